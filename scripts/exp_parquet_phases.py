#!/usr/bin/env python3
"""Per-kernel time of one cold scan of an SF-sized lineitem Parquet file (the columns of scripts/bench_parquet.py), device path and host path:
python scripts/exp_parquet_phases.py [--sf 1] [--codec snappy] [--threads 16]"""
import argparse, json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sf", type=float, default=1.0)
    ap.add_argument("--codec", default="snappy")
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--iters", type=int, default=12, help="cold scans per line (the first few warm the pools up: best and median are printed)")
    ap.add_argument("--all-only", action="store_true", help="only the all-columns line of each path")
    args = ap.parse_args()
    import pyarrow as pa, pyarrow.parquet as pq
    from datafusion_amd import _lib, ops, tpch
    from datafusion_amd import parquet as P
    cols = ["l_orderkey", "l_quantity", "l_extendedprice", "l_discount", "l_tax", "l_returnflag", "l_linestatus", "l_shipdate"]
    l = tpch.lineitem(args.sf).select(cols)
    for name in ("l_returnflag", "l_linestatus"):
        codes = l.column(name).to_numpy()
        l = l.set_column(l.column_names.index(name), name, pa.array(codes.view("S1").astype("U1").astype(object), pa.string()))
    _lib.init(0)
    path = os.path.join(tempfile.mkdtemp(), "l.parquet")
    pq.write_table(l, path, compression=args.codec)
    md = pq.ParquetFile(path).metadata
    pages = {c: 0 for c in cols}
    P.CACHE = P.ChunkCache(budget=0)
    f = P.ParquetFile(path)
    for dev in ("1", "0"):
        ops.set_options(parquet__device_decode=dev)
        for single in ((None,) if args.all_only else (cols, None)):
            for c in (single or [None]):
                use = [c] if c else cols
                best, cpu, all_ms = None, None, []
                for _ in range(args.iters):
                    ops.sync()
                    c0 = time.process_time()
                    t0 = time.perf_counter()
                    tab = f.read(use, threads=args.threads)
                    ops.sync()
                    dt = time.perf_counter() - t0
                    dc = time.process_time() - c0
                    all_ms.append(dt * 1e3)
                    if best is None or dt < best:
                        best, cpu = dt, dc
                    tab.free()
                ops.profile_enable(True)
                ops.profile_reset()
                tab = f.read(use, threads=args.threads)
                ops.sync()
                st = ops.profile_stats()
                ops.profile_enable(False)
                tab.free()
                print(json.dumps({"device_decode": dev, "columns": c or "all", "ms": round(best * 1e3, 2), "ms_median": round(sorted(all_ms)[len(all_ms) // 2], 2), "host_cpu_ms": round(cpu * 1e3, 2),
                                  "kernels_ms": {k: round(v["total_ms"], 3) for k, v in st.items() if k.startswith("parquet")},
                                  "calls": {k: v["calls"] for k, v in st.items() if k.startswith("parquet")}}), flush=True)


if __name__ == "__main__":
    main()
