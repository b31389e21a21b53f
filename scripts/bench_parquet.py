#!/usr/bin/env python3
"""Scan -> device measurement: TPC-H-shaped lineitem (the Q1 / Q6 columns, datafusion_amd.tpch — the host mirror of
the device generator; the flag columns as strings) written as Parquet with the codecs the
reference's benchmarks use, decoded (a) by pyarrow's CPU reader on all host cores and (b) by dfgpu_parquet_decode_chunk
(host: page headers + decompression + run headers; device: value decode).  Prints one JSON line per codec.
usage: python scripts/bench_parquet.py [--sf 1] [--threads 1,8]"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sf", type=float, default=1.0)
    ap.add_argument("--threads", default="1,8,16")
    args = ap.parse_args()
    import pyarrow as pa
    import pyarrow.parquet as pq

    from datafusion_amd import _lib, ops
    from datafusion_amd.parquet import ParquetFile
    import numpy as np

    from datafusion_amd import tpch
    cols = ["l_orderkey", "l_quantity", "l_extendedprice", "l_discount", "l_tax", "l_returnflag", "l_linestatus", "l_shipdate"]
    t0 = time.perf_counter()
    l = tpch.lineitem(args.sf).select(cols)
    for name in ("l_returnflag", "l_linestatus"):   # strings, as in the reference's schema (benchmarks/src/tpch/mod.rs:93-122)
        codes = l.column(name).to_numpy()
        l = l.set_column(l.column_names.index(name), name, pa.array(codes.view("S1").astype("U1").astype(object), pa.string()))
    gen_s = time.perf_counter() - t0
    arrow_bytes = sum(l.column(c).nbytes for c in cols)
    _lib.init(0)
    d = tempfile.mkdtemp()
    for codec, kw in (("zstd", dict(compression="zstd", compression_level=1)), ("snappy", dict(compression="snappy")), ("none", dict(compression="none"))):
        path = os.path.join(d, f"lineitem_{codec}.parquet")
        pq.write_table(l, path, **kw)
        size = os.path.getsize(path)
        pa.set_cpu_count(os.cpu_count() or 1)
        best_cpu = None
        for _ in range(2):
            t0 = time.perf_counter()
            ref = pq.read_table(path)
            dt = time.perf_counter() - t0
            best_cpu = dt if best_cpu is None else min(best_cpu, dt)
        line = {"what": "parquet_scan", "codec": codec, "sf": args.sf, "rows": l.num_rows, "file_bytes": size, "arrow_bytes": arrow_bytes,
                "row_groups": pq.ParquetFile(path).metadata.num_row_groups, "pyarrow_read_s": round(best_cpu, 4),
                "pyarrow_rows_per_s": round(l.num_rows / best_cpu), "host_cores": os.cpu_count(), "generate_s": round(gen_s, 2)}
        from datafusion_amd import parquet as P
        P.CACHE.clear()
        P.CACHE = P.ChunkCache(budget=0)          # cold scans below: the device chunk cache is measured separately at the end
        f = ParquetFile(path)
        # host half alone
        t0 = time.perf_counter()
        for g in range(f.num_row_groups):
            for c in cols:
                f.inspect_chunk(g, c)
        line["host_half_s"] = round(time.perf_counter() - t0, 4)
        for nt in [int(x) for x in args.threads.split(",")]:
            best = None
            for _ in range(3):
                ops.sync()
                t0 = time.perf_counter()
                tab = f.read(cols, threads=nt)
                ops.sync()
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            got = tab.to_arrow()
            ok = all((got.column(c).cast(pa.string()) if pa.types.is_dictionary(got.column(c).type) else got.column(c)).equals(ref.column(c)) for c in cols)
            line[f"gpu_read_s_threads{nt}"] = round(best, 4)
            line[f"gpu_rows_per_s_threads{nt}"] = round(l.num_rows / best)
            line[f"gpu_arrow_gb_per_s_threads{nt}"] = round(arrow_bytes / best / 1e9, 2)
            line[f"equal_to_pyarrow_threads{nt}"] = bool(ok)
            tab.free()
        ops.profile_enable(True)
        ops.profile_reset()
        tab = f.read(cols)
        ops.sync()
        st = ops.profile_stats().get("parquet_decode")
        ops.profile_enable(False)
        if st:
            line["decode_kernel"] = {"calls": st["calls"], "total_ms": round(st["total_ms"], 3), "gb_per_s": round(st["bytes"] / (st["total_ms"] * 1e-3) / 1e9, 1)}
        tab.free()
        # repeated scan served from the device chunk cache (datafusion_amd/parquet.py ChunkCache)
        P.CACHE = P.ChunkCache(budget=64 << 30)
        f.read(cols).free()
        ops.sync()
        t0 = time.perf_counter()
        tab = f.read(cols)
        ops.sync()
        line["gpu_cached_rescan_s"] = round(time.perf_counter() - t0, 4)
        line["cache"] = P.CACHE.stats()
        tab.free()
        P.CACHE.clear()
        f.close()
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
