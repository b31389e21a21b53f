#!/usr/bin/env python3
"""Hash-join shapes beyond the TPC-H primary-key join of bench.py, after the reference's own hash-join micro-benchmark
(benchmarks/src/hj.rs:64-: key density 1.0 .. 0.01, probe hit rate 1.0 / 0.1, duplicate build keys, tiny and 100 K-row build
sides against a 60 M-row probe side at SF10) plus what that benchmark does not reach: build sides far beyond every cache
(150 M shuffled keys against 600 M probe rows = the SF100 sizes), duplicate keys at that size (M:N), multi-column and
Decimal128 keys.  Every shape runs under each join-table kind that applies:

  auto    — rank map / ArrayMap when the reference's gating (widened, include/dfgpu.h) allows, else the chained table
  chained — JoinHashMap-style chained table in HBM (table_mode 1)
  radix   — LDS-staged radix-partitioned join (table_mode 4)

with probe_mode 4 ("no ancestor needs the probe order", the hj.rs queries feed an aggregate-free projection whose order is
unobserved).  Output = the probe key column (hj.rs: SELECT l.k), so algorithmic bytes = nb*W + np*W + M*W (key columns read
once + output written once; partition passes, tables, pairs and gathers are overhead).  One JSON line per (shape, table)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HBM_PEAK_GBS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--only", default="")
    ap.add_argument("--big", type=int, default=1, help="include the SF100-sized shapes (150 M build / 600 M probe rows)")
    ap.add_argument("--md", default="")
    ap.add_argument("--tables", default="", help="comma-separated subset of auto,chained,radix,array_map")
    ap.add_argument("--launches", default="", help="comma-separated profile scopes whose launches are listed one by one (ms, last iteration)")
    args = ap.parse_args()
    only = set(x for x in args.only.split(",") if x)

    import ctypes as C

    import torch

    from datafusion_amd import _lib, ops
    from datafusion_amd._lib import Field
    from datafusion_amd.exchange import _as_tensor
    from datafusion_amd.table import DECIMAL128, INT32, INT64, DeviceTable
    lib = _lib.init(0)
    torch.cuda.set_device(0)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(0xDF55)

    def table(cols):
        """{name: int64 tensor | (int64 tensor, 'i32' | 'd128')} -> DeviceTable (library memory, filled through zero-copy tensor views)"""
        names = list(cols)
        kinds = [cols[n][1] if isinstance(cols[n], tuple) else "i64" for n in names]
        tens = [cols[n][0] if isinstance(cols[n], tuple) else cols[n] for n in names]
        fields = (Field * len(names))(*[Field({"i64": INT64, "i32": INT32, "d128": DECIMAL128}[k], 15 if k == "d128" else 0, 2 if k == "d128" else 0, 0) for k in kinds])
        cnames = (C.c_char_p * len(names))(*[n.encode() for n in names])
        out = C.c_void_p()
        _lib.check(lib.dfgpu_table_alloc(len(names), fields, cnames, C.c_int64(int(tens[0].numel())), C.byref(out)))
        t = DeviceTable(out)
        for i, (k, x) in enumerate(zip(kinds, tens)):
            v = t.column_view(i)
            n = x.numel()
            if k == "i64":
                _as_tensor(v.data, n * 8).view(torch.int64).copy_(x)
            elif k == "i32":
                _as_tensor(v.data, n * 4).view(torch.int32).copy_(x.to(torch.int32))
            else:  # Decimal128: low word = value, high word = sign extension
                d = _as_tensor(v.data, n * 16).view(torch.int64).view(n, 2)
                d[:, 0].copy_(x)
                d[:, 1].copy_(x >> 63)
        torch.cuda.synchronize()
        return t

    def randint(lo, hi, n):
        return torch.randint(lo, hi, (n,), generator=gen, device="cuda", dtype=torch.int64)

    results = []

    def run_shape(name, build, probe, on, modes, w, note="", build_out=(), probe_out=None, row_bytes=None):
        """row_bytes = (build row, probe row, output row) bytes of the referenced columns when the output carries payload"""
        if only and not any(o in name for o in only):
            return
        nb, np_ = build.num_rows, probe.num_rows
        for label, opts in modes:
            if args.tables and label not in args.tables.split(","):
                continue

            opts = dict(opts)
            env = opts.pop("env", {})
            os.environ.update(env)

            def step():
                ht = ops.JoinHashTable(build, [l for l, _ in on], probe_mode=4, **opts)
                out = ht.probe(probe, [r for _, r in on], "Inner", list(build_out), [on[0][1]] if probe_out is None else list(probe_out))
                n = out.num_rows
                kind = ht.info().table_kind
                out.free()
                ht.free()
                return n, kind
            try:
                step()
            except _lib.DfgpuError as e:
                print(json.dumps({"case": name, "table": label, "error": str(e)[:200]}), flush=True)
                for k in env:
                    os.environ.pop(k, None)
                continue
            ops.sync()
            ops.profile_enable(True)
            ops.profile_reset()
            times = []
            for _ in range(args.iters):
                ops.sync()
                t0 = time.perf_counter()
                m, kind = step()
                ops.sync()
                times.append(time.perf_counter() - t0)
            stats = ops.profile_stats()
            launches = {}
            for scope in [x for x in args.launches.split(",") if x]:
                ls = ops.profile_launches(scope)
                per = len(ls) // args.iters if args.iters else 0
                launches[scope] = [round(ms, 3) for ms, _ in ls[-per:]] if per else []
            ops.profile_enable(False)
            best = min(times)
            b = (nb + np_ + m) * w if row_bytes is None else nb * row_bytes[0] + np_ * row_bytes[1] + m * row_bytes[2]
            kern = {k: round(v["total_ms"] / args.iters, 3) for k, v in sorted(stats.items(), key=lambda kv: -kv[1]["total_ms"])[:8]}
            rec = {"case": name, "table": label, "table_kind": {0: "chained", 1: "array_map", 2: "rank_map", 3: "radix_lds", 4: "flat8", 5: "flat16"}[kind], "build_rows": nb, "probe_rows": np_,
                   "output_rows": m, "ms": round(best * 1e3, 3), "rows_per_s": (nb + np_) / best, "algorithmic_bytes": b,
                   "algorithmic_gb_per_s": round(b / best / 1e9, 1), "hbm_frac": round(b / best / 1e9 / HBM_PEAK_GBS, 4), "kernel_ms_per_iter": kern, "note": note, **({"launches_ms": launches} if launches else {})}
            results.append(rec)
            print(json.dumps(rec), flush=True)
            for k in env:
                os.environ.pop(k, None)

    ALL = [("auto", {}), ("chained", {"table_mode": 1}), ("radix", {"table_mode": 4}), ("flat", {"table_mode": 5})]
    # ---- hj.rs shapes at SF10: supplier-sized build side (100 K keys) x lineitem-sized probe side (60 M rows)
    nb, np_ = 100_000, 59_986_052
    for mult in (1, 2, 5, 10, 100):
        for hit in (1.0, 0.1):
            bk = (torch.arange(1, nb + 1, device="cuda", dtype=torch.int64) * mult)[torch.randperm(nb, generator=gen, device="cuda")]
            pk = randint(1, nb + 1, np_) * mult
            if hit < 1.0:
                miss = torch.rand(np_, generator=gen, device="cuda") >= hit
                pk = torch.where(miss, pk + nb * mult + 1_000_000, pk)
            b, p = table({"k": bk}), table({"k2": pk})
            run_shape(f"hj density {1 / mult:g} hit {hit:g} build 100K probe 60M", b, p, [("k", "k2")], ALL, 8)
            b.free()
            p.free()
    # duplicates on the build side (hj.rs: "100K_(20%_dups)")
    bk = randint(1, 80_001, nb) * 5
    pk = randint(1, 80_001, np_) * 5
    miss = torch.rand(np_, generator=gen, device="cuda") >= 0.1
    pk = torch.where(miss, pk + 10_000_000, pk)
    b, p = table({"k": bk}), table({"k2": pk})
    run_shape("hj density 0.2 hit 0.1 build 100K with duplicate keys probe 60M", b, p, [("k", "k2")], ALL, 8)
    b.free()
    p.free()
    # multi-column key (Int32, Int64) and a Decimal128 key: no direct-address table applies
    a, c = randint(0, 1000, 1_000_000), randint(0, 1000, 1_000_000)
    pa_, pc = randint(0, 1000, np_), randint(0, 1000, np_)
    b, p = table({"a": (a, "i32"), "c": c}), table({"a2": (pa_, "i32"), "c2": pc})
    run_shape("two-column key (Int32, Int64) build 1M (duplicates) probe 60M", b, p, [("a", "a2"), ("c", "c2")], ALL, 12)
    b.free()
    p.free()
    b, p = table({"k": (torch.randperm(1_000_000, generator=gen, device="cuda") * 7, "d128")}), table({"k2": (randint(0, 1_000_000, np_) * 7, "d128")})
    run_shape("Decimal128 key build 1M probe 60M", b, p, [("k", "k2")], ALL, 16)
    b.free()
    p.free()
    if args.big:
        # ---- SF100 sizes: build side beyond every cache, rows in random order
        nb, np_ = 150_000_000, 600_000_000
        perm = torch.randperm(nb, generator=gen, device="cuda")
        bk = (perm // 8) * 32 + perm % 8 + 1                    # TPC-H's sparse order keys, shuffled
        fk = randint(0, nb, np_)
        pk = (fk // 8) * 32 + fk % 8 + 1
        del perm, fk
        b, p = table({"k": bk}), table({"k2": pk})
        del bk, pk
        torch.cuda.empty_cache()
        run_shape("SF100 sizes: 150M unique shuffled build keys, 600M random foreign keys", b, p, [("k", "k2")], ALL + [("array_map", {"table_mode": 2})], 8,
                  note="auto = rank map + permutation (the build keys are not in ascending row order)")
        b.free()
        p.free()
        # the same keys WITH TPC-H Q3's payload (SURVEY 8d config 3 ii: 16 B build rows, 40 B probe rows, 48 B output rows = 55.2 GB):
        # what a join of tables that are not clustered on the key looks like
        perm = torch.randperm(nb, generator=gen, device="cuda")
        bk = (perm // 8) * 32 + perm % 8 + 1
        fk = randint(0, nb, np_)
        pk = (fk // 8) * 32 + fk % 8 + 1
        del fk
        b = table({"k": bk, "o_orderdate": (perm % 2406 + 8035, "i32"), "o_shippriority": (perm * 0, "i32")})
        del perm, bk
        p = table({"k2": pk, "l_extendedprice": (randint(90000, 10_000_000, np_), "d128"), "l_discount": (randint(0, 11, np_), "d128")})
        del pk
        torch.cuda.empty_cache()
        run_shape("SF100 sizes with the Q3 payload: 150M unique shuffled build rows x 600M random foreign keys", b, p, [("k", "k2")],
                  [("auto", {}), ("array_map", {"table_mode": 2})], 8,
                  build_out=["o_orderdate", "o_shippriority"], probe_out=["k2", "l_extendedprice", "l_discount"], row_bytes=(16, 40, 48))
        b.free()
        p.free()
        bk = randint(0, 50_000_000, nb) * 3                      # every key ~3 times on the build side: M:N, ~3 matches per hit
        pk = randint(0, 100_000_000, 200_000_000) * 3            # half of the probe keys exist
        b, p = table({"k": bk}), table({"k2": pk})
        del bk, pk
        torch.cuda.empty_cache()
        run_shape("SF100 sizes: 150M build rows with duplicate keys (x3), 200M probe rows, hit 0.5 (M:N)", b, p, [("k", "k2")], ALL, 8)
        b.free()
        p.free()
    if args.md:
        with open(args.md, "w") as f:
            f.write("| shape | table | kind | output rows | ms (best) | G rows/s | algorithmic GB/s | % of 8 TB/s | top kernels (ms/iter) |\n|---|---|---|---:|---:|---:|---:|---:|---|\n")
            for r in results:
                top = ", ".join(f"{k} {v}" for k, v in list(r["kernel_ms_per_iter"].items())[:6])
                f.write(f"| {r['case']} | {r['table']} | {r['table_kind']} | {r['output_rows']} | {r['ms']} | {r['rows_per_s'] / 1e9:.2f} | {r['algorithmic_gb_per_s']} | "
                        f"{100 * r['hbm_frac']:.1f} | {top} |\n")


if __name__ == "__main__":
    main()
