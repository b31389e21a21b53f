#!/usr/bin/env python3
"""Per-operator measurements for the SURVEY.md §8(d) configs that bench.py's single JSON line
does not cover (bench.py = config 3, the Q3 hash join):

  config 2  FilterExec + projection, lineitem SF10, three selectivities, Q3 and Q1 projections
  config 4  TPC-H Q1 (filter -> projection -> grouped aggregate -> sort), SF100, one GPU
  config 4' high-cardinality GROUP BY l_orderkey (150 M groups at SF100)
  config 5  TPC-H Q3 end to end (the reference's physical plan), one GPU
  K10       RepartitionExec(Hash) of the Q3 lineitem projection into 8 partitions
  K11/K12   SortExec of orders by (o_orderdate, o_orderkey) and TopK(10)

For every case: rows/s over the input rows, algorithmic GB/s (referenced input columns read once
+ output written once, per §8d) over the wall time of the device region (inputs resident in HBM,
stream drained on both sides), fraction of the 8.0 TB/s HBM peak, and the per-kernel HIP-event
breakdown recorded by the library.  One JSON object per line on stdout; `--md FILE` also writes
a markdown table (copied into profiles/ by the round's profiling script).

  python scripts/bench_ops.py [--sf 100] [--filter-sf 10] [--iters 3] [--only q1,q3,...]
"""
import argparse
import datetime
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HBM_PEAK_GBS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sf", type=float, default=100.0)
    ap.add_argument("--filter-sf", type=float, default=10.0)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--only", default="")
    ap.add_argument("--md", default="")
    args = ap.parse_args()
    only = set(x for x in args.only.split(",") if x)

    import pyarrow as pa

    from datafusion_amd import _lib, ops, queries
    from datafusion_amd.expr import col, lit
    _lib.init(0)
    results = []

    def want(name):
        return not only or name in only

    def measure(name, fn, rows, alg_bytes, note=""):
        """fn() -> object with .free() or list of them (freed outside the timed region)"""
        def free(o):
            for x in (o if isinstance(o, (list, tuple)) else [o]):
                if x is not None and hasattr(x, "free"):
                    x.free()
        free(fn())  # warm-up (also grows the memory pool)
        ops.sync()
        ops.profile_enable(True)
        ops.profile_reset()
        times = []
        for _ in range(args.iters):
            ops.sync()
            t0 = time.perf_counter()
            o = fn()
            ops.sync()
            times.append(time.perf_counter() - t0)
            free(o)
        stats = ops.profile_stats()
        ops.profile_enable(False)
        best = min(times)
        b = alg_bytes() if callable(alg_bytes) else alg_bytes
        kern = {k: round(v["total_ms"] / args.iters, 3) for k, v in sorted(stats.items(), key=lambda kv: -kv[1]["total_ms"])}
        rec = {"case": name, "rows": rows, "ms": round(best * 1e3, 3), "ms_median": round(sorted(times)[len(times) // 2] * 1e3, 3),
               "rows_per_s": rows / best, "algorithmic_bytes": b, "algorithmic_gb_per_s": round(b / best / 1e9, 1),
               "hbm_frac": round(b / best / 1e9 / HBM_PEAK_GBS, 4), "kernel_ms_per_iter": kern, "note": note}
        results.append(rec)
        print(json.dumps(rec), flush=True)

    # ------------------------------------------------------------------ config 2: filter
    if want("filter"):
        li = ops.tpch_lineitem(args.filter_sf)
        n = li.num_rows
        q3p = ["l_orderkey", "l_extendedprice", "l_discount"]                                             # 40 B/row
        q1p = ["l_quantity", "l_extendedprice", "l_discount", "l_tax", "l_returnflag", "l_linestatus"]    # 66 B/row
        d = lambda y, m, dd: lit(datetime.date(y, m, dd), pa.date32())
        preds = [("shipdate<=1998-09-02", col("l_shipdate") <= d(1998, 9, 2)), ("shipdate>1995-03-15", col("l_shipdate") > d(1995, 3, 15)),
                 ("shipdate in 1994", (col("l_shipdate") >= d(1994, 1, 1)).and_(col("l_shipdate") <= d(1994, 12, 31)))]
        for pname, pred in preds:
            for projname, proj, w in (("q3proj", q3p, 40), ("q1proj", q1p, 66)):
                holder = {}

                plan = ops.FilterPlan(li, pred, proj)   # the node is planned once and executed per iteration, as an ExecutionPlan is

                def run(plan=plan):
                    o = plan.execute(li)
                    holder["n"] = o.num_rows
                    return o
                measure(f"filter[{pname},{projname}] SF{args.filter_sf:g}", run, n, lambda w=w: n * 4 + n * w + holder["n"] * w,
                        note="bytes = N*4 (predicate col) + N*W (projected cols read) + sel*N*W (written)")
                results[-1]["selectivity"] = round(holder["n"] / n, 4)
        li.free()
        ops.sync()

    # ------------------------------------------------------------------ config 4: Q1
    if want("q1") or want("agg_highcard") or want("agg_mediumcard") or want("agg_multikey") or want("partition") or want("filter"):
        li = ops.tpch_lineitem(args.sf)
        n = li.num_rows
        if want("filter"):
            # the same FilterExec at the scale of configs 3-5 (config 2 at SF10 is a 1 ms operator: its fixed costs are a tenth of it)
            q3p = ["l_orderkey", "l_extendedprice", "l_discount"]
            holder = {}

            big_plan = ops.FilterPlan(li, col("l_shipdate") > lit(datetime.date(1995, 3, 15), pa.date32()), q3p)

            def run_big():
                o = big_plan.execute(li)
                holder["n"] = o.num_rows
                return o
            measure(f"filter[shipdate>1995-03-15,q3proj] SF{args.sf:g}", run_big, n, lambda: n * 4 + n * 40 + holder["n"] * 40,
                    note="bytes = N*4 (predicate col) + N*W (projected cols read) + sel*N*W (written)")
        if want("q1"):
            measure(f"Q1 SF{args.sf:g} fused node (filter+project+aggregate in one pass, 8 aggs/4 groups) + sort", lambda: queries.q1(li), n, n * 70,
                    note="bytes = 7 referenced columns: l_shipdate 4 + 4 x Decimal128 64 + 2 x u8 flags = 70 B/row; output 4 rows")
            measure(f"Q1 SF{args.sf:g} operator by operator (FilterExec, ProjectionExec, AggregateExec, SortExec)", lambda: queries.q1(li, fused=False), n, n * 70,
                    note="same bytes; the filter's and the projection's outputs are materialised")
            holder = {}
            f = ops.filter(li, col("l_shipdate") <= lit(queries.DATE_Q1, pa.date32()),
                           ["l_extendedprice", "l_discount", "l_quantity", "l_tax", "l_returnflag", "l_linestatus"])
            p = ops.project(f, [(col("l_extendedprice") * (queries.ONE - col("l_discount")), "__common_expr_1"), (col("l_quantity"), "l_quantity"),
                                (col("l_extendedprice"), "l_extendedprice"), (col("l_discount"), "l_discount"), (col("l_tax"), "l_tax"),
                                (col("l_returnflag"), "l_returnflag"), (col("l_linestatus"), "l_linestatus")])
            f.free()
            m = p.num_rows
            measure(f"Q1 AggregateExec only SF{args.sf:g}", lambda: ops.aggregate(p, queries.Q1_GROUP_BY, queries.q1_aggs(), "Single"), m, m * (16 * 5 + 2),
                    note="input = projected table (5 Decimal128 + 2 u8 = 82 B/row)")
            p.free()
        if want("agg_highcard"):
            t = li.select(["l_orderkey", "l_extendedprice"])
            holder = {}

            def run():
                o = ops.aggregate(t, [(col("l_orderkey"), "l_orderkey")], [("sum", col("l_extendedprice"), "s")], "Single")
                holder["g"] = o.num_rows
                return o
            measure(f"GROUP BY l_orderkey SUM(l_extendedprice) SF{args.sf:g}", run, n, lambda: n * 24 + holder["g"] * 24,
                    note="bytes = N*(8+16) in + groups*(8+16) out")
            results[-1]["groups"] = holder["g"]
            t.free()
        if want("agg_mediumcard"):
            # Q13's shape: orders per customer — 150 M rows, 10 M groups over a key range of 15 M values at SF100 (two-level partitioned
            # LDS aggregation: aggregate.hip dense_accumulate_partitioned)
            oc = ops.tpch_orders(args.sf).select(["o_custkey", "o_orderdate"])
            holder = {}

            def run_mc():
                o = ops.aggregate(oc, [(col("o_custkey"), "o_custkey")], [("count", None, "n"), ("min", col("o_orderdate"), "first_order")], "Single")
                holder["g"] = o.num_rows
                return o
            measure(f"GROUP BY o_custkey COUNT(*), MIN(o_orderdate) SF{args.sf:g}", run_mc, oc.num_rows, lambda: oc.num_rows * 12 + holder["g"] * 20,
                    note="bytes = N*(8+4) in + groups*(8+8+4) out")
            results[-1]["groups"] = holder["g"]
            oc.free()
            # the same key with an 8-byte argument: every customer's latest order (the move carries key, o_orderkey and row number: 16-byte records)
            ok = ops.tpch_orders(args.sf).select(["o_custkey", "o_orderkey"])

            def run_mc8():
                o = ops.aggregate(ok, [(col("o_custkey"), "o_custkey")], [("count", None, "n"), ("max", col("o_orderkey"), "last_order")], "Single")
                holder["g8"] = o.num_rows
                return o
            measure(f"GROUP BY o_custkey COUNT(*), MAX(o_orderkey) SF{args.sf:g}", run_mc8, ok.num_rows, lambda: ok.num_rows * 16 + holder["g8"] * 24,
                    note="bytes = N*(8+8) in + groups*(8+8+8) out")
            results[-1]["groups"] = holder["g8"]
            ok.free()
        if want("agg_multikey"):
            # a daily report: three key columns (hash-interned groups: ~10 K), three aggregates over 600 M rows
            t = li.select(["l_returnflag", "l_linestatus", "l_shipdate", "l_quantity", "l_extendedprice"])
            holder = {}

            def run_mk():
                o = ops.aggregate(t, [(col("l_returnflag"), "l_returnflag"), (col("l_linestatus"), "l_linestatus"), (col("l_shipdate"), "l_shipdate")],
                                  [("sum", col("l_quantity"), "q"), ("sum", col("l_extendedprice"), "p"), ("count", None, "n")], "Single")
                holder["g"] = o.num_rows
                return o
            measure(f"GROUP BY l_returnflag, l_linestatus, l_shipdate SUM(l_quantity), SUM(l_extendedprice), COUNT(*) SF{args.sf:g}", run_mk, n,
                    lambda: n * 38 + holder["g"] * 46, note="bytes = N*(1+1+4+16+16) in + groups*(6+16+16+8) out")
            results[-1]["groups"] = holder["g"]
            t.free()
        if want("partition"):
            t = li.select(["l_orderkey", "l_extendedprice", "l_discount"])
            measure(f"RepartitionExec Hash(l_orderkey) -> 8 partitions SF{args.sf:g}", lambda: ops.partition(t, ["l_orderkey"], 8), n, 2 * n * 40,
                    note="bytes = 2*N*40 (every column read once, written once)")
            t.free()
        li.free()
        ops.sync()

    # ------------------------------------------------------------------ sort / TopK
    if want("sort"):
        o = ops.tpch_orders(args.sf)
        n = o.num_rows
        keys = [("o_orderdate", False, False), ("o_orderkey", True, False)]
        measure(f"SortExec orders by (o_orderdate, o_orderkey DESC) SF{args.sf:g}", lambda: ops.sort(o, keys), n, 2 * n * 24,
                note="bytes = 2*N*24 (4 columns read once, written once); radix passes are overhead")
        measure(f"TopK(10) orders by (o_orderdate, o_orderkey DESC) SF{args.sf:g}", lambda: ops.sort(o, keys, fetch=10), n, n * 12,
                note="bytes = N*12 (key columns read once)")
        o.free()
        ops.sync()

    # ------------------------------------------------------------------ config 3's other shapes: key-only output, in-query shape
    if want("join"):
        o, li = ops.tpch_orders(args.sf), ops.tpch_lineitem(args.sf)
        ok = o.select(["o_orderkey"])
        lk = li.select(["l_orderkey"])
        nb, np_ = o.num_rows, li.num_rows

        def key_only(mode):
            ht = ops.JoinHashTable(ok, ["o_orderkey"], probe_mode=ops.PROBE_MODES[mode])
            out = ht.probe(lk, ["l_orderkey"], "Inner", [], ["l_orderkey"])
            ht.free()
            return out
        for mode in ("order_not_needed", "auto"):
            measure(f"config 3 (i): INNER join orders x lineitem SF{args.sf:g}, key-only output, probe_mode {mode}", lambda: key_only(mode), nb + np_,
                    nb * 8 + np_ * 8 + np_ * 8, note="bytes = build keys + probe keys + 8 B per output row (the key; SURVEY's (u32, u32) index pair is 8 B as well)")
        # the in-query shape: Q3's filters applied first (build = the semi-join's output, probe = lineitem after l_shipdate > 1995-03-15)
        c = ops.tpch_customer(args.sf)
        d = lit(datetime.date(1995, 3, 15), pa.date32())
        cb = ops.filter(c, col("c_mktsegment").eq(lit(queries.SEGMENT_BUILDING, pa.uint8())), ["c_custkey"])
        of = ops.filter(o, col("o_orderdate") < d, ["o_orderkey", "o_custkey", "o_orderdate", "o_shippriority"])
        h0 = ops.JoinHashTable(cb, ["c_custkey"])
        build = h0.probe(of, ["o_custkey"], "RightSemi", probe_cols=["o_orderkey", "o_orderdate", "o_shippriority"])
        h0.free()
        probe = ops.filter(li, col("l_shipdate") > d, ["l_orderkey", "l_extendedprice", "l_discount"])
        nb2, np2 = build.num_rows, probe.num_rows
        outn = {}

        def in_query():
            ht = ops.JoinHashTable(build, ["o_orderkey"], probe_mode=ops.PROBE_MODES["order_not_needed"])
            out = ht.probe(probe, ["l_orderkey"], "Inner", ["o_orderdate", "o_shippriority"], ["l_orderkey", "l_extendedprice", "l_discount"])
            ht.free()
            outn["n"] = out.num_rows
            return out
        in_query().free()
        measure(f"config 3 in-query shape: {nb2} build rows (after the semi-join) x {np2} probe rows (after the shipdate filter), Q3 payload", in_query,
                nb2 + np2, lambda: nb2 * 16 + np2 * 40 + outn["n"] * 48, note="bytes = build 16 B/row + probe 40 B/row + output 48 B/row")
        for t in (c, cb, of, build, probe, ok, lk, o, li):
            t.free()
        ops.sync()

    # ------------------------------------------------------------------ strings in HBM (DFGPU_UTF8)
    if want("strings"):
        import numpy as np

        from datafusion_amd.table import DeviceTable
        rng = np.random.default_rng(0)
        n_s, distinct = 30_000_000, 150_000                        # c_name of TPC-H SF1 is 150 K distinct 18-byte strings
        pool = np.array([f"Customer#{i:09d}" for i in range(distinct)])
        host = pa.table({"name": pa.array(pool[rng.integers(0, distinct, size=n_s)], pa.string()), "v": pa.array(rng.integers(0, 1000, size=n_s))})
        sdev = DeviceTable.from_arrow(host)
        sbytes = host.column("name").nbytes
        measure(f"dictionary_encode (device interning) of {n_s // 10**6} M strings, {distinct // 1000} K distinct, 18 B each", lambda: sdev.dictionary_encode(["name"], sorted=True),
                n_s, sbytes + n_s * 4, note="bytes = string bytes + 64-bit offsets read, Int32 indices written (dictionary sort on the host included)")
        measure("filter name LIKE 'Customer#00001%' on the bytes", lambda: ops.filter(sdev, col("name").like("Customer#00001%"), ["v"]), n_s, sbytes + n_s // 8,
                note="bytes = string bytes + offsets read, mask written")
        measure("filter name = literal on the bytes", lambda: ops.filter(sdev, col("name").eq(lit("Customer#000012345", pa.string())), ["v"]), n_s, sbytes + n_s // 8)
        measure("filter v < 500 moving the string column (take of strings)", lambda: ops.filter(sdev, col("v") < lit(500, pa.int64())), n_s, n_s * 8 + 3 * sbytes // 2,
                note="bytes = predicate column + strings read once, half of them written")
        enc = sdev.dictionary_encode(["name"])
        measure("GROUP BY name (interned) SUM(v)", lambda: ops.aggregate(enc, [(col("name"), "name")], [("sum", col("v"), "s")], "Single"), n_s, n_s * 12)
        enc.free()
        sdev.free()
        ops.sync()

    # ------------------------------------------------------------------ boundary: host RecordBatch <-> device table
    if want("import"):
        from datafusion_amd import tpch
        from datafusion_amd.table import DeviceTable
        host = tpch.lineitem(2.0).select(["l_orderkey", "l_extendedprice", "l_discount", "l_shipdate"])     # 12 M rows, 44 B/row
        nbytes = host.nbytes
        measure("dfgpu_table_import (Arrow C Data Interface -> HBM: hipHostRegister + async copy), 12 M lineitem rows x 44 B",
                lambda: DeviceTable.from_arrow(host), host.num_rows, nbytes, note="bytes = host buffer bytes; PCIe Gen5 x16 = 63 GB/s")
        dev = DeviceTable.from_arrow(host)
        holder = {}

        def export():
            holder["t"] = dev.to_arrow()
            return None
        measure("dfgpu_table_export (HBM -> Arrow C Data Interface), same table", export, host.num_rows, nbytes, note="bytes = host buffer bytes")
        dev.free()

    # ------------------------------------------------------------------ config 5 (one GPU): Q3
    if want("q3"):
        c, o, li = ops.tpch_customer(args.sf), ops.tpch_orders(args.sf), ops.tpch_lineitem(args.sf)
        li4 = li.select(["l_orderkey", "l_extendedprice", "l_discount", "l_shipdate"])
        li.free()
        stats = {}
        queries.q3(c, o, li4, stats=stats, fused=False).free()
        nc, no, nl = c.num_rows, o.num_rows, li4.num_rows
        # §8d config 5: sum over operators of referenced input column bytes + output bytes
        b = (nc * (8 + 1) + stats["customer_filtered"] * 8                      # customer filter
             + no * 24 + stats["orders_filtered"] * 24                          # orders filter (4 cols)
             + stats["customer_filtered"] * 8 + stats["orders_filtered"] * 24 + stats["semi_join"] * 16   # semi join
             + nl * 44 + stats["lineitem_filtered"] * 40                        # lineitem filter
             + stats["semi_join"] * 16 + stats["lineitem_filtered"] * 40 + stats["join"] * 48             # inner join
             + stats["join"] * 48 + stats["groups"] * 32                        # aggregate
             + stats["groups"] * 20)                                            # top-k keys
        measure(f"Q3 SF{args.sf:g} end to end, 1 GPU, FilterExecs fused into the probe sides", lambda: queries.q3(c, o, li4), nc + no + nl, b,
                note="bytes = sum over the reference plan's operators of referenced input columns + outputs (intermediate counts below)")
        results[-1]["intermediate_rows"] = stats
        measure(f"Q3 SF{args.sf:g} end to end, 1 GPU, operator by operator", lambda: queries.q3(c, o, li4, fused=False), nc + no + nl, b,
                note="same plan, FilterExec outputs materialised")
        for t in (c, o, li4):
            t.free()

    if args.md:
        with open(args.md, "w") as f:
            f.write("| case | input rows | ms (best) | G rows/s | algorithmic GB | GB/s | % of 8 TB/s | top kernels (ms/iter) |\n|---|---:|---:|---:|---:|---:|---:|---|\n")
            for r in results:
                top = ", ".join(f"{k} {v}" for k, v in list(r["kernel_ms_per_iter"].items())[:5])
                f.write(f"| {r['case']} | {r['rows']} | {r['ms']} | {r['rows_per_s'] / 1e9:.2f} | {r['algorithmic_bytes'] / 1e9:.2f} | "
                        f"{r['algorithmic_gb_per_s']} | {100 * r['hbm_frac']:.1f} | {top} |\n")


if __name__ == "__main__":
    main()
