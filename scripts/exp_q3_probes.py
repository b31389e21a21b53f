"""Q3's two filtered probes at SF300 (the RightSemi of orders against the BUILDING customers, the Inner of lineitem against its output), each
timed on its own under option variants: per-kernel HIP-event times of 5 runs (experiment harness; `python scripts/exp_q3_probes.py [SF]`)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyarrow as pa
from datafusion_amd import _lib, ops, queries
from datafusion_amd.expr import col, lit
_lib.init(0)
sf = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
c = ops.tpch_customer(sf).select(["c_custkey", "c_mktsegment"])
o = ops.tpch_orders(sf).select(["o_orderkey", "o_custkey", "o_orderdate", "o_shippriority"])
li = ops.tpch_lineitem(sf).select(["l_orderkey", "l_extendedprice", "l_discount", "l_shipdate"])
cf = ops.filter(c, col("c_mktsegment").eq(lit(queries.SEGMENT_BUILDING, pa.uint8())), ["c_custkey"])
pm = ops.PROBE_MODES["single_pass_unordered"]
variants = [dict(), dict(join__counts_nt="0"), dict(join__pred_in_counts="0")] + [dict([kv.split("=")]) for kv in sys.argv[2:]]
for opts in variants:
    ops.reset_options()
    ops.set_options(**opts)
    ht = ops.JoinHashTable(cf, ["c_custkey"], probe_mode=pm)
    semi_fn = lambda: ht.probe(o, ["o_custkey"], "RightSemi", probe_cols=["o_orderkey", "o_orderdate", "o_shippriority"], predicate=col("o_orderdate") < lit(queries.DATE_Q3, pa.date32()))
    semi = semi_fn()
    ht2 = ops.JoinHashTable(semi, ["o_orderkey"], probe_mode=pm)
    inner_fn = lambda: ht2.probe(li, ["l_orderkey"], "Inner", ["o_orderdate", "o_shippriority"], ["l_orderkey", "l_extendedprice", "l_discount"], predicate=col("l_shipdate") > lit(queries.DATE_Q3, pa.date32()))
    for name, fn in (("semi(orders)", semi_fn), ("inner(lineitem)", inner_fn)):
        fn().free(); ops.sync()
        ops.profile_enable(True); ops.profile_reset()
        ts = []
        for _ in range(5):
            ops.sync(); t0 = time.perf_counter(); r = fn(); ops.sync(); ts.append(time.perf_counter() - t0); n = r.num_rows; r.free()
        st = ops.profile_stats(); ops.profile_enable(False)
        print(json.dumps({"opts": opts, "probe": name, "rows_out": n, "ms_min": round(min(ts) * 1e3, 3), "ms_med": round(sorted(ts)[2] * 1e3, 3),
                          "kernels_ms": {k: round(v["total_ms"] / v["calls"], 3) for k, v in sorted(st.items(), key=lambda kv: -kv[1]["total_ms"])[:6]}}), flush=True)
    ht.free(); ht2.free(); semi.free()
