"""Q3 SF100 with the FilterExecs fused into the probe sides: per-kernel times under the probe flavours (experiment harness)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from datafusion_amd import _lib, ops, queries
_lib.init(0)
sf = float(os.environ.get("SF", "100"))
c, o, li = ops.tpch_customer(sf), ops.tpch_orders(sf), ops.tpch_lineitem(sf)
li4 = li.select(["l_orderkey", "l_extendedprice", "l_discount", "l_shipdate"])
li.free()
for label, mode in (("unordered", "single_pass_unordered"), ("placed", "auto")):
    pm = ops.PROBE_MODES[mode]
    from datafusion_amd.expr import lit
    import pyarrow as pa
    fn = lambda: queries._q3_fused_filters(c, o, li4, None, pm, lit(queries.SEGMENT_BUILDING, pa.uint8()))
    fn().free(); ops.sync()
    ops.profile_enable(True); ops.profile_reset()
    ts = []
    for _ in range(3):
        ops.sync(); t0 = time.perf_counter(); r = fn(); ops.sync(); ts.append(time.perf_counter() - t0); r.free()
    st = ops.profile_stats(); ops.profile_enable(False)
    print(label, os.environ.get("TAG", ""), "ms", round(min(ts) * 1e3, 3), {k: round(v["total_ms"] / 3, 3) for k, v in sorted(st.items(), key=lambda kv: -kv[1]["total_ms"])[:8]}, flush=True)
