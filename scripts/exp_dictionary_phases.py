"""Where does dictionary_encode's host time go — alone, and inside a process that has held SF100-sized tables (profiles/r3_strings.md)?
DFGPU_TRACE=dict makes the library print its phases."""
import os
import sys
import time

import numpy as np
import pyarrow as pa

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["DFGPU_TRACE"] = "dict"
from datafusion_amd import ops  # noqa: E402
from datafusion_amd.table import DeviceTable  # noqa: E402

n, distinct = 30_000_000, 150_000
rng = np.random.default_rng(3)
names = pa.array([f"Customer#{i:09d}" for i in range(distinct)])
t = DeviceTable.from_arrow(pa.table({"name": names.take(pa.array(rng.integers(0, distinct, n)))}))
ops.sync()


def run(label):
    for i in range(3):
        print(f"--- {label}, call {i}", file=sys.stderr, flush=True)
        t0 = time.perf_counter()
        e = t.dictionary_encode(["name"], sorted=True)
        ops.sync()
        print(f"--- total {1e3 * (time.perf_counter() - t0):.2f} ms", file=sys.stderr, flush=True)
        e.free()


run("fresh process")
# exactly what bench_ops.py's strings block holds: the host table stays alive, a second column
pool = np.array([f"Customer#{i:09d}" for i in range(distinct)])
rng0 = np.random.default_rng(0)
host = pa.table({"name": pa.array(pool[rng0.integers(0, distinct, size=n)], pa.string()), "v": pa.array(rng0.integers(0, 1000, size=n))})
t_keep, t = t, DeviceTable.from_arrow(host)
run("bench_ops.py's table (host table alive, two columns)")
del host
run("the same after dropping the host table")
t.free()
t = t_keep
run("first table again")
ops.profile_enable(True)
ops.profile_reset()
run("fresh process, per-kernel profiling on (what bench_ops.py's measure() does)")
ops.profile_stats()
ops.profile_enable(False)
run("fresh process, profiling off again")
big = [ops.tpch_lineitem(100.0), ops.tpch_orders(100.0)]        # what bench_ops.py holds before its strings cases
j = ops.hash_join(big[1], big[0], [("o_orderkey", "l_orderkey")], "Inner", build_cols=["o_orderdate"], probe_cols=["l_orderkey", "l_extendedprice"], probe_mode=3)
j.free()
run("with SF100 tables resident")
for b in big:
    b.free()
run("after freeing them (blocks cached in the pool)")
print(ops.mem_stats(), file=sys.stderr)
