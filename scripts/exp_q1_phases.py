#!/usr/bin/env python3
"""Where a TPC-H Q1 step's wall time goes outside its node kernel: times the phases of queries.q1 on one GPU with a
device sync between them — planning (expression lowering, once), node execute (create + update + emit), the 4-row SortExec — next to
the back-to-back step the bench times, and prints the library's own per-kernel HIP-event table for the same steps.
usage: python scripts/exp_q1_phases.py [--sf 100] [--steps 10]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sf", type=float, default=100.0)
    ap.add_argument("--steps", type=int, default=10)
    a = ap.parse_args()
    from datafusion_amd import _lib, ops, queries
    _lib.init(0)
    li = ops.tpch_lineitem(a.sf)
    ops.sync()
    for _ in range(3):
        queries.q1(li).free()
    ops.sync()
    keys = [("l_returnflag", False, False), ("l_linestatus", False, False)]
    ph = {"plan_lookup": 0.0, "node_execute": 0.0, "sort": 0.0, "free": 0.0}
    for _ in range(a.steps):
        t0 = time.perf_counter()
        plan = queries.plan_q1(li)
        t1 = time.perf_counter()
        agg = plan.node.execute(li)
        ops.sync()
        t2 = time.perf_counter()
        out = ops.sort(agg, keys)
        ops.sync()
        t3 = time.perf_counter()
        agg.free()
        out.free()
        t4 = time.perf_counter()
        for k, d in zip(ph, (t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
            ph[k] += d
    ops.sync()
    ops.profile_enable(True)
    ops.profile_reset()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        queries.q1(li).free()
    ops.sync()
    whole = (time.perf_counter() - t0) / a.steps * 1e3
    st = ops.profile_stats()
    ops.profile_enable(False)
    # the same step WITHOUT the event pairs around every kernel (what bench.py's wall clock sees when profiling is off)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        queries.q1(li).free()
    ops.sync()
    bare = (time.perf_counter() - t0) / a.steps * 1e3
    kern = {k: round(v["total_ms"] / a.steps, 4) for k, v in sorted(st.items(), key=lambda kv: -kv[1]["total_ms"])}
    print(json.dumps({"sf": a.sf, "rows": li.num_rows, "phases_ms": {k: round(v / a.steps * 1e3, 4) for k, v in ph.items()},
                      "step_ms_profiled": round(whole, 4), "step_ms_bare": round(bare, 4), "kernel_ms_per_step": kern,
                      "kernel_sum_ms": round(sum(kern.values()), 4)}))


if __name__ == "__main__":
    main()
