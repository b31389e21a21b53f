#!/usr/bin/env python3
"""One step of a profiled bench run as a timeline: every launch of the LAST step in a rocprofv3 kernel-trace database, with its start, the
gap before it and its duration; kernels vs gaps at the end.  `python scripts/step_timeline.py <trace_results.db> <marker kernel substring>`
(the marker is a kernel that runs once per step, first: k_cmp<unsigned char for Q3, agg_node for Q1, k_rank_setbits for the join)."""
import sqlite3
import sys


def short(n):
    return n.replace("void ", "").split("(")[0].replace("dfgpu::", "")


def main():
    db, marker = sys.argv[1], sys.argv[2]
    brief = len(sys.argv) > 3 and sys.argv[3] == "--brief"
    rows = list(sqlite3.connect(db).execute("select name,start,end,grid_x from kernels order by start"))
    idx = [i for i, r in enumerate(rows) if marker in r[0]]
    a, b = idx[-2], idx[-1]
    t0, prev_end, tot_k, tot_gap = rows[a][1], None, 0.0, 0.0
    for n, s, e, g in rows[a:b]:
        gap = (s - prev_end) / 1e3 if prev_end else 0.0
        if not brief or (e - s) / 1e3 > 20 or gap > 20:
            print(f"{(s - t0) / 1e3:9.1f} us  +gap {gap:7.1f}  dur {(e - s) / 1e3:8.1f}  {short(n)[:80]} grid={g}")
        tot_k += (e - s) / 1e3
        tot_gap += max(gap, 0.0)
        prev_end = e
    print(f"launches {b - a}  kernels {tot_k:.1f} us  gaps {tot_gap:.1f} us  span {(rows[b][1] - t0) / 1e3:.1f} us")


if __name__ == "__main__":
    main()
