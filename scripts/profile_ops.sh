#!/bin/bash
# usage (GPU box, via gpurun): scripts/profile_ops.sh <tag>
# HBM traffic per kernel of the per-operator benchmarks: two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) as
# MI355X_MICROARCH.md prescribes, plus a kernel trace for the durations.  Writes gpurun_out/<tag>/ops_traffic.md
set -u
TAG=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CASES=${CASES:-filter,q1,agg_highcard,partition,sort,q3}
(timeout 500 rocprofv3 --kernel-trace --stats -d $OUT/stats -o trace -- python $R/scripts/bench_ops.py --only $CASES --iters 2) > $OUT/stats.log 2>&1
(timeout 500 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- python $R/scripts/bench_ops.py --only $CASES --iters 2) > $OUT/pmc_fetch.log 2>&1
(timeout 500 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- python $R/scripts/bench_ops.py --only $CASES --iters 2) > $OUT/pmc_write.log 2>&1
cd $R
python - <<PY > $OUT/ops_traffic.md 2>&1
import sqlite3
def q(db, sql):
    con = sqlite3.connect(db)
    try:
        return list(con.execute(sql))
    finally:
        con.close()
short = lambda n: n.replace("void ", "").split("(")[0].replace("dfgpu::", "")
kern = {short(r[0]): (r[1], r[2], r[3]) for r in q("$OUT/stats/trace_results.db", "select name,total_calls,total_duration,average from top_kernels")}
# largest launches only: the SF100 cases dominate; take the max over dispatches of each kernel
pm = lambda sub: {short(r[0]): r[1] for r in q("$OUT/" + sub + "/pmc_results.db", "select kernel_name, max(value) from counters_collection group by 1")}
dur = {short(r[0]): r[1] for r in q("$OUT/pmc_fetch/pmc_results.db", "select kernel_name, max(duration) from counters_collection group by 1")}
fetch, write = pm("pmc_fetch"), pm("pmc_write")
print("# HBM traffic of the largest launch of each kernel (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)\n")
print("read bytes = FETCH_SIZE x 1024 x 2 (gfx950 correction calibrated in profiles/r1_sf100_v4.md), written bytes = WRITE_SIZE x 1024; duration = that launch under the FETCH pass\n")
print("| kernel | calls | longest launch ms | read GB | written GB | traffic GB | traffic GB/s |\n|---|---:|---:|---:|---:|---:|---:|")
rows = []
for k, (calls, tot, avg) in kern.items():
    if k not in fetch and k not in write:
        continue
    rd, wr = fetch.get(k, 0) * 1024 * 2 / 1e9, write.get(k, 0) * 1024 / 1e9
    d = dur.get(k, 0) / 1e6
    rows.append((d, k, calls, rd, wr))
for d, k, calls, rd, wr in sorted(rows, reverse=True)[:32]:
    print(f"| {k} | {calls} | {d:.3f} | {rd:.2f} | {wr:.2f} | {rd + wr:.2f} | {(rd + wr) / (d * 1e-3) if d else 0:.0f} |")
PY
find $OUT -name '*.db' -size +16M -delete
cat $OUT/ops_traffic.md | head -45
