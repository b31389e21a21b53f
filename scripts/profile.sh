#!/bin/bash
# usage: scripts_profile.sh <tag> [bench args...]  — runs on the GPU box (via gpurun)
# writes gpurun_out/<tag>/{bench.json,stats,pmc_fetch,pmc_write}
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
(timeout 900 python bench.py "$@") > $OUT/bench.json 2> $OUT/bench.err
export TMPDIR=/tmp
cd /tmp
(timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats -o trace -- python $R/bench.py "$@" --no-cpu) > $OUT/stats.log 2>&1
(timeout 600 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- python $R/bench.py "$@" --no-cpu --steps 2 --warmup 1) > $OUT/pmc_fetch.log 2>&1
(timeout 600 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- python $R/bench.py "$@" --no-cpu --steps 2 --warmup 1) > $OUT/pmc_write.log 2>&1
# per-operator kernel trace (Q1 fused node, dense-key group-by, sort, Q3): the specialised kernels show up by name
if [ -z "${SKIP_OPS:-}" ]; then
(timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats_ops -o trace -- python $R/scripts/bench_ops.py --only q1,agg_highcard,sort,q3 --iters 3) > $OUT/stats_ops.log 2>&1
fi
cd $R
[ -z "${SKIP_OPS:-}" ] && python - <<PY > $OUT/ops_kernels.md 2>&1
import sqlite3
con = sqlite3.connect("$OUT/stats_ops/trace_results.db")
print("| kernel | calls | avg ms | total ms | % |\n|---|---:|---:|---:|---:|")
for n, c, tot, avg, pct in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc limit 40"):
    print(f"| {n.replace('void ', '').split('(')[0].replace('dfgpu::', '')} | {c} | {avg / 1e3:.3f} | {tot / 1e3:.1f} | {pct:.1f} |")
PY
# keep only the small summaries (gpurun_out merge is capped at 64 MiB)
find $OUT -name '*kernel_trace.csv' -size +8M -delete
find $OUT -name '*counter_collection.csv' -size +8M -exec sh -c 'head -c 8000000 "$1" > "$1.head"; rm "$1"' _ {} \;
ls -laR $OUT | head -60
tail -3 $OUT/bench.json
