#!/bin/bash
# usage (GPU box): scripts/bench_timeline.sh <tag> <marker kernel substring> <bench.py args...> — kernel-trace timeline of the last step of a bench run
set -u
TAG=$1; MARK=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
(timeout 900 rocprofv3 --kernel-trace -d $OUT/tr_bench -o trace -- python $R/bench.py "$@" --no-cpu --no-also) > $OUT/tr_bench.log 2>&1
DB=$(find $OUT/tr_bench -name '*results.db' | head -1)
python $R/scripts/step_timeline.py $DB "$MARK" > $OUT/timeline_bench.txt 2>&1
rm -rf $OUT/tr_bench
tail -150 $OUT/timeline_bench.txt
