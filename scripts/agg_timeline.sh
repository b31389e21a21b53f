#!/bin/bash
# usage (GPU box): scripts/agg_timeline.sh <tag> <bench_ops --only list> <marker kernel substring>  — kernel-trace timeline of the last iteration of one bench_ops case
set -u
TAG=$1; ONLY=$2; MARK=$3
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
(timeout 600 rocprofv3 --kernel-trace -d $OUT/tr_$ONLY -o trace -- python $R/scripts/bench_ops.py --only $ONLY --iters 3) > $OUT/tr_$ONLY.log 2>&1
DB=$(find $OUT/tr_$ONLY -name '*results.db' | head -1)
python $R/scripts/step_timeline.py $DB "$MARK" > $OUT/timeline_$ONLY.txt 2>&1
rm -rf $OUT/tr_$ONLY
tail -80 $OUT/timeline_$ONLY.txt
