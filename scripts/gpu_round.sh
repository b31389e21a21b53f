#!/bin/bash
# usage (on the GPU box, via gpurun): scripts/gpu_round.sh <tag> [tests|bench|ops|all]...
# writes gpurun_out/<tag>/{pytest.log,bench.json,ops.jsonl,ops.md}
set -u
TAG=$1; shift
WHAT=${*:-all}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
has() { [[ " $WHAT " == *" $1 "* || " $WHAT " == *" all "* ]]; }
if has tests; then
  (time timeout 900 python -m pytest tests -m gpu -x -q) > $OUT/pytest.log 2>&1
  tail -5 $OUT/pytest.log
fi
if has bench; then
  (time timeout 600 python bench.py) > $OUT/bench.json 2> $OUT/bench.err
  tail -2 $OUT/bench.json; tail -3 $OUT/bench.err
fi
if has bench2; then
  (timeout 600 python bench.py --probe-mode 1 --no-cpu) > $OUT/bench_twopass.json 2> $OUT/bench_twopass.err
  tail -1 $OUT/bench_twopass.json
fi
if has ops; then
  (time timeout 1200 python scripts/bench_ops.py --md $OUT/ops.md ${OPS_ARGS:-}) > $OUT/ops.jsonl 2> $OUT/ops.err
  cat $OUT/ops.md; tail -5 $OUT/ops.err
fi
