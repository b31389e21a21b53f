#!/bin/bash
# usage (on the GPU box, via gpurun): scripts/gpu_r4.sh <tag> [tests] [sf300] [shapes] [ops] [bench] — writes gpurun_out/<tag>/
set -u
TAG=$1; shift
WHAT=${*:-tests sf300 shapes}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
has() { [[ " $WHAT " == *" $1 "* ]]; }
if has tests; then
  (time timeout ${TESTS_TIMEOUT:-1200} python -m pytest tests -m gpu -q --maxfail=${MAXFAIL:-25} -p no:cacheprovider -k "not sf300 ${TESTS_K:-}") > $OUT/pytest.log 2>&1
  tail -40 $OUT/pytest.log
fi
if has sf300; then
  (time timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -s -p no:cacheprovider -k sf300) > $OUT/pytest_sf300.log 2>&1
  tail -15 $OUT/pytest_sf300.log
fi
if has shapes; then
  (time timeout 1200 python scripts/bench_join_shapes.py --md $OUT/join_shapes.md ${SHAPES_ARGS:-}) > $OUT/join_shapes.jsonl 2> $OUT/join_shapes.err
  cat $OUT/join_shapes.md; tail -5 $OUT/join_shapes.err
fi
if has gpbits; then   # the grouped probe's group count: 2^8 .. 2^10 groups on the Q3-payload shape
  for B in 8 9 10; do
    (DFGPU_JOIN_GP_BITS=$B timeout 600 python scripts/bench_join_shapes.py --only payload --tables auto --iters 2) > $OUT/gpbits_$B.jsonl 2> $OUT/gpbits_$B.err
    echo "GP_BITS=$B"; python - <<PY
import json
for ln in open("$OUT/gpbits_$B.jsonl"):
    r = json.loads(ln); print(r.get("ms"), r.get("kernel_ms_per_iter"))
PY
  done
fi
if has aggknobs; then   # round 3's opt-in aggregate paths on the three-key aggregate
  for K in "" "DFGPU_AGG_LDS_CLAIM=1" "DFGPU_AGG_SMALL_TABLE=1" "DFGPU_AGG_LDS_CLAIM=1 DFGPU_AGG_SMALL_TABLE=1"; do
    (env $K timeout 600 python scripts/bench_ops.py --only agg_multikey --iters 3) > "$OUT/agg_${K// /_}.jsonl" 2> $OUT/agg.err
    echo "knobs: [$K]"; python - <<PY
import json
for ln in open("$OUT/agg_${K// /_}.jsonl"):
    r = json.loads(ln); print(r.get("ms"), r.get("kernel_ms_per_iter") or r.get("kernels"))
PY
  done
fi
if has pmc; then   # counters of the shape named by PMC_SHAPE / PMC_TABLES (scripts/profile_sq.sh)
  CMD="python $R/scripts/bench_join_shapes.py --only ${PMC_SHAPE:-payload} --tables ${PMC_TABLES:-auto} --iters 1" scripts/profile_sq.sh $TAG/pmc ${PMC_FILTER:-}
fi
if has ops; then
  (time timeout 1200 python scripts/bench_ops.py --md $OUT/ops.md ${OPS_ARGS:-}) > $OUT/ops.jsonl 2> $OUT/ops.err
  cat $OUT/ops.md; tail -5 $OUT/ops.err
fi
if has bench; then
  (time timeout 600 python bench.py) > $OUT/bench.json 2> $OUT/bench.err
  tail -2 $OUT/bench.json; tail -3 $OUT/bench.err
fi
# (gpvariants: the A/B/C tile shapes of grouped.hip were measured with DFGPU_GP_VARIANT — profiles/r4_group_rows.md — and the losing two
# were removed together with the knob)
