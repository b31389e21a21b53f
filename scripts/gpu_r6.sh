#!/bin/bash
# usage (on the GPU box, via gpurun): scripts/gpu_r6.sh <tag> <what...> — writes gpurun_out/<tag>/
set -u
TAG=$1; shift
WHAT=${*:-q1phases}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
has() { [[ " $WHAT " == *" $1 "* ]]; }
if has smoke; then
  (time timeout 300 python __graft_entry__.py smoke) > $OUT/smoke.log 2>&1; tail -3 $OUT/smoke.log
fi
if has quicktests; then
  (time timeout ${TESTS_TIMEOUT:-600} python -m pytest ${QUICK_TESTS:-tests/test_gpu_queries.py tests/test_gpu_fused.py tests/test_gpu_sort_partition.py tests/test_gpu_filter.py} -m gpu -q -x -p no:cacheprovider ${TESTS_K:+-k "$TESTS_K"}) > $OUT/quicktests.log 2>&1
  tail -15 $OUT/quicktests.log
fi
if has tests; then
  (time timeout ${TESTS_TIMEOUT:-1200} python -m pytest tests -m gpu -q --maxfail=${MAXFAIL:-25} -p no:cacheprovider -n ${TESTS_N:-0} ${TESTS_K:+-k "$TESTS_K"}) > $OUT/pytest.log 2>&1
  tail -30 $OUT/pytest.log
fi
if has q1phases; then
  (timeout 300 python scripts/exp_q1_phases.py --sf 100 --steps 10) > $OUT/q1_phases.json 2> $OUT/q1_phases.err
  cat $OUT/q1_phases.json; tail -3 $OUT/q1_phases.err
fi
if has q1trace; then
  (cd /tmp && timeout 600 rocprofv3 --hip-trace --kernel-trace --stats -d $OUT/q1trace -o t -- python $R/scripts/exp_q1_phases.py --sf 100 --steps 5) > $OUT/q1trace.log 2>&1
  tail -5 $OUT/q1trace.log
  python - <<PY > $OUT/q1trace_hip_api.md 2>&1
import sqlite3, glob
db = glob.glob("$OUT/q1trace/**/*results.db", recursive=True)
con = sqlite3.connect(db[0])
names = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
print(names)
for v in ("top", "top_kernels", "hip_api_stats", "top_hip_api"):
    if v in names:
        print("##", v)
        cur = con.execute(f"select * from {v} limit 40")
        print([d[0] for d in cur.description])
        for r in cur: print(r)
PY
  head -c 6000 $OUT/q1trace_hip_api.md
fi
if has benchq1; then
  (time timeout 900 python bench.py --workload q1 --sf 100 --steps 10 --warmup 3 ${BENCH_EXTRA:-}) > $OUT/bench_q1.json 2> $OUT/bench_q1.err
  tail -1 $OUT/bench_q1.json | head -c 3000; echo; tail -3 $OUT/bench_q1.err
fi
if has benchq3; then
  (time timeout 1200 python bench.py --workload q3 --sf 300 --steps 5 --warmup 2 ${BENCH_EXTRA:-}) > $OUT/bench_q3.json 2> $OUT/bench_q3.err
  tail -1 $OUT/bench_q3.json | head -c 3000; echo; tail -3 $OUT/bench_q3.err
fi
if has bench; then
  (time timeout 900 python bench.py ${BENCH_EXTRA:-}) > $OUT/bench.json 2> $OUT/bench.err
  tail -1 $OUT/bench.json | head -c 3000; echo; tail -3 $OUT/bench.err
fi
if has ops; then
  (time timeout 1200 python scripts/bench_ops.py --md $OUT/ops.md ${OPS_ARGS:-}) > $OUT/ops.jsonl 2> $OUT/ops.err
  cat $OUT/ops.md; tail -5 $OUT/ops.err
fi
if has shapes; then
  (time timeout 1200 python scripts/bench_join_shapes.py --md $OUT/join_shapes.md ${SHAPES_ARGS:-}) > $OUT/join_shapes.jsonl 2> $OUT/join_shapes.err
  cat $OUT/join_shapes.md; tail -5 $OUT/join_shapes.err
fi
if has profjoin; then python scripts/profile_run.py $TAG/prof_join ${PROF_FLAGS:-} -- --steps 20 --warmup 3 --no-also > $OUT/prof_join.log 2>&1; tail -30 $OUT/prof_join.log; fi
if has profq1; then python scripts/profile_run.py $TAG/prof_q1 ${PROF_FLAGS:-} -- --workload q1 --sf 100 --steps 10 --warmup 3 > $OUT/prof_q1.log 2>&1; tail -30 $OUT/prof_q1.log; fi
if has profq3; then python scripts/profile_run.py $TAG/prof_q3 ${PROF_FLAGS:-} -- --workload q3 --sf 300 --steps 5 --warmup 2 > $OUT/prof_q3.log 2>&1; tail -30 $OUT/prof_q3.log; fi
if has extra; then bash -c "${EXTRA_CMD}" > $OUT/extra.log 2>&1; tail -40 $OUT/extra.log; fi
