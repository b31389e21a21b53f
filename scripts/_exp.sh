#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp
export DFGPU_JIT_STRICT=1
(time timeout 500 env DFGPU_JIT_MIN_ROWS=0 python -m pytest tests/test_gpu_fused.py tests/test_gpu_aggregate.py tests/test_gpu_queries.py -x -q) 2>&1 | tail -25
for v in "DFGPU_JIT=1"; do
  echo "== $v"
  env $v timeout 300 python scripts/bench_ops.py --only q1 --md gpurun_out/exp/q1_$v.md 2> gpurun_out/exp/err_$v.log | python -c '
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d.get("case"), d.get("ms"), d.get("kernel_ms_per_iter"))'
  tail -20 gpurun_out/exp/err_$v.log
done
