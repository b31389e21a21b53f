#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp
(time timeout 400 python -m pytest tests/test_gpu_join.py tests/test_gpu_queries.py tests/test_abi.py -x -q --timeout 120) 2>&1 | tail -25
timeout 150 python scripts/bench_ops.py --only q3 2> gpurun_out/exp/err_q3.log | python -c '
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d.get("case"), d.get("ms"), d.get("kernel_ms_per_iter"))'
tail -5 gpurun_out/exp/err_q3.log
