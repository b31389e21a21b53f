#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp
(time timeout 300 python -m pytest tests/test_gpu_fused.py tests/test_gpu_aggregate.py tests/test_gpu_fullsize.py -x -q --timeout 150) 2>&1 | tail -8
timeout 150 python scripts/bench_ops.py --only agg_highcard 2> gpurun_out/exp/err_hc.log | python -c '
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d.get("case"), d.get("ms"), d.get("kernel_ms_per_iter"))'
tail -3 gpurun_out/exp/err_hc.log
(time timeout 300 python bench.py) 2>&1 | tail -6 | cut -c1-2500
