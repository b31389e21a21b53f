cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/dbg
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/dbg/pytest.log 2>&1; grep -E "passed|failed|rror" gpurun_out/dbg/pytest.log | head
timeout 300 python bench.py --no-cpu 2>&1 | tail -1
echo "Q3: $(timeout 300 python scripts/bench_ops.py --only q3 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms"], d["kernel_ms_per_iter"])')"
