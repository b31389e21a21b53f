cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for w in 4 8; do for pm in 3 2; do
echo "W=$w PM=$pm: $(DFGPU_FUSED_WORDS=$w timeout 300 python bench.py --no-cpu --steps 3 --warmup 1 --probe-mode $pm 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["kernels"], d.get("ordered_output_two_pass"))')"
done; done
for pm in "DFGPU_FUSED_WORDS=4" "DFGPU_FUSED_WORDS=8"; do
echo "Q3 $pm: $(env $pm timeout 300 python scripts/bench_ops.py --only q3 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms"], d["kernel_ms_per_iter"])')"
done
