#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 200 python -m pytest tests/test_gpu_sort_partition.py -x -q --timeout 150 -k "rccl or exchange" 2>&1 | tail -5
echo "== bench rehearsal pruned"
timeout 200 python bench.py --no-cpu --exchange pruned --steps 3 --warmup 2 2>&1 | tail -2 | cut -c1-1800
