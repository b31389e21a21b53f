"""BASELINE configs 4 and 5 on several ranks, on CPU: the reference's pinned TPC-H physical plans (Partial aggregate ->
RepartitionExec(Hash) -> FinalPartitioned, Partitioned hash joins fed by RepartitionExec on both sides, CoalescePartitionsExec,
SortPreservingMergeExec) executed by 2 and 3 processes over gloo with the oracle's operators — each rank scans its row range,
rows are routed by hash(keys; seed 0) % N exactly as RepartitionExec / exchange.hash_exchange route them — must print the
reference's answers (tests/golden/tpch_answers.json).  This is the N > 1 protocol of physical_plan.py / queries.py with the
oracle standing in for the device operators; with GpuOffloadRule(world_size=N) the exchange nodes stay in the plan."""
import os
import pickle
import socket
import sys

import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
QUERIES = [f"q{i}" for i in range(1, 23)]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, optimized, outdir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from datafusion_amd import physical_plan as P
    from tests import plan_oracle
    from tests.test_tpch_answers import data, plans, q16_with_many_complaints
    out = {}
    many = q16_with_many_complaints(data())
    out["q16_many"] = plan_oracle.collect(P.GpuOffloadRule(world_size=world).optimize(many) if optimized else many)
    for q, plan in plans(data()).items():
        if optimized:
            plan = P.GpuOffloadRule(world_size=world).optimize(plan)
            assert "RepartitionExec" in _names(plan) or q == "q6", q           # the exchanges stay when there is more than one GPU
        out[q] = plan_oracle.collect(plan)
    pickle.dump(out, open(os.path.join(outdir, f"r{rank}.pkl"), "wb"))
    dist.barrier()
    dist.destroy_process_group()


def _names(plan):
    return [plan.name()] + [n for c in plan.children() for n in _names(c)]


@pytest.mark.parametrize("world,optimized", [(2, False), (2, True), (3, True)])
def test_reference_plans_on_several_ranks_reproduce_the_answers(tmp_path, world, optimized):
    from tests.test_tpch_answers import assert_answer
    mp.spawn(_worker, args=(world, _free_port(), optimized, str(tmp_path)), nprocs=world, join=True)
    res = [pickle.load(open(tmp_path / f"r{r}.pkl", "rb")) for r in range(world)]
    for q in QUERIES:
        for r in range(world):          # the root SortPreservingMergeExec / CoalescePartitionsExec replicates the result on every rank
            assert_answer(q, res[r][q])
    # CollectLeft with build-side emission (Q16's null-aware LeftAnti) where the build side really loses rows
    from tests import plan_oracle
    from tests.test_tpch_answers import data, q16_with_many_complaints
    want = plan_oracle.collect(q16_with_many_complaints(data()))
    assert want.to_pylist() != res[0]["q16"].to_pylist()
    for r in range(world):
        assert res[r]["q16_many"].to_pylist() == want.to_pylist()
