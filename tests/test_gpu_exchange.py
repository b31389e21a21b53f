"""The multi-GPU exchange below the C ABI (dfgpu_comm_* / dfgpu_exchange_*, datafusion_amd/csrc/exchange.hip) on the GPU box.

A 1-GPU box offers two ways to run the N > 1 code: a one-rank RCCL communicator (every slice is a self slice: partitioning,
metadata all-gathers through RCCL, result assembly, validity / Boolean / dictionary handling all run, nothing crosses a
link), and TWO processes sharing the GPU whose communicator uses the host transport over a gloo group
(dfgpu_comm_init_host) — real two-rank exchanges of device tables: counts, offsets, bitmap re-basing and dictionary merging
are exercised with peers, only the wire differs from RCCL.  The device plan nodes of physical_plan.py (RepartitionExec,
CoalescePartitionsExec, SortPreservingMergeExec, Partitioned hash joins, Partial -> FinalPartitioned aggregates) run the
reference's pinned TPC-H plans that way on 2 ranks down to the reference's answers."""
import os
import pickle
import subprocess
import sys

import numpy as np
import pyarrow as pa
import pytest

from tests.util import assert_tables_equal

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def dict_col(codes, values, index_type=pa.int32(), mask=None):
    return pa.DictionaryArray.from_arrays(pa.array(codes, type=index_type, mask=mask), pa.array(values, type=pa.string()))


def decoded(t: pa.Table) -> pa.Table:
    return pa.table({n: (c.cast(pa.string()) if pa.types.is_dictionary(c.type) else c) for n, c in zip(t.column_names, t.columns)})


def mixed_table(seed, n, values=("AIR", "MAIL", "RAIL", "SHIP", "TRUCK")):
    """integer key, Decimal128 payload, a nullable int column, a Boolean column, a dictionary-encoded string column with NULLs and a plain Utf8
    column with NULLs"""
    rng = np.random.default_rng(seed)
    return pa.table({
        "k": pa.array(rng.integers(0, 10**6, n), type=pa.int64()),
        "d": pa.array([None if x % 11 == 0 else int(x) for x in rng.integers(0, 10**9, n)], type=pa.decimal128(15, 2)),
        "q": pa.array(rng.integers(0, 1000, n), type=pa.int32(), mask=rng.random(n) < 0.2),
        "b": pa.array(rng.random(n) < 0.4, type=pa.bool_(), mask=rng.random(n) < 0.1),
        "s": dict_col(rng.integers(0, len(values), n), list(values), mask=rng.random(n) < 0.15),
        # a plain Utf8 column (bytes in HBM: a Parquet comment column): lengths and bytes cross the exchange, offsets are rebuilt
        "u": pa.array([None if x % 13 == 0 else ("żółw " * int(x % 4) + f"comment {x}") for x in rng.integers(0, 10**6, n)], type=pa.string()),
    })


# ---------------------------------------------------------------------------------------- one rank, RCCL communicator
@pytest.mark.parametrize("n", [0, 1, 5000, 100_003])
def test_one_rank_rccl_exchanges_carry_nulls_booleans_and_dictionaries(n):
    from datafusion_amd.exchange import Comm
    from datafusion_amd.table import DeviceTable
    t = mixed_table(1, n)
    c = Comm.single()
    d = DeviceTable.from_arrow(t)
    for out in (c.hash_exchange(d, ["k"]), c.hash_exchange(d, ["s", "q"]), c.broadcast(d)):
        got = out.to_arrow()
        assert pa.types.is_dictionary(got.schema.field("s").type)
        assert_tables_equal(decoded(got), decoded(t), ordered=True)     # one rank: one partition, order kept
    st = c.stats()
    assert st["bytes_sent_to_peers"] == 0 and st["collectives"] >= 3
    c.free()


@pytest.mark.parametrize("n,chunks", [(0, 3), (1, 2), (100_003, 1), (100_003, 5), (4096, 64)])
def test_one_rank_streaming_exchange_hands_over_every_row_chunk_by_chunk(n, chunks):
    """dfgpu_exchange_hash_stream_* with one rank: the chunks, concatenated, are the blocking exchange's rows in order (one partition: the
    row ranges in input order); a table with a Utf8 column arrives as ONE chunk, a table without as `chunks` of them (empty ranges included)"""
    from datafusion_amd.exchange import Comm
    from datafusion_amd.table import DeviceTable
    t = mixed_table(5, n)
    c = Comm.single()
    for cols in (t.column_names, ["k", "d", "q", "b", "s"]):
        d = DeviceTable.from_arrow(t.select(cols))
        parts = [x.to_arrow() for x in c.hash_exchange_stream(d, ["k"], chunks)]
        assert len(parts) == (1 if "u" in cols else chunks)
        assert_tables_equal(decoded(pa.concat_tables(parts)), decoded(t.select(cols)), ordered=True)
    c.free()


def test_one_rank_pruned_broadcast_keeps_rows_inside_the_probe_bounds():
    import pyarrow.compute as pc

    from datafusion_amd.exchange import Comm
    from datafusion_amd.table import DeviceTable
    t = mixed_table(2, 50_000)
    probe = pa.table({"k2": pa.array(np.random.default_rng(3).integers(200_000, 600_001, 20_000), type=pa.int64())})
    c = Comm.single()
    got = c.broadcast_pruned(DeviceTable.from_arrow(t), "k", DeviceTable.from_arrow(probe), "k2").to_arrow()
    lo, hi = pc.min(probe.column("k2")).as_py(), pc.max(probe.column("k2")).as_py()
    exp = t.filter(pc.and_(pc.greater_equal(t.column("k"), lo), pc.less_equal(t.column("k"), hi)))
    assert 0 < exp.num_rows < t.num_rows
    assert_tables_equal(decoded(got), decoded(exp), ordered=True)
    c.free()


# ---------------------------------------------------------------------------------------- two ranks on one GPU (host transport)
_WORKER = r"""
import os, pickle, sys
import numpy as np, pyarrow as pa
import torch.distributed as dist
sys.path.insert(0, os.environ["DFGPU_ROOT"])
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
from datafusion_amd import _lib
_lib.init(0)                                   # both ranks share GPU 0
from datafusion_amd.table import DeviceTable
out = {}
mode = os.environ["DFGPU_TEST_MODE"]
if mode == "exchange":
    from datafusion_amd.exchange import comm_for
    from tests.test_gpu_exchange import mixed_table, decoded
    # every rank encodes its strings with its OWN dictionary (different value sets, different order)
    values = [("RAIL", "AIR", "TRUCK", "FOB"), ("SHIP", "AIR", "MAIL", "REG AIR", "TRUCK")][rank % 2]
    t = mixed_table(10 + rank, 20_000 + 777 * rank, values)
    d = DeviceTable.from_arrow(t)
    c = comm_for()
    assert c.world == world and c.rank == rank
    out["input"] = decoded(t)
    out["hash_k"] = decoded(c.hash_exchange(d, ["k"]).to_arrow())
    out["hash_s"] = decoded(c.hash_exchange(d, ["s"]).to_arrow())
    out["broadcast"] = decoded(c.broadcast(d).to_arrow())
    probe = pa.table({"k2": pa.array(np.random.default_rng(50 + rank).integers(300_000 * rank, 300_000 * rank + 400_000, 5000), type=pa.int64())})
    out["probe"] = probe
    out["pruned"] = decoded(c.broadcast_pruned(d, "k", DeviceTable.from_arrow(probe), "k2").to_arrow())
    out["stats"] = c.stats()
    # the streaming exchange: a join builder consumes the chunks of one side as they land, the probe takes the other side's chunk by chunk
    nk = d.select(["k", "d", "q", "b", "s"])
    out["stream_k"] = [decoded(x.to_arrow()) for x in c.hash_exchange_stream(nk, ["k"], 4)]
    out["stream_s"] = [decoded(x.to_arrow()) for x in c.hash_exchange_stream(nk, ["s", "q"], 3)]
    from datafusion_amd import ops as _o
    bt = DeviceTable.from_arrow(pa.table({"bk": pa.array(np.arange(0, 10**6, 7, dtype=np.int64)[rank::world]), "bv": pa.array(np.arange(0, 10**6, 7, dtype=np.int64)[rank::world] * 3)}))
    builder = _o.JoinBuilder([0])
    for chunk in c.hash_exchange_stream(bt, ["bk"], 3):
        builder.push(chunk)
    ht = builder.finish()
    joined = [ht.probe(chunk, ["k"], "Inner", ["bv"], ["k", "q"]).to_arrow() for chunk in c.hash_exchange_stream(d.select(["k", "q"]), ["k"], 4)]
    out["stream_join"] = pa.concat_tables(joined)
    # the exchange of a distributed ORDER BY (dfgpu_exchange_range) + the local sort: rank order = sort order
    from datafusion_amd import ops as _ops
    out["range_asc"] = decoded(_ops.sort(c.range_exchange(d, "k", False, False), [("k", False, False), ("d", True, False)]).to_arrow())
    out["range_desc_nulls_first"] = decoded(_ops.sort(c.range_exchange(d, "q", True, True), [("q", True, True), ("k", False, False)]).to_arrow())
    from datafusion_amd import queries as _Q
    out["merge_sorted"] = decoded(_Q._merge_sorted(_ops.sort(d, [("k", False, False), ("d", True, False)]), [("k", False, False), ("d", True, False)], None).to_arrow())
    # PartitionMode::Partitioned on a STRING key: the two sides cross in separate exchanges — one dictionary-encoded with this rank's own
    # dictionary, the other as plain Utf8 bytes — and equal strings must still meet on one rank (routing hashes the bytes)
    from datafusion_amd import ops
    rng = np.random.default_rng(70 + rank)
    words = [f"w{i}" for i in range(40)] + list(values)
    left = pa.table({"s": pa.array([words[i] for i in rng.integers(0, len(words), 3000)], pa.string()).dictionary_encode(), "a": pa.array(rng.integers(0, 10**6, 3000))})
    right = pa.table({"s2": pa.array([words[i] for i in rng.integers(0, len(words), 4000)], pa.string()), "b": pa.array(rng.integers(0, 10**6, 4000))})
    lx = c.hash_exchange(DeviceTable.from_arrow(left), ["s"])
    rx = c.hash_exchange(DeviceTable.from_arrow(right), ["s2"])
    out["pjoin"] = decoded(ops.hash_join(lx, rx, [("s", "s2")], "Inner").to_arrow())
    out["pjoin_inputs"] = (decoded(left), right)
elif mode == "stream_abandoned":
    # rank 0's consumer leaves the stream after its first chunk; rank 1 drains.  Nobody hangs: rank 0 enters the next chunk's collective
    # poisoned, rank 1 gets an error at that chunk — and the communicator is still in step: a blocking exchange afterwards works
    from datafusion_amd.exchange import comm_for
    from tests.test_gpu_exchange import mixed_table, decoded
    t = mixed_table(20 + rank, 30_000)
    d = DeviceTable.from_arrow(t).select(["k", "d", "q"])
    c = comm_for()
    got, err = [], None
    try:
        for i, chunk in enumerate(c.hash_exchange_stream(d, ["k"], 6)):
            got.append(chunk.num_rows)
            if rank == 0 and i == 0:
                break
    except Exception as e:
        err = str(e)
    out["chunks"], out["error"] = got, err
    # a collective while a stream is open, and freeing the communicator under it, are refused
    gen = c.hash_exchange_stream(d, ["k"], 2)
    first = next(gen)
    refused = []
    for what in (lambda: c.hash_exchange(d, ["k"]), lambda: c.free()):
        try:
            what()
            refused.append(None)
        except Exception as e:
            refused.append(str(e))
    rest = [first.num_rows] + [x.num_rows for x in gen]
    out["refused"], out["second_stream_rows"] = refused, sum(rest)
    out["after"] = decoded(c.hash_exchange(d, ["k"]).to_arrow())
    out["input"] = decoded(d.to_arrow())
elif mode == "plans":
    from datafusion_amd import physical_plan as P
    from tests.test_tpch_answers import data, plans
    tables = {}
    for name, t in data().items():             # this rank's row range of every table: what `world` scans produce
        lo, hi = t.num_rows * rank // world, t.num_rows * (rank + 1) // world
        tables[name] = DeviceTable.from_arrow(t.slice(lo, hi - lo))
    from tests.test_tpch_answers import GPU_QUERIES, q16_with_many_complaints
    for q, plan in plans(tables).items():
        if q not in GPU_QUERIES:
            continue
        opt = P.GpuOffloadRule(world_size=world).optimize(plan)
        out[q] = P.collect(opt).to_arrow()
        if q in ("q1", "q3"):
            out[q + "_pinned"] = P.collect(plan).to_arrow()
    if "q16" in GPU_QUERIES:     # CollectLeft + build-side emission with a build side that really loses rows (HashJoinExec.execute)
        out["q16_many"] = P.collect(P.GpuOffloadRule(world_size=world).optimize(q16_with_many_complaints(tables))).to_arrow()
pickle.dump(out, open(os.path.join(os.environ["DFGPU_OUT"], f"r{rank}.pkl"), "wb"))
dist.barrier()
print("WORKER_OK", flush=True)
os._exit(0)
"""


def _run_ranks(tmp_path, world, mode, port):
    procs = []
    for r in range(world):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), WORLD_SIZE=str(world), DFGPU_ROOT=ROOT,
                   DFGPU_OUT=str(tmp_path), DFGPU_TEST_MODE=mode)
        procs.append(subprocess.Popen([sys.executable, "-c", _WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT))
    outs = [p.communicate(timeout=900) for p in procs]
    for r, (so, se) in enumerate(outs):
        assert "WORKER_OK" in so, f"rank {r}:\n{so[-2000:]}\n{se[-4000:]}"
    return [pickle.load(open(tmp_path / f"r{r}.pkl", "rb")) for r in range(world)]


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def cut(n, q, chunks):
    """row range boundaries of the streaming exchange (exchange.hip hash_stream_partition): multiples of 64, the last one the row count"""
    return n if q >= chunks else (n * q // chunks) // 64 * 64


def test_two_ranks_exchange_device_tables_with_different_dictionaries_and_nulls(tmp_path):
    import pyarrow.compute as pc

    from oracle import oracle
    world = 2
    res = _run_ranks(tmp_path, world, "exchange", _free_port())
    inputs = [r["input"] for r in res]
    whole = pa.concat_tables(inputs)
    # RepartitionExec(Hash(k)): rank r holds exactly the rows the oracle routes to r, sender by sender in sender order
    for r in range(world):
        exp = pa.concat_tables([oracle.hash_partition(inp.select(["k"]).append_column("row", pa.array(np.arange(inp.num_rows))), ["k"], world)[0][r]
                                for inp in inputs])
        got = res[r]["hash_k"]
        assert got.column("k").to_pylist() == exp.column("k").to_pylist()
    assert_tables_equal(pa.concat_tables([r["hash_k"] for r in res]), whole, ordered=False)
    # the streaming exchange: chunk k of rank r = the rows of every sender's k-th row range that route to r, sender by sender — the chunks of a
    # rank together are exactly the blocking exchange's rows (another order), and a join fed chunk by chunk finds every match once
    for r in range(world):
        assert len(res[r]["stream_k"]) == 4 and len(res[r]["stream_s"]) == 3
        five = ["k", "d", "q", "b", "s"]
        for k_, chunk in enumerate(res[r]["stream_k"]):
            exp = pa.concat_tables([oracle.hash_partition(inp.select(["k"]).slice(cut(inp.num_rows, k_, 4), cut(inp.num_rows, k_ + 1, 4) - cut(inp.num_rows, k_, 4)), ["k"], world)[0][r]
                                    for inp in inputs])
            assert chunk.column("k").to_pylist() == exp.column("k").to_pylist()
        assert_tables_equal(pa.concat_tables(res[r]["stream_k"]), res[r]["hash_k"].select(five), ordered=False)
    assert_tables_equal(pa.concat_tables([x for r in res for x in r["stream_s"]]), whole.select(["k", "d", "q", "b", "s"]), ordered=False)
    got = sorted(zip(*[pa.concat_tables([r["stream_join"] for r in res]).column(c_).to_pylist() for c_ in ("bv", "k", "q")]), key=lambda x: (x[1], x[0], -1 if x[2] is None else x[2]))
    want = sorted(((k_ * 3, k_, q_) for k_, q_ in zip(whole.column("k").to_pylist(), whole.column("q").to_pylist()) if k_ % 7 == 0), key=lambda x: (x[1], x[0], -1 if x[2] is None else x[2]))
    assert len(want) > 1000 and got == want
    # routing on the string column: the ranks' dictionaries differ, yet every string (and NULL) meets on ONE rank and nothing is lost
    assert_tables_equal(pa.concat_tables([r["hash_s"] for r in res]), whole, ordered=False)
    homes = {}
    for r in range(world):
        for v in set(res[r]["hash_s"].column("s").to_pylist()):
            assert homes.setdefault(v, r) == r, f"string {v!r} was routed to two ranks"
    assert len(set(homes.values())) == world                      # and both ranks got something
    # all-gather: every rank holds all rows in rank order
    for r in range(world):
        assert_tables_equal(res[r]["broadcast"], whole, ordered=True)
    # pruned all-gather: rank r holds the build rows (of all ranks, rank order) inside r's own probe bounds
    for r in range(world):
        lo, hi = pc.min(res[r]["probe"].column("k2")).as_py(), pc.max(res[r]["probe"].column("k2")).as_py()
        exp = whole.filter(pc.and_(pc.greater_equal(whole.column("k"), lo), pc.less_equal(whole.column("k"), hi)))
        assert_tables_equal(res[r]["pruned"], exp, ordered=True)
        assert res[r]["stats"]["bytes_sent_to_peers"] > 0 and res[r]["stats"]["rows_received_from_peers"] > 0
    # distributed ORDER BY: every rank owns one key range; read in rank order the sorted ranges are the oracle's sort of everything
    exp = oracle.sort(whole, [("k", False, False), ("d", True, False)])
    got = pa.concat_tables([r["range_asc"] for r in res])
    assert got.column("k").to_pylist() == exp.column("k").to_pylist() and got.column("d").to_pylist() == exp.column("d").to_pylist()
    assert all(r["range_asc"].num_rows > 0.2 * whole.num_rows for r in res)        # the sampled splitters balance the ranks
    exp = oracle.sort(whole, [("q", True, True), ("k", False, False)])
    got = pa.concat_tables([r["range_desc_nulls_first"] for r in res])
    assert got.column("q").to_pylist() == exp.column("q").to_pylist() and got.column("k").to_pylist() == exp.column("k").to_pylist()
    assert res[0]["range_desc_nulls_first"].column("q").null_count == whole.column("q").null_count      # NULLS FIRST: all on the first rank
    exp = oracle.sort(whole, [("k", False, False), ("d", True, False)])
    for r in res:                                                                  # SortPreservingMergeExec: ONE output, on every rank
        assert r["merge_sorted"].column("k").to_pylist() == exp.column("k").to_pylist() and r["merge_sorted"].column("d").to_pylist() == exp.column("d").to_pylist()
    # the partitioned join on the string key: the ranks' results together are the global join
    lefts = pa.concat_tables([r["pjoin_inputs"][0] for r in res])
    rights = pa.concat_tables([r["pjoin_inputs"][1] for r in res])
    by_s = {}
    for s_, b in zip(rights.column("s2").to_pylist(), rights.column("b").to_pylist()):
        by_s.setdefault(s_, []).append(b)
    want = sorted((s_, a, b) for s_, a in zip(lefts.column("s").to_pylist(), lefts.column("a").to_pylist()) for b in by_s.get(s_, []))
    got = sorted((x["s"], x["a"], x["b"]) for r in res for x in r["pjoin"].to_pylist())
    assert len(want) > 100_000 and got == want


def test_device_plans_on_two_ranks_reproduce_the_reference_answers(tmp_path):
    """the product's plan nodes with N = 2 (device operators + C-ABI exchanges; tests/test_plans_gloo.py runs the same protocol
    with the oracle's operators on CPU)"""
    from tests.test_tpch_answers import GPU_QUERIES, assert_answer
    res = _run_ranks(tmp_path, 2, "plans", _free_port())
    for q in GPU_QUERIES:
        for r in range(2):
            assert_answer(q, res[r][q])
    if "q16" in GPU_QUERIES:
        from tests import plan_oracle
        from tests.test_tpch_answers import data, q16_with_many_complaints
        want = plan_oracle.collect(q16_with_many_complaints(data()))
        for r in range(2):
            got = res[r]["q16_many"]
            got = pa.table({n: (c.cast(pa.string()) if pa.types.is_dictionary(c.type) else c) for n, c in zip(got.column_names, got.columns)})
            assert got.to_pylist() == want.to_pylist()
    for q in ("q1_pinned", "q3_pinned"):
        for r in range(2):
            assert_answer(q[:2], res[r][q])


def test_bench_two_ranks_rehearsal_prints_its_line(tmp_path):
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one process per rank), rehearsed on ONE GPU: both ranks on device 0,
    a gloo group for the barriers and the host transport under dfgpu_exchange_* (DFGPU_BENCH_REHEARSAL=1).  Rank 0 prints ONE JSON line:
    the blocking hash exchange is measured first, the streamed form and the pruned broadcast beside it, none of them reports an error,
    and every exchange joins the same number of rows"""
    import json
    env = dict(os.environ, DFGPU_BENCH_REHEARSAL="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--sf", "1", "--steps", "2", "--warmup", "1"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert p.returncode == 0 and len(lines) == 1, (p.returncode, p.stdout[-2000:], p.stderr[-4000:])
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["metric"] == "tpch_q3_hash_join_rows_per_sec" and line["value"] > 0 and line["scaling"] == "strong"
    assert "exchange_errors" not in line, line.get("exchange_errors")
    ex = line["exchanges"]
    assert {"repartition", "repartition_stream", "pruned"} <= set(ex), sorted(ex)
    assert line["config"]["exchange"] in ("repartition", "repartition_stream")
    assert line["config"]["output_rows"] == line["config"]["probe_rows"] > 5_900_000
    for k, v in ex.items():
        assert v["ms_per_step"] > 0 and v["transport"] == "host", (k, v)


def test_bench_two_ranks_without_a_launcher_starts_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher (no RANK / WORLD_SIZE in the environment — how the driver starts N = 1) starts the two ranks
    itself (torch.distributed.run on 127.0.0.1) instead of exiting: one line, three exchanges (rehearsed on one GPU over the host transport)"""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["DFGPU_BENCH_REHEARSAL"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--sf", "1", "--steps", "2", "--warmup", "1"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert p.returncode == 0 and len(lines) == 1, (p.returncode, p.stdout[-2000:], p.stderr[-4000:])
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 0 and "watchdog" not in line and "exchange_errors" not in line
    assert {"repartition", "repartition_stream", "pruned"} <= set(line["exchanges"])


def test_a_rank_that_leaves_a_streamed_exchange_early_takes_every_rank_out_with_an_error(tmp_path):
    """round-5 advice: a consumer that leaves the stream early (or an error on one rank's threads) used to stop that rank's all-to-all(v)s
    while its peers waited in theirs forever.  Now the leaving rank enters the next chunk's collective poisoned: every rank abandons the
    stream at the same chunk with an error, the communicator stays in step (a blocking exchange right after works), and while a stream
    is open the communicator refuses other collectives and dfgpu_comm_free"""
    world = 2
    res = _run_ranks(tmp_path, world, "stream_abandoned", _free_port())
    assert res[0]["error"] is None and len(res[0]["chunks"]) == 1                      # rank 0 left after its first chunk
    assert res[1]["error"] is not None and "abandoned" in res[1]["error"] and 1 <= len(res[1]["chunks"]) <= 4, res[1]   # (rank 0 runs up to two chunks ahead of its consumer)
    for r in range(world):
        assert all(x is not None and "streamed exchange" in x for x in res[r]["refused"]), res[r]["refused"]
    whole = pa.concat_tables([r["input"] for r in res])
    assert sum(r["second_stream_rows"] for r in res) == whole.num_rows
    assert_tables_equal(pa.concat_tables([r["after"] for r in res]), whole, ordered=False)
