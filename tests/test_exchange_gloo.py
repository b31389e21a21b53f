"""The N>1 path on CPU: world_size-2 gloo run of the hash-repartition exchange collectives
(datafusion_amd/exchange.py: exchange_counts + all_to_all_bytes), with the oracle standing in for the
device partition kernel (same hash, same `hash % world` routing — tests/test_gpu_sort_partition.py
pins that equality on the GPU).  Checks the RepartitionExec(Hash) contract end to end: every row
lands on the rank its key hash routes to, nothing is lost or duplicated, and per-rank joins of the
co-partitioned sides add up to the global join (PartitionMode::Partitioned, hash_join/exec.rs:1314-1324)."""
import os
import pickle
import socket
import sys

import numpy as np
import pyarrow as pa
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _table_slice(seed, rank, world):
    rng = np.random.default_rng(seed)
    n_b, n_p = 4000, 15000
    build = pa.table({"a": pa.array(rng.permutation(6000)[:n_b], type=pa.int64()), "x": pa.array(rng.integers(0, 100, n_b), type=pa.int32())})
    probe = pa.table({"b": pa.array(rng.integers(0, 7000, n_p), type=pa.int64()), "y": pa.array(rng.integers(0, 10**6, n_p), type=pa.int64())})
    lo_b, hi_b = n_b * rank // world, n_b * (rank + 1) // world
    lo_p, hi_p = n_p * rank // world, n_p * (rank + 1) // world
    return build, probe, build.slice(lo_b, hi_b - lo_b), probe.slice(lo_p, hi_p - lo_p)


def _exchange_cpu(table, key, world, max_message_bytes=None):
    """RepartitionExec(Hash) over gloo: partition (oracle) -> counts all-to-all -> one all-to-all(v) per column"""
    from datafusion_amd.exchange import all_to_all_bytes, exchange_counts
    from oracle import oracle
    parts, _ = oracle.hash_partition(table, [key], world)
    send_counts = [p.num_rows for p in parts]
    recv_counts = exchange_counts(send_counts)
    out = {}
    for name in table.column_names:
        width = table.schema.field(name).type.bit_width // 8
        send_np = np.concatenate([np.frombuffer(oracle.values_np(p.column(name)).tobytes(), dtype=np.uint8) for p in parts]) if table.num_rows else np.zeros(0, np.uint8)
        send = torch.from_numpy(send_np.copy())
        recv = torch.empty(sum(recv_counts) * width, dtype=torch.uint8)
        all_to_all_bytes(send, send_counts, recv, recv_counts, width, max_message_bytes=max_message_bytes)
        out[name] = pa.Array.from_buffers(table.schema.field(name).type, sum(recv_counts), [None, pa.py_buffer(recv.numpy().tobytes())])
    return pa.table(out)


def _worker(rank, world, port, seed, outdir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from datafusion_amd.exchange import route
    from oracle import oracle
    _, _, my_build, my_probe = _table_slice(seed, rank, world)
    b = _exchange_cpu(my_build, "a", world)
    p = _exchange_cpu(my_probe, "b", world, max_message_bytes=7000)   # several rounds per column (splits are ~40-60 KB)
    # routing contract: hash(key; seed 0) % world == rank for every received row
    for t, k in ((b, "a"), (p, "b")):
        h = oracle.create_hashes([t.column(k)], 0)
        assert (route(h, world) == rank).all()
    joined = oracle.hash_join(b, p, [("a", "b")], "Inner")
    pickle.dump({"build_rows": b.num_rows, "probe_rows": p.num_rows, "join": joined.to_pylist()}, open(os.path.join(outdir, f"r{rank}.pkl"), "wb"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_hash_exchange_over_gloo(tmp_path, world):
    from oracle import oracle
    seed = 1234
    port = _free_port()
    mp.spawn(_worker, args=(world, port, seed, str(tmp_path)), nprocs=world, join=True)
    res = [pickle.load(open(tmp_path / f"r{r}.pkl", "rb")) for r in range(world)]
    build, probe, _, _ = _table_slice(seed, 0, world)
    assert sum(r["build_rows"] for r in res) == build.num_rows
    assert sum(r["probe_rows"] for r in res) == probe.num_rows
    exp = oracle.hash_join(build, probe, [("a", "b")], "Inner").to_pylist()
    got = [row for r in res for row in r["join"]]
    key = lambda d: tuple(d.values())
    assert sorted(got, key=key) == sorted(exp, key=key)


def _gather_worker(rank, world, port, ragged, outdir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from datafusion_amd.exchange import all_gather_bytes, gather_counts
    n = 1000 + (37 * rank if ragged else 0)
    out = {}
    for width, dtype in ((16, np.uint64), (8, np.int64), (4, np.int32), (1, np.uint8)):
        k = max(1, width // np.dtype(dtype).itemsize)
        mine = (np.arange(n * k, dtype=np.int64) + 10**6 * rank).astype(dtype)
        counts = gather_counts(n)
        assert counts == [1000 + (37 * r if ragged else 0) for r in range(world)]
        send = torch.from_numpy(np.frombuffer(mine.tobytes(), dtype=np.uint8).copy())
        recv = torch.empty(sum(counts) * width, dtype=torch.uint8)
        all_gather_bytes(send, recv, counts, width)
        out[width] = recv.numpy().tobytes()
    pickle.dump(out, open(os.path.join(outdir, f"g{rank}.pkl"), "wb"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,ragged", [(2, False), (3, True)])
def test_broadcast_build_all_gather_over_gloo(tmp_path, world, ragged):
    """CollectLeft build side: every rank ends up with all ranks' rows in rank order (equal and ragged shards)"""
    port = _free_port()
    mp.spawn(_gather_worker, args=(world, port, ragged, str(tmp_path)), nprocs=world, join=True)
    res = [pickle.load(open(tmp_path / f"g{r}.pkl", "rb")) for r in range(world)]
    for width, dtype in ((16, np.uint64), (8, np.int64), (4, np.int32), (1, np.uint8)):
        k = max(1, width // np.dtype(dtype).itemsize)
        exp = b"".join((np.arange((1000 + (37 * r if ragged else 0)) * k, dtype=np.int64) + 10**6 * r).astype(dtype).tobytes() for r in range(world))
        for r in range(world):
            assert res[r][width] == exp


def _pruned_worker(rank, world, port, clustered, outdir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from datafusion_amd.exchange import all_to_all_bytes, exchange_counts, gather_key_ranges, nothing_crosses_ranks, pruned_send_ranges
    rng = np.random.default_rng(99)
    nb, npr = 6000, 20000
    bkeys = np.sort(rng.permutation(50_000)[:nb]).astype(np.int64)
    pkeys = rng.integers(0, 50_000, npr).astype(np.int64)
    if clustered:
        pkeys = np.sort(pkeys)          # range-partitioned scans: each rank's probe keys cover a narrow range
    my_b = bkeys[nb * rank // world: nb * (rank + 1) // world]
    my_p = pkeys[npr * rank // world: npr * (rank + 1) // world]
    if rank == world - 1 and not clustered:
        my_p = my_p[:0]                 # a rank without probe rows asks for nothing
    my_b_range = (int(my_b.min()), int(my_b.max())) if len(my_b) else None
    ranges, branges = gather_key_ranges((int(my_p.min()), int(my_p.max())) if len(my_p) else None, my_b_range)
    assert branges[rank] == my_b_range
    assert not nothing_crosses_ranks(ranges, branges)      # random build keys: some always fall inside another rank's probe bounds
    sends = pruned_send_ranges(ranges, my_b_range)
    parts = [my_b[:0] if sr is None else my_b[(my_b >= sr[0]) & (my_b <= sr[1])] for sr in sends]
    send_counts = [len(x) for x in parts]
    recv_counts = exchange_counts(send_counts)
    send = torch.from_numpy(np.frombuffer(np.concatenate(parts).tobytes(), dtype=np.uint8).copy()) if sum(send_counts) else torch.empty(0, dtype=torch.uint8)
    recv = torch.empty(sum(recv_counts) * 8, dtype=torch.uint8)
    all_to_all_bytes(send, send_counts, recv, recv_counts, 8)
    got_b = np.frombuffer(recv.numpy().tobytes(), dtype=np.int64)
    local_join = np.sort(my_p[np.isin(my_p, got_b)])
    pickle.dump({"join": local_join, "received": len(got_b), "sent_to_peers": sum(send_counts) - send_counts[rank]}, open(os.path.join(outdir, f"p{rank}.pkl"), "wb"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("clustered", [True, False], ids=["range_clustered_probe", "spread_probe"])
def test_bounds_pruned_broadcast_over_gloo(tmp_path, clustered):
    """pruned CollectLeft: every rank receives the build keys inside its own probe-key bounds; the union of the local
    joins equals the global join, and clustered probe shards pull far fewer build rows than a full broadcast"""
    world = 3
    port = _free_port()
    mp.spawn(_pruned_worker, args=(world, port, clustered, str(tmp_path)), nprocs=world, join=True)
    res = [pickle.load(open(tmp_path / f"p{r}.pkl", "rb")) for r in range(world)]
    rng = np.random.default_rng(99)
    bkeys = np.sort(rng.permutation(50_000)[:6000]).astype(np.int64)
    pkeys = rng.integers(0, 50_000, 20000).astype(np.int64)
    if clustered:
        pkeys = np.sort(pkeys)
    else:
        pkeys = pkeys[:20000 * (world - 1) // world]      # the last rank dropped its probe rows
    exp = np.sort(pkeys[np.isin(pkeys, bkeys)])
    got = np.sort(np.concatenate([r["join"] for r in res]))
    assert (got == exp).all() and len(got) == len(exp)
    full_broadcast = 6000 * (world - 1)                   # build rows that cross ranks in a plain all-gather
    moved = sum(r["sent_to_peers"] for r in res)
    assert moved <= full_broadcast
    if clustered:
        assert moved < full_broadcast // 4


def test_pruned_broadcast_fast_path_decision():
    from datafusion_amd.exchange import nothing_crosses_ranks
    # aligned range shards (what bench.py's row ranges are): probe bounds inside the local build range
    assert nothing_crosses_ranks([(1, 90), (101, 190), None], [(1, 100), (101, 200), (201, 300)])
    assert not nothing_crosses_ranks([(1, 150), (101, 190)], [(1, 100), (101, 200)])
    assert nothing_crosses_ranks([None, None], [(1, 100), (50, 200)])


def test_broadcast_vs_repartition_choice():
    from datafusion_amd.exchange import broadcast_build_moves_fewer_bytes
    b, p = 150_000_000 * 16, 600_000_000 * 40   # the SF100 Q3 join of bench.py
    assert all(broadcast_build_moves_fewer_bytes(b, p, n) for n in (2, 4, 8))
    assert not broadcast_build_moves_fewer_bytes(p, b, 2)


def test_route_is_hash_mod_world():
    from datafusion_amd.exchange import route
    h = np.array([0, 1, 7, 8, 2**64 - 1], dtype=np.uint64)
    assert route(h, 8).tolist() == [0, 1, 7, 0, 7]


def _bounds_worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from datafusion_amd import ops, physical_plan as P

    class Build:     # this rank's build partition: its own key range (rank 2 of 3 holds no rows)
        schema = pa.schema([pa.field("bk", pa.int64())])
        num_rows = 5

        def index_of(self, c):
            return 0
    local = {0: (100, 200, 5, True), 1: (150, 900, 5, True), 2: (None, None, 0, False)}[rank]
    ops.column_minmax = lambda t, c: local
    scan = P.ParquetExec("/nonexistent.parquet", ["pk"], "probe")
    j = P.HashJoinExec(P.MemoryExec(Build()), P.RepartitionExec(scan, ["pk"], world), [("bk", "pk")], "Inner")
    j._publish_dynamic_bounds(Build())
    pickle.dump(scan.dynamic_bounds, open(os.path.join(outdir, f"b{rank}.pkl"), "wb"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_dynamic_join_bounds_cover_every_ranks_build_partition(tmp_path, world):
    """Partitioned joins on N GPUs: a rank's probe-side scan feeds every rank, so the published bounds are the union over all
    ranks' build partitions (the reference's SharedBuildAccumulator waits for all partitions, hash_join/shared_bounds.rs)"""
    mp.spawn(_bounds_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert pickle.load(open(tmp_path / f"b{r}.pkl", "rb")) == {"pk": (100, 900)}
