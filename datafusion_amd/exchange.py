"""Multi-GPU exchanges = RepartitionExec(Partitioning::Hash) across GPUs (physical-plan/src/repartition/mod.rs:1097-1150,
:1626), the collected build side of PartitionMode::CollectLeft (hash_join/exec.rs:1325-1328) and that all-gather pruned by
probe-key bounds — Python face of the C ABI's dfgpu_comm_* / dfgpu_exchange_* (datafusion_amd/csrc/exchange.hip), which is
where the work happens: partition kernel, RCCL grouped ncclSend / ncclRecv over xGMI issued by the library itself on its
own stream, validity bitmaps, Boolean columns and dictionaries carried across.  One process per GPU.

torch.distributed is used for two things only: handing RCCL's bootstrap id from rank 0 to the other ranks, and — when the
process group is NOT RCCL (gloo: two test ranks sharing one GPU) — as the host transport the library calls back into
(dfgpu_comm_init_host).  The functions of the second half of this file (exchange_counts ... route) are that host transport's
collectives plus the pruning protocol's pure functions; tests/test_exchange_gloo.py runs them with world_size 2 / 3 on CPU.
"""
from __future__ import annotations

import ctypes as C

import numpy as np


class Comm:
    """a dfgpu_comm_t: `world` ranks, this process is `rank`"""

    def __init__(self, handle, world, rank, keep=None):
        self._h, self.world, self.rank, self._keep = handle, world, rank, keep

    # ------------------------------------------------------------------ construction
    @staticmethod
    def single() -> "Comm":
        """a one-rank RCCL communicator (rehearsal of the N > 1 code path on a 1-GPU box)"""
        from . import _lib
        lib = _lib.init()
        uid = (C.c_uint8 * 128)()
        _lib.check(lib.dfgpu_comm_unique_id(uid))
        h = C.c_void_p()
        _lib.check(lib.dfgpu_comm_init_rank(uid, 1, 0, C.byref(h)))
        return Comm(h, 1, 0)

    @staticmethod
    def rccl(group=None) -> "Comm":
        """RCCL communicator over the ranks of a torch.distributed group: rank 0's bootstrap id is broadcast through the group"""
        import torch.distributed as dist

        from . import _lib
        lib = _lib.init()
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        uid = (C.c_uint8 * 128)()
        if rank == 0:
            _lib.check(lib.dfgpu_comm_unique_id(uid))
        box = [bytes(uid)]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        uid = (C.c_uint8 * 128).from_buffer_copy(box[0])
        h = C.c_void_p()
        _lib.check(lib.dfgpu_comm_init_rank(uid, world, rank, C.byref(h)))
        return Comm(h, world, rank)

    @staticmethod
    def host(group=None) -> "Comm":
        """the library's exchange protocol over collectives of `group` on host memory (dfgpu_comm_init_host)"""
        import torch.distributed as dist

        from . import _lib
        lib = _lib.init()
        world, rank = dist.get_world_size(group), dist.get_rank(group)

        def alltoallv(_ctx, send, send_bytes, recv, recv_bytes):
            try:
                host_alltoallv([(send[p], send_bytes[p]) for p in range(world)], [(recv[p], recv_bytes[p]) for p in range(world)], group)
                return 0
            except Exception as e:  # noqa: BLE001 - reported through the C ABI's error channel
                print("host transport alltoallv failed:", e, flush=True)
                return 1

        def allgather(_ctx, mine, nbytes, out):
            try:
                host_allgather(mine, nbytes, out, group)
                return 0
            except Exception as e:  # noqa: BLE001
                print("host transport allgather failed:", e, flush=True)
                return 1
        t = _lib.HostTransport(None, _lib.ALLTOALLV_FN(alltoallv), _lib.ALLGATHER_FN(allgather))
        h = C.c_void_p()
        _lib.check(lib.dfgpu_comm_init_host(C.byref(t), world, rank, C.byref(h)))
        return Comm(h, world, rank, keep=t)

    # ------------------------------------------------------------------ collectives over device tables
    def _call(self, fn, *args):
        from . import _lib
        from .table import DeviceTable
        out = (C.c_void_p * 1)()
        _lib.check(fn(self._h, *args, out))
        return DeviceTable(C.c_void_p(out[0]))

    def hash_exchange(self, table, keys):
        from . import _lib
        idx = [table.index_of(k) for k in keys]
        return self._call(_lib.load().dfgpu_exchange_hash, (C.c_void_p * 1)(table.handle), (C.c_int * len(idx))(*idx), len(idx))

    def hash_exchange_stream(self, table, keys, n_chunks):
        """the same exchange as a stream of `n_chunks` tables (dfgpu_exchange_hash_stream_*): chunk k holds the rows of every rank's k-th
        row range that route here and is handed over while chunk k + 1 crosses the links and chunk k + 2 is being partitioned — the
        consumer (a join builder's push, a probe) overlaps both.  Every rank must pass the same n_chunks and drain the stream."""
        from . import _lib
        from .table import DeviceTable
        lib = _lib.load()
        idx = [table.index_of(k) for k in keys]
        h = C.c_void_p()
        _lib.check(lib.dfgpu_exchange_hash_stream_open(self._h, (C.c_void_p * 1)(table.handle), (C.c_int * len(idx))(*idx), len(idx), int(n_chunks), C.byref(h)))
        try:
            while True:
                out, done = (C.c_void_p * 1)(), C.c_int()
                _lib.check(lib.dfgpu_exchange_hash_stream_next(h, out, C.byref(done)))
                if done.value:
                    break
                yield DeviceTable(C.c_void_p(out[0]))
        finally:
            lib.dfgpu_exchange_hash_stream_free(h)

    def broadcast(self, table):
        from . import _lib
        return self._call(_lib.load().dfgpu_exchange_broadcast, (C.c_void_p * 1)(table.handle))

    def broadcast_pruned(self, build, build_key, probe, probe_key):
        from . import _lib
        return self._call(_lib.load().dfgpu_exchange_broadcast_pruned, (C.c_void_p * 1)(build.handle), build.index_of(build_key),
                          (C.c_void_p * 1)(probe.handle), probe.index_of(probe_key))

    def range_exchange(self, table, key, descending=False, nulls_first=False):
        """the exchange of a distributed ORDER BY (dfgpu_exchange_range): this rank's share of the key range, unsorted"""
        from . import _lib
        return self._call(_lib.load().dfgpu_exchange_range, (C.c_void_p * 1)(table.handle), table.index_of(key), int(descending), int(nulls_first))

    def merge_join_visited(self, ht):
        """OR of the visited marks (and null-aware flags) of a replicated build side's join tables across the ranks
        (dfgpu_exchange_join_visited): after it, emit_unmatched reports the same rows everywhere"""
        from . import _lib
        _lib.check(_lib.load().dfgpu_exchange_join_visited(self._h, (C.c_void_p * 1)(ht._h)))

    def transport_info(self) -> dict:
        """what the transport itself reports (dfgpu_comm_transport_info): RCCL or the host transport, and — under RCCL — how many
        ranks ncclCommCount sees and which one this is"""
        from . import _lib
        a, n, r = C.c_int(), C.c_int(), C.c_int()
        _lib.check(_lib.load().dfgpu_comm_transport_info(self._h, C.byref(a), C.byref(n), C.byref(r)))
        return {"transport": "rccl" if a.value else "host", "n_ranks_seen_by_rccl": n.value if a.value else None, "rccl_rank": r.value if a.value else None}

    def stats(self, reset=False) -> dict:
        from . import _lib
        st = _lib.ExchangeStats()
        _lib.check(_lib.load().dfgpu_comm_stats(self._h, C.byref(st), int(reset)))
        return {n: getattr(st, n) for n, _ in st._fields_}

    def free(self):
        """dfgpu_comm_free; refused (DfgpuError, the communicator stays usable) while a streamed exchange is open on it"""
        if self._h:
            from . import _lib
            _lib.check(_lib.load().dfgpu_comm_free(self._h))
            self._h = None


_COMMS = {}


def comm_for(group=None, force=False):
    """the communicator of a torch.distributed group (cached): RCCL when the group's backend is RCCL, else the host transport
    over the group; None when there is one rank and the caller does not insist (force = one-rank rehearsal on a 1-GPU box)"""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        if not force:
            return None
        if "single" not in _COMMS:
            _COMMS["single"] = Comm.single()
        return _COMMS["single"]
    if dist.get_world_size(group) == 1 and not force:
        return None
    key = ("group", id(group))
    if key not in _COMMS:
        _COMMS[key] = Comm.rccl(group) if "nccl" in str(dist.get_backend(group)) else Comm.host(group)
    return _COMMS[key]


def hash_exchange(table, keys, group=None, force=False):
    """DeviceTable -> DeviceTable holding every row (from all ranks) whose key hash routes here (dfgpu_exchange_hash).
    force=True runs the partition + exchange path even for a single rank (exercises the plumbing on a 1-GPU box)."""
    c = comm_for(group, force)
    return table if c is None else c.hash_exchange(table, keys)


def broadcast_table(table, group=None, force=False):
    """DeviceTable -> DeviceTable holding the rows of ALL ranks in rank order: the build side of a PartitionMode::CollectLeft
    hash join (dfgpu_exchange_broadcast).  Chosen instead of hash-repartitioning both sides when it moves fewer bytes per
    GPU: build_bytes * N < build_bytes + probe_bytes."""
    c = comm_for(group, force)
    return table if c is None else c.broadcast(table)


def pruned_broadcast_table(build, build_key, probe, probe_key, group=None, force=False, stats=None):
    """CollectLeft with the broadcast pruned by the destinations' probe-key bounds (dfgpu_exchange_broadcast_pruned): rank r
    receives from every rank only the build rows whose key lies inside [min, max] of r's probe keys.  `stats` (optional
    dict) receives the rows sent / received across ranks."""
    import pyarrow as pa
    c = comm_for(group, force)
    if c is None:
        return build
    ktype = build.schema.field(build.schema.get_field_index(build_key)).type
    if ktype not in (pa.int64(), pa.int32()):
        return c.broadcast(build)
    before = c.stats()
    out = c.broadcast_pruned(build, build_key, probe, probe_key)
    if stats is not None:
        after = c.stats()
        stats.update(rows_sent_to_peers=after["rows_sent_to_peers"] - before["rows_sent_to_peers"],
                     rows_received_from_peers=after["rows_received_from_peers"] - before["rows_received_from_peers"],
                     bytes_sent_to_peers=after["bytes_sent_to_peers"] - before["bytes_sent_to_peers"],
                     build_rows_local=build.num_rows, build_rows_after_exchange=out.num_rows)
    return out


# ---------------------------------------------------------------------------------------------------------------------
# host transport: the two collectives dfgpu_comm_init_host calls back into, over torch.distributed on host memory
def _host_tensor(ptr, nbytes):
    import torch
    if not nbytes:
        return torch.empty(0, dtype=torch.uint8)
    return torch.from_numpy(np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(int(nbytes),)))


def host_alltoallv(send, recv, group=None):
    """send[p] = (pointer, bytes) for rank p (own entry empty), recv likewise: point-to-point isend / irecv, all in flight"""
    import torch.distributed as dist
    rank = dist.get_rank(group)
    ops, keep = [], []
    for p, (ptr, n) in enumerate(recv):
        if p != rank and n:
            t = _host_tensor(ptr, n)
            keep.append(t)
            ops.append(dist.P2POp(dist.irecv, t, dist.get_global_rank(group, p) if group is not None else p, group))
    for p, (ptr, n) in enumerate(send):
        if p != rank and n:
            t = _host_tensor(ptr, n)
            keep.append(t)
            ops.append(dist.P2POp(dist.isend, t, dist.get_global_rank(group, p) if group is not None else p, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()


def host_allgather(mine, nbytes, out, group=None):
    import torch.distributed as dist
    world = dist.get_world_size(group)
    dist.all_gather_into_tensor(_host_tensor(out, nbytes * world), _host_tensor(mine, nbytes), group=group)


_W = {1: 4, 2: 8, 3: 16, 4: 8, 5: 1, 6: 4, 7: 8, 8: 4}  # dfgpu_type -> bytes


class _DevMem:
    """exposes a raw HBM range through __cuda_array_interface__ (uint8)"""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def _as_tensor(ptr: int, nbytes: int):
    import torch
    if nbytes == 0:
        return torch.empty(0, dtype=torch.uint8, device="cuda")
    return torch.as_tensor(_DevMem(ptr, nbytes), device="cuda")


def exchange_counts(send_counts, group=None):
    """all ranks learn how many rows every peer sends them: returns recv_counts (list[int])"""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    s = torch.tensor(list(send_counts), dtype=torch.int64, device=dev)
    r = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_to_all_single(r, s, group=group)
    return [int(x) for x in r.cpu().tolist()]


MAX_MESSAGE_BYTES = 1 << 30   # per-peer message bound: multi-GB single sends have misbehaved over RCCL (SF100 rehearsal)


def all_to_all_bytes(send, send_counts, recv, recv_counts, width, group=None, max_message_bytes=None):
    """one all-to-all(v) of a column: `send`/`recv` are flat uint8 tensors, counts are rows.
    The buffers are viewed with the widest element type dividing `width` so per-peer element
    counts stay small (a 2.4 GB split of Decimal128 values is 300 M int64 elements, not 2.4 G bytes).
    Splits larger than `max_message_bytes` go in several rounds: round k moves the k-th fraction of every
    per-peer slice through contiguous staging tensors."""
    import torch
    import torch.distributed as dist
    limit = MAX_MESSAGE_BYTES if max_message_bytes is None else max_message_bytes
    unit, dtype = (8, torch.int64) if width % 8 == 0 else (4, torch.int32) if width % 4 == 0 else (1, torch.uint8)
    if send.data_ptr() % unit or recv.data_ptr() % unit:
        unit, dtype = 1, torch.uint8
    k = width // unit
    world = len(send_counts)
    # every rank must run the same number of rounds: agree on the largest split anywhere
    biggest = max(list(send_counts) + list(recv_counts) + [0]) * width
    dev = send.device
    t = torch.tensor([biggest], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    rounds = max(1, -(-int(t.item()) // limit))
    if rounds == 1:
        dist.all_to_all_single(recv.view(dtype), send.view(dtype), output_split_sizes=[c * k for c in recv_counts],
                               input_split_sizes=[c * k for c in send_counts], group=group)
        return
    s_off = [0] * world
    r_off = [0] * world
    for j in range(1, world):
        s_off[j] = s_off[j - 1] + send_counts[j - 1]
        r_off[j] = r_off[j - 1] + recv_counts[j - 1]
    for rd in range(rounds):
        s_rng = [(c * rd // rounds, c * (rd + 1) // rounds) for c in send_counts]
        r_rng = [(c * rd // rounds, c * (rd + 1) // rounds) for c in recv_counts]
        stage_in = torch.cat([send[(s_off[j] + lo) * width:(s_off[j] + hi) * width] for j, (lo, hi) in enumerate(s_rng)])
        n_in = sum(hi - lo for lo, hi in r_rng)
        stage_out = torch.empty(n_in * width, dtype=torch.uint8, device=dev)
        dist.all_to_all_single(stage_out.view(dtype) if stage_out.data_ptr() % unit == 0 else stage_out,
                               stage_in.view(dtype) if stage_in.data_ptr() % unit == 0 else stage_in,
                               output_split_sizes=[(hi - lo) * (k if stage_out.data_ptr() % unit == 0 else width) for lo, hi in r_rng],
                               input_split_sizes=[(hi - lo) * (k if stage_in.data_ptr() % unit == 0 else width) for lo, hi in s_rng], group=group)
        o = 0
        for j, (lo, hi) in enumerate(r_rng):
            nb = (hi - lo) * width
            recv[(r_off[j] + lo) * width:(r_off[j] + hi) * width].copy_(stage_out[o:o + nb])
            o += nb


def gather_counts(count, group=None):
    """every rank learns every rank's row count (rank order)"""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    mine = torch.tensor([int(count)], dtype=torch.int64, device=dev)
    allc = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(allc, mine, group=group)
    return [int(x) for x in allc.cpu().tolist()]


def all_gather_bytes(send, recv, counts, width, group=None):
    """all-gather of one column: `send` = this rank's rows, `recv` = all rows in rank order (flat uint8 tensors,
    counts in rows).  Equal shards take one all_gather_into_tensor straight into `recv`; ragged shards are
    broadcast rank by rank into their slice (no staging copy either way)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    unit, dtype = (8, torch.int64) if width % 8 == 0 else (4, torch.int32) if width % 4 == 0 else (1, torch.uint8)
    if send.data_ptr() % unit or recv.data_ptr() % unit:
        unit, dtype = 1, torch.uint8
    if len(set(counts)) == 1 and counts[0] * width <= MAX_MESSAGE_BYTES:
        if counts[0]:
            dist.all_gather_into_tensor(recv.view(dtype), send.view(dtype), group=group)
        return
    off = 0
    step = max(1, MAX_MESSAGE_BYTES // width)   # rows per message
    for r in range(world):
        src = dist.get_global_rank(group, r) if group is not None else r
        for lo in range(0, counts[r], step):
            hi = min(counts[r], lo + step)
            piece = recv[(off + lo) * width:(off + hi) * width]
            if r == rank:
                piece.copy_(send[lo * width:hi * width])
            dist.broadcast(piece.view(dtype) if (piece.data_ptr() % unit == 0) else piece, src=src, group=group)
        off += counts[r]


def gather_key_ranges(probe_range, build_range, group=None):
    """every rank learns every rank's closed probe-key and build-key ranges (None = no rows) in ONE all-gather:
    returns (probe_ranges, build_ranges), rank order"""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    row = []
    for r in (probe_range, build_range):
        row += [int(r[0]), int(r[1]), 1] if r is not None else [0, 0, 0]
    mine = torch.tensor(row, dtype=torch.int64, device=dev)
    allr = torch.empty(6 * world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(allr, mine, group=group)
    v = allr.cpu().tolist()
    pr = [(v[6 * r], v[6 * r + 1]) if v[6 * r + 2] else None for r in range(world)]
    br = [(v[6 * r + 3], v[6 * r + 4]) if v[6 * r + 5] else None for r in range(world)]
    return pr, br


def nothing_crosses_ranks(probe_ranges, build_ranges):
    """True when no rank's build-key range meets another rank's probe-key bounds: every rank can decide this from the
    gathered ranges alone, so the pruned broadcast needs no further collective"""
    for i, b in enumerate(build_ranges):
        for j, p in enumerate(probe_ranges):
            if i != j and b is not None and p is not None and max(b[0], p[0]) <= min(b[1], p[1]):
                return False
    return True


def pruned_send_ranges(probe_ranges, my_build_range):
    """which slice of this rank's build keys each destination needs: the intersection of the destination's probe
    key range with this rank's build key range, or None.  A destination can only match build keys inside the
    [min, max] of ITS OWN probe keys — DataFusion's dynamic join filter (hash_join/shared_bounds.rs:277-284:
    build-side bounds prune the probe-side scan) applied in the other direction, to prune the broadcast."""
    out = []
    for pr in probe_ranges:
        if pr is None or my_build_range is None:
            out.append(None)
            continue
        lo, hi = max(pr[0], my_build_range[0]), min(pr[1], my_build_range[1])
        out.append((lo, hi) if lo <= hi else None)
    return out


def broadcast_build_moves_fewer_bytes(build_bytes, probe_bytes, world):
    """per-GPU received bytes: all-gather of the build side = B (N-1)/N, hash repartition of both = (B+P)(N-1)/N^2"""
    return build_bytes * world < build_bytes + probe_bytes


def route(hashes: np.ndarray, world: int) -> np.ndarray:
    """destination rank of a row = hash % world (BatchPartitioner hash arm, repartition/mod.rs:1111-1150)"""
    return (hashes % np.uint64(world)).astype(np.int64)
