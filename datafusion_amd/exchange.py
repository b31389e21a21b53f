"""Multi-GPU hash-repartition exchange = RepartitionExec(Partitioning::Hash) across GPUs
(physical-plan/src/repartition/mod.rs:1097-1150, :1626): each rank splits its rows by
hash(keys; seed 0) % world_size with the K10 partition kernel (contiguous per-destination
slices of one buffer per column), then one all-to-all(v) per column moves slice r of every
rank to rank r.  In-process tokio channels of the reference become RCCL over xGMI
(torch.distributed backend "nccl"); on CPU the same collective code runs over gloo in tests.

One process per GPU.  torch is plumbing only: it wraps the library's HBM buffers as tensors
(zero-copy, __cuda_array_interface__) so RCCL can send from / receive into them.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

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


def hash_exchange(table, keys, group=None, force=False):
    """DeviceTable -> DeviceTable holding every row (from all ranks) whose key hash routes here.
    force=True runs the partition + all-to-all path even for a single rank (used to exercise the
    RCCL plumbing on a 1-GPU box)."""
    import torch
    import torch.distributed as dist

    from . import _lib, ops
    from ._lib import Field, check
    from .table import DeviceTable

    world = dist.get_world_size(group)
    if world == 1 and not force:
        return table
    lib = _lib.load()
    parts = ops.partition(table, keys, world)
    send_counts = [p.num_rows for p in parts]
    recv_counts = exchange_counts(send_counts, group)
    total = sum(recv_counts)
    ncols = table.num_columns
    views0 = [parts[0].column_view(i) for i in range(ncols)]
    fields = (Field * ncols)(*[v.field for v in views0])
    names = (C.c_char_p * ncols)(*[v.name for v in views0])
    out = C.c_void_p()
    check(lib.dfgpu_table_alloc(ncols, fields, names, C.c_int64(total), C.byref(out)))
    result = DeviceTable(out)
    ops.sync()  # partition kernels ran on the library stream; RCCL runs on torch's
    n_send = sum(send_counts)
    for i in range(ncols):
        v = views0[i]
        if v.validity:
            raise _lib.DfgpuError("hash_exchange: nullable columns are not supported yet")
        w = _W[v.field.type]
        # partition outputs are consecutive slices of one buffer: partition 0's pointer is its start
        send = _as_tensor(v.data, n_send * w)
        rv = result.column_view(i)
        recv = _as_tensor(rv.data, total * w)
        all_to_all_bytes(send, send_counts, recv, recv_counts, w, group)
    torch.cuda.synchronize()
    for p in parts:
        p.free()
    return result


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


def broadcast_table(table, group=None, force=False):
    """DeviceTable -> DeviceTable holding the rows of ALL ranks in rank order: the build side of a
    PartitionMode::CollectLeft hash join (hash_join/exec.rs:1325-1328: one side collected whole, the probe side
    stays partitioned), as one all-gather per column over RCCL.  Chosen instead of hash-repartitioning both sides
    when it moves fewer bytes per GPU: build_bytes * N < build_bytes + probe_bytes."""
    import torch
    import torch.distributed as dist

    from . import _lib, ops
    from ._lib import Field, check
    from .table import DeviceTable

    world = dist.get_world_size(group)
    if world == 1 and not force:
        return table
    lib = _lib.load()
    counts = gather_counts(table.num_rows, group)
    total = sum(counts)
    ncols = table.num_columns
    views = [table.column_view(i) for i in range(ncols)]
    fields = (Field * ncols)(*[v.field for v in views])
    names = (C.c_char_p * ncols)(*[v.name for v in views])
    out = C.c_void_p()
    check(lib.dfgpu_table_alloc(ncols, fields, names, C.c_int64(total), C.byref(out)))
    result = DeviceTable(out)
    ops.sync()  # the producer kernels ran on the library stream; RCCL runs on torch's
    for i in range(ncols):
        v = views[i]
        if v.validity:
            raise _lib.DfgpuError("broadcast_table: nullable columns are not supported yet")
        w = _W[v.field.type]
        send = _as_tensor(v.data, table.num_rows * w)
        recv = _as_tensor(result.column_view(i).data, total * w)
        all_gather_bytes(send, recv, counts, w, group)
    torch.cuda.synchronize()
    return result


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


def _key_range(table, key):
    """(min, max) of an integer key column of a DeviceTable, None if it has no non-null row"""
    from . import ops
    lo, hi, n, _ = ops.column_minmax(table, key)
    return None if n == 0 else (lo, hi)


def pruned_broadcast_table(build, build_key, probe, probe_key, group=None, force=False, stats=None):
    """CollectLeft with the broadcast pruned by the destinations' probe-key bounds: rank r receives from every rank
    only the build rows whose key lies inside [min, max] of r's probe keys (a superset of what r can match, so the
    local join is unchanged).  Clustered inputs — TPC-H orders / lineitem in key order, any range-partitioned scan —
    move almost nothing; uniformly spread keys degrade to the full all-gather (same bytes as broadcast_table).
    One all-to-all(v) per column; `stats` (optional dict) receives the rows sent / received across ranks."""
    import pyarrow as pa
    import torch
    import torch.distributed as dist

    from . import _lib, ops
    from ._lib import Field, check
    from .expr import col, lit
    from .table import DeviceTable

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1 and not force:
        return build
    ktype = build.schema.field(build.schema.get_field_index(build_key)).type
    if ktype not in (pa.int64(), pa.int32()):
        return broadcast_table(build, group, force)
    lib = _lib.load()
    my_range = _key_range(build, build_key)
    probe_ranges, build_ranges = gather_key_ranges(_key_range(probe, probe_key), my_range, group=group)
    sends = pruned_send_ranges(probe_ranges, my_range)

    def part_for(sr):
        if sr is None:
            return build.slice(0, 0)
        if sr == my_range:
            return build.select(list(range(build.num_columns)))  # the bounds cover this whole shard: a zero-copy view, no filter pass
        return ops.filter(build, (col(build_key) >= lit(sr[0], ktype)).and_(col(build_key) <= lit(sr[1], ktype)))

    if nothing_crosses_ranks(probe_ranges, build_ranges):
        own = part_for(sends[rank])                             # the local shard (pruned to the local bounds) is the build side
        if stats is not None:
            stats.update(rows_sent_to_peers=0, rows_received_from_peers=0, build_rows_local=build.num_rows, build_rows_after_exchange=own.num_rows)
        return own
    parts = [part_for(sr) for sr in sends]
    send_counts = [p.num_rows for p in parts]
    recv_counts = exchange_counts(send_counts, group)
    total = sum(recv_counts)
    if stats is not None:
        stats.update(rows_sent_to_peers=sum(send_counts) - send_counts[rank], rows_received_from_peers=total - recv_counts[rank],
                     build_rows_local=build.num_rows, build_rows_after_exchange=total)
    nonempty = [p for p in parts if p.num_rows]
    packed = nonempty[0].select(list(range(nonempty[0].num_columns))) if len(nonempty) == 1 else (DeviceTable.concat(parts) if nonempty else build.slice(0, 0))
    ncols = build.num_columns
    views = [packed.column_view(i) for i in range(ncols)]
    fields = (Field * ncols)(*[v.field for v in views])
    names = (C.c_char_p * ncols)(*[v.name for v in views])
    out = C.c_void_p()
    check(lib.dfgpu_table_alloc(ncols, fields, names, C.c_int64(total), C.byref(out)))
    result = DeviceTable(out)
    ops.sync()
    n_send = sum(send_counts)
    for i in range(ncols):
        v = views[i]
        if v.validity:
            raise _lib.DfgpuError("pruned_broadcast_table: nullable columns are not supported yet")
        w = _W[v.field.type]
        send = _as_tensor(v.data, n_send * w)
        recv = _as_tensor(result.column_view(i).data, total * w)
        all_to_all_bytes(send, send_counts, recv, recv_counts, w, group)
    torch.cuda.synchronize()
    for p in parts:
        p.free()
    packed.free()
    return result


def broadcast_build_moves_fewer_bytes(build_bytes, probe_bytes, world):
    """per-GPU received bytes: all-gather of the build side = B (N-1)/N, hash repartition of both = (B+P)(N-1)/N^2"""
    return build_bytes * world < build_bytes + probe_bytes


def route(hashes: np.ndarray, world: int) -> np.ndarray:
    """destination rank of a row = hash % world (BatchPartitioner hash arm, repartition/mod.rs:1111-1150)"""
    return (hashes % np.uint64(world)).astype(np.int64)
