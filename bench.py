#!/usr/bin/env python3
"""bench.py — BASELINE.json metric: rows/sec + GB/s HBM of the TPC-H Q3 hash join
(orders ⋈ lineitem on orderkey, Q3 payload projection) at SF100 on 1/2/4/8 MI355X.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one full pass of the hot path over device-resident inputs:
  N = 1 : HashJoinExec build (collect_left_input) + probe (whole lineitem) -> output table.
  N > 1 : each rank holds a 1/N row range of both tables (what N scans of tables stored in key order produce; the scan
          boundaries of the two tables do not line up exactly: --shard-skew, default 1 % of a shard).  Two exchanges
          are timed, K steps each, and BOTH are on the JSON line under "exchanges" with the bytes that crossed per rank:
            repartition — PartitionMode::Partitioned as the reference plans this join (hash_join/exec.rs:1312-1324):
                          dfgpu_exchange_hash of BOTH sides (partition kernel + RCCL all-to-all(v) over xGMI), then
                          the local build + probe.  Every row crosses with probability (N-1)/N.
            pruned      — PartitionMode::CollectLeft with the build-side all-gather pruned by each rank's probe-key
                          bounds (dfgpu_exchange_broadcast_pruned): inputs clustered by key move only the rows
                          around the shard boundaries; keys spread uniformly degrade to the full all-gather.
          `value` is north_star's exchange — the RCCL all-to-all hash repartition of both sides — and the one a byte-counting
          planner would pick (broadcast the build side while B*N < B+P, the SF100 Q3 join up to N = 10, pruned by
          bounds) is the secondary entry, named in config.planner_choice; --exchange forces one.
          Total work is fixed (SF100) => "strong" scaling.
  --workload q1 / q3: BASELINE configs 4 and 5 — the whole TPC-H Q1 / Q3 plan per step (queries.q1 / q3: Partial
          aggregate -> hash exchange of the states -> FinalPartitioned; Q3's four repartitions in two phases), rows of
          the scanned tables per second.
value = (input rows summed over ranks) / max-over-ranks wall time of the K steps.
Inputs are generated on device (no dataset download possible) before the timed region.

Extra objects on the JSON line:
  roofline     — dominant kernel (join_probe_fused = k_join_probe_fused: lookup + output offsets +
                 materialisation in one pass; join_probe_placed = the same kernel placed by pre-computed tile
                 offsets when --probe-mode 0): algorithmic bytes per launch (probe columns read once
                 np*40 + build payload read once nb*8 + output written once nout*48) / average launch duration
                 measured with HIP events on the library stream, against the 8.0 TB/s HBM3E spec peak
                 (frac) and the measured 6.29 TB/s copy ceiling (frac_vs_copy_ceiling).
  ordered_output — the same step with the output in probe order (what the reference emits), with its own
                 roofline object.
  cpu_baseline — the CPU restatement (oracle/, kind "port") of DataFusion's plan for the same join — RepartitionExec(Hash) of
                 both sides, HashJoinExec(Partitioned), build_batch_from_indices of the five Q3 payload columns — timed on this
                 box's host cores on the GPU leg's own tables at the GPU leg's scale factor when host RAM allows (the SF is
                 stated; `speedup_vs_cpu_port` is only reported between equal workloads).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_COPY_CEILING_GBS = 6290.0  # MI355X_MICROARCH.md: measured float4 copy ceiling (79 % of spec)
TABLE_KINDS = {0: "hash_map", 1: "array_map", 2: "rank_map", 3: "radix_lds", 4: "flat_hash_map", 5: "flat_hash_map_16"}

# what the watchdog thread prints if a collective never returns (N > 1): the line as of the last finished measurement, the phase
# that was running and since when.  run_join / run_query update it through checkpoint().
WATCHDOG = {"line": None, "phase": "setup", "since": time.monotonic()}


def start_watchdog(args, rank, world):
    """N > 1 only: a rank stuck inside a collective (a peer died, RCCL never connected) cannot be helped from Python — but the run can
    still leave a record.  Every rank runs a timer thread (the main thread is inside a C call with the GIL released); when one phase
    has taken longer than --watchdog-s, rank 0 prints ONE JSON line — the line of the measurements that DID finish, or a line with
    value null — carrying `watchdog: {phase, seconds}`, and every rank leaves with os._exit (rank 0 first, the others 5 s later)."""
    if world == 1 or args.watchdog_s <= 0:
        return
    import threading

    def watch():
        while True:
            time.sleep(1.0)
            if WATCHDOG["phase"] == "done":
                return
            waited = time.monotonic() - WATCHDOG["since"]
            if waited > args.watchdog_s + (0 if rank == 0 else 5):
                line = WATCHDOG["line"]
                if rank == 0:
                    measured = line is not None
                    if line is None:
                        line = {"metric": "tpch_q3_hash_join_rows_per_sec" if args.workload == "join" else f"tpch_{args.workload}_rows_per_sec", "value": None,
                                "unit": "rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True,
                                "scaling": "strong", "vs_baseline": None, "data": "synthetic", "config": {"workload": args.workload, "sf": args.sf}}
                    line["watchdog"] = {"fired_in_phase": WATCHDOG["phase"], "after_s": round(waited, 1), "limit_s": args.watchdog_s,
                                        "measured_before_it_fired": measured}
                    sys.stdout.flush()
                    print(json.dumps(line), flush=True)
                    os._exit(0 if measured else 3)
                os._exit(3)
    threading.Thread(target=watch, daemon=True, name="bench-watchdog").start()


def checkpoint_phase(phase, line=None):
    WATCHDOG["phase"] = phase
    WATCHDOG["since"] = time.monotonic()
    if line is not None:
        WATCHDOG["line"] = line


BUILD_COLS = ["o_orderdate", "o_shippriority"]
PROBE_COLS = ["l_orderkey", "l_extendedprice", "l_discount"]


def algorithmic_bytes(nb, np_, nout):
    """read each referenced input column once + write the output once:
    build {o_orderkey 8, o_orderdate 4, o_shippriority 4} = 16 B/row, probe {l_orderkey 8,
    l_extendedprice 16, l_discount 16} = 40 B/row, output 48 B/row"""
    return nb * 16 + np_ * 40 + nout * 48


def cpu_quota():
    """CPUs' worth of time the container may use (cgroup v2 cpu.max: "<quota> <period>"), None when unlimited or unknown: a
    256-thread host behind a 16-CPU quota runs 16 threads at a time whatever os.cpu_count() says"""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if quota == "max" else round(int(quota) / int(period), 2)
    except Exception:  # noqa: BLE001
        return None


def host_memory_available():
    """bytes of host RAM this process may still take: MemAvailable, capped by the cgroup's memory.max minus memory.current"""
    avail = None
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable:"):
                avail = int(ln.split()[1]) * 1024
    except OSError:
        pass
    try:
        mx = open("/sys/fs/cgroup/memory.max").read().strip()
        if mx != "max":
            room = int(mx) - int(open("/sys/fs/cgroup/memory.current").read())
            avail = room if avail is None else min(avail, room)
    except (OSError, ValueError):
        pass
    return avail


def _host_q3_columns(orders, lineitem):
    """the Q3 join's referenced columns of two device tables as host numpy arrays (Decimal128 as (n, 2) uint64, low word first)"""
    import numpy as np
    o = orders.select(["o_orderkey", "o_orderdate", "o_shippriority"]).to_arrow()
    l = lineitem.select(["l_orderkey", "l_extendedprice", "l_discount"]).to_arrow()

    def dec(c):
        a = c.combine_chunks()
        return np.frombuffer(a.buffers()[1], dtype=np.uint64)[2 * a.offset:2 * (a.offset + len(a))].reshape(-1, 2)
    import pyarrow as pa
    return (o.column("o_orderkey").to_numpy(), o.column("o_orderdate").cast(pa.int32()).to_numpy(), o.column("o_shippriority").to_numpy(),
            l.column("l_orderkey").to_numpy(), dec(l.column("l_extendedprice")), dec(l.column("l_discount"))), (o, l)


def cpu_baseline(gpu_sf, max_sf, hw_threads, orders_dev=None, lineitem_dev=None):
    """oracle leg (kind "port"): the SAME join as the GPU leg, as the reference plans it on a CPU — RepartitionExec(Hash) of every
    column of both sides -> HashJoinExec(Partitioned), one thread per partition, 8192-row probe batches, build_batch_from_indices of
    the five Q3 payload columns per batch (oracle/dforacle.c orc_partitioned_q3_join) — on the GPU leg's own tables copied to the
    host, at the GPU leg's scale factor when host RAM allows (inputs + their repartitioned copies: 2 x 26.4 GB at SF100), else at the
    largest of SF50 / 30 / 10 that fits.  The partition count is chosen on an SF10 sample first (one partition per core is not the
    fastest setting on a many-core NUMA host behind a CPU quota); the big run uses that count once."""
    import numpy as np

    from datafusion_amd import ops
    from oracle import oracle
    quota = cpu_quota()
    cores = max(1, int(min(hw_threads, quota) if quota else hw_threads))   # threads that can actually run at a time
    avail = host_memory_available()
    need = lambda sf: int(sf * 1.0e6 * (1.5 * 16 + 6.0 * 40) * 2.6)   # inputs + repartitioned copies + Arrow export buffers, 30 % slack
    sample_sf = next((sf for sf in (gpu_sf, 50.0, 30.0, 10.0) if sf <= min(gpu_sf, max_sf) and (avail is None or need(sf) < avail)), min(gpu_sf, 1.0))
    small_sf = min(10.0, sample_sf)

    def tables(sf):
        if sf == gpu_sf and orders_dev is not None:
            return orders_dev, lineitem_dev, False
        return ops.tpch_orders(sf), ops.tpch_lineitem(sf), True
    o, l, own = tables(small_sf)
    cols, keep = _host_q3_columns(o, l)
    if own:
        o.free()
        l.free()
    tried, best_t, best = {}, None, None
    for threads in sorted({cores, 2 * cores, max(1, cores // 2), min(hw_threads, 32)}, reverse=True):
        t0 = time.perf_counter()
        rows, chk_small = oracle.partitioned_q3_join(*cols, threads)
        dt = time.perf_counter() - t0
        assert rows == len(cols[3])
        tried[threads] = round((len(cols[0]) + len(cols[3])) / dt)
        if best is None or dt < best:
            best, best_t = dt, threads
    nb, np_, dt_big = len(cols[0]), len(cols[3]), best
    if sample_sf > small_sf:
        del cols, keep
        o, l, own = tables(sample_sf)
        cols, keep = _host_q3_columns(o, l)
        if own:
            o.free()
            l.free()
        t0 = time.perf_counter()
        rows, _chk = oracle.partitioned_q3_join(*cols, best_t)
        dt_big = time.perf_counter() - t0
        nb, np_ = len(cols[0]), len(cols[3])
        assert rows == np_
    out = {"value": (nb + np_) / dt_big, "unit": "rows/s", "cores": min(best_t, cores), "threads": best_t, "kind": "port", "sf": sample_sf,
           "same_workload_as_gpu_leg": bool(sample_sf == gpu_sf),
           "sample": f"orders x lineitem at SF{sample_sf:g} with the Q3 payload ({nb} build rows x 16 B, {np_} probe rows x 40 B, {np_} output rows x 48 B): "
                     f"RepartitionExec(Hash) of all columns of both sides -> HashJoinExec(Partitioned) -> build_batch_from_indices per 8192-row "
                     f"probe batch (output batches produced, checksummed and dropped like the reference's stream), {best_t} partitions/threads "
                     f"chosen on SF{small_sf:g} (rows/s by thread count: {tried}); host has {hw_threads} hardware threads, cgroup CPU quota {quota}, "
                     f"{'unknown' if avail is None else round(avail / 2**30)} GiB of host RAM available", "cpu_quota": quota}
    try:  # independent production CPU engine: Arrow Acero hash join WITH the payload, on the SF10-sized sample
        import pyarrow as pa
        pa.set_cpu_count(cores)
        so, sl, sown = tables(small_sf)
        bt = so.select(["o_orderkey", "o_orderdate", "o_shippriority"]).to_arrow()
        pt = sl.select(["l_orderkey", "l_extendedprice", "l_discount"]).to_arrow()
        if sown:
            so.free()
            sl.free()
        t_best = None
        for _ in range(2):
            t0 = time.perf_counter()
            j = pt.join(bt, keys="l_orderkey", right_keys="o_orderkey", join_type="inner")
            dt = time.perf_counter() - t0
            assert j.num_rows == pt.num_rows
            t_best = dt if t_best is None else min(t_best, dt)
        out["acero_rows_per_s"] = (bt.num_rows + pt.num_rows) / t_best
        out["acero_sample"] = f"SF{small_sf:g}, the same five payload columns, {cores} threads"
    except Exception as e:  # noqa: BLE001 - the Acero leg is a sanity bound, never required
        out["acero_error"] = str(e)[:200]
    return out


def setup_dist(args):
    """rank / world from the launcher's environment; torch.distributed (backend nccl = RCCL) carries the barrier, the
    max-over-ranks timing and RCCL's bootstrap id for the library's own communicator — the data path is dfgpu_exchange_*"""
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and args.gpus > 1:
        raise SystemExit("--gpus N needs N ranks: main() starts them itself when no launcher did")
    # DFGPU_BENCH_REHEARSAL=1: every rank on GPU 0, a gloo group and the host transport under dfgpu_exchange_* — the N > 1 control
    # flow (shards, both exchanges, max-over-ranks timing, the JSON line) on a box with one GPU.  Its numbers mean nothing.
    rehearsal = os.environ.get("DFGPU_BENCH_REHEARSAL", "0") == "1"
    if rehearsal:
        local_rank = 0
        os.environ["LOCAL_RANK"] = "0"   # (what the library binds to when nothing else is said)
    torch.cuda.set_device(local_rank)
    from datafusion_amd import _lib
    _lib.init(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    return rank, world, local_rank, dist


def timed(step, steps, barrier):
    barrier()
    t0 = time.perf_counter()
    last = None
    for _ in range(steps):
        last = step()
    barrier()
    return time.perf_counter() - t0, last


def max_over_ranks(dist, dt, *sums):
    """(max dt over ranks, sums over ranks)"""
    import torch
    if dist is None:
        return dt, [int(x) for x in sums]
    where = "cpu" if "gloo" in str(dist.get_backend()) else "cuda"
    mx = torch.tensor([dt], dtype=torch.float64, device=where)
    tot = torch.tensor([float(x) for x in sums], dtype=torch.float64, device=where)
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    return float(mx[0]), [int(x) for x in tot]


def shard_bounds(n_orders, rank, world, skew):
    """this rank's order range for `orders`, and the (shifted) order range whose lines it holds of `lineitem`: two scans of
    tables stored in key order cut them at row counts that do not fall on the same keys"""
    def cut(r):
        return n_orders * r // world
    b, e = cut(rank), cut(rank + 1)
    shift = int(skew * (n_orders // world)) if world > 1 else 0
    lb = b + shift if rank > 0 else 0
    le = e + shift if rank < world - 1 else n_orders
    return (b, e), (lb, le)


def run_join(args, rank, world, dist):
    import torch

    from datafusion_amd import ops, tpch
    from datafusion_amd.exchange import broadcast_build_moves_fewer_bytes, comm_for
    n_orders = tpch.n_orders(args.sf)
    (b, e), (lb, le) = shard_bounds(n_orders, rank, world, args.shard_skew)
    orders = ops.tpch_orders(args.sf, b, e).select(["o_orderkey", "o_orderdate", "o_shippriority"])
    lineitem = ops.tpch_lineitem(args.sf, lb, le).select(["l_orderkey", "l_extendedprice", "l_discount"])
    nb_local, np_local = orders.num_rows, lineitem.num_rows
    ops.sync()
    forced = world == 1 and args.exchange != "auto"   # one-rank rehearsal of the N > 1 code on a 1-GPU box
    comm = comm_for(None, force=forced) if (world > 1 or forced) else None

    def barrier():
        ops.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    def make_step(exchange, probe_mode):
        def step():
            o, l = orders, lineitem
            if exchange == "pruned":
                o = comm.broadcast_pruned(orders, "o_orderkey", lineitem, "l_orderkey")
            elif exchange == "broadcast":
                o = comm.broadcast(orders)
            elif exchange == "repartition":
                o = comm.hash_exchange(orders, ["o_orderkey"])
                l = comm.hash_exchange(lineitem, ["l_orderkey"])
            elif exchange == "repartition_stream":
                # the same exchange, streamed (dfgpu_exchange_hash_stream_*): the build side's chunks go into the join builder as they
                # land, the probe side's chunks are probed as they land — partition kernel, all-to-all(v) and build / probe of
                # neighbouring chunks overlap on the library's own threads and streams
                builder = ops.JoinBuilder([0], probe_mode=probe_mode)
                for chunk in comm.hash_exchange_stream(orders, ["o_orderkey"], args.exchange_chunks):
                    builder.push(chunk)
                    chunk.free()
                ht = builder.finish()
                n_out = 0
                for chunk in comm.hash_exchange_stream(lineitem, ["l_orderkey"], args.exchange_chunks):
                    out = ht.probe(chunk, ["l_orderkey"], "Inner", BUILD_COLS, PROBE_COLS)
                    n_out += out.num_rows
                    out.free()
                    chunk.free()
                res = (n_out, ht.info())
                ht.free()
                return res
            # join-table choice follows the library default (direct-address down to key density 1/64, see
            # DFGPU_DEFAULT_MIN_KEY_DENSITY in include/dfgpu.h): hash routing leaves each rank 1/N of the keys over the
            # same key range (density 0.25/N)
            ht = ops.JoinHashTable(o, ["o_orderkey"], probe_mode=probe_mode)
            out = ht.probe(l, ["l_orderkey"], "Inner", BUILD_COLS, PROBE_COLS)
            res = (out.num_rows, ht.info())
            out.free()
            ht.free()
            if o is not orders:
                o.free()
            if l is not lineitem:
                l.free()
            return res
        return step

    def measure(exchange, probe_mode, profile):
        step = make_step(exchange, probe_mode)
        for _ in range(args.warmup):
            step()
        if comm is not None:
            comm.stats(reset=True)
        if profile:
            ops.profile_enable(True)
            ops.profile_reset()
        dt, (n_out, info) = timed(step, args.steps, barrier)
        st = ops.profile_stats() if profile else {}
        if profile:
            ops.profile_enable(False)
        xs = comm.stats(reset=True) if comm is not None else None
        dt, (nb, np_, nout) = max_over_ranks(dist, dt, nb_local, np_local, n_out)
        return {"dt": dt, "nb": nb, "np": np_, "nout": nout, "info": info, "stats": st, "xstats": xs}

    if world == 1 and not forced:
        primary = "none"
    elif args.exchange != "auto":
        primary = args.exchange
    else:
        # `value` = north_star's exchange: hash repartition of both sides by key, RCCL all-to-all(v) over xGMI (dfgpu_exchange_hash).
        # The exchange a byte-counting planner would pick instead for these sizes (broadcast the build side while
        # B*N < B+P, pruned by the destinations' probe-key bounds) is timed as the secondary entry of "exchanges" and named in
        # config.planner_choice — on clustered shards it moves almost nothing, which is not what the scaling curve is about.
        primary = "repartition_stream" if args.exchange_chunks > 1 else "repartition"
    planner_choice = None
    if world > 1:
        tot = max_over_ranks(dist, 0.0, nb_local, np_local)[1]
        planner_choice = "pruned" if broadcast_build_moves_fewer_bytes(tot[0] * 16, tot[1] * 40, world) else "repartition"
    others, exchange_errors = {}, {}

    def make_line(m, primary, ordered, final):
        """the JSON line from what has been measured so far (final=False: the watchdog's partial line, no CPU leg)"""
        nb, np_, nout, dt, stats, info = m["nb"], m["np"], m["nout"], m["dt"], m["stats"], m["info"]
        ms_per_step = dt / args.steps * 1e3
        rows_per_s = (nb + np_) / (dt / args.steps)
        alg = algorithmic_bytes(nb, np_, nout)

        def roofline_of(st):
            """the dominant kernel of a step: algorithmic bytes per launch (probe columns once + build payload
            once + output once, computed by the library per launch) / its average HIP-event duration on the library stream"""
            name = next((k for k in ("join_probe_fused", "join_probe_placed", "join_probe_materialize") if k in st), None)
            if name is None or not st[name]["calls"]:
                return None
            d = st[name]
            avg_ms = d["total_ms"] / d["calls"]
            per_launch = d["bytes"] / d["calls"]
            achieved = per_launch / (avg_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "frac_vs_copy_ceiling": round(achieved / HBM_COPY_CEILING_GBS, 4),
                    "traffic": None, "avg_launch_ms": round(avg_ms, 4), "algorithmic_bytes_per_launch": int(per_launch)}
            attach_traffic(roof, name, {"build_rows": nb, "probe_rows": np_, "output_rows": nout}, world)
            return roof

        def kernel_table(st):
            return {k: {"calls": v["calls"], "avg_ms": round(v["total_ms"] / max(1, v["calls"]), 4)} for k, v in st.items()}

        def exchange_summary(mm):
            """one exchange flavour: its step time, what crossed per step (rank 0's view), and — so that a first run on real multi-GPU
            hardware diagnoses itself — what the transport reports and what the wires allow: xGMI is point-to-point, 7 links x ~153 GB/s
            per GPU (MI355X_MICROARCH.md / SURVEY 8e), so an all-to-all(v) in which rank 0 sends B bytes to its N - 1 peers cannot
            finish before max(bytes to one peer) / 153 GB/s; the same bound for what it receives"""
            per_step = {k: v // args.steps for k, v in mm["xstats"].items()} if mm["xstats"] else None
            out = {"ms_per_step": round(mm["dt"] / args.steps * 1e3, 3), "rows_per_s": (mm["nb"] + mm["np"]) / (mm["dt"] / args.steps),
                   "join_table": TABLE_KINDS[mm["info"].table_kind], "crossed_per_step_rank0": per_step}
            if mm["stats"]:
                # the step's three phases as the library's HIP events saw them on rank 0 (summed kernel time per step; under the streamed
                # exchange they run on three streams at once, so their sum may exceed the step): partition kernels, the all-to-all(v), the join
                def phase(pred):
                    return round(sum(v["total_ms"] for k, v in mm["stats"].items() if pred(k)) / args.steps, 3)
                ph = {"partition_ms": phase(lambda k: k.startswith("partition") or k == "scan_u32"), "exchange_ms": phase(lambda k: k.startswith("exchange")),
                      "join_ms": phase(lambda k: k.startswith("join") or k in ("scan_mask_popcounts", "column_minmax", "gather", "concat"))}
                out["phases_ms_rank0"] = ph
                out["phases_sum_ms"] = round(sum(ph.values()), 3)
                out["longest_phase_ms"] = max(ph.values())
                out["step_over_longest_phase"] = round(out["ms_per_step"] / max(ph.values()), 3) if max(ph.values()) > 0 else None
            if comm is not None:
                out.update(comm.transport_info())
            if per_step and world > 1:
                link_gbs = 153.0
                sent, recv = per_step.get("bytes_sent_to_peers", 0), per_step.get("bytes_received_from_peers", 0)
                per_link = max(sent, recv) / (world - 1)          # an even all-to-all spreads a rank's bytes over its N - 1 links
                out["bytes_per_link_per_step_rank0"] = int(per_link)
                out["predicted_exchange_ms_at_153_GBps_per_link"] = round(per_link / (link_gbs * 1e9) * 1e3, 3)
                ex_ms = sum(v["total_ms"] for k, v in mm["stats"].items() if k.startswith("exchange")) / args.steps if mm["stats"] else None
                out["measured_exchange_ms"] = round(ex_ms, 3) if ex_ms else None
            return out

        parallelism = {"none": "single GPU",
                       "repartition_stream": f"Partitioned x{world}: hash repartition of both sides STREAMED in {args.exchange_chunks} chunks (dfgpu_exchange_hash_stream: partition "
                                             "kernel, RCCL all-to-all(v) and the join's build / probe of neighbouring chunks overlap)",
                       "pruned": f"CollectLeft x{world}: build-side all-gather pruned by each rank's probe-key bounds (dfgpu_exchange_broadcast_pruned, RCCL send/recv), probe side stays in place",
                       "broadcast": f"CollectLeft x{world}: RCCL all-gather of the build side (dfgpu_exchange_broadcast), probe side stays in place",
                       "repartition": f"Partitioned x{world}: hash repartition of both sides (dfgpu_exchange_hash: partition kernel + RCCL all-to-all(v))"}[primary]
        line = {
            "metric": "tpch_q3_hash_join_rows_per_sec", "value": rows_per_s, "unit": "rows/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "int64 keys / decimal128 payload", "data": "synthetic",
            "config": {"workload": f"INNER hash-join orders⋈lineitem on o_orderkey, TPC-H SF{args.sf:g}, Q3 payload "
                                   "(o_orderdate,o_shippriority,l_orderkey,l_extendedprice,l_discount), device-resident inputs",
                       "build_rows": nb, "probe_rows": np_, "output_rows": nout,
                       "join_table": TABLE_KINDS[info.table_kind],
                       "probe": {0: "placed_ordered", 1: "placed_ordered", 2: "single_pass_ordered_lookback", 3: "single_pass_unordered"}[args.probe_mode],
                       "parallelism": parallelism, "exchange": primary, "planner_choice": planner_choice, "shard_skew": args.shard_skew if world > 1 else None},
            "algorithmic_gb_per_s": round(alg / (dt / args.steps) / 1e9, 1),
            "hbm_frac_whole_step": round(alg / (dt / args.steps) / 1e9 / (HBM_PEAK_GBS * world), 4),
            "roofline": roofline_of(stats), "kernels": kernel_table(stats),
        }
        if world > 1 or forced:
            line["exchanges"] = {primary: exchange_summary(m), **{k: exchange_summary(v) for k, v in others.items()}}
            if exchange_errors:
                line["exchange_errors"] = exchange_errors
        if ordered is not None:
            # the same step with the output in probe order, exactly as the reference emits it (hash_join/exec.rs:3349):
            # tile counts -> scan -> the fused kernel with known tile offsets
            oms = ordered["dt"] / args.steps * 1e3
            line["ordered_output"] = {"probe": "placed (tile counts -> scan -> fused materialise)", "ms_per_step": round(oms, 3),
                                      "rows_per_s": (nb + np_) / (oms * 1e-3),
                                      "hbm_frac_whole_step": round(alg / (oms * 1e-3) / 1e9 / (HBM_PEAK_GBS * world), 4),
                                      "roofline": roofline_of(ordered["stats"]), "kernels": kernel_table(ordered["stats"])}
        if final and not args.no_cpu and world == 1:  # the CPU baseline is reported by the single-GPU run only
            threads = os.cpu_count() or 1
            line["cpu_baseline"] = cpu_baseline(args.sf, args.sf if args.cpu_sf is None else args.cpu_sf, threads, orders, lineitem)
            # a ratio only between equal workloads: the same tables, the same payload, the same scale factor
            line["speedup_vs_cpu_port"] = round(rows_per_s / line["cpu_baseline"]["value"], 1) if line["cpu_baseline"]["same_workload_as_gpu_leg"] else None
        return line

    def checkpoint(m, primary, phase):
        """rank 0 keeps the line as it would be printed NOW: if a later exchange hangs, the watchdog prints this one"""
        checkpoint_phase(phase, make_line(m, primary, None, False) if (rank == 0 and m is not None) else None)
    if world > 1 and args.exchange == "auto":
        # First the blocking exchange — the shortest code path through RCCL, so that a first run on a real multi-GPU node has a result in
        # hand — then the streamed form of the same exchange and the planner's alternative, each allowed to fail with an error (the same
        # error on every rank: a rank that hangs inside a collective is the watchdog's case).  `value` = north_star's hash repartition
        # at its best: the streamed form when it ran and was faster (dt is the maximum over ranks, so every rank decides alike).
        checkpoint(None, None, "repartition")
        m = measure("repartition", args.probe_mode, True)
        primary = "repartition"
        for ex in (("repartition_stream",) if args.exchange_chunks > 1 else ()) + ("pruned",):
            checkpoint(m, primary, ex)
            if exchange_errors:
                # an exchange that failed part-way may have left the ranks out of step inside the communicator: nothing more runs on it
                exchange_errors[ex] = "skipped: an earlier exchange failed on this communicator"
                continue
            try:
                others[ex] = measure(ex, args.probe_mode, True)
            except Exception as e:  # noqa: BLE001
                exchange_errors[ex] = repr(e)[:300]
        if "repartition_stream" in others and others["repartition_stream"]["dt"] < m["dt"]:
            others["repartition"], m, primary = m, others.pop("repartition_stream"), "repartition_stream"
    else:
        checkpoint(None, None, primary)
        m = measure(primary, args.probe_mode, True)
        if forced and primary == "repartition_stream":
            checkpoint(m, primary, "repartition")
            others["repartition"] = measure("repartition", args.probe_mode, True)
    # secondary, outside the contract's timed region: the same step with the output in probe order
    checkpoint(m, primary, "ordered_output")
    ordered = measure(primary, 0, True) if (args.probe_mode == 3 and world == 1) else None
    checkpoint_phase("done")
    line = make_line(m, primary, ordered, True) if rank == 0 else None
    orders.free()
    lineitem.free()
    return line


# ProfileScope name (the library's HIP-event table) -> the device kernel's name as rocprofv3 prints it (prefix match)
DEVICE_KERNEL_OF = {"agg_fused_jit": "agg_node", "agg_fused_tile": "k_agg_fused_tile", "join_probe_tile_counts": "k_join_tile_counts",
                    "join_probe_listed": "k_join_emit_listed", "join_probe_fused": "k_join_probe_fused", "join_probe_placed": "k_join_probe_fused",
                    "cmp": "k_cmp", "compact": "k_compact", "agg_fused_global": "k_agg_fused", "agg_fused_lds": "k_agg_fused"}


def attach_traffic(roof, name, workload_key, world):
    """HBM bytes per launch from the committed PMC passes (scripts/profile.sh -> profiles/traffic.json): this workload, this device
    kernel, and the kernel's SOURCE as it was when the passes were taken (a kernel edit without a fresh PMC pass reports no traffic)"""
    try:
        import hashlib
        for tr in json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["kernels"]:
            if tr["kernel"] != name or any(tr["workload"].get(k) != v for k, v in workload_key.items()) or world != 1:
                continue
            cur = hashlib.sha256(open(os.path.join(ROOT, tr.get("kernel_source", "datafusion_amd/csrc/join.hip")), "rb").read()).hexdigest()[:16]
            if tr.get("kernel_source_sha16") != cur:
                roof["traffic_note"] = f"the PMC passes in profiles/{tr['source']} (commit {tr.get('commit')}) predate the kernel's source: not attached"
                continue
            roof["traffic"] = tr["traffic_bytes_per_launch"]
            roof["traffic_source"] = (f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of {tr['device_kernel']} at commit {tr.get('commit')}, "
                                      f"profiles/{tr['source']}")
    except (OSError, KeyError, ValueError, TypeError):
        pass
    return roof


def q3_algorithmic_table(n_customer, n_orders, n_lineitem, st):
    """sum over the plan's operators of (referenced input column bytes + output bytes), intermediate row counts
    as measured (queries.q3's `stats`).  c_mktsegment is a 1-byte dictionary code here (the string layout in use)."""
    c, semi, j, g = st["customer_filtered"], st["semi_join"], st["join"], st["groups"]
    rows = [("FilterExec customer (c_mktsegment = BUILDING) -> [c_custkey]", n_customer * (8 + 1), c * 8),
            ("FilterExec orders (o_orderdate < 1995-03-15) + HashJoinExec RightSemi (c_custkey = o_custkey) -> [o_orderkey, o_orderdate, o_shippriority]",
             c * 8 + n_orders * (8 + 8 + 4 + 4), semi * 16),
            ("FilterExec lineitem (l_shipdate > 1995-03-15) + HashJoinExec Inner (o_orderkey = l_orderkey) -> Q3 payload",
             semi * 16 + n_lineitem * (8 + 16 + 16 + 4), j * 48),
            ("AggregateExec gby [l_orderkey, o_orderdate, o_shippriority] SUM(l_extendedprice * (1 - l_discount))", j * 48, g * (8 + 4 + 4 + 16)),
            ("SortExec TopK(10) [revenue DESC, o_orderdate]", g * (16 + 4), 10 * 32)]
    return [{"operator": o, "input_bytes": int(i), "output_bytes": int(w)} for o, i, w in rows]


def cpu_baseline_query(workload, gpu_sf, hw_threads, device_tables=None, max_sf=None, budget_s=150.0):
    """oracle leg (kind "port") of configs 4 / 5: the reference's pinned plan (q1.slt.part:42-58 / q3.slt.part:44-76) run with the CPU
    restatement's operators (oracle/plans.py over oracle/oracle.py, oracle/dforacle.c) on this box's host cores, the way DataFusion runs
    it: the scan cut into `target_partitions` row ranges, one thread per partition — Partial aggregate / filter per partition,
    RepartitionExec(Hash) between the stages, FinalPartitioned aggregate / partitioned joins per hash partition, a merge of the
    per-partition sorted runs.  It runs on the GPU leg's OWN tables copied to the host, at the GPU leg's scale factor, when host RAM
    holds them and their intermediates (Q1 SF100: 42 GB + 39 GB; Q3 SF300: 91 GB + 2 x 39 GB) and SF1's time projects to less than
    `budget_s` (the projection overestimates: SF1's run is mostly fixed costs — Q1 SF100 takes ~20 s, Q3 SF300 ~13 s on 16 threads); else on the largest of SF 100 / 30 / 10 / 3 / 1 that does (same_workload_as_gpu_leg says which)."""
    from datafusion_amd import ops
    from oracle import plans
    quota = cpu_quota()
    cores = max(1, int(min(hw_threads, quota) if quota else hw_threads))
    P = cores
    avail = host_memory_available()
    per_sf = (70 * 6.0e6 * 2.2) if workload == "q1" else ((9 * 0.15e6 + 24 * 1.5e6 + 44 * 6.0e6) * 2.4)   # tables + filtered + repartitioned copies

    q1_cols = ["l_quantity", "l_extendedprice", "l_discount", "l_tax", "l_returnflag", "l_linestatus", "l_shipdate"]

    def fits(sf):
        return avail is None or sf * per_sf < 0.8 * avail

    def host_tables(sf):
        """the sample's tables: the GPU leg's own when the scale factors agree, else from the same DEVICE generator, copied to the host"""
        if device_tables is not None and sf == gpu_sf:
            return tuple(t.to_arrow() for t in device_tables)
        gens = ((ops.tpch_lineitem, q1_cols),) if workload == "q1" else ((ops.tpch_customer, ["c_custkey", "c_mktsegment"]),
                (ops.tpch_orders, ["o_orderkey", "o_custkey", "o_orderdate", "o_shippriority"]), (ops.tpch_lineitem, ["l_orderkey", "l_extendedprice", "l_discount", "l_shipdate"]))
        out = []
        for g, cols in gens:
            t = g(sf)
            s = t.select(cols)
            t.free()
            out.append(s.to_arrow())
            s.free()
        return tuple(out)

    def once(sf):
        ts = host_tables(sf)
        st = {}
        t0 = time.perf_counter()
        out = plans.run_q1(*ts, P, st) if workload == "q1" else plans.run_q3(*ts, P, st)
        dt = time.perf_counter() - t0
        return dt, sum(t.num_rows for t in ts), out, st
    dt, rows, out, st = once(min(1.0, gpu_sf))
    sf = min(1.0, gpu_sf)
    cap = gpu_sf if max_sf is None else min(gpu_sf, max_sf)
    for cand in (gpu_sf, 100.0, 30.0, 10.0, 3.0):
        if 1.0 < cand <= cap and dt * cand <= budget_s and fits(cand):
            sf = cand
            dt, rows, out, st = once(sf)
            break
    origin = "the GPU leg's own tables" if (device_tables is not None and sf == gpu_sf) else "tables from the GPU leg's generator"
    return {"value": rows / dt, "unit": "rows/s", "cores": cores, "threads": P, "kind": "port", "sf": sf, "seconds": round(dt, 2),
            "same_workload_as_gpu_leg": bool(sf == gpu_sf), "cpu_quota": quota, "intermediate_rows": st,
            "result": out,
            "sample": f"TPC-H {workload.upper()} at SF{sf:g} ({rows} scanned rows, {out.num_rows} output rows), "
                      f"{origin} copied to "
                      f"the host: the reference's pinned plan on the oracle's operators (oracle/plans.py), "
                      f"{P} partitions / threads (host has {hw_threads} hardware threads, cgroup CPU quota {quota}, "
                      f"{'unknown' if avail is None else round(avail / 2**30)} GiB of host RAM available)"}


def run_query(args, rank, world, dist):
    """BASELINE config 4 (TPC-H Q1, grouped hash aggregate, 1 -> 8 GPUs with the hash repartition of the partial states) and
    config 5 (TPC-H Q3 end to end: 3-way join + aggregate + top-k; four repartitions in two exchange phases at N > 1)"""
    import torch

    from datafusion_amd import ops, queries, tpch
    n_orders = tpch.n_orders(args.sf)
    b, e = n_orders * rank // world, n_orders * (rank + 1) // world
    # only the columns the plan references stay resident (the generator's other columns are dropped with the full table: SF300's full
    # lineitem alone is ~200 GB)
    if args.workload == "q1":
        lineitem = ops.tpch_lineitem(args.sf, b, e).select(["l_quantity", "l_extendedprice", "l_discount", "l_tax", "l_returnflag", "l_linestatus", "l_shipdate"])
        tables = [lineitem]
    else:
        nc = tpch.n_customers(args.sf)
        lineitem = ops.tpch_lineitem(args.sf, b, e).select(["l_orderkey", "l_extendedprice", "l_discount", "l_shipdate"])
        tables = [ops.tpch_customer(args.sf, nc * rank // world, nc * (rank + 1) // world).select(["c_custkey", "c_mktsegment"]),
                  ops.tpch_orders(args.sf, b, e).select(["o_orderkey", "o_custkey", "o_orderdate", "o_shippriority"]), lineitem]
    rows_local = sum(t.num_rows for t in tables)
    bytes_local = sum(t.nbytes() for t in tables)
    ops.sync()

    def barrier():
        ops.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    group = None  # the default group: queries.* route their RepartitionExecs through exchange.hash_exchange -> dfgpu_exchange_hash
    q3_stats = {}

    # Q1's fused node is planned ONCE, here (expressions lowered to the C ABI, AVG return types): the timed steps execute a held plan, as a
    # DataFusion PhysicalPlan executed repeatedly would (`config.plan` says so; Q3's operators take their expressions per call)
    q1_plan = queries.plan_q1(lineitem, group) if args.workload == "q1" else None

    def step():
        out = queries.q1(lineitem, group, plan=q1_plan) if args.workload == "q1" else queries.q3(*tables, group=group)
        n = out.num_rows
        out.free()
        return n
    checkpoint_phase(f"{args.workload} warm-up")
    for _ in range(args.warmup):
        step()
    checkpoint_phase(f"{args.workload} timed steps")
    if args.workload == "q3":   # the intermediate row counts of the algorithmic-bytes table (8d config 5), outside the timed region
        res = queries.q3(*tables, group=group, stats=q3_stats)
    else:
        res = queries.q1(lineitem, group, plan=q1_plan)
    gpu_result = res.to_arrow()      # what the CPU leg's result is compared with (outside the timed region)
    res.free()
    comm = None
    if world > 1:
        from datafusion_amd.exchange import comm_for
        comm = comm_for(group)
        comm.stats(reset=True)
    ops.profile_enable(True)
    ops.profile_reset()
    ops.metrics_reset()
    dt, n_out = timed(step, args.steps, barrier)
    stats = ops.profile_stats()
    launch_recs = {k: ops.profile_launches(k) for k, v in stats.items() if v["calls"] > args.steps}
    alg_local = ops.metrics()["hbm_bytes_algorithmic"] // args.steps   # sum of the kernels' algorithmic bytes of one step, as the library counts them
    ops.profile_enable(False)
    xs = comm.stats(reset=True) if comm is not None else None
    dt, (rows, nbytes, alg_kernels) = max_over_ranks(dist, dt, rows_local, bytes_local, alg_local)
    counts = None
    if args.workload == "q3":
        counts = max_over_ranks(dist, 0.0, *(q3_stats.get(k, 0) for k in ("customer_filtered", "semi_join", "join", "groups")), *(t.num_rows for t in tables))[1]
    checkpoint_phase("done")
    if rank != 0:
        return None
    step_s = dt / args.steps
    top = sorted(stats.items(), key=lambda kv: -kv[1]["total_ms"])[:8]
    # config 4 = the Arrow buffer bytes of the 7 referenced columns (l_shipdate 4 + 4 x Decimal128 64 + 2 flags as 1-byte
    # codes = 70 B/row; the 4 x 10 output values are noise); config 5 = the operator table with the measured intermediate row counts
    if args.workload == "q1":
        alg_table = [{"operator": "FilterExec + ProjectionExec + AggregateExec (fused node) over [l_shipdate, l_quantity, l_extendedprice, l_discount, l_tax, "
                                  "l_returnflag (u8), l_linestatus (u8)]", "input_bytes": rows * 70, "output_bytes": n_out * 10 * 16}]
    else:
        alg_table = q3_algorithmic_table(counts[4], counts[5], counts[6], dict(zip(("customer_filtered", "semi_join", "join", "groups"), counts[:4])))
    alg = sum(r["input_bytes"] + r["output_bytes"] for r in alg_table)
    roof, roof_launches = None, None
    if top and top[0][1]["calls"]:
        name, d = top[0]

        def roofline_of(avg_ms, per_launch, launches_per_step, which):
            achieved = per_launch / (avg_ms * 1e-3) / 1e9
            return {"bound": "hbm", "kernel": name, "device_kernel": DEVICE_KERNEL_OF.get(name), "launch": which, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "frac_vs_copy_ceiling": round(achieved / HBM_COPY_CEILING_GBS, 4), "traffic": None,
                    "avg_launch_ms": round(avg_ms, 4), "launches_per_step": launches_per_step, "algorithmic_bytes_per_launch": int(per_launch),
                    "share_of_step": round(avg_ms * launches_per_step / (step_s * 1e3), 3)}
        lps = d["calls"] // args.steps
        if lps > 1 and lps * args.steps == d["calls"] and len(launch_recs.get(name, ())) == d["calls"]:
            # the same kernel over unlike inputs within one step (Q3: the counts pass over orders, then over lineitem): one roofline per
            # position in the step, `roofline` = the position that takes longest
            recs = launch_recs[name]
            roof_launches = []
            for k in range(lps):
                mine = recs[k::lps]
                roof_launches.append(roofline_of(sum(m for m, _ in mine) / len(mine), sum(b for _, b in mine) / len(mine), 1, f"launch {k + 1} of {lps} in a step"))
            roof = max(roof_launches, key=lambda x: x["avg_launch_ms"])
        else:
            roof = roofline_of(d["total_ms"] / d["calls"], d["bytes"] / d["calls"], d["calls"] / args.steps, "every launch")
        key = {"query": args.workload, "sf": args.sf, "input_rows": rows}
        for rf in (roof_launches or [roof]):
            attach_traffic(rf, name, {**key, **({"algorithmic_bytes_per_launch": rf["algorithmic_bytes_per_launch"]} if roof_launches else {})}, world)
    line = {
        "metric": f"tpch_{args.workload}_rows_per_sec", "value": rows / step_s, "unit": "rows/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": step_s * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "decimal128 / int64", "data": "synthetic",
        "config": {"workload": ("TPC-H Q1 (FilterExec + ProjectionExec + grouped AggregateExec fused, Partial -> hash exchange -> FinalPartitioned at N > 1)" if args.workload == "q1"
                                else "TPC-H Q3 end to end (2 hash joins + aggregate + top-k; at N > 1 four hash repartitions in two exchange phases)") + f", SF{args.sf:g}, device-resident inputs",
                   "input_rows": rows, "output_rows": n_out, "parallelism": "single GPU" if world == 1 else f"{world} ranks, dfgpu_exchange_hash (RCCL all-to-all(v))",
                   **({"plan": "the fused node is planned once before the timed region (expressions lowered, AVG return types); the timed steps execute the held plan"}
                      if args.workload == "q1" else {})},
        "scanned_table_gb_per_s": round(nbytes / step_s / 1e9, 1),
        # whole-step fraction of peak by the bytes the step's kernels have to move (each kernel's own algorithmic bytes, summed by the
        # library): a late-materialising plan never reads the columns of rows it drops, so SURVEY 8(d)'s operator-table formula (every
        # referenced column read once) is kept beside it under its own name and is NOT a roofline fraction
        "kernel_algorithmic_bytes_per_step": int(alg_kernels),
        "algorithmic_gb_per_s": round(alg_kernels / step_s / 1e9, 1), "hbm_frac_whole_step": round(alg_kernels / step_s / 1e9 / (HBM_PEAK_GBS * world), 4),
        "survey_8d_formula": {"bytes_per_step": int(alg), "table": alg_table, "gb_per_s": round(alg / step_s / 1e9, 1),
                              "note": "referenced input column bytes + output bytes per operator; not a roofline fraction when the plan skips columns of dropped rows"},
        "roofline": roof, **({"roofline_per_launch": roof_launches} if roof_launches else {}),
        "kernels": {k: {"calls": v["calls"], "avg_ms": round(v["total_ms"] / max(1, v["calls"]), 4)} for k, v in top},
        "kernel_ms_per_step": round(sum(v["total_ms"] for v in stats.values()) / args.steps, 4),
        "crossed_per_step_rank0": {k: v // args.steps for k, v in xs.items()} if xs else None,
    }
    if counts is not None:
        line["config"]["intermediate_rows"] = dict(zip(("customer_filtered", "semi_join", "join", "groups"), counts[:4]))
    if not args.no_cpu and world == 1:
        cb = cpu_baseline_query(args.workload, args.sf, os.cpu_count() or 1, device_tables=tables, max_sf=args.cpu_sf)
        cpu_result = cb.pop("result")
        if cb["same_workload_as_gpu_leg"]:
            # the oracle as the CHECKER of the GPU leg at the benchmark's own size: the same tables, so the same rows (bit-exact)
            cb["result_equals_gpu_leg"] = bool(cpu_result.to_pylist() == gpu_result.to_pylist())
            if counts is not None:
                cb["intermediate_rows_equal_gpu_leg"] = bool(all(cb["intermediate_rows"].get(k) == v for k, v in line["config"]["intermediate_rows"].items()))
        line["cpu_baseline"] = cb
        line["speedup_vs_cpu_port"] = round(line["value"] / cb["value"], 1) if cb["same_workload_as_gpu_leg"] else None
    for t in tables:
        t.free()
    return line


ALSO_DEFAULT = "q1:100,q3:300"   # BASELINE config 4 (TPC-H Q1 SF100) and config 5 (TPC-H Q3 SF300: the one-GPU anchor of the 8-GPU configuration)


def also_legs(args, spec, deadline_s=900.0):
    """BASELINE configs 4 and 5 beside the headline, on the default single-GPU line: each runs as its own process (`bench.py --workload q1 /
    q3` at the configuration's scale factor, the same --steps / --warmup, its CPU leg included) AFTER the headline's timed region and after
    its tables are freed, and its line is folded into `also` — ms_per_step, rows/s, the dominant kernel's roofline, the whole-step
    fraction, the CPU leg on the same tables and whether the oracle's result equals the GPU leg's.  A leg that fails or runs out of time
    leaves an `error` entry; the headline line is printed whatever happens here."""
    import subprocess
    out = {}
    t_start = time.perf_counter()
    for item in spec.split(","):
        workload, sf = item.split(":")[0], float(item.split(":")[1])
        what = f"BASELINE config {4 if workload == 'q1' else 5}: TPC-H {workload.upper()} at SF{sf:g} on one GPU"
        left = deadline_s - (time.perf_counter() - t_start)
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--workload", workload, "--sf", str(sf), "--steps", str(args.steps),
               "--warmup", str(args.warmup), "--no-also"] + (["--no-cpu"] if args.no_cpu else []) + ([] if args.cpu_sf is None else ["--cpu-sf", str(args.cpu_sf)])
        try:
            if left < 30:
                raise TimeoutError("no time left for this leg")
            t0 = time.perf_counter()
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=left)
            lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
            if p.returncode != 0 or not lines:
                raise RuntimeError(f"rc {p.returncode}: {(p.stderr or p.stdout)[-300:]}")
            d = json.loads(lines[-1])
            keep = ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype", "config", "kernel_algorithmic_bytes_per_step", "algorithmic_gb_per_s",
                    "hbm_frac_whole_step", "roofline", "roofline_per_launch", "kernels", "kernel_ms_per_step", "cpu_baseline", "speedup_vs_cpu_port")
            out[workload] = {"what": what, **{k: d[k] for k in keep if k in d}, "survey_8d_formula_gb_per_s": d.get("survey_8d_formula", {}).get("gb_per_s"),
                             "wall_s": round(time.perf_counter() - t0, 1)}
        except Exception as e:  # noqa: BLE001 - the headline must not depend on a secondary leg
            out[workload] = {"what": what, "error": repr(e)[:400]}
    return out


def self_spawn(args):
    """`bench.py --gpus N` started WITHOUT a launcher (no RANK / WORLD_SIZE in the environment): start the N ranks ourselves — the
    command line the driver documents, `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` — and pass
    their output through, so the run produces its line either way."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--sf", type=float, default=100.0, help="TPC-H scale factor (BASELINE: 100)")
    ap.add_argument("--cpu-sf", type=float, default=None, help="largest scale factor the CPU-baseline leg may run at (default: the GPU leg's, taken when host RAM allows)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--workload", choices=["join", "q1", "q3"], default="join",
                    help="join = the BASELINE metric (config 3 ii); q1 / q3 = the whole TPC-H Q1 / Q3 plan per step (configs 4 and 5)")
    ap.add_argument("--exchange-chunks", type=int, default=4,
                    help="N > 1: row ranges the streamed hash repartition cuts every input into (1 = the blocking exchange: partition, all-to-all(v), join one after the other)")
    ap.add_argument("--exchange", choices=["auto", "pruned", "broadcast", "repartition", "repartition_stream"], default="auto",
                    help="N > 1, join workload: auto = time both repartition (Partitioned: hash exchange of both sides) and pruned (CollectLeft, "
                         "build side pruned by probe-key bounds), `value` = the one a byte-counting planner picks; anything else forces that "
                         "exchange (at N = 1: a one-rank rehearsal of its code path)")
    ap.add_argument("--shard-skew", type=float, default=0.01,
                    help="N > 1: the lineitem shards start this fraction of a shard later than the orders shards (scan boundaries of two tables "
                         "do not fall on the same keys)")
    ap.add_argument("--probe-mode", type=int, default=3,
                    help="3 single pass, unordered output (default: in Q3 the join feeds AggregateExec, no ancestor needs the probe "
                         "order); 0/1 output in probe order (tile counts -> scan -> placed); 2 single pass ordered (look-back)")
    ap.add_argument("--watchdog-s", type=float, default=300.0,
                    help="N > 1: seconds one phase (RCCL bootstrap, table generation, one exchange flavour's warm-up + timed steps) may take before rank 0 "
                         "prints the line of what HAS been measured with a `watchdog` object and every rank exits (0 = no watchdog)")
    ap.add_argument("--also", default=None,
                    help=f"workload:SF pairs run after the headline and folded into `also` (default {ALSO_DEFAULT} when --sf is 100, nothing otherwise)")
    ap.add_argument("--no-also", action="store_true",
                    help="the default single-GPU join line also carries BASELINE configs 4 and 5 (Q1 SF100, Q3 SF300, each run as its own process after "
                         "the headline's timed region) under `also`; this leaves them out")
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_spawn(args))
    start_watchdog(args, int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")))
    rank, world, _local, dist = setup_dist(args)
    checkpoint_phase("generate tables")
    line = run_join(args, rank, world, dist) if args.workload == "join" else run_query(args, rank, world, dist)
    also = "" if args.no_also else (args.also if args.also is not None else (ALSO_DEFAULT if args.sf == 100.0 else ""))
    if world == 1 and args.workload == "join" and args.exchange == "auto" and also:
        from datafusion_amd import _lib
        _lib.load().dfgpu_mem_trim()    # the headline's HBM goes back to the driver before the other configurations take theirs
        line["also"] = also_legs(args, also)
    # the JSON line is the LAST thing on stdout: whatever native libraries buffered in C stdio (RCCL prints its version banner
    # there) leaves every rank's buffer first
    import ctypes
    sys.stdout.flush()
    ctypes.CDLL(None).fflush(None)
    if dist is not None:
        dist.barrier()
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        sys.stdout.flush()
        sys.stderr.flush()
        # RCCL teardown has aborted on some boxes after all work was done and checked; the line is out, leave now
        os._exit(0)


if __name__ == "__main__":
    main()
