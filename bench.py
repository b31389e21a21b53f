#!/usr/bin/env python3
"""bench.py — BASELINE.json metric: rows/sec + GB/s HBM of the TPC-H Q3 hash join
(orders ⋈ lineitem on orderkey, Q3 payload projection) at SF100 on 1/2/4/8 MI355X.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one full pass of the hot path over device-resident inputs:
  N = 1 : HashJoinExec build (collect_left_input) + probe (whole lineitem) -> output table.
  N > 1 : each rank holds a 1/N row range of both tables (what N scans would produce).  The exchange that
          moves fewer bytes per GPU is used: PartitionMode::CollectLeft = the build side is
          broadcast (B(N-1)/N bytes received per GPU at most) when B*N < B+P — the SF100 Q3 join up to N = 8 —
          else PartitionMode::Partitioned = K10 partition kernel + RCCL all-to-all of both sides; then the local
          build + probe.  The broadcast is pruned by each rank's probe-key bounds (exchange.pruned_broadcast_table:
          a rank only receives build rows inside [min, max] of its own probe keys — the reference's dynamic join
          filter turned around); shards that arrive clustered by key, like the row ranges used here and like any
          scan of TPC-H tables in their natural order, therefore move almost nothing, while spread keys fall back
          to the full all-gather's volume.  `exchange_rank0` on the JSON line says how many rows crossed ranks;
          --exchange broadcast / repartition force the unpruned exchanges.  Total work is fixed (SF100) =>
          "strong" scaling.
value = (build rows + probe rows summed over ranks) / max-over-ranks wall time of the K steps.
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
  cpu_baseline — the CPU restatement (oracle/, kind "port") of DataFusion's partitioned hash
                 join, timed on this box's host cores on a bounded sample of the same workload.
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

BUILD_COLS = ["o_orderdate", "o_shippriority"]
PROBE_COLS = ["l_orderkey", "l_extendedprice", "l_discount"]


def algorithmic_bytes(nb, np_, nout):
    """read each referenced input column once + write the output once:
    build {o_orderkey 8, o_orderdate 4, o_shippriority 4} = 16 B/row, probe {l_orderkey 8,
    l_extendedprice 16, l_discount 16} = 40 B/row, output 48 B/row"""
    return nb * 16 + np_ * 40 + nout * 48


def cpu_baseline(sample_sf, cores):
    """oracle leg: RepartitionExec(Hash) x2 -> HashJoinExec(Partitioned), one thread per partition
    (target_partitions); the best of a few partition counts up to the core count is reported, since one
    partition per core is not the fastest setting on a many-core NUMA host"""
    import numpy as np

    from datafusion_amd import tpch
    from oracle import oracle
    i = np.arange(tpch.n_orders(sample_sf), dtype=np.int64)
    bk = tpch.order_key(i)
    pk = np.repeat(bk, tpch.line_count(i))
    best, best_t, tried = None, None, {}
    for threads in sorted({max(1, cores), max(1, cores // 2), max(1, cores // 4), min(cores, 32)}, reverse=True):
        t0 = time.perf_counter()
        pairs, _chk = oracle.partitioned_inner_join_i64(bk, pk, threads)
        dt = time.perf_counter() - t0
        assert pairs == len(pk)
        tried[threads] = round((len(bk) + len(pk)) / dt)
        if best is None or dt < best:
            best, best_t = dt, threads
    out = {"value": (len(bk) + len(pk)) / best, "unit": "rows/s", "cores": best_t, "kind": "port",
           "sample": f"orders x lineitem keys at SF{sample_sf:g} ({len(bk)} build + {len(pk)} probe rows), "
                     f"{best_t} partitions/threads (best of {tried} rows/s; host has {cores} cores), 8192-row probe batches, key-only pairs (no payload gather)"}
    try:  # independent production CPU engine on the same sample: Arrow Acero hash join
        import pyarrow as pa
        pa.set_cpu_count(cores)
        b = pa.table({"k": bk, "bi": np.arange(len(bk), dtype=np.int64)})
        p = pa.table({"k": pk, "pi": np.arange(len(pk), dtype=np.int64)})
        t_best = None
        for _ in range(2):
            t0 = time.perf_counter()
            j = p.join(b, keys="k", join_type="inner")
            dt = time.perf_counter() - t0
            assert j.num_rows == len(pk)
            t_best = dt if t_best is None else min(t_best, dt)
        out["acero_rows_per_s"] = (len(bk) + len(pk)) / t_best
    except Exception as e:  # noqa: BLE001 - the Acero leg is a sanity bound, never required
        out["acero_error"] = str(e)[:200]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--sf", type=float, default=100.0, help="TPC-H scale factor (BASELINE: 100)")
    ap.add_argument("--cpu-sf", type=float, default=10.0, help="scale factor of the CPU-baseline sample")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--exchange", choices=["auto", "pruned", "broadcast", "repartition"], default="auto",
                    help="N > 1: pruned = CollectLeft with the build-side broadcast pruned by each rank's probe-key bounds; broadcast = plain "
                         "all-gather of the build side; repartition = hash-repartition both sides (Partitioned); auto = pruned when a broadcast "
                         "moves fewer bytes than a repartition, else repartition")
    ap.add_argument("--probe-mode", type=int, default=3,
                    help="3 single pass, unordered output (default: in Q3 the join feeds AggregateExec, no ancestor needs the probe "
                         "order); 0/1 two passes, output in probe order; 2 single pass ordered (look-back)")
    args = ap.parse_args()

    import torch

    from datafusion_amd import _lib, ops

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    torch.cuda.set_device(local_rank)
    _lib.init(local_rank)
    dist = None
    forced = world == 1 and args.exchange != "auto"  # one-rank rehearsal of the N > 1 exchange code on a 1-GPU box
    if world > 1 or forced:
        import torch.distributed as dist
        if forced:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29544")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from datafusion_amd import tpch
    n_orders = tpch.n_orders(args.sf)
    b, e = n_orders * rank // world, n_orders * (rank + 1) // world
    orders = ops.tpch_orders(args.sf, b, e).select(["o_orderkey", "o_orderdate", "o_shippriority"])
    lineitem = ops.tpch_lineitem(args.sf, b, e).select(["l_orderkey", "l_extendedprice", "l_discount"])
    nb_local, np_local = orders.num_rows, lineitem.num_rows
    ops.sync()

    exchange = args.exchange if forced else "none"
    if world > 1:
        from datafusion_amd.exchange import broadcast_build_moves_fewer_bytes
        t4 = torch.tensor([float(nb_local), float(np_local)], dtype=torch.float64, device="cuda")
        dist.all_reduce(t4)
        exchange = args.exchange
        if exchange == "auto":
            exchange = "pruned" if broadcast_build_moves_fewer_bytes(int(t4[0]) * 16, int(t4[1]) * 40, world) else "repartition"

    xstats = {}

    def step(probe_mode=args.probe_mode):
        o, l = orders, lineitem
        if exchange == "pruned":
            from datafusion_amd.exchange import pruned_broadcast_table
            o = pruned_broadcast_table(orders, "o_orderkey", lineitem, "l_orderkey", force=forced, stats=xstats)
        elif exchange == "broadcast":
            from datafusion_amd.exchange import broadcast_table
            o = broadcast_table(orders, force=forced)
        elif exchange == "repartition":
            from datafusion_amd.exchange import hash_exchange
            o = hash_exchange(orders, ["o_orderkey"], force=forced)
            l = hash_exchange(lineitem, ["l_orderkey"], force=forced)
        # join-table choice follows the library default (direct-address down to key density 1/64, see
        # DFGPU_DEFAULT_MIN_KEY_DENSITY in include/dfgpu.h): at N > 1 hash routing leaves each rank 1/N of
        # the keys over the same key range (density 0.25/N)
        ht = ops.JoinHashTable(o, ["o_orderkey"], probe_mode=probe_mode)
        out = ht.probe(l, ["l_orderkey"], "Inner", BUILD_COLS, PROBE_COLS)
        n_out = out.num_rows
        info = ht.info()
        out.free()
        ht.free()
        if exchange in ("broadcast", "pruned"):
            o.free()
        elif exchange == "repartition":
            o.free()
            l.free()
        return n_out, info

    def barrier():
        ops.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    ops.profile_enable(True)
    ops.profile_reset()
    barrier()
    t0 = time.perf_counter()
    n_out = 0
    for _ in range(args.steps):
        n_out, info = step()
    barrier()
    dt = time.perf_counter() - t0
    stats = ops.profile_stats()
    ops.profile_enable(False)
    # secondary, outside the contract's timed region: the same step with output in probe order
    # (two passes), for plans where an ancestor does need HashJoinExec's probe-side ordering
    ordered_ms, ordered_stats = None, {}
    if args.probe_mode == 3 and world == 1:
        step(0)
        ops.profile_enable(True)
        ops.profile_reset()
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step(0)
        barrier()
        ordered_ms = (time.perf_counter() - t1) / args.steps * 1e3
        ordered_stats = ops.profile_stats()
        ops.profile_enable(False)

    tot = torch.tensor([float(nb_local), float(np_local), float(n_out), dt], dtype=torch.float64, device="cuda")
    if dist is not None:
        mx = tot.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dt = float(mx[3])
    nb, np_, nout = int(tot[0]), int(tot[1]), int(tot[2])

    if rank == 0:
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
            return {"bound": "hbm", "kernel": name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "frac_vs_copy_ceiling": round(achieved / HBM_COPY_CEILING_GBS, 4),
                    "traffic": None, "avg_launch_ms": round(avg_ms, 4), "algorithmic_bytes_per_launch": int(per_launch)}

        def attach_traffic(roof):
            """HBM bytes per launch from the committed PMC passes (scripts/profile.sh), when taken on this very workload and kernel"""
            try:
                for tr in json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["kernels"]:
                    wl = tr["workload"]
                    if tr["kernel"] == roof["kernel"] and (wl["build_rows"], wl["probe_rows"], wl["output_rows"]) == (nb, np_, nout) and world == 1:
                        roof["traffic"] = tr["traffic_bytes_per_launch"]
                        roof["traffic_source"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, profiles/" + tr["source"]
            except (OSError, KeyError, ValueError, TypeError):
                pass
            return roof

        roof = roofline_of(stats)
        if roof:
            attach_traffic(roof)
        kernels = {k: {"calls": v["calls"], "avg_ms": round(v["total_ms"] / max(1, v["calls"]), 4)} for k, v in stats.items()}
        line = {
            "metric": "tpch_q3_hash_join_rows_per_sec", "value": rows_per_s, "unit": "rows/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "int64 keys / decimal128 payload", "data": "synthetic",
            "config": {"workload": f"INNER hash-join orders⋈lineitem on o_orderkey, TPC-H SF{args.sf:g}, Q3 payload "
                                   "(o_orderdate,o_shippriority,l_orderkey,l_extendedprice,l_discount), device-resident inputs",
                       "build_rows": nb, "probe_rows": np_, "output_rows": nout,
                       "join_table": {0: "hash_map", 1: "array_map", 2: "rank_map"}[info.table_kind],
                       "probe": {0: "placed_ordered", 1: "placed_ordered", 2: "single_pass_ordered_lookback", 3: "single_pass_unordered"}[args.probe_mode],
                       "parallelism": "single GPU" if world == 1 else
                       (f"CollectLeft x{world}: build side broadcast pruned by each rank's probe-key bounds (RCCL all-to-all(v)), probe side stays in place"
                        if exchange == "pruned" else
                        f"CollectLeft x{world}: RCCL all-gather of the build side, probe side stays in place" if exchange == "broadcast"
                        else f"Partitioned: hash-repartition all-to-all of both sides x{world}")},
            "algorithmic_gb_per_s": round(alg / (dt / args.steps) / 1e9, 1),
            "hbm_frac_whole_step": round(alg / (dt / args.steps) / 1e9 / (HBM_PEAK_GBS * world), 4),
            "roofline": roof, "kernels": kernels,
        }
        if xstats:
            line["exchange_rank0"] = {**xstats, "build_row_bytes": 16}
        if ordered_ms is not None:
            # the same step with the output in probe order, exactly as the reference emits it (hash_join/exec.rs:3349):
            # tile counts -> scan -> the fused kernel with known tile offsets
            oroof = roofline_of(ordered_stats)
            line["ordered_output"] = {"probe": "placed (tile counts -> scan -> fused materialise)", "ms_per_step": round(ordered_ms, 3),
                                      "rows_per_s": (nb + np_) / (ordered_ms * 1e-3),
                                      "hbm_frac_whole_step": round(alg / (ordered_ms * 1e-3) / 1e9 / (HBM_PEAK_GBS * world), 4),
                                      "roofline": attach_traffic(oroof) if oroof else None,
                                      "kernels": {k: {"calls": v["calls"], "avg_ms": round(v["total_ms"] / max(1, v["calls"]), 4)} for k, v in ordered_stats.items()}}
        if not args.no_cpu and world == 1:  # the CPU baseline is reported by the single-GPU run only
            threads = os.cpu_count() or 1
            line["cpu_baseline"] = cpu_baseline(args.cpu_sf, threads)
            line["speedup_vs_cpu_port"] = round(rows_per_s / line["cpu_baseline"]["value"], 1)
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        sys.stdout.flush()
        sys.stderr.flush()
        # RCCL teardown has aborted on some boxes after all work was done and checked; the line is out, leave now
        os._exit(0)


if __name__ == "__main__":
    main()
