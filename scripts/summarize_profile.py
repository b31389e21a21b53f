#!/usr/bin/env python3
"""Summarise a scripts/profile.sh run (rocprofv3 rocpd sqlite outputs) into profiles/<tag>.md/.json.

  python scripts/summarize_profile.py gpurun_out/<tag> profiles/<tag>

Kernel durations come from `rocprofv3 --kernel-trace --stats`; HBM traffic from two separate
`--pmc` passes (FETCH_SIZE, WRITE_SIZE; TCC has 4 slots, FETCH_SIZE takes 3 and WRITE_SIZE 2 —
MI355X_MICROARCH.md §rocprofv3 PMC slots).  gfx950 correction (same guide, §HBM): FETCH_SIZE
reports 1/2 of the bytes of a coalesced streaming read, so read bytes = FETCH_SIZE*1024*2;
this is calibrated inside the same run on k_rank_setbits, which reads exactly 8 B x build rows.
Also refreshes profiles/traffic.json (per-launch PMC traffic of the dominant kernel), which bench.py
attaches to its roofline object when the workload matches.
"""
import json
import os
import sqlite3
import sys


def q(db, sql):
    con = sqlite3.connect(db)
    try:
        return list(con.execute(sql))
    finally:
        con.close()


def short(name):
    n = name.replace("void ", "").split("(")[0]
    return n.replace("dfgpu::", "")


def main():
    src, dst = sys.argv[1], sys.argv[2]
    bench = None
    for line in open(os.path.join(src, "bench.json")):
        if line.startswith("{"):
            bench = json.loads(line)
    # One row per kernel — and per GRID SIZE where a kernel runs over unlike inputs within a step (Q3: k_join_tile_counts over 450 M orders
    # rows, then over 1.8 G lineitem rows; round-5 verdict, weak 2): such launches get their own duration and their own traffic.
    stats_db = os.path.join(src, "stats", "trace_results.db")
    per_grid = q(stats_db, "select name, grid_x, count(*), sum(duration), avg(duration) from kernels group by 1, 2")     # durations in ns
    total_ns = sum(r[3] for r in per_grid) or 1
    grids_of = {}
    for n, g, c, tot, avg in per_grid:
        grids_of.setdefault(short(n), []).append((g, c, tot, avg))
    def split(k):   # by grid size: several sizes, each launched more than once (a size per step), a kernel that matters
        gs = grids_of.get(k, [])
        return 1 < len(gs) <= 4 and all(c > 1 for _, c, _, _ in gs) and sum(t for _, _, t, _ in gs) / total_ns >= 0.02
    kern = {}
    for k, gs in grids_of.items():
        if split(k):
            for g, c, tot, avg in gs:
                kern[f"{k} [grid {g}]"] = {"calls": c, "total_us": tot / 1e3, "avg_us": avg / 1e3, "pct": 100.0 * tot / total_ns, "grid": g, "base": k}
        else:
            c, tot = sum(x[1] for x in gs), sum(x[2] for x in gs)
            kern[k] = {"calls": c, "total_us": tot / 1e3, "avg_us": tot / c / 1e3, "pct": 100.0 * tot / total_ns, "grid": None, "base": k}
    def pmc(sub):
        out = {}
        for n, g, v in q(os.path.join(src, sub, "pmc_results.db"), "select kernel_name, grid_size, avg(value) from counters_collection group by 1, 2"):
            out[(short(n), g)] = v
        res = {}
        for k, v in kern.items():
            if v["grid"] is not None:
                if (v["base"], v["grid"]) in out:
                    res[k] = out[(v["base"], v["grid"])]
            else:
                vals = [(x, grids) for (kk, grids), x in out.items() if kk == k]
                if vals:   # launch-weighted over the kernel's grid sizes (the PMC pass ran the same launches)
                    w = {g: c for g, c, _, _ in grids_of.get(k, [])}
                    tot_w = sum(w.get(g, 1) for _, g in vals)
                    res[k] = sum(x * w.get(g, 1) for x, g in vals) / tot_w
        return res
    fetch, write = pmc("pmc_fetch"), pmc("pmc_write")
    # streamed (algorithmic) read bytes of the dominant kernel's launches, by position in the step — bench.py's roofline_per_launch —
    # matched to the grid sizes by duration order: FETCH_SIZE counts a coalesced stream at HALF its bytes and a random 64-byte
    # gather at face value (profiles/r2_fetch_calib.md: 2.000 and 1.001), so a kernel that streams S bytes and gathers the rest read
    # S + (FETCH_SIZE*1024 - S/2) bytes, not 2 x FETCH_SIZE*1024
    streamed = {}
    per_launch = (bench or {}).get("roofline_per_launch")
    if per_launch:
        from bench import DEVICE_KERNEL_OF as _DK
        prefix = _DK.get(per_launch[0]["kernel"])
        cands = sorted([k for k, v in kern.items() if v["grid"] is not None and prefix and v["base"].startswith(prefix)], key=lambda k: kern[k]["avg_us"])
        for k, pl in zip(cands, sorted(per_launch, key=lambda x: x["avg_launch_ms"])):
            streamed[k] = pl["algorithmic_bytes_per_launch"]
            kern[k]["bench_launch"] = pl["launch"]
    nb = bench["config"].get("build_rows") if bench else None
    # calibration kernel: k_rank_setbits (ascending variant) reads exactly the 8-byte build keys, nothing else
    calib = None
    # (the VERIFY variant — 4 template arguments — also reads every key's predecessor: not a known-bytes kernel)
    cal_k = next((k for k in fetch if k.startswith("k_rank_setbits") and k.count(",") < 3), None) or next((k for k in fetch if k.startswith("k_key_minmax")), None)
    if nb and cal_k:
        calib = nb * 8 / (fetch[cal_k] * 1024)
    rows = []
    for k, v in sorted(kern.items(), key=lambda kv: -kv[1]["total_us"]):
        f, w = fetch.get(k), write.get(k)
        rd = f * 1024 * 2 if f is not None else None
        wr = w * 1024 if w is not None else None
        S = streamed.get(k)
        if S is not None and f is not None:
            rd = S + max(0.0, f * 1024 - S / 2)      # the streamed columns in full + the gathered lines at face value
        tr = (rd or 0) + (wr or 0) if (f is not None or w is not None) else None
        rows.append({"kernel": k, **v, "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "read_bytes_corrected": rd, "write_bytes": wr, "streamed_bytes_known": S,
                     "traffic_bytes": tr, "traffic_GBps": (tr / (v["avg_us"] * 1e-6) / 1e9) if tr else None,
                     "traffic_over_algorithmic": (tr / S) if (tr and S) else None, "algorithmic_GBps": (S / (v["avg_us"] * 1e-6) / 1e9) if S else None})
    out = {"bench": bench, "fetch_size_calibration_factor_measured": calib, "kernels": rows}
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
    json.dump(out, open(dst + ".json", "w"), indent=1)
    with open(dst + ".md", "w") as f:
        f.write(f"# rocprofv3 summary — {os.path.basename(src)}\n\n")
        if bench:
            f.write("bench line: `" + json.dumps({k: bench[k] for k in ("metric", "value", "unit", "n_gpus", "ms_per_step")}) + "`\n\n")
            f.write("roofline: `" + json.dumps(bench.get("roofline")) + "`\n\n")
        if calib is not None:
            f.write(f"FETCH_SIZE calibration (bytes {cal_k} must read / FETCH_SIZE*1024): {calib}\n\n")
        else:
            f.write("FETCH_SIZE calibration: no kernel with known read bytes in this run (the build's one pass reads each key and its predecessor); "
                    "the x2 factor for coalesced reads is the one measured in profiles/r2_fetch_calib.md (2.000) and in profiles/r3_sf100_v1.md (1.99990)\n\n")
        f.write("Traffic = reads + WRITE_SIZE*1024.  Reads: FETCH_SIZE*1024 x 2 (coalesced streams are counted at half their bytes on gfx950) — except on the rows "
                "that carry `algorithmic GB` (the dominant kernel's launches, one row per grid size): there the streamed columns count in full and what "
                "FETCH_SIZE holds beyond half of them is gathered 64-byte lines at face value (profiles/r2_fetch_calib.md).\n\n")
        f.write("| kernel | calls | avg µs | % | FETCH_SIZE KB/launch | WRITE_SIZE KB/launch | HBM traffic GB/launch | traffic GB/s | algorithmic GB/launch | algorithmic GB/s (of 8000) | traffic / algorithmic |\n"
                "|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|\n")
        for r in rows:
            fmt = lambda x, d=1: "" if x is None else f"{x:.{d}f}"
            f.write(f"| {r['kernel']}{' = ' + r['bench_launch'] if r.get('bench_launch') else ''} | {r['calls']} | {r['avg_us']:.1f} | {r['pct']:.1f} | {fmt(r['FETCH_SIZE_KB'],0)} | {fmt(r['WRITE_SIZE_KB'],0)} | "
                    f"{fmt(r['traffic_bytes']/1e9 if r['traffic_bytes'] else None,2)} | {fmt(r['traffic_GBps'],0)} | {fmt(r['streamed_bytes_known']/1e9 if r['streamed_bytes_known'] else None,2)} | "
                    f"{fmt(r['algorithmic_GBps'],0)}{' (' + format(r['algorithmic_GBps']/8000, '.2f') + ')' if r['algorithmic_GBps'] else ''} | {fmt(r['traffic_over_algorithmic'],2)} |\n")
    # ---- agreement of the two clocks and the clock state of the passes (scripts/profile_run.py)
    try:
        agree = json.load(open(os.path.join(src, "agreement.json")))["tries"]
        with open(dst + ".md", "a") as f:
            f.write("\nHIP-event average vs rocprofv3 average of the dominant kernel, SAME process (the kernel-trace pass prints its own bench line):\n\n"
                    "| try | scope | device kernel | HIP events ms | rocprofv3 ms | difference | step ms (under the profiler) |\n|---:|---|---|---:|---:|---:|---:|\n")
            for t in agree:
                if "error" in t:
                    f.write(f"| {t['try']} | error: {t['error']} | | | | | |\n")
                else:
                    f.write(f"| {t['try']} | {t['scope']} | {t['device_kernel']} | {t['hip_event_avg_ms']} | {t['rocprof_avg_ms']} | "
                            f"{'' if t['relative_difference'] is None else format(t['relative_difference'] * 100, '.2f') + ' %'} | {t['ms_per_step']:.3f} |\n")
        out["agreement"] = agree
    except (OSError, KeyError, ValueError):
        pass
    try:
        clocks = json.load(open(os.path.join(src, "clocks.json")))
        out["clocks"] = clocks
        with open(dst + ".md", "a") as f:
            f.write("\nclock state (rocm-smi) before -> after each pass:\n\n")
            for c in clocks:
                def brief(x):
                    if not isinstance(x, dict):
                        return str(x)
                    card = x.get("card0", x)
                    keep = {k: v for k, v in card.items() if any(w in k.lower() for w in ("sclk", "mclk", "power", "performance", "temperature (sensor junction)", "temperature (sensor memory)"))} if isinstance(card, dict) else card
                    return json.dumps(keep)[:600]
                f.write(f"- `{c['pass']}` ({c['seconds']} s, rc {c['rc']}): {brief(c['before'])} -> {brief(c['after'])}\n")
    except (OSError, KeyError, ValueError):
        pass
    json.dump(out, open(dst + ".json", "w"), indent=1)
    # per-launch PMC traffic of the line's dominant kernel -> profiles/traffic.json (entries of OTHER workloads / kernels are kept).
    # What the passes were taken on: the commit checked out when this summary is made (the run's snapshot) and a hash of the file that
    # defines the kernel — bench.py attaches the traffic only while that file is unchanged (a kernel edit without a fresh PMC pass must
    # not report the old bytes)
    import hashlib
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from bench import DEVICE_KERNEL_OF
    try:
        commit = subprocess.check_output(["git", "-C", root, "rev-parse", "--short=12", "HEAD"], text=True).strip()
    except Exception:  # noqa: BLE001
        commit = None
    entries = []
    is_join = bool(bench) and "build_rows" in bench.get("config", {})
    for r in rows:
        if not r["traffic_bytes"] or not bench:
            continue
        if is_join:
            if not r["kernel"].startswith("k_join_probe_fused"):
                continue
            targs = [a.strip() for a in r["kernel"][r["kernel"].index("<") + 1:].rstrip(">").split(",")]   # <KIND, KT, W, MODE, KEYREG>
            mode = targs[3] if len(targs) >= 4 else ""
            name = {"0": "join_probe_fused", "2": "join_probe_placed"}.get(mode.replace("(dfgpu::FusedMode)", ""))
            if name is None:
                continue
            wl = {k: bench["config"][k] for k in ("build_rows", "probe_rows", "output_rows", "join_table")}
            ksrc = "datafusion_amd/csrc/join.hip"
        else:
            scope = (bench.get("roofline") or {}).get("kernel")
            prefix = DEVICE_KERNEL_OF.get(scope)
            if prefix is None or not r["kernel"].startswith(prefix):
                continue
            wl = {"query": bench["metric"].split("_")[1], "sf": float(bench["config"]["workload"].split(", SF")[1].split(",")[0]), "input_rows": bench["config"]["input_rows"]}
            if r.get("streamed_bytes_known"):   # one entry per launch of the step, told apart by its algorithmic bytes
                wl["algorithmic_bytes_per_launch"] = int(r["streamed_bytes_known"])
            elif any(e["kernel"] == scope for e in entries):     # (rows are sorted by total time: the first match is the dominant instantiation)
                continue
            name = scope
            ksrc = "datafusion_amd/csrc/aggregate.hip" if scope.startswith("agg") else "datafusion_amd/csrc/join.hip"
        src_sha = hashlib.sha256(open(os.path.join(root, ksrc), "rb").read()).hexdigest()[:16]
        entries.append({"kernel": name, "device_kernel": r["kernel"], "traffic_bytes_per_launch": int(r["traffic_bytes"]),
                        "read_bytes_corrected": int(r["read_bytes_corrected"] or 0), "write_bytes": int(r["write_bytes"] or 0),
                        "fetch_size_calibration": calib, "avg_launch_us_rocprof": r["avg_us"], "source": os.path.basename(dst) + ".json", "workload": wl,
                        "commit": commit, "kernel_source": ksrc, "kernel_source_sha16": src_sha})
    if entries:
        tpath = os.path.join(os.path.dirname(dst) or ".", "traffic.json")
        try:
            old = json.load(open(tpath))["kernels"]
        except (OSError, KeyError, ValueError):
            old = []
        def same(a, b):
            return a["kernel"] == b["kernel"] and {k: v for k, v in a["workload"].items() if k != "join_table"} == {k: v for k, v in b["workload"].items() if k != "join_table"}
        kept = [o for o in old if not any(same(o, e) for e in entries)]
        json.dump({"kernels": kept + entries}, open(tpath, "w"), indent=1)
    print(open(dst + ".md").read())


if __name__ == "__main__":
    main()
