"""The call sequences of the DataFusion-side shim (shim/src/hash_join.rs, operators.rs) executed from PLAIN C through include/dfgpu.h:
tests/c/plan_driver.c is compiled with gcc (strict C99), linked against libdfgpu.so and run as its own process — no Python, ctypes or
torch between the caller and the C ABI.  Python only writes the case files (from the reference's snapshot tests, tests/golden/) and
compares what the driver printed.

  join   builder push x k -> finish (ONE table per join) -> probe per probe partition -> emit_unmatched ONCE -> export_batch slices:
         the reference's join_* / partitioned_join_* / join_*_with_filter / null_aware snapshots (hash_join/exec.rs:3311-7411) with the
         build side arriving in 1 or 3 batches and the probe side split over 1 or 3 partitions (CollectLeft's shared build side,
         exec.rs:772,1312-1330,1503)
  chain  FilterExec -> AggregateExec -> SortExec over device handles + an ArrowDeviceArray hand-off: zero PCIe bytes in dfgpu_metrics
"""
import os
import subprocess

import pytest

from tests.util import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JOIN_TYPES = {"Inner": 0, "Left": 1, "Right": 2, "Full": 3, "LeftSemi": 4, "RightSemi": 5, "LeftAnti": 6, "RightAnti": 7, "LeftMark": 8, "RightMark": 9}
CMP = {"=": 20, "!=": 21, "<": 22, "<=": 23, ">": 24, ">=": 25}


def build_driver(tmp_path) -> str:
    exe = str(tmp_path / "plan_driver")
    lib_dir = os.path.join(ROOT, "datafusion_amd")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-D_POSIX_C_SOURCE=200809L", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "plan_driver.c"), "-o", exe, "-L", lib_dir, "-ldfgpu", f"-Wl,-rpath,{lib_dir}"])
    return exe


def test_c_driver_compiles_as_strict_c99_and_links(tmp_path):
    """CPU leg: the driver is plain C over the header alone and resolves every entry point it calls in libdfgpu.so"""
    exe = build_driver(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 3 and "usage" in out.stderr


def _cases():
    """int32-only snapshot cases of the reference (the driver's host tables are Int32, like build_table_i32)"""
    out = []
    for c in load_golden("hash_join_exec.json") + load_golden("hash_join_exec_more.json") + load_golden("hash_join_filter.json"):
        if "types" in c or c.get("force_hash_collisions") or "expected_rows" not in c:
            continue
        out.append(c)
    return out


def _table_text(t):
    rep = t.get("repeat", 1)
    cols = [list(v) * rep for v in t["data"]]
    n = len(cols[0]) if cols else 0
    lines = [f"table {len(cols)} {n}"]
    for name, vals in zip(t["columns"], cols):
        lines.append(name + " " + " ".join("N" if v is None else str(v) for v in vals))
    return "\n".join(lines)


def write_case(path, case, build_batches, probe_partitions, table_mode, batch_size=3):
    jt = case["join_type"]
    lcols, rcols = case["left"]["columns"], case["right"]["columns"]
    bo = [] if jt in ("RightSemi", "RightAnti", "RightMark") else list(range(len(lcols)))
    po = [] if jt in ("LeftSemi", "LeftAnti", "LeftMark") else list(range(len(rcols)))
    lines = [f"join_type {JOIN_TYPES[jt]}", f"null_equality {1 if case['null_equality'] == 'NullEqualsNull' else 0}", f"null_aware {int(case.get('null_aware', False))}",
             f"table_mode {table_mode}", f"build_batches {build_batches}", f"probe_partitions {probe_partitions}", f"batch_size {batch_size}",
             f"on {len(case['on'])} " + " ".join(f"{lcols.index(l)} {rcols.index(r)}" for l, r in case["on"]),
             f"build_out {len(bo)} " + " ".join(map(str, bo)), f"probe_out {len(po)} " + " ".join(map(str, po))]
    f = case.get("filter")
    if f is None:
        lines.append("filter 0")
    else:
        e = f["expr"]
        right_is_col = "right_col" in e
        lines.append(f"filter {len(f['columns'])} " + " ".join(f"{i} {0 if side == 'Left' else 1}" for i, side in f["columns"]) +
                     f" {CMP[e['op']]} {e['left']} {int(right_is_col)} {e['right_col'] if right_is_col else e['right_lit']}")
    lines += [_table_text(case["left"]), _table_text(case["right"])]
    open(path, "w").write("\n".join(lines) + "\n")


def parse_output(text):
    """{case path: (columns, rows, info)}"""
    out, cur = {}, None
    for line in text.splitlines():
        tok = line.split()
        if not tok:
            continue
        if tok[0] == "case":
            cur = out.setdefault(tok[1], dict(columns=None, rows=[], info=None))
        elif tok[0] == "columns":
            cur["columns"] = tok[1:]
        elif tok[0] == "row":
            cur["rows"].append(tuple(None if v == "NULL" else True if v == "true" else False if v == "false" else int(v) for v in tok[1:]))
        elif tok[0] == "info":
            cur["info"] = dict(zip(tok[1::2], map(int, tok[2::2])))
    return out


_key = lambda row: tuple((v is None, 0 if v is None else v) for v in row)


@pytest.mark.gpu
@pytest.mark.parametrize("table_mode", [0, 1], ids=["phj_auto", "hash_map"])
@pytest.mark.parametrize("shape", [(1, 1), (3, 1), (1, 3), (3, 3)], ids=["1_build_batch_1_probe_partition", "3_build_batches", "3_probe_partitions", "3_batches_3_partitions"])
def test_join_call_sequence_from_plain_c_matches_the_reference_snapshots(tmp_path, shape, table_mode):
    exe = build_driver(tmp_path)
    cases, paths = _cases(), []
    assert len(cases) >= 55
    for i, c in enumerate(cases):
        p = str(tmp_path / f"case_{i}.txt")
        write_case(p, c, shape[0], shape[1], table_mode)
        paths.append(p)
    run = subprocess.run([exe, "join"] + paths, capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stderr[-2000:]
    got = parse_output(run.stdout)
    assert len(got) == len(cases)
    seen_types = set()
    for p, c in zip(paths, cases):
        g = got[p]
        exp_rows = sorted([tuple(r) for r in c["expected_rows"]], key=_key)
        assert g["columns"] == c["expected_columns"], (c["name"], c["source"])
        assert sorted(g["rows"], key=_key) == exp_rows, (c["name"], c["source"], shape)
        assert g["info"]["rows_printed"] == len(exp_rows)
        seen_types.add(c["join_type"] + ("+filter" if "filter" in c else "") + ("+null_aware" if c.get("null_aware") else ""))
    for need in ("Left", "Full", "LeftAnti+filter", "LeftSemi+filter", "Left+filter", "Full+filter", "LeftMark", "RightMark", "LeftAnti+null_aware", "RightAnti+null_aware"):
        assert need in seen_types, need


@pytest.mark.gpu
def test_three_node_chain_from_plain_c_moves_no_table_bytes_over_pcie(tmp_path):
    """FilterExec -> (ArrowDeviceArray hand-off) -> AggregateExec -> SortExec: dfgpu_metrics reports h2d = d2h = 0 for the chain (the
    driver exits non-zero otherwise), and the rows it prints equal the oracle's over the same generated lineitem"""
    import datetime

    import pyarrow as pa

    from datafusion_amd import tpch
    from oracle import oracle
    exe = build_driver(tmp_path)
    sf = 0.05
    run = subprocess.run([exe, "chain", str(sf)], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stderr[-2000:]
    lines = run.stdout.splitlines()
    m = dict(zip(lines[0].split()[1::2], map(int, lines[0].split()[2::2])))
    assert lines[0].startswith("metrics") and m["h2d_bytes"] == 0 and m["d2h_bytes"] == 0 and m["calls"] >= 8
    assert lines[1].split() == ["columns", "l_returnflag", "l_linestatus", "sum_qty", "sum_base_price", "count_order"]
    got = [tuple(int(v) for v in l.split()[1:]) for l in lines[2:] if l.startswith("row")]
    li = tpch.lineitem(sf)
    f = oracle.filter(li, ("bin", "<=", ("col", "l_shipdate"), ("lit", datetime.date(1998, 9, 2), pa.date32())), li.column_names)
    assert m["rows_filtered"] == f.num_rows
    exp = oracle.aggregate(f, [(("col", "l_returnflag"), "l_returnflag"), (("col", "l_linestatus"), "l_linestatus")],
                           [("sum", ("col", "l_quantity"), "sum_qty"), ("sum", ("col", "l_extendedprice"), "sum_base_price"), ("count", None, "count_order")], "Single")
    exp = oracle.sort(exp, [("l_returnflag", False, False), ("l_linestatus", False, False)])
    unscaled = lambda d: int(d.scaleb(2))
    want = [(r["l_returnflag"], r["l_linestatus"], unscaled(r["sum_qty"]), unscaled(r["sum_base_price"]), r["count_order"]) for r in exp.to_pylist()]
    assert got == want


@pytest.mark.gpu
@pytest.mark.parametrize("nparts", [1, 4])
def test_partial_repartition_final_and_grouping_sets_from_plain_c(tmp_path, nparts):
    """the two-phase aggregate's call sequence (shim/src/operators.rs; aggregates/mod.rs:28-47) from plain C: three input partitions of two
    batches each -> AggregateExec(Partial) -> RepartitionExec(Hash(keys, P)) -> AggregateExec(FinalPartitioned) per output partition,
    with SUM / AVG(Decimal128) / COUNT(*) / MIN — and the same under GROUPING SETS ((flag), (status), (flag, status)), whose Final node
    groups by (keys, __grouping_id).  The printed rows equal the oracle's Single aggregates over the same generated lineitem."""
    from datafusion_amd import tpch
    from oracle import oracle
    exe = build_driver(tmp_path)
    sf = 0.03
    run = subprocess.run([exe, "partial_final", str(sf), str(nparts)], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stderr[-2000:]
    sections, cur = {}, None
    for ln in run.stdout.splitlines():
        w = ln.split()
        if ln in ("two_phase", "grouping_sets"):
            cur = sections.setdefault(ln, {"columns": None, "rows": []})
        elif w and w[0] == "columns":
            cur["columns"] = w[1:]
        elif w and w[0] == "row":
            cur["rows"].append(tuple(None if v == "NULL" else int(v) for v in w[1:]))
        elif w and w[0] == "rows":
            assert int(w[1]) == len(cur["rows"])
    li = tpch.lineitem(sf)
    gb = [(("col", "l_returnflag"), "l_returnflag"), (("col", "l_linestatus"), "l_linestatus")]
    aggs = [("sum", ("col", "l_quantity"), "sum_qty"), ("avg", ("col", "l_extendedprice"), "avg_price"), ("count", None, "count_order"), ("min", ("col", "l_quantity"), "min_qty")]
    exp = oracle.aggregate(li, gb, aggs, "Single")
    unscaled = lambda d, s: int(d.scaleb(s))
    want = sorted((r["l_returnflag"], r["l_linestatus"], unscaled(r["sum_qty"], 2), unscaled(r["avg_price"], 6), r["count_order"], unscaled(r["min_qty"], 2)) for r in exp.to_pylist())
    two = sections["two_phase"]
    assert two["columns"] == ["l_returnflag", "l_linestatus", "sum_qty", "avg_price", "count_order", "min_qty"]
    assert sorted(two["rows"]) == want
    # grouping sets: every set is the Single aggregate over its own keys, the other key NULL, __grouping_id = the NULLed columns' bits
    gs = sections["grouping_sets"]
    assert gs["columns"] == ["l_returnflag", "l_linestatus", "__grouping_id", "sum_qty", "avg_price", "count_order", "min_qty"]
    want_gs = [(a, b, 0) + tuple(rest) for (a, b, *rest) in [tuple(r) for r in want]]
    for keep, gid in ((0, 1), (1, 2)):   # (flag): linestatus NULLed -> bit 0; (status): returnflag NULLed -> bit 1
        e = oracle.aggregate(li, [gb[keep]], aggs, "Single")
        for r in e.to_pylist():
            key = (r["l_returnflag"], None) if keep == 0 else (None, r["l_linestatus"])
            want_gs.append(key + (gid, unscaled(r["sum_qty"], 2), unscaled(r["avg_price"], 6), r["count_order"], unscaled(r["min_qty"], 2)))
    norm = lambda rows: sorted(rows, key=lambda t: tuple((v is None, 0 if v is None else v) for v in t))
    assert norm(gs["rows"]) == norm([tuple(x) for x in want_gs])
