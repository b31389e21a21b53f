"""Second batch of the reference's HashJoinExec known-answer tests (tests/golden/hash_join_exec_more.json, made by
tests/golden/extract_reference_goldens_more.py): multi-partition inputs, empty sides, all-NULL build keys under every
join type, Date32 / i64::MIN..MAX keys, forced hash collisions and the null_aware (NOT IN) anti joins
(hash_join/exec.rs:429-455, stream.rs:755-808,937-955,1016-1076).  Oracle on the CPU, the HIP path on the GPU box."""
import numpy as np
import pyarrow as pa
import pytest

from tests.util import assert_tables_equal, load_golden, random_table, rows, sorted_rows

CASES = load_golden("hash_join_exec_more.json")
TYPES = {"date32": pa.date32(), "int64": pa.int64()}


def side(case, which):
    t, types = case[which], case.get("types", {})

    def column(name, values):
        kind = types.get(name, "int32")
        return pa.array(values, type=pa.int32()).cast(pa.date32()) if kind == "date32" else pa.array(values, type=TYPES.get(kind, pa.int32()))
    return pa.Table.from_arrays([column(n, v) for n, v in zip(t["columns"], t["data"])], names=t["columns"])


def plain(v):
    import datetime
    return (v - datetime.date(1970, 1, 1)).days if isinstance(v, datetime.date) else v


def check(case, out, ordered=None):
    assert out.column_names == case["expected_columns"], case["source"]
    if "expected_num_rows" in case:
        assert out.num_rows == case["expected_num_rows"], case["source"]
    got = [tuple(plain(v) for v in r) for r in rows(out)]
    expected = [tuple(r) for r in case["expected_rows"]]
    key = lambda row: tuple((v is None, 0 if v is None else v) for v in row)
    if case["ordered"] if ordered is None else ordered:
        assert got == expected, case["source"]
    else:
        assert sorted(got, key=key) == sorted(expected, key=key), case["source"]


@pytest.mark.parametrize("mode", [0, 1], ids=["phj_auto", "hash_map"])
@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_oracle_matches_reference(case, mode):
    from oracle import oracle
    out = oracle.hash_join(side(case, "left"), side(case, "right"), [tuple(p) for p in case["on"]], case["join_type"], case["null_equality"],
                           mode=mode, null_aware=case.get("null_aware", False))
    check(case, out)


def test_oracle_null_aware_validation():
    """HashJoinExec::try_new (exec.rs:429-455; tests :7588-7745)"""
    from oracle import oracle
    t = pa.table({"c1": pa.array([1], pa.int32()), "c2": pa.array([1], pa.int32())})
    with pytest.raises(ValueError, match="null_aware can only be true for LeftAnti joins and RightAnti joins"):
        oracle.hash_join(t, t, [("c1", "c1")], "Inner", null_aware=True)
    with pytest.raises(ValueError, match="null_aware anti join only supports single column join key"):
        oracle.hash_join(t, t, [("c1", "c1"), ("c2", "c2")], "LeftAnti", null_aware=True)
    with pytest.raises(ValueError, match="null_aware RightAnti join does not support a join filter"):
        oracle.hash_join(t, t, [("c1", "c1")], "RightAnti", null_aware=True, join_filter=(("bin", ">", ("col", "f0"), ("col", "f1")), [(1, "Left"), (1, "Right")]))


# ------------------------------------------------------------------------------------------------ GPU

def gpu_join(left, right, on, join_type, null_equality="NullEqualsNothing", **opts):
    from datafusion_amd import ops
    from datafusion_amd.table import DeviceTable
    return ops.hash_join(DeviceTable.from_arrow(left), DeviceTable.from_arrow(right), on, join_type, null_equality, **opts).to_arrow()


@pytest.mark.gpu
@pytest.mark.parametrize("opts", [dict(table_mode=0), dict(table_mode=1), dict(table_mode=1, force_hash_collisions=True), dict(table_mode=4),
                                  dict(table_mode=4, force_hash_collisions=True)],
                         ids=["phj_auto", "hash_map", "forced_collisions", "radix_lds", "radix_lds_forced_collisions"])
@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_gpu_matches_reference(case, opts):
    if case.get("force_hash_collisions"):
        opts = dict(opts, force_hash_collisions=True, table_mode=1)
    out = gpu_join(side(case, "left"), side(case, "right"), [tuple(p) for p in case["on"]], case["join_type"], case["null_equality"],
                   null_aware=case.get("null_aware", False), **opts)
    # the build rows matching one probe row come out in chain order, which a parallel build does not fix (the reference's
    # is ascending build index): compared as multisets, like tests/test_gpu_join.py::test_reference_snapshots
    check(case, out, ordered=False)


@pytest.mark.gpu
def test_gpu_null_aware_validation():
    from datafusion_amd import _lib, expr as X
    t = pa.table({"c1": pa.array([1], pa.int32()), "c2": pa.array([1], pa.int32())})
    with pytest.raises(_lib.DfgpuError, match="null_aware can only be true for LeftAnti joins and RightAnti joins"):
        gpu_join(t, t, [("c1", "c1")], "Inner", null_aware=True)
    with pytest.raises(_lib.DfgpuError, match="null_aware anti join only supports single column join key, got 2 columns"):
        gpu_join(t, t, [("c1", "c1"), ("c2", "c2")], "LeftAnti", null_aware=True)
    with pytest.raises(_lib.DfgpuError, match="null_aware RightAnti join does not support a join filter"):
        gpu_join(t, t, [("c1", "c1")], "RightAnti", null_aware=True, join_filter=(X.col("f0") > X.col("f1"), [(1, "Left"), (1, "Right")]))


@pytest.mark.gpu
@pytest.mark.parametrize("join_type", ["LeftAnti", "RightAnti"])
@pytest.mark.parametrize("nulls", ["none", "left", "right", "both"])
@pytest.mark.parametrize("table_mode", [0, 1, 4])
def test_gpu_null_aware_random_vs_oracle(join_type, nulls, table_mode):
    """NOT IN over a few thousand rows, NULL keys on either / both / neither side, empty sides, multi-batch probing"""
    from oracle import oracle
    rng = np.random.default_rng(len(join_type) * 10 + len(nulls))
    spec_l = {"a": (pa.int64(), 0, 900), "x": (pa.decimal128(15, 2), 0, 10**6)}
    spec_r = {"b": (pa.int64(), 0, 1200), "w": (pa.date32(), 8000, 9000)}
    left = random_table(rng, 2500, spec_l, null_frac=0.02 if nulls in ("left", "both") else 0.0)
    right = random_table(rng, 4100, spec_r, null_frac=0.001 if nulls in ("right", "both") else 0.0)
    for l, r in ((left, right), (left.slice(0, 0), right), (left, right.slice(0, 0))):
        got = gpu_join(l, r, [("a", "b")], join_type, null_aware=True, table_mode=table_mode)
        exp = oracle.hash_join(l, r, [("a", "b")], join_type, null_aware=True)
        assert_tables_equal(got, exp)


@pytest.mark.gpu
def test_gpu_null_aware_left_anti_flags_span_probe_tables():
    """probe_side_has_null / probe_side_non_empty are shared by every probe partition (JoinLeftData, stream.rs:769-803):
    a NULL key in ANY probe table empties the LeftAnti output; a filter fused below the probe side decides which NULLs count"""
    from datafusion_amd import ops, expr as X
    from datafusion_amd.table import DeviceTable
    left = DeviceTable.from_arrow(pa.table({"c1": pa.array([1, 4, None, 9], pa.int32()), "d": pa.array([10, 40, 0, 90], pa.int32())}))
    p1 = DeviceTable.from_arrow(pa.table({"c2": pa.array([1, 2], pa.int32()), "e": pa.array([1, 1], pa.int32())}))
    p2 = DeviceTable.from_arrow(pa.table({"c2": pa.array([None, 9], pa.int32()), "e": pa.array([0, 1], pa.int32())}))
    ht = ops.JoinHashTable(left, ["c1"], null_aware=True)
    ht.probe(p1, ["c2"], "LeftAnti")
    ht.probe(p2, ["c2"], "LeftAnti", predicate=X.col("e") > X.lit(0, pa.int32()))   # the NULL key is filtered out below the join
    assert rows(ht.emit_unmatched("LeftAnti").to_arrow()) == [(4, 40)]   # 1 and 9 matched, NULL dropped (probe non-empty)
    ht = ops.JoinHashTable(left, ["c1"], null_aware=True)
    ht.probe(p1, ["c2"], "LeftAnti")
    ht.probe(p2, ["c2"], "LeftAnti")
    assert ht.emit_unmatched("LeftAnti").num_rows == 0
    ht = ops.JoinHashTable(left, ["c1"], null_aware=True)                # no probe rows at all: NULL NOT IN (empty) is TRUE
    assert sorted_rows(ht.emit_unmatched("LeftAnti").to_arrow()) == [(1, 10), (4, 40), (9, 90), (None, 0)]
