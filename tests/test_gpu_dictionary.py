"""Dictionary-encoded string columns (Arrow Dictionary(index, Utf8)) through the boundary: indices on the device,
dictionary on the host, handed on by every operator that selects or reorders rows, re-attached on export
(include/dfgpu.h dfgpu_table_dictionary_lookup; reference: group_values/multi_group_by/dictionary.rs)."""
import numpy as np
import pyarrow as pa
import pyarrow.compute as pc
import pytest

from tests.util import assert_tables_equal

pytestmark = pytest.mark.gpu


def dict_col(codes, values, index_type=pa.int32(), mask=None):
    return pa.DictionaryArray.from_arrays(pa.array(codes, type=index_type, mask=mask), pa.array(values, type=pa.string()))


def decoded(t: pa.Table) -> pa.Table:
    """dictionary columns -> plain strings (comparison form)"""
    return pa.table({n: (c.cast(pa.string()) if pa.types.is_dictionary(c.type) else c) for n, c in zip(t.column_names, t.columns)})


def test_import_export_round_trip_with_nulls_and_both_index_widths():
    from datafusion_amd.table import DeviceTable
    rng = np.random.default_rng(1)
    n = 10_001
    seg = dict_col(rng.integers(0, 5, n), ["AUTOMOBILE", "BUILDING", "FURNITURE", "HOUSEHOLD", "MACHINERY"], mask=rng.random(n) < 0.1)
    flag = dict_col(rng.integers(0, 3, n).astype(np.uint8), ["A", "N", "R"], pa.uint8())
    t = pa.table({"seg": seg, "flag": flag, "v": pa.array(rng.integers(0, 100, n), type=pa.int64())})
    back = DeviceTable.from_arrow(t).to_arrow()
    assert pa.types.is_dictionary(back.schema.field("seg").type) and back.schema.field("flag").type.index_type == pa.uint8()
    assert_tables_equal(decoded(back), decoded(t), ordered=True)
    assert_tables_equal(decoded(DeviceTable.from_arrow(t.slice(17, 4000)).to_arrow()), decoded(t.slice(17, 4000)), ordered=True)   # non-zero offset


def test_q1_with_string_flag_columns():
    """TPC-H Q1 with l_returnflag / l_linestatus as strings, the reference's schema (benchmarks/src/tpch/mod.rs:93-122)"""
    from datafusion_amd import queries, tpch
    from datafusion_amd.table import DeviceTable
    from tests.test_gpu_queries import oracle_q1
    li = tpch.lineitem(0.05)

    def as_strings(col):
        codes = col.to_numpy()
        vals = sorted(set(int(x) for x in codes))                      # ascending dictionary: ORDER BY on the column stays valid
        return dict_col(np.searchsorted(vals, codes).astype(np.uint8), [chr(v) for v in vals], pa.uint8())
    names = li.column_names
    t = pa.table({n: (as_strings(li.column(n)) if n in ("l_returnflag", "l_linestatus") else li.column(n)) for n in names})
    got = queries.q1(DeviceTable.from_arrow(t)).to_arrow()
    assert pa.types.is_dictionary(got.schema.field("l_returnflag").type)
    exp = oracle_q1(li)
    exp = exp.set_column(0, "l_returnflag", pa.array([chr(v) for v in exp.column("l_returnflag").to_pylist()])) \
             .set_column(1, "l_linestatus", pa.array([chr(v) for v in exp.column("l_linestatus").to_pylist()]))
    assert_tables_equal(decoded(got), exp, ordered=True)


def test_filter_on_a_dictionary_code_join_payload_group_key_sort_and_partition():
    from datafusion_amd import ops
    from datafusion_amd.expr import col, lit
    from datafusion_amd.table import DeviceTable
    rng = np.random.default_rng(5)
    segs = ["AUTOMOBILE", "BUILDING", "FURNITURE", "HOUSEHOLD", "MACHINERY"]
    nc, no = 3000, 20_000
    cust = pa.table({"c_custkey": pa.array(np.arange(nc), type=pa.int64()), "c_mktsegment": dict_col(rng.integers(0, 5, nc), segs)})
    orders = pa.table({"o_custkey": pa.array(rng.integers(0, nc, no), type=pa.int64()), "o_total": pa.array(rng.integers(1, 1000, no), type=pa.int64())})
    dc, do = DeviceTable.from_arrow(cust), DeviceTable.from_arrow(orders)
    # FilterExec: c_mktsegment = 'BUILDING' lowered to the literal's dictionary index
    code = dc.dictionary_code("c_mktsegment", "BUILDING")
    assert code == 1 and dc.dictionary_code("c_mktsegment", "nope") is None
    f = ops.filter(dc, col("c_mktsegment").eq(lit(code, pa.int32()))).to_arrow()
    assert_tables_equal(decoded(f), decoded(cust.filter(pc.equal(cust.column("c_mktsegment").cast(pa.string()), "BUILDING"))), ordered=True)
    # HashJoinExec with the string column as build-side payload, then AggregateExec grouped by it
    j = ops.hash_join(dc, do, [("c_custkey", "o_custkey")], "Inner")
    ja = j.to_arrow()
    assert pa.types.is_dictionary(ja.schema.field("c_mktsegment").type) and ja.num_rows == no
    want_seg = cust.column("c_mktsegment").cast(pa.string()).take(ja.column("o_custkey"))
    assert ja.column("c_mktsegment").cast(pa.string()).to_pylist() == want_seg.to_pylist()
    g = ops.aggregate(j, [(col("c_mktsegment"), "seg")], [("sum", col("o_total"), "s"), ("count", None, "n")], "Single").to_arrow()
    ref = decoded(ja).group_by("c_mktsegment").aggregate([("o_total", "sum"), ("o_total", "count")])
    assert sorted(zip(g.column("seg").cast(pa.string()).to_pylist(), g.column("s").to_pylist(), g.column("n").to_pylist())) == \
        sorted(zip(ref.column("c_mktsegment").to_pylist(), ref.column("o_total_sum").to_pylist(), ref.column("o_total_count").to_pylist()))
    # SortExec on another key keeps the payload's dictionary; ORDER BY the string column itself works because this dictionary is sorted
    s = ops.sort(dc, [("c_mktsegment", False, False), ("c_custkey", True, False)]).to_arrow()
    d = decoded(cust)
    order = pc.sort_indices(d, sort_keys=[("c_mktsegment", "ascending"), ("c_custkey", "descending")])
    assert_tables_equal(decoded(s), d.take(order), ordered=True)
    parts = ops.partition(dc, ["c_custkey"], 3)
    assert sum(p.num_rows for p in parts) == nc and all(pa.types.is_dictionary(p.to_arrow().schema.field("c_mktsegment").type) for p in parts)


def test_rejected_dictionaries():
    from datafusion_amd import _lib, ops
    from datafusion_amd.table import DeviceTable
    dup = pa.table({"s": dict_col([0, 1, 2], ["a", "b", "a"])})
    with pytest.raises(_lib.DfgpuError, match="duplicate dictionary value"):
        DeviceTable.from_arrow(dup)
    unsorted = DeviceTable.from_arrow(pa.table({"s": dict_col([0, 1, 2, 1], ["b", "a", "c"]), "v": pa.array([1, 2, 3, 4], type=pa.int64())}))
    with pytest.raises(_lib.DfgpuError, match="dictionary in ascending order"):
        ops.sort(unsorted, [("s", False, False)])
    assert ops.sort(unsorted, [("v", True, False)]).to_arrow().column("s").cast(pa.string()).to_pylist() == ["a", "c", "a", "b"]


def test_string_literal_predicates_are_bound_to_dictionary_indices():
    """FilterExec / fused aggregate predicate / fused probe-side predicate written with the string itself"""
    from datafusion_amd import ops
    from datafusion_amd.expr import col, lit
    from datafusion_amd.table import DeviceTable
    rng = np.random.default_rng(9)
    n = 30_000
    segs = ["AUTOMOBILE", "BUILDING", "FURNITURE", "HOUSEHOLD", "MACHINERY"]
    t = pa.table({"k": pa.array(rng.integers(0, 500, n), type=pa.int64()), "seg": dict_col(rng.integers(0, 5, n).astype(np.uint8), segs, pa.uint8(), mask=rng.random(n) < 0.1),
                  "v": pa.array(rng.integers(0, 1000, n), type=pa.int64())})
    dt = DeviceTable.from_arrow(t)
    plain = decoded(t)
    is_b = pc.equal(plain.column("seg"), "BUILDING")
    for pred, mask in ((col("seg").eq(lit("BUILDING", pa.string())), is_b), (col("seg").ne(lit("BUILDING", pa.string())), pc.invert(is_b)),
                       (col("seg").eq(lit("NOT THERE", pa.string())), pc.and_(is_b, pc.invert(is_b))),
                       (col("seg").ne(lit("NOT THERE", pa.string())), pc.is_valid(plain.column("seg"))),
                       (col("seg").eq(lit("BUILDING", pa.string())).or_(col("v") > lit(990)), pc.or_kleene(is_b, pc.greater(plain.column("v"), 990)))):
        got = ops.filter(dt, pred).to_arrow()
        assert_tables_equal(decoded(got), plain.filter(mask), ordered=True)          # NULL predicate rows are dropped on both sides
    agg = ops.aggregate(dt, [], [("sum", col("v"), "s"), ("count", None, "n")], "Single", predicate=col("seg").eq(lit("MACHINERY", pa.string()))).to_arrow()
    m = plain.filter(pc.equal(plain.column("seg"), "MACHINERY"))
    assert agg.to_pylist() == [{"s": pc.sum(m.column("v")).as_py(), "n": m.num_rows}]
    build = DeviceTable.from_arrow(pa.table({"bk": pa.array(np.arange(500), type=pa.int64())}))
    ht = ops.JoinHashTable(build, ["bk"], probe_mode=3)
    j = ht.probe(dt, ["k"], "Inner", ["bk"], ["k", "v"], predicate=col("seg").eq(lit("FURNITURE", pa.string()))).to_arrow()
    assert j.num_rows == plain.filter(pc.equal(plain.column("seg"), "FURNITURE")).num_rows


@pytest.mark.parametrize("table_mode", [0, 1])
@pytest.mark.parametrize("join_type", ["Inner", "RightAnti", "LeftSemi"])
def test_join_on_string_keys_with_different_dictionaries(join_type, table_mode):
    """two tables encode the same strings with different dictionaries (every Parquet file / table brings its own): the join
    must compare strings, not raw indices — the probe side's indices are rewritten into the build side's dictionary"""
    from datafusion_amd import ops
    from datafusion_amd.table import DeviceTable
    from oracle import oracle
    rng = np.random.default_rng(7)
    bvals, pvals = ["pear", "apple", "fig", "kiwi"], ["apple", "banana", "cherry", "fig", "grape", "pear"]
    nb, np_ = 50, 4000
    b = pa.table({"k": dict_col(rng.integers(0, len(bvals), nb), bvals), "bi": pa.array(np.arange(nb), pa.int64())})
    p = pa.table({"k2": dict_col(rng.integers(0, len(pvals), np_), pvals, mask=rng.random(np_) < 0.05), "pi": pa.array(np.arange(np_), pa.int64())})
    got = ops.hash_join(DeviceTable.from_arrow(b), DeviceTable.from_arrow(p), [("k", "k2")], join_type, table_mode=table_mode).to_arrow()
    # the oracle joins integer keys: one global code per distinct string stands for the string on both sides
    code = {v: i for i, v in enumerate(sorted(set(bvals) | set(pvals)))}

    def coded(t, name):
        return t.set_column(t.schema.get_field_index(name), name, pa.array([None if v is None else code[v] for v in decoded(t).column(name).to_pylist()], pa.int32()))
    exp = oracle.hash_join(coded(b, "k"), coded(p, "k2"), [("k", "k2")], join_type)
    ids = [n for n in ("bi", "pi") if n in exp.column_names]
    assert got.num_rows == exp.num_rows
    assert_tables_equal(got.select(ids), exp.select(ids), ordered=False)
    if "k" in got.column_names and "k2" in got.column_names:
        assert decoded(got).column("k").to_pylist() == decoded(got).column("k2").to_pylist()


def test_comparing_columns_of_different_dictionaries_is_refused():
    from datafusion_amd import _lib, ops
    from datafusion_amd.expr import col
    from datafusion_amd.table import DeviceTable
    t = pa.table({"a": dict_col([0, 1, 0], ["x", "y"]), "b": dict_col([0, 1, 1], ["y", "x"])})
    with pytest.raises(_lib.DfgpuError, match="different dictionaries"):
        ops.filter(DeviceTable.from_arrow(t), col("a").eq(col("b")))
    same = pa.table({"a": dict_col([0, 1, 0], ["x", "y"]), "b": dict_col([0, 0, 1], ["x", "y"])})
    assert ops.filter(DeviceTable.from_arrow(same), col("a").eq(col("b"))).num_rows == 1


def test_group_keys_with_a_changed_dictionary_are_refused():
    from datafusion_amd import _lib, ops
    from datafusion_amd.expr import col
    from datafusion_amd.table import DeviceTable
    t1 = pa.table({"k": dict_col([0, 1, 0], ["x", "y"]), "v": pa.array([1, 2, 3], pa.int64())})
    t2 = pa.table({"k": dict_col([0, 1, 1], ["y", "z"]), "v": pa.array([1, 2, 3], pa.int64())})
    agg = ops.GroupedAggregate("Single", ["k", "v"], [(col("k"), "k")], [("sum", col("v"), "s")])
    agg.update(DeviceTable.from_arrow(t1))
    with pytest.raises(_lib.DfgpuError, match="dictionary changed"):
        agg.update(DeviceTable.from_arrow(t2))


@pytest.mark.gpu
@pytest.mark.parametrize("pattern,negated,ci", [("%green%", False, False), ("forest%", False, False), ("%e_ %", True, False), ("%GREEN%", False, True), ("%nothing%", False, False)])
def test_like_over_a_dictionary_with_one_value_per_row(pattern, negated, ci):
    """LIKE over a dictionary-encoded column whose matching indices do not form a few runs (TPC-H p_name: one value per part): the
    library matches the pattern against the dictionary and the rows look their index up; NULL rows stay NULL"""
    import numpy as np
    import pyarrow.compute as pc
    from datafusion_amd import ops
    from datafusion_amd.expr import col
    from datafusion_amd.table import DeviceTable
    from oracle import dbgen
    names = dbgen.part_names(6000)
    rng = np.random.default_rng(2)
    arr = pa.array(names, pa.string(), mask=rng.random(6000) < 0.05)
    t = pa.table({"p_name": arr.dictionary_encode(), "k": pa.array(np.arange(6000))})
    got = ops.filter(DeviceTable.from_arrow(t), col("p_name").like(pattern, negated=negated, case_insensitive=ci)).to_arrow()
    hit = pc.match_like(arr, pattern, ignore_case=ci)
    if negated:
        hit = pc.invert(hit)
    want = [k for k, h in zip(range(6000), hit.to_pylist()) if h]
    assert got.column("k").to_pylist() == want
