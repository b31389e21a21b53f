"""K6/K7 parity: AggregateExec on the GPU vs the reference's check_aggregates snapshot and the CPU
oracle.  Integer / Decimal128 aggregates bit-exact; Float64 SUM/AVG within 1e-6 relative
(accumulation order differs on the GPU)."""
import math
from decimal import Decimal

import numpy as np
import pyarrow as pa
import pytest

from tests.util import load_golden, random_table, rows, sorted_rows, to_oracle_expr

pytestmark = pytest.mark.gpu
G = load_golden("aggregate_check_aggregates.json")
REL = 1e-6  # north_star tolerance for SUM/AVG(float64)


def gpu_agg(table, group_by, aggs, mode="Single", return_types=None):
    from datafusion_amd import ops
    from datafusion_amd.table import DeviceTable
    return ops.aggregate(DeviceTable.from_arrow(table), group_by, aggs, mode, return_types=return_types).to_arrow()


def oracle_agg(table, group_by, aggs, mode="Single"):
    from oracle import oracle
    return oracle.aggregate(table, [(to_oracle_expr(e), n) for e, n in group_by],
                            [(f, None if e is None else to_oracle_expr(e), n) for f, e, n in aggs], mode)


def assert_agg_equal(got, exp, ordered=True):
    assert got.column_names == exp.column_names
    for fa, fe in zip(got.schema, exp.schema):
        assert fa.type == fe.type, (fa, fe)
    gr, er = (rows(got), rows(exp)) if ordered else (sorted_rows(got), sorted_rows(exp))
    assert len(gr) == len(er)
    for a, b in zip(gr, er):
        for x, y in zip(a, b):
            if isinstance(y, float) and x is not None:
                assert math.isclose(x, y, rel_tol=REL, abs_tol=1e-12), (a, b)
            else:
                assert x == y, (a, b)


def test_reference_check_aggregates_partial_and_final():
    from datafusion_amd.expr import col
    t = pa.table({"a": pa.array(G["input"]["a"], type=pa.uint32()), "b": pa.array(G["input"]["b"], type=pa.float64())})
    gb, aggs = [(col("a"), "a")], [("avg", col("b"), "AVG(b)")]
    partial = gpu_agg(t, gb, aggs, "Partial")
    assert partial.column_names == G["partial"]["columns"]
    assert sorted_rows(partial) == [tuple(r) for r in G["partial"]["rows"]]
    parts = [gpu_agg(t.slice(lo, hi - lo), gb, aggs, "Partial") for lo, hi in G["batches"]]
    final = gpu_agg(pa.concat_tables(parts), gb, aggs, "Final")
    assert final.column_names == G["final"]["columns"]
    assert sorted_rows(final) == [tuple(r) for r in G["final"]["rows"]]
    assert sorted_rows(gpu_agg(t, gb, aggs, "Single")) == [tuple(r) for r in G["final"]["rows"]]


@pytest.mark.parametrize("ngroups", [1, 4, 300, 5000, 200_000])
def test_group_by_int_key_all_functions(ngroups):
    from datafusion_amd.expr import col
    rng = np.random.default_rng(ngroups)
    t = random_table(rng, 300_000, {"k": (pa.int64(), 0, ngroups), "d": (pa.decimal128(15, 2), -10**9, 10**9), "i": (pa.int32(), -1000, 1000),
                                    "f": (pa.float64(), -10**6, 10**6), "dt": (pa.date32(), 8000, 10000)}, null_frac=0.1)
    aggs = [("sum", col("d"), "sd"), ("avg", col("d"), "ad"), ("min", col("d"), "mind"), ("max", col("d"), "maxd"), ("sum", col("i"), "si"),
            ("count", col("i"), "ci"), ("count", None, "cstar"), ("sum", col("f"), "sf"), ("avg", col("f"), "af"), ("min", col("f"), "minf"),
            ("max", col("dt"), "maxdt")]
    got = gpu_agg(t, [(col("k"), "k")], aggs)
    exp = oracle_agg(t, [(col("k"), "k")], aggs)
    # group ids in first-seen order, like the reference (group_values/mod.rs:88-92)
    assert_agg_equal(got, exp, ordered=True)


def test_multi_column_group_keys_q3_shape():
    """Q3 aggregate: GROUP BY (l_orderkey Int64, o_orderdate Date32, o_shippriority Int32), SUM(Decimal128(38,4))"""
    from datafusion_amd.expr import col, lit
    rng = np.random.default_rng(3)
    t = random_table(rng, 100_000, {"l_orderkey": (pa.int64(), 0, 20_000), "o_orderdate": (pa.date32(), 9000, 9003), "o_shippriority": (pa.int32(), 0, 2),
                                    "p": (pa.decimal128(15, 2), 90000, 10_000_000), "d": (pa.decimal128(15, 2), 0, 11)})
    rev = col("p") * (lit(1, pa.decimal128(20, 0)) - col("d"))
    gb = [(col("l_orderkey"), "l_orderkey"), (col("o_orderdate"), "o_orderdate"), (col("o_shippriority"), "o_shippriority")]
    got = gpu_agg(t, gb, [("sum", rev, "revenue")])
    exp = oracle_agg(t, gb, [("sum", rev, "revenue")])
    assert got.schema.field("revenue").type == pa.decimal128(38, 4)
    assert_agg_equal(got, exp)


def test_tpch_q1_aggregate_small_sf():
    """BASELINE config 4 at a scale the oracle finishes in seconds: 8 aggregates, 4 groups"""
    from datafusion_amd import ops, tpch
    from datafusion_amd.expr import col, lit
    sf = 0.01
    li = tpch.lineitem(sf)
    one = lit(1, pa.decimal128(20, 0))
    disc = col("l_extendedprice") * (one - col("l_discount"))
    aggs = [("sum", col("l_quantity"), "sum_qty"), ("sum", col("l_extendedprice"), "sum_base_price"), ("sum", disc, "sum_disc_price"),
            ("sum", disc * (one + col("l_tax")), "sum_charge"), ("avg", col("l_quantity"), "avg_qty"), ("avg", col("l_extendedprice"), "avg_price"),
            ("avg", col("l_discount"), "avg_disc"), ("count", None, "count_order")]
    gb = [(col("l_returnflag"), "l_returnflag"), (col("l_linestatus"), "l_linestatus")]
    got = ops.aggregate(ops.tpch_lineitem(sf), gb, aggs).to_arrow()
    exp = oracle_agg(li, gb, aggs)
    assert got.num_rows == 4
    assert got.schema.field("sum_charge").type == pa.decimal128(38, 6) and got.schema.field("avg_disc").type == pa.decimal128(19, 6)
    assert_agg_equal(got, exp)


def test_partial_final_composition_matches_single():
    """GPU Partial -> (concat of partitions) -> GPU Final == Single; state schema = the reference's state_fields"""
    from datafusion_amd.expr import col
    rng = np.random.default_rng(12)
    t = random_table(rng, 50_000, {"k": (pa.int32(), 0, 700), "d": (pa.decimal128(15, 2), -10**8, 10**8), "f": (pa.float64(), 0, 1000)}, null_frac=0.05)
    gb = [(col("k"), "k")]
    aggs = [("sum", col("d"), "s"), ("avg", col("d"), "a"), ("count", col("f"), "c"), ("min", col("d"), "mn"), ("max", col("f"), "mx"), ("avg", col("f"), "af")]
    parts = [gpu_agg(t.slice(o, 12_500), gb, aggs, "Partial") for o in range(0, 50_000, 12_500)]
    assert parts[0].column_names == ["k", "s[sum]", "a[count]", "a[sum]", "c[count]", "mn[value]", "mx[value]", "af[count]", "af[sum]"]
    exp_partial = oracle_agg(t.slice(0, 12_500), gb, aggs, "Partial")
    assert_agg_equal(parts[0], exp_partial)
    assert parts[0].schema.field("a[sum]").type == pa.decimal128(38, 2)       # avg_sum_data_type, average.rs:131-172
    with pytest.raises(Exception, match="declared return type"):
        gpu_agg(pa.concat_tables(parts), gb, aggs, "Final")
    final = gpu_agg(pa.concat_tables(parts), gb, aggs, "Final", return_types={"a": pa.decimal128(19, 6)})
    single = oracle_agg(t, gb, aggs, "Single")
    assert_agg_equal(final, single, ordered=False)


def test_no_group_by_and_empty_input():
    from datafusion_amd.expr import col
    t = random_table(np.random.default_rng(1), 10_000, {"d": (pa.decimal128(15, 2), 0, 10**6), "i": (pa.int64(), 0, 100)}, null_frac=0.2)
    aggs = [("sum", col("d"), "s"), ("count", None, "c"), ("avg", col("i"), "a"), ("max", col("i"), "m")]
    assert_agg_equal(gpu_agg(t, [], aggs), oracle_agg(t, [], aggs))
    empty = t.slice(0, 0)
    assert_agg_equal(gpu_agg(empty, [], aggs), oracle_agg(empty, [], aggs))          # one row: NULL sums, count 0
    assert gpu_agg(empty, [(col("i"), "i")], aggs).num_rows == 0


def test_group_by_without_aggregates_is_distinct():
    """gby=[...], aggr=[] — the inner level of COUNT(DISTINCT x) (q16.slt.part:75-77: Partial and FinalPartitioned with no
    aggregate): the distinct key rows in first-seen order, then COUNT(column) over them per outer group"""
    from datafusion_amd.expr import col
    rng = np.random.default_rng(16)
    t = random_table(rng, 40_000, {"a": (pa.int32(), 0, 40), "b": (pa.int64(), 0, 300), "d": (pa.decimal128(15, 2), 0, 5)}, null_frac=0.03)
    gb = [(col("a"), "a"), (col("d"), "d"), (col("b"), "alias1")]
    for mode in ("Single", "Partial"):
        assert_agg_equal(gpu_agg(t, gb, [], mode), oracle_agg(t, gb, [], mode))
    parts = [gpu_agg(t.slice(o, 10_000), gb, [], "Partial") for o in range(0, 40_000, 10_000)]
    distinct = gpu_agg(pa.concat_tables(parts), gb, [], "FinalPartitioned")
    assert_agg_equal(distinct, oracle_agg(t, gb, [], "Single"))
    assert distinct.num_rows == len({(r["a"], r["d"], r["b"]) for r in t.to_pylist()})
    outer = [(col("a"), "a"), (col("d"), "d")]
    cnt = [("count", col("alias1"), "count(alias1)")]
    assert_agg_equal(gpu_agg(distinct, outer, cnt), oracle_agg(distinct, outer, cnt))
    assert gpu_agg(t.slice(0, 0), gb, [], "Single").num_rows == 0


def test_wrapping_i128_sum_is_order_independent():
    """SUM(Decimal128) uses add_wrapping (sum.rs:308-320): overflow wraps identically on GPU and CPU"""
    from datafusion_amd.expr import col
    big = Decimal("9" * 38)
    t = pa.table({"k": pa.array([1] * 64 + [2] * 64, type=pa.int32()), "d": pa.array([big] * 128, type=pa.decimal128(38, 0))})
    aggs = [("sum", col("d"), "s")]
    got, exp = gpu_agg(t, [(col("k"), "k")], aggs), oracle_agg(t, [(col("k"), "k")], aggs)
    from oracle import oracle
    assert (oracle.values_np(got.column("s")) == oracle.values_np(exp.column("s"))).all()


def test_min_max_over_wide_decimals_check_that_values_fit():
    """MIN / MAX over Decimal128(38, 4) (TPC-H Q15: MAX over a SUM's type): 64-bit atomics after a device-side check that every
    value is representable in 64 bits; a value beyond that is an error, never a wrong answer"""
    from datafusion_amd import _lib, ops
    from datafusion_amd.expr import col
    from datafusion_amd.table import DeviceTable
    from decimal import Decimal
    rng = np.random.default_rng(8)
    n = 5000
    vals = [Decimal(int(x)).scaleb(-4) for x in rng.integers(-10**17, 10**17, n)]
    t = pa.table({"g": pa.array(rng.integers(0, 40, n)), "d": pa.array(vals, pa.decimal128(38, 4))})
    got = ops.aggregate(DeviceTable.from_arrow(t), [(col("g"), "g")], [("max", col("d"), "mx"), ("min", col("d"), "mn")], "Single").to_arrow()
    want = {}
    for g, v in zip(t.column("g").to_pylist(), vals):
        a, b = want.get(g, (v, v))
        want[g] = (max(a, v), min(b, v))
    assert got.schema.field("mx").type == pa.decimal128(38, 4)
    assert {r["g"]: (r["mx"], r["mn"]) for r in got.to_pylist()} == want
    one = ops.aggregate(DeviceTable.from_arrow(t), [], [("max", col("d"), "mx")], "Single").to_arrow()
    assert one.column("mx")[0].as_py() == max(vals)
    big = pa.table({"d": pa.array([Decimal(1), Decimal(2**70)], pa.decimal128(38, 4))})
    with pytest.raises(_lib.DfgpuError, match="does not fit in 64 bits"):
        ops.aggregate(DeviceTable.from_arrow(big), [], [("max", col("d"), "mx")], "Single")


def _oracle_grouping_sets(table, group_by, null_types, groups, aggs, mode):
    """the oracle's aggregate per grouping set: keys NULLed out by the set, `__grouping_id` appended (PhysicalGroupBy semantics)"""
    from oracle import oracle
    n = len(group_by)
    parts = []
    for g in groups:
        gid = sum(1 << (n - 1 - i) for i, nulled in enumerate(g) if nulled)
        t = table
        keys = []
        for i, ((e, name), nulled) in enumerate(zip(group_by, g)):
            if nulled:
                t = t.append_column(f"__null_{i}", pa.nulls(t.num_rows, null_types[i]))
                keys.append((("col", f"__null_{i}"), name))
            else:
                keys.append((to_oracle_expr(e), name))
        t = t.append_column("__gid", pa.array(np.full(t.num_rows, gid, dtype=np.uint8)))
        keys.append((("col", "__gid"), "__grouping_id"))
        parts.append(oracle.aggregate(t, keys, [(f, None if e is None else to_oracle_expr(e), nm) for f, e, nm in aggs], mode))
    return pa.concat_tables(parts)


def test_reference_check_grouping_sets_partial_and_final():
    """aggregates/mod.rs check_grouping_sets (:3428-3590): GROUPING SETS ((a), (b), (a, b)) with COUNT(1) over some_data(), the
    reference's Partial and Final snapshots (tests/golden/aggregate_grouping_sets.json, extracted mechanically)"""
    from datafusion_amd import ops
    from datafusion_amd.expr import col, lit
    from datafusion_amd.table import DeviceTable
    S = load_golden("aggregate_grouping_sets.json")
    t = pa.table({"a": pa.array(G["input"]["a"], type=pa.uint32()), "b": pa.array(G["input"]["b"], type=pa.float64())})
    gb = [(col("a"), "a"), (col("b"), "b")]
    nulls = [lit(None, pa.uint32()), lit(None, pa.float64())]
    aggs = [("count", None, "COUNT(1)")]       # count(lit(1)) in the reference: a non-NULL literal counts rows
    key = lambda row: tuple((v is None, 0 if v is None else v) for v in row)
    partial = ops.aggregate_grouping_sets(DeviceTable.from_arrow(t), gb, nulls, S["groups"], aggs, "Partial").to_arrow()
    assert partial.column_names == S["partial"]["columns"]
    assert partial.schema.field("__grouping_id").type == pa.uint8() and partial.schema.field("a").type == pa.uint32()
    assert sorted(rows(partial), key=key) == sorted([tuple(r) for r in S["partial"]["rows"]], key=key)
    # Final over the partial states of two input batches: a plain aggregate grouping by (a, b, __grouping_id)
    parts = [ops.aggregate_grouping_sets(DeviceTable.from_arrow(t.slice(lo, hi - lo)), gb, nulls, S["groups"], aggs, "Partial").to_arrow() for lo, hi in G["batches"]]
    final = gpu_agg(pa.concat_tables(parts), gb + [(col("__grouping_id"), "__grouping_id")], aggs, "Final")
    assert final.column_names == S["final"]["columns"]
    assert sorted(rows(final), key=key) == sorted([tuple(r) for r in S["final"]["rows"]], key=key)
    single = ops.aggregate_grouping_sets(DeviceTable.from_arrow(t), gb, nulls, S["groups"], aggs, "Single").to_arrow()
    assert sorted(rows(single), key=key) == sorted([tuple(r) for r in S["final"]["rows"]], key=key)


@pytest.mark.parametrize("shape", ["rollup_3", "cube_2_with_predicate", "nullable_keys"])
def test_grouping_sets_vs_oracle(shape):
    """ROLLUP / CUBE shapes with several aggregate functions, a fused predicate and keys that are themselves NULL (a NULL key of the
    data and a key NULLed out by the set differ in `__grouping_id` only) against the oracle's per-set aggregates"""
    from datafusion_amd import ops
    from datafusion_amd.expr import col, lit
    from datafusion_amd.table import DeviceTable
    rng = np.random.default_rng(len(shape))
    n = 60_000
    t = random_table(rng, n, {"k1": (pa.int64(), 0, 7), "k2": (pa.int32(), 0, 5), "k3": (pa.uint8(), 0, 3), "d": (pa.decimal128(15, 2), -10**6, 10**6), "f": (pa.float64(), -100, 100)},
                     null_frac=0.15 if shape == "nullable_keys" else 0.0)
    aggs = [("sum", col("d"), "s"), ("avg", col("f"), "af"), ("count", None, "n"), ("min", col("d"), "mn"), ("max", col("f"), "mx")]
    if shape == "rollup_3":
        gb = [(col("k1"), "k1"), (col("k2"), "k2"), (col("k3"), "k3")]
        groups = [[False, False, False], [False, False, True], [False, True, True], [True, True, True]]
    else:
        gb = [(col("k1"), "k1"), (col("k2"), "k2")]
        groups = [[False, False], [False, True], [True, False], [True, True]]
    types = [t.schema.field(nm).type for _, nm in gb]
    pred = (col("f") > lit(0.0, pa.float64())) if shape == "cube_2_with_predicate" else None
    got = ops.aggregate_grouping_sets(DeviceTable.from_arrow(t), gb, [lit(None, ty) for ty in types], groups, aggs, "Single", predicate=pred).to_arrow()
    host = t
    if pred is not None:
        from oracle import oracle
        host = oracle.filter(t, to_oracle_expr(pred), t.column_names)
    exp = _oracle_grouping_sets(host, gb, types, groups, aggs, "Single")
    assert_agg_equal(got, exp, ordered=False)


def test_partial_reduce_merges_states_into_states():
    """AggregateMode::PartialReduce (aggregates/mod.rs:340-361): partial states in, partial states out — a tree of
    Partial -> PartialReduce -> Final gives what Single gives, and PartialReduce's output has Partial's schema"""
    from datafusion_amd.expr import col
    rng = np.random.default_rng(41)
    t = random_table(rng, 40_000, {"k": (pa.int64(), 0, 300), "d": (pa.decimal128(15, 2), -10**6, 10**6), "f": (pa.float64(), -100, 100), "q": (pa.int32(), -50, 50)}, null_frac=0.1)
    gb = [(col("k"), "k")]
    aggs = [("sum", col("d"), "s"), ("avg", col("d"), "ad"), ("avg", col("f"), "af"), ("count", col("q"), "c"), ("count", None, "n"), ("min", col("q"), "mn"), ("max", col("f"), "mx")]
    from datafusion_amd import ops
    from datafusion_amd.table import DeviceTable
    rt = ops.aggregate_return_types(DeviceTable.from_arrow(t), aggs)
    leaves = [gpu_agg(t.slice(lo, 10_000), gb, aggs, "Partial") for lo in range(0, 40_000, 10_000)]
    mids = [gpu_agg(pa.concat_tables(leaves[i:i + 2]), gb, aggs, "PartialReduce") for i in (0, 2)]
    for m in mids:
        assert m.schema == leaves[0].schema                       # states in, states out
    final = gpu_agg(pa.concat_tables(mids), gb, aggs, "Final", return_types=rt)
    single = gpu_agg(t, gb, aggs, "Single")
    assert_agg_equal(final, single, ordered=False)
    assert_agg_equal(single, oracle_agg(t, gb, aggs, "Single"), ordered=False)


@pytest.mark.parametrize("null_frac", [0.0, 0.2])
def test_boolean_group_keys(null_frac):
    """GROUP BY over Boolean columns (alone, with another key, through Partial -> Final): TRUE / FALSE / NULL groups in first-seen order"""
    from datafusion_amd.expr import col
    rng = np.random.default_rng(5)
    n = 30_000
    t = pa.table({"b": pa.array(rng.random(n) < 0.3, mask=(rng.random(n) < null_frac) if null_frac else None), "c": pa.array(rng.random(n) < 0.5),
                  "k": pa.array(rng.integers(0, 4, n)), "v": pa.array(rng.integers(-100, 100, n).astype(np.int32), mask=rng.random(n) < 0.1)})
    aggs = [("sum", col("v"), "s"), ("count", None, "n"), ("avg", col("v"), "a")]
    as_bytes = pa.table({c: (t.column(c).cast(pa.uint8()) if c in ("b", "c") else t.column(c)) for c in t.column_names})     # the oracle groups by the bytes 0 / 1

    def expected(gb):
        e = oracle_agg(as_bytes, gb, aggs, "Single")
        return pa.table({c: (e.column(c).cast(pa.bool_()) if c in ("b", "c") else e.column(c)) for c in e.column_names})
    for gb in ([(col("b"), "b")], [(col("b"), "b"), (col("k"), "k")], [(col("c"), "c"), (col("b"), "b")]):
        got = gpu_agg(t, gb, aggs, "Single")
        assert got.schema.field("b").type == pa.bool_()
        assert_agg_equal(got, expected(gb), ordered=True)
    gb = [(col("b"), "b"), (col("k"), "k")]
    parts = [gpu_agg(t.slice(lo, 10_000), gb, aggs, "Partial") for lo in range(0, n, 10_000)]
    assert_agg_equal(gpu_agg(pa.concat_tables(parts), gb, aggs, "Final"), expected(gb), ordered=False)


@pytest.mark.gpu
def test_dense_key_node_takes_a_known_key_range_without_a_pass():
    """GROUP BY on a dictionary-encoded column (codes lie in [0, dictionary size)) or on a column whose min / max are cached needs
    no pass over the keys for their range: the dense-key node goes straight to its bitmap — same groups, same sums, in first-seen order"""
    from datafusion_amd import ops
    from datafusion_amd.expr import col
    from datafusion_amd.table import DeviceTable
    rng = np.random.default_rng(77)
    n, distinct = 5_000_000, 40_000
    codes = rng.integers(0, distinct, n)
    names = pa.array([f"name{i:06d}" for i in range(distinct)])
    v = rng.integers(-1000, 1000, n)
    t = DeviceTable.from_arrow(pa.table({"s": pa.DictionaryArray.from_arrays(pa.array(codes.astype(np.int32)), names), "k": pa.array(codes * 3 + 11), "v": pa.array(v)}))
    want = {}
    first = {}
    for i, (c, x) in enumerate(zip(codes.tolist(), v.tolist())):
        want[c] = want.get(c, 0) + x
        first.setdefault(c, i)
    order = sorted(want, key=lambda c: first[c])
    for key, label, convert in (("s", "dictionary", lambda c: f"name{c:06d}"), ("k", "cached statistics", lambda c: c * 3 + 11)):
        if key == "k":
            assert ops.column_minmax(t, "k")[:2] == (11, (distinct - 1) * 3 + 11)       # fills the column's cached statistics
        ops.profile_enable(True)
        ops.profile_reset()
        got = ops.aggregate(t, [(col(key), key)], [("sum", col("v"), "sv")], "Single").to_arrow()
        stats = ops.profile_stats()
        ops.profile_enable(False)
        assert "agg_dense_accumulate" in stats and "agg_dense_key_range" not in stats, (label, sorted(stats))
        assert got.column(key).to_pylist() == [convert(c) for c in order], label
        assert got.column("sv").to_pylist() == [want[c] for c in order], label


@pytest.mark.gpu
@pytest.mark.parametrize("key_type", ["int64", "int32_dictionary_codes"])
def test_medium_cardinality_group_by_partitions_rows_then_accumulates_in_lds(key_type):
    """GROUP BY over tens of thousands of groups whose key and arguments are columns as they stand: the rows are moved into <= 64
    key-range partitions first and accumulated in LDS per partition (one global atomic per value and workgroup instead of one per row
    and aggregate).  Same groups in first-seen order, same SUM / COUNT / MIN / MAX / AVG as the specialised kernel with its global
    atomics (DFGPU_AGG_PARTITIONED=0 is read once per process, so the comparison is against values computed on the host)."""
    from decimal import Decimal

    from datafusion_amd import ops
    from datafusion_amd.expr import col
    from datafusion_amd.table import DeviceTable
    rng = np.random.default_rng(123)
    n, distinct = 9_000_000, 50_000
    codes = rng.integers(0, distinct, n)
    i64 = rng.integers(-10**9, 10**9, n)
    i32 = rng.integers(-1000, 1000, n).astype(np.int32)
    f64 = rng.random(n) * 100.0
    dec = rng.integers(-10**8, 10**8, n)
    cols = {"v": pa.array(i64), "w": pa.array(i32), "f": pa.array(f64), "d": pa.array([Decimal(int(x)).scaleb(-2) for x in dec[:0]], pa.decimal128(15, 2))}
    # Decimal128(15, 2) column from unscaled int64 values without 9 M Python objects: low word = value, high word = sign
    raw = np.empty((n, 2), dtype=np.int64)
    raw[:, 0] = dec
    raw[:, 1] = dec >> 63
    cols["d"] = pa.Array.from_buffers(pa.decimal128(15, 2), n, [None, pa.py_buffer(raw.tobytes())])
    if key_type == "int64":
        keys = codes * 3 + 1000
        cols["k"] = pa.array(keys)
        label = lambda c: c * 3 + 1000
    else:
        names = pa.array([f"n{i:06d}" for i in range(distinct)])
        cols["k"] = pa.DictionaryArray.from_arrays(pa.array(codes.astype(np.int32)), names)
        label = lambda c: f"n{c:06d}"
    t = DeviceTable.from_arrow(pa.table(cols))
    ops.profile_enable(True)
    ops.profile_reset()
    got = ops.aggregate(t, [(col("k"), "k")], [("sum", col("v"), "sv"), ("count", None, "n"), ("count", col("w"), "nw"), ("min", col("w"), "lo"), ("max", col("v"), "hi"),
                                              ("sum", col("d"), "sd"), ("avg", col("d"), "ad"), ("sum", col("f"), "sf"),
                                              ("sum", col("v") + col("v"), "s2"), ("avg", col("w"), "aw")], "Single").to_arrow()      # an expression; AVG(Int32) sums doubles
    stats = ops.profile_stats()
    ops.profile_enable(False)
    assert "agg_dense_accumulate_partitioned" in stats and "agg_dense_accumulate" not in stats, sorted(stats)
    first = np.full(distinct, n, dtype=np.int64)
    np.minimum.at(first, codes, np.arange(n))
    present = np.nonzero(first < n)[0]
    order = present[np.argsort(first[present], kind="stable")]
    assert got.column("k").to_pylist() == [label(int(c)) for c in order]
    sv = np.zeros(distinct, dtype=np.int64); np.add.at(sv, codes, i64)
    cnt = np.bincount(codes, minlength=distinct)
    lo = np.full(distinct, 2**31, dtype=np.int64); np.minimum.at(lo, codes, i32)
    hi = np.full(distinct, -2**62, dtype=np.int64); np.maximum.at(hi, codes, i64)
    sd = np.zeros(distinct, dtype=np.int64); np.add.at(sd, codes, dec)
    sf = np.bincount(codes, weights=f64, minlength=distinct)
    assert got.column("sv").to_pylist() == sv[order].tolist()
    assert got.column("n").to_pylist() == cnt[order].tolist() == got.column("nw").to_pylist()
    assert got.column("lo").to_pylist() == lo[order].tolist() and got.column("hi").to_pylist() == hi[order].tolist()
    assert [int(x.scaleb(2)) for x in got.column("sd").to_pylist()] == sd[order].tolist()
    # AVG(Decimal128(15, 2)) -> Decimal128(19, 6) (DecimalAverager::avg, functions-aggregate-common/src/utils.rs): the rule is the
    # ORACLE's, pinned by the reference's avg_cases — its Final mode turns the host's per-group (count, sum) state into the averages
    from oracle import oracle
    praw = np.empty((len(order), 2), dtype=np.int64)
    praw[:, 0] = sd[order]
    praw[:, 1] = sd[order] >> 63
    state = pa.table({"k": pa.array(np.arange(len(order))), "ad[count]": pa.array(cnt[order].astype(np.uint64)),
                      "ad[sum]": pa.Array.from_buffers(pa.decimal128(25, 2), len(order), [None, pa.py_buffer(praw.tobytes())])})
    want = oracle.aggregate(state, [(to_oracle_expr(col("k")), "k")], [("avg", to_oracle_expr(col("k")), "ad")], "Final", return_types={"ad": pa.decimal128(19, 6)})
    assert got.column("ad").to_pylist() == want.column("ad").to_pylist()
    assert np.allclose(got.column("sf").to_numpy(), sf[order], rtol=1e-9)
    assert got.column("s2").to_pylist() == (2 * sv[order]).tolist()
    sw = np.zeros(distinct, dtype=np.int64); np.add.at(sw, codes, i32)
    assert np.allclose(got.column("aw").to_numpy(), sw[order] / cnt[order], rtol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("move", ["grouped", "two_level"])
def test_medium_cardinality_group_by_over_a_wide_key_range_moves_the_rows_twice(monkeypatch, move):
    """a key range too wide for 64 LDS-sized windows: up to 2048 windows are reached by ONE move of the rows (grouped.hip's pass,
    round 4), up to 4096 by two 64-way moves (low 6 bits of the window number, then the high 6; DFGPU_AGG_GROUPED_MOVE=0 forces them
    here) before the LDS accumulation — 600 K groups over a range of 1.8 M values here, COUNT(*) / SUM / MIN in first-seen order"""
    from datafusion_amd import ops
    from datafusion_amd.expr import col
    from datafusion_amd.table import DeviceTable
    ops.set_options(agg__partitioned_min_rows="1000000")
    if move == "two_level":
        ops.set_options(agg__grouped_move="0")
    rng = np.random.default_rng(9)
    n, distinct = 6_000_000, 600_000
    codes = rng.integers(0, distinct, n)
    keys = codes * 3 + 5
    v = rng.integers(-10**6, 10**6, n)
    d = rng.integers(0, 3000, n).astype(np.int32)
    t = DeviceTable.from_arrow(pa.table({"k": pa.array(keys), "v": pa.array(v), "d": pa.array(d, pa.date32())}))
    ops.profile_enable(True)
    ops.profile_reset()
    got = ops.aggregate(t, [(col("k"), "k")], [("count", None, "n"), ("sum", col("v"), "sv"), ("min", col("d"), "first_day")], "Single").to_arrow()
    stats = ops.profile_stats()
    ops.profile_enable(False)
    assert "agg_dense_accumulate_partitioned" in stats and "agg_dense_accumulate" not in stats, sorted(stats)
    if move == "grouped":
        assert stats["agg_group_rows"]["calls"] == 1 and "partition_scatter" not in stats, sorted(stats)
    else:
        assert stats["partition_scatter"]["calls"] == 2 and "agg_group_rows" not in stats, sorted(stats)
    first = np.full(distinct, n, dtype=np.int64)
    np.minimum.at(first, codes, np.arange(n))
    present = np.nonzero(first < n)[0]
    order = present[np.argsort(first[present], kind="stable")]
    sv = np.zeros(distinct, dtype=np.int64); np.add.at(sv, codes, v)
    lo = np.full(distinct, 10**6, dtype=np.int64); np.minimum.at(lo, codes, d)
    assert got.column("k").to_pylist() == (order * 3 + 5).tolist()
    assert got.column("n").to_pylist() == np.bincount(codes, minlength=distinct)[order].tolist()
    assert got.column("sv").to_pylist() == sv[order].tolist()
    assert got.column("first_day").cast(pa.int32()).to_pylist() == lo[order].tolist()


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["gather", "renumber_forced", "skewed_windows"])
def test_partitioned_group_by_emits_groups_in_first_seen_order_by_either_form(case):
    """the two emits of the partitioned dense-key node: when every key window had ONE workgroup, the first rows are marked over the
    input rows by the accumulation itself and the groups are gathered in first-seen order by one pass over the marks (round 6); when
    a window's rows were split over several workgroups (skew: half of the rows in the first two thousand keys) or agg.gather_emit=0,
    per-value first rows are merged and the groups renumbered (rounds 3-5).  Same groups, order and values either way."""
    from datafusion_amd import ops
    from datafusion_amd.expr import col
    from datafusion_amd.table import DeviceTable
    rng = np.random.default_rng(61)
    n, distinct = 6_000_000, 300_000
    codes = rng.integers(0, distinct, n)
    if case == "skewed_windows":
        codes[: n // 2] = rng.integers(0, 2000, n // 2)
        rng.shuffle(codes)
    v = rng.integers(-10**6, 10**6, n)
    d = rng.integers(0, 3000, n).astype(np.int32)
    t = DeviceTable.from_arrow(pa.table({"k": pa.array(codes * 3 + 5), "v": pa.array(v), "d": pa.array(d, pa.date32())}))
    try:
        ops.set_options(agg__partitioned_min_rows="1000000")
        if case == "renumber_forced":
            ops.set_options(agg__gather_emit="0")
        ops.profile_enable(True)
        ops.profile_reset()
        got = ops.aggregate(t, [(col("k"), "k")], [("count", None, "n"), ("sum", col("v"), "sv"), ("min", col("d"), "first_day"), ("avg", col("v"), "av")], "Single").to_arrow()
        stats = ops.profile_stats()
        ops.profile_enable(False)
    finally:
        ops.reset_options()
    assert "agg_dense_accumulate_partitioned" in stats and "agg_dense_accumulate" not in stats, sorted(stats)
    assert ("agg_dense_gather_emit" in stats) == (case == "gather") and ("agg_dense_emit" in stats) == (case != "gather"), sorted(stats)
    first = np.full(distinct, n, dtype=np.int64)
    np.minimum.at(first, codes, np.arange(n))
    present = np.nonzero(first < n)[0]
    order = present[np.argsort(first[present], kind="stable")]
    cnt = np.bincount(codes, minlength=distinct)
    sv = np.zeros(distinct, dtype=np.int64); np.add.at(sv, codes, v)
    lo = np.full(distinct, 10**6, dtype=np.int64); np.minimum.at(lo, codes, d)
    assert got.column("k").to_pylist() == (order * 3 + 5).tolist()
    assert got.column("n").to_pylist() == cnt[order].tolist()
    assert got.column("sv").to_pylist() == sv[order].tolist()
    assert got.column("first_day").cast(pa.int32()).to_pylist() == lo[order].tolist()
    assert np.allclose(got.column("av").to_numpy(), sv[order] / cnt[order], rtol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["all_rows", "under_a_filter", "int64_argument"])
@pytest.mark.parametrize("form", ["records", "columns"])
def test_grouped_move_of_narrow_rows_as_records_or_columns(form, case):
    """Q13's shape — COUNT(*), MIN(Date32) by an integer key over a range of hundreds of thousands of values: the key (as a 32-bit offset),
    the 4-byte arguments and the row number are moved into the key windows as ONE record per row (grouped.hip's record form, round 6:
    12 bytes under the filter, 16 with a second argument; an 8-byte argument — MAX(Int64), SUM(Int64) — takes two of a record's four
    words), or column by column (group.records=0); with a fused FilterExec only the passing rows move.  Same groups in first-seen order,
    same counts, minima, maxima and sums."""
    from datafusion_amd import ops
    from datafusion_amd.expr import col, lit
    from datafusion_amd.table import DeviceTable
    rng = np.random.default_rng(71)
    filtered = case == "under_a_filter"
    n, distinct = 5_000_000, 400_000
    codes = rng.integers(0, distinct, n)
    d = rng.integers(-500, 3000, n).astype(np.int32)
    w = rng.integers(0, 100, n).astype(np.int32)
    v = rng.integers(-2**40, 2**40, n)
    t = DeviceTable.from_arrow(pa.table({"k": pa.array(codes * 2 - 300_000), "d": pa.array(d, pa.date32()), "w": pa.array(w), "v": pa.array(v)}))
    if case == "int64_argument":
        aggs = [("count", None, "n"), ("max", col("v"), "mv"), ("sum", col("v"), "sv")]      # (MAX and SUM read the same moved column)
    else:
        aggs = [("count", None, "n"), ("min", col("d"), "first_day")] + ([] if filtered else [("max", col("w"), "mw")])
    try:
        ops.set_options(agg__partitioned_min_rows="1000000")
        if form == "columns":
            ops.set_options(group__records="0")
        ops.profile_enable(True)
        ops.profile_reset()
        got = ops.aggregate(t, [(col("k"), "k")], aggs, "Single", predicate=(col("w") < lit(70, pa.int32())) if filtered else None).to_arrow()
        stats = ops.profile_stats()
        ops.profile_enable(False)
    finally:
        ops.reset_options()
    assert "agg_dense_accumulate_partitioned" in stats and stats["agg_group_rows"]["calls"] == 1 and "agg_dense_gather_emit" in stats, sorted(stats)
    keep = (w < 70) if filtered else np.ones(n, dtype=bool)
    kc, kd = codes[keep], d[keep]
    first = np.full(distinct, n, dtype=np.int64)
    np.minimum.at(first, kc, np.nonzero(keep)[0])
    present = np.nonzero(first < n)[0]
    order = present[np.argsort(first[present], kind="stable")]
    assert got.column("k").to_pylist() == (order * 2 - 300_000).tolist()
    assert got.column("n").to_pylist() == np.bincount(kc, minlength=distinct)[order].tolist()
    if case == "int64_argument":
        mv = np.full(distinct, -2**62, dtype=np.int64); np.maximum.at(mv, kc, v[keep])
        sv = np.zeros(distinct, dtype=np.int64); np.add.at(sv, kc, v[keep])
        assert got.column("mv").to_pylist() == mv[order].tolist()
        assert got.column("sv").to_pylist() == sv[order].tolist()
        return
    lo = np.full(distinct, 10**6, dtype=np.int64); np.minimum.at(lo, kc, kd)
    assert got.column("first_day").cast(pa.int32()).to_pylist() == lo[order].tolist()
    if not filtered:
        mw = np.full(distinct, -1, dtype=np.int64); np.maximum.at(mw, kc, w[keep])
        assert got.column("mw").to_pylist() == mw[order].tolist()


@pytest.mark.gpu
def test_final_merge_of_many_partial_states_moves_rows_by_group_number():
    """Final over millions of partial-state rows with a two-column key (hash-interned groups): every row's group number is looked up
    once, the rows are moved into LDS-sized windows of group numbers, accumulated there and merged per group — SUM / AVG / COUNT / MIN
    states of 40 K groups come out as the sums / minima of the partial rows"""
    from datafusion_amd import ops
    from datafusion_amd.expr import col
    rng = np.random.default_rng(31)
    n, g1, g2 = 9_000_000, 200, 200
    k1 = rng.integers(0, g1, n)
    k2 = rng.integers(0, g2, n).astype(np.int32)
    s = rng.integers(-10**6, 10**6, n)                 # SUM(int64) state
    c = rng.integers(0, 50, n).astype(np.uint64)       # COUNT state
    mn = rng.integers(-10**9, 10**9, n)                # MIN(int64) state
    ac = rng.integers(1, 9, n).astype(np.uint64)       # AVG(float64) state: count, sum
    asum = rng.random(n) * 10.0
    t = pa.table({"k1": pa.array(k1), "k2": pa.array(k2), "s[sum]": pa.array(s), "c[count]": pa.array(c.astype(np.int64)), "m[value]": pa.array(mn),
                  "a[count]": pa.array(ac), "a[sum]": pa.array(asum)})
    gb = [(col("k1"), "k1"), (col("k2"), "k2")]
    aggs = [("sum", col("v"), "s"), ("count", col("v"), "c"), ("min", col("v"), "m"), ("avg", col("f"), "a")]
    ops.profile_enable(True)
    ops.profile_reset()
    got = gpu_agg(t, gb, aggs, "Final")
    stats = ops.profile_stats()
    ops.profile_enable(False)
    assert "agg_dense_accumulate_partitioned" in stats and "agg_row_gids" in stats and "agg_accumulate_global" not in stats, sorted(stats)
    gid = k1 * g2 + k2
    ss = np.zeros(g1 * g2, dtype=np.int64); np.add.at(ss, gid, s)
    cc = np.bincount(gid, weights=c.astype(np.float64), minlength=g1 * g2).astype(np.int64)
    mm = np.full(g1 * g2, 2**62, dtype=np.int64); np.minimum.at(mm, gid, mn)
    an = np.bincount(gid, weights=ac.astype(np.float64), minlength=g1 * g2)
    asm = np.bincount(gid, weights=asum, minlength=g1 * g2)
    rows = {(a, b): (x, y, z, w) for a, b, x, y, z, w in zip(got.column("k1").to_pylist(), got.column("k2").to_pylist(), got.column("s").to_pylist(),
                                                        got.column("c").to_pylist(), got.column("m").to_pylist(), got.column("a").to_pylist())}
    assert len(rows) == got.num_rows == int((np.bincount(gid, minlength=g1 * g2) > 0).sum())
    for (a, b), (x, y, z, w) in list(rows.items())[::37]:
        g = a * g2 + b
        assert (x, y, z) == (int(ss[g]), int(cc[g]), int(mm[g]))
        assert abs(w - asm[g] / an[g]) <= 1e-9 * abs(asm[g] / an[g])


@pytest.mark.gpu
def test_medium_cardinality_group_by_under_a_fused_filter_moves_only_the_rows_that_pass():
    """a FilterExec fused below the aggregate (predicate with NULLs in its column): the rows it drops are not moved at all — the
    partitions hold the passing rows, the groups are those with a passing row, in the order their first passing rows come"""
    from datafusion_amd import ops
    from datafusion_amd.expr import col, lit
    from datafusion_amd.table import DeviceTable
    rng = np.random.default_rng(55)
    n, distinct = 9_000_000, 30_000
    codes = rng.integers(0, distinct, n)
    v = rng.integers(-10**6, 10**6, n)
    w = rng.integers(-100, 100, n).astype(np.int32)
    wnull = rng.random(n) < 0.1
    t = DeviceTable.from_arrow(pa.table({"k": pa.array(codes * 2 + 1), "v": pa.array(v), "w": pa.array(w, mask=wnull), "x": pa.array(rng.integers(0, 9, n))}))
    ops.profile_enable(True)
    ops.profile_reset()
    got = ops.aggregate(t, [(col("k"), "k")], [("sum", col("v"), "sv"), ("count", None, "n"), ("max", col("x"), "mx")], "Single", predicate=col("w") > lit(10, pa.int32())).to_arrow()
    stats = ops.profile_stats()
    ops.profile_enable(False)
    assert "agg_dense_accumulate_partitioned" in stats and "agg_dense_accumulate" not in stats, sorted(stats)
    keep = (w > 10) & ~wnull
    kc, kv, kx = codes[keep], v[keep], t.to_arrow().column("x").to_numpy()[keep]
    first = np.full(distinct, n, dtype=np.int64)
    np.minimum.at(first, kc, np.nonzero(keep)[0])
    present = np.nonzero(first < n)[0]
    order = present[np.argsort(first[present], kind="stable")]
    sv = np.zeros(distinct, dtype=np.int64); np.add.at(sv, kc, kv)
    mx = np.full(distinct, -1, dtype=np.int64); np.maximum.at(mx, kc, kx)
    assert got.column("k").to_pylist() == (order * 2 + 1).tolist()
    assert got.column("sv").to_pylist() == sv[order].tolist()
    assert got.column("n").to_pylist() == np.bincount(kc, minlength=distinct)[order].tolist()
    assert got.column("mx").to_pylist() == mx[order].tolist()


@pytest.mark.gpu
@pytest.mark.parametrize("key_type", ["int32", "int64"])
def test_one_grouped_move_under_a_fused_filter_with_a_decimal_sum(monkeypatch, key_type):
    """the single move into up to 2048 windows (grouped.hip) under a fused predicate: only passing rows move; Int32 and Int64 keys
    (the moved keys are widened), a 128-bit SUM(Decimal128) beside COUNT(*) and MAX — groups in the order of their first passing row"""
    from datafusion_amd import ops
    from datafusion_amd.expr import col, lit
    from datafusion_amd.table import DeviceTable
    ops.set_options(agg__partitioned_min_rows="1000000")
    rng = np.random.default_rng(77)
    n, distinct = 5_000_000, 400_000
    codes = rng.integers(0, distinct, n)
    keys = (codes * 5 - 700_000).astype(np.int32 if key_type == "int32" else np.int64)     # negative keys too: range 2 M values
    cents = rng.integers(-10**9, 10**9, n)
    x = rng.integers(0, 1000, n).astype(np.int32)
    w = rng.integers(0, 100, n).astype(np.int32)
    raw = np.empty((n, 2), dtype=np.int64)                                                  # Decimal128(15, 2) from its two words
    raw[:, 0] = cents
    raw[:, 1] = cents >> 63
    dcol = pa.Array.from_buffers(pa.decimal128(15, 2), n, [None, pa.py_buffer(raw.tobytes())])
    t = DeviceTable.from_arrow(pa.table({"k": pa.array(keys), "p": dcol, "x": pa.array(x), "w": pa.array(w)}))
    ops.profile_enable(True)
    ops.profile_reset()
    got = ops.aggregate(t, [(col("k"), "k")], [("sum", col("p"), "sp"), ("count", None, "n"), ("max", col("x"), "mx")], "Single", predicate=col("w") < lit(60, pa.int32())).to_arrow()
    stats = ops.profile_stats()
    ops.profile_enable(False)
    assert "agg_dense_accumulate_partitioned" in stats and stats["agg_group_rows"]["calls"] == 1 and "partition_scatter" not in stats, sorted(stats)
    keep = w < 60
    kc = codes[keep]
    first = np.full(distinct, n, dtype=np.int64)
    np.minimum.at(first, kc, np.nonzero(keep)[0])
    present = np.nonzero(first < n)[0]
    order = present[np.argsort(first[present], kind="stable")]
    sp = np.zeros(distinct, dtype=np.int64); np.add.at(sp, kc, cents[keep])
    mx = np.full(distinct, -1, dtype=np.int64); np.maximum.at(mx, kc, x[keep])
    assert got.column("k").to_pylist() == (order * 5 - 700_000).tolist()
    assert [int(v.scaleb(2)) for v in got.column("sp").to_pylist()] == sp[order].tolist()
    assert got.column("n").to_pylist() == np.bincount(kc, minlength=distinct)[order].tolist()
    assert got.column("mx").to_pylist() == mx[order].tolist()


@pytest.mark.gpu
@pytest.mark.parametrize("filtered", [False, True], ids=["no_predicate", "fused_filter"])
def test_two_column_key_medium_cardinality_moves_rows_by_group_number(filtered):
    """GROUP BY (k1, k2) — hash-interned groups, 40 K of them over 9 M rows — in Single mode: after the keys are interned, every row's
    group number is looked up once, the rows are moved by group number into LDS-sized windows and accumulated there; under a fused
    FilterExec only the passing rows are interned, looked up and moved.  Against sums computed on the host."""
    from datafusion_amd import ops
    from datafusion_amd.expr import col, lit
    from datafusion_amd.table import DeviceTable
    rng = np.random.default_rng(77 + filtered)
    n, g1, g2 = 9_000_000, 200, 200
    k1 = rng.integers(0, g1, n)
    k2 = rng.integers(0, g2, n).astype(np.int32)
    v = rng.integers(-10**6, 10**6, n)
    w = rng.integers(-100, 100, n).astype(np.int32)
    t = DeviceTable.from_arrow(pa.table({"k1": pa.array(k1), "k2": pa.array(k2), "v": pa.array(v), "w": pa.array(w)}))
    aggs = [("sum", col("v"), "sv"), ("count", None, "n"), ("avg", col("w"), "aw"), ("min", col("v"), "lo")]
    if not filtered:
        aggs.append(("sum", col("v") + col("v"), "s2"))       # an expression: evaluated column-at-a-time first
    pred = (col("w") >= lit(0, pa.int32())) if filtered else None
    ops.profile_enable(True)
    ops.profile_reset()
    got = ops.aggregate(t, [(col("k1"), "k1"), (col("k2"), "k2")], aggs, "Single", predicate=pred).to_arrow()
    stats = ops.profile_stats()
    ops.profile_enable(False)
    assert "agg_dense_accumulate_partitioned" in stats and "agg_row_gids" in stats and "agg_fused_global" not in stats, sorted(stats)
    keep = (w >= 0) if filtered else np.ones(n, dtype=bool)
    gid = (k1 * g2 + k2)[keep]
    first = np.full(g1 * g2, n, dtype=np.int64)
    np.minimum.at(first, gid, np.nonzero(keep)[0])
    present = np.nonzero(first < n)[0]
    order = present[np.argsort(first[present], kind="stable")]                      # first-seen order, like the reference
    assert got.column("k1").to_pylist() == (order // g2).tolist() and got.column("k2").to_pylist() == (order % g2).tolist()
    sv = np.zeros(g1 * g2, dtype=np.int64); np.add.at(sv, gid, v[keep])
    cnt = np.bincount(gid, minlength=g1 * g2)
    sw = np.zeros(g1 * g2, dtype=np.int64); np.add.at(sw, gid, w[keep])
    lo = np.full(g1 * g2, 2**62, dtype=np.int64); np.minimum.at(lo, gid, v[keep])
    assert got.column("sv").to_pylist() == sv[order].tolist() and got.column("n").to_pylist() == cnt[order].tolist()
    assert got.column("lo").to_pylist() == lo[order].tolist()
    assert np.allclose(got.column("aw").to_numpy(), sw[order] / cnt[order], rtol=1e-12)
    if not filtered:
        assert got.column("s2").to_pylist() == (2 * sv[order]).tolist()


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["u8_u8_date32", "i32_i32_with_negatives"])
@pytest.mark.parametrize("filtered", [False, True], ids=["no_predicate", "fused_filter"])
def test_narrow_key_columns_are_interned_through_one_packed_word(shape, filtered):
    """key columns without NULLs that fit 64 bits together are interned through their packed form, which the table keeps in its slots
    (one hash and one comparison per row, no trip to the representative row's columns) — the groups, their first-seen order and the key
    VALUES that come out (taken from the original columns, negative ones included) are those of the column-by-column path; four updates, so the later ones intern against
    existing groups, and the last (small) one goes column by column against groups found through packed keys"""
    from datafusion_amd import ops
    from datafusion_amd.expr import col, lit
    from datafusion_amd.table import DeviceTable
    rng = np.random.default_rng(5 + filtered)
    n = 3_430_000
    if shape == "u8_u8_date32":
        dom = [4, 50, 60]
        cols = {"a": pa.array(rng.integers(0, dom[0], n).astype(np.uint8) + 250), "b": pa.array((rng.integers(0, dom[1], n) * 5).astype(np.uint8)),
                "c": pa.array((rng.integers(0, dom[2], n) + 9000).astype(np.int32), pa.date32())}
        raw = [cols["a"].to_numpy().astype(np.int64), cols["b"].to_numpy().astype(np.int64), cols["c"].cast(pa.int32()).to_numpy().astype(np.int64)]
    else:
        dom = [300, 40]
        cols = {"a": pa.array((rng.integers(0, dom[0], n) - 150).astype(np.int32) * 7_000_000), "b": pa.array((rng.integers(0, dom[1], n) - 39).astype(np.int32))}
        raw = [cols["a"].to_numpy().astype(np.int64), cols["b"].to_numpy().astype(np.int64)]
    names = list(cols)
    v = rng.integers(-10**6, 10**6, n)
    w = rng.integers(-100, 100, n).astype(np.int32)
    table = pa.table({**cols, "v": pa.array(v), "w": pa.array(w)})
    gb = [(col(c), c) for c in names]
    aggs = [("sum", col("v"), "sv"), ("count", None, "cnt"), ("max", col("w"), "hi")]
    pred = (col("w") < lit(20, pa.int32())) if filtered else None
    a = ops.GroupedAggregate("Single", table.column_names, gb, aggs)
    ops.profile_enable(True)
    ops.profile_reset()
    for lo, hi in ((0, 1_300_000), (1_300_000, 2_500_000), (2_500_000, 3_400_000), (3_400_000, n)):      # the last one is below the threshold
        a.update(DeviceTable.from_arrow(table.slice(lo, hi - lo)), pred)                                # (existing groups + rows < 2^16): column by column
    got = a.emit().to_arrow()
    stats = ops.profile_stats()
    ops.profile_enable(False)
    assert stats["agg_intern_claim_keyed"]["calls"] == 3 and stats["agg_intern_claim"]["calls"] == 1, sorted(stats)
    keep = (w < 20) if filtered else np.ones(n, dtype=bool)
    _, dense = np.unique(np.stack(raw, axis=1), axis=0, return_inverse=True)
    dense = dense.reshape(-1)
    G = int(dense.max()) + 1
    gid = dense[keep]
    first = np.full(G, n, dtype=np.int64)
    np.minimum.at(first, gid, np.nonzero(keep)[0])
    present = np.nonzero(first < n)[0]
    order = present[np.argsort(first[present], kind="stable")]
    rows = first[order]
    assert got.num_rows == len(order)
    for c in names:
        assert got.column(c).to_pylist() == table.column(c).take(pa.array(rows)).to_pylist(), c
    sv = np.zeros(G, dtype=np.int64); np.add.at(sv, gid, v[keep])
    hi = np.full(G, -1000, dtype=np.int64); np.maximum.at(hi, gid, w[keep])
    assert got.column("sv").to_pylist() == sv[order].tolist()
    assert got.column("cnt").to_pylist() == np.bincount(gid, minlength=G)[order].tolist()
    assert got.column("hi").to_pylist() == hi[order].tolist()


@pytest.mark.gpu
@pytest.mark.parametrize("filtered", [False, True], ids=["no_predicate", "fused_filter"])
@pytest.mark.parametrize("keys", ["dense_int64", "two_columns"])
def test_a_few_thousand_groups_are_accumulated_in_lds_where_the_rows_lie(keys, filtered):
    """a key range (or a number of hash-interned groups) small enough for ONE workgroup's LDS: no row is moved — every workgroup
    accumulates a slice of the rows in place (the fused predicate's mask read row by row, NULL predicate values dropping the row)
    and the copies merge; groups in first-seen order, SUM / COUNT / MIN / AVG / SUM(Decimal128) as computed on the host"""
    from datafusion_amd import ops
    from datafusion_amd.expr import col, lit
    from datafusion_amd.table import DeviceTable
    rng = np.random.default_rng(1234 + filtered)
    n = 9_000_000
    if keys == "dense_int64":
        G = 5000
        gid_all = rng.integers(0, G, n)
        cols = {"k": pa.array(gid_all * 1 + 77)}
        gb = [(col("k"), "k")]
    else:
        g2 = 60
        G = 50 * g2
        k1 = rng.integers(0, 50, n)
        k2 = rng.integers(0, g2, n).astype(np.int32)
        gid_all = k1 * g2 + k2
        cols = {"k1": pa.array(k1 - 25), "k2": pa.array(k2)}
        gb = [(col("k1"), "k1"), (col("k2"), "k2")]
    v = rng.integers(-10**6, 10**6, n)
    w = rng.integers(-100, 100, n).astype(np.int32)
    wnull = rng.random(n) < 0.05
    dec = rng.integers(-10**8, 10**8, n)
    raw = np.empty((n, 2), dtype=np.int64)
    raw[:, 0] = dec
    raw[:, 1] = dec >> 63
    table = pa.table({**cols, "v": pa.array(v), "w": pa.array(w, mask=wnull if filtered else None),
                      "d": pa.Array.from_buffers(pa.decimal128(15, 2), n, [None, pa.py_buffer(raw.tobytes())])})
    aggs = [("sum", col("v"), "sv"), ("count", None, "cnt"), ("min", col("v"), "lo"), ("avg", col("v"), "av")]
    if keys == "two_columns":
        aggs.append(("sum", col("d"), "sd"))           # a 128-bit sum: two cell words, its own launch beside the others
    pred = (col("w") >= lit(-20, pa.int32())) if filtered else None
    t = DeviceTable.from_arrow(table)
    ops.profile_enable(True)
    ops.profile_reset()
    got = ops.aggregate(t, gb, aggs, "Single", predicate=pred).to_arrow()
    stats = ops.profile_stats()
    ops.profile_enable(False)
    assert "agg_dense_accumulate_partitioned" in stats and "partition_scatter" not in stats and "agg_fused_global" not in stats, sorted(stats)
    if keys == "two_columns":   # (k1: 50 values) x (k2: 60 values): a direct table of 3000 slots (round 4), no hash table
        assert "agg_intern_claim_direct" in stats and "agg_intern_claim_keyed" not in stats, sorted(stats)
    keep = ((w >= -20) & ~wnull) if filtered else np.ones(n, dtype=bool)
    gid = gid_all[keep]
    first = np.full(G, n, dtype=np.int64)
    np.minimum.at(first, gid, np.nonzero(keep)[0])
    present = np.nonzero(first < n)[0]
    order = present[np.argsort(first[present], kind="stable")]
    for c in cols:
        assert got.column(c).to_pylist() == table.column(c).take(pa.array(first[order])).to_pylist(), c
    sv = np.zeros(G, dtype=np.int64); np.add.at(sv, gid, v[keep])
    cnt = np.bincount(gid, minlength=G)
    lo = np.full(G, 2**62, dtype=np.int64); np.minimum.at(lo, gid, v[keep])
    assert got.column("sv").to_pylist() == sv[order].tolist() and got.column("cnt").to_pylist() == cnt[order].tolist()
    assert got.column("lo").to_pylist() == lo[order].tolist()
    assert np.allclose(got.column("av").to_numpy(), sv[order] / cnt[order], rtol=1e-12)
    if keys == "two_columns":
        sd = np.zeros(G, dtype=np.int64); np.add.at(sd, gid, dec[keep])
        assert [int(x.scaleb(2)) for x in got.column("sd").to_pylist()] == sd[order].tolist()


@pytest.mark.gpu
@pytest.mark.parametrize("table", ["direct", "keyed_by_knob"])
@pytest.mark.parametrize("filtered", [False, True], ids=["no_predicate", "fused_filter"])
def test_key_columns_with_small_value_ranges_take_a_direct_table(monkeypatch, table, filtered):
    """round 4: key columns whose value ranges multiply to a few thousand slots — two UInt8 flags holding a few letters each (numbered by
    the values that occur: 'A', 'N', 'R' -> 0, 1, 2), a Date32 over ~1500 days, an Int64 with negative values — are interned without a
    hash table: a row's slot is its mixed-radix number, the smallest row per slot is kept in LDS.  Groups, first-seen order, key values
    and totals equal the host's and those of the keyed table (DFGPU_AGG_DIRECT_TABLE=0); a second update interns against the groups
    of the first through the hash table again"""
    from datafusion_amd import ops
    from datafusion_amd.expr import col, lit
    from datafusion_amd.table import DeviceTable
    if table == "keyed_by_knob":
        ops.set_options(agg__direct_table="0")
    rng = np.random.default_rng(41 + filtered)
    n, n2 = 4_600_000, 300_000
    flag = np.frombuffer(b"ANR", dtype=np.uint8)[rng.integers(0, 3, n + n2)]
    status = np.frombuffer(b"FO", dtype=np.uint8)[rng.integers(0, 2, n + n2)]
    day = (np.sort(rng.integers(0, 1500, n + n2)) + rng.integers(0, 60, n + n2) + 8000).astype(np.int32)     # clustered like l_shipdate
    small = rng.integers(-2, 1, n + n2)                                                                       # Int64: -2, -1, 0
    v = rng.integers(-10**6, 10**6, n + n2)
    w = rng.integers(-100, 100, n + n2).astype(np.int32)
    cents = rng.integers(-10**8, 10**8, n + n2)
    raw = np.empty((n + n2, 2), dtype=np.int64)
    raw[:, 0] = cents
    raw[:, 1] = cents >> 63
    full = pa.table({"f": pa.array(flag), "s": pa.array(status), "d": pa.array(day, pa.date32()), "z": pa.array(small), "v": pa.array(v), "w": pa.array(w),
                     "p": pa.Array.from_buffers(pa.decimal128(15, 2), n + n2, [None, pa.py_buffer(raw.tobytes())])})
    gb = [(col(c), c) for c in ("f", "s", "d", "z")]
    aggs = [("sum", col("p"), "sp"), ("sum", col("v"), "sv"), ("count", None, "cnt"), ("max", col("w"), "hi")]
    pred = (col("w") < lit(30, pa.int32())) if filtered else None
    a = ops.GroupedAggregate("Single", full.column_names, gb, aggs)
    ops.profile_enable(True)
    ops.profile_reset()
    a.update(DeviceTable.from_arrow(full.slice(0, n)), pred)
    a.update(DeviceTable.from_arrow(full.slice(n, n2)), pred)
    got = a.emit().to_arrow()
    stats = ops.profile_stats()
    ops.profile_enable(False)
    if table == "direct":
        assert stats["agg_intern_claim_direct"]["calls"] == 1 and "column_u8_presence" in stats, sorted(stats)
    else:
        # (four key columns of 14 bytes do not fit the keyed table's one word: column by column)
        assert "agg_intern_claim_direct" not in stats and ("agg_intern_claim_keyed" in stats or "agg_intern_claim" in stats), sorted(stats)
    keep = (w < 30) if filtered else np.ones(n + n2, dtype=bool)
    key = ((flag.astype(np.int64) * 256 + status) * 100_000 + day) * 8 + (small + 2)
    uniq, inv = np.unique(key, return_inverse=True)
    inv = inv.reshape(-1)
    G = len(uniq)
    gid = inv[keep]
    first = np.full(G, n + n2, dtype=np.int64)
    np.minimum.at(first, gid, np.nonzero(keep)[0])
    present = np.nonzero(first < n + n2)[0]
    order = present[np.argsort(first[present], kind="stable")]
    rows = first[order]
    assert got.num_rows == len(order) > 10_000
    for c in ("f", "s", "d", "z"):
        assert got.column(c).to_pylist() == full.column(c).take(pa.array(rows)).to_pylist(), c
    sv = np.zeros(G, dtype=np.int64); np.add.at(sv, gid, v[keep])
    sp = np.zeros(G, dtype=np.int64); np.add.at(sp, gid, cents[keep])
    hi = np.full(G, -1000, dtype=np.int64); np.maximum.at(hi, gid, w[keep])
    assert got.column("sv").to_pylist() == sv[order].tolist()
    assert [int(x.scaleb(2)) for x in got.column("sp").to_pylist()] == sp[order].tolist()
    assert got.column("cnt").to_pylist() == np.bincount(gid, minlength=G)[order].tolist()
    assert got.column("hi").to_pylist() == hi[order].tolist()


@pytest.mark.gpu
@pytest.mark.parametrize("streamed", [False, True], ids=["one_update", "three_updates"])
def test_the_packed_key_of_all_ones_has_a_slot_of_its_own(streamed):
    """(Int32 -1, Int32 -1) packs to the one 64-bit word an empty slot of the keyed table holds: its group lives in a slot past the
    table's end — it is found, counted once, and keeps its place in the first-seen order"""
    from datafusion_amd import ops
    from datafusion_amd.expr import col
    from datafusion_amd.table import DeviceTable
    rng = np.random.default_rng(2)
    n = 300_000
    a = rng.integers(-1, 2, n).astype(np.int32)
    b = (rng.integers(-1, 3, n) * np.where(rng.random(n) < 0.5, 1, 7)).astype(np.int32)
    a[:5] = [3, 3, -1, 3, -1]
    b[:5] = [3, 3, -1, 3, -1]               # (-1, -1) is the second group seen
    v = rng.integers(-1000, 1000, n)
    table = pa.table({"a": pa.array(a), "b": pa.array(b), "v": pa.array(v)})
    agg = ops.GroupedAggregate("Single", table.column_names, [(col("a"), "a"), (col("b"), "b")], [("sum", col("v"), "sv"), ("count", None, "cnt")])
    ops.profile_enable(True)
    ops.profile_reset()
    for lo, hi in (((0, 100_000), (100_000, 200_000), (200_000, n)) if streamed else ((0, n),)):
        agg.update(DeviceTable.from_arrow(table.slice(lo, hi - lo)))
    got = agg.emit().to_arrow()
    stats = ops.profile_stats()
    ops.profile_enable(False)
    assert "agg_intern_claim_keyed" in stats and "agg_intern_claim" not in stats, sorted(stats)
    want = {}
    for x, y, z in zip(a.tolist(), b.tolist(), v.tolist()):
        e = want.setdefault((x, y), [0, 0])
        e[0] += z
        e[1] += 1
    assert list(zip(got.column("a").to_pylist(), got.column("b").to_pylist())) == list(want)           # dict order = first-seen order
    assert list(want)[1] == (-1, -1)
    assert got.column("sv").to_pylist() == [e[0] for e in want.values()] and got.column("cnt").to_pylist() == [e[1] for e in want.values()]


@pytest.mark.gpu
def test_keyed_table_sized_from_a_misleading_sample_overflows_and_is_rebuilt_larger():
    """the first million rows carry 1000 distinct (a, b) keys — the table is sized for a few thousand groups — and the rows after them
    300 K more: the claim pass reports the table too full, it is rebuilt 16 x larger, and the groups, their first-seen order and
    their counts / sums are those computed on the host"""
    from datafusion_amd import ops
    from datafusion_amd.expr import col
    from datafusion_amd.table import DeviceTable
    rng = np.random.default_rng(99)
    n, head = 6_000_000, 2_000_000
    a = np.concatenate([rng.integers(0, 50, head), rng.integers(0, 600, n - head)]).astype(np.int32)
    b = np.concatenate([rng.integers(0, 20, head), rng.integers(0, 500, n - head)]).astype(np.int32) - 3
    v = rng.integers(-1000, 1000, n)
    t = DeviceTable.from_arrow(pa.table({"a": pa.array(a), "b": pa.array(b), "v": pa.array(v)}))
    ops.profile_enable(True)
    ops.profile_reset()
    got = ops.aggregate(t, [(col("a"), "a"), (col("b"), "b")], [("sum", col("v"), "sv"), ("count", None, "cnt")], "Single").to_arrow()
    stats = ops.profile_stats()
    ops.profile_enable(False)
    assert stats["agg_intern_claim_keyed"]["calls"] >= 2, sorted(stats)
    key = a.astype(np.int64) * 1000 + (b + 3)
    uniq, first, inv = np.unique(key, return_index=True, return_inverse=True)
    order = np.argsort(first, kind="stable")
    assert got.num_rows == len(uniq) > 250_000
    assert got.column("a").to_pylist() == a[first[order]].tolist() and got.column("b").to_pylist() == b[first[order]].tolist()
    sv = np.zeros(len(uniq), dtype=np.int64); np.add.at(sv, inv.reshape(-1), v)
    assert got.column("sv").to_pylist() == sv[order].tolist()
    assert got.column("cnt").to_pylist() == np.bincount(inv.reshape(-1), minlength=len(uniq))[order].tolist()


@pytest.mark.gpu
@pytest.mark.parametrize("keys", ["one_nullable_int64", "two_columns_one_nullable", "nullable_only_in_a_later_batch"])
def test_streamed_updates_over_nullable_group_keys(keys):
    """round 4: group keys with NULLs across SEVERAL updates (the reference interns NULL as a key value of its own and keeps doing
    so batch after batch: group_values/multi_group_by/mod.rs:595-745): the groups that exist already and the new rows are
    concatenated WITH their validity — a side without a bitmap counts as all valid — so a NULL key of batch 3 finds the NULL group
    of batch 1.  Groups in first-seen order; checked against pyarrow's group_by (NULL keys form a group there too)"""
    from datafusion_amd import ops
    from datafusion_amd.expr import col
    from datafusion_amd.table import DeviceTable
    rng = np.random.default_rng(77)
    n = 240_000
    k1 = rng.integers(0, 3000, n) * 7 - 500
    k2 = rng.integers(0, 5, n).astype(np.int32)
    m1 = rng.random(n) < 0.03
    m2 = rng.random(n) < 0.10
    if keys == "nullable_only_in_a_later_batch":
        m1[:100_000] = False          # the first updates carry no validity bitmap at all
    v = rng.integers(-1000, 1000, n)
    cols = {"k1": pa.array(k1, mask=m1)}
    gb = [(col("k1"), "k1")]
    if keys == "two_columns_one_nullable":
        cols["k2"] = pa.array(k2, mask=m2)
        gb.append((col("k2"), "k2"))
    table = pa.table({**cols, "v": pa.array(v)})
    a = ops.GroupedAggregate("Single", table.column_names, gb, [("sum", col("v"), "s"), ("count", None, "c"), ("min", col("v"), "lo")])
    cuts = [0, 100_000, 100_007, 180_000, n]
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        part = table.slice(lo, hi - lo)
        if keys == "nullable_only_in_a_later_batch" and hi <= 100_000:
            part = pa.table({"k1": pa.array(k1[lo:hi]), "v": part.column("v")})      # no bitmap on this batch
        a.update(DeviceTable.from_arrow(part))
    got = a.emit().to_arrow()
    names = [nm for _, nm in gb]
    exp = table.group_by(names, use_threads=False).aggregate([("v", "sum"), ("v", "count"), ("v", "min")])
    assert got.num_rows == exp.num_rows
    key = lambda row: tuple((x is None, 0 if x is None else x) for x in row)
    g = sorted(zip(*[got.column(c).to_pylist() for c in names + ["s", "c", "lo"]]), key=key)
    e = sorted(zip(*[exp.column(c).to_pylist() for c in names + ["v_sum", "v_count", "v_min"]]), key=key)
    assert g == e
    # first-seen order: the k-th group's key is the k-th distinct key of the input
    seen, order = set(), []
    for row in zip(*[table.column(c).to_pylist() for c in names]):
        if row not in seen:
            seen.add(row)
            order.append(row)
    assert list(zip(*[got.column(c).to_pylist() for c in names])) == order
