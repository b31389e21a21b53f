"""One LARGE exact check (BASELINE config 3 at SF10: 15 M orders x 60 M lineitem rows): the join's output compared ROW BY ROW with a
host restatement, every column of every row, and TPC-H Q3 end to end against the oracle's operators over the same tables — the
size-independent properties of tests/test_gpu_fullsize.py (SF100) would not notice, say, an output row pairing the right key with
a neighbour's payload in a way that preserves column sums.

The inputs are generated on the device (tpch.hip, the generator the benchmarks use) and exported once; the expected join is
`take(orders, searchsorted(o_orderkey, l_orderkey))` next to the lineitem columns (numpy / Arrow on the host), cross-checked by the
oracle's partitioned hash join (pair count).  Probe-order flavours compare positionally; the unordered flavours as multisets of
whole rows (a 64-bit mix of every column of the row, sorted)."""
import numpy as np
import pyarrow as pa
import pytest

pytestmark = pytest.mark.gpu

SF = 10.0
BUILD_COLS, PROBE_COLS = ["o_orderdate", "o_shippriority"], ["l_orderkey", "l_extendedprice", "l_discount"]


@pytest.fixture(scope="module")
def sf10():
    from datafusion_amd import ops
    orders = ops.tpch_orders(SF).select(["o_orderkey", "o_custkey", "o_orderdate", "o_shippriority"])
    lineitem = ops.tpch_lineitem(SF).select(["l_orderkey", "l_extendedprice", "l_discount", "l_shipdate"])
    customer = ops.tpch_customer(SF)
    host = dict(orders=orders.to_arrow(), lineitem=lineitem.to_arrow(), customer=customer.to_arrow())
    ok, lk = host["orders"].column("o_orderkey").to_numpy(), host["lineitem"].column("l_orderkey").to_numpy()
    assert (np.diff(ok) > 0).all()
    pos = np.searchsorted(ok, lk)
    assert (ok[pos] == lk).all()                                           # every line has its order
    expected = pa.table({**{c: host["orders"].column(c).take(pa.array(pos)) for c in BUILD_COLS}, **{c: host["lineitem"].column(c) for c in PROBE_COLS}})
    yield dict(orders=orders, lineitem=lineitem, customer=customer, host=host, expected=expected)
    for t in (orders, lineitem, customer):
        t.free()


def _row_mix(table: pa.Table) -> np.ndarray:
    """one 64-bit value per row mixing every column of the row (Decimal128 through both halves)"""
    h = np.zeros(table.num_rows, np.uint64)
    for c in table.columns:
        a = c.combine_chunks()
        if pa.types.is_decimal128(a.type):
            words = np.frombuffer(a.buffers()[1], np.uint64, 2 * len(a)).reshape(-1, 2)
            parts = [words[:, 0], words[:, 1]]
        elif pa.types.is_date32(a.type):
            parts = [a.cast(pa.int32()).to_numpy().astype(np.uint64)]
        else:
            parts = [a.to_numpy().astype(np.int64).view(np.uint64)]
        for p in parts:
            h = (h ^ p) * np.uint64(0x9E3779B97F4A7C15)
            h ^= h >> np.uint64(29)
    return h


@pytest.mark.parametrize("probe_mode", [0, 1], ids=["placed_probe_order", "two_pass_probe_order"])
def test_sf10_join_equals_the_host_join_row_by_row(sf10, probe_mode):
    from datafusion_amd import ops
    from oracle import oracle
    ht = ops.JoinHashTable(sf10["orders"], ["o_orderkey"], probe_mode=probe_mode)
    out = ht.probe(sf10["lineitem"], ["l_orderkey"], "Inner", BUILD_COLS, PROBE_COLS)
    ht.free()
    got = out.to_arrow()
    out.free()
    exp = sf10["expected"]
    assert got.num_rows == exp.num_rows > 59_000_000
    for c in exp.column_names:                                             # probe order: row i of the output is probe row i
        assert got.column(c).combine_chunks().equals(exp.column(c).combine_chunks()), c
    pairs, _ = oracle.partitioned_inner_join_i64(sf10["host"]["orders"].column("o_orderkey").to_numpy(), sf10["host"]["lineitem"].column("l_orderkey").to_numpy(), 32)
    assert pairs == got.num_rows


@pytest.mark.parametrize("flavour", ["single_pass_unordered", "hash_map", "radix_lds"])
def test_sf10_join_unordered_flavours_hold_the_same_rows(sf10, flavour):
    from datafusion_amd import ops
    kw = {"single_pass_unordered": dict(probe_mode=3), "hash_map": dict(table_mode=1, probe_mode=3), "radix_lds": dict(table_mode=4, probe_mode=4)}[flavour]
    ht = ops.JoinHashTable(sf10["orders"], ["o_orderkey"], **kw)
    out = ht.probe(sf10["lineitem"], ["l_orderkey"], "Inner", BUILD_COLS, PROBE_COLS)
    ht.free()
    got = out.to_arrow().select(sf10["expected"].column_names)
    out.free()
    assert got.num_rows == sf10["expected"].num_rows
    assert np.array_equal(np.sort(_row_mix(got)), np.sort(_row_mix(sf10["expected"])))


@pytest.mark.parametrize("fused", [True, False])
def test_sf10_q3_equals_the_oracle(sf10, fused):
    """the reference's pinned Q3 plan over the same 15 M / 60 M-row tables: the oracle's operators on the host vs the device"""
    from datafusion_amd import queries
    from tests import tpch_plans as T
    from datafusion_amd.expr import lit
    from tests import plan_oracle
    if "q3" not in sf10:
        h = sf10["host"]
        sf10["q3"] = plan_oracle.collect(T.q3_plan(h["customer"], h["orders"], h["lineitem"], segment_literal=lit(queries.SEGMENT_BUILDING, pa.uint8())))
    got = queries.q3(sf10["customer"], sf10["orders"], sf10["lineitem"], fused=fused).to_arrow()
    assert got.to_pylist() == sf10["q3"].to_pylist()


# ------------------------------------------------------------------------------------------------------------------------------------
# BASELINE configs 2 and 4 at SF10, exact (round-2 verdict, weak 1): FilterExec positionally against the oracle's filter over all
# 60 M lineitem rows, Q1 digit for digit through all three evaluators of the fused node, and the Float64 variant of the money
# columns (where the order of the device's atomics really varies) within north_star's 1e-6 of the oracle's row-order sums.

Q3_PROJECTION = ["l_orderkey", "l_extendedprice", "l_discount"]


@pytest.fixture(scope="module")
def lineitem10():
    """the whole SF10 lineitem table (8 columns) on the device and, exported once, on the host"""
    from datafusion_amd import ops
    dev = ops.tpch_lineitem(SF)
    host = dev.to_arrow()
    assert host.num_rows > 59_000_000
    yield dict(dev=dev, host=host)
    dev.free()


def _config2_predicates():
    """the three selectivities of BASELINE config 2 / SURVEY §8(d): Q1's (98 %), Q3's (54 %) and a one-year band (15 %)"""
    import datetime

    from datafusion_amd.expr import col, lit
    d = lambda s: lit(datetime.date.fromisoformat(s), pa.date32())
    return {"q1_le_1998_09_02": col("l_shipdate") <= d("1998-09-02"), "q3_gt_1995_03_15": col("l_shipdate") > d("1995-03-15"),
            "year_1994": (col("l_shipdate") >= d("1994-01-01")).and_(col("l_shipdate") <= d("1994-12-31"))}


@pytest.mark.parametrize("projection", [None, Q3_PROJECTION], ids=["all_columns", "q3_projection"])
@pytest.mark.parametrize("which", ["q1_le_1998_09_02", "q3_gt_1995_03_15", "year_1994"])
def test_sf10_filter_equals_the_oracle_row_by_row(lineitem10, which, projection):
    """FilterExec (filter.rs:1339-1444) over 60 M rows: every output row at its position, every column (config 2)"""
    from datafusion_amd import ops
    from oracle import oracle
    from tests.util import to_oracle_expr
    pred = _config2_predicates()[which]
    out = ops.filter(lineitem10["dev"], pred, projection)
    got = out.to_arrow()
    out.free()
    exp = oracle.filter(lineitem10["host"], to_oracle_expr(pred), projection)
    assert got.schema.names == exp.schema.names
    assert 0 < got.num_rows == exp.num_rows < lineitem10["host"].num_rows
    for name in exp.column_names:
        assert got.schema.field(name).type == exp.schema.field(name).type, name
        assert got.column(name).combine_chunks().equals(exp.column(name).combine_chunks()), name


@pytest.fixture(params=["specialised", "interpreted", "column_at_a_time"])
def evaluator(request):
    """the three evaluators of the fused FilterExec + ProjectionExec + AggregateExec node (tests/test_gpu_fused.py `fusion`)"""
    import os

    from datafusion_amd import ops
    ops.set_fusion(request.param != "column_at_a_time")
    if request.param == "specialised":
        ops.set_options(jit="1", jit__strict="1")
    else:
        ops.set_options(jit="0")
    yield request.param
    ops.reset_options()
    ops.set_fusion(True)


def test_sf10_q1_equals_the_oracle_digit_for_digit(lineitem10, evaluator):
    """config 4's query at SF10 (aggregates/mod.rs:1167-1253): the pinned Q1 plan, Decimal128 sums and averages bit-exact"""
    from datafusion_amd import queries
    from tests.test_gpu_queries import oracle_q1
    if "q1" not in lineitem10:
        lineitem10["q1"] = oracle_q1(lineitem10["host"])
    out = queries.q1(lineitem10["dev"])
    got = out.to_arrow()
    out.free()
    exp = lineitem10["q1"]
    assert got.schema == exp.schema
    assert got.num_rows == exp.num_rows == 4
    assert got.to_pylist() == exp.to_pylist()
    out = queries.q1(lineitem10["dev"], fused=False)                        # operator by operator: FilterExec -> ProjectionExec -> AggregateExec
    assert out.to_arrow().to_pylist() == exp.to_pylist()
    out.free()


@pytest.fixture(scope="module")
def lineitem10_float():
    """SF10 lineitem with Float64 money columns + a ProjectionExec'd `bucket` column, on the device and on the host"""
    from datafusion_amd import ops
    from datafusion_amd.expr import col, lit
    raw = ops.tpch_lineitem(SF, float_money=True)
    dev = ops.project(raw, [(col(n), n) for n in raw.column_names] + [(col("l_orderkey") % lit(100_003, pa.int64()), "bucket")])
    raw.free()
    yield dict(dev=dev, host=dev.to_arrow(), expected={})
    dev.free()


@pytest.mark.parametrize("group_key", ["flags", "l_orderkey", "dense_bucket", "hashed"])
def test_sf10_float64_sums_and_averages_within_1e_6(lineitem10_float, evaluator, group_key):
    """Float64 money columns (dfgpu_tpch_lineitem(float_money=1)): SUM / AVG accumulate in whatever order the device's atomics land,
    the oracle in row order — north_star's tolerance is 1e-6 relative.  Four group shapes = the four accumulation paths: 4 groups
    (LDS cells, ~15 M addends per group), one run per order (runs node), a dense integer key in no order (rank interning, 100 003
    groups), two key columns (hash interning, global atomics)."""
    from datafusion_amd import ops
    from datafusion_amd.expr import col, lit
    from oracle import oracle
    from tests.util import to_oracle_expr
    one = lit(1.0, pa.float64())
    disc_price = col("l_extendedprice") * (one - col("l_discount"))
    aggs = [("sum", col("l_extendedprice"), "sum_base_price"), ("sum", disc_price, "sum_disc_price"), ("sum", disc_price * (one + col("l_tax")), "sum_charge"),
            ("avg", col("l_quantity"), "avg_qty"), ("avg", col("l_discount"), "avg_disc"), ("count", None, "n")]
    gb = {"flags": [(col("l_returnflag"), "l_returnflag"), (col("l_linestatus"), "l_linestatus")], "l_orderkey": [(col("l_orderkey"), "l_orderkey")],
          "dense_bucket": [(col("bucket"), "bucket")], "hashed": [(col("bucket"), "bucket"), (col("l_linestatus"), "l_linestatus")]}[group_key]
    out = ops.aggregate(lineitem10_float["dev"], gb, aggs, "Single")
    got = out.to_arrow()
    out.free()
    if group_key not in lineitem10_float["expected"]:
        lineitem10_float["expected"][group_key] = oracle.aggregate(
            lineitem10_float["host"], [(to_oracle_expr(e), n) for e, n in gb], [(f, None if e is None else to_oracle_expr(e), n) for f, e, n in aggs], "Single")
    exp = lineitem10_float["expected"][group_key]
    assert got.schema == exp.schema and got.num_rows == exp.num_rows
    for name in exp.column_names:                                           # first-seen group order on both sides
        g, e = got.column(name).to_numpy(), exp.column(name).to_numpy()
        if exp.schema.field(name).type == pa.float64():
            assert np.isfinite(g).all(), name
            rel = np.abs(g - e) / np.maximum(np.abs(e), 1e-300)
            assert rel.max() <= 1e-6, (name, float(rel.max()))
        else:
            assert np.array_equal(g, e), name
