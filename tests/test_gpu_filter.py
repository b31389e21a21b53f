"""K8/K9 parity: FilterExec / ProjectionExec on the GPU vs the CPU oracle (bit-exact)."""
import datetime

import numpy as np
import pyarrow as pa
import pytest

from tests.util import assert_tables_equal, random_table, to_oracle_expr

pytestmark = pytest.mark.gpu


def run_filter(table, pred, projection=None):
    from datafusion_amd import ops
    from datafusion_amd.table import DeviceTable
    from oracle import oracle
    dev = DeviceTable.from_arrow(table)
    got = ops.filter(dev, pred, projection).to_arrow()
    exp = oracle.filter(table, to_oracle_expr(pred), projection)
    # FilterExec preserves input order (maintains_input_order) -> ordered comparison
    assert_tables_equal(got, exp, ordered=True)
    return got


SPEC = {"k": (pa.int64(), 0, 1000), "d": (pa.date32(), 8000, 10500), "p": (pa.decimal128(15, 2), -10**6, 10**6),
        "q": (pa.int32(), -50, 50), "f": (pa.float64(), -1000, 1000), "c": (pa.uint8(), 0, 5)}


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 1000, 100_003])
def test_filter_date_predicate_sizes(n):
    from datafusion_amd.expr import col, lit
    t = random_table(np.random.default_rng(n), n, SPEC)
    run_filter(t, col("d") > lit(datetime.date(1995, 3, 15), pa.date32()), ["k", "p", "q"])


@pytest.mark.parametrize("op", ["=", "!=", "<", "<=", ">", ">="])
@pytest.mark.parametrize("colname", ["k", "d", "p", "q", "f", "c"])
def test_filter_all_comparisons_all_types(op, colname):
    from datafusion_amd.expr import BinaryExpr, col, lit
    t = random_table(np.random.default_rng(7), 5000, SPEC)
    typ = t.schema.field(colname).type
    value = {"k": 500, "d": 9200, "p": "12.50", "q": 3, "f": 10.0 / 7.0, "c": 2}[colname]
    run_filter(t, BinaryExpr(col(colname), op, lit(value, typ)))


def test_filter_with_nulls_drops_null_predicate_rows():
    from datafusion_amd.expr import col, lit
    t = random_table(np.random.default_rng(3), 20_000, SPEC, null_frac=0.2)
    got = run_filter(t, col("q") >= lit(0, pa.int32()))
    assert got.num_rows > 0 and got.column("q").null_count == 0
    assert got.column("p").null_count > 0  # payload NULLs survive the compaction


def test_filter_and_or_kleene():
    from datafusion_amd.expr import col, lit
    t = random_table(np.random.default_rng(4), 30_000, SPEC, null_frac=0.15)
    p1 = (col("q") > lit(0, pa.int32())).and_(col("k") < lit(500, pa.int64()))
    p2 = (col("q") > lit(10, pa.int32())).or_(col("d") <= lit(9000, pa.date32()))
    p3 = p1.or_(p2.not_())
    for p in (p1, p2, p3):
        run_filter(t, p)


def test_filter_is_null():
    from datafusion_amd.expr import col
    t = random_table(np.random.default_rng(5), 10_000, SPEC, null_frac=0.3)
    run_filter(t, col("p").is_null())
    run_filter(t, col("p").is_not_null())


def test_filter_compound_predicates_over_a_million_rows():
    """compound predicates over 1.1 M rows (the band l_shipdate >= a AND l_shipdate <= b, Kleene AND / OR / NOT over NULLs, a comparison of
    computed values, a division): the rows that pass are those of the oracle's FilterExec, in order"""
    from datafusion_amd.expr import col, lit
    t = random_table(np.random.default_rng(11), 1_100_000, SPEC, null_frac=0.1)
    band = (col("d") >= lit(9000, pa.date32())).and_(col("d") <= lit(9365, pa.date32()))
    kleene = ((col("q") > lit(0, pa.int32())).and_(col("k") < lit(500, pa.int64()))).or_((col("q") > lit(10, pa.int32())).or_(col("d") <= lit(9000, pa.date32())).not_())
    computed = (col("q").cast(pa.int64()) + col("k")) > lit(400, pa.int64())
    for pred in (band, kleene, computed, (col("k") / lit(7, pa.int64())) > lit(50, pa.int64())):
        run_filter(t, pred, ["k", "p", "d"])


def test_filter_between_q_shapes():
    """the three selectivities of BASELINE config 2 (l_shipdate predicates of Q1 / Q3 / a 1-year band)"""
    from datafusion_amd import tpch
    from datafusion_amd.expr import col, lit
    li = tpch.lineitem(0.01)
    d = lambda s: lit(datetime.date.fromisoformat(s), pa.date32())
    q3 = ["l_orderkey", "l_extendedprice", "l_discount"]
    a = run_filter(li, col("l_shipdate") <= d("1998-09-02"), None)
    b = run_filter(li, col("l_shipdate") > d("1995-03-15"), q3)
    c = run_filter(li, (col("l_shipdate") >= d("1994-01-01")).and_(col("l_shipdate") <= d("1994-12-31")), q3)
    assert a.num_rows > b.num_rows > c.num_rows > 0


def test_projection_decimal_arithmetic_types_and_values():
    """Q1/Q3 expressions: l_extendedprice * (1 - l_discount) -> Decimal128(38,4); * (1 + l_tax) -> (38,6)
    (scales pinned by tpch/answers/q1.slt.part:42-45, literal type by tpch/plans/q1.slt.part:45-46)"""
    from datafusion_amd import ops, tpch
    from datafusion_amd.expr import col, lit
    from datafusion_amd.table import DeviceTable
    from oracle import oracle
    li = tpch.lineitem(0.002)
    one = lit(1, pa.decimal128(20, 0))
    disc_price = col("l_extendedprice") * (one - col("l_discount"))
    charge = disc_price * (one + col("l_tax"))
    exprs = [(disc_price, "disc_price"), (charge, "charge"), (col("l_quantity") + col("l_tax"), "s"), (col("l_orderkey") * lit(3), "k3")]
    got = ops.project(DeviceTable.from_arrow(li), exprs).to_arrow()
    exp = oracle.project(li, [(to_oracle_expr(e), n) for e, n in exprs])
    assert got.schema.field("disc_price").type == pa.decimal128(38, 4)
    assert got.schema.field("charge").type == pa.decimal128(38, 6)
    assert_tables_equal(got, exp, ordered=True)


def test_projection_cast_and_float():
    from datafusion_amd import ops
    from datafusion_amd.expr import col, lit
    from datafusion_amd.table import DeviceTable
    from oracle import oracle
    t = random_table(np.random.default_rng(9), 4097, SPEC, null_frac=0.1)
    exprs = [(col("k").cast(pa.decimal128(20, 0)) * col("p"), "a"), (col("f") * lit(2.5) + col("f"), "b"),
             (col("q").cast(pa.int64()) - col("k"), "c"), (col("p").cast(pa.decimal128(20, 4)) - col("p"), "d")]
    got = ops.project(DeviceTable.from_arrow(t), exprs).to_arrow()
    exp = oracle.project(t, [(to_oracle_expr(e), n) for e, n in exprs])
    assert_tables_equal(got, exp, ordered=True)


def test_unsupported_type_is_an_error_not_a_fallback():
    from datafusion_amd import DfgpuError
    from datafusion_amd.table import DeviceTable
    with pytest.raises(DfgpuError, match="unsupported Arrow type"):
        DeviceTable.from_arrow(pa.table({"s": pa.array([[1], [2, 3]], pa.list_(pa.int32()))}))      # (strings are imported since ABI 8: tests/test_gpu_strings.py)


def test_float64_to_decimal_cast_values_and_errors():
    """CastExpr Float64 -> Decimal128 with safe = false (arrow-cast cast_floating_point_to_decimal128): (v * 10^scale).round(), an error —
    not garbage — for NaN / inf / beyond i128 and for values beyond the declared precision; NULL rows are never looked at.  Values are
    checked against pyarrow's cast of the same column."""
    from datafusion_amd import _lib, ops
    from datafusion_amd.expr import col, lit
    from datafusion_amd.table import DeviceTable
    rng = np.random.default_rng(8)
    v = np.round(rng.uniform(-1e6, 1e6, 50_000), 3)
    t = pa.table({"f": pa.array(v, mask=rng.random(len(v)) < 0.1)})
    got = ops.project(DeviceTable.from_arrow(t), [(col("f").cast(pa.decimal128(15, 2)), "d")]).to_arrow()
    import decimal
    half_away = lambda y: int(decimal.Decimal(y).quantize(decimal.Decimal(1), rounding=decimal.ROUND_HALF_UP))    # f64::round on the exact value of the double
    exp = pa.array([None if x is None else half_away(x * 100.0) for x in t.column("f").to_pylist()], pa.int64())
    unscaled = pa.array([None if x is None else int(x.scaleb(2)) for x in got.column("d").to_pylist()], pa.int64())
    assert unscaled.equals(exp)
    for bad, msg in ((float("nan"), "Cannot cast to Decimal128"), (float("inf"), "Cannot cast to Decimal128"), (1e40, "Cannot cast to Decimal128"),
                     (1e14, "too large to store in a Decimal128")):
        tb = pa.table({"f": pa.array([1.0, bad, 2.0])})
        with pytest.raises(_lib.DfgpuError, match=msg):
            ops.project(DeviceTable.from_arrow(tb), [(col("f").cast(pa.decimal128(15, 2)), "d")])
        with pytest.raises(_lib.DfgpuError, match=msg):                                                      # the literal is folded on the host
            ops.project(DeviceTable.from_arrow(tb), [(lit(bad, pa.float64()).cast(pa.decimal128(15, 2)), "d")])
        masked = pa.table({"f": pa.array([1.0, bad, 2.0], mask=np.array([False, True, False]))})             # the offending slot is NULL: no error
        assert ops.project(DeviceTable.from_arrow(masked), [(col("f").cast(pa.decimal128(15, 2)), "d")]).to_arrow().column("d").null_count == 1
    with pytest.raises(_lib.DfgpuError, match="too large to store"):                                         # scalar scale-down beyond the precision
        ops.project(DeviceTable.from_arrow(pa.table({"f": pa.array([1.0])})), [(lit("999.99", pa.decimal128(7, 2)).cast(pa.decimal128(2, 0)), "d")])


def test_short_scans_take_one_launch_and_long_ones_three_with_the_same_result():
    """round 6: prefix sums of up to 8 x 4096 elements run as ONE chained kernel (decoupled look-back over the tiles), longer ones as reduce /
    sums / down; forcing either form on a 3 M-row filter gives the same rows (the scan places every compacted row)"""
    import numpy as np
    import pyarrow as pa
    from datafusion_amd import ops
    from datafusion_amd.expr import col, lit
    from datafusion_amd.table import DeviceTable
    rng = np.random.default_rng(8)
    n = 3_000_017
    t = pa.table({"a": pa.array(rng.integers(0, 100, n), type=pa.int32()), "b": pa.array(np.arange(n), type=pa.int64())})
    dev = DeviceTable.from_arrow(t)
    want = t.filter(pa.compute.less(t.column("a"), 37))
    for tiles in ("0", "1000000", None):
        ops.set_options(scan__chained_max_tiles=tiles)
        got = ops.filter(dev, col("a") < lit(37, pa.int32())).to_arrow()
        assert got.equals(want), tiles
    ops.set_options(scan__chained_max_tiles=None)
