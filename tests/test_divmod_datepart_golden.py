"""BinaryExpr Divide / Modulo and date_part: the oracle (CPU) and the device evaluators against the reference's own known answers
(tests/golden/binary_expr_divmod.json, extracted from physical-expr/src/expressions/binary.rs by
tests/golden/extract_reference_divmod_goldens.py) and against each other on random inputs; date_part against Python's calendar."""
import datetime
from decimal import Decimal

import numpy as np
import pyarrow as pa
import pytest

from tests.test_oracle_expr_golden import column, parse_type, unscaled
from tests.util import load_golden

CASES = load_golden("binary_expr_divmod.json")
OK = [c for c in CASES if "error" not in c]
ERR = [c for c in CASES if "error" in c]


def _oracle_expr(case):
    a = ("cast", ("col", "a"), parse_type(case["cast_a"])) if "cast_a" in case else ("col", "a")
    b = ("lit", case["b_scalar"]["value"], parse_type(case["b_scalar"]["type"])) if "b_scalar" in case else ("col", "b")
    return ("bin", case["op"], a, b)


def _table(case):
    cols = {"a": column(case["a"])}
    if "b" in case:
        cols["b"] = column(case["b"])
    return pa.table(cols)


def _expected(case):
    return case["expected_unscaled"] if "expected_unscaled" in case else case["expected"]


def _values(arr):
    arr = arr.combine_chunks() if isinstance(arr, pa.ChunkedArray) else arr
    return unscaled(arr) if pa.types.is_decimal128(arr.type) else arr.to_pylist()


@pytest.mark.parametrize("case", OK, ids=[c["name"] for c in OK])
def test_oracle_divide_modulo_match_reference(case):
    from oracle import oracle
    out = oracle.project(_table(case), [(_oracle_expr(case), "r")]).column("r")
    assert out.type == parse_type(case["expected_type"]), case["source"]
    assert _values(out) == _expected(case), case["source"]


@pytest.mark.parametrize("case", ERR, ids=[c["name"] for c in ERR])
def test_oracle_divide_by_zero_is_an_error(case):
    from oracle import oracle
    with pytest.raises(ZeroDivisionError, match=case["error"]):
        oracle.project(_table(case), [(_oracle_expr(case), "r")])


def test_oracle_date_part_matches_the_calendar():
    from oracle import oracle
    days = [-719162, -1, 0, 58, 59, 60, 365, 789, 8035, 9298, 10591, 11016, 11017, 19000, 2932896]   # 0001-01-01 .. 9999-12-31, leap days
    t = pa.table({"d": pa.array(days, pa.int32()).cast(pa.date32())})
    for part, f in (("year", lambda d: d.year), ("month", lambda d: d.month), ("day", lambda d: d.day)):
        got = oracle.project(t, [(("date_part", part, ("col", "d")), "r")]).column("r").to_pylist()
        assert got == [f(datetime.date(1970, 1, 1) + datetime.timedelta(days=x)) for x in days], part


def _gpu_expr(case):
    from datafusion_amd.expr import BinaryExpr, col, lit
    a = col("a").cast(parse_type(case["cast_a"])) if "cast_a" in case else col("a")
    b = lit(case["b_scalar"]["value"], parse_type(case["b_scalar"]["type"])) if "b_scalar" in case else col("b")
    return BinaryExpr(a, case["op"], b)


@pytest.mark.gpu
@pytest.mark.parametrize("case", OK, ids=[c["name"] for c in OK])
def test_gpu_divide_modulo_match_reference(case):
    from datafusion_amd import ops
    from datafusion_amd.table import DeviceTable
    out = ops.project(DeviceTable.from_arrow(_table(case)), [(_gpu_expr(case), "r")]).to_arrow().column("r")
    assert out.type == parse_type(case["expected_type"]), case["source"]
    assert _values(out) == _expected(case), case["source"]


@pytest.mark.gpu
@pytest.mark.parametrize("case", ERR, ids=[c["name"] for c in ERR])
def test_gpu_divide_by_zero_is_an_error(case):
    from datafusion_amd import _lib, ops
    from datafusion_amd.table import DeviceTable
    with pytest.raises(_lib.DfgpuError, match=case["error"]):
        ops.project(DeviceTable.from_arrow(_table(case)), [(_gpu_expr(case), "r")])


@pytest.mark.gpu
def test_gpu_zero_divisor_under_a_null_is_not_an_error():
    """arrow-arith visits valid rows only (try_binary): a NULL row's data slot may hold anything, 0 included"""
    from datafusion_amd import ops
    from datafusion_amd.expr import col
    from datafusion_amd.table import DeviceTable
    b = pa.array([2, 0, 4], pa.int32(), mask=np.array([False, True, False]))
    t = pa.table({"a": pa.array([8, 9, None], pa.int32()), "b": b})
    assert ops.project(DeviceTable.from_arrow(t), [(col("a") / col("b"), "q"), (col("a") % col("b"), "m")]).to_arrow().to_pydict() == {"q": [4, None, None], "m": [0, None, None]}


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["int32", "int64", "float64", "dec_dec", "dec_scalar", "scalar_dec"])
def test_gpu_divide_modulo_random_vs_oracle(kind):
    from datafusion_amd import ops
    from datafusion_amd.expr import col, lit
    from datafusion_amd.table import DeviceTable
    from oracle import oracle
    from tests.util import random_table, to_oracle_expr
    rng = np.random.default_rng(11)
    n = 3000
    if kind in ("int32", "int64"):
        typ = pa.int32() if kind == "int32" else pa.int64()
        t = random_table(rng, n, {"a": (typ, -10**6, 10**6), "b": (typ, 1, 999)}, null_frac=0.1)
        neg = pa.array(np.where(rng.random(n) < 0.5, -1, 1), typ)
        import pyarrow.compute as pc
        t = t.set_column(1, "b", pc.multiply(t.column("b"), neg))
        exprs = [(col("a") / col("b"), "q"), (col("a") % col("b"), "m"), (col("a") / lit(7, typ), "qs"), (lit(1000, typ) % col("b"), "ms")]
    elif kind == "float64":
        t = random_table(rng, n, {"a": (pa.float64(), -10**6, 10**6), "b": (pa.float64(), -50, 50)}, null_frac=0.1)   # b holds zeros: inf / NaN
        exprs = [(col("a") / col("b"), "q"), (col("a") % col("b"), "m")]
    else:
        t = random_table(rng, n, {"a": (pa.decimal128(15, 2), -10**12, 10**12), "b": (pa.decimal128(12, 4), 1, 10**9)}, null_frac=0.1)
        if kind == "dec_dec":
            exprs = [(col("a") / col("b"), "q"), (col("a") % col("b"), "m"), (col("b") / col("a"), "q2")]
            # a may be zero in q2: keep it away from zero
            import pyarrow.compute as pc
            a = t.column("a").combine_chunks()
            a = pc.if_else(pc.equal(a, pa.scalar(Decimal("0.00"), a.type)), pa.scalar(Decimal("1.00"), a.type), a)
            t = t.set_column(0, "a", a)
        elif kind == "dec_scalar":
            exprs = [(col("a") / lit(Decimal("7.0"), pa.decimal128(2, 1)), "q"), (col("a") % lit(Decimal("0.37"), pa.decimal128(3, 2)), "m")]
        else:
            exprs = [(lit(Decimal("100.00"), pa.decimal128(5, 2)) / col("b"), "q"), (lit(Decimal("12345.678"), pa.decimal128(8, 3)) % col("b"), "m")]
    got = ops.project(DeviceTable.from_arrow(t), exprs).to_arrow()
    exp = oracle.project(t, [(to_oracle_expr(e), nm) for e, nm in exprs])
    assert got.schema.types == exp.schema.types
    for nm in got.column_names:
        g, e = got.column(nm).to_pylist(), exp.column(nm).to_pylist()
        if kind == "float64":
            assert all((x is None and y is None) or (x != x and y != y) or x == y for x, y in zip(g, e)), nm
        else:
            assert g == e, nm


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [True, False])
def test_gpu_date_part_column_at_a_time_and_in_the_fused_node(fused):
    """date_part(YEAR) as a projection and as a group key of the fused aggregate node (Q7 / Q8 / Q9's l_year / o_year)"""
    import os

    from datafusion_amd import ops
    from datafusion_amd.expr import col, date_part
    from datafusion_amd.table import DeviceTable
    from oracle import oracle
    from tests.test_gpu_aggregate import assert_agg_equal, oracle_agg
    from tests.util import random_table, to_oracle_expr
    rng = np.random.default_rng(3)
    t = random_table(rng, 20000, {"d": (pa.date32(), -800000, 2900000), "v": (pa.int64(), 0, 1000)}, null_frac=0.05)
    dev = DeviceTable.from_arrow(t)
    exprs = [(date_part(p, col("d")), p) for p in ("year", "month", "day")]
    got = ops.project(dev, exprs).to_arrow()
    exp = oracle.project(t, [(to_oracle_expr(e), nm) for e, nm in exprs])
    assert got.schema.types == exp.schema.types and got.to_pydict() == exp.to_pydict()
    t2 = random_table(rng, 50000, {"d": (pa.date32(), 8035, 10591), "v": (pa.int64(), 0, 1000)})
    gb, aggs = [(date_part("year", col("d")), "l_year")], [("sum", col("v"), "s"), ("count", None, "n")]
    ops.set_fusion(fused)
    try:
        ops.set_options(jit="1", jit__min_rows="0", jit__strict="1")
        g = ops.aggregate(DeviceTable.from_arrow(t2), gb, aggs, "Single", predicate=col("v") > 10).to_arrow()
    finally:
        ops.set_fusion(True)
        ops.set_options(jit=None, jit__min_rows=None, jit__strict=None)
    src = oracle.filter(t2, to_oracle_expr(col("v") > 10), t2.column_names)
    assert_agg_equal(g, oracle_agg(src, gb, aggs), ordered=True)


@pytest.mark.gpu
def test_gpu_like_on_a_dictionary_column_matches_pyarrow():
    import pyarrow.compute as pc

    from datafusion_amd import ops
    from datafusion_amd.expr import col
    from datafusion_amd.table import DeviceTable
    words = sorted({a + " " + b + " " + c for a in ("STANDARD", "SMALL", "PROMO", "ECONOMY") for b in ("ANODIZED", "BRUSHED", "plated") for c in ("TIN", "STEEL", "COPPER%", "a_b")}
                   | {"", "PROMO", "été", "naïve PROMO", "50% off"})
    rng = np.random.default_rng(1)
    codes = rng.integers(0, len(words), size=5000).astype(np.int32)
    mask = rng.random(5000) < 0.1
    arr = pa.DictionaryArray.from_arrays(pa.array(codes, mask=mask), pa.array(words, pa.string()))
    t = pa.table({"s": arr, "i": pa.array(np.arange(5000, dtype=np.int64))})
    dev = DeviceTable.from_arrow(t)
    plain = arr.cast(pa.string())
    for pattern, ci in (("PROMO%", False), ("%STEEL", False), ("%BRUSHED%", False), ("promo%", True), ("S_ALL %", False), ("%", False), ("", False), ("PROMO", False),
                        ("%\\%", False), ("%a\\_b", False), ("_t_", False), ("%ï%", False), ("no such", False), ("%PLATED%", True)):
        for negated in (False, True):
            keep = pc.match_like(plain, pattern, ignore_case=ci)
            keep = pc.invert(keep) if negated else keep
            want = t.filter(pc.fill_null(keep, False)).column("i").to_pylist()
            got = ops.filter(dev, col("s").like(pattern, negated=negated, case_insensitive=ci), ["i"]).to_arrow().column("i").to_pylist()
            assert got == want, (pattern, ci, negated)
