"""CaseExpr (physical-expr/src/expressions/case.rs:274, `case_when_no_expr` :895-980): CASE WHEN c THEN a [WHEN ...]
[ELSE e] END through every evaluator — ProjectionExec / FilterExec column-at-a-time (expr.hip k_select), and as
aggregate arguments / predicates of the fused node (register program, LDS-register-file program, specialised HIP
source: GATE / MERGE instructions, rowprog.hip) — against the oracle's restatement: THEN where the condition is TRUE,
ELSE where it is FALSE or NULL, NULL without ELSE."""
import os
from decimal import Decimal

import numpy as np
import pyarrow as pa
import pytest

from tests.test_gpu_aggregate import assert_agg_equal, oracle_agg
from tests.util import assert_tables_equal, random_table, to_oracle_expr

pytestmark = pytest.mark.gpu

SPEC = {"i": (pa.int32(), -1000, 1000), "k": (pa.int64(), -10**9, 10**9), "g": (pa.int64(), 0, 700), "d": (pa.decimal128(15, 2), -10**9, 10**9),
        "e": (pa.decimal128(15, 2), 0, 11), "f": (pa.float64(), -10**6, 10**6), "dt": (pa.date32(), 8000, 10000)}


def _table(seed, n, nulls):
    rng = np.random.default_rng(seed)
    t = random_table(rng, n, SPEC, nulls)
    return t.append_column("b", pa.array(rng.integers(65, 68, size=n).astype(np.uint8)))


def _cases():
    from datafusion_amd.expr import case, col, lit
    d2 = pa.decimal128(15, 2)
    c1, c2 = col("i") > lit(0, pa.int32()), col("d") < col("e")      # both NULL where their inputs are
    return [
        (case([(c1, lit(1))], lit(0)), "int_literals"),                                   # Q12's shape
        (case([(c1, col("k"))], col("g")), "i64_columns"),
        (case([(c1, col("i"))]), "i32_no_else"),
        (case([(c2, col("d") + col("e"))], col("d") - col("e")), "decimal_expr"),
        (case([(c1, col("d"))], lit(Decimal("0.00"), d2)), "decimal_vs_literal"),
        (case([(c1, lit(None, d2))], col("e")), "null_then"),
        (case([(c2, col("f"))], lit(-1.5)), "f64"),
        (case([(c1, col("dt"))]), "date_no_else"),
        (case([(c1, col("b"))], lit(0, pa.uint8())), "u8"),
        (case([(c1, c2)], col("k").is_null()), "bool_result"),
        (case([(c1, lit(1)), (c2, lit(2)), (col("f") > lit(0.0), lit(3))], lit(4)), "three_whens"),
        (case([(c1.or_(c2), col("k") * lit(2))], case([(col("f") > lit(0.0), col("g"))], lit(-7))) + lit(1), "nested_in_arithmetic"),
        (case([(lit(True), col("k"))], col("g")), "constant_condition"),
        (case([(lit(None, pa.bool_()), col("k"))], col("g")), "null_condition"),
    ]


@pytest.mark.parametrize("nulls", [0.0, 0.2])
@pytest.mark.parametrize("n", [1, 64, 10_007])
def test_case_projection_and_filter(n, nulls):
    from datafusion_amd import ops
    from datafusion_amd.expr import case, col, lit
    from datafusion_amd.table import DeviceTable
    from oracle import oracle as O
    t = _table(7 + n, n, nulls)
    dev = DeviceTable.from_arrow(t)
    exprs = _cases()
    got = ops.project(dev, exprs).to_arrow()
    exp = O.project(t, [(to_oracle_expr(e), nm) for e, nm in exprs])
    assert_tables_equal(got, exp, ordered=True)
    pred = case([(col("i") > lit(0, pa.int32()), col("d") < col("e"))], col("k") > lit(0))
    assert_tables_equal(ops.filter(dev, pred).to_arrow(), O.filter(t, to_oracle_expr(pred), t.column_names), ordered=True)


@pytest.mark.parametrize("evaluator", ["specialised", "interpreted", "column_at_a_time"])
@pytest.mark.parametrize("keys", ["none", "one_byte", "int64", "two_keys"])
@pytest.mark.parametrize("nulls", [0.0, 0.2])
def test_case_in_aggregate_arguments_and_predicate(keys, nulls, evaluator):
    from datafusion_amd import ops
    from datafusion_amd.expr import case, col, lit
    from datafusion_amd.table import DeviceTable
    from oracle import oracle as O
    t = _table(31, 20_011, nulls)
    gb = {"none": [], "one_byte": [(col("b"), "b")], "int64": [(col("g"), "g")], "two_keys": [(col("g"), "g"), (col("b"), "b")]}[keys]
    numeric = [(e, nm) for e, nm in _cases() if nm not in ("bool_result", "date_no_else")]
    aggs = [("sum", e, f"s_{nm}") for e, nm in numeric[:7]] + [("count", e, f"c_{nm}") for e, nm in numeric[7:]] + [("avg", numeric[3][0], "avg_dec"), ("count", None, "n")]
    pred = case([(col("i") > lit(0, pa.int32()), col("d") < col("e"))], col("k") > lit(0))
    ops.set_fusion(evaluator != "column_at_a_time")
    if evaluator == "specialised":
        ops.set_options(jit="1", jit__min_rows="0", jit__strict="1")
    else:
        ops.set_options(jit="0")
    try:
        got = ops.aggregate(DeviceTable.from_arrow(t), gb, aggs, "Single", predicate=pred).to_arrow()
    finally:
        ops.set_fusion(True)
        ops.reset_options()
    src = O.filter(t, to_oracle_expr(pred), t.column_names)
    assert_agg_equal(got, oracle_agg(src, gb, aggs, "Single"), ordered=True)


def test_case_type_errors():
    from datafusion_amd import _lib, ops
    from datafusion_amd.expr import case, col, lit
    from datafusion_amd.table import DeviceTable
    dev = DeviceTable.from_arrow(_table(1, 10, 0.0))
    with pytest.raises(_lib.DfgpuError, match="CASE branch types differ"):
        ops.project(dev, [(case([(col("i") > lit(0, pa.int32()), col("k"))], col("i")), "x")])
    with pytest.raises(_lib.DfgpuError, match="CASE WHEN condition must be Boolean"):
        ops.project(dev, [(case([(col("i"), col("k"))], col("k")), "x")])
