"""Pins the oracle's expression evaluator (and, on the GPU box, the device evaluators) against the reference's own
known-answer tests of BinaryExpr over Decimal128 (physical-expr/src/expressions/binary.rs: arithmetic_decimal_expr_test
:4728-4804 — result precision / scale of arrow-arith's add / sub / mul — and comparison_decimal_expr_test :4355-4428);
tests/golden/binary_expr_decimal.json carries the vectors with their source lines."""
from decimal import Decimal

import pyarrow as pa
import pytest

from tests.util import load_golden

CASES = load_golden("binary_expr_decimal.json")


def parse_type(s):
    if s == "int32":
        return pa.int32()
    p, sc = s[len("decimal128("):-1].split(",")
    return pa.decimal128(int(p), int(sc))


def column(spec):
    t = parse_type(spec["type"])
    if "values" in spec:
        return pa.array(spec["values"], type=t)
    return pa.array([None if v is None else Decimal(v).scaleb(-t.scale) for v in spec["unscaled"]], type=t)


def unscaled(arr):
    t = arr.type
    return [None if v is None else int(v.scaleb(t.scale)) for v in arr.to_pylist()]


ARITH = [c for c in CASES if "op" in c]
CMP = [c for c in CASES if "comparisons" in c]


@pytest.mark.parametrize("case", ARITH, ids=[c["name"] for c in ARITH])
def test_oracle_decimal_arithmetic_matches_reference(case):
    from oracle import oracle
    t = pa.table({"a": column(case["a"]), "b": column(case["b"])})
    side = lambda n: ("cast", ("col", "a"), parse_type(case["cast_a"])) if n == "a" else ("col", "b")
    out = oracle.project(t, [(("bin", case["op"], side(case["left"]), side(case["right"])), "r")]).column("r")
    assert out.type == parse_type(case["expected_type"]), case["source"]
    assert unscaled(out.combine_chunks() if isinstance(out, pa.ChunkedArray) else out) == case["expected_unscaled"], case["source"]


@pytest.mark.parametrize("case", CMP, ids=[c["name"] for c in CMP])
def test_oracle_decimal_comparisons_match_reference(case):
    from oracle import oracle
    b = column(case["b"])
    t = pa.table({"b": b})
    lit = ("lit", Decimal(case["scalar_unscaled"]).scaleb(-b.type.scale), b.type)
    for op, expected in case["comparisons"].items():
        out = oracle.project(t, [(("bin", op, ("col", "b"), lit), "r")]).column("r")
        assert out.to_pylist() == expected, (case["source"], op)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ARITH, ids=[c["name"] for c in ARITH])
def test_gpu_decimal_arithmetic_matches_reference(case):
    from datafusion_amd import ops
    from datafusion_amd.expr import BinaryExpr, col
    from datafusion_amd.table import DeviceTable
    t = pa.table({"a": column(case["a"]), "b": column(case["b"])})
    side = lambda n: col("a").cast(parse_type(case["cast_a"])) if n == "a" else col("b")
    out = ops.project(DeviceTable.from_arrow(t), [(BinaryExpr(side(case["left"]), case["op"], side(case["right"])), "r")]).to_arrow().column("r")
    assert out.type == parse_type(case["expected_type"]), case["source"]
    assert unscaled(out.combine_chunks()) == case["expected_unscaled"], case["source"]


@pytest.mark.gpu
@pytest.mark.parametrize("case", CMP, ids=[c["name"] for c in CMP])
def test_gpu_decimal_comparisons_match_reference(case):
    from datafusion_amd import ops
    from datafusion_amd.expr import BinaryExpr, col, lit
    from datafusion_amd.table import DeviceTable
    b = column(case["b"])
    dt = DeviceTable.from_arrow(pa.table({"b": b}))
    scalar = lit(str(Decimal(case["scalar_unscaled"]).scaleb(-b.type.scale)), b.type)
    for op, expected in case["comparisons"].items():
        out = ops.project(dt, [(BinaryExpr(col("b"), op, scalar), "r")]).to_arrow().column("r")
        assert out.to_pylist() == expected, (case["source"], op)


# ---------------------------------------------------------------------------------- Kleene logic and Int32 arithmetic
LOGIC = load_golden("binary_expr_logic.json")


def _logic_table(rec):
    typ = pa.bool_() if rec["type"] == "bool" else pa.int32()
    return pa.table({"a": pa.array(rec["a"], typ), "b": pa.array(rec["b"], typ)})


def _logic_expr(rec):
    from datafusion_amd.expr import BinaryExpr, col
    return BinaryExpr(col("a"), rec["op"], col("b"))


@pytest.mark.parametrize("rec", LOGIC, ids=[r["name"] for r in LOGIC])
def test_oracle_binary_expr_logic_and_int_arithmetic(rec):
    """and_with_nulls_op / or_with_nulls_op (binary.rs:3469-3765: and_kleene / or_kleene over all nine combinations),
    plus_op / minus_op / multiply_op (binary.rs:2238-2760)"""
    from oracle import oracle as O
    from tests.util import to_oracle_expr
    got = O.project(_logic_table(rec), [(to_oracle_expr(_logic_expr(rec)), "r")]).column("r")
    assert got.to_pylist() == rec["expected"], rec["source"]


@pytest.mark.gpu
@pytest.mark.parametrize("rec", LOGIC, ids=[r["name"] for r in LOGIC])
def test_gpu_binary_expr_logic_and_int_arithmetic(rec):
    """column-at-a-time (ProjectionExec) and inside the fused aggregate node: as its predicate (Kleene) or its argument (arithmetic)"""
    from datafusion_amd import ops
    from datafusion_amd.table import DeviceTable
    dev = DeviceTable.from_arrow(_logic_table(rec))
    e = _logic_expr(rec)
    got = ops.project(dev, [(e, "r")]).to_arrow().column("r")
    assert got.to_pylist() == rec["expected"], rec["source"]
    if rec["type"] == "bool":
        n = ops.aggregate(dev, [], [("count", None, "n")], "Single", predicate=e).to_arrow().to_pylist()[0]["n"]
        assert n == sum(1 for v in rec["expected"] if v is True)
    else:
        s = ops.aggregate(dev, [], [("sum", e.cast(pa.int64()), "s")], "Single").to_arrow().to_pylist()[0]["s"]
        assert s == sum(rec["expected"])
