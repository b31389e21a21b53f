"""CaseExpr pinned against the reference's own unit tests (physical-expr/src/expressions/case.rs:1586-2179; fixture
tests/golden/case_expr.json).  `CASE x WHEN v THEN ...` is lowered to `CASE WHEN x = v THEN ...` — the form the
C ABI carries (DFGPU_EXPR_CASE) and the rewrite DataFusion's own simplifier applies.  CPU leg: the oracle; GPU leg:
ProjectionExec through the C ABI (string column dictionary-encoded)."""
import pyarrow as pa
import pytest

from tests.util import load_golden

GOLD = load_golden("case_expr.json")
TYPES = {"i32": pa.int32(), "f64": pa.float64(), "utf8": pa.string(), "bool": pa.bool_()}


def build_table(batch):
    cols = {}
    for name, c in batch.items():
        if c["type"] == "utf8":   # dictionary-encoded (ascending dictionary), NULL = null index
            values = sorted({v for v in c["values"] if v is not None})
            idx = pa.array([None if v is None else values.index(v) for v in c["values"]], pa.uint8())
            cols[name] = pa.DictionaryArray.from_arrays(idx, pa.array(values, pa.string()))
        else:                     # raw values + validity bits: NULL rows keep their raw value, as in case_test_batch_nulls
            import numpy as np
            vals = np.array(c["values"], dtype=np.float64)
            valid = np.packbits(np.array(c["valid"], dtype=np.uint8), bitorder="little")
            cols[name] = pa.Array.from_buffers(pa.float64(), len(vals), [pa.py_buffer(valid.tobytes()), pa.py_buffer(vals.tobytes())])
    return pa.table(cols)


def to_expr(node):
    from datafusion_amd.expr import col, lit
    kind = node[0]
    if kind == "col":
        return col(node[1])
    if kind == "lit":
        return lit(node[1], TYPES[node[2]])
    a, b = to_expr(node[1]), to_expr(node[2])
    return a.eq(b) if kind == "=" else a.and_(b)


def case_expr(rec):
    from datafusion_amd.expr import case
    whens = []
    for w, t in rec["whens"]:
        cond = to_expr(w) if rec["base"] is None else to_expr(rec["base"]).eq(to_expr(w))
        whens.append((cond, to_expr(t)))
    return case(whens, None if rec["else_"] is None else to_expr(rec["else_"]))


def check(rec, got: pa.ChunkedArray):
    if rec["expected_type"] == "utf8":
        assert pa.types.is_dictionary(got.type) and got.null_count == len(rec["expected"])
        return
    assert got.type == TYPES[rec["expected_type"]]
    assert got.to_pylist() == rec["expected"], (rec["name"], rec["source"])


@pytest.mark.parametrize("rec", GOLD, ids=[r["name"] for r in GOLD])
def test_oracle_case_known_answers(rec):
    from datafusion_amd import physical_plan as P
    from tests import plan_oracle
    plan = P.ProjectionExec([(case_expr(rec), "r")], P.MemoryExec(build_table(rec["batch"]), "t"))
    check(rec, plan_oracle.collect(plan).column("r"))


@pytest.mark.gpu
@pytest.mark.parametrize("rec", GOLD, ids=[r["name"] for r in GOLD])
def test_gpu_case_known_answers(rec):
    from datafusion_amd import ops
    from datafusion_amd.table import DeviceTable
    dev = DeviceTable.from_arrow(build_table(rec["batch"]))
    check(rec, ops.project(dev, [(case_expr(rec), "r")]).to_arrow().column("r"))
    if rec["name"] == "case_with_scalar_predicate":     # "one row" leg of the reference test
        one = DeviceTable.from_arrow(pa.table({"load4": pa.array([1.1], pa.float64())}))
        assert ops.project(one, [(case_expr(rec), "r")]).to_arrow().column("r").to_pylist() == [1.1]
