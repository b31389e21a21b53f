"""InListExpr pinned against the reference's own unit tests (physical-expr/src/expressions/in_list.rs `run_test_cases`
:725-855 for Int32 / Int64 / UInt8 / UInt32 / UInt64 / Utf8 / Date32 / Decimal128, `in_list_float64` with NaN / -NaN / NULL
list members; fixture tests/golden/in_list.json).  `x [NOT] IN (...)` crosses the boundary as the Kleene OR of equalities
(expr.InListExpr.lowered), so these also pin `=` / OR / NOT NULL semantics and the total-order equality of Float64.
CPU leg: the oracle; GPU leg: ProjectionExec through the C ABI.  The reference's Boolean case is not taken: the device
path has no Boolean = Boolean comparison."""
import struct
from decimal import Decimal

import numpy as np
import pyarrow as pa
import pytest

from tests.util import load_golden

GOLD = [r for r in load_golden("in_list.json") if r["type"] != "bool"]
TYPES = {"i32": pa.int32(), "i64": pa.int64(), "u8": pa.uint8(), "u32": pa.uint32(), "u64": pa.uint64(), "f64": pa.float64(), "date32": pa.date32(),
         "decimal128(10,2)": pa.decimal128(10, 2), "utf8": pa.string()}
NAN, NEG_NAN = struct.unpack("<d", struct.pack("<Q", 0x7FF8000000000000))[0], struct.unpack("<d", struct.pack("<Q", 0xFFF8000000000000))[0]


def value(v, typ):
    if v is None:
        return None
    if typ == "f64":
        return {"NaN": NAN, "-NaN": NEG_NAN}.get(v, v)
    if typ.startswith("decimal"):
        return Decimal(v).scaleb(-2)
    return v


def column(rec) -> pa.Array:
    typ, vals = rec["type"], [value(v, rec["type"]) for v in rec["column"]]
    if typ == "utf8":
        d = sorted(v for v in vals if v is not None)
        return pa.DictionaryArray.from_arrays(pa.array([None if v is None else d.index(v) for v in vals], pa.uint8()), pa.array(d, pa.string()))
    if typ == "f64":     # exact bit patterns (the sign of a NaN matters: Float64 equality is total-order equality)
        raw = np.array([0.0 if v is None else v for v in vals], dtype=np.float64)
        valid = np.packbits(np.array([v is not None for v in vals], dtype=np.uint8), bitorder="little")
        return pa.Array.from_buffers(pa.float64(), len(vals), [pa.py_buffer(valid.tobytes()), pa.py_buffer(raw.tobytes())])
    if typ == "date32":
        return pa.array(vals, pa.int32()).cast(pa.date32())
    return pa.array(vals, TYPES[typ])


def in_list_expr(rec):
    from datafusion_amd.expr import col, lit
    return col("a").in_list([lit(value(v, rec["type"]), TYPES[rec["type"]]) for v in rec["list"]], rec["negated"])


@pytest.mark.parametrize("rec", GOLD, ids=[r["name"] for r in GOLD])
def test_oracle_in_list_known_answers(rec):
    from datafusion_amd import physical_plan as P
    from tests import plan_oracle
    plan = P.ProjectionExec([(in_list_expr(rec), "r")], P.MemoryExec(pa.table({"a": column(rec)}), "t"))
    got = plan_oracle.collect(plan).column("r")
    assert got.type == pa.bool_() and got.to_pylist() == rec["expected"], rec["source"]


@pytest.mark.gpu
@pytest.mark.parametrize("rec", GOLD, ids=[r["name"] for r in GOLD])
def test_gpu_in_list_known_answers(rec):
    from datafusion_amd import ops
    from datafusion_amd.table import DeviceTable
    t = pa.table({"a": column(rec), "one": pa.array([1] * len(rec["column"]), pa.int64())})
    dev = DeviceTable.from_arrow(t)
    got = ops.project(dev, [(in_list_expr(rec), "r")]).to_arrow().column("r")
    assert got.type == pa.bool_() and got.to_pylist() == rec["expected"], rec["source"]
    # as a FilterExec predicate (NULL drops the row) and inside the fused aggregate node (register program)
    kept = ops.filter(dev, in_list_expr(rec), ["one"]).num_rows
    assert kept == sum(1 for e in rec["expected"] if e is True)
    from datafusion_amd.expr import col
    n = ops.aggregate(dev, [], [("count", None, "n"), ("sum", col("one"), "s")], "Single", predicate=in_list_expr(rec)).to_arrow().to_pylist()[0]
    assert n["n"] == kept
