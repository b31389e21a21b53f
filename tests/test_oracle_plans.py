"""oracle/plans.py — the partitioned CPU plans of TPC-H Q1 / Q3 that bench.py's cpu_baseline leg times and the full-size GPU tests
(tests/test_gpu_fullsize.py: Q1 SF100, Q3 SF300) compare with — pinned against the reference's own SF0.1 answers
(sqllogictest/test_files/tpch/answers/q1.slt.part:42-45, q3.slt.part:44-53) on dbgen-exact data, for several partition counts: the
Partial -> RepartitionExec(Hash) -> FinalPartitioned split, the partitioned joins and the merge of per-partition top rows must not
change a digit."""
import numpy as np
import pyarrow as pa
import pytest

from tests.test_tpch_answers import assert_answer


@pytest.fixture(scope="module")
def codes_tables():
    """dbgen's SF0.1 tables in the device generator's layout: flags as their ASCII byte, c_mktsegment as a UInt8 code"""
    from oracle import dbgen
    return dbgen.tables(0.1, "codes")


def _letters(t: pa.Table, names):
    for n in names:
        i = t.schema.get_field_index(n)
        t = t.set_column(i, n, pa.array([chr(v) for v in t.column(n).to_pylist()]))
    return t


@pytest.mark.parametrize("P", [1, 3, 8])
def test_partitioned_q1_prints_the_reference_answer(codes_tables, P):
    from oracle import plans
    st = {}
    out = plans.run_q1(codes_tables[2], P, st)
    assert out.schema.field("sum_charge").type == pa.decimal128(38, 6) and out.schema.field("avg_qty").type == pa.decimal128(19, 6)
    assert_answer("q1", _letters(out, ["l_returnflag", "l_linestatus"]))
    assert st["filtered"] == sum(out.column("count_order").to_pylist())


@pytest.mark.parametrize("P", [1, 3, 8])
def test_partitioned_q3_prints_the_reference_answer(codes_tables, P):
    from oracle import plans
    st = {}
    out = plans.run_q3(*codes_tables, P, st)
    assert out.column_names == ["l_orderkey", "revenue", "o_orderdate", "o_shippriority"] and out.schema.field("revenue").type == pa.decimal128(38, 4)
    assert_answer("q3", out)
    assert st["semi_join"] > st["groups"] >= 10 and st["join"] >= st["groups"]
    if P > 1:   # the intermediate row counts do not depend on the partition count
        st1 = {}
        plans.run_q3(*codes_tables, 1, st1)
        assert st1 == st
