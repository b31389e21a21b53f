"""Operators the reference runs and the device path does not: GpuOffloadRule leaves them to the CPU (round 4).

Every DFGPU_CHECK(... "not supported") on a §8(a)-row path that a legal plan can reach has a plan-time twin in
physical_plan.unsupported_reason — the rule asks BEFORE it substitutes a GPU node, so the query keeps running on the reference's own
operator (which handles these inputs: Decimal256 accumulation, 128-bit comparisons, arrow-row encoded sort keys) instead of
failing at run time.  Each case feeds a plan through the rule and looks at what came out; the same plan with a supported type
right next to it is rewritten as usual."""
from decimal import Decimal

import pyarrow as pa
import pytest

pytestmark = pytest.mark.gpu


def _leaf(table):
    from datafusion_amd import physical_plan as P
    from datafusion_amd.table import DeviceTable
    return P.MemoryExec(DeviceTable.from_arrow(table), "t")


def _wide_table():
    return pa.table({"g": pa.array([1, 2, 1], type=pa.int32()),
                     "wide": pa.array([Decimal("1.00"), Decimal("2.50"), Decimal("3.25")], type=pa.decimal128(30, 2)),     # 30 + 13 > 38
                     "mid": pa.array([Decimal("1.00"), Decimal("2.50"), Decimal("3.25")], type=pa.decimal128(20, 2)),      # > 18 digits declared
                     "ok": pa.array([Decimal("1.00"), Decimal("2.50"), Decimal("3.25")], type=pa.decimal128(15, 2))})


@pytest.mark.parametrize("func, column, needle", [("avg", "wide", "Decimal256"), ("min", "mid", "64-bit"), ("max", "wide", "64-bit")])
@pytest.mark.parametrize("below", ["plain", "under_filter_and_projection"])
def test_aggregates_the_device_cannot_accumulate_stay_on_the_cpu(func, column, needle, below):
    """AVG(Decimal128(p > 25)) accumulates in Decimal256 in the reference (average.rs:131-172); MIN / MAX over Decimal128(p > 18)
    compares 128-bit values (the device compares 64-bit words and would only find out at run time, aggregate.hip:739)"""
    from datafusion_amd import physical_plan as P
    from datafusion_amd.expr import col, lit
    leaf = _leaf(_wide_table())
    inp = leaf
    if below == "under_filter_and_projection":     # the shape the rule would otherwise fuse into one node
        inp = P.ProjectionExec([(col("g"), "g"), (col(column), column), (col("ok"), "ok")], P.FilterExec(col("g") > lit(0, pa.int32()), leaf))
    plan = P.AggregateExec("Single", [(col("g"), "g")], [(func, col(column), "a")], inp)
    rule = P.GpuOffloadRule()
    out = rule.optimize(plan)
    assert isinstance(out, P.AggregateExec) and not isinstance(out, P.GpuFusedAggregateExec) and getattr(out, "kept_on_cpu", False)
    assert len(rule.declined) == 1 and needle in rule.declined[0][1], rule.declined
    # the same aggregate over a column the device does accumulate is offloaded as usual, and runs
    good = P.AggregateExec("Single", [(col("g"), "g")], [(func, col("ok"), "a")], inp)
    rule2 = P.GpuOffloadRule()
    opt = rule2.optimize(good)
    assert not getattr(opt, "kept_on_cpu", False) and not rule2.declined
    assert P.collect(opt).num_rows == 2


def test_partial_aggregate_is_declined_like_the_single_one():
    from datafusion_amd import physical_plan as P
    from datafusion_amd.expr import col
    plan = P.AggregateExec("Partial", [(col("g"), "g")], [("avg", col("wide"), "a"), ("sum", col("ok"), "s")], _leaf(_wide_table()))
    rule = P.GpuOffloadRule()
    out = rule.optimize(plan)
    assert getattr(out, "kept_on_cpu", False) and "Decimal256" in rule.declined[0][1]


def test_a_sort_key_wider_than_the_packed_key_stays_on_the_cpu():
    """sort.hip packs the key columns into at most 192 bits; two Decimal128 columns (2 x 129 bits by type) may not fit"""
    from datafusion_amd import physical_plan as P
    t = pa.table({"a": pa.array([Decimal("3"), Decimal("1")], type=pa.decimal128(38, 0)), "b": pa.array([Decimal("2"), Decimal("2")], type=pa.decimal128(38, 0)),
                  "c": pa.array([5, 6], type=pa.int32())})
    leaf = _leaf(t)
    rule = P.GpuOffloadRule()
    out = rule.optimize(P.SortExec([("a", False, False), ("b", True, False)], leaf))
    assert getattr(out, "kept_on_cpu", False) and "192" in rule.declined[0][1]
    rule2 = P.GpuOffloadRule()
    ok = rule2.optimize(P.SortExec([("a", False, False), ("c", True, False)], leaf))     # 129 + 33 bits: fits, as it is
    assert not getattr(ok, "kept_on_cpu", False) and not rule2.declined
    assert P.collect(ok).to_arrow().column("c").to_pylist() == [6, 5]


def test_plan_schema_follows_the_reference_typing_rules():
    """what the declines are decided on: node output schemas (ExecutionPlan::schema) typed by the library's own expression typing"""
    from datafusion_amd import physical_plan as P
    from datafusion_amd.expr import col, lit
    leaf = _leaf(_wide_table())
    proj = P.ProjectionExec([(col("ok") * col("ok"), "sq"), (col("g"), "g")], P.FilterExec(col("g") > lit(0, pa.int32()), leaf))
    assert P.plan_schema(proj).types == [pa.decimal128(31, 4), pa.int32()]
    agg = P.AggregateExec("Single", [(col("g"), "g")], [("sum", col("sq"), "s"), ("avg", col("sq"), "a"), ("count", None, "c"), ("min", col("g"), "lo")], proj)
    assert P.plan_schema(agg).types == [pa.int32(), pa.decimal128(38, 4), pa.decimal128(35, 8), pa.int64(), pa.int32()]
    part = P.AggregateExec("Partial", [(col("g"), "g")], [("avg", col("ok"), "a")], leaf)
    assert P.plan_schema(part).types == [pa.int32(), pa.uint64(), pa.decimal128(38, 2)]     # avg_sum_data_type: 38 digits
    join = P.HashJoinExec(leaf, proj, [("g", "g")], "RightSemi")
    assert P.plan_schema(join).names == ["sq", "g"]
