"""The ExecutionPlan / PhysicalOptimizerRule surface (datafusion_amd/physical_plan.py): the reference's pinned
TPC-H Q1 and Q3 physical plans (sqllogictest/test_files/tpch/plans/q1.slt.part:50-58, q3.slt.part:61-76) built node
for node, rewritten by GpuOffloadRule, executed on the GPU and compared with the same plan composed from the CPU
oracle's operators — and with the unrewritten plan, whose results the rule must not change (schema_check)."""
import pyarrow as pa
import pyarrow.compute as pc
import pytest

from tests.test_gpu_queries import oracle_q1, oracle_q3
from tests.util import assert_tables_equal

pytestmark = pytest.mark.gpu


def q1_plan(lineitem):
    """q1.slt.part:50-58 — one statement of the plan for the product and the oracle: tests/tpch_plans.py"""
    from tests import tpch_plans as T
    return T.q1_plan(lineitem)


def q3_plan(customer, orders, lineitem):
    """q3.slt.part:61-76 over the device generator's layout (c_mktsegment as a UInt8 code)"""
    from datafusion_amd import queries as Q
    from tests import tpch_plans as T
    from datafusion_amd.expr import lit
    return T.q3_plan(customer, orders, lineitem, segment_literal=lit(Q.SEGMENT_BUILDING, pa.uint8()))


def names(plan):
    return [plan.name()] + [n for c in plan.children() for n in names(c)]


@pytest.mark.parametrize("sf", [0.002, 0.05])
def test_q1_reference_plan_through_the_rule(sf):
    from datafusion_amd import ops, physical_plan as P, tpch
    li = ops.tpch_lineitem(sf)
    plan = q1_plan(li)
    rule = P.GpuOffloadRule()
    assert rule.name() == "gpu_offload_amd" and rule.schema_check()
    opt = rule.optimize(plan)
    ns = names(opt)
    assert ns == ["SortExec", "AggregateExec", "GpuFusedAggregateExec", "MemoryExec"], P.displayable(opt)
    assert names(rule.optimize(opt)) == ns                      # idempotent
    exp = oracle_q1(tpch.lineitem(sf))
    got, plain = P.collect(opt).to_arrow(), P.collect(plan).to_arrow()
    assert_tables_equal(got, exp, ordered=True)
    assert_tables_equal(plain, exp, ordered=True)               # the rule changes the plan, not the result


@pytest.mark.parametrize("sf", [0.002, 0.05])
def test_q3_reference_plan_through_the_rule(sf):
    from datafusion_amd import ops, physical_plan as P, tpch
    c, o, l = ops.tpch_customer(sf), ops.tpch_orders(sf), ops.tpch_lineitem(sf)
    plan = q3_plan(c, o, l)
    opt = P.GpuOffloadRule().optimize(plan)
    ns = names(opt)
    assert ns.count("GpuHashJoinExec") == 2 and "RepartitionExec" not in ns and "CoalesceBatchesExec" not in ns, P.displayable(opt)
    assert ns.count("FilterExec") == 1                          # only the customer filter (a build side) stays a separate operator
    joins = [n for n in _walk(opt) if isinstance(n, P.HashJoinExec)]
    assert all(j.probe_mode == ops.PROBE_MODES["order_not_needed"] for j in joins)   # both feed an aggregate / another join's build side
    exp, _ = oracle_q3(tpch.customer(sf), tpch.orders(sf), tpch.lineitem(sf))
    assert_tables_equal(P.collect(opt).to_arrow(), exp, ordered=True)
    assert_tables_equal(P.collect(plan).to_arrow(), exp, ordered=True)
    assert c.num_rows and o.num_rows and l.num_rows             # leaf tables are never freed by a plan


def _walk(plan):
    yield plan
    for ch in plan.children():
        yield from _walk(ch)


def test_rule_leaves_other_shapes_alone_and_keeps_probe_order_when_observed():
    from datafusion_amd import ops, physical_plan as P
    from datafusion_amd.expr import col, lit
    from datafusion_amd.table import DeviceTable
    from oracle import oracle
    b = pa.table({"k": pa.array([1, 2, 3, 4], type=pa.int64()), "v": pa.array([10, 20, 30, 40], type=pa.int64())})
    p = pa.table({"k2": pa.array([4, 1, 9, 2, 4], type=pa.int64()), "w": pa.array([1, 2, 3, 4, 5], type=pa.int64())})
    bt, pt = DeviceTable.from_arrow(b), DeviceTable.from_arrow(p)
    join = P.HashJoinExec(P.MemoryExec(bt), P.FilterExec(col("w") > lit(1), P.MemoryExec(pt)), [("k", "k2")], "Inner")
    opt = P.GpuOffloadRule().optimize(join)                     # the root's order is observed: ordered (two-pass) probe, filter still fused
    assert isinstance(opt, P.GpuHashJoinExec) and opt.probe_mode == 0
    got = P.collect(opt).to_arrow()
    exp = oracle.hash_join(b, p.filter(pc.greater(p.column("w"), 1)), [("k", "k2")], "Inner")
    assert_tables_equal(got, exp, ordered=True)
    # Final aggregates and build-side filters are not fusion patterns
    fin = P.AggregateExec("Final", [(col("k"), "k")], [("sum", col("v"), "s")], P.FilterExec(col("v") > lit(0), P.MemoryExec(bt)))
    assert names(P.GpuOffloadRule().optimize(fin)) == ["AggregateExec", "FilterExec", "MemoryExec"]
