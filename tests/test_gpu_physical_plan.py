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
    """q1.slt.part:50-58 (4 target partitions in the reference's file; the partition count only shows in RepartitionExec)"""
    from datafusion_amd import physical_plan as P, queries as Q
    from datafusion_amd.expr import col, lit
    scan = P.MemoryExec(lineitem, "lineitem")
    f = P.FilterExec(col("l_shipdate") <= lit(Q.DATE_Q1, pa.date32()), scan,
                     projection=["l_quantity", "l_extendedprice", "l_discount", "l_tax", "l_returnflag", "l_linestatus"])
    proj = P.ProjectionExec([(col("l_extendedprice") * (Q.ONE - col("l_discount")), "__common_expr_1"), (col("l_quantity"), "l_quantity"),
                             (col("l_extendedprice"), "l_extendedprice"), (col("l_discount"), "l_discount"), (col("l_tax"), "l_tax"),
                             (col("l_returnflag"), "l_returnflag"), (col("l_linestatus"), "l_linestatus")], P.CoalesceBatchesExec(f))
    partial = P.AggregateExec("Partial", Q.Q1_GROUP_BY, Q.q1_aggs(), proj)
    rep = P.CoalesceBatchesExec(P.RepartitionExec(partial, ["l_returnflag", "l_linestatus"], 4))
    final = P.AggregateExec("FinalPartitioned", Q.Q1_GROUP_BY, Q.q1_aggs(), rep)
    return P.SortExec([("l_returnflag", False, False), ("l_linestatus", False, False)], final)


def q3_plan(customer, orders, lineitem):
    """q3.slt.part:61-76"""
    from datafusion_amd import physical_plan as P, queries as Q
    from datafusion_amd.expr import col, lit
    c = P.RepartitionExec(P.CoalesceBatchesExec(P.FilterExec(col("c_mktsegment").eq(lit(Q.SEGMENT_BUILDING, pa.uint8())), P.MemoryExec(customer, "customer"),
                                                              projection=["c_custkey"])), ["c_custkey"], 4)
    o = P.RepartitionExec(P.CoalesceBatchesExec(P.FilterExec(col("o_orderdate") < lit(Q.DATE_Q3, pa.date32()), P.MemoryExec(orders, "orders"),
                                                              projection=["o_orderkey", "o_custkey", "o_orderdate", "o_shippriority"])), ["o_custkey"], 4)
    semi = P.HashJoinExec(P.CoalesceBatchesExec(c), P.CoalesceBatchesExec(o), [("c_custkey", "o_custkey")], "RightSemi",
                          projection=(None, ["o_orderkey", "o_orderdate", "o_shippriority"]))
    l = P.RepartitionExec(P.CoalesceBatchesExec(P.FilterExec(col("l_shipdate") > lit(Q.DATE_Q3, pa.date32()), P.MemoryExec(lineitem, "lineitem"),
                                                              projection=["l_orderkey", "l_extendedprice", "l_discount"])), ["l_orderkey"], 4)
    j = P.HashJoinExec(P.CoalesceBatchesExec(P.RepartitionExec(P.CoalesceBatchesExec(semi), ["o_orderkey"], 4)), P.CoalesceBatchesExec(l),
                       [("o_orderkey", "l_orderkey")], "Inner", projection=(["o_orderdate", "o_shippriority"], ["l_orderkey", "l_extendedprice", "l_discount"]))
    gb = [(col("l_orderkey"), "l_orderkey"), (col("o_orderdate"), "o_orderdate"), (col("o_shippriority"), "o_shippriority")]
    agg = P.AggregateExec("SinglePartitioned", gb, [("sum", col("l_extendedprice") * (Q.ONE - col("l_discount")), "revenue")], P.CoalesceBatchesExec(j))
    top = P.SortExec(Q.Q3_SORT, agg, fetch=10)
    return P.ProjectionExec([(col(n), n) for n in ["l_orderkey", "revenue", "o_orderdate", "o_shippriority"]], top)


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
    assert all(j.probe_mode == ops.PROBE_MODES["single_pass_unordered"] for j in joins)   # both feed an aggregate / another join's build side
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
