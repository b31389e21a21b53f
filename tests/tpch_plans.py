"""The reference's pinned TPC-H physical plans, node for node, over this package's ExecutionPlan mirror.

Source of every plan: `datafusion/sqllogictest/test_files/tpch/plans/q{N}.slt.part` (the `physical_plan` block; 4
target partitions in the reference's files — the count only shows in RepartitionExec).  These are what a DataFusion
session hands to a PhysicalOptimizerRule: `GpuOffloadRule.optimize(plan)` rewrites them, `physical_plan.collect` runs
them.  String columns are compared with their string literals, as in the reference's plans; on the device they are
dictionary-encoded (indices on the device, `expr.bind_string_literals`) or, for the device generator's layout, UInt8
codes — `segment_literal` lets Q3 take either.

The leaf tables may be DeviceTables (product) or anything with `num_rows` (plan-shape tests, the oracle's plan
interpreter in tests/plan_oracle.py).
"""
from __future__ import annotations

import datetime
from decimal import Decimal

import pyarrow as pa

from datafusion_amd import physical_plan as P
from datafusion_amd.expr import case, col, date_part, lit

DATE = pa.date32()
D15_2 = pa.decimal128(15, 2)
ONE = lit(1, pa.decimal128(20, 0))  # Int64(1) coerced to Decimal128(20,0), type_coercion/binary.rs:1257-1273
ASC = (False, False)                # ASC NULLS LAST
DESC = (True, True)                 # DESC (NULLS FIRST)


def _d(y, m, d):
    return lit(datetime.date(y, m, d), DATE)


def _scan(table, name):
    return P.MemoryExec(table, name)


def _cb(x):
    return P.CoalesceBatchesExec(x)


def _hash(x, keys):
    return P.RepartitionExec(x, keys, 4)


# ------------------------------------------------------------------------------------------ Q1
def q1_group_by():
    return [(col("l_returnflag"), "l_returnflag"), (col("l_linestatus"), "l_linestatus")]


def q1_aggs():
    ce = col("__common_expr_1")
    return [("sum", col("l_quantity"), "sum_qty"), ("sum", col("l_extendedprice"), "sum_base_price"), ("sum", ce, "sum_disc_price"),
            ("sum", ce * (ONE + col("l_tax")), "sum_charge"), ("avg", col("l_quantity"), "avg_qty"),
            ("avg", col("l_extendedprice"), "avg_price"), ("avg", col("l_discount"), "avg_disc"), ("count", None, "count_order")]


def q1_plan(lineitem):
    """q1.slt.part:50-58"""
    f = P.FilterExec(col("l_shipdate") <= _d(1998, 9, 2), _scan(lineitem, "lineitem"),
                     projection=["l_quantity", "l_extendedprice", "l_discount", "l_tax", "l_returnflag", "l_linestatus"])
    proj = P.ProjectionExec([(col("l_extendedprice") * (ONE - col("l_discount")), "__common_expr_1"), (col("l_quantity"), "l_quantity"),
                             (col("l_extendedprice"), "l_extendedprice"), (col("l_discount"), "l_discount"), (col("l_tax"), "l_tax"),
                             (col("l_returnflag"), "l_returnflag"), (col("l_linestatus"), "l_linestatus")], _cb(f))
    partial = P.AggregateExec("Partial", q1_group_by(), q1_aggs(), proj)
    final = P.AggregateExec("FinalPartitioned", q1_group_by(), q1_aggs(), _cb(_hash(partial, ["l_returnflag", "l_linestatus"])))
    keys = [("l_returnflag",) + ASC, ("l_linestatus",) + ASC]
    return P.SortPreservingMergeExec(keys, P.SortExec(keys, final))


# ------------------------------------------------------------------------------------------ Q3
Q3_SORT = [("revenue",) + DESC, ("o_orderdate",) + ASC]


def q3_plan(customer, orders, lineitem, segment_literal=None):
    """q3.slt.part:61-76.  segment_literal: the literal `c_mktsegment` is compared with (default: the string
    'BUILDING' for a string / dictionary column; the device generator's UInt8 code layout passes lit(1, uint8))"""
    seg = lit("BUILDING", pa.string()) if segment_literal is None else segment_literal
    c = _hash(_cb(P.FilterExec(col("c_mktsegment").eq(seg), _scan(customer, "customer"), projection=["c_custkey"])), ["c_custkey"])
    o = _hash(_cb(P.FilterExec(col("o_orderdate") < _d(1995, 3, 15), _scan(orders, "orders"),
                               projection=["o_orderkey", "o_custkey", "o_orderdate", "o_shippriority"])), ["o_custkey"])
    semi = P.HashJoinExec(_cb(c), _cb(o), [("c_custkey", "o_custkey")], "RightSemi", projection=(None, ["o_orderkey", "o_orderdate", "o_shippriority"]))
    l = _hash(_cb(P.FilterExec(col("l_shipdate") > _d(1995, 3, 15), _scan(lineitem, "lineitem"),
                               projection=["l_orderkey", "l_extendedprice", "l_discount"])), ["l_orderkey"])
    j = P.HashJoinExec(_cb(_hash(_cb(semi), ["o_orderkey"])), _cb(l), [("o_orderkey", "l_orderkey")], "Inner",
                       projection=(["o_orderdate", "o_shippriority"], ["l_orderkey", "l_extendedprice", "l_discount"]))
    gb = [(col("l_orderkey"), "l_orderkey"), (col("o_orderdate"), "o_orderdate"), (col("o_shippriority"), "o_shippriority")]
    agg = P.AggregateExec("SinglePartitioned", gb, [("sum", col("l_extendedprice") * (ONE - col("l_discount")), "revenue")], _cb(j))
    top = P.SortExec(Q3_SORT, agg, fetch=10)
    proj = P.ProjectionExec([(col(n), n) for n in ["l_orderkey", "revenue", "o_orderdate", "o_shippriority"]], top)
    return P.SortPreservingMergeExec(Q3_SORT, proj, fetch=10)


# ------------------------------------------------------------------------------------------ Q4
def q4_plan(orders, lineitem):
    """q4.slt.part:61-73: EXISTS subquery decorrelated to a LeftSemi join (build = filtered orders)"""
    o = _hash(_cb(P.FilterExec((col("o_orderdate") >= _d(1993, 7, 1)).and_(col("o_orderdate") < _d(1993, 10, 1)), _scan(orders, "orders"),
                               projection=["o_orderkey", "o_orderpriority"])), ["o_orderkey"])
    l = _hash(_cb(P.FilterExec(col("l_receiptdate") > col("l_commitdate"), _scan(lineitem, "lineitem"), projection=["l_orderkey"])), ["l_orderkey"])
    semi = P.HashJoinExec(_cb(o), _cb(l), [("o_orderkey", "l_orderkey")], "LeftSemi", projection=(["o_orderpriority"], None))
    gb = [(col("o_orderpriority"), "o_orderpriority")]
    aggs = [("count", None, "count(Int64(1))")]
    partial = P.AggregateExec("Partial", gb, aggs, _cb(semi))
    final = P.AggregateExec("FinalPartitioned", gb, aggs, _cb(_hash(partial, ["o_orderpriority"])))
    keys = [("o_orderpriority",) + ASC]
    proj = P.ProjectionExec([(col("o_orderpriority"), "o_orderpriority"), (col("count(Int64(1))"), "order_count")], P.SortExec(keys, final))
    return P.SortPreservingMergeExec(keys, proj)


# ------------------------------------------------------------------------------------------ Q5
def q5_plan(customer, orders, lineitem, supplier, nation, region):
    """q5.slt.part:70-99: five joins (one on two key columns), LeftSemi against the filtered region"""
    c = _hash(_scan(customer, "customer").project(["c_custkey", "c_nationkey"]), ["c_custkey"])
    o = _hash(_cb(P.FilterExec((col("o_orderdate") >= _d(1994, 1, 1)).and_(col("o_orderdate") < _d(1995, 1, 1)), _scan(orders, "orders"),
                               projection=["o_orderkey", "o_custkey"])), ["o_custkey"])
    j1 = P.HashJoinExec(_cb(c), _cb(o), [("c_custkey", "o_custkey")], "Inner", projection=(["c_nationkey"], ["o_orderkey"]))
    l = _hash(_scan(lineitem, "lineitem").project(["l_orderkey", "l_suppkey", "l_extendedprice", "l_discount"]), ["l_orderkey"])
    j2 = P.HashJoinExec(_cb(_hash(_cb(j1), ["o_orderkey"])), _cb(l), [("o_orderkey", "l_orderkey")], "Inner",
                        projection=(["c_nationkey"], ["l_suppkey", "l_extendedprice", "l_discount"]))
    s = _hash(_scan(supplier, "supplier").project(["s_suppkey", "s_nationkey"]), ["s_suppkey", "s_nationkey"])
    j3 = P.HashJoinExec(_cb(_hash(_cb(j2), ["l_suppkey", "c_nationkey"])), _cb(s), [("l_suppkey", "s_suppkey"), ("c_nationkey", "s_nationkey")], "Inner",
                        projection=(["l_extendedprice", "l_discount"], ["s_nationkey"]))
    n = _hash(_scan(nation, "nation").project(["n_nationkey", "n_name", "n_regionkey"]), ["n_nationkey"])
    j4 = P.HashJoinExec(_cb(_hash(_cb(j3), ["s_nationkey"])), _cb(n), [("s_nationkey", "n_nationkey")], "Inner",
                        projection=(["l_extendedprice", "l_discount"], ["n_name", "n_regionkey"]))
    r = _hash(_cb(P.FilterExec(col("r_name").eq(lit("ASIA", pa.string())), _scan(region, "region"), projection=["r_regionkey"])), ["r_regionkey"])
    semi = P.HashJoinExec(_cb(_hash(_cb(j4), ["n_regionkey"])), _cb(r), [("n_regionkey", "r_regionkey")], "LeftSemi",
                          projection=(["l_extendedprice", "l_discount", "n_name"], None))
    gb = [(col("n_name"), "n_name")]
    name = "sum(lineitem.l_extendedprice * Int64(1) - lineitem.l_discount)"
    aggs = [("sum", col("l_extendedprice") * (ONE - col("l_discount")), name)]
    partial = P.AggregateExec("Partial", gb, aggs, _cb(semi))
    final = P.AggregateExec("FinalPartitioned", gb, aggs, _cb(_hash(partial, ["n_name"])))
    srt = P.SortExec([(name,) + DESC], final)
    proj = P.ProjectionExec([(col("n_name"), "n_name"), (col(name), "revenue")], srt)
    return P.SortPreservingMergeExec([("revenue",) + DESC], proj)


# ------------------------------------------------------------------------------------------ Q7
def q7_plan(supplier, lineitem, orders, customer, nation):
    """q7.slt.part:87-119: five Inner joins, the last one with a JoinFilter over the two nation names
    (n_name@0 = FRANCE AND n_name@1 = GERMANY OR n_name@0 = GERMANY AND n_name@1 = FRANCE), date_part(YEAR, l_shipdate) as a group key.
    This mirror addresses columns by name where the reference's plan uses indices, so the second nation scan carries an aliasing
    ProjectionExec (n_nationkey / n_name -> n2_nationkey / n2_name) that the index-addressed plan does not need."""
    s_ = lambda v: lit(v, pa.string())                                     # noqa: E731
    sup = _hash(_scan(supplier, "supplier").project(["s_suppkey", "s_nationkey"]), ["s_suppkey"])
    li = _hash(_cb(P.FilterExec((col("l_shipdate") >= _d(1995, 1, 1)).and_(col("l_shipdate") <= _d(1996, 12, 31)),
                                _scan(lineitem, "lineitem").project(["l_orderkey", "l_suppkey", "l_extendedprice", "l_discount", "l_shipdate"]))), ["l_suppkey"])
    j1 = P.HashJoinExec(_cb(sup), _cb(li), [("s_suppkey", "l_suppkey")], "Inner",
                        projection=(["s_nationkey"], ["l_orderkey", "l_extendedprice", "l_discount", "l_shipdate"]))
    o = _hash(_scan(orders, "orders").project(["o_orderkey", "o_custkey"]), ["o_orderkey"])
    j2 = P.HashJoinExec(_cb(_hash(_cb(j1), ["l_orderkey"])), _cb(o), [("l_orderkey", "o_orderkey")], "Inner",
                        projection=(["s_nationkey", "l_extendedprice", "l_discount", "l_shipdate"], ["o_custkey"]))
    c = _hash(_scan(customer, "customer").project(["c_custkey", "c_nationkey"]), ["c_custkey"])
    j3 = P.HashJoinExec(_cb(_hash(_cb(j2), ["o_custkey"])), _cb(c), [("o_custkey", "c_custkey")], "Inner",
                        projection=(["s_nationkey", "l_extendedprice", "l_discount", "l_shipdate"], ["c_nationkey"]))
    n1 = _hash(_cb(P.FilterExec(col("n_name").eq(s_("FRANCE")).or_(col("n_name").eq(s_("GERMANY"))), _scan(nation, "nation").project(["n_nationkey", "n_name"]))),
               ["n_nationkey"])
    j4 = P.HashJoinExec(_cb(_hash(_cb(j3), ["s_nationkey"])), _cb(n1), [("s_nationkey", "n_nationkey")], "Inner",
                        projection=(["l_extendedprice", "l_discount", "l_shipdate", "c_nationkey"], ["n_name"]))
    n2scan = P.ProjectionExec([(col("n_nationkey"), "n2_nationkey"), (col("n_name"), "n2_name")], _scan(nation, "nation").project(["n_nationkey", "n_name"]))
    n2 = _hash(_cb(P.FilterExec(col("n2_name").eq(s_("GERMANY")).or_(col("n2_name").eq(s_("FRANCE"))), n2scan)), ["n2_nationkey"])
    # JoinFilter over the intermediate columns f0 = n_name (Left 4), f1 = n2_name (Right 1)
    f0, f1 = col("f0"), col("f1")
    jf = (f0.eq(s_("FRANCE")).and_(f1.eq(s_("GERMANY")))).or_(f0.eq(s_("GERMANY")).and_(f1.eq(s_("FRANCE"))))
    j5 = P.HashJoinExec(_cb(_hash(_cb(j4), ["c_nationkey"])), _cb(n2), [("c_nationkey", "n2_nationkey")], "Inner",
                        projection=(["n_name", "l_shipdate", "l_extendedprice", "l_discount"], ["n2_name"]), filter=(jf, [(4, "Left"), (1, "Right")]))
    proj = P.ProjectionExec([(col("n_name"), "supp_nation"), (col("n2_name"), "cust_nation"), (date_part("year", col("l_shipdate")), "l_year"),
                             (col("l_extendedprice") * (ONE - col("l_discount")), "volume")], _cb(j5))
    gb = [(col("supp_nation"), "supp_nation"), (col("cust_nation"), "cust_nation"), (col("l_year"), "l_year")]
    aggs = [("sum", col("volume"), "sum(shipping.volume)")]
    partial = P.AggregateExec("Partial", gb, aggs, proj)
    final = P.AggregateExec("FinalPartitioned", gb, aggs, _cb(_hash(partial, ["supp_nation", "cust_nation", "l_year"])))
    keys = [("supp_nation",) + ASC, ("cust_nation",) + ASC, ("l_year",) + ASC]
    out = P.ProjectionExec([(col("supp_nation"), "supp_nation"), (col("cust_nation"), "cust_nation"), (col("l_year"), "l_year"), (col("sum(shipping.volume)"), "revenue")],
                           P.SortExec(keys, final))
    return P.SortPreservingMergeExec(keys, out)


# ------------------------------------------------------------------------------------------ Q8
def q8_plan(part, supplier, lineitem, orders, customer, nation, region):
    """q8.slt.part:93-132: RightSemi against the filtered part, five Inner joins, LeftSemi against the filtered region,
    date_part(YEAR, o_orderdate) as the group key, SUM(CASE WHEN nation = 'BRAZIL' ...) and the final
    CAST(CAST(a AS Decimal128(12, 2)) / CAST(b AS Decimal128(12, 2)) AS Decimal128(15, 2)): two scale-down casts around arrow-arith's
    decimal division (Decimal128(12,2) / Decimal128(12,2) -> Decimal128(18,6))"""
    s_ = lambda v: lit(v, pa.string())                                     # noqa: E731
    p = _hash(_cb(P.FilterExec(col("p_type").eq(s_("ECONOMY ANODIZED STEEL")), _scan(part, "part").project(["p_partkey", "p_type"]), projection=["p_partkey"])), ["p_partkey"])
    li = _hash(_scan(lineitem, "lineitem").project(["l_orderkey", "l_partkey", "l_suppkey", "l_extendedprice", "l_discount"]), ["l_partkey"])
    j1 = P.HashJoinExec(_cb(p), _cb(li), [("p_partkey", "l_partkey")], "RightSemi", projection=(None, ["l_orderkey", "l_suppkey", "l_extendedprice", "l_discount"]))
    sup = _hash(_scan(supplier, "supplier").project(["s_suppkey", "s_nationkey"]), ["s_suppkey"])
    j2 = P.HashJoinExec(_cb(_hash(_cb(j1), ["l_suppkey"])), _cb(sup), [("l_suppkey", "s_suppkey")], "Inner",
                        projection=(["l_orderkey", "l_extendedprice", "l_discount"], ["s_nationkey"]))
    o = _hash(_cb(P.FilterExec((col("o_orderdate") >= _d(1995, 1, 1)).and_(col("o_orderdate") <= _d(1996, 12, 31)),
                               _scan(orders, "orders").project(["o_orderkey", "o_custkey", "o_orderdate"]))), ["o_orderkey"])
    j3 = P.HashJoinExec(_cb(_hash(_cb(j2), ["l_orderkey"])), _cb(o), [("l_orderkey", "o_orderkey")], "Inner",
                        projection=(["l_extendedprice", "l_discount", "s_nationkey"], ["o_custkey", "o_orderdate"]))
    c = _hash(_scan(customer, "customer").project(["c_custkey", "c_nationkey"]), ["c_custkey"])
    j4 = P.HashJoinExec(_cb(_hash(_cb(j3), ["o_custkey"])), _cb(c), [("o_custkey", "c_custkey")], "Inner",
                        projection=(["l_extendedprice", "l_discount", "s_nationkey", "o_orderdate"], ["c_nationkey"]))
    n1 = _hash(_scan(nation, "nation").project(["n_nationkey", "n_regionkey"]), ["n_nationkey"])
    j5 = P.HashJoinExec(_cb(_hash(_cb(j4), ["c_nationkey"])), _cb(n1), [("c_nationkey", "n_nationkey")], "Inner",
                        projection=(["l_extendedprice", "l_discount", "s_nationkey", "o_orderdate"], ["n_regionkey"]))
    n2 = _hash(_scan(nation, "nation").project(["n_nationkey", "n_name"]), ["n_nationkey"])
    j6 = P.HashJoinExec(_cb(_hash(_cb(j5), ["s_nationkey"])), _cb(n2), [("s_nationkey", "n_nationkey")], "Inner",
                        projection=(["l_extendedprice", "l_discount", "o_orderdate", "n_regionkey"], ["n_name"]))
    r = _hash(_cb(P.FilterExec(col("r_name").eq(s_("AMERICA")), _scan(region, "region"), projection=["r_regionkey"])), ["r_regionkey"])
    semi = P.HashJoinExec(_cb(_hash(_cb(j6), ["n_regionkey"])), _cb(r), [("n_regionkey", "r_regionkey")], "LeftSemi",
                          projection=(["o_orderdate", "l_extendedprice", "l_discount", "n_name"], None))
    proj = P.ProjectionExec([(date_part("year", col("o_orderdate")), "o_year"), (col("l_extendedprice") * (ONE - col("l_discount")), "volume"), (col("n_name"), "nation")], _cb(semi))
    a_name = 'sum(CASE WHEN all_nations.nation = Utf8("BRAZIL") THEN all_nations.volume ELSE Int64(0) END)'
    b_name = "sum(all_nations.volume)"
    zero = lit(Decimal("0.0000"), pa.decimal128(38, 4))
    gb = [(col("o_year"), "o_year")]
    aggs = [("sum", case([(col("nation").eq(s_("BRAZIL")), col("volume"))], zero), a_name), ("sum", col("volume"), b_name)]
    partial = P.AggregateExec("Partial", gb, aggs, proj)
    final = P.AggregateExec("FinalPartitioned", gb, aggs, _cb(_hash(partial, ["o_year"])))
    d12, d15 = pa.decimal128(12, 2), pa.decimal128(15, 2)
    share = P.ProjectionExec([(col("o_year"), "o_year"), ((col(a_name).cast(d12) / col(b_name).cast(d12)).cast(d15), "mkt_share")], final)
    keys = [("o_year",) + ASC]
    return P.SortPreservingMergeExec(keys, P.SortExec(keys, share))


# ----------------------------------------------------------------------------------------- Q14
def q14_plan(lineitem, part):
    """q14.slt.part:39-49: Inner join lineitem x part, SUM(CASE WHEN p_type LIKE 'PROMO%' THEN ... ELSE 0.0000 END) and the final
    Float64 division 100 * CAST(a AS Float64) / CAST(b AS Float64)"""
    li = _hash(_cb(P.FilterExec((col("l_shipdate") >= _d(1995, 9, 1)).and_(col("l_shipdate") < _d(1995, 10, 1)),
                                _scan(lineitem, "lineitem").project(["l_partkey", "l_extendedprice", "l_discount", "l_shipdate"]),
                                projection=["l_partkey", "l_extendedprice", "l_discount"])), ["l_partkey"])
    p = _hash(_scan(part, "part").project(["p_partkey", "p_type"]), ["p_partkey"])
    j = P.HashJoinExec(_cb(li), _cb(p), [("l_partkey", "p_partkey")], "Inner", projection=(["l_extendedprice", "l_discount"], ["p_type"]))
    proj = P.ProjectionExec([(col("l_extendedprice") * (ONE - col("l_discount")), "__common_expr_1"), (col("p_type"), "p_type")], _cb(j))
    a_name = 'sum(CASE WHEN part.p_type LIKE Utf8("PROMO%") THEN lineitem.l_extendedprice * Int64(1) - lineitem.l_discount ELSE Int64(0) END)'
    b_name = "sum(lineitem.l_extendedprice * Int64(1) - lineitem.l_discount)"
    zero = lit(Decimal("0.0000"), pa.decimal128(38, 4))
    aggs = [("sum", case([(col("p_type").like("PROMO%"), col("__common_expr_1"))], zero), a_name), ("sum", col("__common_expr_1"), b_name)]
    partial = P.AggregateExec("Partial", [], aggs, proj)
    final = P.AggregateExec("Final", [], aggs, P.CoalescePartitionsExec(partial))
    f64 = pa.float64()
    return P.ProjectionExec([(lit(100.0, f64) * col(a_name).cast(f64) / col(b_name).cast(f64), "promo_revenue")], final)


# ----------------------------------------------------------------------------------------- Q17
def q17_plan(lineitem, part):
    """q17.slt.part:54-69: the correlated scalar subquery decorrelated to a LeftSemi join whose JoinFilter compares
    CAST(l_quantity AS Decimal128(30, 15)) with CAST(0.2 * CAST(avg(l_quantity) AS Float64) AS Decimal128(30, 15)) per part.
    (The subquery's key column is aliased l_partkey2: this mirror addresses columns by name, the reference's plan by index.)"""
    f64, d30 = pa.float64(), pa.decimal128(30, 15)
    li = _hash(_scan(lineitem, "lineitem").project(["l_partkey", "l_quantity", "l_extendedprice"]), ["l_partkey"])
    p = _hash(_cb(P.FilterExec(col("p_brand").eq(lit("Brand#23", pa.string())).and_(col("p_container").eq(lit("MED BOX", pa.string()))),
                               _scan(part, "part").project(["p_partkey", "p_brand", "p_container"]), projection=["p_partkey"])), ["p_partkey"])
    j1 = P.HashJoinExec(_cb(li), _cb(p), [("l_partkey", "p_partkey")], "Inner", projection=(["l_quantity", "l_extendedprice"], ["p_partkey"]))
    gb = [(col("l_partkey"), "l_partkey")]
    aggs = [("avg", col("l_quantity"), "avg(lineitem.l_quantity)")]
    partial = P.AggregateExec("Partial", gb, aggs, _scan(lineitem, "lineitem").project(["l_partkey", "l_quantity"]))
    final = P.AggregateExec("FinalPartitioned", gb, aggs, _cb(_hash(partial, ["l_partkey"])))
    name = "Float64(0.2) * avg(lineitem.l_quantity)"
    thr = P.ProjectionExec([((lit(0.2, f64) * col("avg(lineitem.l_quantity)").cast(f64)).cast(d30), name), (col("l_partkey"), "l_partkey2")], final)
    # JoinFilter over the intermediate columns f0 = l_quantity (Left 0), f1 = the threshold (Right 0)
    jf = (col("f0").cast(d30) < col("f1"), [(0, "Left"), (0, "Right")])
    semi = P.HashJoinExec(_cb(j1), _cb(thr), [("p_partkey", "l_partkey2")], "LeftSemi", projection=(["l_extendedprice"], None), filter=jf)
    a = [("sum", col("l_extendedprice"), "sum(lineitem.l_extendedprice)")]
    fin = P.AggregateExec("Final", [], a, P.CoalescePartitionsExec(P.AggregateExec("Partial", [], a, _cb(semi))))
    return P.ProjectionExec([(col("sum(lineitem.l_extendedprice)").cast(f64) / lit(7.0, f64), "avg_yearly")], fin)


# ----------------------------------------------------------------------------------------- Q11
def q11_plan(partsupp, supplier, nation):
    """q11.slt.part:75-108: partsupp (build) x supplier, LeftSemi against the GERMANY row of nation, SUM(ps_supplycost *
    CAST(ps_availqty AS Decimal128(10, 0))) per part, HAVING it above 0.0001 of the same sum over all parts — an uncorrelated scalar
    subquery: ScalarSubqueryExec runs it first, the FilterExec reads it as CAST(CAST(sum AS Float64) * 0.0001 AS Decimal128(38, 15))"""
    from datafusion_amd.expr import ScalarSubqueryExpr, ScalarSubqueryResults
    f64, d38 = pa.float64(), pa.decimal128(38, 15)
    name = "sum(partsupp.ps_supplycost * partsupp.ps_availqty)"
    value = col("ps_supplycost") * col("ps_availqty").cast(pa.decimal128(10, 0))

    def german_parts(with_partkey):
        pcols = (["ps_partkey"] if with_partkey else []) + ["ps_availqty", "ps_supplycost"]
        ps = _hash(_scan(partsupp, "partsupp").project((["ps_partkey"] if with_partkey else []) + ["ps_suppkey", "ps_availqty", "ps_supplycost"]), ["ps_suppkey"])
        su = _hash(_scan(supplier, "supplier").project(["s_suppkey", "s_nationkey"]), ["s_suppkey"])
        j = P.HashJoinExec(_cb(ps), _cb(su), [("ps_suppkey", "s_suppkey")], "Inner", projection=(pcols, ["s_nationkey"]))
        n = _hash(_cb(P.FilterExec(col("n_name").eq(lit("GERMANY", pa.string())), _scan(nation, "nation").project(["n_nationkey", "n_name"]),
                                   projection=["n_nationkey"])), ["n_nationkey"])
        return P.HashJoinExec(_cb(_hash(_cb(j), ["s_nationkey"])), _cb(n), [("s_nationkey", "n_nationkey")], "LeftSemi", projection=(pcols, None))

    aggs = [("sum", value, name)]
    # the subquery: the ungrouped sum, scaled
    sub = P.AggregateExec("Final", [], aggs, P.CoalescePartitionsExec(P.AggregateExec("Partial", [], aggs, _cb(german_parts(False)))))
    sub = P.ProjectionExec([((col(name).cast(f64) * lit(0.0001, f64)).cast(d38), name + " * Float64(0.0001)")], sub)
    results = ScalarSubqueryResults(1)
    # the main plan
    gb = [(col("ps_partkey"), "ps_partkey")]
    partial = P.AggregateExec("Partial", gb, aggs, _cb(german_parts(True)))
    final = P.AggregateExec("FinalPartitioned", gb, aggs, _cb(_hash(partial, ["ps_partkey"])))
    having = P.FilterExec(col(name).cast(d38) > ScalarSubqueryExpr(results, 0, d38), final)
    keys = [(name,) + DESC]
    top = P.ProjectionExec([(col("ps_partkey"), "ps_partkey"), (col(name), "value")], P.SortExec(keys, _cb(having), fetch=10))
    return P.ScalarSubqueryExec(P.SortPreservingMergeExec([("value",) + DESC], top, fetch=10), [(sub, 0)], results)


# ------------------------------------------------------------------------------------------ Q9
def q9_plan(part, supplier, lineitem, partsupp, orders, nation):
    """q9.slt.part:77-104: RightSemi against the parts named LIKE '%green%', then four Inner joins in which the running result is
    always the BUILD side (duplicate keys: l_suppkey, then the two-column (l_suppkey, l_partkey), l_orderkey, s_nationkey), profit =
    l_extendedprice * (1 - l_discount) - ps_supplycost * l_quantity per nation and date_part(YEAR, o_orderdate), top 10"""
    p = _hash(_cb(P.FilterExec(col("p_name").like("%green%"), _scan(part, "part").project(["p_partkey", "p_name"]), projection=["p_partkey"])), ["p_partkey"])
    lcols = ["l_orderkey", "l_partkey", "l_suppkey", "l_quantity", "l_extendedprice", "l_discount"]
    li = _hash(_scan(lineitem, "lineitem").project(lcols), ["l_partkey"])
    j1 = P.HashJoinExec(_cb(p), _cb(li), [("p_partkey", "l_partkey")], "RightSemi", projection=(None, lcols))
    su = _hash(_scan(supplier, "supplier").project(["s_suppkey", "s_nationkey"]), ["s_suppkey"])
    j2 = P.HashJoinExec(_cb(_hash(_cb(j1), ["l_suppkey"])), _cb(su), [("l_suppkey", "s_suppkey")], "Inner", projection=(lcols, ["s_nationkey"]))
    ps = _hash(_scan(partsupp, "partsupp").project(["ps_partkey", "ps_suppkey", "ps_supplycost"]), ["ps_suppkey", "ps_partkey"])
    j3 = P.HashJoinExec(_cb(_hash(_cb(j2), ["l_suppkey", "l_partkey"])), _cb(ps), [("l_suppkey", "ps_suppkey"), ("l_partkey", "ps_partkey")], "Inner",
                        projection=(["l_orderkey", "l_quantity", "l_extendedprice", "l_discount", "s_nationkey"], ["ps_supplycost"]))
    o = _hash(_scan(orders, "orders").project(["o_orderkey", "o_orderdate"]), ["o_orderkey"])
    j4 = P.HashJoinExec(_cb(_hash(_cb(j3), ["l_orderkey"])), _cb(o), [("l_orderkey", "o_orderkey")], "Inner",
                        projection=(["l_quantity", "l_extendedprice", "l_discount", "s_nationkey", "ps_supplycost"], ["o_orderdate"]))
    n = _hash(_scan(nation, "nation").project(["n_nationkey", "n_name"]), ["n_nationkey"])
    j5 = P.HashJoinExec(_cb(_hash(_cb(j4), ["s_nationkey"])), _cb(n), [("s_nationkey", "n_nationkey")], "Inner",
                        projection=(["l_quantity", "l_extendedprice", "l_discount", "ps_supplycost", "o_orderdate"], ["n_name"]))
    amount = col("l_extendedprice") * (ONE - col("l_discount")) - col("ps_supplycost") * col("l_quantity")
    proj = P.ProjectionExec([(col("n_name"), "nation"), (date_part("year", col("o_orderdate")), "o_year"), (amount, "amount")], _cb(j5))
    gb = [(col("nation"), "nation"), (col("o_year"), "o_year")]
    aggs = [("sum", col("amount"), "sum(profit.amount)")]
    partial = P.AggregateExec("Partial", gb, aggs, proj)
    final = P.AggregateExec("FinalPartitioned", gb, aggs, _cb(_hash(partial, ["nation", "o_year"])))
    keys = [("nation",) + ASC, ("o_year",) + DESC]
    top = P.ProjectionExec([(col("nation"), "nation"), (col("o_year"), "o_year"), (col("sum(profit.amount)"), "sum_profit")], P.SortExec(keys, final, fetch=10))
    return P.SortPreservingMergeExec(keys, top, fetch=10)


# ----------------------------------------------------------------------------------------- Q20
def q20_plan(supplier, nation, partsupp, part, lineitem):
    """q20.slt.part:84-111: suppliers of CANADA (LeftSemi against nation) that supply (LeftSemi) a part named LIKE 'forest%' whose
    ps_availqty exceeds half of what was shipped of it by that supplier in 1994 — a LeftSemi join on (partkey, suppkey) with the
    JoinFilter CAST(ps_availqty AS Float64) > 0.5 * sum(l_quantity)"""
    f64 = pa.float64()
    su = _hash(_scan(supplier, "supplier").project(["s_suppkey", "s_name", "s_address", "s_nationkey"]), ["s_nationkey"])
    n = _hash(_cb(P.FilterExec(col("n_name").eq(lit("CANADA", pa.string())), _scan(nation, "nation").project(["n_nationkey", "n_name"]), projection=["n_nationkey"])),
              ["n_nationkey"])
    canada = P.HashJoinExec(_cb(su), _cb(n), [("s_nationkey", "n_nationkey")], "LeftSemi", projection=(["s_suppkey", "s_name", "s_address"], None))
    ps = _hash(_scan(partsupp, "partsupp").project(["ps_partkey", "ps_suppkey", "ps_availqty"]), ["ps_partkey"])
    p = _hash(_cb(P.FilterExec(col("p_name").like("forest%"), _scan(part, "part").project(["p_partkey", "p_name"]), projection=["p_partkey"])), ["p_partkey"])
    forest = P.HashJoinExec(_cb(ps), _cb(p), [("ps_partkey", "p_partkey")], "LeftSemi")
    pred = (col("l_shipdate") >= _d(1994, 1, 1)).and_(col("l_shipdate") < _d(1995, 1, 1))
    lf = P.FilterExec(pred, _scan(lineitem, "lineitem").project(["l_partkey", "l_suppkey", "l_quantity", "l_shipdate"]), projection=["l_partkey", "l_suppkey", "l_quantity"])
    gb = [(col("l_partkey"), "l_partkey"), (col("l_suppkey"), "l_suppkey")]
    aggs = [("sum", col("l_quantity"), "sum(lineitem.l_quantity)")]
    shipped = P.AggregateExec("FinalPartitioned", gb, aggs, _cb(_hash(P.AggregateExec("Partial", gb, aggs, _cb(lf)), ["l_partkey", "l_suppkey"])))
    half_name = "Float64(0.5) * sum(lineitem.l_quantity)"
    half = P.ProjectionExec([(lit(0.5, f64) * col("sum(lineitem.l_quantity)").cast(f64), half_name), (col("l_partkey"), "l_partkey"), (col("l_suppkey"), "l_suppkey")], shipped)
    # JoinFilter over the intermediate columns f0 = ps_availqty (Left 2), f1 = the half of the shipped quantity (Right 0)
    jf = (col("f0").cast(f64) > col("f1"), [(2, "Left"), (0, "Right")])
    excess = P.HashJoinExec(_cb(_hash(_cb(forest), ["ps_partkey", "ps_suppkey"])), _cb(half), [("ps_partkey", "l_partkey"), ("ps_suppkey", "l_suppkey")], "LeftSemi",
                            projection=(["ps_suppkey"], None), filter=jf)
    out = P.HashJoinExec(_cb(_hash(_cb(canada), ["s_suppkey"])), _cb(_hash(_cb(excess), ["ps_suppkey"])), [("s_suppkey", "ps_suppkey")], "LeftSemi",
                         projection=(["s_name", "s_address"], None))
    keys = [("s_name",) + ASC]
    return P.SortPreservingMergeExec(keys, P.SortExec(keys, _cb(out)))


# ----------------------------------------------------------------------------------------- Q15
def q15_plan(supplier, lineitem):
    """q15.slt.part:73-94: the revenue0 view (revenue per supplier over one quarter) twice — once under MAX as an uncorrelated scalar
    subquery, once filtered to the suppliers whose revenue EQUALS it — joined to supplier (build side, string payload)"""
    from datafusion_amd.expr import ScalarSubqueryExpr, ScalarSubqueryResults
    name = "sum(lineitem.l_extendedprice * Int64(1) - lineitem.l_discount)"
    d38 = pa.decimal128(38, 4)

    def revenue0():
        pred = (col("l_shipdate") >= _d(1996, 1, 1)).and_(col("l_shipdate") < _d(1996, 4, 1))
        f = P.FilterExec(pred, _scan(lineitem, "lineitem"), projection=["l_suppkey", "l_extendedprice", "l_discount"])
        gb = [(col("l_suppkey"), "l_suppkey")]
        aggs = [("sum", col("l_extendedprice") * (ONE - col("l_discount")), name)]
        return P.AggregateExec("FinalPartitioned", gb, aggs, _cb(_hash(P.AggregateExec("Partial", gb, aggs, _cb(f)), ["l_suppkey"])))

    mx = [("max", col("total_revenue"), "max(revenue0.total_revenue)")]
    sub = P.AggregateExec("Final", [], mx, P.CoalescePartitionsExec(P.AggregateExec("Partial", [], mx, P.ProjectionExec([(col(name), "total_revenue")], revenue0()))))
    results = ScalarSubqueryResults(1)
    best = P.ProjectionExec([(col("l_suppkey"), "supplier_no"), (col(name), "total_revenue")],
                            _cb(P.FilterExec(col(name).eq(ScalarSubqueryExpr(results, 0, d38)), revenue0())))
    su = _hash(_scan(supplier, "supplier").project(["s_suppkey", "s_name", "s_address", "s_phone"]), ["s_suppkey"])
    j = P.HashJoinExec(_cb(su), _cb(best), [("s_suppkey", "supplier_no")], "Inner", projection=(["s_suppkey", "s_name", "s_address", "s_phone"], ["total_revenue"]))
    keys = [("s_suppkey",) + ASC]
    return P.ScalarSubqueryExec(P.SortPreservingMergeExec(keys, P.SortExec(keys, _cb(j))), [(sub, 0)], results)


# ----------------------------------------------------------------------------------------- Q22
def q22_plan(customer, orders):
    """q22.slt.part:76-95: customers of seven country codes (substr(c_phone, 1, 2)) whose balance is above the average positive
    balance of those countries (uncorrelated scalar subquery) and who have no orders (LeftAnti: the filtered customers are the
    build side), counted and summed per country code"""
    from datafusion_amd.expr import ScalarSubqueryExpr, ScalarSubqueryResults, substr
    codes = [lit(c, pa.string()) for c in ("13", "31", "23", "29", "30", "18", "17")]
    cc = substr(col("c_phone"), 1, 2)
    d19 = pa.decimal128(19, 6)
    # the subquery: avg(c_acctbal) over the positive balances of those countries
    avg_name = "avg(customer.c_acctbal)"
    a = [("avg", col("c_acctbal"), avg_name)]
    sub_f = P.FilterExec((col("c_acctbal") > lit(Decimal("0.00"), D15_2)).and_(cc.in_list(codes)), _scan(customer, "customer").project(["c_phone", "c_acctbal"]),
                         projection=["c_acctbal"])
    sub = P.AggregateExec("Final", [], a, P.CoalescePartitionsExec(P.AggregateExec("Partial", [], a, _cb(sub_f))))
    results = ScalarSubqueryResults(1)
    # the main plan
    f = P.FilterExec(cc.in_list(codes).and_(col("c_acctbal").cast(d19) > ScalarSubqueryExpr(results, 0, d19)),
                     _scan(customer, "customer").project(["c_custkey", "c_phone", "c_acctbal"]))
    anti = P.HashJoinExec(_cb(_hash(_cb(f), ["c_custkey"])), _cb(_hash(_scan(orders, "orders").project(["o_custkey"]), ["o_custkey"])),
                          [("c_custkey", "o_custkey")], "LeftAnti", projection=(["c_phone", "c_acctbal"], None))
    proj = P.ProjectionExec([(substr(col("c_phone"), 1, 2), "cntrycode"), (col("c_acctbal"), "c_acctbal")], _cb(anti))
    gb = [(col("cntrycode"), "cntrycode")]
    aggs = [("count", None, "count(Int64(1))"), ("sum", col("c_acctbal"), "sum(custsale.c_acctbal)")]
    partial = P.AggregateExec("Partial", gb, aggs, proj)
    final = P.AggregateExec("FinalPartitioned", gb, aggs, _cb(_hash(partial, ["cntrycode"])))
    keys = [("cntrycode",) + ASC]
    out = P.ProjectionExec([(col("cntrycode"), "cntrycode"), (col("count(Int64(1))"), "numcust"), (col("sum(custsale.c_acctbal)"), "totacctbal")], P.SortExec(keys, final))
    return P.ScalarSubqueryExec(P.SortPreservingMergeExec(keys, out), [(sub, 0)], results)


# ------------------------------------------------------------------------------------------ Q2
def q2_plan(part, supplier, partsupp, nation, region):
    """q2.slt.part:101-144: the BRASS parts of size 15 with their suppliers of EUROPE (three Inner joins, the running result the
    build side, then LeftSemi against the EUROPE row of region), kept where the supply cost EQUALS the minimum over EUROPE's suppliers
    of that part: a LeftSemi join on the two columns (p_partkey, ps_supplycost) = (ps_partkey, min(ps_supplycost)) against the
    decorrelated subquery (partsupp x supplier x nation, LeftSemi region, MIN per part); top 10"""
    def europe():
        f = P.FilterExec(col("r_name").eq(lit("EUROPE", pa.string())), _scan(region, "region").project(["r_regionkey", "r_name"]), projection=["r_regionkey"])
        return _hash(_cb(f), ["r_regionkey"])

    p = _hash(_cb(P.FilterExec(col("p_size").eq(lit(15, pa.int32())).and_(col("p_type").like("%BRASS")),
                               _scan(part, "part").project(["p_partkey", "p_mfgr", "p_type", "p_size"]), projection=["p_partkey", "p_mfgr"])), ["p_partkey"])
    ps = _hash(_scan(partsupp, "partsupp").project(["ps_partkey", "ps_suppkey", "ps_supplycost"]), ["ps_partkey"])
    j1 = P.HashJoinExec(_cb(p), _cb(ps), [("p_partkey", "ps_partkey")], "Inner", projection=(["p_partkey", "p_mfgr"], ["ps_suppkey", "ps_supplycost"]))
    su = _hash(_scan(supplier, "supplier").project(["s_suppkey", "s_name", "s_address", "s_nationkey", "s_phone", "s_acctbal", "s_comment"]), ["s_suppkey"])
    j2 = P.HashJoinExec(_cb(_hash(_cb(j1), ["ps_suppkey"])), _cb(su), [("ps_suppkey", "s_suppkey")], "Inner",
                        projection=(["p_partkey", "p_mfgr", "ps_supplycost"], ["s_name", "s_address", "s_nationkey", "s_phone", "s_acctbal", "s_comment"]))
    n = _hash(_scan(nation, "nation").project(["n_nationkey", "n_name", "n_regionkey"]), ["n_nationkey"])
    j3 = P.HashJoinExec(_cb(_hash(_cb(j2), ["s_nationkey"])), _cb(n), [("s_nationkey", "n_nationkey")], "Inner",
                        projection=(["p_partkey", "p_mfgr", "s_name", "s_address", "s_phone", "s_acctbal", "s_comment", "ps_supplycost"], ["n_name", "n_regionkey"]))
    cols = ["p_partkey", "p_mfgr", "s_name", "s_address", "s_phone", "s_acctbal", "s_comment", "ps_supplycost", "n_name"]
    j4 = P.HashJoinExec(_cb(_hash(_cb(j3), ["n_regionkey"])), _cb(europe()), [("n_regionkey", "r_regionkey")], "LeftSemi", projection=(cols, None))
    # the subquery: the cheapest supply cost of every part among EUROPE's suppliers
    ps2 = _hash(_scan(partsupp, "partsupp").project(["ps_partkey", "ps_suppkey", "ps_supplycost"]), ["ps_suppkey"])
    su2 = _hash(_scan(supplier, "supplier").project(["s_suppkey", "s_nationkey"]), ["s_suppkey"])
    k1 = P.HashJoinExec(_cb(ps2), _cb(su2), [("ps_suppkey", "s_suppkey")], "Inner", projection=(["ps_partkey", "ps_supplycost"], ["s_nationkey"]))
    n2 = _hash(_scan(nation, "nation").project(["n_nationkey", "n_regionkey"]), ["n_nationkey"])
    k2 = P.HashJoinExec(_cb(_hash(_cb(k1), ["s_nationkey"])), _cb(n2), [("s_nationkey", "n_nationkey")], "Inner", projection=(["ps_partkey", "ps_supplycost"], ["n_regionkey"]))
    k3 = P.HashJoinExec(_cb(_hash(_cb(k2), ["n_regionkey"])), _cb(europe()), [("n_regionkey", "r_regionkey")], "LeftSemi", projection=(["ps_partkey", "ps_supplycost"], None))
    gb = [(col("ps_partkey"), "ps_partkey")]
    mn = "min(partsupp.ps_supplycost)"
    aggs = [("min", col("ps_supplycost"), mn)]
    cheapest = P.AggregateExec("FinalPartitioned", gb, aggs, _cb(_hash(P.AggregateExec("Partial", gb, aggs, _cb(k3)), ["ps_partkey"])))
    cheapest = P.ProjectionExec([(col(mn), mn), (col("ps_partkey"), "ps_partkey")], cheapest)
    out_cols = ["s_acctbal", "s_name", "n_name", "p_partkey", "p_mfgr", "s_address", "s_phone", "s_comment"]
    j5 = P.HashJoinExec(_cb(_hash(_cb(j4), ["p_partkey", "ps_supplycost"])), _cb(_hash(cheapest, ["ps_partkey", mn])),
                        [("p_partkey", "ps_partkey"), ("ps_supplycost", mn)], "LeftSemi", projection=(out_cols, None))
    keys = [("s_acctbal",) + DESC, ("n_name",) + ASC, ("s_name",) + ASC, ("p_partkey",) + ASC]
    return P.SortPreservingMergeExec(keys, P.SortExec(keys, _cb(j5), fetch=10), fetch=10)


# ----------------------------------------------------------------------------------------- Q10
def q10_plan(customer, orders, lineitem, nation):
    """q10.slt.part:70-91: customer (build) x the orders of 1993 Q4 x the returned lines x nation, the lost revenue per customer —
    grouped by SEVEN columns (the key, the name, the balance and four more strings), top 10 by revenue"""
    ccols = ["c_custkey", "c_name", "c_address", "c_nationkey", "c_phone", "c_acctbal", "c_comment"]
    c = _hash(_scan(customer, "customer").project(ccols), ["c_custkey"])
    of = P.FilterExec((col("o_orderdate") >= _d(1993, 10, 1)).and_(col("o_orderdate") < _d(1994, 1, 1)),
                      _scan(orders, "orders").project(["o_orderkey", "o_custkey", "o_orderdate"]), projection=["o_orderkey", "o_custkey"])
    j1 = P.HashJoinExec(_cb(c), _cb(_hash(_cb(of), ["o_custkey"])), [("c_custkey", "o_custkey")], "Inner", projection=(ccols, ["o_orderkey"]))
    lf = P.FilterExec(col("l_returnflag").eq(lit("R", pa.string())), _scan(lineitem, "lineitem").project(["l_orderkey", "l_extendedprice", "l_discount", "l_returnflag"]),
                      projection=["l_orderkey", "l_extendedprice", "l_discount"])
    j2 = P.HashJoinExec(_cb(_hash(_cb(j1), ["o_orderkey"])), _cb(_hash(_cb(lf), ["l_orderkey"])), [("o_orderkey", "l_orderkey")], "Inner",
                        projection=(ccols, ["l_extendedprice", "l_discount"]))
    n = _hash(_scan(nation, "nation").project(["n_nationkey", "n_name"]), ["n_nationkey"])
    j3 = P.HashJoinExec(_cb(_hash(_cb(j2), ["c_nationkey"])), _cb(n), [("c_nationkey", "n_nationkey")], "Inner",
                        projection=(["c_custkey", "c_name", "c_address", "c_phone", "c_acctbal", "c_comment", "l_extendedprice", "l_discount"], ["n_name"]))
    gcols = ["c_custkey", "c_name", "c_acctbal", "c_phone", "n_name", "c_address", "c_comment"]
    gb = [(col(g), g) for g in gcols]
    name = "sum(lineitem.l_extendedprice * Int64(1) - lineitem.l_discount)"
    aggs = [("sum", col("l_extendedprice") * (ONE - col("l_discount")), name)]
    final = P.AggregateExec("FinalPartitioned", gb, aggs, _cb(_hash(P.AggregateExec("Partial", gb, aggs, _cb(j3)), gcols)))
    top = P.SortExec([(name,) + DESC], final, fetch=10)
    out = P.ProjectionExec([(col("c_custkey"), "c_custkey"), (col("c_name"), "c_name"), (col(name), "revenue"), (col("c_acctbal"), "c_acctbal"), (col("n_name"), "n_name"),
                            (col("c_address"), "c_address"), (col("c_phone"), "c_phone"), (col("c_comment"), "c_comment")], top)
    return P.SortPreservingMergeExec([("revenue",) + DESC], out, fetch=10)


# ----------------------------------------------------------------------------------------- Q13
def q13_plan(customer, orders):
    """q13.slt.part:55-68: customer LEFT JOIN the orders whose comment is NOT LIKE '%special%requests%' (customer is the build side:
    its unmatched rows come out with a NULL o_orderkey), COUNT(o_orderkey) per customer (SinglePartitioned: the join output is
    already partitioned on c_custkey), then the distribution of those counts; top 10"""
    c = _hash(_scan(customer, "customer").project(["c_custkey"]), ["c_custkey"])
    of = P.FilterExec(col("o_comment").like("%special%requests%", negated=True), _scan(orders, "orders").project(["o_orderkey", "o_custkey", "o_comment"]),
                      projection=["o_orderkey", "o_custkey"])
    j = P.HashJoinExec(_cb(c), _cb(_hash(_cb(of), ["o_custkey"])), [("c_custkey", "o_custkey")], "Left", projection=(["c_custkey"], ["o_orderkey"]))
    per_customer = P.AggregateExec("SinglePartitioned", [(col("c_custkey"), "c_custkey")], [("count", col("o_orderkey"), "count(orders.o_orderkey)")], _cb(j))
    counts = P.ProjectionExec([(col("count(orders.o_orderkey)"), "c_count")], per_customer)
    gb = [(col("c_count"), "c_count")]
    aggs = [("count", None, "count(Int64(1))")]
    final = P.AggregateExec("FinalPartitioned", gb, aggs, _cb(_hash(P.AggregateExec("Partial", gb, aggs, counts), ["c_count"])))
    keys = [("count(Int64(1))",) + DESC, ("c_count",) + DESC]
    top = P.ProjectionExec([(col("c_count"), "c_count"), (col("count(Int64(1))"), "custdist")], P.SortExec(keys, final, fetch=10))
    return P.SortPreservingMergeExec([("custdist",) + DESC, ("c_count",) + DESC], top, fetch=10)


# ----------------------------------------------------------------------------------------- Q16
def q16_plan(partsupp, part, supplier):
    """q16.slt.part:69-88: partsupp x the parts kept by three predicates (<> on a string, IN over eight sizes, NOT LIKE), a null-aware
    LeftAnti join (NOT IN) against the suppliers whose comment is LIKE '%Customer%Complaints%', then COUNT(DISTINCT ps_suppkey) as two
    aggregate levels: the distinct (brand, type, size, supplier) groups with no aggregate, counted per (brand, type, size); top 10"""
    sizes = [lit(v, pa.int32()) for v in (49, 14, 23, 45, 19, 3, 36, 9)]
    pf = P.FilterExec(col("p_brand").ne(lit("Brand#45", pa.string())).and_(col("p_size").in_list(sizes)).and_(col("p_type").like("MEDIUM POLISHED%", negated=True)),
                      _scan(part, "part").project(["p_partkey", "p_brand", "p_type", "p_size"]))
    ps = _hash(_scan(partsupp, "partsupp").project(["ps_partkey", "ps_suppkey"]), ["ps_partkey"])
    j = P.HashJoinExec(_cb(ps), _cb(_hash(_cb(pf), ["p_partkey"])), [("ps_partkey", "p_partkey")], "Inner", projection=(["ps_suppkey"], ["p_brand", "p_type", "p_size"]))
    bad = P.FilterExec(col("s_comment").like("%Customer%Complaints%"), _scan(supplier, "supplier").project(["s_suppkey", "s_comment"]), projection=["s_suppkey"])
    anti = P.HashJoinExec(P.CoalescePartitionsExec(_cb(j)), _cb(bad), [("ps_suppkey", "s_suppkey")], "LeftAnti", null_aware=True)
    g4 = [(col("p_brand"), "p_brand"), (col("p_type"), "p_type"), (col("p_size"), "p_size"), (col("ps_suppkey"), "alias1")]
    d_partial = P.AggregateExec("Partial", g4, [], _cb(anti))
    g4f = [(col("p_brand"), "p_brand"), (col("p_type"), "p_type"), (col("p_size"), "p_size"), (col("alias1"), "alias1")]
    distinct = P.AggregateExec("FinalPartitioned", g4f, [], _cb(_hash(d_partial, ["p_brand", "p_type", "p_size", "alias1"])))
    g3 = [(col("p_brand"), "p_brand"), (col("p_type"), "p_type"), (col("p_size"), "p_size")]
    aggs = [("count", col("alias1"), "count(alias1)")]
    final = P.AggregateExec("FinalPartitioned", g3, aggs, _cb(_hash(P.AggregateExec("Partial", g3, aggs, distinct), ["p_brand", "p_type", "p_size"])))
    keys = [("count(alias1)",) + DESC, ("p_brand",) + ASC, ("p_type",) + ASC, ("p_size",) + ASC]
    top = P.ProjectionExec([(col("p_brand"), "p_brand"), (col("p_type"), "p_type"), (col("p_size"), "p_size"), (col("count(alias1)"), "supplier_cnt")], P.SortExec(keys, final, fetch=10))
    return P.SortPreservingMergeExec([("supplier_cnt",) + DESC, ("p_brand",) + ASC, ("p_type",) + ASC, ("p_size",) + ASC], top, fetch=10)


# ------------------------------------------------------------------------------------------ Q6
def q6_plan(lineitem):
    """q6.slt.part:38-43: one filter, one ungrouped SUM"""
    pred = (col("l_shipdate") >= _d(1994, 1, 1)).and_(col("l_shipdate") < _d(1995, 1, 1)) \
        .and_(col("l_discount") >= lit(Decimal("0.05"), D15_2)).and_(col("l_discount") <= lit(Decimal("0.07"), D15_2)) \
        .and_(col("l_quantity") < lit(Decimal("24.00"), D15_2))
    f = P.FilterExec(pred, _scan(lineitem, "lineitem"), projection=["l_extendedprice", "l_discount"])
    name = "sum(lineitem.l_extendedprice * lineitem.l_discount)"
    aggs = [("sum", col("l_extendedprice") * col("l_discount"), name)]
    partial = P.AggregateExec("Partial", [], aggs, _cb(f))
    final = P.AggregateExec("Final", [], aggs, P.CoalescePartitionsExec(partial))
    return P.ProjectionExec([(col(name), "revenue")], final)


# ----------------------------------------------------------------------------------------- Q12
def q12_plan(orders, lineitem):
    """q12.slt.part:62-73: build = the filtered lineitem rows (several per order), probe = orders; two SUM(CASE ...)"""
    mode = col("l_shipmode")
    pred = (mode.eq(lit("MAIL", pa.string())).or_(mode.eq(lit("SHIP", pa.string())))).and_(col("l_receiptdate") > col("l_commitdate")) \
        .and_(col("l_shipdate") < col("l_commitdate")).and_(col("l_receiptdate") >= _d(1994, 1, 1)).and_(col("l_receiptdate") < _d(1995, 1, 1))
    l = _hash(_cb(P.FilterExec(pred, _scan(lineitem, "lineitem"), projection=["l_orderkey", "l_shipmode"])), ["l_orderkey"])
    o = _hash(_scan(orders, "orders").project(["o_orderkey", "o_orderpriority"]), ["o_orderkey"])
    j = P.HashJoinExec(_cb(l), _cb(o), [("l_orderkey", "o_orderkey")], "Inner", projection=(["l_shipmode"], ["o_orderpriority"]))
    prio = col("o_orderpriority")
    urgent, high = lit("1-URGENT", pa.string()), lit("2-HIGH", pa.string())
    one, zero = lit(1, pa.int64()), lit(0, pa.int64())
    hi_name = 'sum(CASE WHEN orders.o_orderpriority = Utf8("1-URGENT") OR orders.o_orderpriority = Utf8("2-HIGH") THEN Int64(1) ELSE Int64(0) END)'
    lo_name = 'sum(CASE WHEN orders.o_orderpriority != Utf8("1-URGENT") AND orders.o_orderpriority != Utf8("2-HIGH") THEN Int64(1) ELSE Int64(0) END)'
    aggs = [("sum", case([(prio.eq(urgent).or_(prio.eq(high)), one)], zero), hi_name),
            ("sum", case([(prio.ne(urgent).and_(prio.ne(high)), one)], zero), lo_name)]
    gb = [(col("l_shipmode"), "l_shipmode")]
    partial = P.AggregateExec("Partial", gb, aggs, _cb(j))
    final = P.AggregateExec("FinalPartitioned", gb, aggs, _cb(_hash(partial, ["l_shipmode"])))
    keys = [("l_shipmode",) + ASC]
    proj = P.ProjectionExec([(col("l_shipmode"), "l_shipmode"), (col(hi_name), "high_line_count"), (col(lo_name), "low_line_count")], P.SortExec(keys, final))
    return P.SortPreservingMergeExec(keys, proj)


# ----------------------------------------------------------------------------------------- Q18
def q18_plan(customer, orders, lineitem):
    """q18.slt.part:70-87: two Inner joins, a LeftSemi join against `GROUP BY l_orderkey HAVING sum(l_quantity) > 300`
    (150 k·SF groups: the dense-integer-key aggregate node), a five-column GROUP BY, a two-key sort"""
    c = _hash(_scan(customer, "customer").project(["c_custkey", "c_name"]), ["c_custkey"])
    o = _hash(_scan(orders, "orders").project(["o_orderkey", "o_custkey", "o_totalprice", "o_orderdate"]), ["o_custkey"])
    j1 = P.HashJoinExec(_cb(c), _cb(o), [("c_custkey", "o_custkey")], "Inner", projection=(["c_custkey", "c_name"], ["o_orderkey", "o_totalprice", "o_orderdate"]))
    l = _hash(_scan(lineitem, "lineitem").project(["l_orderkey", "l_quantity"]), ["l_orderkey"])
    j2 = P.HashJoinExec(_cb(_hash(_cb(j1), ["o_orderkey"])), _cb(l), [("o_orderkey", "l_orderkey")], "Inner",
                        projection=(["c_custkey", "c_name", "o_orderkey", "o_totalprice", "o_orderdate"], ["l_quantity"]))
    gb = [(col("l_orderkey"), "l_orderkey")]
    qsum = "sum(lineitem.l_quantity)"
    aggs = [("sum", col("l_quantity"), qsum)]
    partial = P.AggregateExec("Partial", gb, aggs, _scan(lineitem, "lineitem").project(["l_orderkey", "l_quantity"]))
    final = P.AggregateExec("FinalPartitioned", gb, aggs, _cb(_hash(partial, ["l_orderkey"])))
    having = _cb(P.FilterExec(col(qsum) > lit(Decimal("300.00"), pa.decimal128(25, 2)), final, projection=["l_orderkey"]))
    semi = P.HashJoinExec(_cb(j2), having, [("o_orderkey", "l_orderkey")], "LeftSemi")
    gb2 = [(col(n), n) for n in ["c_name", "c_custkey", "o_orderkey", "o_orderdate", "o_totalprice"]]
    agg = P.AggregateExec("SinglePartitioned", gb2, aggs, _cb(semi))
    keys = [("o_totalprice",) + DESC, ("o_orderdate",) + ASC]
    return P.SortPreservingMergeExec(keys, P.SortExec(keys, agg))


# ----------------------------------------------------------------------------------------- Q19
def q19_plan(lineitem, part):
    """q19.slt.part:67-77: LeftSemi join (build = filtered lineitem) whose JoinFilter mixes both sides — brand / container / size of
    the part with the quantity of the line — IN lists over string columns, one ungrouped SUM"""
    qty, s = col("l_quantity"), lambda v: lit(v, pa.string())            # noqa: E731
    q = lambda v: lit(Decimal(v), D15_2)                                  # noqa: E731
    between = lambda e, lo, hi: (e >= q(lo)).and_(e <= q(hi))             # noqa: E731
    lpred = (col("l_shipmode").eq(s("AIR")).or_(col("l_shipmode").eq(s("AIR REG")))).and_(col("l_shipinstruct").eq(s("DELIVER IN PERSON"))) \
        .and_(between(qty, "1.00", "11.00").or_(between(qty, "10.00", "20.00")).or_(between(qty, "20.00", "30.00")))
    l = _hash(_cb(P.FilterExec(lpred, _scan(lineitem, "lineitem").project(["l_partkey", "l_quantity", "l_extendedprice", "l_discount", "l_shipinstruct", "l_shipmode"]),
                               projection=["l_partkey", "l_quantity", "l_extendedprice", "l_discount"])), ["l_partkey"])
    groups = [("Brand#12", ["SM CASE", "SM BOX", "SM PACK", "SM PKG"], 5, "1.00", "11.00"),
              ("Brand#23", ["MED BAG", "MED BOX", "MED PKG", "MED PACK"], 10, "10.00", "20.00"),
              ("Brand#34", ["LG CASE", "LG BOX", "LG PACK", "LG PKG"], 15, "20.00", "30.00")]

    def part_side(brand, container, size):
        alts = None
        for b, cs, mx, _, _ in groups:
            a = brand.eq(s(b)).and_(container.in_list([s(c) for c in cs])).and_(size <= lit(mx, pa.int32()))
            alts = a if alts is None else alts.or_(a)
        return alts
    ppred = (col("p_size") >= lit(1, pa.int32())).and_(part_side(col("p_brand"), col("p_container"), col("p_size")))
    p = _hash(_cb(P.FilterExec(ppred, _scan(part, "part").project(["p_partkey", "p_brand", "p_size", "p_container"]))), ["p_partkey"])
    # JoinFilter over the intermediate columns f0 = l_quantity (Left 1), f1 = p_brand, f2 = p_size, f3 = p_container (Right 1, 2, 3)
    f_qty, f_brand, f_size, f_cont = col("f0"), col("f1"), col("f2"), col("f3")
    jf = None
    for b, cs, mx, lo, hi in groups:
        a = f_brand.eq(s(b)).and_(f_cont.in_list([s(c) for c in cs])).and_(f_qty >= q(lo)).and_(f_qty <= q(hi)).and_(f_size <= lit(mx, pa.int32()))
        jf = a if jf is None else jf.or_(a)
    semi = P.HashJoinExec(_cb(l), _cb(p), [("l_partkey", "p_partkey")], "LeftSemi", projection=(["l_extendedprice", "l_discount"], None),
                          filter=(jf, [(1, "Left"), (1, "Right"), (2, "Right"), (3, "Right")]))
    name = "sum(lineitem.l_extendedprice * Int64(1) - lineitem.l_discount)"
    aggs = [("sum", col("l_extendedprice") * (ONE - col("l_discount")), name)]
    partial = P.AggregateExec("Partial", [], aggs, _cb(semi))
    final = P.AggregateExec("Final", [], aggs, P.CoalescePartitionsExec(partial))
    return P.ProjectionExec([(col(name), "revenue")], final)


# ----------------------------------------------------------------------------------------- Q21
def q21_plan(supplier, lineitem, orders, nation):
    """q21.slt.part:92-122: EXISTS / NOT EXISTS decorrelated to LeftSemi / LeftAnti joins carrying the JoinFilter
    `l_suppkey != l_suppkey` (joins/join_filter.rs), two more LeftSemi joins, COUNT(*) per supplier name"""
    late = col("l_receiptdate") > col("l_commitdate")
    s = _hash(_scan(supplier, "supplier").project(["s_suppkey", "s_name", "s_nationkey"]), ["s_suppkey"])
    l1 = _hash(_cb(P.FilterExec(late, _scan(lineitem, "lineitem"), projection=["l_orderkey", "l_suppkey"])), ["l_suppkey"])
    j = P.HashJoinExec(_cb(s), _cb(l1), [("s_suppkey", "l_suppkey")], "Inner", projection=(["s_name", "s_nationkey"], ["l_orderkey", "l_suppkey"]))
    o = _hash(_cb(P.FilterExec(col("o_orderstatus").eq(lit("F", pa.string())), _scan(orders, "orders"), projection=["o_orderkey"])), ["o_orderkey"])
    semi_o = P.HashJoinExec(_cb(_hash(_cb(j), ["l_orderkey"])), _cb(o), [("l_orderkey", "o_orderkey")], "LeftSemi")
    n = _hash(_cb(P.FilterExec(col("n_name").eq(lit("SAUDI ARABIA", pa.string())), _scan(nation, "nation").project(["n_nationkey", "n_name"]),
                               projection=["n_nationkey"])), ["n_nationkey"])
    semi_n = P.HashJoinExec(_cb(_hash(_cb(semi_o), ["s_nationkey"])), _cb(n), [("s_nationkey", "n_nationkey")], "LeftSemi",
                            projection=(["s_name", "l_orderkey", "l_suppkey"], None))
    other_supplier = (col("f0").ne(col("f1")), [(2, "Left"), (1, "Right")])          # left.l_suppkey != right.l_suppkey
    l2 = _hash(_scan(lineitem, "lineitem").project(["l_orderkey", "l_suppkey"]), ["l_orderkey"])
    exists = P.HashJoinExec(_cb(_hash(_cb(semi_n), ["l_orderkey"])), _cb(l2), [("l_orderkey", "l_orderkey")], "LeftSemi", filter=other_supplier)
    l3 = _hash(_cb(P.FilterExec(late, _scan(lineitem, "lineitem"), projection=["l_orderkey", "l_suppkey"])), ["l_orderkey"])
    not_exists = P.HashJoinExec(_cb(exists), _cb(l3), [("l_orderkey", "l_orderkey")], "LeftAnti", projection=(["s_name"], None), filter=other_supplier)
    gb = [(col("s_name"), "s_name")]
    aggs = [("count", None, "count(Int64(1))")]
    partial = P.AggregateExec("Partial", gb, aggs, _cb(not_exists))
    final = P.AggregateExec("FinalPartitioned", gb, aggs, _cb(_hash(partial, ["s_name"])))
    keys = [("count(Int64(1))",) + DESC, ("s_name",) + ASC]
    proj = P.ProjectionExec([(col("s_name"), "s_name"), (col("count(Int64(1))"), "numwait")], P.SortExec(keys, final))
    return P.SortPreservingMergeExec([("numwait",) + DESC, ("s_name",) + ASC], proj)
