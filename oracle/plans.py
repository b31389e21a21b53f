"""TPC-H Q1 and Q3 as the reference plans them on a CPU, on the oracle's operators.

TEST INFRASTRUCTURE ONLY (see oracle/oracle.py): imported by tests/ (the full-size parity tests) and by bench.py's
`cpu_baseline` leg, never by the product.

The plans are the reference's pinned physical plans, node for node, run the way DataFusion runs them — the scan cut
into `target_partitions` row ranges, one thread per partition:
  Q1  datafusion/sqllogictest/test_files/tpch/plans/q1.slt.part:42-58 — FilterExec + ProjectionExec + AggregateExec(Partial) per
      partition, RepartitionExec(Hash) + AggregateExec(FinalPartitioned), SortExec / SortPreservingMergeExec
  Q3  datafusion/sqllogictest/test_files/tpch/plans/q3.slt.part:44-76 — three FilterExecs per partition, four RepartitionExec(Hash)
      in two phases, HashJoinExec(Partitioned, RightSemi) and HashJoinExec(Partitioned, Inner) per hash partition,
      AggregateExec(SinglePartitioned), SortExec(TopK fetch=10) per partition, SortPreservingMergeExec(fetch=10)
String columns are the generator's 1-byte codes (l_returnflag / l_linestatus = the ASCII byte, c_mktsegment = a UInt8 code), the
layout of the GPU leg's tables (datafusion_amd/queries.py)."""
from __future__ import annotations

import datetime
from concurrent.futures import ThreadPoolExecutor

import pyarrow as pa

from . import oracle

DATE_Q1 = datetime.date(1998, 9, 2)      # q1.slt.part:52  l_shipdate <= 1998-09-02
DATE_Q3 = datetime.date(1995, 3, 15)     # q3.slt.part:72,75
SEGMENT_BUILDING = 1                     # tpch.SEGMENTS.index("BUILDING")
ONE = ("lit", 1, pa.decimal128(20, 0))   # Int64(1) coerced to Decimal128(20,0): expr-common/src/type_coercion/binary.rs:1257-1273
DISC_PRICE = ("bin", "*", ("col", "l_extendedprice"), ("bin", "-", ONE, ("col", "l_discount")))
Q3_SORT = [("revenue", True, True), ("o_orderdate", False, False)]   # revenue DESC (nulls first), o_orderdate ASC NULLS LAST

Q1_GROUP_BY = [(("col", "l_returnflag"), "l_returnflag"), (("col", "l_linestatus"), "l_linestatus")]
Q1_AGGS = [("sum", ("col", "l_quantity"), "sum_qty"), ("sum", ("col", "l_extendedprice"), "sum_base_price"), ("sum", DISC_PRICE, "sum_disc_price"),
           ("sum", ("bin", "*", DISC_PRICE, ("bin", "+", ONE, ("col", "l_tax"))), "sum_charge"), ("avg", ("col", "l_quantity"), "avg_qty"),
           ("avg", ("col", "l_extendedprice"), "avg_price"), ("avg", ("col", "l_discount"), "avg_disc"), ("count", None, "count_order")]
# AVG(Decimal128(15,2)) -> Decimal128(19,6): functions-aggregate/src/average.rs:131-172 (what the Final node is planned with)
Q1_RETURN_TYPES = {"avg_qty": pa.decimal128(19, 6), "avg_price": pa.decimal128(19, 6), "avg_disc": pa.decimal128(19, 6)}


def _slices(t: pa.Table, P: int):
    n = t.num_rows
    return [t.slice(n * k // P, n * (k + 1) // P - n * k // P) for k in range(P)]


def _by_hash(parts, key: str, P: int):
    """RepartitionExec(Hash([key], P)): every input partition splits its rows by the reference's hash routing
    (repartition/mod.rs:875-935); output partition q is the concatenation of the q-th pieces"""
    def split(t):
        if t.num_rows == 0:
            return [t] * P
        return oracle.hash_partition(t, [key], P)[0]
    with ThreadPoolExecutor(P) as ex:
        pieces = list(ex.map(split, parts))
    return [pa.concat_tables([pieces[i][q] for i in range(len(parts))]) for q in range(P)]


def run_q1(lineitem: pa.Table, P: int, stats: dict | None = None) -> pa.Table:
    def partial(t):
        f = oracle.filter(t, ("bin", "<=", ("col", "l_shipdate"), ("lit", DATE_Q1, pa.date32())),
                          ["l_quantity", "l_extendedprice", "l_discount", "l_tax", "l_returnflag", "l_linestatus"])
        return f.num_rows, oracle.aggregate(f, Q1_GROUP_BY, Q1_AGGS, "Partial")
    with ThreadPoolExecutor(P) as ex:
        parts = list(ex.map(partial, _slices(lineitem, P)))
    if stats is not None:
        stats.update(filtered=sum(n for n, _ in parts))
    fin = oracle.aggregate(pa.concat_tables([t for _, t in parts]), Q1_GROUP_BY, Q1_AGGS, "FinalPartitioned", return_types=Q1_RETURN_TYPES)
    return oracle.sort(fin, [("l_returnflag", False, False), ("l_linestatus", False, False)])


def run_q3(customer: pa.Table, orders: pa.Table, lineitem: pa.Table, P: int, stats: dict | None = None) -> pa.Table:
    """`stats` (optional) receives the intermediate row counts the GPU leg reports (queries.q3's `stats`).  Intermediates are dropped as
    soon as their consumer has run: at SF300 the filtered lineitem and its repartitioned copy are 39 GB each."""
    with ThreadPoolExecutor(P) as ex:
        c = list(ex.map(lambda t: oracle.filter(t, ("bin", "=", ("col", "c_mktsegment"), ("lit", SEGMENT_BUILDING, pa.uint8())), ["c_custkey"]), _slices(customer, P)))
        o = list(ex.map(lambda t: oracle.filter(t, ("bin", "<", ("col", "o_orderdate"), ("lit", DATE_Q3, pa.date32())),
                                                ["o_orderkey", "o_custkey", "o_orderdate", "o_shippriority"]), _slices(orders, P)))
    c_r, o_r = _by_hash(c, "c_custkey", P), _by_hash(o, "o_custkey", P)
    n_c = sum(t.num_rows for t in c)
    del c, o
    with ThreadPoolExecutor(P) as ex:
        semi = list(ex.map(lambda ab: oracle.hash_join(ab[0], ab[1], [("c_custkey", "o_custkey")], "RightSemi").select(["o_orderkey", "o_orderdate", "o_shippriority"]),
                           zip(c_r, o_r)))
    del c_r, o_r
    s_r = _by_hash(semi, "o_orderkey", P)
    n_semi = sum(t.num_rows for t in semi)
    del semi
    with ThreadPoolExecutor(P) as ex:
        l = list(ex.map(lambda t: oracle.filter(t, ("bin", ">", ("col", "l_shipdate"), ("lit", DATE_Q3, pa.date32())),
                                                ["l_orderkey", "l_extendedprice", "l_discount"]), _slices(lineitem, P)))
    l_r = _by_hash(l, "l_orderkey", P)
    del l
    gb = [(("col", "l_orderkey"), "l_orderkey"), (("col", "o_orderdate"), "o_orderdate"), (("col", "o_shippriority"), "o_shippriority")]

    def tail(ab):
        j = oracle.hash_join(ab[0], ab[1], [("o_orderkey", "l_orderkey")], "Inner").select(["o_orderdate", "o_shippriority", "l_orderkey", "l_extendedprice", "l_discount"])
        a = oracle.aggregate(j, gb, [("sum", DISC_PRICE, "revenue")], "SinglePartitioned")
        return j.num_rows, a.num_rows, oracle.sort(a, Q3_SORT, fetch=10)
    with ThreadPoolExecutor(P) as ex:
        tops = list(ex.map(tail, zip(s_r, l_r)))
    del s_r, l_r
    if stats is not None:
        stats.update(customer_filtered=n_c, semi_join=n_semi, join=sum(t[0] for t in tops), groups=sum(t[1] for t in tops))
    out = oracle.sort(pa.concat_tables([t[2] for t in tops]), Q3_SORT, fetch=10)
    return out.select(["l_orderkey", "revenue", "o_orderdate", "o_shippriority"])
