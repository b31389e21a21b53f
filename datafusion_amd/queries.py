"""TPC-H Q1 and Q3 as the operator DAGs DataFusion's planner produces, executed operator by
operator on device tables (BASELINE.json configs 4 and 5).

The plans are the reference's pinned physical plans, node for node:
  Q1  datafusion/sqllogictest/test_files/tpch/plans/q1.slt.part:42-58
  Q3  datafusion/sqllogictest/test_files/tpch/plans/q3.slt.part:44-76
Each `RepartitionExec(Hash)` of the plan is `exchange.hash_exchange` (partition kernel + RCCL
all-to-all) when a process group with more than one rank is active and a no-op otherwise; the
`SortPreservingMergeExec` at the root gathers the per-rank top rows to every rank and merges.

String columns: l_returnflag / l_linestatus are 1-byte codes (UInt8 = the ASCII byte) and
c_mktsegment is a UInt8 dictionary code (tpch.SEGMENTS order), as produced by the generator
(SURVEY.md §7 "Strings: first pass").
"""
from __future__ import annotations

import datetime

import pyarrow as pa

from . import ops
from .expr import col, lit
from .table import DeviceTable

DATE_Q1 = datetime.date(1998, 9, 2)
DATE_Q3 = datetime.date(1995, 3, 15)
SEGMENT_BUILDING = 1  # tpch.SEGMENTS.index("BUILDING")
ONE = lit(1, pa.decimal128(20, 0))  # Int64(1) coerced to Decimal128(20,0), type_coercion/binary.rs:1257-1273


def _world(group=None) -> int:
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(group)
    except ImportError:
        pass
    return 1


def _repartition(table: DeviceTable, keys, group=None) -> DeviceTable:
    """RepartitionExec: partitioning=Hash(keys, N)"""
    if _world(group) == 1:
        return table
    from .exchange import hash_exchange
    return hash_exchange(table, keys, group)


def _merge_sorted(table: DeviceTable, keys, fetch, group=None) -> DeviceTable:
    """SortPreservingMergeExec(fetch): every rank contributes its (already sorted, <= fetch rows)
    partition; the merged result is replicated on all ranks: a device all-gather (dfgpu_exchange_broadcast:
    values, validity bitmaps and dictionaries cross below the C ABI) and one more device sort."""
    if _world(group) == 1:
        return table
    from .exchange import broadcast_table, comm_for
    import pyarrow as pa
    first = table.schema.field(table.index_of(keys[0][0])).type if keys else None
    int_like = first is not None and (pa.types.is_integer(first) or pa.types.is_date32(first)) and first not in (pa.uint64(), pa.int8(), pa.int16(), pa.uint16()) and \
        table.dictionary_size(keys[0][0]) is None       # (a dictionary-encoded string key is ordered by its strings, not by its indices on every rank)
    if fetch is None and int_like:
        # an unbounded ORDER BY: sample sort.  Rows move to the rank that owns their first key's range (dfgpu_exchange_range:
        # splitters from all ranks' samples, all-to-all(v)), every rank sorts ITS range, and the ranges read in rank order are the
        # result — no rank sorts everything.  SortPreservingMergeExec has ONE output partition: the sorted ranges are gathered to
        # every rank (a concatenation in rank order, no second sort); an engine that streams the result from rank 0 keeps them apart.
        comm = comm_for(group)
        mine = comm.range_exchange(table, keys[0][0], bool(keys[0][1]), bool(keys[0][2]))
        ranged = ops.sort(mine, keys)
        mine.free()
        out = broadcast_table(ranged, group)
        if out is not ranged:
            ranged.free()
        return out
    merged = broadcast_table(table, group)          # TopK: at most `fetch` rows per rank
    out = ops.sort(merged, keys, fetch=fetch)
    merged.free()
    return out


# ------------------------------------------------------------------------------------ Q1
Q1_GROUP_BY = [(col("l_returnflag"), "l_returnflag"), (col("l_linestatus"), "l_linestatus")]


def q1_aggs():
    ce = col("__common_expr_1")
    return [("sum", col("l_quantity"), "sum_qty"), ("sum", col("l_extendedprice"), "sum_base_price"), ("sum", ce, "sum_disc_price"),
            ("sum", ce * (ONE + col("l_tax")), "sum_charge"), ("avg", col("l_quantity"), "avg_qty"),
            ("avg", col("l_extendedprice"), "avg_price"), ("avg", col("l_discount"), "avg_disc"), ("count", None, "count_order")]


def q1_aggs_inlined():
    """q1_aggs() with the ProjectionExec inlined: `__common_expr_1` spelled out over lineitem's columns
    (the fused node's value numbering finds the common subexpression again)"""
    ce = col("l_extendedprice") * (ONE - col("l_discount"))
    return [("sum", col("l_quantity"), "sum_qty"), ("sum", col("l_extendedprice"), "sum_base_price"), ("sum", ce, "sum_disc_price"),
            ("sum", ce * (ONE + col("l_tax")), "sum_charge"), ("avg", col("l_quantity"), "avg_qty"),
            ("avg", col("l_extendedprice"), "avg_price"), ("avg", col("l_discount"), "avg_disc"), ("count", None, "count_order")]


class _Q1Planned:
    """the fused FilterExec + ProjectionExec + AggregateExec node of Q1, planned: expressions lowered to the C ABI once, the AVG
    return types the Final node needs computed once (planning is per query, execution per `execute()` — the reference times both,
    tpch/run.rs:165-215, but its planner does not re-derive a plan it holds)"""

    def __init__(self, lineitem: DeviceTable, mode: str):
        pred = col("l_shipdate") <= lit(DATE_Q1, pa.date32())
        self.return_types = ops.aggregate_return_types(lineitem, q1_aggs_inlined())   # what the Final node is planned with (AVG types)
        self.node = ops.AggregatePlan(lineitem, Q1_GROUP_BY, q1_aggs_inlined(), mode, predicate=pred)


def plan_q1(lineitem: DeviceTable, group=None) -> _Q1Planned:
    """Q1's fused node planned for `lineitem`'s schema — what a caller that runs the query more than once holds on to and hands to q1()
    (bench.py plans once, outside its timed region, and says so on its line; q1() without a plan plans per call).  No cache is kept
    anywhere: the plan lives as long as the caller keeps it."""
    return _Q1Planned(lineitem, "Single" if _world(group) == 1 else "Partial")


def q1(lineitem: DeviceTable, group=None, fused: bool = True, plan: _Q1Planned = None) -> DeviceTable:
    """q1.slt.part:50-58, bottom-up: FilterExec(l_shipdate <= 1998-09-02, projection) ->
    ProjectionExec(__common_expr_1 = l_extendedprice * (1 - l_discount), ...) ->
    AggregateExec(Partial) -> RepartitionExec(Hash(flag, status)) -> AggregateExec(FinalPartitioned)
    -> SortExec -> SortPreservingMergeExec.  With one partition DataFusion plans a single
    AggregateExec(Single) instead of Partial/Final; so do we.

    fused=True (what the optimizer rule substitutes): FilterExec + ProjectionExec + AggregateExec as ONE
    node — predicate, projection expressions and accumulation in a single pass over lineitem's seven
    referenced columns (dfgpu_agg_update_filtered).  fused=False runs the three operators one after the
    other, materialising the filter's and the projection's outputs."""
    first_mode = "Single" if _world(group) == 1 else "Partial"
    if fused:
        if plan is None:
            plan = _Q1Planned(lineitem, first_mode)
        return_types = plan.return_types
        first = plan.node.execute(lineitem)
    else:
        pred = col("l_shipdate") <= lit(DATE_Q1, pa.date32())
        return_types = ops.aggregate_return_types(lineitem, q1_aggs_inlined())   # what the Final node is planned with (AVG types)
        f = ops.filter(lineitem, pred, ["l_extendedprice", "l_discount", "l_quantity", "l_tax", "l_returnflag", "l_linestatus"])
        p = ops.project(f, [(col("l_extendedprice") * (ONE - col("l_discount")), "__common_expr_1"), (col("l_quantity"), "l_quantity"),
                            (col("l_extendedprice"), "l_extendedprice"), (col("l_discount"), "l_discount"), (col("l_tax"), "l_tax"),
                            (col("l_returnflag"), "l_returnflag"), (col("l_linestatus"), "l_linestatus")])
        f.free()
        first = ops.aggregate(p, Q1_GROUP_BY, q1_aggs(), first_mode)
        p.free()
    keys = [("l_returnflag", False, False), ("l_linestatus", False, False)]
    if _world(group) == 1:
        agg = first
    else:
        routed = _repartition(first, ["l_returnflag", "l_linestatus"], group)
        agg = ops.aggregate(routed, Q1_GROUP_BY, q1_aggs(), "FinalPartitioned", return_types=return_types)
        if routed is not first:
            routed.free()
        first.free()
    out = ops.sort(agg, keys)
    agg.free()
    return _merge_sorted(out, keys, None, group)


# ------------------------------------------------------------------------------------ Q3
def _q3_fused_filters(customer, orders, lineitem, stats, probe_mode, segment):
    c = ops.filter(customer, col("c_mktsegment").eq(segment), ["c_custkey"])
    ht = ops.JoinHashTable(c, ["c_custkey"], probe_mode=probe_mode)
    # 12) FilterExec o_orderdate < 1995-03-15 + 07) HashJoinExec RightSemi on (c_custkey, o_custkey)
    semi = ht.probe(orders, ["o_custkey"], "RightSemi", probe_cols=["o_orderkey", "o_orderdate", "o_shippriority"],
                    predicate=col("o_orderdate") < lit(DATE_Q3, pa.date32()))
    ht.free()
    # 15) FilterExec l_shipdate > 1995-03-15 + 05) HashJoinExec Inner on (o_orderkey, l_orderkey)
    ht2 = ops.JoinHashTable(semi, ["o_orderkey"], probe_mode=probe_mode)
    j = ht2.probe(lineitem, ["l_orderkey"], "Inner", ["o_orderdate", "o_shippriority"], ["l_orderkey", "l_extendedprice", "l_discount"],
                  predicate=col("l_shipdate") > lit(DATE_Q3, pa.date32()))
    ht2.free()
    if stats is not None:
        stats.update(customer_filtered=c.num_rows, semi_join=semi.num_rows, join=j.num_rows)
    gb = [(col("l_orderkey"), "l_orderkey"), (col("o_orderdate"), "o_orderdate"), (col("o_shippriority"), "o_shippriority")]
    agg = ops.aggregate(j, gb, [("sum", col("l_extendedprice") * (ONE - col("l_discount")), "revenue")], "SinglePartitioned")
    if stats is not None:
        stats.update(groups=agg.num_rows)
    for t in (c, semi, j):
        t.free()
    top = ops.sort(agg, Q3_SORT, fetch=10)
    agg.free()
    return top.select(["l_orderkey", "revenue", "o_orderdate", "o_shippriority"])


Q3_SORT = [("revenue", True, True), ("o_orderdate", False, False)]  # revenue DESC (NULLS FIRST), o_orderdate ASC NULLS LAST


def q3(customer: DeviceTable, orders: DeviceTable, lineitem: DeviceTable, group=None, stats: dict | None = None,
       probe_mode: int = ops.PROBE_MODES["single_pass_unordered"], fused: bool = True, segment_literal=None) -> DeviceTable:
    """q3.slt.part:61-76, bottom-up.  `stats` (optional) receives intermediate row counts.

    fused=True (what the optimizer rule substitutes on one GPU): the FilterExecs on orders and lineitem are fused
    below the probe side of their HashJoinExec (dfgpu_join_probe_filtered) — the single-pass probe applies the
    predicate's row mask in the probe kernel, so neither filtered table is materialised.  With a repartition
    between filter and join (N > 1) the filters stay separate operators, as in the reference plan.
    segment_literal: what c_mktsegment is compared with — default the UInt8 code of 'BUILDING' (the device generator's
    layout); lit("BUILDING", pa.string()) for a dictionary-encoded string column."""
    segment = lit(SEGMENT_BUILDING, pa.uint8()) if segment_literal is None else segment_literal
    fuse_filters = fused and _world(group) == 1 and probe_mode in (ops.PROBE_MODES["single_pass_unordered"], ops.PROBE_MODES["single_pass_ordered"])
    if fuse_filters:
        return _q3_fused_filters(customer, orders, lineitem, stats, probe_mode, segment)
    # 09) FilterExec: c_mktsegment = BUILDING, projection=[c_custkey]; 08) Repartition Hash(c_custkey)
    c = ops.filter(customer, col("c_mktsegment").eq(segment), ["c_custkey"])
    c_r = _repartition(c, ["c_custkey"], group)
    # 12) FilterExec: o_orderdate < 1995-03-15; 11) Repartition Hash(o_custkey)
    o = ops.filter(orders, col("o_orderdate") < lit(DATE_Q3, pa.date32()), ["o_orderkey", "o_custkey", "o_orderdate", "o_shippriority"])
    o_r = _repartition(o, ["o_custkey"], group)
    # 07) HashJoinExec RightSemi on (c_custkey, o_custkey), projection=[o_orderkey, o_orderdate, o_shippriority]
    # (both joins feed a RepartitionExec / AggregateExec: no ancestor needs the probe-side order, so the
    # planner may take the unordered single-pass probe)
    ht = ops.JoinHashTable(c_r, ["c_custkey"], probe_mode=probe_mode)
    semi = ht.probe(o_r, ["o_custkey"], "RightSemi", probe_cols=["o_orderkey", "o_orderdate", "o_shippriority"])
    ht.free()
    # 06) Repartition Hash(o_orderkey)
    semi_r = _repartition(semi, ["o_orderkey"], group)
    # 15) FilterExec: l_shipdate > 1995-03-15, projection=[l_orderkey, l_extendedprice, l_discount]; 14) Repartition Hash(l_orderkey)
    l = ops.filter(lineitem, col("l_shipdate") > lit(DATE_Q3, pa.date32()), ["l_orderkey", "l_extendedprice", "l_discount"])
    l_r = _repartition(l, ["l_orderkey"], group)
    # 05) HashJoinExec Inner on (o_orderkey, l_orderkey), projection=[o_orderdate, o_shippriority, l_orderkey, l_extendedprice, l_discount]
    ht2 = ops.JoinHashTable(semi_r, ["o_orderkey"], probe_mode=probe_mode)
    j = ht2.probe(l_r, ["l_orderkey"], "Inner", ["o_orderdate", "o_shippriority"], ["l_orderkey", "l_extendedprice", "l_discount"])
    ht2.free()
    if stats is not None:
        stats.update(customer_filtered=c.num_rows, orders_filtered=o.num_rows, semi_join=semi.num_rows, lineitem_filtered=l.num_rows, join=j.num_rows)
    # 04) AggregateExec SinglePartitioned gby=[l_orderkey, o_orderdate, o_shippriority], sum(l_extendedprice * (1 - l_discount))
    gb = [(col("l_orderkey"), "l_orderkey"), (col("o_orderdate"), "o_orderdate"), (col("o_shippriority"), "o_shippriority")]
    agg = ops.aggregate(j, gb, [("sum", col("l_extendedprice") * (ONE - col("l_discount")), "revenue")], "SinglePartitioned")
    if stats is not None:
        stats.update(groups=agg.num_rows)
    for t in {id(x): x for x in (c, c_r, o, o_r, semi, semi_r, l, l_r, j)}.values():
        t.free()
    # 03) SortExec TopK(fetch=10) [revenue DESC, o_orderdate ASC NULLS LAST]; 02) ProjectionExec reorder
    top = ops.sort(agg, Q3_SORT, fetch=10)
    agg.free()
    out = top.select(["l_orderkey", "revenue", "o_orderdate", "o_shippriority"])
    # 01) SortPreservingMergeExec fetch=10
    return _merge_sorted(out, Q3_SORT, 10, group)
