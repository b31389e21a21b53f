"""Pin the oracle's AggregateExec restatement against the reference's check_aggregates test
(aggregates/mod.rs:3591-3706: Partial state (count,sum) then Final AVG) and its decimal typing
against the TPC-H Q1 plan/answer files."""
from decimal import Decimal

import pyarrow as pa

from oracle import oracle
from tests.util import load_golden, sorted_rows

G = load_golden("aggregate_check_aggregates.json")


def _input():
    return pa.table({"a": pa.array(G["input"]["a"], type=pa.uint32()), "b": pa.array(G["input"]["b"], type=pa.float64())})


def test_partial_then_final_avg_matches_reference_snapshots():
    t = _input()
    gb, aggs = [(("col", "a"), "a")], [("avg", ("col", "b"), "AVG(b)")]
    partial = oracle.aggregate(t, gb, aggs, mode="Partial")
    assert partial.column_names == G["partial"]["columns"]
    assert sorted_rows(partial) == [tuple(r) for r in G["partial"]["rows"]]
    assert partial.schema.field("AVG(b)[count]").type == pa.uint64()
    # one partial per input batch, merged by the Final aggregate (CoalescePartitionsExec in the test)
    parts = [oracle.aggregate(t.slice(lo, hi - lo), gb, aggs, mode="Partial") for lo, hi in G["batches"]]
    final = oracle.aggregate(pa.concat_tables(parts), gb, aggs, mode="Final")
    assert final.column_names == G["final"]["columns"]
    assert sorted_rows(final) == [tuple(r) for r in G["final"]["rows"]]
    single = oracle.aggregate(t, gb, aggs, mode="Single")
    assert sorted_rows(single) == [tuple(r) for r in G["final"]["rows"]]


def test_groups_are_numbered_in_first_seen_order():
    """GroupValues contract (group_values/mod.rs:88-92)"""
    t = pa.table({"k": pa.array([7, 3, 7, 9, 3, None, 9, None], type=pa.int64()), "v": pa.array(range(8), type=pa.int64())})
    out = oracle.aggregate(t, [(("col", "k"), "k")], [("sum", ("col", "v"), "s"), ("count", None, "c")])
    assert out.column("k").to_pylist() == [7, 3, 9, None]      # NULL is a group of its own
    assert out.column("s").to_pylist() == [2, 5, 9, 12] and out.column("c").to_pylist() == [2, 2, 2, 2]


def test_decimal_typing_follows_q1_plan():
    """sum(l_extendedprice * (1 - l_discount)) : Decimal128(38,4); * (1 + l_tax) : (38,6); avg(l_quantity):
    Decimal128(19,6) — scales visible in tpch/answers/q1.slt.part:42-45, literal Decimal128(Some(1),20,0) in
    tpch/plans/q1.slt.part:45-46"""
    D = lambda s: Decimal(s)
    t = pa.table({"p": pa.array([D("100.00"), D("200.50")], type=pa.decimal128(15, 2)),
                  "d": pa.array([D("0.05"), D("0.10")], type=pa.decimal128(15, 2)),
                  "x": pa.array([D("0.02"), D("0.08")], type=pa.decimal128(15, 2)),
                  "q": pa.array([D("17.00"), D("36.00")], type=pa.decimal128(15, 2))})
    one = ("lit", 1, pa.decimal128(20, 0))
    disc = ("bin", "*", ("col", "p"), ("bin", "-", one, ("col", "d")))
    charge = ("bin", "*", disc, ("bin", "+", one, ("col", "x")))
    out = oracle.aggregate(t, [], [("sum", disc, "sum_disc_price"), ("sum", charge, "sum_charge"), ("avg", ("col", "q"), "avg_qty"),
                                   ("avg", ("col", "d"), "avg_disc"), ("count", None, "count_order")])
    assert out.schema.field("sum_disc_price").type == pa.decimal128(38, 4)
    assert out.schema.field("sum_charge").type == pa.decimal128(38, 6)
    assert out.schema.field("avg_qty").type == pa.decimal128(19, 6)
    row = out.to_pylist()[0]
    assert row["sum_disc_price"] == D("100.00") * D("0.95") + D("200.50") * D("0.90")
    assert row["sum_charge"] == D("100.00") * D("0.95") * D("1.02") + D("200.50") * D("0.90") * D("1.08")
    assert row["avg_qty"] == D("26.500000") and row["avg_disc"] == D("0.075000") and row["count_order"] == 2
    # truncating division (DecimalAverager::avg, functions-aggregate-common/src/utils.rs:157-176)
    t3 = pa.table({"q": pa.array([D("0.01"), D("0.01"), D("0.02")], type=pa.decimal128(15, 2))})
    assert oracle.aggregate(t3, [], [("avg", ("col", "q"), "a")]).to_pylist()[0]["a"] == D("0.013333")
