"""Pin the oracle's AggregateExec restatement against the reference's check_aggregates test
(aggregates/mod.rs:3591-3706: Partial state (count,sum) then Final AVG) and its decimal typing
against the TPC-H Q1 plan/answer files."""
from decimal import Decimal

import pyarrow as pa
import pytest

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


# ----------------------------------------------------------------------------------------------- AVG: avg_cases
AVG = load_golden("avg_cases.json")


def _dec(unscaled: int, scale: int):
    """exact Decimal (scaleb would round to the context's 28 digits)"""
    from decimal import Decimal
    return Decimal((0 if unscaled >= 0 else 1, tuple(int(c) for c in str(abs(unscaled))), -scale))


def _avg_case_table(rec):
    if rec["input_type"][0] == "Float64":
        return pa.table({"v": pa.array(rec["values"], pa.float64())})
    p, s = rec["input_type"][1:]
    if "values_unscaled" in rec:
        return pa.table({"v": pa.array([_dec(v, s) for v in rec["values_unscaled"]], pa.decimal128(p, s))})
    return pa.table({"v": pa.array([_dec(rec["value_repeated"], s)] * 64, pa.decimal128(p, s))})   # the type decides, not the row count


def _avg_types(rec):
    mk = lambda t: pa.float64() if t[0] == "Float64" else pa.decimal128(t[1], t[2])   # noqa: E731
    return mk(rec["return_type"]), (None if rec["sum_type"][0] == "Decimal256" else mk(rec["sum_type"]))


@pytest.mark.parametrize("rec", AVG, ids=[r["name"] for r in AVG])
def test_oracle_avg_cases_values_and_state_types(rec):
    """avg_cases (functions-aggregate/src/average.rs:1242-1330): return type, the sum-state type of avg_sum_data_type (:131-172) and the value;
    the Decimal128(34,0) case accumulates in Decimal256 in the reference and must be refused, not wrapped"""
    from decimal import Decimal

    from oracle import oracle
    t = _avg_case_table(rec)
    ret, sum_t = _avg_types(rec)
    aggs = [("avg", ("col", "v"), "a")]
    if sum_t is None:
        with pytest.raises(NotImplementedError, match="Decimal256"):
            oracle.aggregate(t, [], aggs, "Partial")
        return
    part = oracle.aggregate(t, [], aggs, "Partial")
    assert part.column_names == ["a[count]", "a[sum]"] and part.schema.field("a[sum]").type == sum_t and part.schema.field("a[count]").type == pa.uint64()
    single = oracle.aggregate(t, [], aggs, "Single")
    final = oracle.aggregate(part, [], aggs, "Final", return_types={"a": ret})
    want = rec["expected"] if "expected" in rec else _dec(rec["expected_unscaled"], ret.scale)
    for got in (single, final):
        assert got.schema.field("a").type == ret and got.column("a").to_pylist() == [want]


@pytest.mark.gpu
@pytest.mark.parametrize("rec", AVG, ids=[r["name"] for r in AVG])
def test_gpu_avg_cases_values_and_state_types(rec):
    from decimal import Decimal

    from datafusion_amd import _lib, ops
    from datafusion_amd.expr import col
    from datafusion_amd.table import DeviceTable
    dev = DeviceTable.from_arrow(_avg_case_table(rec))
    ret, sum_t = _avg_types(rec)
    aggs = [("avg", col("v"), "a")]
    if sum_t is None:
        for mode in ("Partial", "Single"):
            with pytest.raises(_lib.DfgpuError, match="Decimal256"):
                ops.aggregate(dev, [], aggs, mode)
        return
    part = ops.aggregate(dev, [], aggs, "Partial")
    pa_part = part.to_arrow()
    assert pa_part.column_names == ["a[count]", "a[sum]"] and pa_part.schema.field("a[sum]").type == sum_t
    want = rec["expected"] if "expected" in rec else _dec(rec["expected_unscaled"], ret.scale)
    single = ops.aggregate(dev, [], aggs, "Single").to_arrow()
    final = ops.aggregate(part, [], aggs, "Final", return_types=ops.aggregate_return_types(dev, aggs)).to_arrow()
    for got in (single, final):
        assert got.schema.field("a").type == ret and got.column("a").to_pylist() == [want]
