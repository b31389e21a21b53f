"""One LARGE exact check (BASELINE config 3 at SF10: 15 M orders x 60 M lineitem rows): the join's output compared ROW BY ROW with a
host restatement, every column of every row, and TPC-H Q3 end to end against the oracle's operators over the same tables — the
size-independent properties of tests/test_gpu_fullsize.py (SF100) would not notice, say, an output row pairing the right key with
a neighbour's payload in a way that preserves column sums.

The inputs are generated on the device (tpch.hip, the generator the benchmarks use) and exported once; the expected join is
`take(orders, searchsorted(o_orderkey, l_orderkey))` next to the lineitem columns (numpy / Arrow on the host), cross-checked by the
oracle's partitioned hash join (pair count).  Probe-order flavours compare positionally; the unordered flavours as multisets of
whole rows (a 64-bit mix of every column of the row, sorted)."""
import numpy as np
import pyarrow as pa
import pytest

pytestmark = pytest.mark.gpu

SF = 10.0
BUILD_COLS, PROBE_COLS = ["o_orderdate", "o_shippriority"], ["l_orderkey", "l_extendedprice", "l_discount"]


@pytest.fixture(scope="module")
def sf10():
    from datafusion_amd import ops
    orders = ops.tpch_orders(SF).select(["o_orderkey", "o_custkey", "o_orderdate", "o_shippriority"])
    lineitem = ops.tpch_lineitem(SF).select(["l_orderkey", "l_extendedprice", "l_discount", "l_shipdate"])
    customer = ops.tpch_customer(SF)
    host = dict(orders=orders.to_arrow(), lineitem=lineitem.to_arrow(), customer=customer.to_arrow())
    ok, lk = host["orders"].column("o_orderkey").to_numpy(), host["lineitem"].column("l_orderkey").to_numpy()
    assert (np.diff(ok) > 0).all()
    pos = np.searchsorted(ok, lk)
    assert (ok[pos] == lk).all()                                           # every line has its order
    expected = pa.table({**{c: host["orders"].column(c).take(pa.array(pos)) for c in BUILD_COLS}, **{c: host["lineitem"].column(c) for c in PROBE_COLS}})
    yield dict(orders=orders, lineitem=lineitem, customer=customer, host=host, expected=expected)
    for t in (orders, lineitem, customer):
        t.free()


def _row_mix(table: pa.Table) -> np.ndarray:
    """one 64-bit value per row mixing every column of the row (Decimal128 through both halves)"""
    h = np.zeros(table.num_rows, np.uint64)
    for c in table.columns:
        a = c.combine_chunks()
        if pa.types.is_decimal128(a.type):
            words = np.frombuffer(a.buffers()[1], np.uint64, 2 * len(a)).reshape(-1, 2)
            parts = [words[:, 0], words[:, 1]]
        elif pa.types.is_date32(a.type):
            parts = [a.cast(pa.int32()).to_numpy().astype(np.uint64)]
        else:
            parts = [a.to_numpy().astype(np.int64).view(np.uint64)]
        for p in parts:
            h = (h ^ p) * np.uint64(0x9E3779B97F4A7C15)
            h ^= h >> np.uint64(29)
    return h


@pytest.mark.parametrize("probe_mode", [0, 1], ids=["placed_probe_order", "two_pass_probe_order"])
def test_sf10_join_equals_the_host_join_row_by_row(sf10, probe_mode):
    from datafusion_amd import ops
    from oracle import oracle
    ht = ops.JoinHashTable(sf10["orders"], ["o_orderkey"], probe_mode=probe_mode)
    out = ht.probe(sf10["lineitem"], ["l_orderkey"], "Inner", BUILD_COLS, PROBE_COLS)
    ht.free()
    got = out.to_arrow()
    out.free()
    exp = sf10["expected"]
    assert got.num_rows == exp.num_rows > 59_000_000
    for c in exp.column_names:                                             # probe order: row i of the output is probe row i
        assert got.column(c).combine_chunks().equals(exp.column(c).combine_chunks()), c
    pairs, _ = oracle.partitioned_inner_join_i64(sf10["host"]["orders"].column("o_orderkey").to_numpy(), sf10["host"]["lineitem"].column("l_orderkey").to_numpy(), 32)
    assert pairs == got.num_rows


@pytest.mark.parametrize("flavour", ["single_pass_unordered", "hash_map", "radix_lds"])
def test_sf10_join_unordered_flavours_hold_the_same_rows(sf10, flavour):
    from datafusion_amd import ops
    kw = {"single_pass_unordered": dict(probe_mode=3), "hash_map": dict(table_mode=1, probe_mode=3), "radix_lds": dict(table_mode=4, probe_mode=4)}[flavour]
    ht = ops.JoinHashTable(sf10["orders"], ["o_orderkey"], **kw)
    out = ht.probe(sf10["lineitem"], ["l_orderkey"], "Inner", BUILD_COLS, PROBE_COLS)
    ht.free()
    got = out.to_arrow().select(sf10["expected"].column_names)
    out.free()
    assert got.num_rows == sf10["expected"].num_rows
    assert np.array_equal(np.sort(_row_mix(got)), np.sort(_row_mix(sf10["expected"])))


@pytest.mark.parametrize("fused", [True, False])
def test_sf10_q3_equals_the_oracle(sf10, fused):
    """the reference's pinned Q3 plan over the same 15 M / 60 M-row tables: the oracle's operators on the host vs the device"""
    from datafusion_amd import queries, tpch_plans as T
    from datafusion_amd.expr import lit
    from tests import plan_oracle
    if "q3" not in sf10:
        h = sf10["host"]
        sf10["q3"] = plan_oracle.collect(T.q3_plan(h["customer"], h["orders"], h["lineitem"], segment_literal=lit(queries.SEGMENT_BUILDING, pa.uint8())))
    got = queries.q3(sf10["customer"], sf10["orders"], sf10["lineitem"], fused=fused).to_arrow()
    assert got.to_pylist() == sf10["q3"].to_pylist()
