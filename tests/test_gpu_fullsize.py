"""BASELINE.json's full sizes (SF100: 150 M orders, 600 M lineitem rows), where the CPU oracle cannot finish in
seconds: size-independent properties of the domain instead of row-by-row comparison.

  * join: every lineitem row matches exactly one order => |output| = |probe|; the probe payload is a permutation
    of the input (wrapping column sums equal, computed by the GPU aggregate without GROUP BY — a different kernel
    path); the gathered build payload is a known function of the key (o_orderdate = tpch.order_date(index of the
    key)), checked on a sample of output rows; joining with the FilterExec fused into the probe gives exactly the
    rows of filter-then-join (count + checksums).
  * Q1: the per-group counts add up to the rows passing the predicate; per-group sums add up to the ungrouped sum.
  * sort: output is a permutation (checksum) and is ordered (adjacent-pair check on device).
"""
import datetime

import numpy as np
import pyarrow as pa
import pytest

pytestmark = pytest.mark.gpu

SF = 100.0


def _sums(table, cols):
    """wrapping column sums through the GPU aggregate (no GROUP BY); Date32 columns are summed as their day numbers"""
    from datafusion_amd import ops
    from datafusion_amd.expr import col
    dates = {f.name for f in table.schema if f.type == pa.date32()}
    out = ops.aggregate(table, [], [("sum", col(c).cast(pa.int32()) if c in dates else col(c), c) for c in cols] + [("count", None, "n")], "Single")
    row = out.to_arrow().to_pylist()[0]
    out.free()
    return row


@pytest.fixture(scope="module")
def tables():
    from datafusion_amd import ops
    orders = ops.tpch_orders(SF).select(["o_orderkey", "o_orderdate", "o_shippriority"])
    lineitem = ops.tpch_lineitem(SF).select(["l_orderkey", "l_extendedprice", "l_discount", "l_shipdate"])
    yield orders, lineitem
    orders.free()
    lineitem.free()


@pytest.mark.parametrize("probe_mode", [3, 0], ids=["single_pass_unordered", "two_pass_ordered"])
def test_sf100_join_properties(tables, probe_mode):
    from datafusion_amd import ops, tpch
    orders, lineitem = tables
    ht = ops.JoinHashTable(orders, ["o_orderkey"], probe_mode=probe_mode)
    out = ht.probe(lineitem, ["l_orderkey"], "Inner", ["o_orderdate", "o_shippriority"], ["l_orderkey", "l_extendedprice", "l_discount"])
    ht.free()
    assert out.num_rows == lineitem.num_rows == 599_960_064
    cols = ["l_orderkey", "l_extendedprice", "l_discount"]
    assert _sums(out, cols) == _sums(lineitem, cols)
    # build payload = f(key) on a sample: key = (i >> 3) * 32 + (i & 7) + 1  =>  i = ((key - 1) >> 5) * 8 + ((key - 1) & 31)
    sample = pa.concat_tables([out.slice(o, 1000).to_arrow() for o in range(0, out.num_rows - 1000, out.num_rows // 200)])
    key = sample.column("l_orderkey").to_numpy() - 1
    idx = (key >> 5) * 8 + (key & 31)
    assert (key & 31 < 8).all()
    assert (sample.column("o_orderdate").cast(pa.int32()).to_numpy() == tpch.order_date(idx.astype(np.int64))).all()
    assert (sample.column("o_shippriority").to_numpy() == 0).all()
    if probe_mode == 0:   # probe order preserved: l_orderkey ascending like the input
        k = sample.slice(0, 1000).column("l_orderkey").to_numpy()
        assert (np.diff(k) >= 0).all()
    out.free()


def test_sf100_join_over_keys_in_no_order(tables):
    """the same join with both sides in NO key order (orders sorted by o_orderdate with ties broken by a hash-like key, lineitem by
    l_extendedprice): the build guesses nothing (its sample does not ascend), marks bytes instead of bits, leaves the row permutation
    unbuilt until a probe wants build rows; a key-only probe whose order nobody observes is grouped by key range first.  Properties:
    every probe row still finds its order; key checksums survive; the build payload is still the function of the key it was
    generated as; the grouped and the ungrouped probe hold the same rows."""
    import os

    from datafusion_amd import ops, tpch
    orders, lineitem = tables
    so = ops.sort(orders, [("o_orderdate", False, False), ("o_orderkey", True, False)])
    sl = ops.sort(lineitem.select(["l_orderkey", "l_extendedprice"]), [("l_extendedprice", False, False)])
    k_sum = _sums(lineitem, ["l_orderkey"])
    ops.profile_enable(True)
    ops.profile_reset()
    ht = ops.JoinHashTable(so, ["o_orderkey"], probe_mode=4)
    keys = ht.probe(sl, ["l_orderkey"], "Inner", [], ["l_orderkey"])
    stats = ops.profile_stats()
    ops.profile_enable(False)
    assert ht.info().table_kind == 2 and "join_build_key_stats" in stats and "join_build_speculation_missed" not in stats   # rank map, measured statistics
    assert "radix_sort_pass" in stats                                      # probe keys grouped by key range
    assert keys.num_rows == lineitem.num_rows and _sums(keys, ["l_orderkey"]) == k_sum
    keys.free()
    os.environ["DFGPU_JOIN_GROUPED_PROBE"] = "0"
    try:
        plain = ht.probe(sl, ["l_orderkey"], "Inner", [], ["l_orderkey"])
    finally:
        del os.environ["DFGPU_JOIN_GROUPED_PROBE"]
    assert plain.num_rows == lineitem.num_rows and _sums(plain, ["l_orderkey"]) == k_sum
    plain.free()
    # a probe that gathers build payload: the permutation is built now; payload = f(key) on a sample
    out = ht.probe(sl, ["l_orderkey"], "Inner", ["o_orderdate"], ["l_orderkey", "l_extendedprice"])
    ht.free()
    assert out.num_rows == lineitem.num_rows
    assert _sums(out, ["l_orderkey", "l_extendedprice"]) == _sums(lineitem, ["l_orderkey", "l_extendedprice"])
    sample = pa.concat_tables([out.slice(o, 1000).to_arrow() for o in range(0, out.num_rows - 1000, out.num_rows // 200)])
    key = sample.column("l_orderkey").to_numpy() - 1
    idx = (key >> 5) * 8 + (key & 31)
    assert (sample.column("o_orderdate").cast(pa.int32()).to_numpy() == tpch.order_date(idx.astype(np.int64))).all()
    for t in (out, so, sl):
        t.free()


def test_sf100_filter_fused_into_probe_equals_filter_then_join(tables):
    from datafusion_amd import ops, queries
    from datafusion_amd.expr import col, lit
    orders, lineitem = tables
    pred = col("l_shipdate") > lit(queries.DATE_Q3, pa.date32())
    ht = ops.JoinHashTable(orders, ["o_orderkey"], probe_mode=3)
    pcols = ["l_orderkey", "l_extendedprice", "l_discount"]
    fused = ht.probe(lineitem, ["l_orderkey"], "Inner", ["o_orderdate", "o_shippriority"], pcols, predicate=pred)
    f = ops.filter(lineitem, pred, pcols)
    plain = ht.probe(f, ["l_orderkey"], "Inner", ["o_orderdate", "o_shippriority"], pcols)
    ht.free()
    assert fused.num_rows == plain.num_rows == f.num_rows
    cols = pcols + ["o_orderdate", "o_shippriority"]
    assert _sums(fused, cols) == _sums(plain, cols)
    for t in (fused, plain, f):
        t.free()


def test_sf100_q1_group_totals_add_up():
    from datafusion_amd import ops, queries
    from datafusion_amd.expr import col, lit
    li = ops.tpch_lineitem(SF)
    pred = col("l_shipdate") <= lit(queries.DATE_Q1, pa.date32())
    q1 = queries.q1(li).to_arrow()
    assert q1.num_rows == 4
    f = ops.filter(li, pred, ["l_quantity", "l_extendedprice", "l_discount"])
    tot = _sums(f, ["l_quantity", "l_extendedprice"])
    assert sum(q1.column("count_order").to_pylist()) == f.num_rows == tot["n"]
    assert sum(q1.column("sum_qty").to_pylist()) == tot["l_quantity"]
    assert sum(q1.column("sum_base_price").to_pylist()) == tot["l_extendedprice"]
    # sum_disc_price: same expression, ungrouped, through the specialised node
    ungrouped = ops.aggregate(li, [], [("sum", col("l_extendedprice") * (queries.ONE - col("l_discount")), "s")], "Single", predicate=pred).to_arrow()
    assert sum(q1.column("sum_disc_price").to_pylist()) == ungrouped.column("s").to_pylist()[0]
    f.free()
    li.free()


def test_sf100_group_by_orderkey_dense_node_properties():
    from datafusion_amd import ops
    from datafusion_amd.expr import col
    li = ops.tpch_lineitem(SF).select(["l_orderkey", "l_extendedprice"])
    g = ops.aggregate(li, [(col("l_orderkey"), "l_orderkey")], [("sum", col("l_extendedprice"), "s"), ("count", None, "c")], "Single")
    assert g.num_rows == 150_000_000            # every order has at least one line
    tot, gt = _sums(li, ["l_extendedprice"]), _sums(g, ["s", "c", "l_orderkey"])
    assert gt["s"] == tot["l_extendedprice"] and gt["c"] == li.num_rows
    head = g.slice(0, 100_000).to_arrow()       # first-seen order = key order for input sorted by key
    k = head.column("l_orderkey").to_numpy()
    assert (np.diff(k) > 0).all() and 1 <= head.column("c").to_numpy().min() and head.column("c").to_numpy().max() <= 7
    g.free()
    li.free()


def test_sf100_sort_is_an_ordered_permutation():
    from datafusion_amd import ops
    o = ops.tpch_orders(SF).select(["o_orderkey", "o_orderdate", "o_custkey"])
    s = ops.sort(o, [("o_orderdate", False, False), ("o_orderkey", True, False)])
    assert s.num_rows == o.num_rows
    assert _sums(s, ["o_orderkey", "o_custkey"]) == _sums(o, ["o_orderkey", "o_custkey"])
    for off in (0, 75_000_000, 149_000_000):
        w = s.slice(off, 1_000_000).to_arrow()
        d, k = w.column("o_orderdate").cast(pa.int32()).to_numpy().astype(np.int64), w.column("o_orderkey").to_numpy()
        assert ((np.diff(d) > 0) | ((np.diff(d) == 0) & (np.diff(k) < 0))).all()
    s.free()
    o.free()


def test_sf100_group_by_custkey_partitioned_equals_global_atomics(monkeypatch):
    """Q13's shape at SF100 (150 M orders, 10 M customers with orders): the dense-key node with its rows moved into LDS-sized key
    windows (two moves) gives exactly what the same node gives with one global atomic per row — groups in the same first-seen order,
    same counts, same first order dates — and the counts add up to the orders"""
    from datafusion_amd import ops
    from datafusion_amd.expr import col
    orders = ops.tpch_orders(SF).select(["o_custkey", "o_orderdate"])
    aggs = [("count", None, "cnt"), ("min", col("o_orderdate"), "first_order")]

    def run():
        ops.profile_enable(True)
        ops.profile_reset()
        out = ops.aggregate(orders, [(col("o_custkey"), "o_custkey")], aggs, "Single")
        stats = ops.profile_stats()
        ops.profile_enable(False)
        return out, stats
    moved, s1 = run()
    assert "agg_dense_accumulate_partitioned" in s1 and s1["partition_scatter"]["calls"] == 2
    monkeypatch.setenv("DFGPU_AGG_PARTITIONED_MIN_ROWS", str(2**31 - 1))
    plain, s2 = run()
    assert "agg_dense_accumulate" in s2 and "agg_dense_accumulate_partitioned" not in s2
    assert moved.num_rows == plain.num_rows
    cols = ["o_custkey", "cnt", "first_order"]
    assert _sums(moved, cols) == _sums(plain, cols)
    assert _sums(moved, ["cnt"])["cnt"] == orders.num_rows
    for off in range(0, moved.num_rows - 1000, moved.num_rows // 50):          # the same rows in the same (first-seen) order
        assert moved.slice(off, 1000).to_arrow().equals(plain.slice(off, 1000).to_arrow())
    for t in (moved, plain, orders):
        t.free()
