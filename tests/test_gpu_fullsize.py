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
import os

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


def _col_tensor(table, name, dtype):
    """zero-copy torch view of a fixed-width, non-nullable device column (test plumbing: the checker below is plain torch)"""
    import torch

    from datafusion_amd.exchange import _as_tensor
    v = table.column_view(table.index_of(name))
    width = {torch.int64: 8, torch.int32: 4}[dtype]
    per_row = 16 if str(table.schema.field(table.index_of(name)).type).startswith("decimal128") else width
    return _as_tensor(v.data, table.num_rows * per_row).view(dtype)


def _order_date_torch(idx):
    """tpch.order_date (datafusion_amd/tpch.py; csrc/tpch.hip) over an int64 CUDA tensor of order indices, in wrapping int64
    arithmetic: fmix64((i * 16 + 1) ^ seed ^ golden) % 2406 + DATE_START — an independent restatement of the generator, so every
    output row's o_orderdate can be recomputed from its l_orderkey"""
    import torch

    from datafusion_amd import tpch

    def c(v):   # 64-bit constant as the int64 with the same bits
        return v - (1 << 64) if v >= (1 << 63) else v

    def lsr33(x):
        return (x >> 33) & 0x7FFFFFFF
    x = (idx * 16 + 1) ^ c((tpch.SEED_BASE + tpch.T_ORDERS) ^ 0x9E3779B97F4A7C15)
    x = x ^ lsr33(x)
    x = x * c(0xff51afd7ed558ccd)
    x = x ^ lsr33(x)
    x = x * c(0xc4ceb9fe1a85ec53)
    x = x ^ lsr33(x)
    m = tpch.DATE_END - tpch.DATE_START + 1
    # unsigned x mod m: x = 2 * (x >>> 1) + (x & 1)
    hi = (x >> 1) & 0x7FFFFFFFFFFFFFFF
    r = ((hi % m) * 2 + (x & 1)) % m
    return (r + tpch.DATE_START).to(torch.int32)


def _assert_every_joined_row(out, lineitem=None):
    """EVERY output row of the orders x lineitem join, on the device: the build payload is the generator's function of the key
    (o_orderdate = order_date(index of l_orderkey), o_shippriority = 0); with `lineitem`: the output is the probe side row for row"""
    import torch
    key = _col_tensor(out, "l_orderkey", torch.int64) - 1
    assert bool(((key & 31) < 8).all())
    idx = (key >> 5) * 8 + (key & 31)
    got = _col_tensor(out, "o_orderdate", torch.int32)
    step = 1 << 27   # bounded temporaries
    for lo in range(0, out.num_rows, step):
        assert bool((got[lo:lo + step] == _order_date_torch(idx[lo:lo + step])).all()), lo
    assert bool((_col_tensor(out, "o_shippriority", torch.int32) == 0).all())
    del key, idx
    if lineitem is not None:
        assert out.num_rows == lineitem.num_rows
        for name in ("l_orderkey", "l_extendedprice", "l_discount"):
            a, b = _col_tensor(out, name, torch.int64), _col_tensor(lineitem, name, torch.int64)
            for lo in range(0, a.numel(), 1 << 28):
                assert torch.equal(a[lo:lo + (1 << 28)], b[lo:lo + (1 << 28)]), (name, lo)
    torch.cuda.empty_cache()


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
    # all 600 M rows, not a sample: the build payload recomputed from the key; in probe-order mode the output IS the probe side
    _assert_every_joined_row(out, lineitem if probe_mode == 0 else None)
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
    assert "join_probe_group_keys" in stats or "radix_sort_pass" in stats   # probe keys grouped by key range (grouped.hip; round 3: a radix pass)
    assert keys.num_rows == lineitem.num_rows and _sums(keys, ["l_orderkey"]) == k_sum
    keys.free()
    ops.set_options(join__grouped_probe="0")
    try:
        plain = ht.probe(sl, ["l_orderkey"], "Inner", [], ["l_orderkey"])
    finally:
        ops.set_options(join__grouped_probe=None)
    assert plain.num_rows == lineitem.num_rows and _sums(plain, ["l_orderkey"]) == k_sum
    plain.free()
    # a probe that gathers build payload: payload = f(key)
    ops.profile_enable(True)
    ops.profile_reset()
    out = ht.probe(sl, ["l_orderkey"], "Inner", ["o_orderdate"], ["l_orderkey", "l_extendedprice"])
    stats2 = ops.profile_stats()
    ops.profile_enable(False)
    ht.free()
    assert out.num_rows == lineitem.num_rows
    assert _sums(out, ["l_orderkey", "l_extendedprice"]) == _sums(lineitem, ["l_orderkey", "l_extendedprice"])
    sample = pa.concat_tables([out.slice(o, 1000).to_arrow() for o in range(0, out.num_rows - 1000, out.num_rows // 200)])
    key = sample.column("l_orderkey").to_numpy() - 1
    idx = (key >> 5) * 8 + (key & 31)
    assert (sample.column("o_orderdate").cast(pa.int32()).to_numpy() == tpch.order_date(idx.astype(np.int64))).all()
    # (round 4) this probe — payload on both sides, keys in no order, a table beyond the caches — goes through the grouped lookup
    # and comes back in probe order: every row checked on the device, build payload recomputed from the key, probe side row for row
    import torch
    assert "join_probe_grouped_lookup" in stats2 and "join_build_rank_payload" in stats2 and "join_build_rank_perm" not in stats2, sorted(stats2)
    okey = _col_tensor(out, "l_orderkey", torch.int64) - 1
    oidx = (okey >> 5) * 8 + (okey & 31)
    got = _col_tensor(out, "o_orderdate", torch.int32)
    for lo in range(0, out.num_rows, 1 << 27):
        assert bool((got[lo:lo + (1 << 27)] == _order_date_torch(oidx[lo:lo + (1 << 27)])).all()), lo
    del okey, oidx
    for name in ("l_orderkey", "l_extendedprice"):
        a, b = _col_tensor(out, name, torch.int64), _col_tensor(sl, name, torch.int64)
        assert torch.equal(a, b), name
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
    windows — by ONE grouped move into 1831 windows (round 4), or by round 3's two 64-way moves (DFGPU_AGG_GROUPED_MOVE=0) — gives exactly
    what the same node gives with one global atomic per row: groups in the same first-seen order, same counts, same first order dates,
    and the counts add up to the orders"""
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
    assert "agg_dense_accumulate_partitioned" in s1 and s1["agg_group_rows"]["calls"] == 1 and "partition_scatter" not in s1, sorted(s1)
    ops.set_options(agg__grouped_move="0")
    twice, s3 = run()
    assert "agg_dense_accumulate_partitioned" in s3 and s3["partition_scatter"]["calls"] == 2 and "agg_group_rows" not in s3, sorted(s3)
    ops.set_options(agg__partitioned_min_rows=str(2**31 - 1))
    plain, s2 = run()
    assert "agg_dense_accumulate" in s2 and "agg_dense_accumulate_partitioned" not in s2
    assert moved.num_rows == plain.num_rows == twice.num_rows
    cols = ["o_custkey", "cnt", "first_order"]
    assert _sums(moved, cols) == _sums(plain, cols) == _sums(twice, cols)
    assert _sums(moved, ["cnt"])["cnt"] == orders.num_rows
    for off in range(0, moved.num_rows - 1000, moved.num_rows // 50):          # the same rows in the same (first-seen) order
        want = plain.slice(off, 1000).to_arrow()
        assert moved.slice(off, 1000).to_arrow().equals(want) and twice.slice(off, 1000).to_arrow().equals(want)
    for t in (moved, twice, plain, orders):
        t.free()


def test_two_column_many_to_many_join_at_60m_probe_rows_one_pass_equals_two_passes():
    """the benchmark's two-column shape (1 M build rows with duplicate (Int32, Int64) keys x 60 M probe rows, flat table) at its size:
    the pairs made in ONE pass under probe_mode 4 (round 4, k_probe_pairs_single) and the counts + pairs passes of the ordered probe
    give the same number of rows and the same wrapping sums of every output column — and both equal what the host derives from the
    key counts: rows = sum over probe rows of the build rows carrying their key, and sum(v * w) over the joined rows = sum over probe rows
    of w x (the sum of v over the key's build rows), which a wrong pairing of build and probe rows cannot reproduce"""
    from datafusion_amd import ops
    from datafusion_amd.expr import col
    from datafusion_amd.table import DeviceTable
    rng = np.random.default_rng(2024)
    nb, npr, nkeys, nprobe_keys = 1_000_000, 60_000_000, 400_000, 500_000
    bk = rng.integers(0, nkeys, nb)
    v = rng.integers(0, 10**6, nb)
    build = DeviceTable.from_arrow(pa.table({"a": pa.array(bk.astype(np.int32)), "b": pa.array(bk * 7 - 2**40), "v": pa.array(v)}))
    pk = rng.integers(0, nprobe_keys, npr)
    w = rng.integers(0, 1000, npr)
    probe = DeviceTable.from_arrow(pa.table({"c": pa.array(pk.astype(np.int32)), "d": pa.array(pk * 7 - 2**40), "w": pa.array(w)}))
    rows_per_key = np.bincount(bk, minlength=nprobe_keys)
    v_per_key = np.zeros(nprobe_keys, dtype=np.int64)
    np.add.at(v_per_key, bk, v)
    want_rows = int(rows_per_key[pk].sum())
    want_vw = int((v_per_key[pk] * w).sum())
    assert 100_000_000 < want_rows < 200_000_000
    on = [("a", "c"), ("b", "d")]
    seen = {}
    for mode in (4, 0):
        ops.profile_enable(True)
        ops.profile_reset()
        out = ops.hash_join(build, probe, on, "Inner", probe_mode=mode)
        stats = ops.profile_stats()
        ops.profile_enable(False)
        assert ("join_probe_pairs_single" in stats) == (mode == 4) and ("join_probe_count" in stats) == (mode == 0), sorted(stats)
        sums = _sums(out, ["a", "b", "v", "c", "d", "w"])
        vw = ops.project(out, [(col("v") * col("w"), "vw")])
        sums["vw"] = _sums(vw, ["vw"])["vw"]
        vw.free()
        out.free()
        assert sums["n"] == want_rows and sums["vw"] == want_vw, (mode, sums, want_rows, want_vw)
        seen[mode] = sums
    assert seen[4] == seen[0]
    build.free()
    probe.free()


def test_sf300_q3_on_one_gpu():
    """BASELINE config 5's N = 1 anchor: TPC-H Q3 at SF300 (45 M customers, 450 M orders, 1.8 G lineitem rows; ~91 GB of referenced
    columns) on ONE MI355X, as the plan of tpch/plans/q3.slt.part:60-76.  No oracle finishes this: the two executions of the plan —
    FilterExecs fused into the probes vs operator by operator (different kernels: row-masked probes vs compaction + plain probes) —
    must agree row for row on the ten result rows and on every intermediate row count; the group count SURVEY §8 a9 asks to be
    measured is asserted against the SF100 measurement scaled (1.1 groups per 100 lineitem rows surviving both joins' filters is a
    property of the date predicates, not of the scale); the revenue of the top row is re-derived from lineitem with a filter on its
    l_orderkey alone (a third path: FilterExec + ungrouped aggregate)"""
    from datafusion_amd import ops, queries
    from datafusion_amd.expr import col, lit
    sf = 300.0
    customer = ops.tpch_customer(sf)
    orders = ops.tpch_orders(sf).select(["o_orderkey", "o_custkey", "o_orderdate", "o_shippriority"])
    lineitem = ops.tpch_lineitem(sf).select(["l_orderkey", "l_extendedprice", "l_discount", "l_shipdate"])
    assert (customer.num_rows, orders.num_rows) == (45_000_000, 450_000_000) and 1_795_000_000 < lineitem.num_rows < 1_805_000_000
    s1, s2 = {}, {}
    fused = queries.q3(customer, orders, lineitem, stats=s1, fused=True).to_arrow()
    plain = queries.q3(customer, orders, lineitem, stats=s2, fused=False).to_arrow()
    assert fused.num_rows == 10 and fused.equals(plain), (fused.to_pylist(), plain.to_pylist())
    for k in ("customer_filtered", "semi_join", "join", "groups"):
        assert s1[k] == s2[k], (k, s1, s2)
    print("Q3 SF300 on one GPU:", {**s2, **s1})
    # measured at SF100: 3.0 M joined rows in 1.13 M groups; the ratios are scale-free
    assert 0.30 < s1["groups"] / s1["join"] < 0.45 and 9_000 * sf < s1["groups"] < 13_000 * sf, s1
    rev = fused.column("revenue").to_pylist()
    assert rev == sorted(rev, reverse=True)
    top_key = fused.column("l_orderkey").to_pylist()[0]
    pred = col("l_orderkey").eq(lit(top_key, pa.int64())).and_(col("l_shipdate") > lit(queries.DATE_Q3, pa.date32()))
    again = ops.aggregate(lineitem, [], [("sum", col("l_extendedprice") * (queries.ONE - col("l_discount")), "revenue")], "Single", predicate=pred).to_arrow()
    assert again.column("revenue").to_pylist()[0] == rev[0]
    for t in (customer, orders, lineitem):
        t.free()


# ------------------------------------------------------------------------------------------------------------------------------------
# BASELINE configs 4 and 5 AT THEIR SIZES against the oracle (round-5 verdict, weak 1): the GPU leg's own tables copied to the host,
# the reference's pinned plan on the oracle's operators (oracle/plans.py: target_partitions row ranges, one thread each), every value
# of the result compared bit for bit.  Host RAM: Q1 SF100 holds 42 GB of columns + 39 GB of filtered rows, Q3 SF300 91 GB + 2 x 39 GB.

def _oracle_threads():
    import bench
    q = bench.cpu_quota()
    n = os.cpu_count() or 1
    return max(1, int(min(n, q) if q else n))


def _need_host_gib(gib):
    import bench
    avail = bench.host_memory_available()
    if avail is not None and avail < gib * 2**30:
        pytest.skip(f"the oracle leg needs ~{gib} GiB of host RAM, {avail / 2**30:.0f} GiB available")


def test_sf100_q1_equals_oracle():
    """TPC-H Q1 at SF100 (600 M lineitem rows): the 4 groups x 10 columns of queries.q1 — Decimal128 sums, the truncating Decimal128 averages,
    counts — equal the oracle's Partial -> FinalPartitioned plan over the same rows, digit for digit; so does the filter's row count
    (answers' shape: tpch/answers/q1.slt.part:42-45, types q1.slt.part:45-46)"""
    from datafusion_amd import ops, queries
    from oracle import plans
    _need_host_gib(110)
    li = ops.tpch_lineitem(SF).select(["l_quantity", "l_extendedprice", "l_discount", "l_tax", "l_returnflag", "l_linestatus", "l_shipdate"])
    assert li.num_rows == 599_960_064
    got = queries.q1(li).to_arrow()
    host = li.to_arrow()
    li.free()
    st = {}
    exp = plans.run_q1(host, _oracle_threads(), st)
    assert got.num_rows == exp.num_rows == 4 and got.column_names == exp.column_names
    assert got.schema.field("sum_charge").type == pa.decimal128(38, 6) and got.schema.field("avg_disc").type == pa.decimal128(19, 6)
    assert got.to_pylist() == exp.to_pylist()
    assert sum(got.column("count_order").to_pylist()) == st["filtered"]


def test_sf300_q3_equals_oracle():
    """TPC-H Q3 at SF300 (45 M customers, 450 M orders, 1.8 G lineitem rows) on one GPU: the ten result rows (l_orderkey, revenue
    Decimal128(38,4), o_orderdate, o_shippriority; revenue DESC, o_orderdate ASC) and the four intermediate row counts equal the oracle's
    partitioned plan over the same tables (q3.slt.part:44-76)"""
    from datafusion_amd import ops, queries
    from oracle import plans
    _need_host_gib(230)
    sf = 300.0
    customer = ops.tpch_customer(sf).select(["c_custkey", "c_mktsegment"])
    orders = ops.tpch_orders(sf).select(["o_orderkey", "o_custkey", "o_orderdate", "o_shippriority"])
    lineitem = ops.tpch_lineitem(sf).select(["l_orderkey", "l_extendedprice", "l_discount", "l_shipdate"])
    s1 = {}
    got = queries.q3(customer, orders, lineitem, stats=s1).to_arrow()
    host = []
    for t in (customer, orders, lineitem):
        host.append(t.to_arrow())
        t.free()
    s2 = {}
    exp = plans.run_q3(*host, _oracle_threads(), s2)
    assert got.num_rows == exp.num_rows == 10 and got.column_names == exp.column_names
    assert got.to_pylist() == exp.to_pylist()
    assert {k: s1[k] for k in ("customer_filtered", "semi_join", "join", "groups")} == s2
