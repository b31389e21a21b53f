"""K1-K5 parity: HashJoinExec on the GPU vs (a) the reference's own snapshot tests and (b) the
CPU oracle on random inputs — every JoinType, both table kinds (direct-address / chained hash),
NULL keys under both NullEquality settings, duplicates, forced hash collisions."""
import datetime

import numpy as np
import pyarrow as pa
import pytest

from tests.util import assert_tables_equal, i32_table, load_golden, random_table, rows, sorted_rows

pytestmark = pytest.mark.gpu

CASES = load_golden("hash_join_exec.json")
ALL_TYPES = ["Inner", "Left", "Right", "Full", "LeftSemi", "RightSemi", "LeftAnti", "RightAnti", "LeftMark", "RightMark"]


def gpu_join(left, right, on, join_type, null_equality="NullEqualsNothing", **opts):
    from datafusion_amd import ops
    from datafusion_amd.table import DeviceTable
    return ops.hash_join(DeviceTable.from_arrow(left), DeviceTable.from_arrow(right), on, join_type, null_equality, **opts).to_arrow()


# the reference runs every case with PHJ on/off (exec.rs:2929-2963); plus its force_hash_collisions CI job
@pytest.mark.parametrize("opts", [dict(table_mode=0), dict(table_mode=1), dict(table_mode=1, force_hash_collisions=True), dict(table_mode=4),
                                  dict(table_mode=4, force_hash_collisions=True), dict(table_mode=5), dict(table_mode=5, force_hash_collisions=True)],
                         ids=["phj_auto", "hash_map", "forced_collisions", "radix_lds", "radix_lds_forced_collisions", "flat", "flat_forced_collisions"])
@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_reference_snapshots(case, opts):
    left = i32_table(case["left"]["columns"], case["left"]["data"], case["left"]["repeat"])
    right = i32_table(case["right"]["columns"], case["right"]["data"], case["right"]["repeat"])
    out = gpu_join(left, right, [tuple(p) for p in case["on"]], case["join_type"], case["null_equality"], **opts)
    assert out.column_names == case["expected_columns"], case["source"]
    key = lambda row: tuple((v is None, 0 if v is None else v) for v in row)
    assert sorted_rows(out) == sorted([tuple(r) for r in case["expected_rows"]], key=key), case["source"]


@pytest.mark.parametrize("join_type", ALL_TYPES)
@pytest.mark.parametrize("mode", [0, 1, 4, 5])
def test_random_vs_oracle_all_join_types(join_type, mode):
    from oracle import oracle
    rng = np.random.default_rng(11)
    left = random_table(rng, 3000, {"a": (pa.int64(), 0, 800), "x": (pa.decimal128(15, 2), 0, 10**6), "y": (pa.int32(), 0, 100)}, null_frac=0.05)
    right = random_table(rng, 7000, {"b": (pa.int64(), 0, 1000), "z": (pa.float64(), 0, 1000), "w": (pa.date32(), 8000, 9000)}, null_frac=0.05)
    for ne in ("NullEqualsNothing", "NullEqualsNull"):
        got = gpu_join(left, right, [("a", "b")], join_type, ne, table_mode=mode)
        exp = oracle.hash_join(left, right, [("a", "b")], join_type, ne)
        assert_tables_equal(got, exp)


def test_unique_build_fast_path_preserves_probe_order():
    """N:1 join (the TPC-H shape): fused compaction+gather path; output in probe order like the
    reference ("Inner join output is expected to preserve both inputs order", exec.rs:3349)"""
    from oracle import oracle
    rng = np.random.default_rng(5)
    build = pa.table({"k": pa.array(rng.permutation(5000)[:4000] * 3, type=pa.int64()), "v": pa.array(np.arange(4000), type=pa.int32())})
    probe = random_table(rng, 50_001, {"k2": (pa.int64(), -10, 15100), "p": (pa.decimal128(15, 2), 0, 10**7)})
    exp = oracle.hash_join(build, probe, [("k", "k2")], "Inner")
    for mode in (0, 1, 2):
        for probe_mode in (1, 2):   # two-pass (lookup -> scan -> materialise) and fused single pass
            got = gpu_join(build, probe, [("k", "k2")], "Inner", table_mode=mode, probe_mode=probe_mode)
            assert_tables_equal(got, exp, ordered=True)
        got = gpu_join(build, probe, [("k", "k2")], "Inner", table_mode=mode, probe_mode=3)   # unordered single pass
        assert_tables_equal(got, exp, ordered=False)


@pytest.mark.parametrize("np_rows", [1, 63, 1024, 1025, 70_000, 3_000_001])
@pytest.mark.parametrize("table_mode", [0, 1, 2])
def test_single_pass_probe_lookback_many_tiles(np_rows, table_mode):
    """single-pass probe: 1024-row tiles chained by decoupled look-back; 3 M rows = 2930 tiles, so
    look-back windows span > 64 predecessors; ragged tails; selectivity ~ 50 %; result in probe
    order and identical to the two-pass path and the oracle"""
    from oracle import oracle
    rng = np.random.default_rng(np_rows)
    nb = 40_000
    build = pa.table({"k": pa.array(rng.permutation(2 * nb)[:nb].astype(np.int64) * 2, type=pa.int64()),
                      "v": pa.array(np.arange(nb), type=pa.int32()), "d": pa.array(np.arange(nb) + 8000, type=pa.int32()).cast(pa.date32())})
    probe = random_table(rng, np_rows, {"k2": (pa.int64(), -5, 4 * nb + 5), "p": (pa.decimal128(15, 2), 0, 10**7), "q": (pa.float64(), 0, 1)})
    exp = oracle.hash_join(build, probe, [("k", "k2")], "Inner")
    one = gpu_join(build, probe, [("k", "k2")], "Inner", table_mode=table_mode, probe_mode=2)
    two = gpu_join(build, probe, [("k", "k2")], "Inner", table_mode=table_mode, probe_mode=1)
    assert_tables_equal(one, exp, ordered=True)
    assert_tables_equal(two, exp, ordered=True)
    assert_same_rows_any_tile_order(gpu_join(build, probe, [("k", "k2")], "Inner", table_mode=table_mode, probe_mode=3), exp)
    for jt in ("RightSemi", "RightAnti"):
        e = oracle.hash_join(build, probe, [("k", "k2")], jt)
        assert_tables_equal(gpu_join(build, probe, [("k", "k2")], jt, table_mode=table_mode, probe_mode=2), e, ordered=True)
        assert_tables_equal(gpu_join(build, probe, [("k", "k2")], jt, table_mode=table_mode, probe_mode=1), e, ordered=True)
        assert_same_rows_any_tile_order(gpu_join(build, probe, [("k", "k2")], jt, table_mode=table_mode, probe_mode=3), e)


def assert_same_rows_any_tile_order(got, exp):
    """unordered single-pass output: the same multiset of rows (tiles land in claim order)"""
    assert got.num_rows == exp.num_rows and got.schema.types == exp.schema.types
    names = got.column_names
    key = [(n, "ascending") for n in names]
    assert got.sort_by(key).equals(exp.rename_columns(names).sort_by(key))


def test_single_pass_probe_rejects_inapplicable():
    """duplicate build keys (M:N) cannot use the single-pass kernel: explicit request fails loudly, auto falls back"""
    from datafusion_amd import _lib
    build = pa.table({"k": pa.array([1, 1, 2], type=pa.int64()), "v": pa.array([1, 2, 3], type=pa.int32())})
    probe = pa.table({"k2": pa.array([1, 2, 3], type=pa.int64())})
    for pm in (2, 3):
        with pytest.raises(_lib.DfgpuError):
            gpu_join(build, probe, [("k", "k2")], "Inner", probe_mode=pm)
    assert gpu_join(build, probe, [("k", "k2")], "Inner").num_rows == 3


def test_multi_column_and_decimal_keys():
    from oracle import oracle
    rng = np.random.default_rng(8)
    left = random_table(rng, 2000, {"a": (pa.int32(), 0, 30), "b": (pa.decimal128(15, 2), 0, 20), "v": (pa.int64(), 0, 10**9)}, null_frac=0.03)
    right = random_table(rng, 5000, {"a": (pa.int32(), 0, 30), "b": (pa.decimal128(15, 2), 0, 20), "w": (pa.int64(), 0, 10**9)}, null_frac=0.03)
    for jt in ("Inner", "Left", "RightSemi", "RightAnti", "Full"):
        exp = oracle.hash_join(left, right, [("a", "a"), ("b", "b")], jt)
        for mode in (0, 1, 4):   # (Int32, Decimal128) = 20 bytes: beyond what the flat table packs, auto = chained
            for ne in ("NullEqualsNothing", "NullEqualsNull"):
                got = gpu_join(left, right, [("a", "a"), ("b", "b")], jt, ne, table_mode=mode)
                assert_tables_equal(got, exp if ne == "NullEqualsNothing" else oracle.hash_join(left, right, [("a", "a"), ("b", "b")], jt, ne))


@pytest.mark.parametrize("keys", ["i32_i64", "i32_i32", "d128", "date_u8ish_f64", "i64_i64"])
@pytest.mark.parametrize("join_type", ALL_TYPES)
def test_flat_table_packed_keys(keys, join_type):
    """hash table with the keys inline (round 4): every key set that packs into 16 bytes — one or several columns, NULLs on both
    sides under both NullEquality settings (a NULL flag per nullable column rides in the packed key), duplicates on both sides —
    gives the oracle's rows for every JoinType, also with every hash forced to 0 (one long run of slots)"""
    from datafusion_amd import _lib, ops
    from datafusion_amd.table import DeviceTable
    from oracle import oracle
    rng = np.random.default_rng(len(keys) * 31 + len(join_type))
    spec = {"i32_i64": [(pa.int32(), -20, 20), (pa.int64(), -(2**40), -(2**40) + 25)], "i32_i32": [(pa.int32(), -5, 40), (pa.int32(), 0, 12)],
            "d128": [(pa.decimal128(20, 2), -300, 300)], "date_u8ish_f64": [(pa.date32(), 9000, 9020), (pa.float64(), 0, 1)],
            "i64_i64": [(pa.int64(), -9, 9), (pa.int64(), 2**62, 2**62 + 30)]}[keys]
    lcols = {f"k{i}": t for i, t in enumerate(spec)}
    rcols = {f"j{i}": t for i, t in enumerate(spec)}
    left = random_table(rng, 2500, {**lcols, "v": (pa.int64(), 0, 10**9)}, null_frac=0.04)
    right = random_table(rng, 6000, {**rcols, "w": (pa.decimal128(15, 2), 0, 10**6)}, null_frac=0.04)
    if keys == "date_u8ish_f64":   # few distinct doubles, -0.0 among them (hash_utils.rs:258-276: -0.0 and +0.0 are one key)
        vals = np.array([0.0, -0.0, 1.5, -2.25, 1e300])
        for t, name, n in ((left, "k1", 2500), (right, "j1", 6000)):
            arr = pa.array(vals[rng.integers(0, 5, n)], mask=rng.random(n) < 0.04)
            t = t.set_column(t.schema.get_field_index(name), name, arr)
            if name == "k1":
                left = t
            else:
                right = t
    on = [(f"k{i}", f"j{i}") for i in range(len(spec))]
    assert ops.JoinHashTable(DeviceTable.from_arrow(left), [a for a, _ in on], table_mode=5).info().table_kind == (5 if keys in ("i32_i64", "d128", "i64_i64", "date_u8ish_f64") else 4)
    for ne in ("NullEqualsNothing", "NullEqualsNull"):
        exp = oracle.hash_join(left, right, on, join_type, ne)
        assert_tables_equal(gpu_join(left, right, on, join_type, ne), exp)   # auto: the flat table, or the chained one when the keys do not pack
        if ne == "NullEqualsNull" and keys in ("d128", "i64_i64"):
            # 16 bytes of key values leave no room for the NULL flags NULL == NULL needs: not packable, asked for by name it says so
            with pytest.raises(_lib.DfgpuError):
                gpu_join(left, right, on, join_type, ne, table_mode=5)
            continue
        assert_tables_equal(gpu_join(left, right, on, join_type, ne, table_mode=5), exp)
        assert_tables_equal(gpu_join(left, right, on, join_type, ne, table_mode=5, force_hash_collisions=True), exp)


@pytest.mark.parametrize("shape", ["uniform_duplicates", "sample_sees_no_match", "nulls_both_equalities"])
def test_flat_table_many_to_many_pairs_in_one_pass(shape):
    """round 4: an INNER join over a flat table whose order nobody observes (probe_mode 4) makes its pairs in ONE pass — every tile of
    probe rows reserves its pairs at a device-wide cursor (join.hip k_probe_pairs_single).  Same rows as the oracle (as multisets) over
    duplicate keys on both sides and a two-column key; a probe side whose SAMPLED rows (every 2nd here) match nothing while the others
    match three build rows each makes the first pair buffer too small: the cursor has counted the exact size and a second pass fits;
    NULL keys under both NullEquality settings"""
    from datafusion_amd import ops
    from datafusion_amd.table import DeviceTable
    from oracle import oracle
    rng = np.random.default_rng(len(shape))
    nb, npr = 3000, 140_000                                              # np >= 65536: the single pass applies
    bk = rng.integers(0, 1000, nb)
    left = pa.table({"a": pa.array(bk.astype(np.int32)), "b": pa.array(bk * 7 - 2**40), "v": pa.array(np.arange(nb, dtype=np.int64))})
    pk = rng.integers(0, 1400, npr)
    if shape == "sample_sees_no_match":
        pk = np.where(np.arange(npr) % 2 == 0, 5000 + pk, pk % 1000)     # even rows (the sampled ones) miss, odd rows hit ~3 build rows
    nulls = (rng.random(npr) < 0.03) if shape == "nulls_both_equalities" else None
    right = pa.table({"c": pa.array(pk.astype(np.int32), mask=nulls), "d": pa.array(pk * 7 - 2**40), "w": pa.array(rng.integers(0, 10**6, npr))})
    if shape == "nulls_both_equalities":
        lm = rng.random(nb) < 0.05
        left = left.set_column(0, "a", pa.array(bk.astype(np.int32), mask=lm))
    on = [("a", "c"), ("b", "d")]
    for ne in (("NullEqualsNothing", "NullEqualsNull") if shape == "nulls_both_equalities" else ("NullEqualsNothing",)):
        exp = oracle.hash_join(left, right, on, "Inner", ne)
        ops.profile_enable(True)
        ops.profile_reset()
        got = gpu_join(left, right, on, "Inner", ne, probe_mode=4)
        stats = ops.profile_stats()
        ops.profile_enable(False)
        assert stats["join_probe_pairs_single"]["calls"] == (2 if shape == "sample_sees_no_match" else 1) and "join_probe_count" not in stats, sorted(stats)
        assert_tables_equal(got, exp)                                      # (multisets: the order is the tiles' arrival order)
        assert_tables_equal(gpu_join(left, right, on, "Inner", ne, probe_mode=0), exp)   # the ordered two-pass path beside it


def test_flat_table_selection():
    """auto: key columns that pack into 16 bytes get the hash table with inline keys (kinds 4 / 5), wider key sets the chained one"""
    from datafusion_amd import _lib, ops
    from datafusion_amd.table import DeviceTable
    t = DeviceTable.from_arrow(pa.table({"a": pa.array([1, 2, 3, 2], type=pa.int32()), "b": pa.array([5, 6, 7, 6], type=pa.int64()),
                                         "d": pa.array([1, 2, 3, 4], type=pa.decimal128(15, 2))}))
    i = ops.JoinHashTable(t, ["a", "b"]).info()
    assert (i.table_kind, i.build_keys_unique) == (5, 0)
    assert ops.JoinHashTable(t, ["d"]).info().table_kind == 5
    assert ops.JoinHashTable(t, ["a", "a"]).info().table_kind == 4
    assert ops.JoinHashTable(t, ["a", "d"]).info().table_kind == 0            # 20 bytes
    assert ops.JoinHashTable(t, ["a", "b"], table_mode=1).info().table_kind == 0
    with pytest.raises(_lib.DfgpuError):
        ops.JoinHashTable(t, ["a", "d"], table_mode=5)


def test_empty_sides():
    from oracle import oracle
    empty = pa.table({"a": pa.array([], type=pa.int64()), "v": pa.array([], type=pa.int32())})
    some = pa.table({"b": pa.array([1, 2, 3], type=pa.int64()), "w": pa.array([7, 8, 9], type=pa.int32())})
    for jt in ALL_TYPES:
        for l, r, on in ((empty, some, [("a", "b")]), (some, empty, [("b", "a")])):
            assert_tables_equal(gpu_join(l, r, on, jt), oracle.hash_join(l, r, on, jt))


def test_array_map_gating_matches_reference_rules():
    """try_create_array_map (exec.rs:111-191) with the reference's knob values.  The keys carry one
    duplicate so that the GPU-native rank map (unique keys only) steps aside and the reference's
    own choice between ArrayMap and JoinHashMap is what is observed."""
    from datafusion_amd import ops
    from datafusion_amd.table import DeviceTable
    dense = DeviceTable.from_arrow(pa.table({"k": pa.array(list(range(0, 4000, 2)) + [0], type=pa.int64())}))
    sparse = DeviceTable.from_arrow(pa.table({"k": pa.array(list(range(0, 400000, 200)) + [0], type=pa.int64())}))
    small = DeviceTable.from_arrow(pa.table({"k": pa.array([5, 900, 5], type=pa.int64())}))
    ref = dict(small_build_threshold=1024, min_key_density=0.15)   # the reference's defaults, config.rs:913,923
    assert ops.JoinHashTable(dense, ["k"], **ref).info().used_array_map == 1
    assert ops.JoinHashTable(sparse, ["k"], **ref).info().used_array_map == 0      # density 0.005
    assert ops.JoinHashTable(small, ["k"], **ref).info().used_array_map == 1
    assert ops.JoinHashTable(sparse, ["k"], min_key_density=0.001).info().used_array_map == 1
    assert ops.JoinHashTable(sparse, ["k"]).info().used_array_map == 0             # GPU default 1/64 still rejects 0.005
    neg = DeviceTable.from_arrow(pa.table({"k": pa.array([-(2**63), 2**63 - 1], type=pa.int64())}))
    assert ops.JoinHashTable(neg, ["k"], **ref).info().table_kind == 4  # full-range overflow guard, exec.rs:6907: the hash table (keys inline)


def test_rank_map_selection():
    """GPU-native table for unique integer keys: bitmap + popcount directory; row id = rank when the
    build keys are ascending, perm[rank] otherwise; duplicates fall back to the reference's tables"""
    from datafusion_amd import _lib, ops
    from datafusion_amd.table import DeviceTable
    asc = DeviceTable.from_arrow(pa.table({"k": pa.array(range(-50, 400000, 7), type=pa.int64())}))
    i = ops.JoinHashTable(asc, ["k"]).info()
    assert (i.table_kind, i.build_keys_ascending, i.build_keys_unique) == (2, 1, 1)
    rng = np.random.default_rng(0)
    shuffled = DeviceTable.from_arrow(pa.table({"k": pa.array(rng.permutation(100000) * 3, type=pa.int32())}))
    i = ops.JoinHashTable(shuffled, ["k"]).info()
    assert (i.table_kind, i.build_keys_ascending, i.build_keys_unique) == (2, 0, 1)
    dups = DeviceTable.from_arrow(pa.table({"k": pa.array([1, 2, 3, 2], type=pa.int64())}))
    i = ops.JoinHashTable(dups, ["k"]).info()
    assert (i.table_kind, i.build_keys_unique) == (1, 0)
    with pytest.raises(_lib.DfgpuError):
        ops.JoinHashTable(dups, ["k"], table_mode=3)
    very_sparse = DeviceTable.from_arrow(pa.table({"k": pa.array(range(0, 3_000_000, 1000), type=pa.int64())}))   # density 1/1000 < 1/256
    assert ops.JoinHashTable(very_sparse, ["k"]).info().table_kind == 4    # hash table, keys inline
    assert ops.JoinHashTable(asc, ["k"], table_mode=1).info().table_kind == 0
    assert ops.JoinHashTable(asc, ["k"], table_mode=5).info().table_kind == 4
    assert ops.JoinHashTable(asc, ["k"], table_mode=2).info().table_kind == 1


@pytest.mark.parametrize("join_type", ALL_TYPES)
@pytest.mark.parametrize("order", ["ascending", "shuffled", "with_nulls"])
def test_unique_build_keys_all_table_kinds_agree(join_type, order):
    """unique build keys: rank map (mode 3 / auto), ArrayMap (2) and hash map (1) give the oracle's rows for every JoinType"""
    from oracle import oracle
    rng = np.random.default_rng(21)
    nb = 5000
    keys = np.sort(rng.permutation(3 * nb)[:nb]).astype(np.int64) - 700
    if order != "ascending":
        keys = rng.permutation(keys)
    mask = (rng.random(nb) < 0.05) if order == "with_nulls" else None
    left = pa.table({"a": pa.array(keys, type=pa.int64(), mask=mask), "x": pa.array(np.arange(nb), type=pa.int32())})
    right = random_table(rng, 20_000, {"b": (pa.int64(), -900, 3 * nb), "z": (pa.float64(), 0, 1000)}, null_frac=0.05)
    exp = oracle.hash_join(left, right, [("a", "b")], join_type)
    for mode in (0, 1, 2, 3, 5):
        assert_tables_equal(gpu_join(left, right, [("a", "b")], join_type, table_mode=mode), exp)


def test_tpch_join_shape_small_sf():
    """BASELINE config 3 at a scale the oracle finishes in seconds: orders x lineitem on orderkey,
    Q3 payload projection; bit-exact and in probe order"""
    from datafusion_amd import ops, tpch
    from oracle import oracle
    sf = 0.02
    o, l = tpch.orders(sf), tpch.lineitem(sf)
    od, ld = ops.tpch_orders(sf), ops.tpch_lineitem(sf)
    got = ops.hash_join(od, ld, [("o_orderkey", "l_orderkey")], "Inner", build_cols=["o_orderdate", "o_shippriority"],
                        probe_cols=["l_orderkey", "l_extendedprice", "l_discount"]).to_arrow()
    exp = oracle.hash_join(o, l, [("o_orderkey", "l_orderkey")], "Inner").select(
        ["o_orderdate", "o_shippriority", "l_orderkey", "l_extendedprice", "l_discount"])
    assert got.num_rows == l.num_rows
    assert_tables_equal(got, exp, ordered=True)


@pytest.mark.parametrize("join_type", ["Inner", "RightSemi", "RightAnti", "Right", "LeftSemi"])
@pytest.mark.parametrize("probe_mode", [0, 3], ids=["two_pass", "single_pass"])
def test_probe_side_filter_fused_into_the_probe(join_type, probe_mode):
    """FilterExec below the probe side (dfgpu_join_probe_filtered): with the single-pass probe the predicate's row
    mask is applied in the probe kernel; every other flavour filters first.  NULL predicate rows are dropped."""
    from datafusion_amd import ops
    from datafusion_amd.expr import col, lit
    from datafusion_amd.table import DeviceTable
    from oracle import oracle
    from tests.util import to_oracle_expr
    rng = np.random.default_rng(17)
    build = pa.table({"k": pa.array(rng.permutation(9000)[:6000], type=pa.int64()), "v": pa.array(np.arange(6000), type=pa.int32())})
    probe = random_table(rng, 40_001, {"k2": (pa.int64(), -10, 9100), "p": (pa.decimal128(15, 2), 0, 10**7), "d": (pa.date32(), 8000, 9000)})
    dn = random_table(rng, 40_001, {"dn": (pa.int32(), 0, 100)}, null_frac=0.2).column("dn")   # nullable predicate input
    probe = probe.append_column("dn", dn)
    pred = (col("d") > lit(8500, pa.int32()).cast(pa.date32())).and_(col("dn") < lit(70, pa.int32()))
    if probe_mode == 3 and join_type not in ("Inner", "RightSemi", "RightAnti"):
        pytest.skip("the single-pass probe serves at most one match per probe row and probe-side output only")
    ht = ops.JoinHashTable(DeviceTable.from_arrow(build), ["k"], probe_mode=probe_mode)
    pcols = ["k2", "p"]
    got = ht.probe(DeviceTable.from_arrow(probe), ["k2"], join_type, ["v"], pcols, predicate=pred)
    if join_type == "LeftSemi":
        got = ht.emit_unmatched(join_type, ["v"])
    got = got.to_arrow()
    filtered = oracle.filter(probe, to_oracle_expr(pred), probe.column_names)
    exp = oracle.hash_join(build, filtered, [("k", "k2")], join_type)
    keep = {"Inner": ["v", "k2", "p"], "Right": ["v", "k2", "p"], "RightSemi": pcols, "RightAnti": pcols, "LeftSemi": ["v"]}[join_type]
    assert_tables_equal(got, exp.select(keep))


def test_column_minmax_statistics():
    """dfgpu_column_minmax: the key statistics the join build and the pruned multi-GPU exchange use"""
    from datafusion_amd import ops
    from datafusion_amd.table import DeviceTable
    rng = np.random.default_rng(3)
    v = rng.integers(-10**12, 10**12, 100_001)
    t = DeviceTable.from_arrow(pa.table({"a": pa.array(v, type=pa.int64()), "s": pa.array(np.arange(100_001) * 3 - 7, type=pa.int32()),
                                         "n": pa.array([None if i % 5 == 0 else int(x % 1000) for i, x in enumerate(v)], type=pa.int64())}))
    assert ops.column_minmax(t, "a") == (int(v.min()), int(v.max()), 100_001, False)
    assert ops.column_minmax(t, "s") == (-7, 300_000 - 7, 100_001, True)
    lo, hi, n, asc = ops.column_minmax(t, "n")
    keep = [int(x % 1000) for i, x in enumerate(v) if i % 5]
    assert (lo, hi, n, asc) == (min(keep), max(keep), len(keep), False)
    empty = DeviceTable.from_arrow(pa.table({"a": pa.array([], type=pa.int64())}))
    assert ops.column_minmax(empty, "a") == (None, None, 0, False)


FILTER_CASES = load_golden("hash_join_filter.json")


def _filter_of(case):
    """(GPU expression over f0.., oracle expression, column list) of a golden JoinFilter"""
    from datafusion_amd.expr import col, lit
    f, e = case["filter"], case["filter"]["expr"]
    lhs = col(f"f{e['left']}")
    rhs = col(f"f{e['right_col']}") if "right_col" in e else lit(e["right_lit"], pa.int32())
    gpu = {">": lhs > rhs, "!=": lhs.ne(rhs), "<": lhs < rhs, "=": lhs.eq(rhs), ">=": lhs >= rhs, "<=": lhs <= rhs}[e["op"]]
    return gpu, [(i, side) for i, side in f["columns"]]


@pytest.mark.parametrize("opts", [dict(table_mode=0), dict(table_mode=1), dict(table_mode=1, force_hash_collisions=True), dict(table_mode=4),
                                  dict(table_mode=4, force_hash_collisions=True), dict(table_mode=5), dict(table_mode=5, force_hash_collisions=True)],
                         ids=["phj_auto", "hash_map", "forced_collisions", "radix_lds", "radix_lds_forced_collisions", "flat", "flat_forced_collisions"])
@pytest.mark.parametrize("case", FILTER_CASES, ids=[c["name"] for c in FILTER_CASES])
def test_reference_snapshots_with_join_filter(case, opts):
    """the reference's join_*_with_filter tests (hash_join/exec.rs:4422-5830) through dfgpu_join_probe_with_filter"""
    left = i32_table(case["left"]["columns"], case["left"]["data"])
    right = i32_table(case["right"]["columns"], case["right"]["data"])
    out = gpu_join(left, right, [tuple(p) for p in case["on"]], case["join_type"], case["null_equality"], join_filter=_filter_of(case), **opts)
    assert out.column_names == case["expected_columns"], case["source"]
    key = lambda row: tuple((v is None, 0 if v is None else v) for v in row)
    assert sorted_rows(out) == sorted([tuple(r) for r in case["expected_rows"]], key=key), case["source"]


@pytest.mark.parametrize("join_type", ALL_TYPES)
def test_random_join_filter_vs_oracle(join_type):
    from datafusion_amd.expr import col
    from oracle import oracle
    from tests.util import to_oracle_expr
    rng = np.random.default_rng(23)
    left = random_table(rng, 2500, {"a": (pa.int64(), 0, 300), "x": (pa.decimal128(15, 2), 0, 10**6), "y": (pa.int32(), 0, 100)}, null_frac=0.05)
    right = random_table(rng, 6000, {"b": (pa.int64(), 0, 350), "z": (pa.decimal128(15, 2), 0, 10**6), "w": (pa.int32(), 0, 100)}, null_frac=0.05)
    # residual predicate over both sides with NULLs on both: left.x > right.z AND left.y != right.w
    gpu_expr = (col("f0") > col("f1")).and_(col("f2").ne(col("f3")))
    cols = [(1, "Left"), (1, "Right"), (2, "Left"), (2, "Right")]
    for mode in (0, 1, 4, 5):
        got = gpu_join(left, right, [("a", "b")], join_type, join_filter=(gpu_expr, cols), table_mode=mode)
        exp = oracle.hash_join(left, right, [("a", "b")], join_type, join_filter=(to_oracle_expr(gpu_expr), cols))
        assert_tables_equal(got, exp)


def test_array_map_and_hash_map_known_answers_on_the_device():
    """the map-level known answers of tests/test_oracle_join_golden.py (array_map.rs:428-599, join_hash_map.rs:518-572) as device joins"""
    from datafusion_amd import ops
    from datafusion_amd.table import DeviceTable
    i32, i64, u64 = pa.int32(), pa.int64(), pa.uint64()
    cases = [([1, 1, 2], [1, 2], i32, [(0, 0), (0, 1), (1, 2)]), ([1, 2], [10, 1, 2], i32, [(1, 0), (2, 1)]),
             ([1, 1], [10, 1, 20, 1], i32, [(1, 0), (1, 1), (3, 0), (3, 1)]), (list(range(11)), [3, (1 << 32) + 3, 11, None], u64, [(0, 3)]),
             ([-5, 0, 5, -2, 3, 10], [0, -5, 10, -1], i64, [(0, 1), (1, 0), (2, 5)]),
             ([10, 20, 30], [10, None, 30], i64, [(0, 0), (2, 2)]), ([10, 20, 10, 20], [None, 20], i64, [(1, 1), (1, 3)])]
    for build, probe, typ, want in cases:
        b = DeviceTable.from_arrow(pa.table({"k": pa.array(build, typ), "bi": pa.array(range(len(build)), pa.int64())}))
        p = DeviceTable.from_arrow(pa.table({"k2": pa.array(probe, typ), "pi": pa.array(range(len(probe)), pa.int64())}))
        for table_mode in (0, 1, 4):
            j = ops.hash_join(b, p, [("k", "k2")], "Inner", table_mode=table_mode).to_arrow()
            assert sorted(zip(j.column("pi").to_pylist(), j.column("bi").to_pylist())) == sorted(want), (build, probe, table_mode)


@pytest.mark.parametrize("shape", ["many_partitions", "skewed_duplicates", "no_matches"])
def test_radix_lds_join_many_partitions_and_skew(shape):
    """the LDS radix join over inputs large enough for several partition bits, several tasks per partition and (skew) several
    LDS chunks per partition; M:N output compared with the oracle as a multiset"""
    from oracle import oracle
    rng = np.random.default_rng(31)
    if shape == "many_partitions":
        nb, np_, dom = 300_000, 1_000_000, 400_000
        bk, pk = rng.integers(0, dom, nb), rng.integers(0, dom, np_)
    elif shape == "skewed_duplicates":
        nb, np_ = 60_000, 200_000
        bk = np.where(rng.random(nb) < 0.1, 7, rng.integers(0, 50_000, nb))          # 10 % of the build rows share one key
        pk = np.where(rng.random(np_) < 0.001, 7, rng.integers(0, 50_000, np_))
    else:
        nb, np_ = 100_000, 300_000
        bk, pk = rng.integers(0, 10**6, nb) * 2, rng.integers(0, 10**6, np_) * 2 + 1
    left = pa.table({"a": pa.array(bk, type=pa.int64()), "bi": pa.array(np.arange(nb), type=pa.int32())})
    right = pa.table({"b": pa.array(pk, type=pa.int64()), "pi": pa.array(np.arange(np_), type=pa.int32())})
    for jt in ("Inner", "RightAnti", "LeftSemi"):
        got = gpu_join(left, right, [("a", "b")], jt, table_mode=4)
        exp = oracle.hash_join(left, right, [("a", "b")], jt)
        assert got.num_rows == exp.num_rows
        assert_tables_equal(got, exp)


@pytest.mark.parametrize("keys", ["int64_nullable", "int32", "two_columns", "decimal128"])
@pytest.mark.parametrize("opts", [dict(join__radix_partition_rows=8), dict(join__radix_partition_rows=1), dict(join__radix_partition_rows=8, join__radix_tile_threads=256, join__radix_tile_items=8), dict(join__radix_partition_rows=8, join__radix_tile_threads=256),
                                  dict(join__radix_partition_rows=8, join__radix_onesweep=0)], ids=["two_passes", "three_passes", "tiles_256x8", "tiles_256x16", "six_bit_passes"])
def test_radix_partitioner_passes_masks_and_key_kinds(keys, opts):
    """the radix join's partitioner (round 6: 8-bit passes straight off the key column, look-back from the second pass on) forced to two and to
    three passes on a 70 K-row build side, over NULL keys (the validity mask decides which rows become records), 32-bit keys, and keys that
    are hashed (two columns, Decimal128: records carry the row hash and matches are re-checked); every join type's pairs against the oracle"""
    from oracle import oracle
    from datafusion_amd import ops
    rng = np.random.default_rng(77)
    nb, np_ = 70_000, 120_000
    if keys == "int64_nullable":
        left = random_table(rng, nb, {"a": (pa.int64(), -40_000, 40_000), "x": (pa.int32(), 0, 100)}, null_frac=0.07)
        right = random_table(rng, np_, {"b": (pa.int64(), -50_000, 50_000), "z": (pa.int32(), 0, 100)}, null_frac=0.07)
        on = [("a", "b")]
    elif keys == "int32":
        left = random_table(rng, nb, {"a": (pa.int32(), 0, 90_000), "x": (pa.int64(), 0, 100)}, null_frac=0.0)
        right = random_table(rng, np_, {"b": (pa.int32(), 0, 120_000), "z": (pa.int32(), 0, 100)}, null_frac=0.0)
        on = [("a", "b")]
    elif keys == "two_columns":
        left = random_table(rng, nb, {"a": (pa.int32(), 0, 300), "a2": (pa.int64(), 0, 300), "x": (pa.int32(), 0, 100)}, null_frac=0.03)
        right = random_table(rng, np_, {"b": (pa.int32(), 0, 320), "b2": (pa.int64(), 0, 320), "z": (pa.int32(), 0, 100)}, null_frac=0.03)
        on = [("a", "b"), ("a2", "b2")]
    else:
        left = random_table(rng, nb, {"a": (pa.decimal128(20, 2), 0, 80_000), "x": (pa.int32(), 0, 100)}, null_frac=0.02)
        right = random_table(rng, np_, {"b": (pa.decimal128(20, 2), 0, 100_000), "z": (pa.int32(), 0, 100)}, null_frac=0.02)
        on = [("a", "b")]
    ops.set_options(**opts)
    for jt, ne in (("Inner", "NullEqualsNothing"), ("Full", "NullEqualsNothing"), ("LeftAnti", "NullEqualsNothing"), ("Inner", "NullEqualsNull")):
        got, names = _probe_paths(lambda: gpu_join(left, right, on, jt, ne, table_mode=4))
        assert ("radix_join_partition_pass" in names) == (opts.get("join__radix_onesweep", 1) == 1), names
        exp = oracle.hash_join(left, right, on, jt, ne)
        assert got.num_rows == exp.num_rows
        assert_tables_equal(got, exp)


@pytest.mark.parametrize("part_rows", [2400, 16], ids=["one_partition_level", "two_passes"])
def test_radix_inner_join_writes_payload_columns_of_every_width_from_the_emit_walk(part_rows):
    """INNER radix join whose output columns are written by the emit walk itself (no pair list, no gathers): payload of 16, 8, 4 and 1 bytes on
    both sides without NULLs, the key column of either side among the outputs (rebuilt from the record key), duplicates on both sides"""
    from oracle import oracle
    from datafusion_amd import ops
    rng = np.random.default_rng(5)
    nb, np_ = 90_000, 200_000
    left = random_table(rng, nb, {"a": (pa.int64(), -30_000, 30_000), "x": (pa.decimal128(15, 2), 0, 10**6), "y": (pa.int32(), 0, 1000), "u": (pa.uint8(), 0, 255), "v": (pa.int64(), 0, 10**12)})
    right = random_table(rng, np_, {"b": (pa.int64(), -40_000, 40_000), "z": (pa.decimal128(15, 2), 0, 10**6), "w": (pa.date32(), 8000, 9000), "t": (pa.uint8(), 0, 255), "s": (pa.int64(), 0, 10**12)})
    ops.set_options(join__radix_partition_rows=part_rows)
    got, names = _probe_paths(lambda: gpu_join(left, right, [("a", "b")], "Inner", table_mode=4))
    assert "radix_join_emit_columns" in names, names
    exp = oracle.hash_join(left, right, [("a", "b")], "Inner")
    assert got.num_rows == exp.num_rows and got.num_rows > 200_000
    assert_tables_equal(got, exp)
    # the pairs + gathers of round 5 (join.radix_fused_emit = 0) give the same rows
    ops.set_options(join__radix_fused_emit=0)
    got2, names2 = _probe_paths(lambda: gpu_join(left, right, [("a", "b")], "Inner", table_mode=4))
    assert "radix_join_emit_columns" not in names2 and "radix_join_emit" in names2, names2
    assert_tables_equal(got2, exp)


def _probe_paths(fn):
    """run fn() with the library's profile on; returns (result, names of the profiled scopes that ran)"""
    from datafusion_amd import ops
    ops.profile_enable(True)
    ops.profile_reset()
    try:
        out = fn()
        names = set(ops.profile_stats())
    finally:
        ops.profile_enable(False)
    return out, names


@pytest.mark.parametrize("np_rows", [1, 63, 64, 65, 2047, 2049, 8191, 8192, 8193, 50_003])
@pytest.mark.parametrize("join_type", ["Inner", "RightSemi", "RightAnti"])
def test_selective_probe_lists_its_output_rows(np_rows, join_type):
    """few output rows expected (a build side that covers little of its key range): counts + hit words, then one thread per listed
    output row (join.hip k_join_emit_listed).  Sizes around the 64-row word, the 2048-row tile and the 8192-row group; output in
    probe order under every probe_mode; the key column written from the reloaded key"""
    from datafusion_amd import ops
    from datafusion_amd.table import DeviceTable
    from oracle import oracle
    rng = np.random.default_rng(np_rows)
    build = pa.table({"k": pa.array(rng.permutation(100_000)[:4000] * 3 - 50_000, type=pa.int64()), "v": pa.array(np.arange(4000), type=pa.int32()),
                      "w": pa.array(rng.integers(0, 10**9, 4000), type=pa.int64())})
    probe = pa.table({"k2": pa.array(rng.integers(-60_000, 260_000, np_rows), type=pa.int64()),
                      "p": pa.array(rng.integers(0, 10**9, np_rows), type=pa.int32()).cast(pa.decimal128(15, 2)),
                      "q": pa.array(rng.integers(0, 100, np_rows).astype(np.uint8))})
    for probe_mode in (0, 3):
        ht = ops.JoinHashTable(DeviceTable.from_arrow(build), ["k"], probe_mode=probe_mode)
        got, names = _probe_paths(lambda: ht.probe(DeviceTable.from_arrow(probe), ["k2"], join_type, ["w", "v"], ["q", "k2", "p"]).to_arrow())
        exp = oracle.hash_join(build, probe, [("k", "k2")], join_type)
        keep = ["w", "v", "q", "k2", "p"] if join_type == "Inner" else ["q", "k2", "p"]
        assert_tables_equal(got, exp.select(keep), ordered=True)
        if join_type != "RightAnti" and exp.num_rows:       # (the anti join emits most rows here: the counts pass sends it to the placed kernel)
            assert "join_probe_listed" in names, names
        ht.free()


@pytest.mark.parametrize("table_mode", ["array_map", "rank_map"])
@pytest.mark.parametrize("key_type", [pa.int32(), pa.int64()], ids=["i32", "i64"])
def test_selective_probe_clustered_hits_nullable_keys_and_masks(table_mode, key_type):
    """the listed path where a whole 8192-row group is output (ascending probe keys: the hits sit in one cluster, 32 rounds of one
    thread per row), NULL probe keys (never a match; emitted by RightAnti), a fused FilterExec, an all-false mask, and a mask so
    permissive that the counts pass sends the probe to the placed kernel instead"""
    from datafusion_amd import ops
    from datafusion_amd.expr import col, lit
    from datafusion_amd.table import DeviceTable
    from oracle import oracle
    from tests.util import to_oracle_expr
    rng = np.random.default_rng(5)
    n = 120_000
    build = pa.table({"k": pa.array(np.concatenate([np.arange(0, 9000), [99_999]]), type=key_type), "v": pa.array(rng.integers(0, 10**6, 9001), type=pa.int32())})
    keys = np.sort(rng.integers(0, 100_000, n))          # ~9 % of the probe rows match, all of them at the front
    probe = pa.table({"k2": pa.array(keys, type=key_type, mask=rng.random(n) < 0.03), "p": pa.array(rng.integers(0, 10**9, n), type=pa.int64()),
                      "f": pa.array(rng.integers(0, 100, n), type=pa.int32())})
    tm = ops.TABLE_MODES[table_mode]
    for join_type in ("Inner", "RightSemi", "RightAnti"):
        keep = ["v", "f", "p"] if join_type == "Inner" else ["f", "p"]     # (a nullable payload column would take the general path)
        for pred, expect_listed in ((None, join_type != "RightAnti"), (col("f") < lit(20, pa.int32()), True), (col("f") < lit(0, pa.int32()), None),
                                    (col("f") < lit(99, pa.int32()), None)):
            ht = ops.JoinHashTable(DeviceTable.from_arrow(build), ["k"], probe_mode=3, table_mode=tm)
            got, names = _probe_paths(lambda: ht.probe(DeviceTable.from_arrow(probe), ["k2"], join_type, ["v"], ["f", "p"], predicate=pred).to_arrow())
            src = probe if pred is None else oracle.filter(probe, to_oracle_expr(pred), probe.column_names)
            exp = oracle.hash_join(build, src, [("k", "k2")], join_type).select(keep)
            assert_tables_equal(got, exp, ordered="join_probe_fused" not in names)    # (probe_mode 3 leaves the order to the flavour that runs)
            if expect_listed is True:
                assert "join_probe_listed" in names, (join_type, names)
            ht.free()


def test_rank_map_over_unordered_keys_builds_its_permutation_on_the_first_probe_that_needs_build_rows():
    """build keys in no particular order: a probe that only asks whether the key is there (SELECT l.k, RightSemi / RightAnti)
    runs on the bitmap alone; the rank -> row permutation appears with the first probe that gathers build columns or marks
    build rows — also when several probe partitions arrive at once"""
    import threading

    from datafusion_amd import ops
    from datafusion_amd.table import DeviceTable
    from oracle import oracle
    rng = np.random.default_rng(12)
    nb, npr = 200_000, 1_000_003
    keys = rng.permutation(nb).astype(np.int64) * 4 + 1                      # unique, shuffled, density 1/4
    build = pa.table({"k": pa.array(keys), "pay": pa.array(rng.integers(0, 10**6, nb), type=pa.int32())})
    pk = rng.integers(0, nb * 5, npr).astype(np.int64)                       # ~ 1/5 of the probe keys exist
    probe = pa.table({"k2": pa.array(pk), "v": pa.array(rng.integers(0, 100, npr), type=pa.int64())})
    b, p = DeviceTable.from_arrow(build), DeviceTable.from_arrow(probe)
    ht = ops.JoinHashTable(b, ["k"], probe_mode=4)
    i0 = ht.info()
    assert (i0.table_kind, i0.build_keys_ascending, i0.build_keys_unique) == (2, 0, 1)
    bitmap_only = i0.table_bytes
    for jt in ("Inner", "RightSemi", "RightAnti"):
        got = ht.probe(p, ["k2"], jt, [], ["k2", "v"]).to_arrow()
        exp = oracle.hash_join(build, probe, [("k", "k2")], jt).select(["k2", "v"])
        assert sorted_rows(got) == sorted_rows(exp), jt
    # no permutation was needed — only the interleaved {bits, prefix} view of the same 16 bytes per word, which the first probe that
    # asks every row for its rank builds (round 6: it used to be built with the table)
    assert ht.info().table_bytes == 2 * bitmap_only
    results, errors = {}, []

    def worker(w, jt):
        try:
            results[w] = ht.probe(p, ["k2"], jt, ["pay"] if jt != "LeftSemi" else ["k", "pay"], ["k2"] if jt != "LeftSemi" else []).to_arrow()
        except Exception as e:      # noqa: BLE001
            errors.append(e)
    threads = [threading.Thread(target=worker, args=(w, "Inner")) for w in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors
    assert ht.info().table_bytes == 2 * bitmap_only + nb * 4                 # built once
    exp = oracle.hash_join(build, probe, [("k", "k2")], "Inner").select(["pay", "k2"])
    for w in range(4):
        assert sorted_rows(results[w]) == sorted_rows(exp)
    ht.free()
    ht2 = ops.JoinHashTable(b, ["k"])
    ht2.probe(p, ["k2"], "LeftSemi", ["k", "pay"], [])
    got = ht2.emit_unmatched("LeftSemi", ["k", "pay"]).to_arrow()            # visited marks are per build ROW: the permutation is there
    assert sorted_rows(got) == sorted_rows(oracle.hash_join(build, probe, [("k", "k2")], "LeftSemi"))
    ht2.free()


@pytest.mark.parametrize("join_type", ["Inner", "RightSemi", "RightAnti"])
def test_key_only_probe_of_unordered_keys_is_grouped_by_key_range(join_type):
    """probe keys in no order against a table beyond the caches, nothing read but the key, order unobserved (probe_mode 4): the
    keys are grouped by the top bits of their range before the lookup — the same rows come out (as a multiset), whichever way"""
    import os

    from datafusion_amd import ops
    from datafusion_amd.table import DeviceTable
    rng = np.random.default_rng(14)
    nb, npr = 3_000_000, 6_000_011
    bkeys = (rng.permutation(nb).astype(np.int64) // 8) * 32 + rng.permutation(nb) % 8 + 1          # sparse, shuffled; duplicates possible -> made unique
    bkeys = np.unique(bkeys)
    rng.shuffle(bkeys)
    pk = rng.integers(0, int(bkeys.max()) + 1000, npr).astype(np.int64)
    b = DeviceTable.from_arrow(pa.table({"k": pa.array(bkeys)}))
    p = DeviceTable.from_arrow(pa.table({"k2": pa.array(pk)}))
    member = np.isin(pk, bkeys)
    exp = np.sort(pk[member if join_type != "RightAnti" else ~member])
    outs = {}
    for grouped in ("1", "0"):
        ops.set_options(join__grouped_probe=grouped)
        ops.set_options(join__beyond_cache_bytes="1000000")  # this test's 3 MB table counts as beyond the caches
        try:
            ht = ops.JoinHashTable(b, ["k"], probe_mode=4)
            assert ht.info().table_kind == 2
            out = ht.probe(p, ["k2"], join_type, [], ["k2"])
            outs[grouped] = out.to_arrow().column("k2").to_numpy()
            assert out.column_names == ["k2"]
            ht.free()
        finally:
            ops.set_options(join__grouped_probe=None)
            ops.set_options(join__beyond_cache_bytes=None)
        assert np.array_equal(np.sort(outs[grouped]), exp), grouped
    if join_type == "Inner":
        assert not np.array_equal(outs["1"], outs["0"])       # the grouped flavour really ran: its rows come out in group order


@pytest.mark.parametrize("gp_bits", ["3", "9", "10"])
@pytest.mark.parametrize("build_order", ["shuffled", "ascending"])
def test_unclustered_probe_with_payload_goes_through_groups_and_comes_back_in_probe_order(build_order, gp_bits):
    """round 4: a probe whose keys arrive in no order against a rank map beyond the caches, WITH payload on both sides: the probe keys
    are grouped by their position in the table's key range, looked up group by group (build payload read at rank positions — from a
    rank-ordered copy when the build keys are shuffled), and what they found returns to the probe rows through `dest`.  The output
    is in probe order like the reference's (exec.rs:3349) under every probe_mode; NULL keys, keys outside the range, misses, a fused
    probe-side predicate, RightSemi / RightAnti all take the same route"""
    import os

    from datafusion_amd import ops
    from datafusion_amd.expr import col, lit
    from datafusion_amd.table import DeviceTable
    from oracle import oracle
    rng = np.random.default_rng(91 + len(build_order))
    nb, npr = 200_000, 1_000_003
    bkeys = np.unique((rng.permutation(4 * nb)[:nb].astype(np.int64)) * 3 - 1000)
    if build_order == "shuffled":
        rng.shuffle(bkeys)
    nb = len(bkeys)
    build = pa.table({"k": pa.array(bkeys), "d": pa.array((bkeys % 9000).astype(np.int32), type=pa.int32()).cast(pa.date32()),
                      "p": pa.array((bkeys % 7).astype(np.int32), type=pa.int32()), "w": pa.array(bkeys * 11, type=pa.int64()),
                      "x": random_table(rng, nb, {"x": (pa.decimal128(15, 2), -10**6, 10**6)}).column("x"),
                      "u": pa.array((np.arange(nb) % 251).astype(np.uint8), type=pa.uint8())})
    probe = random_table(rng, npr, {"k2": (pa.int64(), -5000, 12 * nb + 5000), "e": (pa.decimal128(15, 2), 0, 10**7), "q": (pa.int32(), 0, 50)})
    # NULL keys; the payload stays non-nullable (a nullable output column takes the pairs path): the key travels as the copy "kk"
    probe = probe.append_column("kk", probe.column("k2")).set_column(0, "k2", pa.array(probe.column("k2").to_numpy(), mask=rng.random(npr) < 0.02))
    b, p = DeviceTable.from_arrow(build), DeviceTable.from_arrow(probe)
    # (test knobs: this table counts as beyond the caches, grouping from the first row on, "near" = within 1000 key values)
    ops.set_options(join__beyond_cache_bytes="0", join__grouped_min_rows="0", join__grouped_bits=gp_bits, join__near_window="1000")
    try:
        for payload in (["d", "p"], ["x", "u", "w", "p", "d"], ["w"], []):
            exp = oracle.hash_join(build, probe, [("k", "k2")], "Inner").select(payload + ["kk", "e", "q"])
            for probe_mode in (0, 3, 4):
                ht = ops.JoinHashTable(b, ["k"], probe_mode=probe_mode)
                assert ht.info().table_kind == 2
                ops.profile_enable(True)
                ops.profile_reset()
                got = ht.probe(p, ["k2"], "Inner", payload, ["kk", "e", "q"]).to_arrow()
                stats = ops.profile_stats()
                ops.profile_enable(False)
                assert "join_probe_grouped_lookup" in stats and "join_build_rank_perm" not in stats, sorted(stats)
                assert ("join_build_rank_payload" in stats) == (build_order == "shuffled" and bool(payload)), sorted(stats)
                assert_tables_equal(got, exp, ordered=probe_mode == 0)
                ht.free()
        ht = ops.JoinHashTable(b, ["k"])
        for jt in ("RightSemi", "RightAnti"):
            got = ht.probe(p, ["k2"], jt, [], ["kk", "q"]).to_arrow()
            assert_tables_equal(got, oracle.hash_join(build, probe, [("k", "k2")], jt).select(["kk", "q"]), ordered=True)
        # a FilterExec fused below the probe side (its row mask rides through the grouping)
        pred = col("q") < lit(20, pa.int32())
        got = ht.probe(p, ["k2"], "Inner", ["d", "w"], ["kk", "e"], predicate=pred).to_arrow()
        import pyarrow.compute as pc
        kept = probe.filter(pc.less(probe.column("q"), 20))
        assert_tables_equal(got, oracle.hash_join(build, kept, [("k", "k2")], "Inner").select(["d", "w", "kk", "e"]), ordered=True)
        ht.free()
        # every probe row finds its key (a foreign key): the lookup's hit count lets the placed probe run without a counts pass
        fk = pa.table({"k2": pa.array(bkeys[rng.integers(0, nb, npr)]), "e": probe.column("e")})
        ht = ops.JoinHashTable(b, ["k"])
        ops.profile_enable(True)
        ops.profile_reset()
        got = ht.probe(DeviceTable.from_arrow(fk), ["k2"], "Inner", ["d", "p"], ["k2", "e"]).to_arrow()
        stats = ops.profile_stats()
        ops.profile_enable(False)
        assert "join_probe_grouped_lookup" in stats and "join_probe_tile_counts" not in stats and "join_probe_speculation_missed" not in stats, sorted(stats)
        assert_tables_equal(got, oracle.hash_join(build, fk, [("k", "k2")], "Inner").select(["d", "p", "k2", "e"]), ordered=True)
        ht.free()
    finally:
        ops.set_options(join__beyond_cache_bytes=None, join__grouped_min_rows=None, join__grouped_bits=None, join__near_window=None)


@pytest.mark.parametrize("dangling", [0, 3])
def test_probe_order_speculation_every_row_finds_its_key(dangling):
    """the probe-order (placed) flavour skips its counts pass when 64 K sampled probe rows all find their key and verifies the
    assumption tile by tile while it writes: a foreign-key probe comes out in one pass; three dangling keys between the samples make
    the kernel raise its flag and the host run the counted probe — same rows, same order, either way"""
    from datafusion_amd import ops
    from datafusion_amd.table import DeviceTable
    rng = np.random.default_rng(31)
    nb, npr = 500_000, 5_000_000
    okeys = (np.arange(nb, dtype=np.int64) // 8) * 32 + np.arange(nb) % 8 + 1
    fk = np.sort(rng.integers(0, nb, npr))
    pk = okeys[fk].copy()
    pay = rng.integers(0, 10**6, npr).astype(np.int64)
    if dangling:
        for pos in (70_001, 2_345_679, 4_999_998):       # none of them on a sampled word
            pk[pos] = -5
    build = DeviceTable.from_arrow(pa.table({"o_orderkey": pa.array(okeys), "o_flag": pa.array((np.arange(nb) % 7).astype(np.int32))}))
    probe = DeviceTable.from_arrow(pa.table({"l_orderkey": pa.array(pk), "l_pay": pa.array(pay)}))
    ops.profile_enable(True)
    ops.profile_reset()
    ht = ops.JoinHashTable(build, ["o_orderkey"], probe_mode=0)
    out = ht.probe(probe, ["l_orderkey"], "Inner", ["o_flag"], ["l_orderkey", "l_pay"]).to_arrow()
    stats = ops.profile_stats()
    ops.profile_enable(False)
    ht.free()
    keep = pk > 0
    assert out.num_rows == int(keep.sum())
    assert np.array_equal(out.column("l_orderkey").to_numpy(), pk[keep]) and np.array_equal(out.column("l_pay").to_numpy(), pay[keep])      # probe order
    assert np.array_equal(out.column("o_flag").to_numpy(), (fk[keep] % 7).astype(np.int32))
    assert ("join_probe_speculation_missed" in stats) == bool(dangling), sorted(stats)
    assert ("join_probe_tile_counts" in stats) == bool(dangling)                                  # no counts pass when the speculation holds


@pytest.mark.parametrize("flaw", ["none", "swap", "duplicate", "out_of_range"])
def test_build_speculates_on_ascending_keys_and_verifies_while_it_builds(flaw):
    """a large build guesses min / max / order from its first key, its last key and a sample of neighbours and builds the rank map in
    one pass that checks every key against its predecessor and the guessed range; keys in table order take that pass alone (no
    statistics pass), a single swapped pair, a repeated key or a key beyond the last one — none of them on a sampled position — raise
    the flag and the build starts over from measured statistics: the same join either way"""
    from datafusion_amd import ops
    from datafusion_amd.table import DeviceTable
    rng = np.random.default_rng(44)
    nb, npr = 5_000_000, 3_000_000
    keys = np.arange(nb, dtype=np.int64) * 3 + 7
    if flaw == "swap":
        keys[[1_234_567, 1_234_568]] = keys[[1_234_568, 1_234_567]]
    elif flaw == "duplicate":
        keys[3_000_001] = keys[3_000_000]
    elif flaw == "out_of_range":
        keys[2_222_223] = keys[-1] + 1_000_003
    pay = rng.integers(0, 10**6, nb).astype(np.int64)
    pk = rng.integers(0, nb * 3 + 50, npr).astype(np.int64)
    build = DeviceTable.from_arrow(pa.table({"k": pa.array(keys), "v": pa.array(pay)}))
    probe = DeviceTable.from_arrow(pa.table({"pk": pa.array(pk), "row": pa.array(np.arange(npr, dtype=np.int64))}))
    ops.profile_enable(True)
    ops.profile_reset()
    ht = ops.JoinHashTable(build, ["k"], probe_mode=0)
    out = ht.probe(probe, ["pk"], "Inner", ["v"], ["pk", "row"]).to_arrow()
    stats = ops.profile_stats()
    ops.profile_enable(False)
    info = ht.info()
    ht.free()
    assert ("join_build_speculation_missed" in stats) == (flaw != "none"), sorted(stats)
    assert ("join_build_key_stats" in stats) == (flaw != "none")                      # no statistics pass when the guess holds
    # expected pairs from a host dictionary (the duplicate key matches twice)
    order = np.argsort(keys, kind="stable")
    sk = keys[order]
    lo, hi = np.searchsorted(sk, pk, "left"), np.searchsorted(sk, pk, "right")
    exp = sorted((int(r), int(pay[order[j]])) for r in np.nonzero(hi > lo)[0] for j in range(lo[r], hi[r]))
    got = sorted(zip(out.column("row").to_pylist(), out.column("v").to_pylist()))
    assert got == exp
    assert bool(info.build_keys_unique) == (flaw != "duplicate")


@pytest.mark.parametrize("shape", ["dense_unique_int64", "two_column_duplicates", "sparse_int64_with_duplicates"])
def test_auto_takes_a_table_kind_within_reach_of_the_fastest(shape):
    """`auto` (table_mode 0) against every table kind that applies to the shape — chained (1), ArrayMap (2), rank map (3), LDS radix (4),
    flat (5) — build + probe timed on the device, best of five after a warm-up: what `auto` takes may not be more than 25 % (and 0.3 ms)
    slower than the fastest forced kind.  4 M-row builds against 16 M-row probes: tables beyond an XCD's L2, where the kinds differ.
    The dispatch thresholds were tuned on the benchmark shapes (profiles/r4_join_shapes_final.md); this is what keeps them honest"""
    import time

    from datafusion_amd import ops
    from datafusion_amd.table import DeviceTable
    rng = np.random.default_rng(len(shape))
    nb, npr = 4_000_000, 16_000_000
    if shape == "dense_unique_int64":
        build = pa.table({"k": pa.array(rng.permutation(nb).astype(np.int64)), "v": pa.array(np.arange(nb, dtype=np.int64))})
        probe = pa.table({"k": pa.array(rng.integers(0, nb, npr)), "w": pa.array(np.arange(npr, dtype=np.int64))})
        on, kinds = ["k"], [1, 2, 3, 5]
    elif shape == "two_column_duplicates":
        build = pa.table({"a": pa.array(rng.integers(0, 2000, nb).astype(np.int32)), "b": pa.array(rng.integers(0, 1000, nb)), "v": pa.array(np.arange(nb, dtype=np.int64))})
        probe = pa.table({"a": pa.array(rng.integers(0, 2000, npr).astype(np.int32)), "b": pa.array(rng.integers(0, 1000, npr)), "w": pa.array(np.arange(npr, dtype=np.int64))})
        on, kinds = ["a", "b"], [1, 4, 5]
    else:
        build = pa.table({"k": pa.array(rng.integers(0, 2**40, nb // 2).repeat(2)), "v": pa.array(np.arange(nb, dtype=np.int64))})
        probe = pa.table({"k": pa.array(np.concatenate([build.column("k").to_numpy()[rng.integers(0, nb, npr // 2)], rng.integers(0, 2**40, npr // 2)])),
                          "w": pa.array(np.arange(npr, dtype=np.int64))})
        on, kinds = ["k"], [1, 4, 5]
    b, p = DeviceTable.from_arrow(build), DeviceTable.from_arrow(probe)

    def run(mode):
        best, rows = None, None
        for it in range(6):
            ops.sync()
            t0 = time.perf_counter()
            jt = ops.JoinHashTable(b, on, table_mode=mode, probe_mode=4)
            out = jt.probe(p, on, "Inner", build_cols=["v"], probe_cols=["w"])
            ops.sync()
            dt = time.perf_counter() - t0
            rows = out.num_rows
            out.free()
            jt.free() if hasattr(jt, "free") else None
            if it and (best is None or dt < best):
                best = dt
        return best, rows

    seen = []
    for attempt in range(3):   # (a timing on a shared box: a systematic gap shows in every attempt, a hiccup in one)
        t_auto, rows_auto = run(0)
        forced = {}
        for m in kinds:
            forced[m], rows = run(m)
            assert rows == rows_auto, (shape, m, rows, rows_auto)
        fastest = min(forced.values())
        seen.append({"auto_ms": round(t_auto * 1e3, 3), "forced_ms": {m: round(t * 1e3, 3) for m, t in forced.items()}})
        if t_auto <= fastest * 1.25 + 0.3e-3:
            break
    else:
        raise AssertionError({"shape": shape, "attempts": seen})


@pytest.mark.parametrize("table_mode", ["array_map", "rank_map"])
def test_simple_predicates_are_evaluated_inside_the_counts_pass(table_mode):
    """round 6: a FilterExec below the probe side whose predicate is an AND of `column <op> literal` over fixed-width integer-like columns
    is evaluated per row by the selective probe's counts pass (RowPred) — no k_cmp launch, no row mask — for every comparison, either
    operand order, every supported column type, nullable predicate columns, the inverted (anti) probe, and a predicate so permissive
    that the counts pass hands its output words to the placed kernel as the row mask.  Same rows as filter-then-join by the oracle, and
    as the mask path (join.pred_in_counts=0)."""
    from datafusion_amd import ops
    from datafusion_amd.expr import col, lit
    from datafusion_amd.table import DeviceTable
    from oracle import oracle
    from tests.util import to_oracle_expr
    rng = np.random.default_rng(606)
    n = 150_001
    build = pa.table({"k": pa.array(np.sort(rng.permutation(400_000)[:50_000]), type=pa.int64()), "v": pa.array(rng.integers(0, 10**6, 50_000), type=pa.int32())})
    probe = pa.table({"k2": pa.array(rng.integers(-3, 400_010, n), type=pa.int64()), "p": pa.array(rng.integers(0, 10**9, n), type=pa.int64()),
                      "d": pa.array(rng.integers(8000, 9000, n).astype(np.int32), type=pa.date32()), "i": pa.array(rng.integers(-50, 50, n), type=pa.int32()),
                      "b": pa.array(rng.integers(0, 256, n).astype(np.uint8)), "u": pa.array(rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)),
                      "w": pa.array(rng.integers(-2**40, 2**40, n), type=pa.int64()),
                      "dn": pa.array(rng.integers(0, 100, n), type=pa.int32(), mask=rng.random(n) < 0.2)})
    d = lambda v: lit(v, pa.int32()).cast(pa.date32())
    preds = {"date_gt": col("d") > d(8500), "date_le_and_i_ne": (col("d") <= lit(datetime.date(1970, 1, 1) + datetime.timedelta(days=8300), pa.date32())).and_(col("i").ne(lit(7, pa.int32()))), "lit_on_the_left": lit(10, pa.int32()) >= col("i"),
             "u8_eq": col("b").eq(lit(17, pa.uint8())), "u32_ge": col("u") >= lit(3_000_000_000, pa.uint32()), "i64_lt_negative": col("w") < lit(-2**39, pa.int64()),
             "nullable": (col("dn") < lit(30, pa.int32())).and_(col("d") >= d(8100)), "permissive": col("i") >= lit(-49, pa.int32())}
    dev_b, dev_p = DeviceTable.from_arrow(build), DeviceTable.from_arrow(probe)
    ht = ops.JoinHashTable(dev_b, ["k"], probe_mode=3, table_mode=ops.TABLE_MODES[table_mode])
    for name, pred in preds.items():
        filtered = oracle.filter(probe, to_oracle_expr(pred), probe.column_names)
        for join_type in ("Inner", "RightSemi", "RightAnti"):
            keep = ["v", "k2", "p"] if join_type == "Inner" else ["k2", "p"]
            exp = oracle.hash_join(build, filtered, [("k", "k2")], join_type).select(keep)
            got, names = _probe_paths(lambda: ht.probe(dev_p, ["k2"], join_type, ["v"], ["k2", "p"], predicate=pred).to_arrow())
            assert "cmp" not in names and "join_probe_tile_counts" in names, (name, join_type, names)
            assert ("join_probe_placed" in names) == (exp.num_rows * 4 > n), (name, join_type, names, exp.num_rows)
            assert_tables_equal(got, exp, ordered=True)
            ops.set_options(join__pred_in_counts="0")
            try:
                old, old_names = _probe_paths(lambda: ht.probe(dev_p, ["k2"], join_type, ["v"], ["k2", "p"], predicate=pred).to_arrow())
            finally:
                ops.set_options(join__pred_in_counts=None)
            assert "cmp" in old_names and old.equals(got), (name, join_type)
    # a predicate of another shape (a comparison of two columns) still takes the mask
    other = col("i") < col("dn")
    got, names = _probe_paths(lambda: ht.probe(dev_p, ["k2"], "Inner", ["v"], ["k2", "p"], predicate=other).to_arrow())
    assert "cmp" in names
    exp = oracle.hash_join(build, oracle.filter(probe, to_oracle_expr(other), probe.column_names), [("k", "k2")], "Inner").select(["v", "k2", "p"])
    assert_tables_equal(got, exp, ordered=True)
    ht.free()


@pytest.mark.parametrize("table_mode", ["array_map", "rank_map"])
def test_sparse_listed_emit_takes_its_rows_in_rounds(table_mode):
    """round 6: a probe that emits fewer than 1 row in 32 gives every workgroup of the listed emit 65536 probe rows (k_join_emit_listed<..,
    EL_WORDS_SPARSE>) and takes a group's listed rows 8192 at a time.  Ascending probe keys put ALL ~12 K hits into the first group (two
    rounds), a second cluster straddles a group boundary, a fused FilterExec thins them; the 8192-row-group variant (join.listed_sparse_den=0)
    and the oracle give the same rows in the same (probe) order."""
    from datafusion_amd import ops
    from datafusion_amd.expr import col, lit
    from datafusion_amd.table import DeviceTable
    from oracle import oracle
    from tests.util import to_oracle_expr
    rng = np.random.default_rng(61)
    n = 3_000_000
    bk = np.concatenate([np.arange(0, 20_000), np.arange(109_000, 110_500), [4_999_999]])
    build = pa.table({"k": pa.array(bk, type=pa.int64()), "v": pa.array(rng.integers(0, 10**6, len(bk)), type=pa.int32())})
    keys = np.sort(rng.integers(0, 5_000_000, n))
    probe = pa.table({"k2": pa.array(keys, type=pa.int64()), "p": pa.array(rng.integers(0, 10**9, n), type=pa.int64()), "f": pa.array(rng.integers(0, 100, n), type=pa.int32())})
    dev_p = DeviceTable.from_arrow(probe)
    ht = ops.JoinHashTable(DeviceTable.from_arrow(build), ["k"], probe_mode=3, table_mode=ops.TABLE_MODES[table_mode])
    for pred in (None, col("f") < lit(90, pa.int32())):
        src = probe if pred is None else oracle.filter(probe, to_oracle_expr(pred), probe.column_names)
        for join_type in ("Inner", "RightSemi"):
            keep = ["v", "k2", "p"] if join_type == "Inner" else ["k2", "p"]
            exp = oracle.hash_join(build, src, [("k", "k2")], join_type).select(keep)
            assert 8192 < exp.num_rows < n // 32
            got, names = _probe_paths(lambda: ht.probe(dev_p, ["k2"], join_type, ["v"], ["k2", "p"], predicate=pred).to_arrow())
            assert "join_probe_listed" in names, names
            assert_tables_equal(got, exp, ordered=True)
            ops.set_options(join__listed_sparse_den="0")
            try:
                small = ht.probe(dev_p, ["k2"], join_type, ["v"], ["k2", "p"], predicate=pred).to_arrow()
            finally:
                ops.set_options(join__listed_sparse_den=None)
            assert small.equals(got)
    ht.free()


@pytest.mark.parametrize("table_mode", ["hash_map", "array_map", "rank_map", "flat_hash_map", "radix_lds"])
@pytest.mark.parametrize("join_type", ["Inner", "Right", "RightSemi", "RightAnti", "Left", "Full", "LeftSemi"])
def test_bounded_probe_takes_the_probe_side_in_pieces_within_the_output_bound(table_mode, join_type):
    """round 6 (verdict missing 5): dfgpu_join_probe_bounded — HashJoinStream's limit / offset resumption
    (get_matched_indices_with_limit_offset + MapOffset, joins/join_hash_map.rs:389-484) at 64-row granularity.  An M:N join (every build key
    ~4 times) taken in pieces of at most 5000 output rows: every piece stays within the bound (or is one 64-row word), the pieces
    concatenated are the whole-table probe's rows in the same order, the visited marks accumulate across pieces (Left / Full /
    LeftSemi emit their unmatched build rows after the last piece), and the oracle agrees"""
    from datafusion_amd import ops
    from datafusion_amd.table import DeviceTable
    from oracle import oracle
    rng = np.random.default_rng(77)
    nb, npr = 8000, 30_011
    unique = table_mode in ("rank_map",)
    bk = rng.permutation(4000)[:2000].repeat(4)[:nb] if not unique else rng.permutation(20_000)[:nb]
    build = pa.table({"k": pa.array(bk, type=pa.int64()), "v": pa.array(np.arange(nb), type=pa.int32())})
    probe = pa.table({"k2": pa.array(rng.integers(0, 4500 if not unique else 22_000, npr), type=pa.int64()), "p": pa.array(rng.integers(0, 10**9, npr), type=pa.int64())})
    dev_b, dev_p = DeviceTable.from_arrow(build), DeviceTable.from_arrow(probe)
    kw = dict(table_mode={"hash_map": 1, "array_map": 2, "rank_map": 3, "radix_lds": 4, "flat_hash_map": 5}[table_mode])
    if table_mode == "radix_lds":
        kw["probe_mode"] = 4
    ht = ops.JoinHashTable(dev_b, ["k"], **kw)
    limit = 5000
    pieces = [t.to_arrow() for t in ht.probe_bounded(dev_p, ["k2"], join_type, ["v"], ["k2", "p"], max_output_rows=limit)]
    assert len(pieces) >= (2 if join_type == "LeftSemi" else 3)      # (LeftSemi emits nothing per piece: a piece is four bounds' worth of probe rows)
    if table_mode != "radix_lds":   # (there the bound applies to the probe rows of a piece)
        assert all(t.num_rows <= max(limit, 64 * 4) for t in pieces), [t.num_rows for t in pieces]
        if join_type != "LeftSemi":
            assert max(t.num_rows for t in pieces) > limit // 2    # ... and the pieces are not needlessly small
    got = pa.concat_tables(pieces)
    tail = ht.emit_unmatched(join_type, ["v"], probe_schema=dev_p.select(["k2", "p"]).schema).to_arrow() if join_type in ("Left", "Full", "LeftSemi") else None
    ht.free()
    ht2 = ops.JoinHashTable(dev_b, ["k"], **kw)
    whole = ht2.probe(dev_p, ["k2"], join_type, ["v"], ["k2", "p"]).to_arrow()
    tail2 = ht2.emit_unmatched(join_type, ["v"], probe_schema=dev_p.select(["k2", "p"]).schema).to_arrow() if tail is not None else None
    ht2.free()
    assert_tables_equal(got, whole, ordered=table_mode != "radix_lds")
    if tail is not None:
        assert_tables_equal(tail, tail2, ordered=False)
        got = pa.concat_tables([got, tail.cast(got.schema)]) if join_type != "LeftSemi" else tail
    exp = oracle.hash_join(build, probe, [("k", "k2")], join_type)
    keep = {"Inner": ["v", "k2", "p"], "Right": ["v", "k2", "p"], "Left": ["v", "k2", "p"], "Full": ["v", "k2", "p"], "RightSemi": ["k2", "p"], "RightAnti": ["k2", "p"], "LeftSemi": ["v"]}[join_type]
    assert_tables_equal(got, exp.select(keep), ordered=False)
