"""Variable-length strings in HBM (DFGPU_UTF8): import / export of Utf8, LargeUtf8 and Utf8View, comparisons and LIKE on the bytes,
row-selecting operators (filter, take through joins / sort), concatenation, and device-side interning
(dfgpu_table_dictionary_encode) against the reference's ArrowBytesMap known answers (physical-expr-common/src/binary_map.rs:649-697
test_string_set_basic, :877-893 test_map: distinct values in first-seen order, a NULL kept apart, re-inserting changes nothing) and
against pyarrow on random data.  String keys of joins / GROUP BY / ORDER BY run on the interned indices and are compared with the
oracle over pyarrow-encoded inputs."""
import numpy as np
import pyarrow as pa
import pyarrow.compute as pc
import pytest

pytestmark = pytest.mark.gpu

WORDS = ["", "a", "b", "ab", "abc", "cbcxx", "AAAAAAAA", "BBBBBQBBB", "CXCCCCCCCC", "bcdefghijklmnop", "qrstuvqxyzhjwya", "✨🔥", "🔥", "🔥🔥🔥🔥🔥🔥",
         "PROMO BURNISHED COPPER", "STANDARD ANODIZED TIN", "furiously special requests haggle", "50% off_", "naïve café", "x" * 100, "y" * 257]


def random_strings(rng, n, null_frac=0.0, pool=None):
    pool = WORDS if pool is None else pool
    idx = rng.integers(0, len(pool), size=n)
    mask = rng.random(n) < null_frac if null_frac else None
    return pa.array([pool[i] for i in idx], pa.string(), mask=mask)


@pytest.mark.parametrize("typ", [pa.string(), pa.large_string(), pa.string_view()])
@pytest.mark.parametrize("null_frac", [0.0, 0.2])
def test_import_export_round_trip(typ, null_frac):
    from datafusion_amd.table import DeviceTable
    rng = np.random.default_rng(1)
    for n in (0, 1, 63, 64, 1000):
        s = random_strings(rng, n, null_frac).cast(typ)
        t = pa.table({"i": pa.array(np.arange(n, dtype=np.int64)), "s": s})
        got = DeviceTable.from_arrow(t).to_arrow()
        assert got.column("s").cast(pa.string()).to_pylist() == s.cast(pa.string()).to_pylist()
        assert got.column("i").to_pylist() == list(range(n))
    sliced = pa.table({"s": random_strings(rng, 500, null_frac).cast(typ)}).slice(37, 201)       # an Arrow offset on the way in
    assert DeviceTable.from_arrow(sliced).to_arrow().column("s").cast(pa.string()).to_pylist() == sliced.column("s").cast(pa.string()).to_pylist()


def test_batched_export_of_strings():
    from datafusion_amd.table import DeviceTable
    rng = np.random.default_rng(2)
    s = random_strings(rng, 1000, 0.1)
    dev = DeviceTable.from_arrow(pa.table({"s": s}))
    out = pa.Table.from_batches(list(dev.to_batches(130)))
    assert out.column("s").to_pylist() == s.to_pylist()


def test_reference_string_set_basic_and_test_map():
    """binary_map.rs:655-697 and :877-893"""
    from datafusion_amd.table import DeviceTable
    values = ["a", "b", "CXCCCCCCCC", "", "cbcxx", None, "AAAAAAAA", "BBBBBQBBB", "a", "cbcxx", "b", "cbcxx", "", None, "BBBBBQBBB", "BBBBBQBBB", "AAAAAAAA", "CXCCCCCCCC"]
    enc = DeviceTable.from_arrow(pa.table({"s": pa.array(values, pa.string())})).dictionary_encode(sorted=False).to_arrow().column("s").combine_chunks()
    # "values must appear in the order they were inserted"; the set keeps ONE NULL entry, the dictionary array keeps NULL indices
    assert enc.dictionary.to_pylist() == ["a", "b", "CXCCCCCCCC", "", "cbcxx", "AAAAAAAA", "BBBBBQBBB"]
    assert enc.to_pylist() == values
    seq = ["A", "bcdefghijklmnop", "X", "Y", None, "qrstuvqxyzhjwya", "✨🔥", "🔥", "🔥🔥🔥🔥🔥🔥"]
    enc = DeviceTable.from_arrow(pa.table({"s": pa.array(seq + seq, pa.string())})).dictionary_encode(sorted=False).to_arrow().column("s").combine_chunks()
    assert enc.dictionary.to_pylist() == [v for v in seq if v is not None]             # "put it in twice": no new entries
    assert enc.indices.to_pylist() == [0, 1, 2, 3, None, 4, 5, 6, 7] * 2                # payload = index of first insertion


@pytest.mark.parametrize("null_frac", [0.0, 0.15])
@pytest.mark.parametrize("sorted_", [False, True])
def test_dictionary_encode_matches_pyarrow(null_frac, sorted_):
    from datafusion_amd.table import DeviceTable
    rng = np.random.default_rng(3)
    pool = WORDS + [f"Customer#{i:09d}" for i in range(5000)]
    s = random_strings(rng, 40000, null_frac, pool)
    enc = DeviceTable.from_arrow(pa.table({"s": s})).dictionary_encode(sorted=sorted_).to_arrow().column("s").combine_chunks()
    assert enc.to_pylist() == s.to_pylist()
    want = pc.dictionary_encode(s).dictionary.to_pylist()                               # pyarrow: first-seen order too
    assert enc.dictionary.to_pylist() == (sorted(want) if sorted_ else want)
    assert enc.indices.null_count == s.null_count


def test_dictionary_encode_edge_shapes():
    from datafusion_amd.table import DeviceTable
    for vals in ([], [None], [None] * 11, ["same"] * 1000, [""] * 3 + [None], [str(i) for i in range(3000)]):
        enc = DeviceTable.from_arrow(pa.table({"s": pa.array(vals, pa.string())})).dictionary_encode(sorted=False).to_arrow().column("s").combine_chunks()
        assert enc.to_pylist() == vals
        assert enc.dictionary.to_pylist() == list(dict.fromkeys(v for v in vals if v is not None))


@pytest.mark.parametrize("null_frac", [0.0, 0.2])
def test_comparisons_and_like_on_bytes(null_frac):
    from datafusion_amd import ops
    from datafusion_amd.expr import col, lit
    from datafusion_amd.table import DeviceTable
    rng = np.random.default_rng(4)
    n = 6000
    a, b = random_strings(rng, n, null_frac), random_strings(rng, n, null_frac)
    t = pa.table({"a": a, "b": b, "i": pa.array(np.arange(n, dtype=np.int64))})
    dev = DeviceTable.from_arrow(t)
    fns = {"=": pc.equal, "!=": pc.not_equal, "<": pc.less, "<=": pc.less_equal, ">": pc.greater, ">=": pc.greater_equal}
    from datafusion_amd.expr import BinaryExpr
    for op, f in fns.items():
        for literal in ("cbcxx", "", "zzz", "naïve café", "🔥"):
            want = t.filter(pc.fill_null(f(a, pa.scalar(literal)), False)).column("i").to_pylist()
            assert ops.filter(dev, BinaryExpr(col("a"), op, lit(literal, pa.string())), ["i"]).to_arrow().column("i").to_pylist() == want, (op, literal)
            wantr = t.filter(pc.fill_null(f(pa.scalar(literal), a), False)).column("i").to_pylist()
            assert ops.filter(dev, BinaryExpr(lit(literal, pa.string()), op, col("a")), ["i"]).to_arrow().column("i").to_pylist() == wantr, (op, literal, "literal first")
        want = t.filter(pc.fill_null(f(a, b), False)).column("i").to_pylist()
        assert ops.filter(dev, BinaryExpr(col("a"), op, col("b")), ["i"]).to_arrow().column("i").to_pylist() == want, op
    for pattern, ci in (("PROMO%", False), ("%requests%", False), ("%special%requests%", False), ("_b%", False), ("%", False), ("", False), ("a", False), ("%\\%%", False),
                        ("50\\% off\\_", False), ("%CAFÉ", False), ("promo%copper", True), ("🔥%", False), ("_", False), ("%x", False), ("__", False)):
        for negated in (False, True):
            keep = pc.match_like(a, pattern, ignore_case=ci)
            keep = pc.invert(keep) if negated else keep
            want = t.filter(pc.fill_null(keep, False)).column("i").to_pylist()
            got = ops.filter(dev, col("a").like(pattern, negated=negated, case_insensitive=ci), ["i"]).to_arrow().column("i").to_pylist()
            assert got == want, (pattern, ci, negated)
    # NULL literal: NULL everywhere -> no row passes
    assert ops.filter(dev, col("a").eq(lit(None, pa.string())), ["i"]).num_rows == 0


def test_filter_take_and_concat_move_strings():
    from datafusion_amd import ops
    from datafusion_amd.expr import col, lit
    from datafusion_amd.table import DeviceTable
    rng = np.random.default_rng(5)
    n = 20000
    t = pa.table({"k": pa.array(rng.integers(0, 500, size=n)), "s": random_strings(rng, n, 0.1), "v": pa.array(rng.integers(0, 10**6, size=n))})
    dev = DeviceTable.from_arrow(t)
    got = ops.filter(dev, col("v") < lit(300000, pa.int64())).to_arrow()
    want = t.filter(pc.less(t.column("v"), 300000))
    assert got.column("s").to_pylist() == want.column("s").to_pylist() and got.column("v").to_pylist() == want.column("v").to_pylist()
    # strings as join payload on both sides + sort output (take)
    build = pa.table({"bk": pa.array(np.arange(500, dtype=np.int64)), "name": pa.array([f"Supplier#{i:09d}" if i % 7 else None for i in range(500)], pa.string())})
    j = ops.hash_join(DeviceTable.from_arrow(build), dev, [("bk", "k")], "Inner").to_arrow()
    names = build.column("name").to_pylist()
    assert j.num_rows == n
    rows = sorted(zip(j.column("v").to_pylist(), j.column("k").to_pylist(), j.column("name").to_pylist(), j.column("s").to_pylist()), key=lambda r: (r[0], r[1], str(r[3])))
    exp = sorted(zip(t.column("v").to_pylist(), t.column("k").to_pylist(), [names[k] for k in t.column("k").to_pylist()], t.column("s").to_pylist()), key=lambda r: (r[0], r[1], str(r[3])))
    assert rows == exp
    srt = ops.sort(dev, [("v", False, False), ("k", False, False)]).to_arrow()
    order = np.lexsort((t.column("k").to_numpy(), t.column("v").to_numpy()))
    assert srt.column("s").to_pylist() == t.column("s").take(pa.array(order)).to_pylist()
    cat = DeviceTable.concat([dev, dev.select(["k", "s", "v"]), DeviceTable.from_arrow(t.slice(0, 5))]).to_arrow()
    assert cat.column("s").to_pylist() == t.column("s").to_pylist() * 2 + t.column("s").slice(0, 5).to_pylist()


def test_string_keys_run_on_interned_indices():
    """GROUP BY / join / ORDER BY on string keys: encode on the device, run on the indices; oracle over pyarrow-encoded input"""
    from datafusion_amd import ops
    from datafusion_amd.expr import col
    from datafusion_amd.table import DeviceTable
    rng = np.random.default_rng(6)
    n = 30000
    pool = [f"Customer#{i:09d}" for i in range(700)] + WORDS
    t = pa.table({"name": random_strings(rng, n, 0.05, pool), "v": pa.array(rng.integers(0, 1000, size=n))})
    dev = DeviceTable.from_arrow(t).dictionary_encode(["name"])
    g = ops.aggregate(dev, [(col("name"), "name")], [("sum", col("v"), "s"), ("count", None, "n")], "Single").to_arrow()
    got = {r["name"]: (r["s"], r["n"]) for r in pa.table({"name": g.column("name").cast(pa.string()), "s": g.column("s"), "n": g.column("n")}).to_pylist()}
    want = {}
    for name, v in zip(t.column("name").to_pylist(), t.column("v").to_pylist()):
        s, c = want.get(name, (0, 0))
        want[name] = (s + v, c + 1)
    assert got == want
    srt = ops.sort(dev, [("name", False, False), ("v", True, False)]).to_arrow()
    keys = [(x is None, x or "", -v) for x, v in zip(srt.column("name").cast(pa.string()).to_pylist(), srt.column("v").to_pylist())]
    assert keys == sorted(keys)
    other = pa.table({"name2": pa.array(pool[::3] + ["not there"], pa.string()), "w": pa.array(np.arange(len(pool[::3]) + 1, dtype=np.int64))})
    odev = DeviceTable.from_arrow(other).dictionary_encode(["name2"])                  # a different dictionary: the join unifies them
    j = ops.hash_join(odev, dev, [("name2", "name")], "Inner").to_arrow()
    w_of = dict(zip(other.column("name2").to_pylist(), other.column("w").to_pylist()))
    exp = sorted((nm, w_of[nm], v) for nm, v in zip(t.column("name").to_pylist(), t.column("v").to_pylist()) if nm in w_of)
    assert sorted(zip(j.column("name").cast(pa.string()).to_pylist(), j.column("w").to_pylist(), j.column("v").to_pylist())) == exp


def test_partition_on_string_keys_routes_on_the_bytes():
    """RepartitionExec Hash on a string key: Utf8 bytes and dictionary-encoded columns (whatever their dictionaries) send equal strings
    to the same partition — the partition contents are identical for the three encodings of one column, NULLs travel together"""
    from datafusion_amd import ops
    from datafusion_amd.table import DeviceTable
    rng = np.random.default_rng(31)
    n = 20000
    s = random_strings(rng, n, 0.05, [f"Customer#{i:09d}" for i in range(500)] + WORDS)
    t = pa.table({"s": s, "v": pa.array(np.arange(n, dtype=np.int64))})
    plain = DeviceTable.from_arrow(t)
    views = [plain, plain.dictionary_encode(["s"], sorted=True), plain.dictionary_encode(["s"], sorted=False)]
    results = []
    for d in views:
        parts = [p.to_arrow() for p in ops.partition(d, ["s"], 8)]
        assert [p.column_names for p in parts] == [["s", "v"]] * 8
        results.append([p.column("v").to_pylist() for p in parts])
        home = {}
        for q, p in enumerate(parts):
            for x in set(p.column("s").cast(pa.string()).to_pylist()):
                assert home.setdefault(x, q) == q
        assert sum(len(r) for r in results[-1]) == n and sum(1 for r in results[-1] if r) >= 6
    assert results[0] == results[1] == results[2]


@pytest.mark.parametrize("join_type", ["Inner", "Left", "LeftAnti", "RightSemi", "Full"])
def test_joins_on_utf8_keys_intern_them_inside_the_operator(join_type):
    """join keys that arrive as plain Utf8 columns on both sides (and Utf8 against dictionary-encoded): interned inside the join, the
    string columns come out as Utf8; duplicates on both sides, NULL keys, strings only one side holds; a second (integer) key"""
    from datafusion_amd import ops
    from datafusion_amd.table import DeviceTable
    from oracle import oracle
    rng = np.random.default_rng(21)
    pool = [f"Supplier#{i:09d}" for i in range(300)] + WORDS
    b = pa.table({"name": random_strings(rng, 2000, 0.05, pool[:250]), "k": pa.array(rng.integers(0, 3, size=2000), pa.int32()), "w": pa.array(np.arange(2000, dtype=np.int64))})
    p = pa.table({"name2": random_strings(rng, 9000, 0.05, pool[100:]), "k2": pa.array(rng.integers(0, 3, size=9000), pa.int32()), "v": pa.array(np.arange(9000, dtype=np.int64))})

    def codes(t, c):      # the oracle joins on integer codes of one shared dictionary (NULL stays NULL)
        at = {v: i for i, v in enumerate(pool)}
        return t.set_column(t.column_names.index(c), c, pa.array([None if v is None else at[v] for v in t.column(c).to_pylist()], pa.int32()))

    def decoded(t):
        cols = {n: (pa.array([None if v is None else pool[v] for v in t.column(n).to_pylist()], pa.string()) if n in ("name", "name2") else t.column(n)) for n in t.column_names}
        return pa.table(cols)
    for on in ([("name", "name2")], [("name", "name2"), ("k", "k2")]):
        want = decoded(oracle.hash_join(codes(b, "name"), codes(p, "name2"), on, join_type))
        for bd, pd in ((DeviceTable.from_arrow(b), DeviceTable.from_arrow(p)), (DeviceTable.from_arrow(b).dictionary_encode(["name"]), DeviceTable.from_arrow(p))):
            got = ops.hash_join(bd, pd, on, join_type).to_arrow()
            got = pa.table({n: (c.cast(pa.string()) if pa.types.is_dictionary(c.type) else c) for n, c in zip(got.column_names, got.columns)})
            assert got.column_names == want.column_names
            assert sorted(map(str, got.to_pylist())) == sorted(map(str, want.to_pylist()))
        if "name2" in want.column_names:
            assert ops.hash_join(DeviceTable.from_arrow(b), DeviceTable.from_arrow(p), on, join_type).schema.field("name2").type == pa.string()


@pytest.mark.parametrize("null_frac", [0.0, 0.1])
def test_group_by_and_order_by_on_utf8_columns_intern_them_inside_the_operator(null_frac):
    """GROUP BY / ORDER BY keys that arrive as plain Utf8 columns: the operator interns them (ascending dictionary), works on the
    indices and hands Utf8 back — Single, Partial -> Final over concatenated partitions, several keys, TopK"""
    from datafusion_amd import ops
    from datafusion_amd.expr import col
    from datafusion_amd.table import DeviceTable
    rng = np.random.default_rng(8)
    n = 20000
    pool = [f"Brand#{i}" for i in range(40)] + WORDS
    t = pa.table({"s": random_strings(rng, n, null_frac, pool), "k": pa.array(rng.integers(0, 5, size=n), pa.int32()), "u": random_strings(rng, n, null_frac, WORDS[:6]),
                  "v": pa.array(rng.integers(0, 1000, size=n))})
    dev = DeviceTable.from_arrow(t)

    def expected(keys):
        want = {}
        for row in t.to_pylist():
            kk = tuple(row[c] for c in keys)
            s, c = want.get(kk, (0, 0))
            want[kk] = (s + row["v"], c + 1)
        return want

    aggs = [("sum", col("v"), "t"), ("count", None, "n")]
    for keys in (["s"], ["s", "k", "u"]):
        gb = [(col(c), c) for c in keys]
        g = ops.aggregate(dev, gb, aggs, "Single").to_arrow()
        assert all(g.schema.field(c).type == t.schema.field(c).type for c in keys)            # Utf8 in, Utf8 out
        assert {tuple(r[c] for c in keys): (r["t"], r["n"]) for r in g.to_pylist()} == expected(keys)
        parts = [ops.aggregate(DeviceTable.from_arrow(t.slice(o, n // 4)), gb, aggs, "Partial") for o in range(0, n, n // 4)]
        f = ops.aggregate(DeviceTable.concat(parts), gb, aggs, "Final").to_arrow()
        assert {tuple(r[c] for c in keys): (r["t"], r["n"]) for r in f.to_pylist()} == expected(keys)
    # ORDER BY s ASC NULLS LAST, v DESC; then s DESC NULLS FIRST with a fetch
    srt = ops.sort(dev, [("s", False, False), ("v", True, False)]).to_arrow()
    assert srt.schema.field("s").type == pa.string()
    ks = [(x is None, (x or "").encode(), -v) for x, v in zip(srt.column("s").to_pylist(), srt.column("v").to_pylist())]
    assert ks == sorted(ks)
    assert sorted(map(str, srt.to_pylist())) == sorted(map(str, t.to_pylist()))
    top = ops.sort(dev, [("s", True, True), ("u", False, False), ("v", False, False)], fetch=50).to_arrow()
    allr = sorted(t.to_pylist(), key=lambda r: (r["s"] is not None, [-b for b in (r["s"] or "").encode()] + [1], r["u"] is None, (r["u"] or "").encode(), r["v"]))
    assert [(r["s"], r["u"], r["v"]) for r in top.to_pylist()] == [(r["s"], r["u"], r["v"]) for r in allr[:50]]


def test_slice_of_nullable_boolean_and_string_columns():
    """RecordBatch::slice on the device (dfgpu_table_slice): validity bits, bit-packed Booleans and strings at offsets that are not
    multiples of 64"""
    from datafusion_amd.table import DeviceTable
    rng = np.random.default_rng(7)
    n = 5000
    t = pa.table({"i": pa.array(rng.integers(0, 1000, size=n), mask=rng.random(n) < 0.2), "b": pa.array(rng.random(n) < 0.5, mask=rng.random(n) < 0.1),
                  "s": random_strings(rng, n, 0.15), "d": pa.array(rng.integers(0, 10**6, size=n).astype(np.int32)).cast(pa.decimal128(15, 2))})
    dev = DeviceTable.from_arrow(t)
    for off, ln in ((0, n), (1, 63), (63, 130), (64, 64), (1000, 3999), (4999, 1), (77, 0)):
        got = dev.slice(off, ln).to_arrow()
        want = t.slice(off, ln)
        for c in t.column_names:
            assert got.column(c).to_pylist() == want.column(c).to_pylist(), (off, ln, c)


@pytest.mark.parametrize("start,count", [(1, 2), (1, None), (3, 4), (0, 3), (-2, 5), (-5, 2), (4, 0), (50, 3), (2, 1000)])
def test_substr_utf8_and_dictionary_columns(start, count):
    """substr(string, start[, count]) (functions/src/unicode/substr.rs): 1-based start in characters, positions below 1 eat into the
    count; over a Utf8 column (device kernels, multi-byte characters, NULLs, empty strings) and over a dictionary-encoded column (the
    dictionary of the distinct substrings, ascending), both against Python's slicing of the code points"""
    from datafusion_amd import ops
    from datafusion_amd.expr import col, substr
    from datafusion_amd.table import DeviceTable
    rng = np.random.default_rng(3)
    pool = ["", "a", "ab", "13-989-741-2988", "31-768-687-3665", "żółć-gęślą", "日本語のテキスト", "x" * 70, "naïve café", "🙂🙃 ok"] + WORDS
    s = random_strings(rng, 4000, 0.1, pool)
    t = pa.table({"s": s, "v": pa.array(np.arange(4000))})

    def want(v):
        if v is None:
            return None
        first = max(start - 1, 0)
        return v[first:] if count is None else v[first:max(start - 1 + count, first)]
    exp = [want(v) for v in s.to_pylist()]
    got = ops.project(DeviceTable.from_arrow(t), [(substr(col("s"), start, count), "p"), (col("v"), "v")]).to_arrow()
    assert got.column("p").to_pylist() == exp
    enc = ops.project(DeviceTable.from_arrow(t).dictionary_encode(["s"]), [(substr(col("s"), start, count), "p")]).to_arrow()
    assert pa.types.is_dictionary(enc.schema.field("p").type)
    assert enc.column("p").cast(pa.string()).to_pylist() == exp
    d = enc.column("p").combine_chunks().dictionary.to_pylist()
    live = [x for x in d if x is not None]
    assert live == sorted(set(live))                              # distinct, ascending


def test_substr_results_compare_with_string_literals_and_group():
    """`substr(c_phone, 1, 2) IN ('13', '31', ...)` and GROUP BY on it (TPC-H Q22): a string literal compared with a computed
    dictionary column is bound to that column's dictionary inside the library; an absent string matches no row"""
    from datafusion_amd import _lib, ops
    from datafusion_amd.expr import col, lit, substr
    from datafusion_amd.table import DeviceTable
    rng = np.random.default_rng(4)
    phones = [f"{int(c):02d}-{int(a):03d}-{int(b):04d}" for c, a, b in zip(rng.integers(10, 35, 6000), rng.integers(100, 999, 6000), rng.integers(1000, 9999, 6000))]
    t = pa.table({"phone": pa.array(phones, pa.string()), "bal": pa.array(rng.integers(-999, 9999, 6000))})
    dev = DeviceTable.from_arrow(t).dictionary_encode(["phone"])
    cc = substr(col("phone"), 1, 2)
    codes = ["13", "31", "23", "99"]                                # "99" is in no row
    f = ops.filter(dev, cc.in_list([lit(c, pa.string()) for c in codes]))
    keep = [p[:2] in codes for p in phones]
    assert f.num_rows == sum(keep)
    g = ops.aggregate(ops.project(f, [(cc, "cc"), (col("bal"), "bal")]), [(col("cc"), "cc")], [("count", None, "n"), ("sum", col("bal"), "s")], "Single").to_arrow()
    got = {r["cc"]: (r["n"], r["s"]) for r in pa.table({"cc": g.column("cc").cast(pa.string()), "n": g.column("n"), "s": g.column("s")}).to_pylist()}
    want = {}
    for p, b, k in zip(phones, t.column("bal").to_pylist(), keep):
        if k:
            n, s = want.get(p[:2], (0, 0))
            want[p[:2]] = (n + 1, s + b)
    assert got == want
    assert ops.filter(dev, cc.ne(lit("99", pa.string()))).num_rows == 6000
    with pytest.raises(_lib.DfgpuError, match="negative substring length"):
        ops.project(dev, [(substr(col("phone"), 1, -1), "p")])


def test_sorted_dictionary_order_long_and_prefix_strings():
    """dictionary_encode(sorted=True): the dictionary is in byte order (= code point order) also for strings that agree in their
    first 24 bytes, that are prefixes of one another, that hold NUL or non-ASCII bytes — the host sort decides on three big-endian
    words first and must fall back correctly"""
    from datafusion_amd.table import DeviceTable
    rng = np.random.default_rng(12)
    base = "a fairly long common prefix, 30+"   # 32 bytes
    pool = [base + suffix for suffix in ("", "a", "b", "aa", "\x00", "\x00\x00", "z", "é", "zz")] + ["", "a", "a\x00", "a\x00b", "ab", "b", "é", "éa", "日本", "日本語", "x" * 24, "x" * 25, "x" * 23]
    pool += [f"Customer#{i:09d}" for i in rng.integers(0, 10**6, 9000)]
    pool = list(dict.fromkeys(pool))
    rows = [pool[int(j)] for j in rng.integers(0, len(pool), 50_000)]
    enc = DeviceTable.from_arrow(pa.table({"s": pa.array(rows, pa.string())})).dictionary_encode(["s"], sorted=True).to_arrow()
    d = enc.column("s").combine_chunks()
    values = d.dictionary.to_pylist()
    assert values == sorted(set(rows), key=lambda v: v.encode())
    assert d.cast(pa.string()).to_pylist() == rows


@pytest.mark.parametrize("pattern", ["Customer#00001%", "%45", "%er#0000%", "%", "%%", "a%", "%a", "%ż%", "żółw%", "%日本", "x%y", "_b%", "100\\%%", "%requests%", "%slyly ironic%"])
def test_like_affix_fast_paths_match_the_general_matcher(pattern):
    """LIKE 'lit%' / '%lit' / '%lit%' take a prefix / suffix compare or a literal search; every other shape takes the general
    matcher; both are checked against pyarrow's match_like on strings with multi-byte characters, empty strings and NULLs"""
    import pyarrow.compute as pc
    from datafusion_amd import ops
    from datafusion_amd.expr import col
    from datafusion_amd.table import DeviceTable
    rng = np.random.default_rng(21)
    pool = ["", "a", "ab", "xay", "x%y", "100%", "100% sure", "żółw", "żółw w wodzie", "日本", "東京と日本", "slyly ironic requests", "carefully ironic requests sleep"] + \
           [f"Customer#{i:09d}" for i in rng.integers(0, 10**5, 300)]
    s = random_strings(rng, 20_000, 0.1, pool)
    t = pa.table({"s": s, "k": pa.array(np.arange(20_000))})
    got = ops.filter(DeviceTable.from_arrow(t), col("s").like(pattern)).to_arrow()
    want = [i for i, h in enumerate(pc.match_like(s, pattern).to_pylist()) if h]
    assert got.column("k").to_pylist() == want


@pytest.mark.parametrize("distinct", [300, 1_200_000], ids=["few_distinct", "mostly_distinct"])
def test_dictionary_encode_first_seen_order_from_few_to_many_distinct_strings(distinct):
    """device interning numbers the strings in first-seen order (ArrowBytesMap::insert_if_new hands out payloads in that order,
    binary_map.rs) whether a handful of strings repeat a million times or most rows carry a string of their own"""
    import pyarrow.compute as pc

    from datafusion_amd.table import DeviceTable
    rng = np.random.default_rng(distinct)
    n = 1_500_000
    pool = pa.array([f"value-{i:07d}-{'x' * (i % 9)}" for i in range(distinct)], pa.string())
    s = pool.take(pa.array(rng.integers(0, distinct, n)))
    enc = DeviceTable.from_arrow(pa.table({"s": s})).dictionary_encode(sorted=False).to_arrow().column("s").combine_chunks()
    want = pc.dictionary_encode(s)
    assert enc.dictionary.to_pylist() == want.dictionary.to_pylist()
    assert enc.indices.to_numpy().tolist() == want.indices.to_numpy().tolist()


@pytest.mark.parametrize("shape", ["short_keys", "long_common_prefixes", "prefix_of_each_other"])
def test_ascending_dictionary_of_many_distinct_strings_is_ordered_on_the_device(shape):
    """from 16 Ki distinct strings on, the ascending dictionary is ordered by the device (three big-endian prefix words through the
    sort operator) and the host only settles runs that agree in their first 24 bytes: byte order (= pyarrow's sort of the distinct
    strings) for short keys, for URLs that share 30 bytes, and for strings that are prefixes of each other around the 24-byte mark"""
    import pyarrow.compute as pc

    from datafusion_amd.table import DeviceTable
    rng = np.random.default_rng(5)
    distinct = 40_000
    if shape == "short_keys":
        pool = [f"k{int(x):09d}" for x in rng.permutation(10**9)[:distinct] % 10**9]
    elif shape == "long_common_prefixes":
        pool = [f"https://www.example.com/catalog/item/{int(x):07d}/{'z' * int(x % 5)}" for x in rng.permutation(distinct * 3)[:distinct]]
    else:
        base = "abcdefghijklmnopqrstuvw"                                      # 23 bytes
        pool = list({base[: int(a)] + "x" * int(b) + str(int(c)) for a, b, c in zip(rng.integers(20, 24, distinct * 2), rng.integers(0, 6, distinct * 2), rng.integers(0, 3000, distinct * 2))})[:distinct]
    pool = sorted(set(pool), key=lambda s: rng.random())                      # first-seen order is not the sorted order
    s = pa.array(pool, pa.string()).take(pa.array(rng.integers(0, len(pool), 400_000)))
    enc = DeviceTable.from_arrow(pa.table({"s": s})).dictionary_encode(sorted=True).to_arrow().column("s").combine_chunks()
    want = pc.sort_indices(pc.unique(s))
    assert enc.dictionary.to_pylist() == pc.unique(s).take(want).to_pylist()
    assert enc.dictionary.take(enc.indices).to_pylist() == s.to_pylist()
