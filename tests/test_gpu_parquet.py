"""Scan -> device on the GPU: every column chunk of files written by pyarrow — three codecs x data page
v1 / v2 x dictionary on / off, small pages, several row groups, NULLs — decoded by dfgpu_parquet_decode_chunk and compared
with pyarrow's own reader (an independent production decoder of the same bytes), bit for bit."""
import ctypes.util  # noqa: F401

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq
import pytest

from tests.parquet_cases import WRITER_MATRIX, case_id, sample_table, write
from tests.util import assert_tables_equal

pytestmark = pytest.mark.gpu


def plain(t: pa.Table) -> pa.Table:
    """dictionary-encoded string columns -> strings (comparison form)"""
    return pa.table({n: (c.cast(pa.string()) if pa.types.is_dictionary(c.type) else c) for n, c in zip(t.column_names, t.columns)})


@pytest.mark.parametrize("writer", WRITER_MATRIX, ids=[case_id(w) for w in WRITER_MATRIX])
def test_decode_matches_pyarrow(tmp_path, writer):
    from datafusion_amd.parquet import read_table
    t = sample_table(40_000)
    path = write(t, tmp_path, "t.parquet", data_page_size=8 * 1024, row_group_size=17_000, **writer)
    got = read_table(path).to_arrow()
    assert pa.types.is_dictionary(got.schema.field("s").type)
    assert_tables_equal(plain(got), pq.read_table(path), ordered=True)


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 1000])
def test_small_tables_and_word_boundaries(tmp_path, n):
    from datafusion_amd.parquet import read_table
    t = sample_table(n, seed=n)
    path = write(t, tmp_path, "s.parquet", compression="snappy", row_group_size=max(1, n // 3 + 1))   # row groups that split validity words
    assert_tables_equal(plain(read_table(path).to_arrow()), pq.read_table(path), ordered=True)


def test_decimals_stored_as_integers_and_column_projection(tmp_path):
    from decimal import Decimal

    from datafusion_amd.parquet import read_table
    rng = np.random.default_rng(3)
    n = 20_000
    t = pa.table({"d9": pa.array([Decimal(int(x)) / 100 for x in rng.integers(-10**6, 10**6, n)], pa.decimal128(9, 2)),
                  "d18": pa.array([Decimal(int(x)) / 1000 for x in rng.integers(-10**15, 10**15, n)], pa.decimal128(18, 3)),
                  "k": pa.array(rng.integers(0, 10**12, n))})
    path = str(tmp_path / "d.parquet")
    pq.write_table(t, path, store_decimal_as_integer=True, compression="zstd")
    meta = pq.ParquetFile(path).metadata.row_group(0)
    assert meta.column(0).physical_type == "INT32" and meta.column(1).physical_type == "INT64"
    assert_tables_equal(read_table(path).to_arrow(), t, ordered=True)
    assert_tables_equal(read_table(path, ["k", "d9"]).to_arrow(), t.select(["k", "d9"]), ordered=True)


@pytest.mark.parametrize("version", ["1.0", "2.0"])
@pytest.mark.parametrize("compression", ["none", "snappy", "zstd"])
def test_delta_byte_stream_split_and_boolean_pages(tmp_path, compression, version):
    """the encodings beyond PLAIN / dictionary: DELTA_BINARY_PACKED (Int32 / Int64 / Date32 incl. wrapping deltas and NULLs),
    BYTE_STREAM_SPLIT (Float64), BOOLEAN columns (PLAIN bits in v1 pages, RLE runs in v2) — against pyarrow's reader"""
    from datafusion_amd.parquet import read_table
    rng = np.random.default_rng(11)
    n = 50_003
    wide = rng.integers(-2**62, 2**62, n)
    wide[:4] = [np.iinfo(np.int64).max, np.iinfo(np.int64).min, 0, -1]                            # deltas that wrap
    t = pa.table({"d64": pa.array(np.cumsum(rng.integers(-3, 50, n)).astype(np.int64)), "wide": pa.array(wide),
                  "d32": pa.array(rng.integers(-2**31, 2**31 - 1, n).astype(np.int32)), "d32n": pa.array(rng.integers(0, 1000, n).astype(np.int32), mask=rng.random(n) < 0.3),
                  "dt": pa.array(np.sort(rng.integers(8000, 10000, n)).astype(np.int32), pa.int32()).cast(pa.date32()),
                  "f": pa.array(rng.normal(size=n)), "fn": pa.array(rng.random(n), mask=rng.random(n) < 0.2),
                  "b": pa.array(rng.random(n) < 0.5), "bn": pa.array(rng.random(n) < 0.1, mask=rng.random(n) < 0.25), "runs": pa.array(np.repeat([True, False, True], [20_000, 20_000, 10_003]))})
    path = str(tmp_path / "e.parquet")
    pq.write_table(t, path, compression=compression, data_page_version=version, use_dictionary=False, data_page_size=16 * 1024, row_group_size=21_000,
                   column_encoding={"d64": "DELTA_BINARY_PACKED", "wide": "DELTA_BINARY_PACKED", "d32": "DELTA_BINARY_PACKED", "d32n": "DELTA_BINARY_PACKED", "dt": "DELTA_BINARY_PACKED",
                                    "f": "BYTE_STREAM_SPLIT", "fn": "BYTE_STREAM_SPLIT", "b": "PLAIN", "bn": "PLAIN", "runs": "PLAIN" if version == "1.0" else "RLE"})
    got = read_table(path).to_arrow()
    assert_tables_equal(got, pq.read_table(path), ordered=True)
    assert got.column("bn").null_count == t.column("bn").null_count and got.schema.field("b").type == pa.bool_()


def test_q6_straight_from_a_parquet_file(tmp_path):
    """dbgen lineitem -> Parquet (ZSTD, the reference's benchmark setting) -> device -> the reference's Q6 plan -> the reference's answer"""
    from datafusion_amd import physical_plan as P
    from tests import tpch_plans as T
    from datafusion_amd.parquet import read_table
    from oracle import dbgen
    from tests.test_tpch_answers import assert_answer
    _, _, l = dbgen.tables(0.1, "utf8")
    path = str(tmp_path / "lineitem.parquet")
    pq.write_table(l, path, compression="zstd", compression_level=1)
    li = read_table(path, ["l_quantity", "l_extendedprice", "l_discount", "l_shipdate", "l_returnflag", "l_linestatus", "l_tax"])
    assert li.num_rows == l.num_rows
    assert_answer("q6", P.collect(P.GpuOffloadRule().optimize(T.q6_plan(li))).to_arrow())
    assert_answer("q1", P.collect(P.GpuOffloadRule().optimize(T.q1_plan(li))).to_arrow())
    # the same with the scan as the plan's leaf: ParquetExec = DataSourceExec over the file, owned output
    leaf = P.ParquetExec(path, ["l_quantity", "l_extendedprice", "l_discount", "l_shipdate"], "lineitem")
    node = T.q6_plan(li)
    while not isinstance(node, P.FilterExec):
        node = node.children()[0]
    f = P.FilterExec(node.predicate, leaf, projection=["l_extendedprice", "l_discount"])
    name = "sum(lineitem.l_extendedprice * lineitem.l_discount)"
    from datafusion_amd.expr import col
    plan = P.ProjectionExec([(col(name), "revenue")], P.AggregateExec("Single", [], [("sum", col("l_extendedprice") * col("l_discount"), name)], f))
    assert_answer("q6", P.collect(P.GpuOffloadRule().optimize(plan)).to_arrow())


@pytest.mark.parametrize("join_type", ["Inner", "RightSemi", "Right"])
def test_join_bounds_prune_the_probe_side_scan(tmp_path, join_type):
    """the hash join's dynamic filter (hash_join/shared_bounds.rs:277-284): [min, max] of the build keys reaches the probe-side
    ParquetExec, which skips row groups by their footer statistics — same join result, fewer row groups read; join types that
    emit unmatched probe rows (Right) must not prune"""
    from datafusion_amd import physical_plan as P
    from datafusion_amd.expr import col, lit
    from datafusion_amd.table import DeviceTable
    from tests import plan_oracle
    rng = np.random.default_rng(5)
    n = 60_000
    probe = pa.table({"l_orderkey": pa.array(np.sort(rng.integers(0, 20_000, n)).astype(np.int64)), "l_qty": pa.array(rng.integers(1, 50, n).astype(np.int32))})
    path = str(tmp_path / "probe.parquet")
    pq.write_table(probe, path, row_group_size=5_000, compression="snappy")
    build = pa.table({"o_orderkey": pa.array(np.arange(7_000, 7_400, dtype=np.int64)), "o_flag": pa.array(np.arange(400, dtype=np.int32) % 3)})

    def plan(build_leaf):
        scan = P.ParquetExec(path, ["l_orderkey", "l_qty"], "lineitem")
        probe_side = P.CoalesceBatchesExec(P.FilterExec(col("l_qty") > lit(10, pa.int32()), scan))
        return P.HashJoinExec(build_leaf, probe_side, [("o_orderkey", "l_orderkey")], join_type), scan
    dev_build = DeviceTable.from_arrow(build)
    j, scan = plan(P.MemoryExec(dev_build, "orders"))
    got = P.collect(j).to_arrow()
    exp = plan_oracle.collect(plan(P.MemoryExec(build, "orders"))[0])
    assert_tables_equal(got, exp, ordered=False)
    assert scan.metrics["row_groups_total"] == 12
    if join_type == "Right":
        assert scan.dynamic_bounds == {} and scan.metrics["row_groups_read"] == 12
    else:
        assert scan.dynamic_bounds == {"l_orderkey": (7_000, 7_399)} and 1 <= scan.metrics["row_groups_read"] <= 3
    # through the rule (filter fused into the probe) the scan is still pruned
    j2, scan2 = plan(P.MemoryExec(dev_build, "orders"))
    opt = P.GpuOffloadRule().optimize(P.AggregateExec("Single", [], [("count", None, "n")], j2))
    n_rows = P.collect(opt).to_arrow().to_pylist()[0]["n"]
    assert n_rows == exp.num_rows and scan2.metrics["row_groups_read"] == scan.metrics["row_groups_read"]
    # an empty build side reads nothing
    j3, scan3 = plan(P.MemoryExec(DeviceTable.from_arrow(build.slice(0, 0)), "orders"))
    out3 = P.collect(j3).to_arrow()
    if join_type != "Right":
        assert out3.num_rows == 0 and scan3.metrics["row_groups_read"] == 0


def test_small_build_side_pushes_an_in_list_that_prunes_between_the_bounds(tmp_path):
    """PushdownStrategy::InList (hash_join/shared_bounds.rs:275-284; limits of exec.rs:2727-2751: 128 KiB and 150 distinct keys):
    three far-apart build keys have bounds that cover every row group, the IN list keeps only the groups that hold one of them;
    a build side beyond the limits pushes bounds only (Map strategy)"""
    from datafusion_amd import ops, physical_plan as P
    from datafusion_amd.table import DeviceTable
    from tests import plan_oracle
    n = 60_000
    probe = pa.table({"l_orderkey": pa.array(np.arange(n, dtype=np.int64) // 3), "l_qty": pa.array((np.arange(n) % 50).astype(np.int32))})
    path = str(tmp_path / "probe.parquet")
    pq.write_table(probe, path, row_group_size=5_000, compression="snappy")
    for keys, want_list, groups in (([10, 9_999, 19_990, 10, None], [10, 9_999, 19_990], 3), (list(range(0, 20_000, 100)), None, 12), ([], [], 0)):
        build = pa.table({"o_orderkey": pa.array(keys, pa.int64()), "o_flag": pa.array(np.arange(len(keys), dtype=np.int32))})
        dev_build = DeviceTable.from_arrow(build)
        assert ops.column_inlist(dev_build, "o_orderkey") == want_list
        scan = P.ParquetExec(path, ["l_orderkey", "l_qty"], "lineitem")
        j = P.HashJoinExec(P.MemoryExec(dev_build, "orders"), scan, [("o_orderkey", "l_orderkey")], "Inner")
        got = P.collect(j).to_arrow()
        exp = plan_oracle.collect(P.HashJoinExec(P.MemoryExec(build, "orders"), P.MemoryExec(probe, "lineitem"), [("o_orderkey", "l_orderkey")], "Inner"))
        assert_tables_equal(got, exp, ordered=False)
        assert scan.metrics["row_groups_total"] == 12 and scan.metrics["row_groups_read"] == groups, (keys[:3], scan.metrics)
        assert scan.dynamic_in_lists == ({} if want_list is None else {"l_orderkey": want_list})
    # the reference's knobs
    small = DeviceTable.from_arrow(pa.table({"k": pa.array(np.arange(200, dtype=np.int64))}))
    assert ops.column_inlist(small, "k") is None and ops.column_inlist(small, "k", max_distinct_values=200) == list(range(200))
    assert ops.column_inlist(small, "k", max_size=1000, max_distinct_values=1000) is None and ops.column_inlist(small, "k", max_distinct_values=0) is None


@pytest.mark.parametrize("join_type", ["Inner", "RightSemi"])
def test_large_build_side_pushes_its_table_as_a_membership_filter(tmp_path, join_type):
    """PushdownStrategy::Map (hash_join/shared_bounds.rs:275-284, exec.rs:2727-2751; HashTableLookupExpr, partitioned_hash_eval.rs:278):
    a build side beyond the IN-list limits pushes the built table itself.  300 keys in two far-apart clusters: their bounds cover
    nearly every row group, the membership test — asked on each row group's key chunk before anything else is decoded — keeps the two
    that hold them, and only the matching rows of those leave the scan.  Same join result."""
    from datafusion_amd import physical_plan as P
    from datafusion_amd.expr import col, lit
    from datafusion_amd.table import DeviceTable
    from tests import plan_oracle
    n = 60_000
    probe = pa.table({"l_orderkey": pa.array(np.arange(n, dtype=np.int64) // 3), "l_qty": pa.array((np.arange(n) % 50).astype(np.int32)),
                      "l_tag": pa.array([["a", "bb", "ccc"][i % 3] for i in range(n)], pa.string())})
    path = str(tmp_path / "probe.parquet")
    pq.write_table(probe, path, row_group_size=5_000, compression="snappy")
    keys = list(range(1_000, 1_150)) + list(range(18_000, 18_150))
    build = pa.table({"o_orderkey": pa.array(keys, pa.int64()), "o_flag": pa.array(np.arange(len(keys), dtype=np.int32))})
    dev_build = DeviceTable.from_arrow(build)
    scan = P.ParquetExec(path, ["l_orderkey", "l_qty", "l_tag"], "lineitem")
    j = P.HashJoinExec(P.MemoryExec(dev_build, "orders"), scan, [("o_orderkey", "l_orderkey")], join_type)
    got = P.collect(j).to_arrow()
    exp = plan_oracle.collect(P.HashJoinExec(P.MemoryExec(build, "orders"), P.MemoryExec(probe, "lineitem"), [("o_orderkey", "l_orderkey")], join_type))
    strings = lambda t: pa.table({c: (t.column(c).cast(pa.string()) if c == "l_tag" else t.column(c)) for c in t.column_names})
    assert_tables_equal(strings(got), strings(exp), ordered=False)
    m = scan.metrics
    assert scan.dynamic_in_lists == {} and scan.dynamic_bounds == {"l_orderkey": (1_000, 18_149)} and "l_orderkey" in scan.dynamic_membership
    assert m["row_groups_total"] == 12 and m["row_groups_read"] == 11            # the statistics drop one row group (keys >= 18 333)
    assert m["row_groups_skipped_by_membership"] == 9 and m["rows_passed"] == 900 and m["rows_scanned"] == 55_000
    # a FilterExec between the join and the scan, fused into the probe by the rule: still pushed down
    scan2 = P.ParquetExec(path, ["l_orderkey", "l_qty", "l_tag"], "lineitem")
    j2 = P.HashJoinExec(P.MemoryExec(dev_build, "orders"), P.FilterExec(col("l_qty") > lit(10, pa.int32()), scan2), [("o_orderkey", "l_orderkey")], join_type)
    opt = P.GpuOffloadRule().optimize(P.AggregateExec("Single", [], [("count", None, "n")], j2))
    n_rows = P.collect(opt).to_arrow().to_pylist()[0]["n"]
    exp2 = plan_oracle.collect(P.HashJoinExec(P.MemoryExec(build, "orders"), P.FilterExec(col("l_qty") > lit(10, pa.int32()), P.MemoryExec(probe, "lineitem")),
                                              [("o_orderkey", "l_orderkey")], join_type))
    assert n_rows == exp2.num_rows and scan2.metrics["row_groups_skipped_by_membership"] == 9
    # a second run of the same plan over a SMALL build side: the IN list takes over and the stale table is gone
    small = DeviceTable.from_arrow(build.slice(0, 5))
    j.left = P.MemoryExec(small, "orders")
    P.collect(j)
    assert scan.dynamic_membership == {} and scan.dynamic_in_lists == {"l_orderkey": list(range(1_000, 1_005))}


def test_device_chunk_cache_serves_repeated_scans(tmp_path):
    """a second scan of the same file takes its column chunks from HBM (no decode); a rewritten file is a new identity; the
    byte budget evicts least recently used chunks"""
    from datafusion_amd import parquet as P
    from datafusion_amd.parquet import ChunkCache, read_table
    n = 40_000
    t = pa.table({"k": pa.array(np.arange(n, dtype=np.int64)), "v": pa.array((np.arange(n) % 97).astype(np.int32)),
                  "s": pa.array([["x", "yy", "zzz"][i % 3] for i in range(n)], pa.string())})
    path = str(tmp_path / "t.parquet")
    pq.write_table(t, path, row_group_size=10_000, compression="snappy")
    saved = P.CACHE
    try:
        P.CACHE = ChunkCache(budget=1 << 30)
        a = read_table(path).to_arrow()
        assert P.CACHE.stats()["misses"] == 12 and P.CACHE.stats()["hits"] == 0 and P.CACHE.stats()["chunks"] == 12
        b = read_table(path, ["v", "k"]).to_arrow()
        assert P.CACHE.stats()["hits"] == 8 and P.CACHE.stats()["misses"] == 12
        assert_tables_equal(a, pa.table({"k": t.column("k"), "v": t.column("v"), "s": t.column("s").dictionary_encode().cast(pa.dictionary(pa.int32(), pa.string()))}), ordered=True)
        assert b.column("v").to_pylist() == t.column("v").to_pylist() and b.column("k").to_pylist() == t.column("k").to_pylist()
        t2 = t.set_column(1, "v", pa.array(np.zeros(n, dtype=np.int32)))
        import time
        time.sleep(0.01)
        pq.write_table(t2, path, row_group_size=10_000, compression="snappy")
        assert read_table(path, ["v"]).to_arrow().column("v").to_pylist() == [0] * n          # not the stale chunks
        P.CACHE = ChunkCache(budget=200_000)                                                     # ~ two 80 KB chunks of `k`
        read_table(path, ["k"])
        st = P.CACHE.stats()
        assert 1 <= st["chunks"] < 4 and st["bytes"] <= 200_000
        P.CACHE = ChunkCache(budget=0)
        read_table(path, ["k"])
        assert P.CACHE.stats()["chunks"] == 0
    finally:
        P.CACHE.clear()
        P.CACHE = saved


@pytest.mark.parametrize("fmt", ["file", "stream", "file_zstd"])
def test_arrow_ipc_scan(tmp_path, fmt):
    """ArrowSource (datasource-arrow/src/source.rs:260): IPC file / stream -> device, projection, several record batches, strings and
    dictionaries, buffer compression; the second scan is served from the device cache; a plan over the scan gives the oracle's answer"""
    import pyarrow.ipc as ipc

    from datafusion_amd import parquet as P, physical_plan as PP
    from datafusion_amd.expr import col, lit
    from datafusion_amd.ipc import read_table
    from tests import plan_oracle
    rng = np.random.default_rng(3)
    n = 25_000
    t = pa.table({"k": pa.array(rng.integers(0, 1000, n)), "v": pa.array(rng.integers(0, 10**6, n).astype(np.int32), mask=rng.random(n) < 0.1),
                  "s": pa.array([["x", "yy", None, "zzz"][i % 4] for i in range(n)], pa.string()),
                  "d": pa.array([["a", "b", "c"][i % 3] for i in range(n)], pa.string()).dictionary_encode()})
    path = str(tmp_path / "t.arrow")
    opts = ipc.IpcWriteOptions(compression="zstd") if fmt == "file_zstd" else None
    with (ipc.new_stream(path, t.schema) if fmt == "stream" else ipc.new_file(path, t.schema, options=opts)) as w:
        for b in t.to_batches(max_chunksize=6000):
            w.write_batch(b)
    saved = P.CACHE
    try:
        P.CACHE = P.ChunkCache(budget=1 << 30)
        stats = {}
        got = read_table(path, ["k", "s", "v", "d"], stats=stats).to_arrow()
        assert stats == {"record_batches": 5, "record_batches_from_cache": 0}
        for c in ("k", "s", "v"):
            assert got.column(c).to_pylist() == t.column(c).to_pylist(), c
        assert got.column("d").cast(pa.string()).to_pylist() == t.column("d").cast(pa.string()).to_pylist()
        read_table(path, ["k", "s", "v", "d"], stats=stats)
        assert stats["record_batches_from_cache"] == 5
        scan = PP.ArrowIpcExec(path, ["k", "v"], "t")
        plan = PP.AggregateExec("Single", [(col("k"), "k")], [("sum", col("v"), "s"), ("count", None, "n")], PP.FilterExec(col("v") > lit(1000, pa.int32()), scan))
        got = PP.collect(PP.GpuOffloadRule().optimize(plan)).to_arrow()
        exp = plan_oracle.collect(PP.AggregateExec("Single", [(col("k"), "k")], [("sum", col("v"), "s"), ("count", None, "n")],
                                                   PP.FilterExec(col("v") > lit(1000, pa.int32()), PP.MemoryExec(t.select(["k", "v"]), "t"))))
        assert_tables_equal(got, exp, ordered=False)
    finally:
        P.CACHE.clear()
        P.CACHE = saved


@pytest.mark.parametrize("compression", [None, "zstd", "lz4"])
@pytest.mark.parametrize("fmt", ["file", "stream"])
def test_arrow_ipc_reader_below_the_c_abi_every_type(tmp_path, fmt, compression):
    """dfgpu_ipc_open / dfgpu_ipc_read_batch (csrc/ipc.hip) over every column type the device knows — integers, Float64, Date32, Decimal128,
    Boolean, Utf8 / LargeUtf8 / Utf8View, dictionary-encoded strings, NULLs — in both framings and with compressed buffers, batch by
    batch and with a projection, against the table pyarrow wrote"""
    import ctypes as C

    from datafusion_amd import _lib
    from datafusion_amd.ipc import IpcFile, read_table
    from tests.test_ipc_host import sample, write
    t = sample(10_001)
    path = str(tmp_path / "t.arrow")
    write(t, path, fmt, compression)
    import ctypes.util
    if compression == "lz4" and not ctypes.util.find_library("lz4"):
        with pytest.raises(_lib.DfgpuError, match="liblz4"):
            read_table(path)
        return
    norm = lambda x: pa.table({n: (c.cast(pa.string()) if (pa.types.is_dictionary(c.type) or c.type in (pa.large_string(), pa.string_view())) else c)
                               for n, c in zip(x.column_names, x.columns)})
    got = read_table(path).to_arrow()
    assert_tables_equal(norm(got), norm(t), ordered=True)
    f = IpcFile(path)
    b1 = f.read_batch(1, ["dec", "sv", "b", "dict"]).to_arrow()
    assert_tables_equal(norm(b1), norm(t.slice(4000, 4000).select(["dec", "sv", "b", "dict"])), ordered=True)
    f.close()


@pytest.mark.parametrize("damage", ["node_longer_than_its_buffers", "node_differs_from_batch", "string_offsets_beyond_data", "dictionary_index_out_of_range"])
def test_arrow_ipc_reader_rejects_files_whose_metadata_overruns_their_buffers(tmp_path, damage):
    """a truncated / corrupt IPC file must fail in dfgpu_ipc_read_batch — the field node's row count is checked against every
    buffer's length, the last string offset against the data buffer, dictionary indices against the dictionary — instead of
    letting dfgpu_table_import read beyond the host mapping (arrow-ipc validates the same for untrusted input)"""
    import struct

    import pyarrow.ipc as ipc

    from datafusion_amd import _lib
    from datafusion_amd.ipc import IpcFile
    n = 12_345   # a row count whose 8 little-endian bytes appear nowhere else in the file
    t = pa.table({"a": pa.array(np.arange(n, dtype=np.int64) * 3), "s": pa.array(["v%d" % (i % 50) for i in range(n)], pa.string()),
                  "d": pa.array([["x", "yy", "zzz"][i % 3] for i in range(n)], pa.string()).dictionary_encode()})
    path = str(tmp_path / "t.arrow")
    with ipc.new_stream(path, t.schema) as w:
        w.write_batch(t.to_batches()[0])
    raw = bytearray(open(path, "rb").read())
    IpcFile(path).read_batch(0).free()       # intact: reads
    pat = struct.pack("<q", n)
    hits = [i for i in range(len(raw) - 8) if raw[i:i + 8] == pat]
    assert len(hits) >= 4                    # RecordBatch.length + one FieldNode.length per column
    if damage == "node_longer_than_its_buffers":
        for h in hits:                        # batch and nodes agree on a row count the buffers do not hold
            raw[h:h + 8] = struct.pack("<q", n * 1000)
        needle = "shorter than its rows"
    elif damage == "node_differs_from_batch":
        raw[hits[1]:hits[1] + 8] = struct.pack("<q", n - 1)
        needle = "differs from the record batch"
    elif damage == "string_offsets_beyond_data":
        last = struct.pack("<i", sum(len("v%d" % (i % 50)) for i in range(n)))     # the final offset of column s
        at = raw.rfind(last)
        assert at > 0
        raw[at:at + 4] = struct.pack("<i", 2**30)
        needle = "last string offset"
    else:
        idx = np.frombuffer(t.column("d").chunk(0).indices.buffers()[1], dtype=np.int32).tobytes()
        at = raw.find(idx[:64])
        assert at > 0
        raw[at:at + 4] = struct.pack("<i", 77)
        needle = "dictionary index"
    bad = str(tmp_path / "bad.arrow")
    open(bad, "wb").write(bytes(raw))
    f = IpcFile(bad)
    with pytest.raises(_lib.DfgpuError, match=needle):
        f.read_batch(0)
    f.close()


def test_parquet_chunk_with_dictionary_fallback_pages(tmp_path):
    """one column chunk holding dictionary-encoded pages followed by PLAIN pages (the writer's dictionary limit was reached)"""
    from datafusion_amd.parquet import ParquetFile, read_table
    n = 50_000
    t = pa.table({"k": pa.array(np.arange(n, dtype=np.int64) * 7), "v": pa.array((np.arange(n) % 97).astype(np.int32))})
    path = str(tmp_path / "fallback.parquet")
    pq.write_table(t, path, dictionary_pagesize_limit=4096, data_page_size=8192, compression="snappy")
    f = ParquetFile(path)
    info = f.inspect_chunk(0, "k")
    f.close()
    assert info["n_dictionary_encoded_pages"] >= 1 and info["n_plain_pages"] >= 1
    assert_tables_equal(read_table(path).to_arrow(), t, ordered=True)


def test_scan_sharded_by_rank(tmp_path):
    """ParquetFile.row_groups_for_rank: the shares of a 3-GPU scan decode to exactly the file, in order"""
    from datafusion_amd.parquet import ParquetFile
    from datafusion_amd.table import DeviceTable
    n = 20_000
    t = pa.table({"k": pa.array(np.arange(n, dtype=np.int64)), "d": pa.array((np.arange(n) % 13).astype(np.int32))})
    path = str(tmp_path / "s.parquet")
    pq.write_table(t, path, row_group_size=1500)
    f = ParquetFile(path)
    parts = [f.read(row_groups=f.row_groups_for_rank(r, 3)) for r in range(3)]
    f.close()
    assert_tables_equal(DeviceTable.concat(parts).to_arrow(), t, ordered=True)


@pytest.mark.parametrize("version", ["1.0", "2.0"])
@pytest.mark.parametrize("compression", ["none", "snappy", "zstd"])
def test_plain_string_pages_and_dictionary_fallback(tmp_path, compression, version):
    """BYTE_ARRAY columns outside a dictionary: written PLAIN (use_dictionary off), or dictionary-encoded until the writer's
    dictionary page limit and PLAIN from there (TPC-H's comment columns).  Such a chunk arrives as a Utf8 column (the host finds
    every value's offset and length, the device spreads them over the NULL rows, scans and copies); a column whose row groups came
    out in both kinds is Utf8 throughout (dfgpu_table_dictionary_decode on the dictionary-encoded ones); a low-cardinality column
    stays dictionary-encoded."""
    from datafusion_amd.parquet import ParquetFile, read_table
    rng = np.random.default_rng(5)
    n = 30_000
    words = ["final", "deposits", "slyly", "żółw", "日本", "", "carefully ironic requests", "x" * 90]
    comment = [" ".join(words[int(j)] for j in rng.integers(0, len(words), int(rng.integers(1, 5)))) + f" #{i}" for i in range(n)]
    low = [words[int(j)] for j in rng.integers(0, 4, n)]
    mixed = [words[int(j)] for j in rng.integers(0, 3, 8000)] + comment[8000:]
    t = pa.table({"k": pa.array(np.arange(n)), "comment": pa.array(comment, pa.string(), mask=rng.random(n) < 0.1), "low": pa.array(low, pa.string()),
                  "mixed": pa.array(mixed, pa.string(), mask=rng.random(n) < 0.05), "plain_low": pa.array(low, pa.string(), mask=rng.random(n) < 0.5)})
    path = str(tmp_path / "strings.parquet")
    pq.write_table(t, path, use_dictionary=["comment", "low", "mixed"], dictionary_pagesize_limit=4096, data_page_size=4096, row_group_size=8000,
                   compression=compression, data_page_version=version)
    f = ParquetFile(path)
    kinds = [f.inspect_chunk(g, "mixed") for g in range(f.num_row_groups)]
    f.close()
    assert kinds[0]["n_plain_pages"] == 0 and kinds[-1]["n_plain_pages"] > 0       # dictionary-only first, fallback later
    got = read_table(path).to_arrow()
    assert pa.types.is_dictionary(got.schema.field("low").type)
    for name in ("comment", "mixed", "plain_low"):
        assert pa.types.is_string(got.schema.field(name).type) or pa.types.is_large_string(got.schema.field(name).type), got.schema.field(name)
    assert_tables_equal(plain(got), pq.read_table(path), ordered=True)
    # the columns of interest alone, and a filter over the Utf8 column on the device
    from datafusion_amd import ops
    from datafusion_amd.expr import col
    dev = read_table(path, columns=["k", "comment"])
    hit = ops.filter(dev, col("comment").like("%slyly%")).to_arrow()
    want = [i for i, c in enumerate(t.column("comment").to_pylist()) if c is not None and "slyly" in c]
    assert hit.column("k").to_pylist() == want


def test_dictionary_decode_round_trip():
    from datafusion_amd.table import DeviceTable
    rng = np.random.default_rng(6)
    vals = ["a", "", "żółć", "longer string " * 3, "b"]
    s = pa.array([vals[int(j)] for j in rng.integers(0, len(vals), 5000)], pa.string(), mask=rng.random(5000) < 0.2)
    t = pa.table({"s": s.dictionary_encode(), "v": pa.array(np.arange(5000))})
    got = DeviceTable.from_arrow(t).dictionary_decode().to_arrow()
    assert pa.types.is_string(got.schema.field("s").type) or pa.types.is_large_string(got.schema.field("s").type)
    assert got.column("s").to_pylist() == s.to_pylist() and got.column("v").to_pylist() == list(range(5000))


def test_scan_in_one_call_thread_counts_cache_hits_and_a_corrupt_chunk(tmp_path):
    """dfgpu_parquet_read_chunks: the chunks of a scan in one call, decoded by host threads inside the library (two chunks in flight
    per worker, largest chunk first) — the same table for every thread count; chunks found in the device chunk cache are counted;
    an error on a worker thread (a chunk cut short) comes back as the call's error, not as a crash or a partial table"""
    import ctypes as C

    from datafusion_amd import _lib
    from datafusion_amd import parquet as P
    from datafusion_amd.parquet import ChunkCache, ParquetFile
    rng = np.random.default_rng(21)
    n = 300_000
    t = pa.table({"k": pa.array(rng.integers(0, 1 << 40, n)), "d": pa.array(rng.integers(0, 50, n).astype(np.int32)),
                  "f": pa.array(rng.random(n), mask=rng.random(n) < 0.2), "s": pa.array([f"v{i % 97}" for i in range(n)])})
    path = str(tmp_path / "t.parquet")
    pq.write_table(t, path, row_group_size=50_000, compression="zstd")
    want = pq.read_table(path)
    saved = P.CACHE
    try:
        P.CACHE = ChunkCache(budget=0)
        f = ParquetFile(path)
        for threads in (1, 3, 16):
            got = f.read(threads=threads).to_arrow()
            for c in want.column_names:
                col = got.column(c)
                assert (col.cast(pa.string()) if pa.types.is_dictionary(col.type) else col).equals(want.column(c)), (threads, c)
        P.CACHE = ChunkCache(budget=1 << 30)
        f.read(["k", "d"], threads=4).free()
        assert f.chunks_from_cache == 0
        f.read(["d", "f"], threads=4).free()
        assert f.chunks_from_cache == 6          # the six row groups' `d` chunks
        # one chunk cut short, on a worker thread
        arr = (_lib.ParquetChunk * 12)()
        keep = []
        for g in range(6):
            for j, c in enumerate(("k", "f")):
                buf, nb, d, alive = f._chunk(g, c)
                keep.append(alive)
                a = arr[g * 2 + j]
                a.bytes, a.n_bytes, a.column = buf, (nb // 2 if (g, c) == (4, "f") else nb), d
        out = C.c_void_p()
        rc = _lib.load().dfgpu_parquet_read_chunks(arr, 6, 2, 4, None, C.byref(out), None)
        assert rc != 0 and b"parquet" in _lib.load().dfgpu_last_error() and not out.value
        f.close()
    finally:
        P.CACHE.clear()
        P.CACHE = saved


@pytest.mark.parametrize("version", ["1.0", "2.0"])
@pytest.mark.parametrize("compression", ["none", "snappy", "snappy_on_the_device", "zstd"])
def test_pages_decoded_on_the_device(tmp_path, compression, version):
    """the device half of the scan (parquet.hip k_pq_decompress / k_pq_levels / k_pq_decode_pages, round 5): for fixed-width targets the host
    reads the page headers and — for ZSTD, and for Snappy by default — undoes the compression straight into the upload buffer; the definition
    levels' runs, the dictionary indices' runs, DELTA_BINARY_PACKED and BYTE_STREAM_SPLIT are decoded by kernels, and with
    parquet.snappy=device the file's bytes cross PCIe as they are and Snappy is decoded by a wave per page as well — 1 MiB pages with the
    shapes a Snappy stream takes: long literals (random data), short and OVERLAPPING copies (constant and periodic columns), copies
    reaching back further than the 32 KB ring the decoder keeps in LDS (a 40 KB period), NULLs in runs and scattered.  Bit for bit
    pyarrow's reading of the same file; the same table through the host path (parquet.device_decode = 0); the profile shows which kernels ran"""
    from datafusion_amd import ops
    from datafusion_amd.parquet import read_table
    rng = np.random.default_rng(5)
    n = 400_000
    period = rng.integers(0, 2**60, 5000)                                 # 40 KB of int64: a copy source behind the ring
    runs = np.repeat(rng.integers(0, 50, n // 1000 + 1), 1000)[:n]
    null_runs = np.repeat(rng.random(n // 700 + 1) < 0.3, 700)[:n]
    t = pa.table({
        "random64": pa.array(rng.integers(-2**62, 2**62, n)),
        "constant": pa.array(np.full(n, 123456789, dtype=np.int64)),
        "periodic": pa.array(period[np.arange(n) % 5000]),
        "short_period": pa.array((np.arange(n) % 7).astype(np.int32)),
        "runs_dict": pa.array(runs.astype(np.int32)),
        "date": pa.array((8000 + np.arange(n) // 97).astype(np.int32), pa.int32()).cast(pa.date32()),
        "f64": pa.array(np.round(rng.random(n) * 100, 2)),
        "null_runs": pa.array(rng.integers(0, 1000, n), mask=null_runs),
        "null_scattered": pa.array(rng.integers(0, 10**9, n).astype(np.int32), mask=rng.random(n) < 0.2),
        "delta": pa.array(np.cumsum(rng.integers(-5, 1000, n))),
        "bss": pa.array(rng.random(n)),
        "s": pa.array(np.array(["alpha", "beta", "gamma", "delta", None], dtype=object)[rng.integers(0, 5, n)], pa.string()),
    })
    path = str(tmp_path / "p.parquet")
    on_device = compression == "snappy_on_the_device"
    compression = "snappy" if on_device else compression
    if on_device:
        ops.set_options(parquet__snappy="device")
    pq.write_table(t, path, compression=compression, data_page_version=version, row_group_size=250_000, data_page_size=1 << 20,
                   use_dictionary=["runs_dict", "short_period", "s", "null_runs", "date"], column_encoding={"delta": "DELTA_BINARY_PACKED", "bss": "BYTE_STREAM_SPLIT"})
    exp = pq.read_table(path)
    ops.profile_enable(True)
    ops.profile_reset()
    got = read_table(path).to_arrow()
    stats = ops.profile_stats()
    ops.profile_enable(False)
    assert_tables_equal(plain(got), exp, ordered=True)
    assert "parquet_levels" in stats and "parquet_decode_pages" in stats and "parquet_decode" not in stats, sorted(stats)
    assert ("parquet_decompress_pages" in stats) == on_device, sorted(stats)
    ops.set_options(parquet__device_decode="0")
    from datafusion_amd import parquet as P
    P.CACHE.clear()   # (the second scan is to decode, not to be served from the device chunk cache)
    ops.profile_enable(True)
    ops.profile_reset()
    host = read_table(path).to_arrow()
    stats = ops.profile_stats()
    ops.profile_enable(False)
    assert_tables_equal(plain(host), exp, ordered=True)
    assert "parquet_decompress_pages" not in stats and "parquet_decode" in stats, sorted(stats)


@pytest.mark.parametrize("compression", ["none", "snappy"])
@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_device_page_decode_survives_corrupt_pages(tmp_path, compression, seed):
    """a column chunk with a stretch of its bytes overwritten — Snappy elements that copy from before the start of the output or end early,
    level lengths and run headers that overrun their page, indices beyond the dictionary: the device path either reports an error or
    decodes what the bytes now say (Parquet pages carry no checksum the readers verify); it never hangs and never writes out of bounds"""
    from datafusion_amd.parquet import ParquetFile
    rng = np.random.default_rng(seed)
    n = 60_000
    t = pa.table({"a": pa.array(np.repeat(rng.integers(0, 9, n // 50), 50).astype(np.int64), mask=np.repeat(rng.random(n // 50) < 0.2, 50)),
                  "b": pa.array(rng.integers(0, 1000, n))})
    path = str(tmp_path / "c.parquet")
    if compression == "snappy":
        from datafusion_amd import ops
        ops.set_options(parquet__snappy="device")   # (the Snappy stream is decoded by the kernel: its bounds checks are what is tested)
    pq.write_table(t, path, compression=compression, use_dictionary=["a"], data_page_size=64 * 1024, data_page_version="2.0" if seed % 2 else "1.0")
    raw = bytearray(open(path, "rb").read())
    for ci in range(2):
        md = pq.ParquetFile(path).metadata.row_group(0).column(ci)
        start = md.dictionary_page_offset if md.dictionary_page_offset else md.data_page_offset
        at = start + 30 + int(rng.integers(0, max(1, md.total_compressed_size - 120)))   # (behind the first page header)
        raw[at: at + 40] = bytes(rng.choice([0xFF, 0xFE, 0x7F, 0x01, 0xF3], 40).astype(np.uint8))
    open(path, "wb").write(bytes(raw))
    try:
        f = ParquetFile(path)
        got = f.read(["a", "b"]).to_arrow()
    except Exception as e:   # noqa: BLE001
        assert "parquet" in str(e) or "Parquet" in str(e), str(e)
    else:
        assert got.num_rows == n


def test_device_page_decode_of_a_chunk_with_hundreds_of_pages(tmp_path):
    """a chunk written with small pages (8 KB: ~400 data pages of 1000 values in one chunk): every page's first value is the sum of the
    earlier pages' non-null counts, which the decode workgroups add up among their threads — more pages than a workgroup has threads"""
    from datafusion_amd.parquet import read_table
    rng = np.random.default_rng(23)
    n = 400_000
    t = pa.table({"a": pa.array(rng.integers(-2**62, 2**62, n), mask=rng.random(n) < 0.1), "b": pa.array(rng.integers(0, 40, n).astype(np.int32))})
    path = str(tmp_path / "m.parquet")
    pq.write_table(t, path, compression="snappy", data_page_size=8 * 1024, row_group_size=n, use_dictionary=["b"])
    assert pq.ParquetFile(path).metadata.row_group(0).column(0).total_uncompressed_size > 300 * 8 * 1024
    assert_tables_equal(plain(read_table(path).to_arrow()), pq.read_table(path), ordered=True)


def test_scan_workers_upload_blocks_go_back_on_trim(tmp_path):
    """round 5 advice: the scan workers keep the device blocks their uploads land in per thread, between chunks and between calls.  They are
    pool blocks on loan: dfgpu_mem_trim (and dfgpu_shutdown) takes them back — after a scan whose tables are freed and a trim, the pool holds
    nothing — and a thread keeps at most parquet.upload_cache_bytes of them"""
    from datafusion_amd import _lib, ops
    from datafusion_amd import parquet as P
    from datafusion_amd.parquet import ChunkCache, ParquetFile
    rng = np.random.default_rng(4)
    n = 2_000_000
    t = pa.table({"a": pa.array(rng.integers(0, 1 << 40, n)), "b": pa.array(rng.integers(0, 1 << 40, n)), "c": pa.array(rng.random(n))})
    path = str(tmp_path / "t.parquet")
    pq.write_table(t, path, row_group_size=500_000, compression="none", use_dictionary=False)
    saved = P.CACHE
    try:
        P.CACHE = ChunkCache(budget=0)
        ops.sync()
        _lib.load().dfgpu_mem_trim()
        base = ops.mem_stats()["in_use"]
        f = ParquetFile(path)
        for _ in range(2):
            got = f.read(threads=4)
            assert got.num_rows == n
            got.free()
        f.close()
        ops.sync()
        held = ops.mem_stats()["in_use"] - base
        assert held > 0                                    # the workers' upload blocks (4 MB chunks -> 4 MiB blocks)
        _lib.load().dfgpu_mem_trim()
        after = ops.mem_stats()
        assert after["in_use"] == base and after["cached"] == 0, (base, held, after)
        # the byte cap: with a cap below one block nothing is kept at all
        ops.set_options(parquet__upload_cache_bytes="1024")
        f = ParquetFile(path)
        f.read(threads=4).free()
        f.close()
        ops.sync()
        assert ops.mem_stats()["in_use"] == base
    finally:
        ops.set_options(parquet__upload_cache_bytes=None)
        P.CACHE.clear()
        P.CACHE = saved
