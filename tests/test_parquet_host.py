"""Scan -> device, the host half (no GPU): dfgpu_parquet_inspect_chunk parses page headers (Thrift compact protocol),
decompresses pages (UNCOMPRESSED / SNAPPY / ZSTD), reads definition levels and walks the run headers of the dictionary
index streams — checked against the footer pyarrow wrote (row counts, NULL counts, dictionary sizes, byte totals)."""
import pyarrow as pa
import pytest

from tests.parquet_cases import WRITER_MATRIX, case_id, sample_table, write


@pytest.mark.parametrize("writer", WRITER_MATRIX, ids=[case_id(w) for w in WRITER_MATRIX])
def test_inspect_matches_the_footer(tmp_path, writer):
    from datafusion_amd.parquet import ParquetFile
    t = sample_table(25_000)
    path = write(t, tmp_path, "t.parquet", data_page_size=16 * 1024, row_group_size=15_000, **writer)
    f = ParquetFile(path)
    assert f.num_row_groups == 2
    for g in range(f.num_row_groups):
        for j, name in enumerate(t.column_names):
            cc = f.meta.row_group(g).column(j)
            info = f.inspect_chunk(g, name)
            assert info["values"] == cc.num_values
            assert info["nulls"] == t.column(name).slice(g * 15_000, cc.num_values).null_count
            assert info["n_dictionary_pages"] == (1 if cc.has_dictionary_page else 0)
            assert info["n_data_pages_v1"] + info["n_data_pages_v2"] == info["n_plain_pages"] + info["n_dictionary_encoded_pages"] >= 1
            assert (info["n_data_pages_v2"] > 0) == (writer["data_page_version"] == "2.0")
            assert info["compressed_bytes"] <= cc.total_compressed_size and info["uncompressed_bytes"] <= cc.total_uncompressed_size   # footer totals include the page headers
            if writer["compression"] == "none":
                assert info["compressed_bytes"] == info["uncompressed_bytes"]
            if cc.has_dictionary_page and info["n_plain_pages"] == 0:
                want = t.column(name).slice(g * 15_000, cc.num_values).drop_null()
                assert info["dictionary_values"] == len(want.unique())
    assert f.inspect_chunk(0, "runs")["n_runs_rle"] >= 100 if writer["use_dictionary"] else True
    f.close()


def test_unsupported_chunks_are_errors_not_wrong_answers(tmp_path):
    from datafusion_amd import _lib
    from datafusion_amd.parquet import ParquetFile
    import pyarrow.parquet as pq
    t = pa.table({"s": pa.array(["a", "b", "c"] * 10), "b": pa.array([True, False, True] * 10), "i": pa.array(list(range(30)), pa.int64()),
                  "l": pa.array([[1, 2], [3], []] * 10, pa.list_(pa.int64()))})
    path = str(tmp_path / "u.parquet")
    pq.write_table(t, path, use_dictionary=False, compression="gzip", column_encoding=None)
    f = ParquetFile(path)
    with pytest.raises(_lib.DfgpuError, match="codec"):
        f.inspect_chunk(0, "i")
    f.close()
    pq.write_table(t.select(["s", "b", "i"]), path, use_dictionary=False, compression="none")
    f = ParquetFile(path)
    assert f.inspect_chunk(0, "s")["n_plain_pages"] == 1      # PLAIN strings are read as Utf8 since round 2 (the dictionary-index reading of the same chunk is refused)
    import ctypes as C
    from datafusion_amd._lib import ParquetChunkInfo
    buf, nb, d, keep = f._chunk(0, "s")
    assert _lib.load().dfgpu_parquet_inspect_chunk(buf, C.c_int64(nb), C.byref(d), C.byref(ParquetChunkInfo())) != 0
    assert b"PLAIN-encoded BYTE_ARRAY" in _lib.load().dfgpu_last_error()
    assert f.inspect_chunk(0, "b")["values"] == 30             # BOOLEAN pages are read since round 3 (PLAIN bits / RLE runs)
    f.close()
    pq.write_table(t.select(["i"]), path, use_dictionary=False, compression="none", column_encoding={"i": "DELTA_BINARY_PACKED"})
    f = ParquetFile(path)
    assert f.inspect_chunk(0, "i")["n_plain_pages"] == 1       # DELTA_BINARY_PACKED is decoded on the host and staged as PLAIN values (round 3)
    f.close()
    pq.write_table(pa.table({"s": pa.array(["a", "bb"] * 15)}), path, use_dictionary=False, compression="none", column_encoding={"s": "DELTA_BYTE_ARRAY"})
    f = ParquetFile(path)
    with pytest.raises(_lib.DfgpuError, match="value encoding 7"):
        f.inspect_chunk(0, "s")
    f.close()


def test_truncated_chunk_is_an_error(tmp_path):
    import ctypes as C

    from datafusion_amd import _lib
    from datafusion_amd._lib import ParquetChunkInfo
    from datafusion_amd.parquet import ParquetFile
    path = write(sample_table(5000, nulls=False), tmp_path, "t.parquet", compression="snappy")
    f = ParquetFile(path)
    buf, n, d, keep = f._chunk(0, "d")
    info = ParquetChunkInfo()
    for cut in (n // 2, 10, 0):
        assert _lib.load().dfgpu_parquet_inspect_chunk(buf, C.c_int64(cut), C.byref(d), C.byref(info)) != 0
        assert b"parquet" in _lib.load().dfgpu_last_error()
    f.close()


def test_row_group_pruning_by_key_bounds(tmp_path):
    """ParquetFile.row_groups_overlapping: footer min / max against closed key bounds (what a hash join's dynamic filter prunes with)"""
    import numpy as np
    import pyarrow.parquet as pq

    from datafusion_amd.parquet import ParquetFile
    n = 10_000
    t = pa.table({"k": pa.array(np.arange(n, dtype=np.int64) * 3), "v": pa.array(np.arange(n, dtype=np.int32) % 7)})
    path = str(tmp_path / "sorted.parquet")
    pq.write_table(t, path, row_group_size=1000)
    f = ParquetFile(path)
    assert f.num_row_groups == 10
    assert f.row_groups_overlapping({}) == list(range(10))
    assert f.row_groups_overlapping({"k": (0, 3 * n)}) == list(range(10))
    assert f.row_groups_overlapping({"k": (2998, 2999)}) == []                  # between row group 0 (max 2997) and row group 1 (min 3000)
    assert f.row_groups_overlapping({"k": (2997, 3000)}) == [0, 1]              # closed bounds touch both neighbours
    assert f.row_groups_overlapping({"k": (12_000, 12_001)}) == [4]
    assert f.row_groups_overlapping({"k": (-50, -1)}) == [] and f.row_groups_overlapping({"k": (10**9, 10**9 + 1)}) == []
    assert f.row_groups_overlapping({"k": (1, 0)}) == []                         # empty build side: empty range
    assert f.row_groups_overlapping({"k": (0, 3 * n), "v": (7, 9)}) == []        # v is 0..6 everywhere
    pq.write_table(t, path, row_group_size=1000, write_statistics=False)
    f2 = ParquetFile(path)
    assert f2.row_groups_overlapping({"k": (12_000, 12_001)}) == list(range(10))  # no statistics: nothing can be pruned
    f.close()
    f2.close()


def test_dictionary_fallback_inside_a_chunk(tmp_path):
    """a writer that gives up on the dictionary mid-chunk (dictionary page limit reached) leaves dictionary-encoded pages followed by
    PLAIN pages in ONE column chunk: the host half plans both kinds page by page"""
    import numpy as np
    import pyarrow.parquet as pq

    from datafusion_amd.parquet import ParquetFile
    n = 50_000
    t = pa.table({"k": pa.array(np.arange(n, dtype=np.int64) * 7)})
    path = str(tmp_path / "fallback.parquet")
    pq.write_table(t, path, dictionary_pagesize_limit=4096, data_page_size=8192, compression="snappy")
    f = ParquetFile(path)
    info = f.inspect_chunk(0, "k")
    assert info["n_dictionary_pages"] == 1 and info["n_dictionary_encoded_pages"] >= 1 and info["n_plain_pages"] >= 1, info
    assert info["values"] == n and info["nulls"] == 0
    assert info["dictionary_values"] < n
    f.close()


def test_corrupt_page_headers_are_errors(tmp_path):
    """bit flips in a chunk (page header fields, level lengths, run headers) end in an error message, never in a crash: every
    byte of the first 64 and a sample of the rest is inverted in turn"""
    import ctypes as C

    import numpy as np

    from datafusion_amd import _lib
    from datafusion_amd._lib import ParquetChunkInfo
    from datafusion_amd.parquet import ParquetFile
    path = write(sample_table(3000), tmp_path, "t.parquet", compression="none", data_page_size=2048)
    f = ParquetFile(path)
    lib = _lib.load()
    rng = np.random.default_rng(0)
    outcomes = {"ok": 0, "error": 0}
    for name in ("nul", "lowcard", "d", "s"):
        buf, n, d, keep = f._chunk(0, name)
        raw = C.string_at(buf, n)
        positions = list(range(min(64, n))) + [int(x) for x in rng.integers(0, n, 200)]
        for pos in positions:
            mutated = bytearray(raw)
            mutated[pos] ^= 0xFF
            mb = (C.c_uint8 * n).from_buffer_copy(bytes(mutated))
            info = ParquetChunkInfo()
            rc = lib.dfgpu_parquet_inspect_chunk(mb, C.c_int64(n), C.byref(d), C.byref(info))
            outcomes["ok" if rc == 0 else "error"] += 1
            if rc != 0:
                assert lib.dfgpu_last_error().startswith(b"parquet") or b"alloc" in lib.dfgpu_last_error().lower() or b"length" in lib.dfgpu_last_error().lower(), lib.dfgpu_last_error()
    assert outcomes["error"] > 50       # header corruption is detected; flips inside value bytes legitimately decode
    f.close()


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_row_groups_are_shared_out_to_ranks_without_gaps_or_overlap(tmp_path, world):
    import numpy as np
    import pyarrow.parquet as pq

    from datafusion_amd.parquet import ParquetFile
    n = 23_456
    path = str(tmp_path / "r.parquet")
    pq.write_table(pa.table({"k": pa.array(np.arange(n, dtype=np.int64))}), path, row_group_size=1000)
    f = ParquetFile(path)
    shares = [f.row_groups_for_rank(r, world) for r in range(world)]
    flat = [g for s in shares for g in s]
    assert flat == list(range(f.num_row_groups))                       # covering, disjoint, in file order, contiguous per rank
    rows = [sum(f.meta.row_group(g).num_rows for g in s) for s in shares]
    assert sum(rows) == n and max(rows) - min(rows) <= 1000            # balanced to one row group
    f.close()


def test_plain_string_chunks_host_half_and_corruption(tmp_path):
    """BYTE_ARRAY chunks with PLAIN pages (read as Utf8): the host half counts the pages, and flipped bytes in the length prefixes /
    page headers end in an error message or a clean parse, never in a crash"""
    import ctypes as C

    import numpy as np
    import pyarrow as pa
    import pyarrow.parquet as pq

    from datafusion_amd import _lib
    from datafusion_amd._lib import ParquetChunkInfo
    from datafusion_amd.parquet import ParquetFile
    rng = np.random.default_rng(1)
    n = 6000
    t = pa.table({"c": pa.array([f"comment {i} " + "x" * int(rng.integers(0, 30)) for i in range(n)], pa.string(), mask=rng.random(n) < 0.1)})
    path = str(tmp_path / "plain.parquet")
    pq.write_table(t, path, use_dictionary=False, data_page_size=2048, compression="none")
    f = ParquetFile(path)
    info = f.inspect_chunk(0, "c")
    assert info["n_plain_pages"] >= 2 and info["values"] == n and info["nulls"] == t.column("c").null_count
    from datafusion_amd.table import UTF8
    buf, nb, d, keep = f._chunk(0, "c")
    d.field.type = UTF8
    raw = C.string_at(buf, nb)
    lib = _lib.load()
    outcomes = {"ok": 0, "error": 0}
    for pos in list(range(min(96, nb))) + [int(x) for x in rng.integers(0, nb, 300)]:
        mutated = bytearray(raw)
        mutated[pos] ^= 0xFF
        mb = (C.c_uint8 * nb).from_buffer_copy(bytes(mutated))
        out = ParquetChunkInfo()
        rc = lib.dfgpu_parquet_inspect_chunk(mb, C.c_int64(nb), C.byref(d), C.byref(out))
        outcomes["ok" if rc == 0 else "error"] += 1
        if rc != 0:
            assert b"parquet" in lib.dfgpu_last_error()
    assert outcomes["error"] > 0
    f.close()
