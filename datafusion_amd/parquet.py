"""Scan -> device: Parquet column chunks decoded on the GPU.

Host side of the GPU scan node a DataFusion shim would put in place of `DataSourceExec` over a Parquet source
(datasource/src/source.rs:366, datasource-parquet): footer metadata -> for every projected column chunk of every row group
its byte range -> `dfgpu_parquet_decode_chunk` (include/dfgpu.h; page headers, decompression and run headers on the host,
value decoding on the device) -> row groups assembled with dfgpu_table_hstack / dfgpu_table_concat.  The footer is read
with pyarrow here (the shim has it from the `parquet` crate's `ParquetMetaData`); no value byte is decoded on the CPU.
Unsupported chunks (nested columns, DELTA / BYTE_STREAM_SPLIT encodings, strings outside a dictionary, other codecs) raise
DfgpuError: the rule keeps the CPU scan for such files.
"""
from __future__ import annotations

import ctypes as C
import mmap
import os
from concurrent.futures import ThreadPoolExecutor

import pyarrow as pa
import pyarrow.parquet as pq

from . import _lib
from ._lib import DfgpuError, ParquetChunkInfo, ParquetColumn, check
from .table import DeviceTable, field_of

PHYSICAL = {"BOOLEAN": 0, "INT32": 1, "INT64": 2, "INT96": 3, "FLOAT": 4, "DOUBLE": 5, "BYTE_ARRAY": 6, "FIXED_LEN_BYTE_ARRAY": 7}
CODEC = {"UNCOMPRESSED": 0, "SNAPPY": 1, "GZIP": 2, "LZO": 3, "BROTLI": 4, "LZ4": 5, "ZSTD": 6, "LZ4_RAW": 7}


def _target_field(arrow_type: pa.DataType):
    """device type of a column: strings become the Int32 indices of a dictionary-encoded column"""
    if pa.types.is_string(arrow_type) or pa.types.is_large_string(arrow_type) or \
            (pa.types.is_dictionary(arrow_type) and pa.types.is_string(arrow_type.value_type)):
        return field_of(pa.int32())
    return field_of(arrow_type)


class ChunkCache:
    """Device-resident column chunks of files that were scanned before: the HBM twin of the reference's keeping hot inputs in
    memory (MemorySourceConfig over cached RecordBatches, datasource/src/memory.rs:58; the file-metadata / statistics caches of
    execution/src/cache).  The cache itself lives BELOW the C ABI (dfgpu_cache_*: LRU under a byte budget, thread-safe, zero-copy
    views on a hit) so that the Rust shim's scan node uses the very same one; this class only spells the keys: (file identity = real
    path + mtime + size, row group, column).  A repeated scan — the same query again, or another query over the same lineitem
    columns — costs no host read, no decompression and no PCIe transfer.  Budget: DFGPU_TABLE_CACHE_BYTES (default 16 GiB, 0 = off;
    288 GB of HBM hold TPC-H SF100's hot columns); a file that changed on disk has a new identity and its old chunks age out."""

    def __init__(self, budget: int | None = None):
        import threading
        self.budget = int(os.environ.get("DFGPU_TABLE_CACHE_BYTES", 16 << 30)) if budget is None else budget
        self._h = None
        self._make = threading.Lock()      # the scan's decode threads arrive together: ONE cache is made

    def _handle(self):
        if self._h is None:
            with self._make:
                if self._h is None:
                    h = C.c_void_p()
                    check(_lib.load().dfgpu_cache_create(C.c_int64(self.budget), C.byref(h)))
                    self._h = h
        return self._h

    @staticmethod
    def _key(key) -> bytes:
        return repr(key).encode()

    def get(self, key):
        k = self._key(key)
        out = C.c_void_p()
        check(_lib.load().dfgpu_cache_get(self._handle(), k, C.c_int64(len(k)), C.byref(out)))
        return DeviceTable(out) if out.value else None

    def put(self, key, table: DeviceTable):
        k = self._key(key)
        check(_lib.load().dfgpu_cache_put(self._handle(), k, C.c_int64(len(k)), table.handle))

    put_table = put     # a whole multi-column table (one record batch of an IPC file) is kept the same way

    def clear(self):
        if self._h is not None:
            check(_lib.load().dfgpu_cache_clear(self._h))

    def stats(self) -> dict:
        st = _lib.CacheStats()
        check(_lib.load().dfgpu_cache_get_stats(self._handle(), C.byref(st)))
        return dict(chunks=st.entries, bytes=st.bytes, hits=st.hits, misses=st.misses, evictions=st.evictions, budget=st.budget_bytes)

    @property
    def bytes(self) -> int:
        return self.stats()["bytes"]

    @property
    def hits(self) -> int:
        return self.stats()["hits"]

    @property
    def misses(self) -> int:
        return self.stats()["misses"]

    def __del__(self):
        try:
            if self._h is not None:
                _lib.load().dfgpu_cache_free(self._h)
                self._h = None
        except Exception:
            pass


CACHE = ChunkCache()


class ParquetFile:
    """one Parquet file: footer via pyarrow, bytes via mmap"""

    def __init__(self, path: str):
        self.path = path
        self.pf = pq.ParquetFile(path)
        self.meta = self.pf.metadata
        self.arrow_schema = self.pf.schema_arrow
        self._f = open(path, "rb")
        self._mm = mmap.mmap(self._f.fileno(), 0, access=mmap.ACCESS_READ)
        import numpy as np
        self._view = np.frombuffer(self._mm, dtype=np.uint8)    # keeps the mapping's address (and the mapping) alive
        self._base = self._view.ctypes.data
        st = os.fstat(self._f.fileno())
        self._identity = (os.path.realpath(path), st.st_mtime_ns, st.st_size)

    def close(self):
        self._view = None
        try:
            self._mm.close()
        except BufferError:      # a view of the mapping is still referenced somewhere: the mapping goes with its last reference
            pass
        self._f.close()

    def _leaf(self, column: str) -> int:
        """Parquet leaf-column index of a flat top-level field, resolved by its path: with a nested (struct / list) column
        ahead of it the leaf indices shift away from the Arrow field indices.  Nested fields are refused (flat columns only)."""
        if column not in self.arrow_schema.names:
            raise KeyError(f"{self.path}: no column {column!r}")
        schema = self.pf.schema
        for j in range(len(schema)):
            if schema.column(j).path == column:
                return j
        raise DfgpuError(f"parquet: column {column!r} of {self.path} is not a flat leaf column (nested types are not supported on the GPU scan path)")

    @property
    def num_row_groups(self) -> int:
        return self.meta.num_row_groups

    @property
    def column_names(self):
        return self.arrow_schema.names

    def _chunk(self, row_group: int, column: str):
        """(pointer, byte length, ParquetColumn descriptor, keep-alive) of one column chunk"""
        j = self._leaf(column)
        cc = self.meta.row_group(row_group).column(j)
        sc = self.pf.schema.column(j)
        start = cc.data_page_offset
        if cc.has_dictionary_page and cc.dictionary_page_offset is not None and 0 < cc.dictionary_page_offset < start:
            start = cc.dictionary_page_offset
        nbytes = cc.total_compressed_size
        if start < 0 or start + nbytes > len(self._mm):
            raise DfgpuError(f"parquet: column chunk {column!r} of row group {row_group} lies outside {self.path}")
        buf = C.c_void_p(self._base + start)                    # the chunk's bytes where the page cache maps them: no host copy
        name = column.encode()
        d = ParquetColumn()
        d.physical_type = PHYSICAL[cc.physical_type]
        d.type_length = sc.length if sc.length is not None and sc.length > 0 else 0
        d.codec = CODEC[cc.compression]
        d.max_definition_level = sc.max_definition_level
        d.max_repetition_level = sc.max_repetition_level
        d.num_values = cc.num_values
        d.field = _target_field(self.arrow_schema.field(column).type)
        d.name = name
        return buf, nbytes, d, (name,)

    def inspect_chunk(self, row_group: int, column: str) -> dict:
        """the host half alone (no GPU): page / run / byte counts of one chunk"""
        buf, n, d, keep = self._chunk(row_group, column)
        info = ParquetChunkInfo()
        lib = _lib.load()
        rc = lib.dfgpu_parquet_inspect_chunk(buf, C.c_int64(n), C.byref(d), C.byref(info))
        if rc != 0 and d.physical_type == PHYSICAL["BYTE_ARRAY"] and b"read the column as Utf8" in lib.dfgpu_last_error():
            from .table import UTF8
            d.field.type = UTF8         # a string chunk with PLAIN pages: inspected the way it will be decoded
            rc = lib.dfgpu_parquet_inspect_chunk(buf, C.c_int64(n), C.byref(d), C.byref(info))
        check(rc)
        return {k: getattr(info, k) for k, _ in ParquetChunkInfo._fields_}

    def _decode(self, row_group: int, column: str) -> DeviceTable:
        """one column chunk on the device.  A string column is read as dictionary indices when every data page of the chunk is
        dictionary-encoded, and as Utf8 bytes when the writer fell back to PLAIN pages in it (high-cardinality text: TPC-H's
        comment columns) — `read` brings the chunks of a column to one kind."""
        key = self._identity + (row_group, column)
        hit = CACHE.get(key)
        if hit is not None:
            return hit
        buf, n, d, keep = self._chunk(row_group, column)
        h = C.c_void_p()
        lib = _lib.load()
        rc = lib.dfgpu_parquet_decode_chunk(buf, C.c_int64(n), C.byref(d), C.byref(h))
        if rc != 0 and d.physical_type == PHYSICAL["BYTE_ARRAY"] and b"read the column as Utf8" in lib.dfgpu_last_error():
            from .table import UTF8
            d.field.type = UTF8
            rc = lib.dfgpu_parquet_decode_chunk(buf, C.c_int64(n), C.byref(d), C.byref(h))
        check(rc)
        out = DeviceTable(h)
        CACHE.put(key, out)
        return out

    @staticmethod
    def _hstack(cols) -> DeviceTable:
        out = cols[0]
        for col in cols[1:]:
            both = C.c_void_p()
            check(_lib.load().dfgpu_table_hstack(out.handle, col.handle, C.byref(both)))
            out.free()
            col.free()
            out = DeviceTable(both)
        return out

    def read_row_group(self, row_group: int, columns=None) -> DeviceTable:
        _lib.init()
        return self._hstack([self._decode(row_group, name) for name in (columns or self.column_names)])

    def row_groups_overlapping(self, bounds: dict, in_lists: dict | None = None) -> list:
        """row groups whose footer statistics admit a value inside every (column -> closed [lo, hi]) bound — the pruning the
        reference's ParquetSource does with a pushed-down predicate (row-group statistics; here the dynamic bounds a hash join
        publishes from its build side, hash_join/shared_bounds.rs:277-284) — and, for a small build side, at least one value of
        the join's `IN (...)` list (column -> ascending values; PushdownStrategy::InList, shared_bounds.rs:275-284: a row group
        between two build keys is skipped although it lies inside their bounds).  An empty range (lo > hi) or an empty list
        prunes everything; a chunk without min / max statistics is kept."""
        import bisect
        keep = []
        for g in range(self.num_row_groups):
            rg = self.meta.row_group(g)
            ok = True
            for name, (lo, hi) in bounds.items():
                if lo > hi:
                    ok = False
                    break
                st = rg.column(self._leaf(name)).statistics
                if st is None or not st.has_min_max:
                    continue
                if st.max < lo or st.min > hi:
                    ok = False
                    break
            for name, values in (in_lists or {}).items():
                if not ok:
                    break
                if not values:
                    ok = False
                    break
                st = rg.column(self._leaf(name)).statistics
                if st is None or not st.has_min_max or not isinstance(st.min, int):
                    continue
                k = bisect.bisect_left(values, st.min)          # first list value >= the chunk's minimum
                if k == len(values) or values[k] > st.max:
                    ok = False
            if ok:
                keep.append(g)
        return keep

    def row_groups_for_rank(self, rank: int, world: int) -> list:
        """this rank's share of the file when `world` GPUs scan it (one process per GPU): contiguous row groups, split where the
        running row count crosses k/world of the total — the file-range split the reference's FileGroups make per partition
        (DataSourceExec: file_groups), at row-group granularity.  Disjoint, covering, in file order; no collective needed."""
        counts = [self.meta.row_group(g).num_rows for g in range(self.num_row_groups)]
        total = sum(counts)
        out, seen = [], 0
        for g, c in enumerate(counts):
            mid = seen + c / 2.0                       # a row group belongs to the rank its midpoint falls in
            owner = min(world - 1, int(mid * world / total)) if total else 0
            if owner == rank:
                out.append(g)
            seen += c
        return out

    def read(self, columns=None, threads: int | None = None, row_groups=None) -> DeviceTable:
        """all row groups (or the given ones), through dfgpu_parquet_read_chunks: the host half of a chunk (decompression above
        all) runs on one core, so the library decodes the chunks from `threads` host threads of its own (default min(16, cores),
        DFGPU_SCAN_THREADS overrides), each on a stream of its own — the way the reference's scan decodes row groups on its
        partition threads."""
        _lib.init()
        names = list(columns or self.column_names)
        groups = list(range(self.num_row_groups)) if row_groups is None else list(row_groups)
        if not groups:   # no row group (left): the schema alone
            sch = pa.schema([self.arrow_schema.field(c) for c in names])
            empty = [pa.array([], pa.dictionary(pa.int32(), pa.string()) if pa.types.is_string(f.type) else f.type) for f in sch]
            return DeviceTable.from_arrow(pa.Table.from_arrays(empty, names=sch.names))
        if threads is None:
            threads = int(os.environ.get("DFGPU_SCAN_THREADS", min(16, os.cpu_count() or 1)))
        # ONE call: the library's own host threads take the chunks off the list (dfgpu_parquet_read_chunks), look them up in the
        # chunk cache, decode the others and put the row groups together on the device
        work = [(g, c) for g in groups for c in names]
        arr = (_lib.ParquetChunk * len(work))()
        keep = []
        for k, (g, c) in enumerate(work):
            buf, n, d, names_alive = self._chunk(g, c)
            key = CACHE._key(self._identity + (g, c))
            keep.append((names_alive, key))
            arr[k].bytes, arr[k].n_bytes, arr[k].column = buf, n, d
            arr[k].cache_key, arr[k].cache_key_bytes = key, len(key)
        out, hits = C.c_void_p(), C.c_int64()
        check(_lib.load().dfgpu_parquet_read_chunks(arr, len(groups), len(names), threads, CACHE._handle() if CACHE.budget > 0 else None, C.byref(out),
                                                    C.byref(hits)))
        self.chunks_from_cache = hits.value
        return DeviceTable(out)


def read_table(path: str, columns=None, threads: int | None = None, bounds: dict | None = None, stats: dict | None = None, in_lists: dict | None = None,
               membership: dict | None = None) -> DeviceTable:
    """the row groups of `path` that can hold rows inside `bounds` and one of `in_lists`' values (all of them without either), the
    given columns, as one device table; `stats` receives row_groups_total / row_groups_read.

    `membership` = {key column: a join's built table (ops.JoinHashTable)}: the Map strategy of the join's dynamic filter
    (PushdownStrategy::Map, hash_join/shared_bounds.rs:275-284 — the reference pushes `HashTableLookupExpr`, partitioned_hash_eval.rs:278,
    to the probe-side scan, where it runs as a row filter).  After the statistics have pruned what they can, every surviving row group's
    KEY chunk is decoded first and asked `contains`; a row group without a single key of the build side is dropped before its other
    column chunks are read, decompressed or sent over PCIe, and the rows of the others are filtered on the device.  `stats` additionally
    receives row_groups_skipped_by_membership / rows_scanned / rows_passed."""
    f = ParquetFile(path)
    try:
        groups = None if not (bounds or in_lists) else f.row_groups_overlapping(bounds or {}, in_lists)
        if stats is not None:
            stats.update(row_groups_total=f.num_row_groups, row_groups_read=f.num_row_groups if groups is None else len(groups))
        if not membership:
            return f.read(columns, threads, groups)
        return _read_with_membership(f, columns, threads, groups, membership, stats)
    finally:
        f.close()


def _read_with_membership(f: "ParquetFile", columns, threads, groups, membership: dict, stats) -> DeviceTable:
    from . import ops
    from .expr import col
    names = list(columns or f.column_names)
    (key, table), = membership.items()           # one pushed-down join per scan (a second one filters what the first lets through)
    if key not in names:
        return f.read(names, threads, groups)
    groups = list(range(f.num_row_groups)) if groups is None else list(groups)
    parts, skipped, scanned, passed = [], 0, 0, 0
    for g in groups:
        keys = f._decode(g, key)
        scanned += keys.num_rows
        mask = table.contains(keys, [key])
        hit = ops.filter(mask, col("contains"), [])      # the count of passing rows without moving a column
        n_hit = hit.num_rows
        hit.free()
        if n_hit == 0:
            skipped += 1
            keys.free()
            mask.free()
            continue
        keys.free()                                       # cached: read() below takes the chunk from the device cache
        rg = f.read(names, threads, [g])
        if n_hit < rg.num_rows:
            both = ParquetFile._hstack([rg.select(list(range(rg.num_columns))), mask])
            rg.free()
            rg = ops.filter(both, col("contains"), names)
            both.free()
        else:
            mask.free()
        passed += rg.num_rows
        parts.append(rg)
    if stats is not None:
        stats.update(row_groups_skipped_by_membership=skipped, rows_scanned=scanned, rows_passed=passed)
    if not parts:
        return f.read(names, threads, [])
    if len(parts) == 1:
        return parts[0]
    out = DeviceTable.concat(parts)
    for p in parts:
        p.free()
    return out
