"""Scan -> device for Arrow IPC files: the GPU-side twin of `DataSourceExec` over an `ArrowSource`
(datasource-arrow/src/source.rs:260-330: the file format `ArrowFileOpener` reads with arrow-ipc's `FileReader`, the stream format with
`StreamReader`).

The reader lives BELOW the C ABI (csrc/ipc.hip, dfgpu_ipc_*: a hand-written walk of the IPC messages' flatbuffers metadata), so the Rust
shim's scan node uses the very same one; this module memory-maps the file, spells the cache keys and concatenates the batches.  An IPC
file already holds Arrow buffers, so there is nothing to decode: every record batch's buffers are imported where the page cache maps them
(one H2D copy per buffer, pinned on the fly), the scan's column projection applied before the copy; per-buffer compression (ZSTD /
LZ4_FRAME) is undone on the host first, as the reference's reader does.  Strings arrive as DFGPU_UTF8 columns or as dictionary-encoded
columns, whichever the file holds.  The device chunk cache of the Parquet scan (parquet.CACHE, dfgpu_cache_*) is shared: a repeated
scan of an unchanged file takes its batches from HBM."""
from __future__ import annotations

import ctypes as C
import mmap
import os

import numpy as np
import pyarrow as pa

from . import _lib
from ._lib import check
from .table import DeviceTable

_FORMATS = {"c": pa.int8(), "C": pa.uint8(), "s": pa.int16(), "S": pa.uint16(), "i": pa.int32(), "I": pa.uint32(), "l": pa.int64(), "L": pa.uint64(),
            "g": pa.float64(), "b": pa.bool_(), "tdD": pa.date32(), "u": pa.string(), "U": pa.large_string(), "vu": pa.string_view()}


class IpcFile:
    """one Arrow IPC file or stream, memory-mapped; `dfgpu_ipc_open` has walked its messages (no GPU needed for that)"""

    def __init__(self, path: str):
        self.path = path
        self._f = open(path, "rb")
        st = os.fstat(self._f.fileno())
        self.identity = (os.path.realpath(path), st.st_mtime_ns, st.st_size)
        self._mm = mmap.mmap(self._f.fileno(), 0, access=mmap.ACCESS_READ)
        self._view = np.frombuffer(self._mm, dtype=np.uint8)
        self._h = C.c_void_p()
        check(_lib.load().dfgpu_ipc_open(C.c_void_p(self._view.ctypes.data), C.c_int64(st.st_size), C.byref(self._h)))
        nb, nc, ff = C.c_int64(), C.c_int32(), C.c_int32()
        check(_lib.load().dfgpu_ipc_info(self._h, C.byref(nb), C.byref(nc), C.byref(ff)))
        self.num_record_batches, self.num_columns, self.is_file_format = nb.value, nc.value, bool(ff.value)
        self.columns = []
        for i in range(self.num_columns):
            name, fmt, nullable, dic = C.c_char_p(), C.c_char_p(), C.c_int32(), C.c_int32()
            check(_lib.load().dfgpu_ipc_column(self._h, i, C.byref(name), C.byref(fmt), C.byref(nullable), C.byref(dic)))
            self.columns.append((name.value.decode(), fmt.value.decode(), bool(nullable.value), bool(dic.value)))

    @property
    def column_names(self):
        return [c[0] for c in self.columns]

    @property
    def schema(self) -> pa.Schema:
        def typ(fmt, dic):
            t = _FORMATS.get(fmt) or pa.decimal128(*[int(x) for x in fmt[2:].split(",")[:2]])
            return pa.dictionary(pa.int32(), t) if dic else t
        return pa.schema([pa.field(n, typ(f, d), nullable) for n, f, nullable, d in self.columns])

    def batch_rows(self, i: int) -> int:
        n = C.c_int64()
        check(_lib.load().dfgpu_ipc_batch_rows(self._h, C.c_int64(i), C.byref(n)))
        return n.value

    def read_batch(self, i: int, columns=None) -> DeviceTable:
        _lib.init()
        idx = None if columns is None else [self.column_names.index(c) if isinstance(c, str) else int(c) for c in columns]
        out = C.c_void_p()
        arr = None if idx is None else (C.c_int * max(1, len(idx)))(*idx)
        check(_lib.load().dfgpu_ipc_read_batch(self._h, C.c_int64(i), arr, 0 if idx is None else len(idx), C.byref(out)))
        return DeviceTable(out)

    def close(self):
        if self._h:
            _lib.load().dfgpu_ipc_close(self._h)
            self._h = None
        self._view = None
        try:
            self._mm.close()
        except BufferError:
            pass
        self._f.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def read_table(path: str, columns=None, stats: dict | None = None) -> DeviceTable:
    """the record batches of an Arrow IPC file (or stream), the given columns, as one device table"""
    from . import parquet as pq_scan
    f = IpcFile(path)
    try:
        names = list(columns) if columns is not None else f.column_names
        parts, hits = [], 0
        for i in range(f.num_record_batches):
            key = f.identity + ("ipc", i, tuple(names))
            cached = pq_scan.CACHE.get(key)
            if cached is not None:
                hits += 1
                parts.append(cached)
                continue
            t = f.read_batch(i, names)
            pq_scan.CACHE.put_table(key, t)
            parts.append(t)
        if stats is not None:
            stats.update(record_batches=len(parts), record_batches_from_cache=hits)
        if not parts:
            sch = f.schema
            return DeviceTable.from_arrow(pa.Table.from_batches([], schema=pa.schema([sch.field(n) for n in names])))
        if len(parts) == 1:
            return parts[0]
        out = DeviceTable.concat(parts)
        for p in parts:
            p.free()
        return out
    finally:
        f.close()
