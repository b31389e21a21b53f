"""Scan -> device for Arrow IPC files: the GPU-side twin of `DataSourceExec` over an `ArrowSource`
(datasource-arrow/src/source.rs:260-330: the file format `ArrowFileOpener` reads with arrow-ipc's `FileReader`, the stream format with
`StreamReader`).

An IPC file already holds Arrow buffers, so there is nothing to decode: each record batch is memory-mapped and handed to
`dfgpu_table_import` (one H2D copy per buffer, pinned on the fly when the range can be registered), with the scan's column projection
applied before the copy; batches are concatenated on the device.  Buffer compression (LZ4_FRAME / ZSTD per buffer) is undone on the host by
pyarrow's reader, as the reference's reader does.  Strings arrive as DFGPU_UTF8 columns or as dictionary-encoded columns, whichever the file
holds.  The device chunk cache of the Parquet scan (parquet.ChunkCache) is shared: a repeated scan of an unchanged file takes its batches
from HBM."""
from __future__ import annotations

import os

import pyarrow as pa
import pyarrow.ipc as ipc

from .table import DeviceTable


def _open(path: str):
    source = pa.memory_map(path, "r")
    try:
        return ipc.open_file(source), True
    except pa.ArrowInvalid:
        source.seek(0)
        return ipc.open_stream(source), False


def read_table(path: str, columns=None, stats: dict | None = None) -> DeviceTable:
    """the record batches of an Arrow IPC file (or stream), the given columns, as one device table"""
    from . import parquet as pq_scan
    st = os.stat(path)
    identity = (os.path.realpath(path), st.st_mtime_ns, st.st_size)
    reader, is_file = _open(path)
    names = list(columns) if columns is not None else reader.schema.names
    parts, hits = [], 0
    batches = (reader.get_batch(i) for i in range(reader.num_record_batches)) if is_file else iter(reader)
    for i, batch in enumerate(batches):
        key = identity + ("ipc", i, tuple(names))
        cached = pq_scan.CACHE.get(key)
        if cached is not None:
            hits += 1
            parts.append(cached)
            continue
        t = DeviceTable.from_arrow(pa.Table.from_batches([batch.select(names)]))
        pq_scan.CACHE.put_table(key, t)
        parts.append(t)
    if stats is not None:
        stats.update(record_batches=len(parts), record_batches_from_cache=hits)
    if not parts:
        return DeviceTable.from_arrow(pa.Table.from_batches([], schema=pa.schema([reader.schema.field(n) for n in names])))
    if len(parts) == 1:
        return parts[0]
    out = DeviceTable.concat(parts)
    for p in parts:
        p.free()
    return out
