"""DeviceTable: a RecordBatch whose column buffers live in HBM (handle into libdfgpu.so).

Crossing the boundary uses the Arrow C Data Interface exactly as the reference's own FFI
streams do (datafusion/ffi/src/record_batch_stream.rs:105-114,156-172 wrap a StructArray in
FFI_ArrowArray + FFI_ArrowSchema).
"""
from __future__ import annotations

import ctypes as C

import pyarrow as pa

from . import _lib
from ._lib import ColumnView, Field, check

# dfgpu_type
INT32, INT64, DECIMAL128, FLOAT64, UINT8, UINT32, UINT64, DATE32, BOOL, UTF8 = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10


class ArrowSchema(C.Structure):
    pass


class ArrowArray(C.Structure):
    pass


ArrowSchema._fields_ = [("format", C.c_char_p), ("name", C.c_char_p), ("metadata", C.c_char_p), ("flags", C.c_int64),
                        ("n_children", C.c_int64), ("children", C.POINTER(C.POINTER(ArrowSchema))),
                        ("dictionary", C.POINTER(ArrowSchema)), ("release", C.c_void_p), ("private_data", C.c_void_p)]
ArrowArray._fields_ = [("length", C.c_int64), ("null_count", C.c_int64), ("offset", C.c_int64), ("n_buffers", C.c_int64),
                       ("n_children", C.c_int64), ("buffers", C.POINTER(C.c_void_p)),
                       ("children", C.POINTER(C.POINTER(ArrowArray))), ("dictionary", C.POINTER(ArrowArray)),
                       ("release", C.c_void_p), ("private_data", C.c_void_p)]


class ArrowDeviceArray(C.Structure):
    """Arrow C Device Data Interface: an ArrowArray whose buffers are device pointers (ARROW_DEVICE_ROCM = 10)"""
    _fields_ = [("array", ArrowArray), ("device_id", C.c_int64), ("device_type", C.c_int32), ("sync_event", C.c_void_p), ("reserved", C.c_int64 * 3)]


ARROW_DEVICE_ROCM = 10


def field_of(t: pa.DataType) -> Field:
    if pa.types.is_int32(t):
        return Field(INT32, 0, 0, 1)
    if pa.types.is_int64(t):
        return Field(INT64, 0, 0, 1)
    if pa.types.is_decimal128(t):
        return Field(DECIMAL128, t.precision, t.scale, 1)
    if pa.types.is_float64(t):
        return Field(FLOAT64, 0, 0, 1)
    if pa.types.is_uint8(t):
        return Field(UINT8, 0, 0, 1)
    if pa.types.is_uint32(t):
        return Field(UINT32, 0, 0, 1)
    if pa.types.is_uint64(t):
        return Field(UINT64, 0, 0, 1)
    if pa.types.is_date32(t):
        return Field(DATE32, 0, 0, 1)
    if pa.types.is_boolean(t):
        return Field(BOOL, 0, 0, 1)
    if pa.types.is_string(t) or pa.types.is_large_string(t) or t == pa.string_view():
        return Field(UTF8, 0, 0, 1)
    raise TypeError(f"type {t} is not supported on the GPU path")


def arrow_type_of(f: Field) -> pa.DataType:
    return {INT32: pa.int32(), INT64: pa.int64(), FLOAT64: pa.float64(), UINT8: pa.uint8(), UINT32: pa.uint32(),
            UINT64: pa.uint64(), DATE32: pa.date32(), BOOL: pa.bool_(), UTF8: pa.string()}.get(f.type) or pa.decimal128(f.precision, f.scale)


class DeviceTable:
    """owning wrapper of a dfgpu_table_t"""

    def __init__(self, handle):
        self._h = C.c_void_p(handle) if not isinstance(handle, C.c_void_p) else handle

    # ------------------------------------------------------------------ construction
    @staticmethod
    def from_arrow(data) -> "DeviceTable":
        lib = _lib.init()
        if isinstance(data, pa.Table):
            batches = data.combine_chunks().to_batches()
            if len(batches) == 0:
                batch = pa.RecordBatch.from_arrays([pa.array([], type=f.type) for f in data.schema], schema=data.schema)
            elif len(batches) == 1:
                batch = batches[0]
            else:
                batch = pa.Table.from_batches(batches).combine_chunks().to_batches()[0]
        else:
            batch = data
        arr, sch = ArrowArray(), ArrowSchema()
        batch._export_to_c(C.addressof(arr), C.addressof(sch))
        out = C.c_void_p()
        check(lib.dfgpu_table_import(C.byref(arr), C.byref(sch), C.byref(out)))
        return DeviceTable(out)

    def to_arrow(self) -> pa.Table:
        lib = _lib.load()
        arr, sch = ArrowArray(), ArrowSchema()
        check(lib.dfgpu_table_export(self._h, C.byref(arr), C.byref(sch)))
        batch = pa.RecordBatch._import_from_c(C.addressof(arr), C.addressof(sch))
        return pa.Table.from_batches([batch])

    def to_batches(self, batch_rows: int = 8192):
        """the table as a stream of RecordBatches of `batch_rows` rows (the last one may be shorter): what the shim's
        poll_next yields (dfgpu_table_export_batch; LimitedBatchCoalescer's target batch size, coalesce/mod.rs:27-120)"""
        lib = _lib.load()
        n = self.num_rows
        for off in range(0, max(n, 1), batch_rows):
            arr, sch = ArrowArray(), ArrowSchema()
            check(lib.dfgpu_table_export_batch(self._h, C.c_int64(off), C.c_int64(min(batch_rows, n - off)), C.byref(arr), C.byref(sch)))
            yield pa.RecordBatch._import_from_c(C.addressof(arr), C.addressof(sch))

    # ------------------------------------------------------------------ device-resident hand-off
    def retain(self) -> "DeviceTable":
        """a second owner of the same HBM buffers (dfgpu_table_retain)"""
        out = C.c_void_p()
        check(_lib.load().dfgpu_table_retain(self.handle, C.byref(out)))
        return DeviceTable(out)

    def export_device(self):
        """(ArrowDeviceArray, ArrowSchema): the table as an Arrow C Device array over its own HBM buffers — what a GPU node hands the
        next GPU node (or any other ROCm component) instead of a host RecordBatch.  The array keeps the buffers alive until its release
        callback runs (DeviceTable.from_device consumes it)."""
        arr, sch = ArrowDeviceArray(), ArrowSchema()
        check(_lib.load().dfgpu_table_export_device(self.handle, C.byref(arr), C.byref(sch)))
        return arr, sch

    @staticmethod
    def from_device(arr: ArrowDeviceArray, sch: ArrowSchema) -> "DeviceTable":
        """dfgpu_table_import_device: zero-copy; consumes both structs"""
        out = C.c_void_p()
        check(_lib.init().dfgpu_table_import_device(C.byref(arr), C.byref(sch), C.byref(out)))
        return DeviceTable(out)

    # ------------------------------------------------------------------ inspection
    @property
    def handle(self) -> C.c_void_p:
        if self._h is None:
            raise _lib.DfgpuError("table already freed")
        return self._h

    @property
    def num_rows(self) -> int:
        n = C.c_int64()
        check(_lib.load().dfgpu_table_num_rows(self.handle, C.byref(n)))
        return n.value

    @property
    def num_columns(self) -> int:
        n = C.c_int()
        check(_lib.load().dfgpu_table_num_columns(self.handle, C.byref(n)))
        return n.value

    def column_view(self, i: int) -> ColumnView:
        v = ColumnView()
        check(_lib.load().dfgpu_table_column(self.handle, i, C.byref(v)))
        return v

    @property
    def column_names(self):
        # a table is immutable: its names are read through the C ABI once (an operator call looks several columns up by name)
        names = self.__dict__.get("_names")
        if names is None:
            names = self.__dict__["_names"] = tuple(self.column_view(i).name.decode() for i in range(self.num_columns))
        return list(names)

    @property
    def schema(self) -> pa.Schema:
        views = [self.column_view(i) for i in range(self.num_columns)]
        return pa.schema([pa.field(v.name.decode(), arrow_type_of(v.field)) for v in views])

    def index_of(self, name_or_index) -> int:
        if isinstance(name_or_index, int):
            return name_or_index
        index = self.__dict__.get("_name_index")
        if index is None:
            names = self.column_names
            index = self.__dict__["_name_index"] = {n: (i if names.count(n) == 1 else -1) for i, n in enumerate(names)}
        i = index.get(name_or_index, -1)
        if i < 0:
            raise KeyError(f"column {name_or_index!r} not found or ambiguous in {self.column_names}")
        return i

    def dictionary_code(self, column, value: str):
        """index of `value` in a dictionary-encoded string column's dictionary, None if absent — what
        `col = 'value'` is lowered to (dfgpu_table_dictionary_lookup)"""
        b = value.encode()
        code = C.c_int64()
        check(_lib.load().dfgpu_table_dictionary_lookup(self.handle, self.index_of(column), b, C.c_int64(len(b)), C.byref(code)))
        return None if code.value < 0 else code.value

    def dictionary_encode(self, columns=None, sorted: bool = True) -> "DeviceTable":
        """the table with its Utf8 columns (all of them, or the named ones) dictionary-encoded on the device
        (dfgpu_table_dictionary_encode): Int32 indices in HBM, the dictionary on the host in ascending order (`sorted`) or in
        first-seen order — the form joins, GROUP BY, ORDER BY and repartition take string keys in"""
        lib = _lib.load()
        names = self.column_names
        todo = [i for i in range(self.num_columns) if self.column_view(i).field.type == UTF8] if columns is None else [self.index_of(c) for c in columns]
        cur, owned = self, False
        for i in todo:
            out = C.c_void_p()
            check(lib.dfgpu_table_dictionary_encode(cur.handle, i, int(sorted), C.byref(out)))
            nxt = DeviceTable(out)
            if owned:
                cur.free()
            cur, owned = nxt, True
        return cur if owned else self.select(names)

    def dictionary_size(self, column):
        """number of values in the dictionary of a dictionary-encoded column, None for any other column"""
        n = C.c_int64(0)
        check(_lib.load().dfgpu_table_dictionary_size(self.handle, self.index_of(column), C.byref(n)))
        return None if n.value < 0 else n.value

    def dictionary_decode(self, columns=None) -> "DeviceTable":
        """the table with its dictionary-encoded string columns (all of them, or the named ones) as Utf8 columns in HBM
        (dfgpu_table_dictionary_decode)"""
        lib = _lib.load()
        names = self.column_names
        todo = [i for i in range(self.num_columns) if self.dictionary_size(i) is not None] if columns is None else [self.index_of(c) for c in columns]
        cur, owned = self, False
        for i in todo:
            out = C.c_void_p()
            check(lib.dfgpu_table_dictionary_decode(cur.handle, i, C.byref(out)))
            nxt = DeviceTable(out)
            if owned:
                cur.free()
            cur, owned = nxt, True
        return cur if owned else self.select(names)

    def dictionary_like(self, column, pattern: str, case_insensitive: bool = False):
        """ascending indices of the dictionary values of a dictionary-encoded string column that match the SQL LIKE
        `pattern` (dfgpu_table_dictionary_like) — what `col LIKE 'pattern'` is lowered to (expr.LikeExpr)"""
        lib, b, n = _lib.load(), pattern.encode(), C.c_int64(0)
        check(lib.dfgpu_table_dictionary_like(self.handle, self.index_of(column), b, C.c_int64(len(b)), int(case_insensitive), None, C.c_int64(0), C.byref(n)))
        codes = (C.c_int64 * max(1, n.value))()
        check(lib.dfgpu_table_dictionary_like(self.handle, self.index_of(column), b, C.c_int64(len(b)), int(case_insensitive), codes, C.c_int64(n.value), C.byref(n)))
        return list(codes[:n.value])

    def nbytes(self) -> int:
        total = 0
        for i in range(self.num_columns):
            v = self.column_view(i)
            w = {INT32: 4, INT64: 8, DECIMAL128: 16, FLOAT64: 8, UINT8: 1, UINT32: 4, UINT64: 8, DATE32: 4}.get(v.field.type)
            if v.field.type == UTF8:
                total += v.length * 24            # 64-bit offsets + an estimate of 16 bytes per string (the exact byte count lives on the device)
            else:
                total += (v.length + 7) // 8 if w is None else v.length * w
        return total

    # ------------------------------------------------------------------ zero-copy ops
    def select(self, cols) -> "DeviceTable":
        idx = [self.index_of(c) for c in cols]
        arr = (C.c_int * len(idx))(*idx)
        out = C.c_void_p()
        check(_lib.load().dfgpu_table_select(self.handle, arr, len(idx), C.byref(out)))
        return DeviceTable(out)

    def slice(self, offset: int, length: int) -> "DeviceTable":
        out = C.c_void_p()
        check(_lib.load().dfgpu_table_slice(self.handle, C.c_int64(offset), C.c_int64(length), C.byref(out)))
        return DeviceTable(out)

    @staticmethod
    def concat(parts) -> "DeviceTable":
        arr = (C.c_void_p * len(parts))(*[p.handle for p in parts])
        out = C.c_void_p()
        check(_lib.load().dfgpu_table_concat(arr, len(parts), C.byref(out)))
        return DeviceTable(out)

    def free(self):
        if self._h is not None:
            _lib.load().dfgpu_table_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
