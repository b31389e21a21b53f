"""ctypes front-end of the CPU oracle (oracle/dforacle.c).

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product (datafusion_amd / libdfgpu.so)
never does.  Tables are pyarrow Tables; every operator returns a pyarrow Table so the
parity tests read like the reference's `assert_batches_sorted_eq!` tests.

Expression trees for the oracle are plain tuples (independent of the product's IR):
    ("col", name) | ("lit", python_value, pa.DataType) | ("cast", expr, pa.DataType)
    ("bin", op, lhs, rhs)   op in + - * = != < <= > >= and or
    ("is_null", expr) | ("not", expr)
    ("case", when_expr, then_expr, else_expr | None)
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from decimal import Decimal

import numpy as np
import pyarrow as pa

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

ORC_I32, ORC_I64, ORC_I128, ORC_F64, ORC_U8, ORC_U32, ORC_U64 = 1, 2, 3, 4, 5, 6, 7
JOIN_TYPES = {
    "Inner": 0, "Left": 1, "Right": 2, "Full": 3, "LeftSemi": 4, "RightSemi": 5,
    "LeftAnti": 6, "RightAnti": 7, "LeftMark": 8, "RightMark": 9,
}
NULL_EQUALITY = {"NullEqualsNothing": 0, "NullEqualsNull": 1}


class OrcCol(C.Structure):
    _fields_ = [("type", C.c_int32), ("_pad", C.c_int32), ("n", C.c_int64),
                ("data", C.c_void_p), ("valid", C.c_void_p)]


def build():
    """(re)build liboracle with the committed Makefile."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libdforacle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.orc_filter_indices.restype = C.c_int64
        _LIB.orc_group_intern.restype = C.c_int64
    return _LIB


def orc_type(t: pa.DataType) -> int:
    if pa.types.is_int32(t) or pa.types.is_date32(t):
        return ORC_I32
    if pa.types.is_int64(t) or pa.types.is_timestamp(t):
        return ORC_I64
    if pa.types.is_decimal128(t):
        return ORC_I128
    if pa.types.is_float64(t):
        return ORC_F64
    if pa.types.is_uint8(t):
        return ORC_U8
    if pa.types.is_uint32(t):
        return ORC_U32
    if pa.types.is_uint64(t):
        return ORC_U64
    raise TypeError(f"oracle: unsupported type {t}")


_NP = {ORC_I32: np.int32, ORC_I64: np.int64, ORC_F64: np.float64, ORC_U8: np.uint8,
       ORC_U32: np.uint32, ORC_U64: np.uint64}


def _flat(arr) -> pa.Array:
    if isinstance(arr, pa.ChunkedArray):
        arr = arr.combine_chunks() if arr.num_chunks != 1 else arr.chunk(0)
    if arr.offset != 0:
        arr = pa.concat_arrays([arr])  # re-materialise at offset 0
    return arr


def values_np(arr) -> np.ndarray:
    """raw fixed-width values as numpy (decimal128 -> (n,2) uint64 lo/hi)."""
    arr = _flat(arr)
    t = orc_type(arr.type)
    buf = arr.buffers()[1]
    n = len(arr)
    if n == 0 or buf is None:
        return np.zeros((0, 2), np.uint64) if t == ORC_I128 else np.zeros(0, _NP[t])
    if t == ORC_I128:
        return np.frombuffer(buf, dtype=np.uint64, count=2 * n).reshape(n, 2)
    return np.frombuffer(buf, dtype=_NP[t], count=n)


class _ColHolder:
    """keeps numpy/arrow buffers alive while the C struct points at them"""

    def __init__(self, arr):
        arr = _flat(arr)
        self.arr = arr
        self.values = np.ascontiguousarray(values_np(arr))
        vb = arr.buffers()[0] if arr.null_count else None
        self.valid = np.frombuffer(vb, dtype=np.uint8).copy() if vb is not None else None
        self.c = OrcCol(orc_type(arr.type), 0, len(arr), self.values.ctypes.data if self.values.size else 0,
                        self.valid.ctypes.data if self.valid is not None else 0)


def _cols(arrs):
    holders = [_ColHolder(a) for a in arrs]
    carr = (OrcCol * len(holders))(*[h.c for h in holders])
    return holders, carr


def _from_values(values: np.ndarray, typ: pa.DataType, valid: np.ndarray | None = None) -> pa.Array:
    """numpy raw values (+ optional bool validity) -> pyarrow array of `typ`."""
    n = values.shape[0]
    data = pa.py_buffer(np.ascontiguousarray(values).tobytes())
    vbuf = None
    nulls = 0
    if valid is not None and not valid.all():
        vbuf = pa.py_buffer(np.packbits(valid.astype(np.uint8), bitorder="little").tobytes())
        nulls = int(n - valid.sum())
    return pa.Array.from_buffers(typ, n, [vbuf, data], null_count=nulls)


def take(table: pa.Table, idx: np.ndarray) -> pa.Table:
    """arrow `take` with -1 = NULL row (joins/utils.rs:1332-1386 build_batch_from_indices)."""
    ind = pa.array(idx, type=pa.int64(), mask=(idx < 0))
    return pa.Table.from_arrays([_flat(table.column(i)).take(ind) for i in range(table.num_columns)],
                                schema=table.schema)


# ----------------------------------------------------------------------------- join

def hash_join(left: pa.Table, right: pa.Table, on, join_type="Inner", null_equality="NullEqualsNothing",
              mode=0, small_build_threshold=1024, min_key_density=0.15, return_indices=False, join_filter=None, null_aware=False):
    """HashJoinExec: left = build side, right = probe side (hash_join/exec.rs:752).
    Output schema = left columns ++ right columns (Inner/Left/Right/Full), left only for
    Left{Semi,Anti}, right only for Right{Semi,Anti}, + `mark` for *Mark joins
    (joins/utils.rs:build_join_schema).
    join_filter = (expr over intermediate columns named f0, f1, ..., [(column index, "Left" | "Right"), ...]):
    JoinFilter (joins/join_filter.rs) — see _hash_join_filtered.
    null_aware = HashJoinExec::null_aware (NOT IN semantics) — see _hash_join_null_aware."""
    if null_aware:
        return _hash_join_null_aware(left, right, on, join_type, null_equality, mode, small_build_threshold, min_key_density, join_filter)
    if join_filter is not None:
        return _hash_join_filtered(left, right, on, join_type, null_equality, mode, small_build_threshold, min_key_density, join_filter)
    L = lib()
    bh, bk = _cols([left.column(l) for l, _ in on])
    ph, pk = _cols([right.column(r) for _, r in on])
    ob, op_, om = C.POINTER(C.c_int64)(), C.POINTER(C.c_int64)(), C.POINTER(C.c_uint8)()
    n = C.c_int64()
    used = C.c_int()
    jt = JOIN_TYPES[join_type]
    rc = L.orc_hash_join(bk, pk, len(on), jt, NULL_EQUALITY[null_equality], mode,
                         C.c_int64(small_build_threshold), C.c_double(min_key_density),
                         C.byref(ob), C.byref(op_), C.byref(om), C.byref(n), C.byref(used))
    if rc != 0:
        raise RuntimeError(f"orc_hash_join rc={rc}")
    cnt = n.value
    bi = np.ctypeslib.as_array(ob, shape=(cnt,)).copy() if cnt else np.zeros(0, np.int64)
    pi = np.ctypeslib.as_array(op_, shape=(cnt,)).copy() if cnt else np.zeros(0, np.int64)
    mk = np.ctypeslib.as_array(om, shape=(cnt,)).copy().astype(bool) if (cnt and om) else np.zeros(0, bool)
    for p in (ob, op_, om):
        if p:
            L.orc_free(p)
    if return_indices:
        return bi, pi, mk, bool(used.value)
    if join_type in ("Inner", "Left", "Right", "Full"):
        lt, rt = take(left, bi), take(right, pi)
        names = [f.name for f in left.schema] + [f.name for f in right.schema]
        out = pa.Table.from_arrays(list(lt.columns) + list(rt.columns), names=names)
    elif join_type in ("LeftSemi", "LeftAnti"):
        out = take(left, bi)
    elif join_type in ("RightSemi", "RightAnti"):
        out = take(right, pi)
    elif join_type == "LeftMark":
        out = take(left, bi).append_column("mark", pa.array(mk))
    else:
        out = take(right, pi).append_column("mark", pa.array(mk))
    return out


def _hash_join_null_aware(left, right, on, join_type, null_equality, mode, small_build_threshold, min_key_density, join_filter):
    """null_aware anti joins = `x NOT IN (subquery)`: validation of HashJoinExec::try_new (hash_join/exec.rs:429-455), then
    process_probe_batch / process_unmatched_build_batch (hash_join/stream.rs:755-808, 937-955, 1016-1076):
      LeftAnti  — a NULL probe key anywhere makes every NOT IN unknown (no output); otherwise left rows with a NULL key
                  are dropped unless the probe side is empty (NULL NOT IN (empty set) is TRUE);
      RightAnti — a NULL build key empties the output; an empty build side emits every probe row (NULL keys
                  included, build_batch_empty_build_side); otherwise probe rows with a NULL key are dropped."""
    if join_type not in ("LeftAnti", "RightAnti"):
        raise ValueError(f"null_aware can only be true for LeftAnti joins and RightAnti joins with `CollectLeft` `PartitionMode`, got {join_type}")
    if len(on) != 1:
        raise ValueError(f"null_aware anti join only supports single column join key, got {len(on)} columns")
    if join_type == "RightAnti" and join_filter is not None:
        raise ValueError("null_aware RightAnti join does not support a join filter")
    lkey, rkey = on[0]
    keep_valid = lambda t, key: take(t, np.flatnonzero(np.asarray(t.column(key).is_valid())).astype(np.int64))
    plain = lambda: hash_join(left, right, on, join_type, null_equality, mode, small_build_threshold, min_key_density, join_filter=join_filter)
    if join_type == "LeftAnti":
        if right.column(rkey).null_count > 0:
            return left.slice(0, 0)
        out = plain()
        return keep_valid(out, lkey) if right.num_rows > 0 else out
    if left.column(lkey).null_count > 0:
        return right.slice(0, 0)
    out = plain()
    return keep_valid(out, rkey) if left.num_rows > 0 else out


def _hash_join_filtered(left, right, on, join_type, null_equality, mode, small_build_threshold, min_key_density, join_filter):
    """HashJoinExec with a JoinFilter.  The reference first finds the key-equal (build, probe) pairs, then
    apply_join_filter_to_indices (joins/utils.rs:1248-1318) builds the intermediate batch from the filter's
    column_indices, evaluates the expression and keeps the pairs whose value is TRUE (NULL = false); only those
    pairs mark build rows visited and go through adjust_indices_by_join_type (joins/utils.rs:1432-1488,
    hash_join/stream.rs:687-1000).  Unmatched rows of outer / anti / mark joins are therefore rows without a
    PASSING pair."""
    expr, cols = join_filter
    bi, pi, _, _ = hash_join(left, right, on, "Inner", null_equality, mode, small_build_threshold, min_key_density, return_indices=True)
    inter = pa.table({f"f{i}": (take(left, bi) if side == "Left" else take(right, pi)).column(idx) for i, (idx, side) in enumerate(cols)}) \
        if len(bi) else pa.table({f"f{i}": pa.array([], type=(left if side == "Left" else right).schema.field(idx).type) for i, (idx, side) in enumerate(cols)})
    d = evaluate(expr, inter)
    vals = np.repeat(d.values, len(bi)) if d.scalar else d.values
    keep = np.asarray(vals, dtype=bool)
    v = None if d.valid is None else (np.repeat(d.valid, len(bi)) if d.scalar else d.valid)
    if v is not None:
        keep = keep & v
    bi, pi = bi[keep], pi[keep]
    nl, nr = left.num_rows, right.num_rows
    l_hit, r_hit = np.zeros(nl, bool), np.zeros(nr, bool)
    l_hit[bi] = True
    r_hit[pi] = True
    l_un, r_un = np.nonzero(~l_hit)[0].astype(np.int64), np.nonzero(~r_hit)[0].astype(np.int64)
    both = lambda b, p: pa.Table.from_arrays(list(take(left, b).columns) + list(take(right, p).columns),
                                             names=[f.name for f in left.schema] + [f.name for f in right.schema])
    neg = lambda k: np.full(k, -1, np.int64)
    if join_type == "Inner":
        return both(bi, pi)
    if join_type == "Left":
        return both(np.concatenate([bi, l_un]), np.concatenate([pi, neg(len(l_un))]))
    if join_type == "Right":
        return both(np.concatenate([bi, neg(len(r_un))]), np.concatenate([pi, r_un]))
    if join_type == "Full":
        return both(np.concatenate([bi, neg(len(r_un)), l_un]), np.concatenate([pi, r_un, neg(len(l_un))]))
    if join_type == "LeftSemi":
        return take(left, np.nonzero(l_hit)[0].astype(np.int64))
    if join_type == "LeftAnti":
        return take(left, l_un)
    if join_type == "RightSemi":
        return take(right, np.nonzero(r_hit)[0].astype(np.int64))
    if join_type == "RightAnti":
        return take(right, r_un)
    if join_type == "LeftMark":
        return left.append_column("mark", pa.array(l_hit))
    return right.append_column("mark", pa.array(r_hit))


def partitioned_inner_join_i64(build_keys: np.ndarray, probe_keys: np.ndarray, nthreads: int):
    """CPU-baseline leg: RepartitionExec(Hash) x2 -> HashJoinExec(Partitioned); returns (pairs, checksum)."""
    L = lib()
    b = np.ascontiguousarray(build_keys, dtype=np.int64)
    p = np.ascontiguousarray(probe_keys, dtype=np.int64)
    pairs, chk = C.c_int64(), C.c_uint64()
    L.orc_partitioned_inner_join_i64(C.c_void_p(b.ctypes.data), C.c_int64(len(b)), C.c_void_p(p.ctypes.data),
                                     C.c_int64(len(p)), nthreads, C.byref(pairs), C.byref(chk))
    return pairs.value, chk.value


def partitioned_q3_join(bkeys, bdate, bprio, pkeys, pprice, pdisc, nthreads: int):
    """CPU-baseline leg with TPC-H Q3's payload: RepartitionExec(Hash) of every column of both sides -> HashJoinExec(Partitioned) ->
    build_batch_from_indices per 8192-row probe batch (dforacle.c orc_partitioned_q3_join).  Keys int64, o_orderdate / o_shippriority
    int32, the two Decimal128 columns as (n, 2) uint64 / int64 arrays (low word first).  Returns (rows, checksum)."""
    L = lib()
    arrs = [np.ascontiguousarray(bkeys, dtype=np.int64), np.ascontiguousarray(bdate, dtype=np.int32), np.ascontiguousarray(bprio, dtype=np.int32),
            np.ascontiguousarray(pkeys, dtype=np.int64), np.ascontiguousarray(pprice), np.ascontiguousarray(pdisc)]
    assert arrs[4].nbytes == 16 * len(arrs[3]) and arrs[5].nbytes == 16 * len(arrs[3]) and len(arrs[1]) == len(arrs[0]) == len(arrs[2])
    rows, chk = C.c_int64(), C.c_uint64()
    L.orc_partitioned_q3_join.restype = C.c_int
    rc = L.orc_partitioned_q3_join(C.c_void_p(arrs[0].ctypes.data), C.c_void_p(arrs[1].ctypes.data), C.c_void_p(arrs[2].ctypes.data), C.c_int64(len(arrs[0])),
                                   C.c_void_p(arrs[3].ctypes.data), C.c_void_p(arrs[4].ctypes.data), C.c_void_p(arrs[5].ctypes.data), C.c_int64(len(arrs[3])),
                                   int(nthreads), C.byref(rows), C.byref(chk))
    if rc != 0:
        raise MemoryError("orc_partitioned_q3_join: allocation failed")
    return rows.value, chk.value


# ---------------------------------------------------------------------- expressions

def _decimal_unscaled(v, scale: int) -> int:
    return int(Decimal(str(v)).scaleb(scale).to_integral_value())


def _i128_np(ints) -> np.ndarray:
    out = np.zeros((len(ints), 2), np.uint64)
    for i, v in enumerate(ints):
        v &= (1 << 128) - 1
        out[i, 0] = v & 0xFFFFFFFFFFFFFFFF
        out[i, 1] = v >> 64
    return out


def _lit_values(value, typ: pa.DataType) -> np.ndarray:
    t = orc_type(typ)
    if t == ORC_I128:
        return _i128_np([_decimal_unscaled(value, typ.scale)])
    if pa.types.is_date32(typ) and not isinstance(value, (int, np.integer)):
        value = pa.scalar(value, type=pa.date32()).cast(pa.int32()).as_py()
    return np.array([value], dtype=_NP[t])


def _clamp_dec(p, s):
    return pa.decimal128(min(38, p), min(38, s))


def arith_result_type(op: str, lt: pa.DataType, rt: pa.DataType) -> pa.DataType:
    """arrow-arith decimal result typing as relied on by BinaryExpr (expr-common/src/
    type_coercion/binary.rs:168-186): add/sub -> scale max(s1,s2), precision
    max(s1,s2)+max(p1-s1,p2-s2)+1; mul -> (p1+p2+1, s1+s2); both clamped to 38 (arrow-rs
    clamps, it does not raise).  Golden: tpch/plans/q1.slt.part:45-46, answers q1:42-45."""
    if pa.types.is_decimal128(lt) and pa.types.is_decimal128(rt):
        p1, s1, p2, s2 = lt.precision, lt.scale, rt.precision, rt.scale
        if op in "+-":
            s = max(s1, s2)
            return _clamp_dec(s + max(p1 - s1, p2 - s2) + 1, s)
        if op == "*":
            return _clamp_dec(p1 + p2 + 1, s1 + s2)
        if op == "/":
            # arrow-arith numeric.rs decimal_op Op::Div ("a fixed scale increment of 4"); pinned by binary.rs:3042-3075
            # (Decimal(10,0) / Decimal(10,0) -> Decimal(14,4)) and :4800-4817 (Decimal(10,0) / Decimal(10,2) -> Decimal(16,4))
            s = min(38, s1 + 4)
            return _clamp_dec(s - s1 + s2 + p1, s)
        if op == "%":
            s = max(s1, s2)   # binary.rs:4819-4836: Decimal(10,0) % Decimal(10,2) -> Decimal(10,2)
            return _clamp_dec(s + min(p1 - s1, p2 - s2), s)
    if lt != rt:
        raise TypeError(f"oracle arith: operand types differ {lt} vs {rt} (planner inserts casts)")
    return lt


class Datum:
    """ColumnarValue (physical-expr-common): array or scalar + validity"""

    def __init__(self, values, typ, valid=None, scalar=False):
        self.values, self.typ, self.valid, self.scalar = values, typ, valid, scalar

    def to_array(self, n) -> pa.Array:
        vals, valid = self.values, self.valid
        if self.scalar:
            vals = np.repeat(vals, n, axis=0)
            valid = None if valid is None else np.repeat(valid, n)
        if pa.types.is_boolean(self.typ):
            return pa.array(vals.astype(bool), mask=None if valid is None else ~valid)
        return _from_values(vals, self.typ, valid)


def _and_valid(a, b, n):
    def ex(d):
        if d.valid is None:
            return None
        return np.repeat(d.valid, n) if d.scalar else d.valid
    va, vb = ex(a), ex(b)
    if va is None:
        return vb
    if vb is None:
        return va
    return va & vb


def evaluate(expr, table: pa.Table) -> Datum:
    """PhysicalExpr::evaluate (physical-expr-common/src/physical_expr.rs:88) for the node
    kinds on the hot path (Column column.rs:121, Literal, CastExpr, BinaryExpr
    binary.rs:536-656, IsNull, Not)."""
    L = lib()
    n = table.num_rows
    kind = expr[0]
    if kind == "col":
        arr = _flat(table.column(expr[1]))
        if pa.types.is_boolean(arr.type):
            vals = np.asarray(arr.fill_null(False).to_numpy(zero_copy_only=False), dtype=bool)
            valid = None if arr.null_count == 0 else ~np.asarray(arr.is_null().to_numpy(zero_copy_only=False))
            return Datum(vals, arr.type, valid)
        valid = None
        if arr.null_count:
            valid = ~np.asarray(arr.is_null().to_numpy(zero_copy_only=False))
        return Datum(values_np(arr), arr.type, valid)
    if kind == "lit":
        _, value, typ = expr
        if value is None and pa.types.is_boolean(typ):
            return Datum(np.array([False]), typ, np.array([False]), scalar=True)
        if value is None:
            t = orc_type(typ)
            z = np.zeros((1, 2), np.uint64) if t == ORC_I128 else np.zeros(1, _NP[t])
            return Datum(z, typ, np.array([False]), scalar=True)
        if pa.types.is_boolean(typ):
            return Datum(np.array([bool(value)]), typ, None, scalar=True)
        return Datum(_lit_values(value, typ), typ, None, scalar=True)
    if kind == "cast":
        d = evaluate(expr[1], table)
        to = expr[2]
        src, dst = orc_type(d.typ), orc_type(to)
        m = d.values.shape[0]
        if dst == ORC_I128 and src == ORC_F64:   # arrow-cast cast_floating_point_to_decimal128: (v * 10^scale).round() as i128
            mv = d.values.astype(np.float64) * float(10 ** to.scale)
            r = np.where(np.abs(mv) < 2.0 ** 52, np.copysign(np.floor(np.abs(mv) + 0.5), mv), mv)
            return Datum(_i128_np([int(x) for x in r.tolist()]), to, d.valid, d.scalar)
        if dst == ORC_I128:
            out = np.zeros((m, 2), np.uint64)
            L.orc_cast_to_i128(src, C.c_void_p(np.ascontiguousarray(d.values).ctypes.data), C.c_int64(m), C.c_void_p(out.ctypes.data))
            from_scale = d.typ.scale if pa.types.is_decimal128(d.typ) else 0
            if to.scale > from_scale:
                out2 = np.zeros_like(out)
                L.orc_decimal_rescale_up(C.c_void_p(out.ctypes.data), C.c_int64(m), to.scale - from_scale, C.c_void_p(out2.ctypes.data))
                out = out2
            elif to.scale < from_scale:
                # arrow-cast cast_decimal_to_decimal, scale reduction: divide by 10^k, round half away from zero; a value beyond the
                # target precision is an error (the reference's CastExpr is not `safe`)
                div = 10 ** (from_scale - to.scale)
                vals = []
                for i, x in enumerate(_py_ints(Datum(out, d.typ, None, d.scalar))):
                    q, r = divmod(abs(x), div)
                    q = q + 1 if 2 * r >= div else q
                    v = -q if x < 0 else q
                    if (d.valid is None or d.valid[i]) and abs(v) >= 10 ** to.precision:
                        raise OverflowError(f"Arrow error: Invalid argument error: {v} is too large to store in a {to}")
                    vals.append(v)
                out = _i128_np(vals)
            return Datum(out, to, d.valid, d.scalar)
        if dst == ORC_F64 and src in (ORC_I32, ORC_I64):
            return Datum(d.values.astype(np.float64), to, d.valid, d.scalar)
        if dst == ORC_F64 and src == ORC_I128:   # arrow-cast cast_decimal_to_float: x as f64 / 10_f64.powi(scale)
            return Datum(np.array([float(v) / float(10 ** d.typ.scale) for v in _py_ints(d)], dtype=np.float64), to, d.valid, d.scalar)
        if dst == ORC_I64 and src in (ORC_I32, ORC_U8, ORC_U32):
            return Datum(d.values.astype(np.int64), to, d.valid, d.scalar)
        if dst == src:
            return Datum(d.values, to, d.valid, d.scalar)
        raise NotImplementedError(f"oracle cast {d.typ} -> {to}")
    if kind == "date_part":
        # date_part(YEAR | MONTH | DAY, Date32) -> Int32 (functions/src/datetime/date_part.rs:165-187), through numpy's calendar
        _, part, arg = expr
        d = evaluate(arg, table)
        if not pa.types.is_date32(d.typ):
            raise TypeError(f"oracle date_part over {d.typ}")
        days = d.values.astype("int64").astype("datetime64[D]")
        months = days.astype("datetime64[M]")
        if part == "year":
            v = days.astype("datetime64[Y]").astype(np.int64) + 1970
        elif part == "month":
            v = months.astype(np.int64) % 12 + 1
        elif part == "day":
            v = (days - months.astype("datetime64[D]")).astype(np.int64) + 1
        else:
            raise NotImplementedError(part)
        return Datum(v.astype(np.int32), pa.int32(), d.valid, d.scalar)
    if kind == "is_null":
        d = evaluate(expr[1], table)
        m = 1 if d.scalar else n
        v = np.zeros(m, bool) if d.valid is None else ~d.valid
        return Datum(v, pa.bool_(), None, d.scalar)
    if kind == "not":
        d = evaluate(expr[1], table)
        return Datum(~d.values.astype(bool), pa.bool_(), d.valid, d.scalar)
    if kind == "case":
        # CaseExpr without base expression, one WHEN (physical-expr/src/expressions/case.rs:895-980 case_when_no_expr,
        # :981-1040 expr_or_expr): THEN where the condition is TRUE; NULL conditions count as FALSE
        # (prep_null_mask_filter) and take ELSE; no ELSE = NULL
        _, ce, te, ee = expr
        c, a = evaluate(ce, table), evaluate(te, table)
        b = evaluate(ee, table) if ee is not None else evaluate(("lit", None, a.typ), table)
        if a.typ != b.typ:
            raise TypeError(f"oracle case: {a.typ} vs {b.typ}")

        def full(d):
            vals = np.repeat(d.values, n, axis=0) if d.scalar else d.values
            valid = np.ones(n, bool) if d.valid is None else (np.repeat(d.valid, n) if d.scalar else d.valid)
            return vals, valid
        cv, cvalid = full(c)
        take = cv.astype(bool) & cvalid
        (av, avalid), (bv, bvalid) = full(a), full(b)
        vals = np.where(take[:, None] if av.ndim == 2 else take, av, bv)
        valid = np.where(take, avalid, bvalid)
        return Datum(vals, a.typ, None if valid.all() else valid)
    if kind == "bin":
        _, op, le, re_ = expr
        a, b = evaluate(le, table), evaluate(re_, table)
        if op in ("and", "or"):
            # Kleene logic (arrow and_kleene / or_kleene, binary.rs:543-603)
            av = np.repeat(a.values, n) if a.scalar else a.values
            bv = np.repeat(b.values, n) if b.scalar else b.values
            avd = np.ones(n, bool) if a.valid is None else (np.repeat(a.valid, n) if a.scalar else a.valid)
            bvd = np.ones(n, bool) if b.valid is None else (np.repeat(b.valid, n) if b.scalar else b.valid)
            at, bt = av.astype(bool) & avd, bv.astype(bool) & bvd
            af, bf = (~av.astype(bool)) & avd, (~bv.astype(bool)) & bvd
            if op == "and":
                val, valid = at & bt, (at & bt) | af | bf
            else:
                val, valid = at | bt, at | bt | (af & bf)
            return Datum(val, pa.bool_(), None if valid.all() else valid)
        scalar = a.scalar and b.scalar
        m = 1 if scalar else n
        valid = _and_valid(a, b, m) if not scalar else (None if (a.valid is None and b.valid is None) else
                                                         np.array([bool((a.valid is None or a.valid[0]) and (b.valid is None or b.valid[0]))]))
        if op in ("+", "-", "*"):
            rt = arith_result_type(op, a.typ, b.typ)
            av, bv = a, b
            if pa.types.is_decimal128(rt) and op in "+-":
                # arrow-arith rescales both sides to the result scale before add/sub
                av = _rescale(a, rt.scale)
                bv = _rescale(b, rt.scale)
            t = orc_type(rt)
            out = np.zeros((m, 2), np.uint64) if t == ORC_I128 else np.zeros(m, _NP[t])
            x, y = np.ascontiguousarray(av.values), np.ascontiguousarray(bv.values)
            rc = L.orc_arith({"+": 0, "-": 1, "*": 2}[op], t, C.c_void_p(x.ctypes.data), int(av.scalar and not scalar),
                             C.c_void_p(y.ctypes.data), int(bv.scalar and not scalar), C.c_int64(m), C.c_void_p(out.ctypes.data))
            assert rc == 0
            return Datum(out, rt, valid, scalar)
        if op in ("/", "%"):
            return _divmod(op, a, b, m, valid, scalar)
        ops = {"=": 0, "!=": 1, "<": 2, "<=": 3, ">": 4, ">=": 5}
        if op in ops:
            av, bv = a, b
            if pa.types.is_decimal128(a.typ) and pa.types.is_decimal128(b.typ) and a.typ.scale != b.typ.scale:
                s = max(a.typ.scale, b.typ.scale)
                av, bv = _rescale(a, s), _rescale(b, s)
            elif orc_type(a.typ) != orc_type(b.typ):
                raise TypeError(f"oracle cmp: {a.typ} vs {b.typ}")
            bits = np.zeros((m + 7) // 8, np.uint8)
            x, y = np.ascontiguousarray(av.values), np.ascontiguousarray(bv.values)
            rc = L.orc_cmp(ops[op], orc_type(av.typ), C.c_void_p(x.ctypes.data), int(av.scalar and not scalar),
                           C.c_void_p(y.ctypes.data), int(bv.scalar and not scalar), C.c_int64(m), C.c_void_p(bits.ctypes.data))
            assert rc == 0
            vals = np.unpackbits(bits, bitorder="little")[:m].astype(bool)
            return Datum(vals, pa.bool_(), valid, scalar)
    raise NotImplementedError(f"oracle expr {expr!r}")


def _py_ints(d: Datum) -> list:
    """the values as Python integers (Decimal128: unscaled)"""
    if orc_type(d.typ) == ORC_I128:
        out = []
        for lo, hi in d.values.tolist():
            v = (int(hi) << 64) | int(lo)
            out.append(v - (1 << 128) if v >> 127 else v)
        return out
    return [int(v) for v in d.values.tolist()]


def _divmod(op, a: Datum, b: Datum, m: int, valid, scalar: bool) -> Datum:
    """BinaryExpr Divide / Modulo = arrow-arith `div` / `rem` (binary.rs:636-637): integers and decimals truncate toward zero
    (Rust `/`, `%`: the remainder takes the dividend's sign), a zero divisor in a valid row is the error the reference tests
    pin (arithmetic_divide_zero, binary.rs:4955-5003); Float64 follows IEEE.  Exact integer arithmetic on Python ints."""
    rt = arith_result_type(op, a.typ, b.typ)
    if pa.types.is_float64(rt):
        x = np.repeat(a.values, m) if (a.scalar and not scalar) else a.values
        y = np.repeat(b.values, m) if (b.scalar and not scalar) else b.values
        with np.errstate(all="ignore"):
            out = np.divide(x, y) if op == "/" else np.fmod(x, y)
        return Datum(out.astype(np.float64), rt, valid, scalar)
    la, lb = 1, 1
    if pa.types.is_decimal128(rt):
        if op == "/":
            k = rt.scale - a.typ.scale + b.typ.scale
            la, lb = (10 ** k, 1) if k >= 0 else (1, 10 ** -k)
        else:
            la, lb = 10 ** (rt.scale - a.typ.scale), 10 ** (rt.scale - b.typ.scale)
    xs, ys = _py_ints(a), _py_ints(b)
    if a.scalar and not scalar:
        xs = xs * m
    if b.scalar and not scalar:
        ys = ys * m
    bits = {ORC_I32: 32, ORC_I64: 64, ORC_I128: 128}[orc_type(rt)]
    out = []
    for i in range(m):
        if valid is not None and not valid[i]:
            out.append(0)
            continue
        x, y = xs[i] * la, ys[i] * lb
        if y == 0:
            raise ZeroDivisionError("Arrow error: Divide by zero error")
        q = abs(x) // abs(y)
        q = -q if (x < 0) != (y < 0) else q
        v = q if op == "/" else x - q * y
        if not -(1 << (bits - 1)) <= v < (1 << (bits - 1)) or abs(x) >= 1 << (bits - 1) and bits == 128:
            raise OverflowError("Arrow error: Arithmetic overflow")
        out.append(v)
    t = orc_type(rt)
    vals = _i128_np(out) if t == ORC_I128 else np.array(out, dtype=_NP[t])
    return Datum(vals, rt, valid, scalar)


def _rescale(d: Datum, scale: int) -> Datum:
    if d.typ.scale == scale:
        return d
    L = lib()
    m = d.values.shape[0]
    out = np.zeros((m, 2), np.uint64)
    L.orc_decimal_rescale_up(C.c_void_p(np.ascontiguousarray(d.values).ctypes.data), C.c_int64(m), scale - d.typ.scale,
                             C.c_void_p(out.ctypes.data))
    return Datum(out, pa.decimal128(min(38, d.typ.precision + scale - d.typ.scale), scale), d.valid, d.scalar)


def project(table: pa.Table, exprs) -> pa.Table:
    """ProjectionExec (projection.rs:713-740): exprs = [(expr, name)]"""
    n = table.num_rows
    return pa.Table.from_arrays([evaluate(e, table).to_array(n) for e, _ in exprs], names=[nm for _, nm in exprs])


def filter(table: pa.Table, predicate, projection=None) -> pa.Table:
    """FilterExec (filter.rs:1367-1444): evaluate predicate, optional embedded projection
    (column indices/names), filter_record_batch; NULL predicate rows are dropped."""
    L = lib()
    n = table.num_rows
    d = evaluate(predicate, table)
    mask = d.values.astype(bool)
    valid = d.valid
    if d.scalar:
        mask = np.repeat(mask, n)
        valid = None if valid is None else np.repeat(valid, n)
    bits = np.packbits(mask.astype(np.uint8), bitorder="little")
    vbits = None if valid is None else np.packbits(valid.astype(np.uint8), bitorder="little")
    idx = np.zeros(max(n, 1), np.int64)
    if bits.size == 0:
        bits = np.zeros(1, np.uint8)
    k = L.orc_filter_indices(C.c_void_p(bits.ctypes.data), C.c_void_p(vbits.ctypes.data) if vbits is not None else None,
                             C.c_int64(n), C.c_void_p(idx.ctypes.data))
    if projection is not None:
        table = table.select(projection)
    return take(table, idx[:k])


# ------------------------------------------------------------------------ aggregate

def sum_result_type(t: pa.DataType) -> pa.DataType:
    """SUM(Decimal128(p,s)) -> Decimal128(min(38,p+10), s) (functions-aggregate/src/sum.rs:232-260)"""
    if pa.types.is_decimal128(t):
        return pa.decimal128(min(38, t.precision + 10), t.scale)
    if pa.types.is_int32(t) or pa.types.is_int64(t):
        return pa.int64()
    return t


def avg_result_type(t: pa.DataType) -> pa.DataType:
    """AVG(Decimal128(p,s)) -> Decimal128(min(38,p+4), min(38,s+4)) (average.rs:219-252)"""
    if pa.types.is_decimal128(t):
        return pa.decimal128(min(38, t.precision + 4), min(38, t.scale + 4))
    return pa.float64()


def avg_sum_type(t: pa.DataType) -> pa.DataType:
    """the type AVG's partial `sum` state carries: avg_sum_data_type (functions-aggregate/src/average.rs:131-172) — the
    input precision plus 13 digits of headroom, never narrower than the input type's maximum: Decimal128(38, s) while
    p + 13 <= 38, Decimal256 beyond that (no device or oracle representation: an error here)"""
    if pa.types.is_decimal128(t):
        if t.precision + 13 > 38:
            raise NotImplementedError(f"AVG({t}) accumulates in Decimal256 in the reference")
        return pa.decimal128(38, t.scale)
    return pa.float64()


def aggregate(table: pa.Table, group_by, aggs, mode="Single", return_types=None) -> pa.Table:
    """AggregateExec (aggregates/mod.rs:839, aggregate_hash_table/common.rs:205-300).
    group_by = [(expr, name)], aggs = [(func, expr_or_None, name)] with func in
    sum/avg/count/min/max.  Output = group columns ++ aggregate columns, groups in first-seen
    order.  mode (AggregateMode, aggregates/mod.rs:289-400):
      Single / SinglePartitioned : raw rows -> final values
      Partial                    : raw rows -> state columns named by format_state_name (AVG -> `name[count]` UInt64 +
                                   `name[sum]`; SUM `name[sum]`, COUNT `name[count]`, MIN / MAX `name[value]`;
                                   average.rs:317-360, sum.rs:281-301, count.rs:317-323, udaf.rs:579-585)
      Final / FinalPartitioned   : state columns (same layout, group columns first) -> final values;
                                   aggregate expressions are ignored (merge_batch path).  `return_types`
                                   {name: type} = the aggregates' declared return types: the reference's AggregateExec
                                   carries them (AggregateFunctionExpr::return_field); AVG over a Decimal128 state needs it,
                                   because the state's sum type no longer tells the argument's precision."""
    L = lib()
    n = table.num_rows
    final = mode in ("Final", "FinalPartitioned")
    partial = mode == "Partial"
    if final:
        key_arrs = [_flat(table.column(i)) for i in range(len(group_by))]
    else:
        key_arrs = [evaluate(e, table).to_array(n) for e, _ in group_by]
    if key_arrs:
        kh, kc = _cols(key_arrs)
        gids = np.zeros(max(n, 1), np.int64)
        first = np.zeros(max(n, 1), np.int64)
        ng = L.orc_group_intern(kc, len(key_arrs), C.c_int64(n), C.c_void_p(gids.ctypes.data), C.c_void_p(first.ctypes.data))
        first = first[:ng]
    else:
        # no GROUP BY -> AggregateStream: exactly one output row even for empty input
        ng, gids, first = 1, np.zeros(max(n, 1), np.int64), np.zeros(0, np.int64)
    out_cols, names = [], []
    for (e, nm), arr in zip(group_by, key_arrs):
        out_cols.append(arr.take(pa.array(first, type=pa.int64())))
        names.append(nm)

    def zeros(tt):
        return np.zeros((max(ng, 1), 2), np.uint64) if tt == ORC_I128 else np.zeros(max(ng, 1), _NP[tt])

    def acc(op, varr):
        """one orc_accumulate call -> (values, seen)"""
        vh, vc = _cols([varr])
        t = orc_type(varr.type)
        out = np.zeros(max(ng, 1), np.int64) if op == 3 else zeros(t)
        seen = np.zeros(max(ng, 1), np.uint8)
        rc = L.orc_accumulate(op, C.byref(vc[0]), C.c_void_p(gids.ctypes.data), C.c_int64(ng), None,
                              C.c_void_p(out.ctypes.data), C.c_void_p(seen.ctypes.data))
        assert rc == 0
        return out, seen[:ng].astype(bool)

    state_col = len(group_by)
    for func, e, nm in aggs:
        cnt_state = None
        if final:
            if func == "avg":
                cnt_state = _flat(table.column(state_col)); state_col += 1
            varr = _flat(table.column(state_col)); state_col += 1
        elif func == "count" and e is None:
            varr = pa.array(np.zeros(n, np.int64))  # COUNT(*) counts rows
        else:
            varr = evaluate(e, table).to_array(n)
        if not final and func in ("sum", "avg") and (pa.types.is_int32(varr.type) or pa.types.is_uint8(varr.type)):
            varr = varr.cast(pa.int64())
        if not final and func == "avg" and pa.types.is_int64(varr.type):
            varr = varr.cast(pa.float64())  # AVG over integers is coerced to Float64 (average.rs coerce_types)
        t = orc_type(varr.type)
        if func == "count":
            if final:
                vals, _ = acc(0, varr.cast(pa.int64()))  # merge partial counts by summing
            else:
                vals, _ = acc(3, varr)
            out_cols.append(pa.array(vals[:ng], type=pa.int64()))
            names.append(nm + "[count]" if partial else nm)          # format_state_name, count.rs:317-323
        elif func in ("sum", "min", "max"):
            vals, seen = acc({"sum": 0, "min": 1, "max": 2}[func], varr)
            rt = varr.type if (final or func != "sum") else sum_result_type(varr.type)
            out_cols.append(_from_values(vals[:ng], rt, seen))
            # state names: `name[sum]` (sum.rs:293-299); MIN / MAX use the default state_fields, `name[value]` (expr/src/udaf.rs:579-585)
            names.append((nm + ("[sum]" if func == "sum" else "[value]")) if partial else nm)
        elif func == "avg":
            sums, seen = acc(0, varr)
            if final:
                cnt, _ = acc(0, cnt_state.cast(pa.int64()))
                sum_t = varr.type
            else:
                cnt, _ = acc(3, varr)
                sum_t = avg_sum_type(varr.type)
            if partial:
                out_cols.append(pa.array(cnt[:ng].astype(np.uint64), type=pa.uint64()))
                names.append(nm + "[count]")
                out_cols.append(_from_values(sums[:ng], sum_t, seen))
                names.append(nm + "[sum]")
                continue
            valid = cnt[:ng] > 0
            if t == ORC_I128:
                # AvgGroupsAccumulator<Decimal128> (average.rs:934-960) + DecimalAverager
                if final:
                    if not return_types or nm not in return_types:
                        raise ValueError(f"Final AVG over the Decimal128 state of '{nm}' needs its declared return type (return_types)")
                    rt = return_types[nm]
                else:
                    rt = avg_result_type(varr.type)
                outv = np.zeros((max(ng, 1), 2), np.uint64)
                rc = L.orc_decimal_avg(C.c_void_p(sums.ctypes.data), C.c_void_p(cnt.ctypes.data), C.c_int64(ng), sum_t.scale, rt.scale,
                                       C.c_void_p(outv.ctypes.data))
                if rc != 0:
                    raise ArithmeticError("Arithmetic Overflow in AvgAccumulator")
                out_cols.append(_from_values(outv[:ng], rt, valid))
            else:
                s = sums.astype(np.float64)
                with np.errstate(divide="ignore", invalid="ignore"):
                    av = s[:ng] / cnt[:ng].astype(np.float64)  # sum / count as f64 (average.rs:374-395)
                out_cols.append(_from_values(np.where(valid, av, 0.0), pa.float64(), valid))
            names.append(nm)
        else:
            raise NotImplementedError(func)
    return pa.Table.from_arrays(out_cols, names=names)


# ---------------------------------------------------------------- repartition / sort

def create_hashes(arrs, seed: int) -> np.ndarray:
    L = lib()
    h, c = _cols(arrs)
    n = len(h[0].arr)
    out = np.zeros(max(n, 1), np.uint64)
    L.orc_create_hashes(c, len(arrs), C.c_int64(n), C.c_uint64(seed), C.c_void_p(out.ctypes.data))
    return out[:n]


def hash_partition(table: pa.Table, key_names, nparts: int):
    """RepartitionExec Hash (repartition/mod.rs:1111-1150) -> list of nparts Tables, input
    order preserved inside each partition."""
    L = lib()
    n = table.num_rows
    h, c = _cols([table.column(k) for k in key_names])
    part = np.zeros(max(n, 1), np.uint32)
    L.orc_hash_partition(c, len(key_names), C.c_int64(n), nparts, C.c_void_p(part.ctypes.data))
    part = part[:n]
    return [take(table, np.nonzero(part == p)[0].astype(np.int64)) for p in range(nparts)], part


def sort(table: pa.Table, keys, fetch=None) -> pa.Table:
    """SortExec (sorts/sort.rs:1366): keys = [(name, descending, nulls_first)], optional
    fetch = TopK (topk/mod.rs:397)."""
    L = lib()
    n = table.num_rows
    h, c = _cols([table.column(k) for k, _, _ in keys])
    desc = np.array([int(d) for _, d, _ in keys], np.uint8)
    nf = np.array([int(f) for _, _, f in keys], np.uint8)
    idx = np.zeros(max(n, 1), np.int64)
    L.orc_lexsort(c, C.c_void_p(desc.ctypes.data), C.c_void_p(nf.ctypes.data), len(keys), C.c_int64(n), C.c_void_p(idx.ctypes.data))
    idx = idx[:n]
    if fetch is not None:
        idx = idx[:fetch]
    return take(table, idx)
