"""shared helpers for the parity tests"""
import json
import os

import pyarrow as pa

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return json.load(open(os.path.join(GOLDEN, name)))


def i32_table(columns, data, repeat=1):
    arrs = [pa.array(list(v) * repeat, type=pa.int32()) for v in data]
    return pa.Table.from_arrays(arrs, names=columns)


def rows(table: pa.Table):
    """list of row tuples (python values)"""
    cols = [[("NaN" if isinstance(v, float) and v != v else v) for v in c.to_pylist()] for c in table.columns]
    return [tuple(c[i] for c in cols) for i in range(table.num_rows)]


def _key(row):
    return tuple((v is None, v == "NaN", 0 if (v is None or v == "NaN") else v) for v in row)


def sorted_rows(table: pa.Table):
    """order-insensitive comparison form (the reference's batches_to_sort_string /
    assert_batches_sorted_eq!, and the fuzzers' sorted formatted rows, join_fuzz.rs:914-925)"""
    return sorted(rows(table), key=_key)


def assert_tables_equal(actual: pa.Table, expected: pa.Table, ordered=False, check_types=True):
    assert actual.num_columns == expected.num_columns, (actual.schema, expected.schema)
    if check_types:
        for fa, fe in zip(actual.schema, expected.schema):
            assert fa.type == fe.type, f"type mismatch {fa} vs {fe}"
    if ordered:
        if actual.num_rows == expected.num_rows and all(a.equals(b) for a, b in zip(actual.columns, expected.columns)):
            return  # fast exact path (large tables); falls through to the row-wise diff otherwise (NaN, messages)
        assert rows(actual) == rows(expected)
    else:
        if actual.num_rows == expected.num_rows and actual.num_rows > 20_000:
            # large multisets: both sides sorted by every column inside Arrow (python row tuples cost ~10 us a value); anything this cannot
            # prove equal (NaN payloads, a real difference) falls through to the row-wise form, which also words the failure
            try:
                names = [f"c{i}" for i in range(actual.num_columns)]
                keys = [(n, "ascending") for n in names]
                a = actual.rename_columns(names).combine_chunks().sort_by(keys)
                e = expected.rename_columns(names).cast(actual.rename_columns(names).schema).combine_chunks().sort_by(keys)
                if all(x.equals(y) for x, y in zip(a.columns, e.columns)):
                    return
            except (pa.ArrowInvalid, pa.ArrowNotImplementedError, pa.ArrowTypeError):
                pass
        assert sorted_rows(actual) == sorted_rows(expected)


def to_oracle_expr(e):
    """product PhysicalExpr tree -> the oracle's tuple AST (test glue only)"""
    from datafusion_amd import expr as X
    if isinstance(e, X.Column):
        return ("col", e.name)
    if isinstance(e, X.Literal):
        return ("lit", e.value, e.type)
    if isinstance(e, X.CastExpr):
        return ("cast", to_oracle_expr(e.expr), e.cast_type)
    if isinstance(e, X.BinaryExpr):
        return ("bin", e.op, to_oracle_expr(e.left), to_oracle_expr(e.right))
    if isinstance(e, X.IsNullExpr):
        return ("is_null", to_oracle_expr(e.arg))
    if isinstance(e, X.IsNotNullExpr):
        return ("not", ("is_null", to_oracle_expr(e.arg)))
    if isinstance(e, X.NotExpr):
        return ("not", to_oracle_expr(e.arg))
    if isinstance(e, X.InListExpr):
        return to_oracle_expr(e.lowered())
    if isinstance(e, X.DatePartExpr):
        return ("date_part", e.part, to_oracle_expr(e.arg))
    if isinstance(e, X.CaseExpr):
        tail = None if e.else_expr is None else to_oracle_expr(e.else_expr)
        for w, t in reversed(e.when_then):
            tail = ("case", to_oracle_expr(w), to_oracle_expr(t), tail)
        return tail
    raise TypeError(e)


def random_table(rng, n, spec, null_frac=0.0):
    """spec: {name: (pa type, low, high)}; uniform ints (decimals as unscaled ints)"""
    import numpy as np
    from decimal import Decimal
    cols = {}
    for name, (typ, lo, hi) in spec.items():
        vals = rng.integers(lo, hi, size=n)
        mask = rng.random(n) < null_frac if null_frac > 0 else None
        if pa.types.is_decimal128(typ) and mask is None and n > 100_000:
            from datafusion_amd.tpch import _decimal_from_int64
            cols[name] = _decimal_from_int64(vals.astype(np.int64), typ)
        elif pa.types.is_decimal128(typ):
            py = [Decimal(int(v)).scaleb(-typ.scale) for v in vals]
            cols[name] = pa.array(py, type=typ, mask=mask)
        elif pa.types.is_float64(typ):
            cols[name] = pa.array(vals.astype(np.float64) / 7.0, type=typ, mask=mask)
        elif pa.types.is_date32(typ):
            cols[name] = pa.array(vals.astype(np.int32), type=pa.int32(), mask=mask).cast(pa.date32())
        else:
            cols[name] = pa.array(vals, type=typ, mask=mask)
    return pa.table(cols)
