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
    cols = [c.to_pylist() for c in table.columns]
    return [tuple(c[i] for c in cols) for i in range(table.num_rows)]


def _key(row):
    return tuple((v is None, 0 if v is None else v) for v in row)


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
        assert rows(actual) == rows(expected)
    else:
        assert sorted_rows(actual) == sorted_rows(expected)
