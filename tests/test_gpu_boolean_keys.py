"""Boolean key columns in HashJoinExec, RepartitionExec(Hash) and SortExec (round 4): the reference hashes and compares a BooleanArray
like any other primitive (common/src/hash_utils.rs:306-345 hash_array; group_values/single_group_by/boolean.rs for grouping); on the
device a Boolean key column is widened to one byte per row on entry and the Boolean column itself travels as payload.  The oracle
has no bit-packed columns: its expectation is computed over the same tables with the Boolean columns cast to UInt8."""
import numpy as np
import pyarrow as pa
import pyarrow.compute as pc
import pytest

from tests.util import assert_tables_equal, sorted_rows

pytestmark = pytest.mark.gpu

ALL_TYPES = ["Inner", "Left", "Right", "Full", "LeftSemi", "RightSemi", "LeftAnti", "RightAnti", "LeftMark", "RightMark"]


def _as_u8(table):
    cols = [c.cast(pa.uint8()) if pa.types.is_boolean(c.type) else c for c in table.columns]
    return pa.Table.from_arrays(cols, names=table.column_names)


def _tables(rng, nl, nr, null_frac):
    def make(n, prefix):
        b = pa.array(rng.random(n) < 0.4, mask=rng.random(n) < null_frac)
        k = pa.array(rng.integers(0, 40, n).astype(np.int32), mask=rng.random(n) < null_frac)
        return pa.table({prefix + "b": b, prefix + "k": k, prefix + "v": pa.array(rng.integers(0, 10**6, n))})
    return make(nl, "l"), make(nr, "r")


@pytest.mark.parametrize("mode", [0, 1, 4, 5])
@pytest.mark.parametrize("join_type", ALL_TYPES)
def test_join_on_a_boolean_key_column(join_type, mode):
    from datafusion_amd import ops
    from datafusion_amd.table import DeviceTable
    from oracle import oracle
    rng = np.random.default_rng(7 + mode)
    left, right = _tables(rng, 700, 1900, 0.05)
    on = [("lb", "rb"), ("lk", "rk")]
    for ne in ("NullEqualsNothing", "NullEqualsNull"):
        got = ops.hash_join(DeviceTable.from_arrow(left), DeviceTable.from_arrow(right), on, join_type, ne, table_mode=mode).to_arrow()
        assert [f.type for f in got.schema][:1] == [pa.bool_()] or join_type in ("RightSemi", "RightAnti", "RightMark")   # the Boolean column comes out as Boolean
        exp = oracle.hash_join(_as_u8(left), _as_u8(right), on, join_type, ne)
        assert sorted_rows(_as_u8(got)) == sorted_rows(exp), (join_type, ne)
    # the Boolean column alone as the key (two key values and NULL: long chains of equal keys)
    got = ops.hash_join(DeviceTable.from_arrow(left.slice(0, 60)), DeviceTable.from_arrow(right.slice(0, 90)), [("lb", "rb")], join_type, table_mode=mode).to_arrow()
    exp = oracle.hash_join(_as_u8(left.slice(0, 60)), _as_u8(right.slice(0, 90)), [("lb", "rb")], join_type)
    assert sorted_rows(_as_u8(got)) == sorted_rows(exp)


def test_hash_repartition_on_a_boolean_key_column():
    from datafusion_amd import ops
    from datafusion_amd.table import DeviceTable
    rng = np.random.default_rng(3)
    t, _ = _tables(rng, 50_000, 1, 0.03)
    parts = [p.to_arrow() for p in ops.partition(DeviceTable.from_arrow(t), ["lb", "lk"], 8)]
    same = [p.to_arrow() for p in ops.partition(DeviceTable.from_arrow(_as_u8(t)), ["lb", "lk"], 8)]
    assert sum(p.num_rows for p in parts) == t.num_rows
    for p, q in zip(parts, same):
        assert p.schema.field("lb").type == pa.bool_() and p.column_names == t.column_names
        assert _as_u8(p).equals(q)          # the same routing, row order kept inside every partition


def test_sort_on_boolean_key_columns():
    from datafusion_amd import ops
    from datafusion_amd.table import DeviceTable
    rng = np.random.default_rng(5)
    t, _ = _tables(rng, 30_000, 1, 0.04)
    t = t.append_column("i", pa.array(np.arange(t.num_rows)))
    dt = DeviceTable.from_arrow(t)
    for desc, nulls_first in ((False, False), (True, True), (True, False)):
        got = ops.sort(dt, [("lb", desc, nulls_first), ("lk", False, False), ("i", False, False)]).to_arrow()
        assert got.schema.field("lb").type == pa.bool_()
        # host ordering: (null rank, value) of lb, then lk (NULLS LAST), then the row number
        lb, lk = t.column("lb").to_pylist(), t.column("lk").to_pylist()

        def key(i):
            b = lb[i]
            first = (0 if nulls_first else 1) if b is None else (1 if nulls_first else 0)
            val = 0 if b is None else ((1 - int(b)) if desc else int(b))
            return (first, val, lk[i] is None, 0 if lk[i] is None else lk[i], i)
        order = sorted(range(t.num_rows), key=key)
        assert got.column("i").to_pylist() == order
    top = ops.sort(dt, [("lb", True, False), ("i", False, False)], fetch=7).to_arrow()
    trues = [i for i, b in enumerate(t.column("lb").to_pylist()) if b]
    assert top.column("i").to_pylist() == trues[:7]
