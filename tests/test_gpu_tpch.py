"""the device generator and its numpy mirror must agree bit for bit (any order range)"""
import pytest

from tests.util import assert_tables_equal

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rng", [(0, -1), (1000, 5000), (7, 8)])
def test_generator_matches_numpy_mirror(rng):
    from datafusion_amd import ops, tpch
    sf = 0.01
    b, e = rng
    assert_tables_equal(ops.tpch_orders(sf, b, e).to_arrow(), tpch.orders(sf, b, e), ordered=True)
    assert_tables_equal(ops.tpch_lineitem(sf, b, e).to_arrow(), tpch.lineitem(sf, b, e), ordered=True)
    assert_tables_equal(ops.tpch_lineitem(sf, b, e, float_money=True).to_arrow(), tpch.lineitem(sf, b, e, float_money=True), ordered=True)
    assert_tables_equal(ops.tpch_customer(sf).to_arrow(), tpch.customer(sf), ordered=True)


def test_arrow_roundtrip():
    import pyarrow as pa
    from datafusion_amd.table import DeviceTable
    t = pa.table({"a": pa.array([1, None, 3], type=pa.int32()), "b": pa.array([True, False, None]),
                  "c": pa.array([1.5, 2.5, None]), "d": pa.array([None, 2, 3], type=pa.decimal128(15, 2))})
    back = DeviceTable.from_arrow(t).to_arrow()
    assert back.to_pylist() == t.to_pylist()
    s = pa.table({"a": pa.array(range(100), type=pa.int64())}).slice(13, 50)   # non-zero offset
    assert DeviceTable.from_arrow(s).to_arrow().column("a").to_pylist() == list(range(13, 63))
