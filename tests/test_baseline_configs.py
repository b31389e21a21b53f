"""BASELINE.json's configs as parity cases.  configs[1]-[4] live in the operator / query / full-size tests;
this file holds configs[0] — `SELECT l_returnflag, SUM(l_extendedprice) FROM lineitem GROUP BY 1` at SF1, the CPU plumbing
case (no GPU): the oracle's AggregateExec over dbgen-exact SF1 lineitem (6,001,215 rows) against Arrow Acero's hash
aggregate on the same table, an independent production CPU engine (the reference itself cannot run here) — and
the same query through the C ABI on the GPU box."""
import time
from decimal import Decimal

import pyarrow as pa
import pytest


def _lineitem_sf1(strings):
    from oracle import dbgen
    _, _, l = dbgen.tables(1, strings)
    return l.select(["l_returnflag", "l_extendedprice"])


def _acero(l: pa.Table):
    flag = l.column("l_returnflag")
    if pa.types.is_dictionary(flag.type):
        l = l.set_column(0, "l_returnflag", flag.cast(pa.string()))
    g = l.group_by("l_returnflag").aggregate([("l_extendedprice", "sum")]).sort_by("l_returnflag")
    key = lambda k: chr(k) if isinstance(k, int) else str(k)   # noqa: E731  (the "codes" layout holds the ASCII byte)
    return {key(k): Decimal(str(v)) for k, v in zip(g.column("l_returnflag").to_pylist(), g.column("l_extendedprice_sum").to_pylist())}


def test_config0_oracle_group_by_returnflag_sf1():
    from oracle import oracle
    l = _lineitem_sf1("codes")
    assert l.num_rows == 6_001_215
    t0 = time.perf_counter()
    got = oracle.aggregate(l, [(("col", "l_returnflag"), "l_returnflag")], [("sum", ("col", "l_extendedprice"), "sum(l_extendedprice)")], "Single")
    dt = time.perf_counter() - t0
    assert got.schema.field("sum(l_extendedprice)").type == pa.decimal128(25, 2)          # SUM(Decimal128(15,2)), sum.rs:247-250
    mine = {chr(k): v for k, v in zip(got.column("l_returnflag").to_pylist(), got.column("sum(l_extendedprice)").to_pylist())}
    assert mine == _acero(l) and sorted(mine) == ["A", "N", "R"]
    print(f"config 0: oracle {l.num_rows / dt / 1e6:.1f} M rows/s (1 thread)")


@pytest.mark.gpu
def test_config0_gpu_group_by_returnflag_sf1():
    from datafusion_amd import ops
    from datafusion_amd.expr import col
    from datafusion_amd.table import DeviceTable
    l = _lineitem_sf1("dictionary")
    got = ops.aggregate(DeviceTable.from_arrow(l), [(col("l_returnflag"), "l_returnflag")], [("sum", col("l_extendedprice"), "sum(l_extendedprice)")], "Single").to_arrow()
    assert got.schema.field("sum(l_extendedprice)").type == pa.decimal128(25, 2)
    mine = dict(zip(got.column("l_returnflag").cast(pa.string()).to_pylist(), got.column("sum(l_extendedprice)").to_pylist()))
    assert mine == _acero(l)
