"""Bit-identical numpy mirror of the device generator in csrc/tpch.hip.

Used by the CPU legs (oracle parity on small scale factors, bench.py's cpu_baseline sample):
same counter-based PRNG, so table(sf)[a:b] on the host equals the device table generated for
the same order range.  Money columns are Decimal128(15,2) (or Float64 when float_money).
"""
from __future__ import annotations

import numpy as np
import pyarrow as pa

SEED_BASE = 0xDF55
T_CUSTOMER, T_ORDERS, T_LINEITEM = 1, 2, 3
DATE_START, DATE_END, DATE_CUTOFF = 8035, 10440, 9298
SEGMENTS = ["AUTOMOBILE", "BUILDING", "FURNITURE", "HOUSEHOLD", "MACHINERY"]
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fmix64(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint64, copy=True)
    with np.errstate(over="ignore"):
        x ^= x >> np.uint64(33)
        x *= np.uint64(0xff51afd7ed558ccd)
        x ^= x >> np.uint64(33)
        x *= np.uint64(0xc4ceb9fe1a85ec53)
        x ^= x >> np.uint64(33)
    return x


def hash_u64(v: np.ndarray, seed: int) -> np.ndarray:
    return _fmix64(v.astype(np.uint64) ^ np.uint64(seed) ^ np.uint64(0x9E3779B97F4A7C15))


def rnd(table: int, row: np.ndarray, stream: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        return hash_u64(row.astype(np.uint64) * np.uint64(16) + np.uint64(stream), SEED_BASE + table)


def n_orders(sf: float) -> int:
    return int(round(1500000.0 * sf))


def n_customers(sf: float) -> int:
    return max(1, int(round(150000.0 * sf)))


def n_parts(sf: float) -> int:
    return max(1, int(round(200000.0 * sf)))


def order_key(i: np.ndarray) -> np.ndarray:
    return (i >> 3) * 32 + (i & 7) + 1


def order_date(i: np.ndarray) -> np.ndarray:
    return (DATE_START + (rnd(T_ORDERS, i, 1) % np.uint64(DATE_END - DATE_START + 1)).astype(np.int64)).astype(np.int32)


def line_count(i: np.ndarray) -> np.ndarray:
    return (1 + (rnd(T_ORDERS, i, 2) % np.uint64(7))).astype(np.int64)


def orders(sf: float, begin=0, end=None) -> pa.Table:
    end = n_orders(sf) if end is None or end < 0 else min(end, n_orders(sf))
    i = np.arange(begin, end, dtype=np.int64)
    c = 1 + (rnd(T_ORDERS, i, 0) % np.uint64(n_customers(sf))).astype(np.int64)
    c = np.where(c % 3 == 0, np.where(c > 1, c - 1, c + 1), c)
    return pa.table({
        "o_orderkey": pa.array(order_key(i), type=pa.int64()),
        "o_custkey": pa.array(c, type=pa.int64()),
        "o_orderdate": pa.array(order_date(i), type=pa.int32()).cast(pa.date32()),
        "o_shippriority": pa.array(np.zeros(len(i), np.int32), type=pa.int32()),
    })


def _decimal_from_int64(v: np.ndarray, typ=pa.decimal128(15, 2)) -> pa.Array:
    buf = np.zeros((len(v), 2), np.int64)
    buf[:, 0] = v
    buf[:, 1] = v >> 63  # sign extension
    return pa.Array.from_buffers(typ, len(v), [None, pa.py_buffer(buf.tobytes())])


def lineitem(sf: float, begin=0, end=None, float_money=False) -> pa.Table:
    end = n_orders(sf) if end is None or end < 0 else min(end, n_orders(sf))
    i = np.arange(begin, end, dtype=np.int64)
    cnt = line_count(i)
    oi = np.repeat(i, cnt)                                   # order index of each line
    starts = np.cumsum(cnt) - cnt
    j = np.arange(len(oi), dtype=np.int64) - np.repeat(starts, cnt)
    line = oi.astype(np.uint64) * np.uint64(8) + j.astype(np.uint64)
    qty = 1 + (rnd(T_LINEITEM, line, 0) % np.uint64(50)).astype(np.int64)
    part = 1 + (rnd(T_LINEITEM, line, 1) % np.uint64(n_parts(sf))).astype(np.int64)
    price = 90000 + ((part // 10) % 20001) + 100 * (part % 1000)
    disc = (rnd(T_LINEITEM, line, 2) % np.uint64(11)).astype(np.int64)
    tax = (rnd(T_LINEITEM, line, 3) % np.uint64(9)).astype(np.int64)
    ship = order_date(oi).astype(np.int64) + 1 + (rnd(T_LINEITEM, line, 4) % np.uint64(121)).astype(np.int64)
    receipt = ship + 1 + (rnd(T_LINEITEM, line, 5) % np.uint64(30)).astype(np.int64)
    ra = np.where((rnd(T_LINEITEM, line, 6) & np.uint64(1)) == 1, ord("R"), ord("A"))
    flag = np.where(receipt <= DATE_CUTOFF, ra, ord("N")).astype(np.uint8)
    status = np.where(ship > DATE_CUTOFF, ord("O"), ord("F")).astype(np.uint8)

    def money(cents):
        return pa.array(cents / 100.0, type=pa.float64()) if float_money else _decimal_from_int64(cents)

    return pa.table({
        "l_orderkey": pa.array(order_key(oi), type=pa.int64()),
        "l_quantity": money(qty * 100),
        "l_extendedprice": money(qty * price),
        "l_discount": money(disc),
        "l_tax": money(tax),
        "l_returnflag": pa.array(flag, type=pa.uint8()),
        "l_linestatus": pa.array(status, type=pa.uint8()),
        "l_shipdate": pa.array(ship.astype(np.int32), type=pa.int32()).cast(pa.date32()),
    })


def customer(sf: float, begin=0, end=None) -> pa.Table:
    end = n_customers(sf) if end is None or end < 0 else min(end, n_customers(sf))
    i = np.arange(begin, end, dtype=np.int64)
    return pa.table({
        "c_custkey": pa.array(i + 1, type=pa.int64()),
        "c_mktsegment": pa.array((rnd(T_CUSTOMER, i, 0) % np.uint64(5)).astype(np.uint8), type=pa.uint8()),
    })
