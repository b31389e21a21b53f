"""TEST INFRASTRUCTURE (checker data, never on the product path): dbgen-exact TPC-H columns.

The reference pins TPC-H *answers* at scale factor 0.1
(`datafusion/sqllogictest/test_files/tpch/answers/q1.slt.part:42-45`, `q3.slt.part:44-53`) over data
written by the TPC's `dbgen` (the reference's harness uses the `tpchgen` crate, a bit-exact Rust
restatement of it — un-vendored, `benchmarks/src/tpch/run.rs:53-58`).  Neither generator is in this
image, so this module restates the published dbgen v2.17 algorithm for exactly the columns Q1 and Q3
read, vectorised with numpy:

  * one Lehmer stream per column:  x' = 16807 * x mod (2^31 - 1)                (rnd.c NextRand)
  * UnifInt(lo, hi) = lo + (long)((double)x' / 2147483647.0 * (double)(hi - lo + 1))   (rnd.c)
  * after every row each stream is advanced to its per-row `boundary` (rnd.c row_stop), so row i
    of a stream starts at seed * 16807^(boundary * i): that is what makes it vectorisable
  * sparse order keys (build.c mk_sparse), retail price (rpb_routine), date offsets and the
    returnflag / linestatus rules of build.c mk_order, market segment of mk_cust.

It is pinned twice (tests/test_dbgen_golden.py): against the first rows of dbgen's SF1 output that
the reference carries as fixtures (`datafusion/core/tests/tpch-csv/{lineitem,orders,customer}.csv`,
copied by `tests/golden/make_tpch_answers_golden.py`), and — end to end — by the oracle's Q1 / Q3
over `table(0.1)` reproducing the reference's pinned answers digit for digit.
"""
from __future__ import annotations

import ctypes
import functools
import os

import numpy as np
import pyarrow as pa

M = 2147483647
A = 16807
# driver.c Seed[]: (initial value, per-row boundary)
O_ODATE_SD = (1066728069, 1)
O_CKEY_SD = (851767375, 1)
O_LCNT_SD = (1434868289, 1)
L_QTY_SD = (209208115, 7)
L_DCNT_SD = (554590007, 7)
L_TAX_SD = (721958466, 7)
L_PKEY_SD = (1808217256, 7)
L_SDTE_SD = (1769349045, 7)
L_CDTE_SD = (904914315, 7)
L_RDTE_SD = (373135028, 7)
L_RFLG_SD = (717419739, 7)
C_MSEG_SD = (1140279430, 1)
O_PRIO_SD = (591449447, 1)
P_MFG_SD = (1, 1)
P_BRND_SD = (46831694, 1)
P_TYPE_SD = (1841581359, 1)
P_SIZE_SD = (1193163244, 1)
P_CNTR_SD = (727633698, 1)
C_NTRG_SD = (1489529863, 1)
S_NTRG_SD = (110356601, 1)
L_SHIP_SD = (1371272478, 7)
L_SMODE_SD = (675466456, 7)
L_SKEY_SD = (2095021727, 7)
S_ADDR_SD = (706178559, 9)      # one length draw + up to eight character draws (a_rnd: a draw per five characters)
S_PHNE_SD = (884434366, 3)
C_ADDR_SD = (881155353, 9)
C_PHNE_SD = (1521138112, 3)     # three draws per customer: area code, exchange, number (bm_utils.c gen_phone)
C_ABAL_SD = (298370230, 1)
P_NAME_SD = (709314158, 92)     # a permutation of the 92 colours per part (permute.c: one draw per position)
PS_QTY_SD = (1671059989, 4)     # driver.c seed table: PSUPP streams advance SUPP_PER_PART draws per part row
PS_SCST_SD = (1051288424, 4)
S_ABAL_SD = (962338209, 1)
BBB_TYPE_SD = (753643799, 1)    # driver.c seed table, streams 45 / 46: mk_supp draws both for EVERY supplier (one draw per row)
BBB_CMNT_SD = (202794285, 1)

STARTDATE_EPOCH = 8035          # 1992-01-01 as days since 1970-01-01 (dss.h STARTDATE 92001)
O_ODATE_SPAN = 2557 - (121 + 30) - 1   # O_ODATE_MAX - O_ODATE_MIN (dss.h TOTDATE, L_SDTE_MAX, L_RDTE_MAX)
CURRENTDATE_EPOCH = 9298        # 1995-06-17 (dss.h CURRENTDATE 95168)
CUST_MORTALITY = 3
DBGEN_SEGMENTS = ["AUTOMOBILE", "BUILDING", "FURNITURE", "HOUSEHOLD", "MACHINERY"]   # dists.dss msegmnt
PRIORITIES = ["1-URGENT", "2-HIGH", "3-MEDIUM", "4-NOT SPECIFIED", "5-LOW"]                # dists.dss o_oprio
SHIP_MODES = ["REG AIR", "AIR", "RAIL", "TRUCK", "MAIL", "FOB", "SHIP"]                   # dists.dss smode
SHIP_INSTRUCT = ["DELIVER IN PERSON", "COLLECT COD", "TAKE BACK RETURN", "NONE"]          # dists.dss instruct
SUPP_PER_PART = 4
# dists.dss p_cntr: 5 x 8 syllable combinations
CONTAINERS = [a + " " + b for a in ("SM", "LG", "MED", "JUMBO", "WRAP") for b in ("CASE", "BOX", "BAG", "JAR", "PKG", "PACK", "CAN", "DRUM")]
# dists.dss nations (name, region) and regions
NATIONS = [("ALGERIA", 0), ("ARGENTINA", 1), ("BRAZIL", 1), ("CANADA", 1), ("EGYPT", 4), ("ETHIOPIA", 0), ("FRANCE", 3), ("GERMANY", 3),
           ("INDIA", 2), ("INDONESIA", 2), ("IRAN", 4), ("IRAQ", 4), ("JAPAN", 2), ("JORDAN", 4), ("KENYA", 0), ("MOROCCO", 0),
           ("MOZAMBIQUE", 0), ("PERU", 1), ("CHINA", 2), ("ROMANIA", 3), ("SAUDI ARABIA", 4), ("VIETNAM", 2), ("RUSSIA", 3),
           ("UNITED KINGDOM", 3), ("UNITED STATES", 1)]
REGIONS = ["AFRICA", "AMERICA", "ASIA", "EUROPE", "MIDDLE EAST"]


def _powers(base: int, n: int) -> np.ndarray:
    """[base^0, base^1, ..., base^(n-1)] mod M, built by doubling (products stay below 2^62)"""
    out = np.empty(max(n, 1), dtype=np.uint64)
    out[0] = 1
    have, step = 1, base % M
    while have < n:
        take = min(have, n - have)
        out[have:have + take] = out[:take] * np.uint64(step) % np.uint64(M)
        have += take
        step = step * step % M
    return out[:n]


def _row_starts(sd, n_rows: int) -> np.ndarray:
    """stream state at the start of rows 0..n_rows-1 (row_stop advances every stream to its boundary)"""
    seed, boundary = sd
    return _powers(pow(A, boundary, M), n_rows) * np.uint64(seed) % np.uint64(M)


def _unif(state: np.ndarray, lo: int, hi: int) -> np.ndarray:
    """UnifInt on already-advanced states: the double arithmetic is the same two IEEE operations as rnd.c"""
    return lo + (state.astype(np.float64) / 2147483647.0 * float(hi - lo + 1)).astype(np.int64)


def _draw(sd, n_rows: int, lo: int, hi: int) -> np.ndarray:
    """one draw per row from a boundary-1 stream"""
    return _unif(_row_starts(sd, n_rows) * np.uint64(A) % np.uint64(M), lo, hi)


def _draw_lines(sd, order_of_line: np.ndarray, call_in_order: np.ndarray, n_orders: int, lo: int, hi: int) -> np.ndarray:
    """draw number `call_in_order` (0-based) of each line's order from a boundary-7 stream"""
    starts = _row_starts(sd, n_orders)
    apow = _powers(A, max(8, sd[1] + 1))
    return _unif(starts[order_of_line] * apow[call_in_order + 1] % np.uint64(M), lo, hi)


def counts(sf: float):
    """dbgen scales the table bases for sf < 1 with integer arithmetic (driver.c: base * int(1000 sf) / 1000)"""
    if sf < 1:
        k = int(1000 * sf)
        return dict(customer=150000 * k // 1000, orders=1500000 * k // 1000, part=200000 * k // 1000)
    k = int(sf)
    return dict(customer=150000 * k, orders=1500000 * k, part=200000 * k)


def order_keys(n: int) -> np.ndarray:
    i = np.arange(1, n + 1, dtype=np.int64)        # build.c mk_sparse: SPARSE_BITS 2, SPARSE_KEEP 3
    return ((i >> 3) << 5) + (i & 7)


def retail_price(partkey: np.ndarray) -> np.ndarray:
    """build.c rpb_routine, in cents"""
    return 90000 + (partkey // 10) % 20001 + (partkey % 1000) * 100


def _decimal(cents: np.ndarray, precision: int = 15, scale: int = 2) -> pa.Array:
    v = cents.astype(np.int64)
    buf = np.empty((len(v), 2), dtype=np.int64)
    buf[:, 0] = v
    buf[:, 1] = v >> 63
    return pa.Array.from_buffers(pa.decimal128(precision, scale), len(v), [None, pa.py_buffer(buf.tobytes())])


def _strings(codes: np.ndarray, names, how: str) -> pa.Array:
    """a pick_str column: `codes` index `names` (the distribution's own order)"""
    if how == "utf8":
        return pa.array(np.array(names, dtype=object)[codes], pa.string())
    order = sorted(names)                              # ascending dictionary: ORDER BY on the indices is ORDER BY on the strings
    remap = np.array([order.index(s) for s in names], dtype=np.uint8)
    if how == "dictionary":
        return pa.DictionaryArray.from_arrays(pa.array(remap[codes], pa.uint8()), pa.array(order, pa.string()))
    return pa.array(remap[codes], pa.uint8())          # "codes": index into sorted(names)


def _flag(ascii_codes: np.ndarray, how: str) -> pa.Array:
    """a one-character column held as its ASCII byte"""
    if how == "codes":
        return pa.array(ascii_codes, pa.uint8())
    letters = sorted(set(int(x) for x in np.unique(ascii_codes)))
    return _strings(np.searchsorted(letters, ascii_codes), [chr(v) for v in letters], how)


def tables(sf: float, strings: str = "codes"):
    """(customer, orders, lineitem) restricted to the columns the pinned queries read, typed as
    `benchmarks/src/tpch/mod.rs:93-122`.  strings: "codes" = the device generator's layout
    (l_returnflag / l_linestatus as their ASCII byte, the other string columns as UInt8 indices into the
    sorted value list), "dictionary" = Arrow Dictionary(UInt8, Utf8) with ascending dictionaries,
    "utf8" = plain strings (CPU checks only)."""
    n = counts(sf)
    nc, no, npart = n["customer"], n["orders"], n["part"]
    nsupp = npart // 20
    # ---- customer (build.c mk_cust)
    seg = _draw(C_MSEG_SD, nc, 1, 5) - 1              # pick_str over five weight-1 entries
    names = [f"Customer#{i:09d}" for i in range(1, nc + 1)]        # build.c mk_cust: C_NAME_FMT "%s%09ld"
    if strings == "utf8":
        c_name = pa.array(names, pa.string())
    elif strings == "dictionary":
        c_name = pa.DictionaryArray.from_arrays(pa.array(np.arange(nc, dtype=np.int32)), pa.array(names, pa.string()))
    else:
        c_name = pa.array(np.arange(nc, dtype=np.int32))
    c_nation = _draw(C_NTRG_SD, nc, 0, 24)
    customer = pa.table({"c_custkey": pa.array(np.arange(1, nc + 1, dtype=np.int64)), "c_name": c_name, "c_nationkey": pa.array(c_nation),
                         "c_mktsegment": _strings(seg, DBGEN_SEGMENTS, strings)})
    if strings != "codes":
        # bm_utils.c gen_phone: PHONE_FMT "%02d-%03d-%03d-%04d" = country code 10 + nation, then three draws of C_PHNE_SD;
        # build.c mk_cust: c_acctbal = RANDOM(-99999, 999999) cents.  (Q22 reads the country code and the balance; its answer pins
        # the balance stream to the cent.)
        c_phone = _string_column(_phones(C_PHNE_SD, c_nation), strings)
        customer = customer.append_column("c_address", _string_column(_v_strings(C_ADDR_SD, nc, 25), strings))
        customer = customer.append_column("c_phone", c_phone).append_column("c_acctbal", _decimal(_draw(C_ABAL_SD, nc, -99999, 999999)))
        customer = customer.append_column("c_comment", _string_column(text_column(C_CMNT_SD, nc, 73), strings))     # mk_cust: TEXT(C_CMNT_LEN = 73)
    # ---- orders (build.c mk_order)
    okey = order_keys(no)
    ckey = _draw(O_CKEY_SD, no, 1, nc)
    dead = ckey % CUST_MORTALITY == 0                 # "while (custkey % 3 == 0) custkey += delta": one step of +1
    ckey = np.where(dead, np.minimum(ckey + 1, nc), ckey)
    dead = ckey % CUST_MORTALITY == 0                 # (only when the +1 was clamped at the maximum: delta flips to -1)
    ckey = np.where(dead, ckey - 1, ckey)
    odate = _draw(O_ODATE_SD, no, 0, O_ODATE_SPAN)    # days after 1992-01-01
    prio = _draw(O_PRIO_SD, no, 1, 5) - 1
    lines = _draw(O_LCNT_SD, no, 1, 7)
    # ---- lineitem
    oi = np.repeat(np.arange(no, dtype=np.int64), lines)
    first = np.cumsum(lines) - lines
    k = np.arange(len(oi), dtype=np.int64) - first[oi]      # line number - 1 = draw index in the order's 7-draw window
    qty = _draw_lines(L_QTY_SD, oi, k, no, 1, 50)
    disc = _draw_lines(L_DCNT_SD, oi, k, no, 0, 10)
    tax = _draw_lines(L_TAX_SD, oi, k, no, 0, 8)
    instruct = _draw_lines(L_SHIP_SD, oi, k, no, 1, 4) - 1
    smode = _draw_lines(L_SMODE_SD, oi, k, no, 1, 7) - 1
    pkey = _draw_lines(L_PKEY_SD, oi, k, no, 1, npart)
    snum = _draw_lines(L_SKEY_SD, oi, k, no, 0, 3)
    skey = (pkey + snum * (nsupp // SUPP_PER_PART + (pkey - 1) // nsupp)) % nsupp + 1     # dss.h PART_SUPP_BRIDGE
    sdate = odate[oi] + _draw_lines(L_SDTE_SD, oi, k, no, 1, 121)
    cdate = odate[oi] + _draw_lines(L_CDTE_SD, oi, k, no, 30, 90)
    rdate = sdate + _draw_lines(L_RDTE_SD, oi, k, no, 1, 30)
    returned = rdate + STARTDATE_EPOCH <= CURRENTDATE_EPOCH
    # L_RFLG_SD is only drawn for received lines: the draw index is the count of earlier received lines of the order
    before = np.cumsum(returned) - returned
    r_idx = before - before[first[oi]]
    pick = _draw_lines(L_RFLG_SD, oi, r_idx, no, 1, 2)
    rflag = np.where(returned, np.where(pick == 1, ord("R"), ord("A")), ord("N")).astype(np.uint8)
    lstatus = np.where(sdate + STARTDATE_EPOCH <= CURRENTDATE_EPOCH, ord("F"), ord("O")).astype(np.uint8)
    eprice = retail_price(pkey) * qty
    # o_totalprice += ((eprice * (100 - discount)) / PENNIES) * (100 + tax) / PENNIES, integer division (build.c mk_order)
    per_line = (eprice * (100 - disc)) // 100 * (100 + tax) // 100
    total = np.add.reduceat(per_line, first) if len(per_line) else np.zeros(0, dtype=np.int64)
    # o_orderstatus: 'F' when every line has shipped by CURRENTDATE, 'O' when none has, else 'P' (build.c mk_order)
    shipped = (lstatus == ord("F")).astype(np.int64)
    n_shipped = np.add.reduceat(shipped, first) if len(shipped) else np.zeros(0, dtype=np.int64)
    ostatus = np.where(n_shipped == lines, ord("F"), np.where(n_shipped == 0, ord("O"), ord("P"))).astype(np.uint8)
    orders = pa.table({"o_orderkey": pa.array(okey), "o_custkey": pa.array(ckey), "o_orderstatus": _flag(ostatus, strings), "o_totalprice": _decimal(total),
                       "o_orderdate": pa.array((odate + STARTDATE_EPOCH).astype(np.int32), pa.date32()),
                       "o_orderpriority": _strings(prio, PRIORITIES, strings),
                       "o_shippriority": pa.array(np.zeros(no, dtype=np.int32))})
    if strings != "codes":
        orders = orders.append_column("o_comment", _string_column(text_column(O_CMNT_SD, no, 49), strings))    # mk_order: TEXT(O_CMNT_LEN = 49)
    lineitem = pa.table({
        "l_orderkey": pa.array(okey[oi]), "l_partkey": pa.array(pkey), "l_suppkey": pa.array(skey), "l_linenumber": pa.array((k + 1).astype(np.int32)),
        "l_quantity": _decimal(qty * 100), "l_extendedprice": _decimal(eprice),
        "l_discount": _decimal(disc), "l_tax": _decimal(tax), "l_returnflag": _flag(rflag, strings), "l_linestatus": _flag(lstatus, strings),
        "l_shipdate": pa.array((sdate + STARTDATE_EPOCH).astype(np.int32), pa.date32()),
        "l_commitdate": pa.array((cdate + STARTDATE_EPOCH).astype(np.int32), pa.date32()),
        "l_receiptdate": pa.array((rdate + STARTDATE_EPOCH).astype(np.int32), pa.date32()),
        "l_shipinstruct": _strings(instruct, SHIP_INSTRUCT, strings), "l_shipmode": _strings(smode, SHIP_MODES, strings)})
    return customer, orders, lineitem


ALPHA_NUM = "0123456789abcdefghijklmnopqrstuvwxyz ABCDEFGHIJKLMNOPQRSTUVWXYZ,"   # bm_utils.c alpha_num (64 characters)


def _v_strings(sd, n_rows: int, avg_len: int) -> list:
    """bm_utils.c a_rnd behind V_STR(avg, sd): length = RANDOM(0.4 avg, 1.6 avg), then one RANDOM(0, MAX_LONG) per five
    characters, six bits per character from the low end.  MAX_LONG - 0 + 1 overflows dbgen's 32-bit range to -2^31, so the value
    the bits are taken from is the NEGATED draw — pinned by the five addresses the reference carries (core/tests/tpch-csv
    customer.csv rows 2 and 3, supplier.csv rows 1 and 8136, answers/q15.slt.part supplier 677): 129 characters."""
    lo, hi = int(avg_len * 0.4), int(avg_len * 1.6)
    rows = np.arange(n_rows)
    zero = np.zeros(n_rows, dtype=np.int64)
    length = _draw_lines(sd, rows, zero, n_rows, lo, hi)
    starts = _row_starts(sd, n_rows)
    apow = _powers(A, 10)
    groups = (hi + 4) // 5
    codes = np.zeros((n_rows, groups * 5), dtype=np.int64)
    for g in range(groups):
        state = (starts * apow[g + 2] % np.uint64(M)).astype(np.int64)       # draw g + 1 of the row (draw 0 is the length)
        v = (np.int64(1) << 31) - state
        for k in range(5):
            codes[:, g * 5 + k] = (v >> (6 * k)) & 63
    table = np.frombuffer(ALPHA_NUM.encode(), dtype=np.uint8)
    chars = table[codes]
    return [bytes(chars[i, :length[i]]).decode() for i in range(n_rows)]


def _phones(sd, nation: np.ndarray) -> list:
    """bm_utils.c gen_phone: PHONE_FMT "%02d-%03d-%03d-%04d" = country code 10 + nation, then three draws"""
    n = len(nation)
    rows, zero = np.arange(n), np.zeros(n, dtype=np.int64)
    area = _draw_lines(sd, rows, zero, n, 100, 999)
    exch = _draw_lines(sd, rows, zero + 1, n, 100, 999)
    numb = _draw_lines(sd, rows, zero + 2, n, 1000, 9999)
    return [f"{10 + int(c):02d}-{int(a):03d}-{int(e):03d}-{int(u):04d}" for c, a, e, u in zip(nation, area, exch, numb)]


def _string_column(values: list, how: str) -> pa.Array:
    """a free-text column: plain Utf8, or dictionary-encoded over the ascending distinct values"""
    if how == "utf8":
        return pa.array(values, pa.string())
    order = sorted(set(values))
    at = {v: i for i, v in enumerate(order)}
    return pa.DictionaryArray.from_arrays(pa.array(np.array([at[v] for v in values], dtype=np.int32)), pa.array(order, pa.string()))


def supplier(sf: float, strings: str = "codes") -> pa.Table:
    """build.c mk_supp: key, name (S_NAME_FMT "%s%09ld") and nation"""
    ns = counts(sf)["part"] // 20
    names = [f"Supplier#{i:09d}" for i in range(1, ns + 1)]
    if strings == "utf8":
        s_name = pa.array(names, pa.string())
    elif strings == "dictionary":
        s_name = pa.DictionaryArray.from_arrays(pa.array(np.arange(ns, dtype=np.int32)), pa.array(names, pa.string()))
    else:
        s_name = pa.array(np.arange(ns, dtype=np.int32))
    s_nation = _draw(S_NTRG_SD, ns, 0, 24)
    t = pa.table({"s_suppkey": pa.array(np.arange(1, ns + 1, dtype=np.int64)), "s_name": s_name, "s_nationkey": pa.array(s_nation)})
    if strings != "codes":     # mk_supp: V_STR(S_ADDR_LEN = 25, S_ADDR_SD), gen_phone(nation, S_PHNE_SD) — Q15 prints both
        t = t.append_column("s_address", _string_column(_v_strings(S_ADDR_SD, ns, 25), strings)).append_column("s_phone", _string_column(_phones(S_PHNE_SD, s_nation), strings))
        # s_acctbal = RANDOM(-99999, 999999) cents (Q2 prints it: pinned to the cent by its answer)
        t = t.append_column("s_acctbal", _decimal(_draw(S_ABAL_SD, ns, -99999, 999999)))
        t = t.append_column("s_comment", _string_column(supplier_comments(ns), strings))
    return t


# ------------------------------------------------------------------------------------------------------------ comment text
# dists.dss, the distributions of dbgen's text grammar (TPC-H specification clause 4.2.2.14): sentence forms, noun and verb
# phrases, and the weighted word lists — in dists.dss's own "word|weight" lines, including its misspelt "whithout".  Restated
# from the published file; pinned, together with oracle/dbgen_text.c, by every comment string the reference carries (165 strings
# at offsets spread over the whole pool: tests/test_dbgen_golden.py).
DISTS = """\
BEGIN grammar
N V T|3
N V P T|3
N V N T|3
N P V N T|1
N P V P T|1
END
BEGIN np
N|10
J N|20
J, J N|10
D J N|50
END
BEGIN vp
V|30
X V|1
V D|40
X V D|1
END
BEGIN nouns
packages|40
requests|40
accounts|40
deposits|40
foxes|20
ideas|20
theodolites|20
pinto beans|20
instructions|20
dependencies|10
excuses|10
platelets|10
asymptotes|10
courts|5
dolphins|5
multipliers|1
sauternes|1
warthogs|1
frets|1
dinos|1
attainments|1
somas|1
Tiresias|1
patterns|1
forges|1
braids|1
frays|1
warhorses|1
dugouts|1
notornis|1
epitaphs|1
pearls|1
tithes|1
waters|1
orbits|1
gifts|1
sheaves|1
depths|1
sentiments|1
decoys|1
realms|1
pains|1
grouches|1
escapades|1
hockey players|1
END
BEGIN verbs
sleep|20
wake|20
are|20
cajole|20
haggle|20
nag|10
use|10
boost|10
affix|5
detect|5
integrate|5
maintain|1
nod|1
was|1
lose|1
sublate|1
solve|1
thrash|1
promise|1
engage|1
hinder|1
print|1
x-ray|1
breach|1
eat|1
grow|1
impress|1
mold|1
poach|1
serve|1
run|1
dazzle|1
snooze|1
doze|1
unwind|1
kindle|1
play|1
hang|1
believe|1
doubt|1
END
BEGIN adjectives
special|20
pending|20
unusual|20
express|20
furious|1
sly|1
careful|1
blithe|1
quick|1
fluffy|1
slow|1
quiet|1
ruthless|1
thin|1
close|1
dogged|1
daring|1
brave|1
stealthy|1
permanent|1
enticing|1
idle|1
busy|1
regular|50
final|40
ironic|40
even|30
bold|20
silent|10
END
BEGIN adverbs
sometimes|1
always|1
never|1
furiously|50
slyly|50
carefully|50
blithely|40
quickly|30
fluffily|20
slowly|1
quietly|1
ruthlessly|1
thinly|1
closely|1
doggedly|1
daringly|1
bravely|1
stealthily|1
permanently|1
enticingly|1
idly|1
busily|1
regularly|1
finally|1
ironically|1
evenly|1
boldly|1
silently|1
END
BEGIN prepositions
about|50
above|50
according to|50
across|50
after|50
against|40
along|40
alongside of|30
among|30
around|20
at|10
atop|1
before|1
behind|1
beneath|1
beside|1
besides|1
between|1
beyond|1
by|1
despite|1
during|1
except|1
for|1
from|1
in place of|1
inside|1
instead of|1
into|1
near|1
of|1
on|1
outside|1
over|1
past|1
since|1
through|1
throughout|1
to|1
toward|1
under|1
until|1
up|1
upon|1
whithout|1
with|1
within|1
END
BEGIN auxillaries
do|1
may|1
might|1
shall|1
will|1
would|1
can|1
could|1
should|1
ought to|1
must|1
will have to|1
shall have to|1
could have to|1
should have to|1
must have to|1
need to|1
try to|1
END
BEGIN terminators
.|50
;|1
:|1
?|1
!|1
--|1
END
"""
TEXT_POOL_SIZE = 300 * 1024 * 1024     # dss.h TEXT_POOL_SIZE
N_CMNT_SD = (606179079, 2)             # driver.c seed table: text columns draw twice per row (offset into the pool, then length)
R_CMNT_SD = (1500869201, 2)
S_CMNT_SD = (1341315363, 2)
C_CMNT_SD = (1335826707, 2)
O_CMNT_SD = (276090261, 2)
P_CMNT_SD = (804159733, 2)
PS_CMNT_SD = (1961692154, 8)            # SUPP_PER_PART rows x 2 draws per part
BBB_JNK_SD = (263032577, 1)
BBB_OFFSET_SD = (715851524, 1)


@functools.lru_cache(maxsize=1)
def text_pool() -> bytes:
    """dbgen's 300 MiB text pool (text.c init_text_pool; restated in oracle/dbgen_text.c, ≈ 2 s)"""
    lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libdforacle.so"))
    out = ctypes.create_string_buffer(TEXT_POOL_SIZE + 4096)
    rc = lib.dbgen_text_pool(ctypes.create_string_buffer(DISTS.encode()), out, ctypes.c_int64(TEXT_POOL_SIZE))
    assert rc == 0, rc
    return out.raw[:TEXT_POOL_SIZE]


def text_column(sd, n_rows: int, avg_len: int, rows=None, call: int = 0) -> list:
    """dss.h TEXT(avg, sd, tgt) = text.c dbg_text(tgt, 0.4 avg, 1.6 avg, sd): offset = RANDOM(0, TEXT_POOL_SIZE - max), then
    length = RANDOM(min, max); the comment is that slice of the pool.  `rows` (default: all n_rows) selects rows of the table;
    `call` = the TEXT call's number within its row (partsupp: four rows of one part share a stream boundary)"""
    lo, hi = int(avg_len * 0.4), int(avg_len * 1.6)
    rows = np.arange(n_rows) if rows is None else np.asarray(rows, dtype=np.int64)
    calls = np.zeros(len(rows), dtype=np.int64) + 2 * np.asarray(call, dtype=np.int64)
    off = _draw_lines(sd, rows, calls, n_rows, 0, TEXT_POOL_SIZE - hi)
    length = _draw_lines(sd, rows, calls + 1, n_rows, lo, hi)
    pool = text_pool()
    return [pool[o:o + n].decode() for o, n in zip(off.tolist(), length.tolist())]


def supplier_complaints(ns: int):
    """mk_supp's Better Business Bureau marks: every supplier draws bad_press = RANDOM(1, 10000, BBB_CMNT_SD) and type =
    RANDOM(0, 100, BBB_TYPE_SD); bad_press <= S_CMNT_BBB (10) overwrites part of the comment with "Customer " … "Complaints"
    (type < BBB_DEADBEATS = 50) or "Customer " … "Recommends".  -> (complaints, recommends) Boolean arrays"""
    marked = _draw(BBB_CMNT_SD, ns, 1, 10000) <= 10
    deadbeat = _draw(BBB_TYPE_SD, ns, 0, 100) < 50
    return marked & deadbeat, marked & ~deadbeat


def supplier_comments(ns: int) -> list:
    """mk_supp: TEXT(S_CMNT_LEN = 63, S_CMNT_SD), then the Better Business Bureau marks of the suppliers supplier_complaints()
    selects: noise = RANDOM(0, len - 19, BBB_JNK_SD), offset = RANDOM(0, len - (19 + noise), BBB_OFFSET_SD) — drawn for every
    supplier — and "Customer " is written at `offset`, "Complaints" / "Recommends" `noise` characters after it.  (The text is
    pinned by the reference's supplier rows and Q2's answer; none of those carries a mark: the marks follow build.c as published.)"""
    text = text_column(S_CMNT_SD, ns, 63)
    bad, good = supplier_complaints(ns)
    length = np.array([len(t) for t in text], dtype=np.int64)
    state = lambda sd: _row_starts(sd, ns) * np.uint64(A) % np.uint64(M)      # noqa: E731 - one draw per row
    junk = state(BBB_JNK_SD).astype(np.float64) / 2147483647.0
    offs = state(BBB_OFFSET_SD).astype(np.float64) / 2147483647.0
    for i in np.nonzero(bad | good)[0]:
        noise = int(junk[i] * float(length[i] - 19 + 1))
        offset = int(offs[i] * float(length[i] - (19 + noise) + 1))
        t = bytearray(text[i].encode())
        t[offset:offset + 9] = b"Customer "
        t[offset + 9 + noise:offset + 9 + noise + 10] = b"Complaints" if bad[i] else b"Recommends"
        text[i] = t.decode()
    return text


def nation(strings: str = "codes") -> pa.Table:
    names = [n for n, _ in NATIONS]
    return pa.table({"n_nationkey": pa.array(np.arange(25, dtype=np.int64)), "n_name": _strings(np.arange(25), names, strings),
                     "n_regionkey": pa.array(np.array([r for _, r in NATIONS], dtype=np.int64))})


def region(strings: str = "codes") -> pa.Table:
    return pa.table({"r_regionkey": pa.array(np.arange(5, dtype=np.int64)), "r_name": _strings(np.arange(5), REGIONS, strings)})


# dists.dss p_types: 6 x 5 x 5 syllable combinations, weight 1 each, in this order (pick_str over the cumulative weights)
PART_TYPES = [a + " " + b + " " + c for a in ("STANDARD", "SMALL", "MEDIUM", "LARGE", "ECONOMY", "PROMO")
              for b in ("ANODIZED", "BURNISHED", "PLATED", "POLISHED", "BRUSHED") for c in ("TIN", "NICKEL", "BRASS", "STEEL", "COPPER")]


# dists.dss colors: the 92 words p_name is made of
COLORS = ("almond antique aquamarine azure beige bisque black blanched blue blush brown burlywood burnished chartreuse chiffon chocolate coral "
          "cornflower cornsilk cream cyan dark deep dim dodger drab firebrick floral forest frosted gainsboro ghost goldenrod green grey honeydew "
          "hot indian ivory khaki lace lavender lawn lemon light lime linen magenta maroon medium metallic midnight mint misty moccasin navajo "
          "navy olive orange orchid pale papaya peach peru pink plum powder puff purple red rose rosy royal saddle salmon sandy seashell sienna "
          "sky slate smoke snow spring steel tan thistle tomato turquoise violet wheat white yellow").split()


def part_names(n: int) -> list:
    """build.c mk_part agg_str(&colors, 5, P_NAME_SD): the colour list is permuted afresh for every part (permute.c: position i
    swaps with a position drawn from [i, 91]) and the first five words make the name — only the first five swaps matter.  Pinned by
    the name the reference's part.csv row carries (part 1: "goldenrod lavender spring chocolate lace") and by Q9 / Q20."""
    nc = len(COLORS)
    rows = np.arange(n)
    perm = np.tile(np.arange(nc, dtype=np.int64), (n, 1))
    for i in range(5):
        src = _draw_lines_wide(P_NAME_SD, rows, i, n, i, nc - 1)
        a, b = perm[rows, i].copy(), perm[rows, src].copy()
        perm[rows, i], perm[rows, src] = b, a
    words = np.array(COLORS, dtype=object)
    return [" ".join(words[perm[r, :5]]) for r in range(n)]


def _draw_lines_wide(sd, rows: np.ndarray, call: int, n_rows: int, lo: int, hi: int) -> np.ndarray:
    """draw number `call` (0-based) of every row from a stream with any boundary"""
    starts = _row_starts(sd, n_rows)
    return _unif(starts[rows] * np.uint64(pow(A, call + 1, M)) % np.uint64(M), lo, hi)


def part(sf: float, strings: str = "codes") -> pa.Table:
    """build.c mk_part: key, brand ("Brand#MN": M = manufacturer 1..5, N = 1..5), type, size 1..50, container"""
    n = counts(sf)["part"]
    mfgr = _draw(P_MFG_SD, n, 1, 5)
    brand = mfgr * 10 + _draw(P_BRND_SD, n, 1, 5)
    brands = [f"Brand#{m}{k}" for m in range(1, 6) for k in range(1, 6)]
    bcode = (brand // 10 - 1) * 5 + (brand % 10 - 1)
    size = _draw(P_SIZE_SD, n, 1, 50)
    cntr = _draw(P_CNTR_SD, n, 1, 40) - 1
    ptype = _draw(P_TYPE_SD, n, 1, len(PART_TYPES)) - 1
    mfgrs = [f"Manufacturer#{m}" for m in range(1, 6)]      # build.c mk_part: P_MFG_FMT "%s%d" over RANDOM(1, 5, P_MFG_SD)
    t = pa.table({"p_partkey": pa.array(np.arange(1, n + 1, dtype=np.int64)), "p_mfgr": _strings(mfgr - 1, mfgrs, strings), "p_brand": _strings(bcode, brands, strings),
                  "p_type": _strings(ptype, PART_TYPES, strings),
                  "p_size": pa.array(size.astype(np.int32)), "p_container": _strings(cntr, CONTAINERS, strings)})
    if strings != "codes":
        t = t.append_column("p_name", _string_column(part_names(n), strings))
    return t


def partsupp(sf: float) -> pa.Table:
    """build.c mk_part, the SUPP_PER_PART = 4 partsupp rows of every part: ps_suppkey by PART_SUPP_BRIDGE (dss.h), ps_availqty =
    RANDOM(1, 9999, PS_QTY_SD), ps_supplycost = RANDOM(100, 100000, PS_SCST_SD) cents.  (ps_comment is text and not generated.)
    Pinned by the SF0.1 answer of Q11 (sqllogictest/test_files/tpch/answers/q11.slt.part: all ten rows, to the cent).  The one
    row of core/tests/tpch-csv/partsupp.csv (`67310,7311,100,993.49`) agrees in its supplier key only — like the part.csv row it
    does not come from this dbgen (tests/test_dbgen_golden.py)."""
    n = counts(sf)["part"]
    tot_scnt = n // 20                                   # suppliers: tdefs[SUPP].base * scale
    part_of = np.repeat(np.arange(n, dtype=np.int64), SUPP_PER_PART)
    s_idx = np.tile(np.arange(SUPP_PER_PART, dtype=np.int64), n)
    p = part_of + 1
    suppkey = (p + s_idx * (tot_scnt // SUPP_PER_PART + (p - 1) // tot_scnt)) % tot_scnt + 1
    qty = _draw_lines(PS_QTY_SD, part_of, s_idx, n, 1, 9999)
    cost = _draw_lines(PS_SCST_SD, part_of, s_idx, n, 100, 100000)
    return pa.table({"ps_partkey": pa.array(p), "ps_suppkey": pa.array(suppkey), "ps_availqty": pa.array(qty.astype(np.int32)),
                     "ps_supplycost": _decimal(cost)})
