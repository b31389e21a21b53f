"""K10/K11/K12 parity: RepartitionExec(Hash), SortExec and TopK on the GPU vs the CPU oracle."""
import numpy as np
import pyarrow as pa
import pytest

from tests.util import assert_tables_equal, random_table

pytestmark = pytest.mark.gpu

SPEC = {"k": (pa.int64(), -500, 500), "d": (pa.decimal128(15, 2), -10**6, 10**6), "q": (pa.int32(), -5, 5), "f": (pa.float64(), -100, 100),
        "dt": (pa.date32(), 9000, 9100), "c": (pa.uint8(), 0, 4)}


def run_sort(t, keys, fetch=None):
    from datafusion_amd import ops
    from datafusion_amd.table import DeviceTable
    from oracle import oracle
    got = ops.sort(DeviceTable.from_arrow(t), keys, fetch).to_arrow()
    exp = oracle.sort(t, keys, fetch)
    # both sides are stable sorts, so even ties compare position by position
    assert_tables_equal(got, exp, ordered=True)


@pytest.mark.parametrize("n", [0, 1, 2, 63, 64, 65, 5000, 200_001])
def test_sort_single_key_sizes(n):
    t = random_table(np.random.default_rng(n), n, SPEC)
    run_sort(t, [("k", False, False)])


@pytest.mark.parametrize("keys", [
    [("d", True, False)], [("f", False, True)], [("q", False, False), ("k", True, False)], [("c", True, True), ("dt", False, False), ("q", True, True)],
    [("d", True, False), ("dt", False, False)],   # TPC-H Q3 ORDER BY revenue DESC, o_orderdate
], ids=["dec_desc", "f64", "2keys", "3keys_nulls", "q3_order"])
def test_sort_multi_key_with_nulls(keys):
    t = random_table(np.random.default_rng(21), 30_000, SPEC, null_frac=0.1)
    run_sort(t, keys)


@pytest.mark.parametrize("k", [1, 10, 100, 5000])
def test_topk(k):
    """SortExec with fetch = TopK (topk/mod.rs:397); Q3 is fetch=10"""
    t = random_table(np.random.default_rng(k), 100_000, SPEC, null_frac=0.02)
    run_sort(t, [("d", True, False), ("dt", False, False)], fetch=k)
    run_sort(t, [("q", False, True)], fetch=k)     # heavy ties


@pytest.mark.parametrize("shape", ["wide_uniform", "skewed_top_bits", "heavy_ties", "two_keys_wide", "tiny_buckets"])
@pytest.mark.parametrize("n", [3000, 70_000, 700_000])
def test_sort_top_digits_in_hbm_buckets_in_lds(shape, n):
    """the two-level sort (sort.hip sorted_ids_local): stable passes over the top digits, every bucket finished in LDS.  Shapes: keys
    that spread over their range (buckets of ~n / 2^top rows), top bits held by three values only (buckets beyond the LDS capacity:
    the all-HBM fallback after the key buffer was used as scratch), few distinct keys (stability inside long runs of ties), two wide
    key columns with NULLs and DESC, and far more buckets than rows"""
    rng = np.random.default_rng(n % 1000 + len(shape))
    v = pa.array(np.arange(n, dtype=np.int64))
    if shape == "wide_uniform":
        t = pa.table({"a": pa.array(rng.integers(-2**40, 2**40, size=n)), "v": v})
        keys = [("a", False, False)]
    elif shape == "skewed_top_bits":
        t = pa.table({"a": pa.array((rng.integers(0, 3, size=n) << 40) + rng.integers(0, 2**20, size=n)), "v": v})
        keys = [("a", True, False)]
    elif shape == "heavy_ties":
        t = pa.table({"a": pa.array(rng.integers(0, 7, size=n) * 10**9), "v": v})
        keys = [("a", False, False)]
    elif shape == "two_keys_wide":
        a = pa.array(rng.integers(8000, 11000, size=n).astype(np.int32), pa.int32(), mask=rng.random(n) < 0.05).cast(pa.date32())
        t = pa.table({"a": a, "b": pa.array(rng.integers(0, 2**33, size=n)), "v": v})
        keys = [("a", False, True), ("b", True, False)]
    else:
        t = pa.table({"a": pa.array(rng.integers(0, 2**62, size=n)), "v": v})
        keys = [("a", False, False)]
    run_sort(t, keys)


def test_sort_special_floats():
    t = pa.table({"f": pa.array([0.0, -0.0, float("inf"), float("-inf"), float("nan"), 1.5, None, -1.5]), "i": pa.array(range(8), type=pa.int32())})
    run_sort(t, [("f", False, False)])
    run_sort(t, [("f", True, True)])


@pytest.mark.parametrize("nparts", [1, 2, 3, 8, 16])
def test_hash_partition_matches_oracle_routing(nparts):
    """same hash (seed 0) and modulo as the oracle => identical partition contents, input order kept"""
    from datafusion_amd import ops
    from datafusion_amd.table import DeviceTable
    from oracle import oracle
    t = random_table(np.random.default_rng(5), 50_000, SPEC)
    parts = ops.partition(DeviceTable.from_arrow(t), ["k"], nparts)
    exp, _ = oracle.hash_partition(t, ["k"], nparts)
    assert sum(p.num_rows for p in parts) == t.num_rows
    for p, e in zip(parts, exp):
        assert_tables_equal(p.to_arrow(), e, ordered=True)


def test_hash_partition_multi_key_and_nulls():
    from datafusion_amd import ops
    from datafusion_amd.table import DeviceTable
    from oracle import oracle
    t = random_table(np.random.default_rng(6), 20_000, SPEC, null_frac=0.1)
    parts = ops.partition(DeviceTable.from_arrow(t), ["q", "d"], 4)
    exp, _ = oracle.hash_partition(t, ["q", "d"], 4)
    for p, e in zip(parts, exp):
        assert_tables_equal(p.to_arrow(), e, ordered=True)


def test_partition_then_join_equals_global_join():
    """PartitionMode::Partitioned (hash_join/exec.rs:1314-1324): co-partitioned sides joined per partition"""
    from datafusion_amd import ops
    from datafusion_amd.table import DeviceTable
    from oracle import oracle
    rng = np.random.default_rng(7)
    l = random_table(rng, 5000, {"a": (pa.int64(), 0, 2000), "x": (pa.int32(), 0, 9)})
    r = random_table(rng, 20_000, {"b": (pa.int64(), 0, 2500), "y": (pa.decimal128(15, 2), 0, 10**5)})
    lp, rp = ops.partition(DeviceTable.from_arrow(l), ["a"], 4), ops.partition(DeviceTable.from_arrow(r), ["b"], 4)
    outs = [ops.hash_join(a, b, [("a", "b")], "Inner").to_arrow() for a, b in zip(lp, rp)]
    assert_tables_equal(pa.concat_tables(outs), oracle.hash_join(l, r, [("a", "b")], "Inner"))


_RCCL_WORKER = r"""
import os, sys
import numpy as np, pyarrow as pa, torch
import torch.distributed as dist
sys.path.insert(0, os.environ["DFGPU_ROOT"])
from datafusion_amd.exchange import broadcast_table, hash_exchange, pruned_broadcast_table
from datafusion_amd.table import DeviceTable
from tests.util import assert_tables_equal, random_table
t = random_table(np.random.default_rng(2), 100_000, {"k": (pa.int64(), 0, 10**6), "d": (pa.decimal128(15, 2), 0, 10**6), "q": (pa.int32(), 0, 9)})
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
out = hash_exchange(DeviceTable.from_arrow(t), ["k"], force=True)
assert_tables_equal(out.to_arrow(), t, ordered=True)
# CollectLeft build side: all-gather of every column (one rank: the gathered table is the input)
bc = broadcast_table(DeviceTable.from_arrow(t), force=True)
assert_tables_equal(bc.to_arrow(), t, ordered=True)
# bounds-pruned CollectLeft: the probe keys cover [200000, 600000] -> only build rows inside those bounds arrive
import pyarrow.compute as pc
probe = pa.table({"k2": pa.array(np.random.default_rng(3).integers(200_000, 600_001, 50_000), type=pa.int64())})
xs = {}
pb = pruned_broadcast_table(DeviceTable.from_arrow(t), "k", DeviceTable.from_arrow(probe), "k2", force=True, stats=xs)
lo, hi = pc.min(probe.column("k2")).as_py(), pc.max(probe.column("k2")).as_py()
exp = t.filter(pc.and_(pc.greater_equal(t.column("k"), lo), pc.less_equal(t.column("k"), hi)))
assert_tables_equal(pb.to_arrow(), exp, ordered=True)
assert xs["build_rows_after_exchange"] == exp.num_rows < t.num_rows
print("EXCHANGE_OK", flush=True)
os._exit(0)   # RCCL teardown in a one-rank group has aborted on some boxes; the result is already checked
"""


def test_exchange_plumbing_single_rank_rccl():
    """one-rank RCCL group on the GPU box: partition -> zero-copy torch views of library HBM ->
    all_to_all_single -> received table; with world=1 the result must equal the input.  Runs in a
    child process so that RCCL state (and its teardown) cannot touch the rest of the session."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", DFGPU_ROOT=root)
    r = subprocess.run([sys.executable, "-c", _RCCL_WORKER], env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert "EXCHANGE_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("shape", ["spread", "two_keys_desc_nulls", "few_distinct", "all_equal"])
@pytest.mark.parametrize("k", [1, 10, 5000])
def test_topk_by_sampled_limit(shape, k):
    """TopK of >= 1 Mi rows with a one-word key (sort.hip: a limit from 64 K sampled keys, one pass over the key columns, the sort of
    the rows under it): keys that spread, two key columns with DESC and NULLs, few distinct values and one value only (ties: the rows
    under the limit are too many and the radix select takes over) — always the first k rows of the stable sort"""
    n = 1_300_003
    rng = np.random.default_rng(k + len(shape))
    v = pa.array(np.arange(n, dtype=np.int64))
    if shape == "spread":
        t = pa.table({"a": pa.array(rng.integers(-2**40, 2**40, size=n)), "v": v})
        keys = [("a", False, False)]
    elif shape == "two_keys_desc_nulls":
        a = pa.array(rng.integers(8000, 11000, size=n).astype(np.int32), pa.int32(), mask=rng.random(n) < 0.01).cast(pa.date32())
        t = pa.table({"a": a, "b": pa.array(rng.integers(0, 2**31, size=n)), "v": v})
        keys = [("a", True, False), ("b", False, False)]
    elif shape == "few_distinct":
        t = pa.table({"a": pa.array(rng.integers(0, 5, size=n)), "v": v})
        keys = [("a", False, False)]
    else:
        t = pa.table({"a": pa.array(np.full(n, 7, dtype=np.int64)), "v": v})
        keys = [("a", True, True)]
    run_sort(t, keys, fetch=k)


@pytest.mark.parametrize("shape", ["orders_shape", "one_bucket", "three_keys_u8_date_desc", "payload_of_exactly_16_bytes", "skewed_top_bits", "heavy_ties",
                                   "payload_too_wide", "nullable_key", "uint32_and_negative_keys"])
@pytest.mark.parametrize("mode", ["onesweep", "records_by_row_id", "records_through_the_passes"])
def test_sort_carried_records_and_keys_decoded_from_the_packed_key(monkeypatch, shape, mode):
    """the carried sort (sort.hip sort_carried_onesweep / sort_carried; forced here for tables of a few MB): the columns the packed key does
    not hold become ONE 16-byte record per row — travelling with the key through onesweep top passes whose first pass reads the source
    columns (round 5, the default), fetched by row id inside the LDS bucket sort (DFGPU_SORT_CARRIED=ids, round 4's default) or travelling
    through the three-kernel passes (DFGPU_SORT_CARRIED=passes) — and the bucket sort writes the output: key columns DECODED from the sorted mixed-radix
    key (ASC and DESC, dates, negative and unsigned values, UInt8), payload fields from the records.  No separate take runs.  Same stable order as the oracle position by position (ties keep their input order: the payload tells).  Shapes
    it must decline and leave to the other paths: a payload beyond 16 bytes, a nullable key column, buckets beyond the LDS capacity"""
    from datafusion_amd import ops
    rng = np.random.default_rng(len(shape) * 7)
    n = 700_000
    carried = True
    if shape == "orders_shape":
        t = pa.table({"o_orderkey": pa.array(rng.permutation(n).astype(np.int64) * 4 + 1), "o_custkey": pa.array(rng.integers(1, 10**6, n)),
                      "o_orderdate": pa.array(rng.integers(8035, 10441, n).astype(np.int32), pa.int32()).cast(pa.date32()),
                      "o_shippriority": pa.array(rng.integers(0, 3, n).astype(np.int32))})
        keys = [("o_orderdate", False, False), ("o_orderkey", True, False)]
    elif shape == "one_bucket":
        n = 1500                                                                                    # fewer rows than one bucket holds: no top pass, records built on their own
        t = pa.table({"a": pa.array(rng.integers(-50, 50, n)), "v": pa.array(np.arange(n, dtype=np.int64)), "w": pa.array(rng.integers(0, 255, n).astype(np.uint8))})
        keys = [("a", True, False)]
    elif shape == "three_keys_u8_date_desc":
        t = pa.table({"c": pa.array(rng.integers(3, 9, n).astype(np.uint8)), "d": pa.array(rng.integers(9000, 9400, n).astype(np.int32), pa.int32()).cast(pa.date32()),
                      "k": pa.array(rng.integers(-10**6, 10**6, n)), "v": pa.array(np.arange(n, dtype=np.int32)), "x": pa.array(rng.integers(0, 2**40, n))})
        keys = [("c", True, False), ("d", False, False), ("k", True, False)]
    elif shape == "payload_of_exactly_16_bytes":
        t = pa.table({"a": pa.array(rng.integers(-2**40, 2**40, size=n)), "v": pa.array(np.arange(n, dtype=np.int64)), "w": pa.array(rng.integers(-2**62, 2**62, n))})
        keys = [("a", False, False)]
    elif shape == "skewed_top_bits":                                                                # three values hold the top bits: buckets beyond the LDS capacity
        t = pa.table({"a": pa.array((rng.integers(0, 3, size=n) << 40) + rng.integers(0, 2**20, size=n)), "v": pa.array(np.arange(n, dtype=np.int64))})
        keys, carried = [("a", True, False)], False
    elif shape == "heavy_ties":                                                                     # seven distinct keys: a bucket holds a seventh of the rows
        t = pa.table({"a": pa.array(rng.integers(0, 7, size=n) * 10**9), "v": pa.array(np.arange(n, dtype=np.int64)), "w": pa.array(rng.integers(0, 255, n).astype(np.uint8))})
        keys, carried = [("a", False, False)], False
    elif shape == "payload_too_wide":
        t = pa.table({"a": pa.array(rng.integers(-2**40, 2**40, size=n)), "v": pa.array(np.arange(n, dtype=np.int64)), "w": pa.array(rng.integers(0, 9, n)), "x": pa.array(rng.integers(0, 9, n).astype(np.int32))})
        keys, carried = [("a", False, False)], False
    elif shape == "nullable_key":
        t = pa.table({"a": pa.array(rng.integers(-2**40, 2**40, size=n), mask=rng.random(n) < 0.01), "v": pa.array(np.arange(n, dtype=np.int64))})
        keys, carried = [("a", False, True)], False
    else:
        t = pa.table({"u": pa.array(rng.integers(2**31, 2**32 - 1, n).astype(np.uint32)), "s": pa.array(rng.integers(-2**31, -2**30, n).astype(np.int32)),
                      "v": pa.array(np.arange(n, dtype=np.int64))})
        keys = [("s", False, False), ("u", True, False)]
    ops.set_options(sort__carried_min_rows="0", sort__lsd="0")   # (narrow keys would take the record passes of the next test)
    ops.set_options(sort__carried={"onesweep": "onesweep", "records_by_row_id": "ids", "records_through_the_passes": "passes"}[mode])
    ops.profile_enable(True)
    ops.profile_reset()
    run_sort(t, keys)
    stats = ops.profile_stats()
    ops.profile_enable(False)
    if carried:
        assert "sort_local_emit" in stats and "take_gather_rows" not in stats and "gather" not in stats, sorted(stats)
        if mode == "onesweep" and shape != "one_bucket":   # (a table of one bucket has no top pass: the older form builds the records on their own)
            assert "sort_onesweep_pass" in stats and "sort_pack_keys" not in stats and "radix_sort_pass" not in stats and "sort_build_records" not in stats, sorted(stats)
        elif mode == "records_through_the_passes":
            assert ("sort_carried_pass" in stats) == (shape != "one_bucket") and "radix_sort_pass" not in stats, sorted(stats)
        elif mode == "records_by_row_id":
            assert "sort_build_records" in stats and ("radix_sort_pass" in stats) == (shape != "one_bucket"), sorted(stats)
    else:
        assert "sort_local_emit" not in stats, sorted(stats)


@pytest.mark.parametrize("shape", ["orders_by_date_then_ascending_key_desc", "orders_by_date_then_ascending_key_asc", "orders_by_date_then_ascending_key_desc_with_look_back",
                                   "two_passes_skewed_digits_16_byte_record", "one_pass_u8_key_16_byte_record",
                                   "three_passes_two_keys_32_byte_record", "key_only_table", "record_too_wide", "key_beyond_32_bits", "nullable_key",
                                   "ragged_tail_and_ties"])
def test_sort_narrow_keys_by_record_passes(shape):
    """the LSD carried sort (sort.hip sort_lsd_carried / k_lsd_pass, round 5): a packed key of at most 32 bits — after dropping a key column that
    is strictly ascending in input order and everything behind it (ASC: the stable passes keep the input order of ties; DESC: the first
    pass reads the rows back to front) — is sorted by two to four stable passes over 16 / 24 / 32-byte records that carry the key in
    their last word; the last pass writes the output columns, key columns decoded from that word.  No bucket sort, no key array, no
    take.  Same stable order as the oracle position by position; wider records, wider keys and nullable keys are left to the other paths"""
    from datafusion_amd import ops
    rng = np.random.default_rng(len(shape) * 11)
    n = 300_001
    lsd = True
    if shape.startswith("orders_by_date"):
        t = pa.table({"o_orderkey": pa.array(np.cumsum(rng.integers(1, 9, n)).astype(np.int64)), "o_custkey": pa.array(rng.integers(1, 10**6, n)),
                      "o_orderdate": pa.array(rng.integers(8035, 10441, n).astype(np.int32), pa.int32()).cast(pa.date32()),
                      "o_shippriority": pa.array(rng.integers(0, 3, n).astype(np.int32))})
        keys = [("o_orderdate", False, False), ("o_orderkey", "_desc" in shape, False)]
        if shape.endswith("with_look_back"):   # (two passes over 12 bits take the offsets-ahead-of-time form by default: this is the other leg)
            ops.set_options(sort__lsd_ahead="0")
    elif shape == "two_passes_skewed_digits_16_byte_record":
        # 10 bits in two passes, most rows on three key values (units of very different sizes, empty ones), more than one coarse segment
        n = 1_200_003
        a = np.where(rng.random(n) < 0.9, rng.choice([5, 700, 701], n), rng.integers(0, 1000, n)).astype(np.int32)
        t = pa.table({"a": pa.array(a), "v": pa.array(np.arange(n, dtype=np.int64)), "w": pa.array(rng.integers(0, 2**31, n).astype(np.int32))})
        keys = [("a", False, False)]
    elif shape == "one_pass_u8_key_16_byte_record":
        t = pa.table({"c": pa.array(rng.integers(3, 200, n).astype(np.uint8)), "v": pa.array(np.arange(n, dtype=np.int64)), "w": pa.array(rng.integers(0, 2**31, n).astype(np.int32))})
        keys = [("c", True, False)]
    elif shape == "three_passes_two_keys_32_byte_record":
        t = pa.table({"d": pa.array(rng.integers(-3000, 3000, n).astype(np.int32)), "e": pa.array(rng.integers(0, 1500, n)), "x": pa.array(rng.integers(-2**62, 2**62, n)),
                      "y": pa.array(rng.integers(0, 255, n).astype(np.uint8)), "dec": pa.array([int(v) for v in rng.integers(-10**15, 10**15, n)], pa.decimal128(30, 2))})
        keys = [("e", False, False), ("d", True, False)]
    elif shape == "key_only_table":
        t = pa.table({"a": pa.array(rng.integers(-40000, 40000, n).astype(np.int32)), "b": pa.array(rng.integers(0, 50, n).astype(np.uint8))})
        keys = [("b", False, False), ("a", False, False)]
    elif shape == "record_too_wide":
        t = pa.table({"a": pa.array(rng.integers(0, 4000, n)), "v": pa.array(np.arange(n, dtype=np.int64)), "w": pa.array(rng.integers(0, 9, n)), "x": pa.array(rng.integers(0, 9, n)),
                      "y": pa.array(rng.integers(0, 9, n))})
        keys, lsd = [("a", False, False)], False
    elif shape == "key_beyond_32_bits":
        t = pa.table({"a": pa.array(rng.integers(0, 2**33, n)), "v": pa.array(np.arange(n, dtype=np.int64))})
        keys, lsd = [("a", False, False)], False
    elif shape == "nullable_key":
        t = pa.table({"a": pa.array(rng.integers(0, 4000, n), mask=rng.random(n) < 0.01), "v": pa.array(np.arange(n, dtype=np.int64))})
        keys, lsd = [("a", False, True)], False
    else:
        n = 2048 * 5 + 3
        t = pa.table({"a": pa.array(rng.integers(0, 5, n)), "v": pa.array(np.arange(n, dtype=np.int64)), "row": pa.array(np.arange(n, dtype=np.int64) * 3)})
        keys = [("a", True, False), ("row", True, False), ("v", False, False)]
    ops.set_options(sort__carried_min_rows="0")
    ops.profile_enable(True)
    ops.profile_reset()
    run_sort(t, keys)
    stats = ops.profile_stats()
    ops.profile_enable(False)
    if lsd:
        assert "sort_lsd_pass_out" in stats and not {"sort_local_emit", "sort_onesweep_pass", "radix_sort_pass", "sort_pack_keys", "gather", "take_gather_rows"} & set(stats), sorted(stats)
    else:
        assert "sort_lsd_pass_out" not in stats, sorted(stats)


def test_sort_and_joins_move_boolean_payload_columns():
    """take of a bit-packed column (SortExec's output, the general join path's gathers): Boolean payload with NULLs"""
    from datafusion_amd import ops
    from datafusion_amd.table import DeviceTable
    from oracle import oracle
    rng = np.random.default_rng(19)
    n = 50_000
    t = pa.table({"k": pa.array(rng.integers(0, 5000, n)), "b": pa.array(rng.random(n) < 0.4, mask=rng.random(n) < 0.1), "c": pa.array(rng.random(n) < 0.5)})
    got = ops.sort(DeviceTable.from_arrow(t), [("k", False, False)]).to_arrow()
    order = np.argsort(t.column("k").to_numpy(), kind="stable")
    assert got.column("b").to_pylist() == t.column("b").take(pa.array(order)).to_pylist() and got.column("c").to_pylist() == t.column("c").take(pa.array(order)).to_pylist()
    build = pa.table({"bk": pa.array(np.arange(5000, dtype=np.int64)), "flag": pa.array(np.arange(5000) % 3 == 0, mask=np.arange(5000) % 7 == 0)})
    j = ops.hash_join(DeviceTable.from_arrow(build), DeviceTable.from_arrow(t), [("bk", "k")], "Right").to_arrow()
    as_u8 = lambda x: pa.table({c: (x.column(c).cast(pa.uint8()) if pa.types.is_boolean(x.schema.field(c).type) else x.column(c)) for c in x.column_names})
    exp = oracle.hash_join(as_u8(build), as_u8(t), [("bk", "k")], "Right")
    assert_tables_equal(as_u8(j), exp, ordered=False)


@pytest.mark.parametrize("n", [1, 2, 3, 100, 1025, 4095, 4096, 4097])
@pytest.mark.parametrize("keys", [[("q", False, False)], [("d", True, False), ("dt", False, False)], [("c", True, True), ("dt", False, False), ("q", True, True)],
                                  [("f", False, True), ("k", True, False)], [("d", True, False), ("k", False, True), ("d", False, False)]],
                         ids=["heavy_ties_one_word", "q3_order_two_words", "3keys_nulls", "f64_then_i64", "three_words"])
def test_small_inputs_are_sorted_by_one_workgroup_in_lds(n, keys):
    """round 6: up to 4096 rows — a small result, a TopK's survivors — are sorted by ONE workgroup in LDS (k_small_sort: a bitonic network over
    (key words, position); the position makes the sort stable) instead of 5 launches per 8-bit digit.  Same rows in the same order as the
    oracle's stable sort and as the radix passes (sort.small=0), with and without fetch; 4097 rows take the old path."""
    from datafusion_amd import ops
    from datafusion_amd.table import DeviceTable
    from oracle import oracle
    t = random_table(np.random.default_rng(n), n, SPEC, null_frac=0.1)
    dev = DeviceTable.from_arrow(t)
    for fetch in (None, 1, 10):
        ops.profile_enable(True)
        ops.profile_reset()
        got = ops.sort(dev, keys, fetch).to_arrow()
        names = set(ops.profile_stats())
        ops.profile_enable(False)
        if 100 <= n <= 4096:
            assert "sort_small" in names and "radix_sort_pass" not in names, (n, names)
        elif n > 4096 and fetch is None:   # (with a fetch the TopK narrows 4097 rows down first: its survivors are sorted in LDS)
            assert "sort_small" not in names, (n, names)
        assert_tables_equal(got, oracle.sort(t, keys, fetch), ordered=True)
        ops.set_options(sort__small="0")
        try:
            old = ops.sort(dev, keys, fetch).to_arrow()
        finally:
            ops.set_options(sort__small=None)
        assert old.equals(got)


def test_topk_survivors_are_sorted_in_lds():
    """Q3's shape: TopK(10) by (Decimal128 DESC, Date32) over 3 M groups — the MSD select narrows to a few thousand survivors, one workgroup
    sorts them"""
    from datafusion_amd import ops
    from datafusion_amd.table import DeviceTable
    from oracle import oracle
    rng = np.random.default_rng(33)
    n = 3_000_000
    t = pa.table({"revenue": pa.array(rng.integers(0, 10**9, n) * 10**4, type=pa.int64()).cast(pa.decimal128(38, 4)), "o_orderdate": pa.array(rng.integers(8000, 9000, n).astype(np.int32), type=pa.date32()),
                  "l_orderkey": pa.array(np.arange(n), type=pa.int64())})
    keys = [("revenue", True, True), ("o_orderdate", False, False)]
    ops.profile_enable(True)
    ops.profile_reset()
    got = ops.sort(DeviceTable.from_arrow(t), keys, 10).to_arrow()
    names = set(ops.profile_stats())
    ops.profile_enable(False)
    assert "sort_small" in names and "radix_sort_pass" not in names, names
    assert_tables_equal(got, oracle.sort(t, keys, 10), ordered=True)
