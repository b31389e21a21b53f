"""The streaming boundary (SURVEY §8 a7 / b): the build side as a stream of batches (CollectBuildSide, hash_join/stream.rs:127-140 over
collect_left_input, exec.rs:2569-2705), output batching (LimitedBatchCoalescer, coalesce/mod.rs:27-120) through pinned exports, and
admission control (MemoryReservation::try_grow, execution/src/memory_pool/mod.rs:188) with the rule's spill-aware fallback."""
import numpy as np
import pyarrow as pa
import pytest

from tests.util import assert_tables_equal, random_table

pytestmark = pytest.mark.gpu


def dict_col(codes, values, mask=None):
    return pa.DictionaryArray.from_arrays(pa.array(codes, type=pa.int32(), mask=mask), pa.array(values, type=pa.string()))


def decoded(t):
    return pa.table({n: (c.cast(pa.string()) if pa.types.is_dictionary(c.type) else c) for n, c in zip(t.column_names, t.columns)})


@pytest.mark.parametrize("batch_rows", [1, 63, 64, 1000, 8192])
def test_export_batches_are_the_slices_of_the_table(batch_rows):
    from datafusion_amd.table import DeviceTable
    rng = np.random.default_rng(3)
    n = 20_011 if batch_rows > 1 else 130
    t = pa.table({"k": pa.array(rng.integers(0, 10**9, n), type=pa.int64()),
                  "q": pa.array(rng.integers(0, 100, n), type=pa.int32(), mask=rng.random(n) < 0.2),
                  "b": pa.array(rng.random(n) < 0.5, type=pa.bool_(), mask=rng.random(n) < 0.1),
                  "d": random_table(rng, n, {"d": (pa.decimal128(15, 2), 0, 10**9)}).column("d"),
                  "s": dict_col(rng.integers(0, 3, n), ["a", "bb", "ccc"], mask=rng.random(n) < 0.1)})
    d = DeviceTable.from_arrow(t)
    batches = list(d.to_batches(batch_rows))
    assert [b.num_rows for b in batches] == [min(batch_rows, n - o) for o in range(0, n, batch_rows)]
    assert_tables_equal(decoded(pa.Table.from_batches(batches)), decoded(t), ordered=True)
    assert_tables_equal(decoded(d.to_arrow()), decoded(t), ordered=True)
    empty = DeviceTable.from_arrow(t.slice(0, 0))
    assert [b.num_rows for b in empty.to_batches(batch_rows)] == [0]


def test_export_into_caller_registered_buffers():
    import ctypes as C

    from datafusion_amd import _lib
    from datafusion_amd.table import DeviceTable
    rng = np.random.default_rng(4)
    n = 100_000
    k, q = rng.integers(0, 10**9, n), rng.integers(0, 100, n).astype(np.int32)
    d = DeviceTable.from_arrow(pa.table({"k": pa.array(k, type=pa.int64()), "q": pa.array(q, type=pa.int32())}))
    lib = _lib.load()
    off, length = 64 * 100, 50_000
    hk, hq = np.zeros(length, np.int64), np.zeros(length, np.int32)
    for a in (hk, hq):
        _lib.check(lib.dfgpu_host_register(C.c_void_p(a.ctypes.data), C.c_size_t(a.nbytes)))
    bufs = (C.c_void_p * 2)(hk.ctypes.data, hq.ctypes.data)
    _lib.check(lib.dfgpu_table_export_into(d.handle, C.c_int64(off), C.c_int64(length), bufs, None))
    for a in (hk, hq):
        _lib.check(lib.dfgpu_host_unregister(C.c_void_p(a.ctypes.data)))
    assert (hk == k[off:off + length]).all() and (hq == q[off:off + length]).all()


@pytest.mark.parametrize("join_type", ["Inner", "Left", "RightAnti"])
def test_build_side_pushed_batch_by_batch_equals_the_whole_build(join_type):
    from datafusion_amd import ops
    from datafusion_amd.table import DeviceTable
    from oracle import oracle
    rng = np.random.default_rng(5)
    build = random_table(rng, 9000, {"a": (pa.int64(), 0, 3000), "x": (pa.decimal128(15, 2), 0, 10**6)}, null_frac=0.03)
    probe = random_table(rng, 20000, {"b": (pa.int64(), 0, 3500), "y": (pa.int32(), 0, 100)}, null_frac=0.03)
    jb = ops.JoinBuilder([0])
    for lo, hi in ((0, 1), (1, 4000), (4000, 4000), (4000, 8999), (8999, 9000)):   # uneven batches, an empty one among them
        jb.push(DeviceTable.from_arrow(build.slice(lo, hi - lo)))
    ht = jb.finish()
    p = DeviceTable.from_arrow(probe)
    out = ht.probe(p, ["b"], join_type)
    if join_type == "Left":
        tail = ht.emit_unmatched("Left", None, probe.schema)
        out = ops.concat_tables([out, tail])
    assert_tables_equal(out.to_arrow(), oracle.hash_join(build, probe, [("a", "b")], join_type))


def test_reservations_and_the_rule_declining_a_join_that_does_not_fit():
    from datafusion_amd import _lib, ops
    from datafusion_amd import physical_plan as P
    from datafusion_amd.table import DeviceTable
    rng = np.random.default_rng(6)
    build = random_table(rng, 50_000, {"a": (pa.int64(), 0, 10**6), "x": (pa.int64(), 0, 10**6)})
    probe = random_table(rng, 200_000, {"b": (pa.int64(), 0, 10**6), "y": (pa.int64(), 0, 10**6)})
    b, p = DeviceTable.from_arrow(build), DeviceTable.from_arrow(probe)
    limit0, reserved0 = ops.mem_limit()
    assert limit0 > 200 << 30 and reserved0 == 0                              # 92 % of 288 GB
    plan = P.HashJoinExec(P.MemoryExec(b, "build"), P.MemoryExec(p, "probe"), [("a", "b")], "Inner")
    try:
        with ops.Reservation(1 << 30):
            assert ops.mem_limit()[1] == 1 << 30
        assert ops.mem_limit()[1] == 0
        ops.mem_set_limit(4 << 20)                                            # a pool the join cannot fit in
        with pytest.raises(_lib.DfgpuError, match="Resources exhausted"):
            ops.Reservation(1 << 30)
        jb = ops.JoinBuilder([0])
        with pytest.raises(_lib.DfgpuError, match="Resources exhausted"):     # collect_left_input's try_grow failing
            jb.push(b)
        jb.free()
        rule = P.GpuOffloadRule()
        kept = rule.optimize(plan)
        assert getattr(kept, "kept_on_cpu", False) and len(rule.declined) == 1 and "Resources exhausted" in rule.declined[0][1]
        ops.mem_set_limit(0)
        rule = P.GpuOffloadRule()
        opt = rule.optimize(plan)
        assert not getattr(opt, "kept_on_cpu", False) and not rule.declined
        assert P.collect(opt).num_rows == ops.hash_join(b, p, [("a", "b")], "Inner").num_rows
    finally:
        ops.mem_set_limit(0)


def test_operator_metrics_are_per_thread():
    """dfgpu_metrics (MetricsSet / BaselineMetrics of a GPU node): rows, PCIe bytes, algorithmic HBM bytes, host and device time of the
    calling thread; another thread's work does not show up"""
    import threading

    import numpy as np
    import pyarrow as pa

    from datafusion_amd import ops
    from datafusion_amd.expr import col, lit
    from datafusion_amd.table import DeviceTable
    n = 200_000
    t = pa.table({"k": pa.array(np.arange(n, dtype=np.int64)), "v": pa.array(np.arange(n, dtype=np.int32) % 7)})
    ops.profile_enable(True)
    ops.metrics_reset()
    dev = DeviceTable.from_arrow(t)
    out = ops.filter(dev, col("v") < lit(3, pa.int32()))
    got = out.to_arrow()
    m = ops.metrics()
    assert m["h2d_bytes"] == n * 12 and m["d2h_bytes"] == got.num_rows * 12
    assert m["rows_out"] == n + got.num_rows and m["rows_in"] == n + got.num_rows   # out: the imported table + the filter's output; in: the filter's input + the exported table
    assert m["hbm_bytes_algorithmic"] >= n * 4 + (n + got.num_rows) * 12 and m["elapsed_ns"] > 0 and m["kernel_ns"] > 0 and m["calls"] >= 3
    seen = {}

    def other():
        ops.metrics_reset()
        ops.filter(dev, col("v") < lit(1, pa.int32())).free()
        seen.update(ops.metrics())
    before = ops.metrics()
    th = threading.Thread(target=other)
    th.start()
    th.join()
    after = ops.metrics()
    assert seen["rows_in"] == n and seen["h2d_bytes"] == 0
    assert after["rows_in"] == before["rows_in"] and after["hbm_bytes_algorithmic"] == before["hbm_bytes_algorithmic"]
    ops.profile_enable(False)


def test_handles_are_usable_concurrently_from_different_threads():
    """different handles usable concurrently from different host threads.  Every thread works on a stream of its own
    (dfgpu_stream differs per thread), builds / probes / aggregates its own tables while the others do the same, shares one join
    table built by the main thread (CollectLeft: one build, many probers), and frees its handles itself; every result equals the
    oracle's.  The pool must never hand a block that one thread's stream still works on to another thread."""
    import threading

    import numpy as np
    import pyarrow as pa

    from datafusion_amd import _lib, ops
    from datafusion_amd.expr import col, lit
    from datafusion_amd.table import DeviceTable
    from oracle import oracle
    from tests.util import assert_tables_equal, random_table, to_oracle_expr
    lib = _lib.init()
    lib.dfgpu_stream.restype = __import__("ctypes").c_void_p
    rng = np.random.default_rng(1)
    shared_build = random_table(rng, 20_000, {"bk": (pa.int64(), 0, 30_000), "bv": (pa.int32(), 0, 1000)})
    shared_build = shared_build.group_by("bk").aggregate([("bv", "min")]).rename_columns(["bk", "bv"])     # unique keys
    ht = ops.JoinHashTable(DeviceTable.from_arrow(shared_build), ["bk"])
    ht_semi = ops.JoinHashTable(DeviceTable.from_arrow(shared_build), ["bk"])   # its visited bytes are marked by all threads at once
    streams, errors, results, probed_keys = {}, [], {}, []

    def work(tid):
        try:
            streams[tid] = lib.dfgpu_stream()
            r = np.random.default_rng(100 + tid)
            for it in range(6):
                t = random_table(r, 60_000 + 1000 * tid, {"k": (pa.int64(), 0, 30_000), "d": (pa.decimal128(15, 2), -10**6, 10**6), "g": (pa.int32(), 0, 50)}, null_frac=0.05)
                dev = DeviceTable.from_arrow(t)
                pred = col("g") < lit(25 + tid, pa.int32())
                f = ops.filter(dev, pred)
                agg = ops.aggregate(f, [(col("g"), "g")], [("sum", col("d"), "s"), ("count", None, "n")], "Single").to_arrow()
                j = ht.probe(f, ["k"], "Inner", ["bv"], ["k", "d"]).to_arrow()
                ht_semi.probe(f, ["k"], "LeftSemi").free()
                srt = ops.sort(f, [("d", True, False), ("k", False, False)], fetch=50).to_arrow()
                ft = oracle.filter(t, to_oracle_expr(pred), t.column_names)
                results[(tid, it)] = (agg, j, srt, ft)
                probed_keys.append(ft.column("k"))
                f.free()
                dev.free()
        except Exception as e:   # noqa: BLE001
            errors.append((tid, repr(e)))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(6)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    assert len(set(streams.values())) == len(streams) and lib.dfgpu_stream() not in streams.values()   # a stream per thread, none is the main thread's
    from tests.test_gpu_aggregate import assert_agg_equal, oracle_agg
    for (tid, it), (agg, j, srt, ft) in results.items():
        assert_agg_equal(agg, oracle_agg(ft, [(col("g"), "g")], [("sum", col("d"), "s"), ("count", None, "n")], "Single"))
        assert_tables_equal(j, oracle.hash_join(shared_build, ft, [("bk", "k")], "Inner").select(["bv", "k", "d"]))
        assert_tables_equal(srt, oracle.sort(ft, [("d", True, False), ("k", False, False)], 50), ordered=True)
    # LeftSemi: the build rows that ANY thread's probe matched (the shared visited bytes, created once, marked concurrently)
    import pyarrow.compute as pc
    seen = pc.unique(pa.chunked_array([c for col_ in probed_keys for c in col_.chunks]))
    want = shared_build.filter(pc.is_in(shared_build.column("bk"), value_set=seen))
    assert_tables_equal(ht_semi.emit_unmatched("LeftSemi", ["bk", "bv"]).to_arrow(), want)
    ht_semi.free()
    ht.free()


# ------------------------------------------------------------------------------------------------------------------------------------
# Device-resident hand-off (ABI 10): dfgpu_table_retain, Arrow C Device Data Interface export / import, the scan cache under the C ABI

def _mixed_table(rng, n):
    return pa.table({"k": pa.array(rng.integers(0, 10**9, n), type=pa.int64()),
                     "q": pa.array(rng.integers(0, 100, n), type=pa.int32(), mask=rng.random(n) < 0.2),
                     "b": pa.array(rng.random(n) < 0.5, type=pa.bool_(), mask=rng.random(n) < 0.1),
                     "d": random_table(rng, n, {"d": (pa.decimal128(15, 2), 0, 10**9)}).column("d"),
                     "s": dict_col(rng.integers(0, 3, n), ["a", "bb", "ccc"], mask=rng.random(n) < 0.1),
                     "u": pa.array([None if i % 11 == 0 else "row-%d" % (i % 977) for i in range(n)], type=pa.string())})


def test_device_array_round_trip_shares_the_buffers_and_moves_nothing_over_pcie():
    from datafusion_amd import ops
    from datafusion_amd.table import ARROW_DEVICE_ROCM, DeviceTable
    rng = np.random.default_rng(21)
    t = _mixed_table(rng, 70_001)
    d = DeviceTable.from_arrow(t)
    ptrs = [d.column_view(i).data for i in range(d.num_columns)]
    ops.metrics_reset()
    arr, sch = d.export_device()
    assert arr.device_type == ARROW_DEVICE_ROCM and arr.sync_event is None and arr.array.length == t.num_rows and arr.array.n_children == t.num_columns
    assert sch.children[5].contents.format == b"U" and sch.children[4].contents.dictionary        # strings: LargeUtf8 over the stored 64-bit offsets
    for i in range(d.num_columns):
        assert arr.array.children[i].contents.buffers[1 if i != 5 else 2] == ptrs[i]             # pointers INTO the table's HBM
    d.free()                                                                                       # the array keeps the buffers alive
    back = DeviceTable.from_device(arr, sch)
    assert not arr.array.release and not sch.release                                               # consumed
    assert [back.column_view(i).data for i in range(back.num_columns)] == ptrs                     # the same buffers
    m = ops.metrics()
    assert m["h2d_bytes"] == 0 and m["d2h_bytes"] == 0
    assert_tables_equal(decoded(back.to_arrow()), decoded(t), ordered=True)                        # dictionary + names + NULLs intact
    twin = back.retain()
    back.free()
    assert_tables_equal(decoded(twin.to_arrow()), decoded(t), ordered=True)


def test_foreign_device_array_from_torch_tensors_is_wrapped_zero_copy():
    """a producer that is not this library: torch tensors in HBM described by a hand-made ArrowDeviceArray (Int64, Float64, a validity
    bitmap, a Boolean column).  The library wraps the pointers, operators run on them, and the producer's release callback runs when
    the last table that refers to the memory is freed — not before."""
    import ctypes as C

    import torch

    from datafusion_amd import ops
    from datafusion_amd.expr import col, lit
    from datafusion_amd.table import ARROW_DEVICE_ROCM, ArrowArray, ArrowDeviceArray, ArrowSchema, DeviceTable
    n = 100_003
    g = torch.Generator(device="cpu").manual_seed(5)
    k = torch.randint(0, 1000, (n,), generator=g, dtype=torch.int64)
    f = torch.rand((n,), generator=g, dtype=torch.float64)
    valid = torch.rand((n,), generator=g) < 0.8
    flag = torch.rand((n,), generator=g) < 0.3
    pack = lambda bits: torch.from_numpy(np.packbits(np.concatenate([bits.numpy(), np.zeros((-n) % 64 + 64, bool)]), bitorder="little").copy())
    dev = [x.to("cuda:0") for x in (k, f, pack(valid), pack(flag))]
    torch.cuda.synchronize()
    released = []
    RELEASE = C.CFUNCTYPE(None, C.POINTER(ArrowArray))

    @RELEASE
    def release(a):
        released.append(1)
        a.contents.release = None
    names = [b"k", b"f", b"flag"]
    fmts = [b"l", b"g", b"b"]
    bufs = [(C.c_void_p * 2)(None, dev[0].data_ptr()), (C.c_void_p * 2)(dev[2].data_ptr(), dev[1].data_ptr()), (C.c_void_p * 2)(None, dev[3].data_ptr())]
    kids = [ArrowArray(length=n, null_count=(0, int((~valid).sum()), 0)[i], offset=0, n_buffers=2, n_children=0, buffers=bufs[i]) for i in range(3)]
    skids = [ArrowSchema(format=fmts[i], name=names[i], flags=2) for i in range(3)]
    kid_ptrs = (C.POINTER(ArrowArray) * 3)(*[C.pointer(x) for x in kids])
    skid_ptrs = (C.POINTER(ArrowSchema) * 3)(*[C.pointer(x) for x in skids])
    root_bufs = (C.c_void_p * 1)(None)
    arr = ArrowDeviceArray(device_id=0, device_type=ARROW_DEVICE_ROCM, sync_event=None)
    arr.array = ArrowArray(length=n, null_count=0, offset=0, n_buffers=1, n_children=3, buffers=root_bufs, children=kid_ptrs, release=C.cast(release, C.c_void_p))
    sch = ArrowSchema(format=b"+s", name=b"", flags=0, n_children=3, children=skid_ptrs)
    t = DeviceTable.from_device(arr, sch)
    assert t.column_view(0).data == dev[0].data_ptr() and t.column_view(1).validity == dev[2].data_ptr()      # wrapped, not copied
    out = ops.filter(t, (col("k") < lit(500, pa.int64())).and_(col("flag")), ["k", "f"]).to_arrow()
    keep = (k.numpy() < 500) & flag.numpy()
    assert out.column("k").to_pylist() == k.numpy()[keep].tolist()
    exp_f = pa.array(f.numpy()[keep], mask=~valid.numpy()[keep])
    assert out.column("f").combine_chunks().equals(exp_f)
    agg = ops.aggregate(t, [(col("k"), "k")], [("sum", col("f"), "s"), ("count", col("f"), "c")], "Single", predicate=col("flag")).to_arrow()
    assert agg.num_rows == len(set(k.numpy()[flag.numpy()].tolist()))
    for row in agg.slice(0, 50).to_pylist():
        sel = (k.numpy() == row["k"]) & flag.numpy() & valid.numpy()
        assert row["c"] == int(sel.sum()) and abs((row["s"] or 0.0) - float(f.numpy()[sel].sum())) <= 1e-6 * max(1.0, abs(row["s"] or 0.0))
    view = t.select(["k"])
    t.free()
    assert released == []                  # `view` still points into the producer's memory
    view.free()
    assert released == [1]


def test_device_import_rejects_what_it_cannot_wrap():
    import ctypes as C

    from datafusion_amd import _lib
    from datafusion_amd.table import ARROW_DEVICE_ROCM, ArrowArray, ArrowDeviceArray, ArrowSchema, DeviceTable
    _lib.init()
    root_bufs = (C.c_void_p * 1)(None)
    for device_type, device_id, msg in ((1, 0, "not in ROCm device memory"), (ARROW_DEVICE_ROCM, 77, "was not given to dfgpu_init")):
        arr = ArrowDeviceArray(device_id=device_id, device_type=device_type)
        arr.array = ArrowArray(length=0, n_buffers=1, buffers=root_bufs)
        with pytest.raises(_lib.DfgpuError, match=msg):
            DeviceTable.from_device(arr, ArrowSchema(format=b"+s", name=b""))


def test_scan_cache_below_the_c_abi_lru_budget_and_threads():
    """dfgpu_cache_*: hits are zero-copy views, least recently used entries leave under the byte budget, many threads at once"""
    import threading

    from datafusion_amd.parquet import ChunkCache
    from datafusion_amd.table import DeviceTable
    mk = lambda i, n=10_000: DeviceTable.from_arrow(pa.table({"v": pa.array(np.full(n, i, dtype=np.int64))}))
    cache = ChunkCache(budget=250_000)                       # three 80 KB tables
    tables = [mk(i) for i in range(5)]
    for i in range(3):
        cache.put(("f", i), tables[i])
    assert cache.stats()["chunks"] == 3 and cache.stats()["bytes"] == 240_000
    hit = cache.get(("f", 0))                                # touches entry 0: entry 1 is now the oldest
    assert hit.column_view(0).data == tables[0].column_view(0).data
    cache.put(("f", 3), tables[3])
    assert cache.get(("f", 1)) is None and cache.get(("f", 0)) is not None and cache.get(("f", 3)) is not None
    st = cache.stats()
    assert st["chunks"] == 3 and st["evictions"] == 1 and st["hits"] == 3 and st["misses"] == 1
    cache.put(("f", 0), tables[4])                           # an existing key keeps its first table
    assert cache.get(("f", 0)).to_arrow().column("v")[0].as_py() == 0
    big = mk(9, 100_000)
    cache.put(("big",), big)                                 # larger than the whole budget: not kept
    assert cache.get(("big",)) is None
    for t in tables:
        t.free()                                             # the cache holds its own references
    assert cache.get(("f", 3)).to_arrow().column("v").to_pylist() == [3] * 10_000
    shared = ChunkCache(budget=1 << 30)
    errors = []

    def worker(w):
        try:
            for i in range(40):
                key = ("t", i % 8)
                got = shared.get(key)
                if got is None:
                    t = mk(i % 8, 2_000)
                    shared.put(key, t)
                    t.free()
                else:
                    assert got.to_arrow().column("v")[0].as_py() == i % 8
                    got.free()
        except Exception as e:      # noqa: BLE001
            errors.append(e)
    threads = [threading.Thread(target=worker, args=(w,)) for w in range(6)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors and shared.stats()["chunks"] == 8
    cache.clear()
    assert cache.stats()["chunks"] == 0 and cache.stats()["bytes"] == 0
