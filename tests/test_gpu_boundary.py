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
