"""ONE process driving TWO GPUs (dfgpu_init(ids, 2) + dfgpu_comm_init_all): every handle lives on the device that was
current when it was made, entry points switch the calling thread to it, the three exchanges run over RCCL between the two devices, and
the hiprtc-specialised nodes are loaded once per device.  Self-skips on a box with fewer than two GPUs (any multi-GPU node runs it; named test_zz_* so that it
comes last); what it checks against is the oracle's routing / join / aggregate over the same rows."""
import ctypes as C

import numpy as np
import pyarrow as pa
import pytest

pytestmark = pytest.mark.gpu


def _two_devices():
    from datafusion_amd import _lib
    lib = _lib.load()
    n = C.c_int(0)
    assert lib.dfgpu_device_count(C.byref(n)) == 0
    if n.value < 2:
        pytest.skip(f"{n.value} GPU visible: the two-device path needs two")
    _lib.init(0)                                                    # what every other test did already (device 0)
    _lib.check(lib.dfgpu_init((C.c_int * 2)(0, 1), 2))             # idempotent for device 0, adds device 1
    return lib


def _on(lib, device, make):
    from datafusion_amd import _lib
    _lib.check(lib.dfgpu_set_device(device))
    try:
        return make()
    finally:
        _lib.check(lib.dfgpu_set_device(0))


def test_two_devices_one_process_exchanges_joins_and_specialised_nodes():
    import os

    from datafusion_amd import _lib, ops, queries
    from datafusion_amd.expr import col, lit
    from datafusion_amd.table import DeviceTable
    from oracle import oracle
    from tests.util import assert_tables_equal
    lib = _two_devices()
    rng = np.random.default_rng(77)
    shards = []
    for r in range(2):
        n = 50_000 + 1_000 * r
        shards.append(pa.table({"k": pa.array(rng.integers(0, 5_000, n).astype(np.int64)), "v": pa.array(rng.integers(0, 100, n).astype(np.int32), mask=rng.random(n) < 0.1),
                                "s": pa.DictionaryArray.from_arrays(pa.array(rng.integers(0, 3, n).astype(np.int32)), pa.array([f"rank{r}-a", "shared", f"rank{r}-z"]))}))
    tables = [_on(lib, r, lambda r=r: DeviceTable.from_arrow(shards[r])) for r in range(2)]
    dev = C.c_int(-1)
    for r in range(2):                                               # a handle remembers its device; views answer from any thread / current device
        assert tables[r].num_rows == shards[r].num_rows
    comm = C.c_void_p()
    _lib.check(lib.dfgpu_comm_init_all(C.byref(comm)))
    world, first, n_local = C.c_int(), C.c_int(), C.c_int()
    _lib.check(lib.dfgpu_comm_info(comm, C.byref(world), C.byref(first), C.byref(n_local)))
    assert (world.value, first.value, n_local.value) == (2, 0, 2)
    ins = (C.c_void_p * 2)(tables[0].handle, tables[1].handle)
    # ---- RepartitionExec(Hash) across the two GPUs: dfgpu_partition's routing, every sender's rows in its own order
    outs = (C.c_void_p * 2)()
    _lib.check(lib.dfgpu_exchange_hash(comm, ins, (C.c_int * 1)(0), 1, outs))
    got = [DeviceTable(C.c_void_p(outs[r])) for r in range(2)]
    strings = lambda t: pa.table({c: (t.column(c).cast(pa.string()) if c == "s" else t.column(c)) for c in t.column_names})
    for r in range(2):
        exp = pa.concat_tables([strings(oracle.hash_partition(strings(shards[src]), ["k"], 2)[0][r]) for src in range(2)])
        assert_tables_equal(strings(got[r].to_arrow()), exp, ordered=True)
        _lib.check(lib.dfgpu_get_device(C.byref(dev)))
    # ---- CollectLeft's build side: all-gather, then pruned by the destinations' probe-key bounds
    _lib.check(lib.dfgpu_exchange_broadcast(comm, ins, outs))
    whole = pa.concat_tables([strings(s) for s in shards])
    for r in range(2):
        t = DeviceTable(C.c_void_p(outs[r]))
        assert_tables_equal(strings(t.to_arrow()), whole, ordered=True)
        t.free()
    probes_host = [pa.table({"pk": pa.array(np.arange(lo, hi, dtype=np.int64))}) for lo, hi in ((0, 1_000), (3_000, 4_500))]
    probes = [_on(lib, r, lambda r=r: DeviceTable.from_arrow(probes_host[r])) for r in range(2)]
    pins = (C.c_void_p * 2)(probes[0].handle, probes[1].handle)
    _lib.check(lib.dfgpu_exchange_broadcast_pruned(comm, ins, 0, pins, 0, outs))
    for r, (lo, hi) in enumerate(((0, 999), (3_000, 4_499))):
        t = DeviceTable(C.c_void_p(outs[r]))
        k = whole.column("k").to_numpy()
        exp = whole.filter(pa.array((k >= lo) & (k <= hi)))
        assert_tables_equal(strings(t.to_arrow()), exp, ordered=True)
        # the local join on device r over what it received
        j = _on(lib, r, lambda: ops.hash_join(t, probes[r], [("k", "pk")], "Inner", build_cols=["v"], probe_cols=["pk"]).to_arrow())
        je = oracle.hash_join(exp.select(["k", "v"]), probes_host[r], [("k", "pk")], "Inner").select(["v", "pk"])
        assert_tables_equal(j, je, ordered=False)
        t.free()
    # ---- the specialised (hiprtc) fused aggregate on BOTH devices: one compile, one module load per device
    ops.set_options(jit="1", jit__min_rows="0", jit__strict="1")
    try:
        before = ops.jit_cache_stats()["modules_loaded"]
        results = []
        for r in range(2):
            def q1_on_this_device():
                li = ops.tpch_lineitem(0.01)                         # generated, aggregated and exported with device r current
                out = ops.aggregate(li, queries.Q1_GROUP_BY, queries.q1_aggs_inlined(), "Single", predicate=col("l_shipdate") <= lit(queries.DATE_Q1, pa.date32())).to_arrow()
                li.free()
                return out
            results.append(_on(lib, r, q1_on_this_device))
        assert results[0].to_pylist() == results[1].to_pylist() and results[0].num_rows == 4
        loaded = ops.jit_cache_stats()["modules_loaded"] - before
        assert loaded >= 2 and loaded % 2 == 0                        # the same code objects, loaded on device 0 AND on device 1
    finally:
        ops.reset_options()
    st = _lib.ExchangeStats()
    _lib.check(lib.dfgpu_comm_stats(comm, C.byref(st), 0))
    assert st.bytes_sent_to_peers > 0 and st.bytes_sent_to_peers == st.bytes_received_from_peers and st.collectives >= 3
    for t in got + tables + probes:
        t.free()
    _lib.check(lib.dfgpu_comm_free(comm))
