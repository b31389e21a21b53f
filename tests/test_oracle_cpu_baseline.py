"""bench.py's cpu_baseline leg (oracle/dforacle.c orc_partitioned_q3_join): the reference's plan for the benchmark's join —
RepartitionExec(Hash) of every column of both sides (repartition/mod.rs:1111-1150), HashJoinExec(Partitioned)
(hash_join/exec.rs:1314-1324), build_batch_from_indices of the five Q3 payload columns (joins/utils.rs:1332-1386) — gives the
rows the plain oracle join gives, whatever the partition count."""
import numpy as np
import pyarrow as pa
import pytest

from datafusion_amd import tpch
from oracle import oracle


def _dec(col):
    a = col.combine_chunks()
    return np.frombuffer(a.buffers()[1], dtype=np.uint64).reshape(-1, 2)


@pytest.mark.parametrize("threads", [1, 2, 7, 16])
def test_partitioned_q3_join_equals_the_single_partition_oracle_join(threads):
    sf = 0.02
    o, l = tpch.orders(sf), tpch.lineitem(sf)
    # a build side in no key order and a probe side with dangling keys: not only the FK -> PK shape of the benchmark
    rng = np.random.default_rng(3)
    o = o.take(pa.array(rng.permutation(o.num_rows)))
    drop = rng.random(l.num_rows) < 0.1
    lk = np.where(drop, l.column("l_orderkey").to_numpy() + 8, l.column("l_orderkey").to_numpy())   # key + 8 is never an order key
    l = l.set_column(l.schema.get_field_index("l_orderkey"), "l_orderkey", pa.array(lk))
    rows, chk = oracle.partitioned_q3_join(o.column("o_orderkey").to_numpy(), o.column("o_orderdate").cast(pa.int32()).to_numpy(),
                                           o.column("o_shippriority").to_numpy(), lk, _dec(l.column("l_extendedprice")), _dec(l.column("l_discount")), threads)
    exp = oracle.hash_join(o.select(["o_orderkey", "o_orderdate", "o_shippriority"]), l.select(["l_orderkey", "l_extendedprice", "l_discount"]),
                           [("o_orderkey", "l_orderkey")], "Inner")
    assert rows == exp.num_rows == int((~drop).sum())
    u = lambda name: np.frombuffer(exp.column(name).combine_chunks().buffers()[1], dtype=np.uint64).reshape(-1, 2)[:, 0]
    want = (exp.column("o_orderdate").cast(pa.int32()).to_numpy().astype(np.int64).astype(np.uint64) + np.uint64(3) * exp.column("o_shippriority").to_numpy().astype(np.uint64)
            + np.uint64(5) * exp.column("l_orderkey").to_numpy().astype(np.uint64) + np.uint64(7) * u("l_extendedprice") + np.uint64(11) * u("l_discount")).sum(dtype=np.uint64)
    assert chk == int(want)
