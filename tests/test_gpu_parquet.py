"""Scan -> device on the GPU: every column chunk of files written by pyarrow — three codecs x data page
v1 / v2 x dictionary on / off, small pages, several row groups, NULLs — decoded by dfgpu_parquet_decode_chunk and compared
with pyarrow's own reader (an independent production decoder of the same bytes), bit for bit."""
import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq
import pytest

from tests.parquet_cases import WRITER_MATRIX, case_id, sample_table, write
from tests.util import assert_tables_equal

pytestmark = pytest.mark.gpu


def plain(t: pa.Table) -> pa.Table:
    """dictionary-encoded string columns -> strings (comparison form)"""
    return pa.table({n: (c.cast(pa.string()) if pa.types.is_dictionary(c.type) else c) for n, c in zip(t.column_names, t.columns)})


@pytest.mark.parametrize("writer", WRITER_MATRIX, ids=[case_id(w) for w in WRITER_MATRIX])
def test_decode_matches_pyarrow(tmp_path, writer):
    from datafusion_amd.parquet import read_table
    t = sample_table(40_000)
    path = write(t, tmp_path, "t.parquet", data_page_size=8 * 1024, row_group_size=17_000, **writer)
    got = read_table(path).to_arrow()
    assert pa.types.is_dictionary(got.schema.field("s").type)
    assert_tables_equal(plain(got), pq.read_table(path), ordered=True)


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 1000])
def test_small_tables_and_word_boundaries(tmp_path, n):
    from datafusion_amd.parquet import read_table
    t = sample_table(n, seed=n)
    path = write(t, tmp_path, "s.parquet", compression="snappy", row_group_size=max(1, n // 3 + 1))   # row groups that split validity words
    assert_tables_equal(plain(read_table(path).to_arrow()), pq.read_table(path), ordered=True)


def test_decimals_stored_as_integers_and_column_projection(tmp_path):
    from decimal import Decimal

    from datafusion_amd.parquet import read_table
    rng = np.random.default_rng(3)
    n = 20_000
    t = pa.table({"d9": pa.array([Decimal(int(x)) / 100 for x in rng.integers(-10**6, 10**6, n)], pa.decimal128(9, 2)),
                  "d18": pa.array([Decimal(int(x)) / 1000 for x in rng.integers(-10**15, 10**15, n)], pa.decimal128(18, 3)),
                  "k": pa.array(rng.integers(0, 10**12, n))})
    path = str(tmp_path / "d.parquet")
    pq.write_table(t, path, store_decimal_as_integer=True, compression="zstd")
    meta = pq.ParquetFile(path).metadata.row_group(0)
    assert meta.column(0).physical_type == "INT32" and meta.column(1).physical_type == "INT64"
    assert_tables_equal(read_table(path).to_arrow(), t, ordered=True)
    assert_tables_equal(read_table(path, ["k", "d9"]).to_arrow(), t.select(["k", "d9"]), ordered=True)


def test_q6_straight_from_a_parquet_file(tmp_path):
    """dbgen lineitem -> Parquet (ZSTD, the reference's benchmark setting) -> device -> the reference's Q6 plan -> the reference's answer"""
    from datafusion_amd import physical_plan as P, tpch_plans as T
    from datafusion_amd.parquet import read_table
    from oracle import dbgen
    from tests.test_tpch_answers import assert_answer
    _, _, l = dbgen.tables(0.1, "utf8")
    path = str(tmp_path / "lineitem.parquet")
    pq.write_table(l, path, compression="zstd", compression_level=1)
    li = read_table(path, ["l_quantity", "l_extendedprice", "l_discount", "l_shipdate", "l_returnflag", "l_linestatus", "l_tax"])
    assert li.num_rows == l.num_rows
    assert_answer("q6", P.collect(P.GpuOffloadRule().optimize(T.q6_plan(li))).to_arrow())
    assert_answer("q1", P.collect(P.GpuOffloadRule().optimize(T.q1_plan(li))).to_arrow())
    # the same with the scan as the plan's leaf: ParquetExec = DataSourceExec over the file, owned output
    leaf = P.ParquetExec(path, ["l_quantity", "l_extendedprice", "l_discount", "l_shipdate"], "lineitem")
    node = T.q6_plan(li)
    while not isinstance(node, P.FilterExec):
        node = node.children()[0]
    f = P.FilterExec(node.predicate, leaf, projection=["l_extendedprice", "l_discount"])
    name = "sum(lineitem.l_extendedprice * lineitem.l_discount)"
    from datafusion_amd.expr import col
    plan = P.ProjectionExec([(col(name), "revenue")], P.AggregateExec("Single", [], [("sum", col("l_extendedprice") * col("l_discount"), name)], f))
    assert_answer("q6", P.collect(P.GpuOffloadRule().optimize(plan)).to_arrow())
