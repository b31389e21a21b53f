"""Arrow IPC scan, the host half (no GPU): dfgpu_ipc_open walks the encapsulated messages of files and streams pyarrow wrote — a
hand-written reader of the flatbuffers metadata (Message / Schema / Field / RecordBatch, csrc/ipc.hip) — and must find the same schema,
batches and row counts as pyarrow's own reader; what the GPU path does not take is an error, not a wrong answer."""
from decimal import Decimal

import numpy as np
import pyarrow as pa
import pyarrow.ipc as ipc
import pytest


def sample(n, seed=2):
    rng = np.random.default_rng(seed)
    return pa.table({
        "i64": pa.array(rng.integers(-10**12, 10**12, n)), "i32n": pa.array(rng.integers(-5, 5, n).astype(np.int32), mask=rng.random(n) < 0.2),
        "u8": pa.array(rng.integers(0, 255, n).astype(np.uint8)), "u32": pa.array(rng.integers(0, 2**32 - 1, n).astype(np.uint32)), "u64": pa.array(rng.integers(0, 2**62, n).astype(np.uint64)),
        "f": pa.array(rng.random(n)), "dt": pa.array(rng.integers(8000, 11000, n).astype(np.int32), pa.int32()).cast(pa.date32()),
        "dec": pa.array([Decimal(int(x)) / 100 for x in rng.integers(-10**9, 10**9, n)], pa.decimal128(15, 2)),
        "b": pa.array(rng.random(n) < 0.5, mask=rng.random(n) < 0.1), "s": pa.array([None if i % 7 == 0 else "row-%d" % (i % 313) for i in range(n)], pa.string()),
        "ls": pa.array(["L%d" % (i % 11) for i in range(n)], pa.large_string()), "sv": pa.array(["a rather long string view value %d" % i if i % 3 else "short" for i in range(n)], pa.string_view()),
        "dict": pa.array([["x", "yy", "zzz"][i % 3] for i in range(n)], pa.string()).dictionary_encode()})


def write(t, path, fmt, compression=None, chunk=4000):
    opts = ipc.IpcWriteOptions(compression=compression) if compression else None
    with (ipc.new_stream(path, t.schema, options=opts) if fmt == "stream" else ipc.new_file(path, t.schema, options=opts)) as w:
        for b in t.to_batches(max_chunksize=chunk):
            w.write_batch(b)


@pytest.mark.parametrize("compression", [None, "zstd", "lz4"])
@pytest.mark.parametrize("fmt", ["file", "stream"])
def test_open_finds_what_pyarrow_wrote(tmp_path, fmt, compression):
    from datafusion_amd.ipc import IpcFile
    t = sample(10_001)
    path = str(tmp_path / "t.arrow")
    write(t, path, fmt, compression)
    f = IpcFile(path)
    assert f.is_file_format == (fmt == "file") and f.num_record_batches == 3 and f.num_columns == t.num_columns
    assert [f.batch_rows(i) for i in range(3)] == [4000, 4000, 2001]
    assert f.column_names == t.column_names
    by_name = {n: (fmt_, nullable, dic) for n, fmt_, nullable, dic in f.columns}
    assert by_name["i64"][0] == "l" and by_name["i32n"][0] == "i" and by_name["u8"][0] == "C" and by_name["u32"][0] == "I" and by_name["u64"][0] == "L"
    assert by_name["f"][0] == "g" and by_name["dt"][0] == "tdD" and by_name["dec"][0] == "d:15,2" and by_name["b"][0] == "b"
    assert by_name["s"][0] == "u" and by_name["ls"][0] == "U" and by_name["sv"][0] == "vu"
    assert by_name["dict"] == ("u", True, True) and not by_name["s"][2]
    sch = f.schema
    for name in ("i64", "f", "dt", "dec", "b", "s", "ls", "sv"):
        assert sch.field(name).type == t.schema.field(name).type, name
    f.close()


def test_what_the_gpu_scan_does_not_take_is_an_error(tmp_path):
    from datafusion_amd import _lib
    from datafusion_amd.ipc import IpcFile
    path = str(tmp_path / "n.arrow")
    for t, msg in ((pa.table({"l": pa.array([[1, 2], [3]], pa.list_(pa.int64()))}), "nested"),
                   (pa.table({"f": pa.array([1.5, 2.5], pa.float32())}), "Float64"),
                   (pa.table({"ts": pa.array([1, 2], pa.timestamp("us"))}), "type id 10"),
                   (pa.table({"d": pa.array([1, 2], pa.int64()).dictionary_encode()}), "string dictionaries")):
        write(t, path, "file")
        with pytest.raises(_lib.DfgpuError, match=msg):
            IpcFile(path)
    open(path, "wb").write(b"definitely not arrow ipc bytes, just text")
    with pytest.raises(_lib.DfgpuError, match="arrow ipc"):
        IpcFile(path)
    write(sample(100), path, "file")
    data = open(path, "rb").read()
    open(path, "wb").write(data[: len(data) // 2])              # a truncated file: an error or fewer batches, never a read past the end
    try:
        f = IpcFile(path)
        assert f.num_record_batches <= 1
        f.close()
    except _lib.DfgpuError as e:
        assert "arrow ipc" in str(e)
