"""shared by the Parquet scan tests: sample tables and the writer settings they are written with"""
import itertools
import os
from decimal import Decimal

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq


def sample_table(n: int, seed: int = 1, nulls: bool = True) -> pa.Table:
    rng = np.random.default_rng(seed)
    modes = np.array(["AIR", "FOB", "MAIL", "RAIL", "REG AIR", "SHIP", "TRUCK"], dtype=object)
    cols = {
        "k": pa.array(np.arange(n, dtype=np.int64)),                                         # unique: dictionary falls back to PLAIN when large
        "i": pa.array(rng.integers(-100, 100, n).astype(np.int32)),
        "lowcard": pa.array(rng.integers(0, 5, n).astype(np.int64)),                        # short dictionary, bit-packed runs
        "runs": pa.array(np.repeat(np.arange(n // 100 + 1), 100)[:n].astype(np.int32)),     # long RLE runs
        "f": pa.array(rng.random(n)),
        "d": pa.array([Decimal(int(x)) / 100 for x in rng.integers(-10**9, 10**9, n)], pa.decimal128(15, 2)),   # FIXED_LEN_BYTE_ARRAY(7)
        "big": pa.array([Decimal(int(x)) * 10**20 for x in rng.integers(-10**9, 10**9, n)], pa.decimal128(38, 4)),   # FIXED_LEN_BYTE_ARRAY(16)
        "dt": pa.array(rng.integers(8000, 10000, n).astype(np.int32), pa.int32()).cast(pa.date32()),
        "u8": pa.array(rng.integers(0, 256, n).astype(np.uint8)),
        "s": pa.array(modes[rng.integers(0, len(modes), n)] if n else [], pa.string()),
    }
    if nulls:
        cols["nul"] = pa.array(rng.integers(0, 1000, n).astype(np.int64), mask=rng.random(n) < 0.2)
        cols["nul_d"] = pa.array([Decimal(int(x)) / 100 for x in rng.integers(-10**6, 10**6, n)], pa.decimal128(15, 2), mask=rng.random(n) < 0.5)
        cols["all_null"] = pa.array([None] * n, pa.int32())
        cols["nul_s"] = pa.array(modes[rng.integers(0, 3, n)] if n else [], pa.string(), mask=rng.random(n) < 0.3)
    return pa.table(cols)


WRITER_MATRIX = [dict(compression=c, data_page_version=v, use_dictionary=d)
                 for c, v, d in itertools.product(["none", "snappy", "zstd"], ["1.0", "2.0"], [True, False])]


def case_id(w):
    return f"{w['compression']}-v{w['data_page_version']}-{'dict' if w['use_dictionary'] else 'plain'}"


def write(table: pa.Table, directory, name: str, **writer) -> str:
    path = os.path.join(str(directory), name)
    writer = dict(writer)
    if writer.get("use_dictionary") is False:
        writer["use_dictionary"] = [c for c in table.column_names if pa.types.is_string(table.schema.field(c).type)]   # strings only exist dictionary-encoded on the device
    pq.write_table(table, path, **writer)
    return path
