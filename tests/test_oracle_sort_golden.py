"""SortExec known answer from the reference's own test (sorts/sort.rs:2860-2964 test_lex_sort_by_float: two float keys
with NaNs and NULLs, DESC NULLS FIRST then ASC NULLS LAST) for the oracle and, on the GPU box, the device sort."""
import pyarrow as pa
import pytest

from tests.util import load_golden, rows

CASES = load_golden("sort_exec.json")


def table_of(case):
    f = lambda v: float("nan") if v == "NaN" else v
    return pa.table({n: pa.array([f(v) for v in vals], type=pa.float64()) for n, vals in case["columns"].items()})


def expected_rows(case):
    return [tuple(v for v in r) for r in case["expected"]]   # tests.util.rows() spells NaN as "NaN" too


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_oracle_sort_matches_reference(case):
    from oracle import oracle
    out = oracle.sort(table_of(case), [tuple(k) for k in case["keys"]])
    assert rows(out) == expected_rows(case), case["source"]


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_gpu_sort_matches_reference(case):
    from datafusion_amd import ops
    from datafusion_amd.table import DeviceTable
    out = ops.sort(DeviceTable.from_arrow(table_of(case)), [tuple(k) for k in case["keys"]]).to_arrow()
    assert rows(out) == expected_rows(case), case["source"]
    top = ops.sort(DeviceTable.from_arrow(table_of(case)), [tuple(k) for k in case["keys"]], fetch=3).to_arrow()
    assert rows(top) == expected_rows(case)[:3], case["source"]
