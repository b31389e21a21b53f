"""Pins the oracle's HashJoinExec-with-JoinFilter restatement against the reference's own snapshot tests
(hash_join/exec.rs join_{inner,left,right,full}_with_filter and join_{left,right}_{semi,anti}_with_filter;
tests/golden/hash_join_filter.json, extracted by tests/golden/extract_reference_filter_goldens.py)."""
import pyarrow as pa
import pytest

from tests.util import i32_table, load_golden, sorted_rows

CASES = load_golden("hash_join_filter.json")


def oracle_filter_of(case):
    f = case["filter"]
    e = f["expr"]
    rhs = ("col", f"f{e['right_col']}") if "right_col" in e else ("lit", e["right_lit"], pa.int32())
    return ("bin", e["op"], ("col", f"f{e['left']}"), rhs), [(i, side) for i, side in f["columns"]]


@pytest.mark.parametrize("mode", [0, 1], ids=["phj_auto", "hash_map"])
@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_oracle_matches_reference_filter_snapshots(case, mode):
    from oracle import oracle
    left = i32_table(case["left"]["columns"], case["left"]["data"])
    right = i32_table(case["right"]["columns"], case["right"]["data"])
    out = oracle.hash_join(left, right, [tuple(p) for p in case["on"]], case["join_type"], case["null_equality"], mode=mode,
                           join_filter=oracle_filter_of(case))
    assert out.column_names == case["expected_columns"], case["source"]
    key = lambda row: tuple((v is None, 0 if v is None else v) for v in row)
    assert sorted_rows(out) == sorted([tuple(r) for r in case["expected_rows"]], key=key), case["source"]
