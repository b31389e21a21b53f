"""Pin the CPU oracle's hash join against the reference's own known-answer tests
(datafusion/physical-plan/src/joins/hash_join/exec.rs snapshot tests, extracted by
tests/golden/extract_reference_goldens.py).  Mirrors the reference's rstest matrix
`hash_join_exec_configs` (exec.rs:2929-2963): every case runs with the perfect-hash
(ArrayMap) path allowed and with it forced off."""
import pytest

from oracle import oracle
from tests.util import i32_table, load_golden, rows, sorted_rows

CASES = load_golden("hash_join_exec.json")


@pytest.mark.parametrize("mode", [0, 1], ids=["phj_auto", "hash_map"])
@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_oracle_matches_reference_snapshot(case, mode):
    left = i32_table(case["left"]["columns"], case["left"]["data"], case["left"]["repeat"])
    right = i32_table(case["right"]["columns"], case["right"]["data"], case["right"]["repeat"])
    out = oracle.hash_join(left, right, [tuple(p) for p in case["on"]], case["join_type"], case["null_equality"], mode=mode)
    assert out.column_names == case["expected_columns"], case["source"]
    expected = [tuple(r) for r in case["expected_rows"]]
    if case["ordered"]:
        # "Inner join output is expected to preserve both inputs order" (exec.rs:3349)
        assert rows(out) == expected, case["source"]
    else:
        key = lambda row: tuple((v is None, 0 if v is None else v) for v in row)
        assert sorted_rows(out) == sorted(expected, key=key), case["source"]


def test_array_map_gating_follows_reference():
    """try_create_array_map (exec.rs:111-191): dense / small range -> ArrayMap, sparse -> hash map"""
    import pyarrow as pa
    dense = pa.table({"k": pa.array(range(0, 4000, 2), type=pa.int64())})       # range 3998, density 0.5
    sparse = pa.table({"k": pa.array(range(0, 400000, 200), type=pa.int64())})  # range ~4e5, density 0.005
    probe = pa.table({"k": pa.array([0, 2, 3, 200], type=pa.int64())})
    *_, used = oracle.hash_join(dense, probe, [("k", "k")], return_indices=True)
    assert used
    *_, used = oracle.hash_join(sparse, probe, [("k", "k")], return_indices=True)
    assert not used
    small = pa.table({"k": pa.array([5, 900], type=pa.int64())})                # range < 1024 -> ArrayMap
    *_, used = oracle.hash_join(small, probe, [("k", "k")], return_indices=True)
    assert used


def test_stream_doc_example():
    """lookup_join_hashmap doc example (hash_join/stream.rs:352-393): LEFT.b1 = RIGHT.b2 ->
    build indices 4, 5, 6, 6 / probe indices 3, 3, 4, 5"""
    import pyarrow as pa
    build = pa.table({"a1": pa.array([1, 3, 5, 7, 9, 11, 13], type=pa.int32()),
                      "b1": pa.array([1, 3, 5, 7, 8, 8, 10], type=pa.int32()),
                      "c1": pa.array([10, 30, 50, 70, 90, 110, 130], type=pa.int32())})
    probe = pa.table({"a2": pa.array([2, 4, 6, 8, 10, 12], type=pa.int32()),
                      "b2": pa.array([2, 4, 6, 8, 10, 10], type=pa.int32()),
                      "c2": pa.array([20, 40, 60, 80, 100, 120], type=pa.int32())})
    for mode in (0, 1):
        bi, pi, _, _ = oracle.hash_join(build, probe, [("b1", "b2")], return_indices=True, mode=mode)
        assert list(bi) == [4, 5, 6, 6] and list(pi) == [3, 3, 4, 5]
    out = oracle.hash_join(build, probe, [("b1", "b2")])
    assert rows(out) == [(9, 8, 90, 8, 8, 80), (11, 8, 110, 8, 8, 80), (13, 10, 130, 10, 10, 100), (13, 10, 130, 12, 10, 120)]


def _pairs(build_keys, probe_keys, typ, mode=0):
    """(probe index, build index) pairs of an Inner join on one key column, in emission order, + whether the ArrayMap was used"""
    import pyarrow as pa
    b = pa.table({"k": pa.array(build_keys, type=typ)})
    p = pa.table({"k": pa.array(probe_keys, type=typ)})
    bi, pi, _, used = oracle.hash_join(b, p, [("k", "k")], return_indices=True, mode=mode)
    return list(zip(pi.tolist(), bi.tolist())), used


def test_array_map_known_answers():
    """ArrayMap's own unit tests (joins/array_map.rs:428-599): matches come out in probe order, the matches of one probe row in
    ascending build order; misses, duplicates on the build side, keys beyond the mapped range, NULL probe keys, negative keys.
    (The tests page through `get_matched_indices_with_limit_offset`; the concatenation of their pages is what is pinned here.)"""
    import pyarrow as pa
    i32, i64, u64 = pa.int32(), pa.int64(), pa.uint64()
    cases = [
        ([1, 1, 2], [1, 2], i32, [(0, 0), (0, 1), (1, 2)]),                                   # :429 limit_offset_duplicate_elements
        ([1, 2], [10, 1, 2], i32, [(1, 0), (2, 1)]),                                          # :460 with_limit_and_misses
        ([1, 1], [10, 1, 20, 1], i32, [(1, 0), (1, 1), (3, 0), (3, 1)]),                      # :493 build_duplicates_and_misses (first three within its limit)
        (list(range(11)), [3, (1 << 32) + 3, 11, None], u64, [(0, 3)]),                       # :519 rejects_large_out_of_range_probe_key
        ([-5, 0, 5, -2, 3, 10], [0, -5, 10, -1], i64, [(0, 1), (1, 0), (2, 5)]),              # :563 i64_with_negative_and_positive_numbers
    ]
    for build, probe, typ, want in cases:
        got, used = _pairs(build, probe, typ)
        assert used and got == want, (build, probe, got)
        got_hm, used_hm = _pairs(build, probe, typ, mode=1)                                   # the chained hash map gives the same pairs
        assert not used_hm and sorted(got_hm) == sorted(want)


def test_hash_map_skips_null_probe_keys():
    """JoinHashMap's unit tests (joins/join_hash_map.rs:518-572): a probe row whose key is NULL matches nothing — not even through a
    chain of duplicate build keys — while valid probe rows still match their whole chain.  (At map level the chain of the second test
    is walked last-inserted-first, [3, 1]; HashJoinExec inserts the build side in reverse so that joins emit ascending build rows —
    the pairs are compared as a set here.)"""
    import pyarrow as pa
    for mode in (0, 1):
        got, _ = _pairs([10, 20, 30], [10, None, 30], pa.int64(), mode)
        assert got == [(0, 0), (2, 2)]
        got, _ = _pairs([10, 20, 10, 20], [None, 20], pa.int64(), mode)
        assert sorted(got) == [(1, 1), (1, 3)]


def test_equal_rows_known_answers():
    """equal_rows_arr's unit tests (joins/utils.rs:4625-4700), at join level: two-column keys (the Utf8 column as dictionary codes)
    and NULL keys under both NullEquality settings — every candidate pair the reference test keeps is a join match here, every
    pair it drops is not"""
    import pyarrow as pa
    code = {"a": 0, "b": 1, "c": 2, "d": 3}
    left = pa.table({"a": pa.array([1, 2, 2, 3], pa.int32()), "b": pa.array([code[x] for x in "abcd"], pa.uint8())})
    right = pa.table({"a": pa.array([2, 2, 3, 4], pa.int32()), "b": pa.array([code[x] for x in "bdda"], pa.uint8())})
    bi, pi, _, _ = oracle.hash_join(left, right, [("a", "a"), ("b", "b")], return_indices=True)
    assert sorted(zip(bi.tolist(), pi.tolist())) == [(1, 0), (3, 2)]            # :4625 filters_candidate_pairs: left [1, 3] / right [0, 2]
    left = pa.table({"k": pa.array([1, None, 2, None], pa.int32())})
    right = pa.table({"k": pa.array([None, 1, 2, None], pa.int32())})
    bi, pi, _, _ = oracle.hash_join(left, right, [("k", "k")], null_equality="NullEqualsNothing", return_indices=True)
    assert sorted(zip(bi.tolist(), pi.tolist())) == [(0, 1), (2, 2)]            # :4666 first half
    bi, pi, _, _ = oracle.hash_join(left, right, [("k", "k")], null_equality="NullEqualsNull", return_indices=True)
    got = sorted(zip(bi.tolist(), pi.tolist()))
    assert {(0, 1), (1, 0), (2, 2), (3, 3)} <= set(got)                         # :4666 second half: all four candidate pairs survive
    assert got == [(0, 1), (1, 0), (1, 3), (2, 2), (3, 0), (3, 3)]              # and the join also pairs the other NULLs
