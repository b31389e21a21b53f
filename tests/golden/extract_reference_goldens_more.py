#!/usr/bin/env python3
"""Second batch of known-answer vectors from the reference's HashJoinExec tests — the shapes the first extractor
(extract_reference_goldens.py) does not parse: inputs made of several batches / partitions, empty sides, one test body
looping over every join type, null_aware (NOT IN) anti joins, Date32 / Int64 keys, forced hash collisions.

Reads (read-only)  /root/reference/datafusion/physical-plan/src/joins/hash_join/exec.rs
and writes         tests/golden/hash_join_exec_more.json   (same record layout as hash_join_exec.json plus the optional
                   keys "types", "null_aware", "force_hash_collisions", "expected_num_rows")

Runs only in the authoring container; the JSON is committed."""
import datetime
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from extract_reference_goldens import balanced, parse_table, parse_vec  # noqa: E402

SRC = "/root/reference/datafusion/physical-plan/src/joins/hash_join/exec.rs"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hash_join_exec_more.json")

TABLE_CALL = re.compile(r"\bbuild_table(?:_i32|_two_cols|_two_batches)?\(")


def tables_in(body):
    """every build_table*( ... ) call of a test body, in source order, as [(column, values)]"""
    out = []
    for m in TABLE_CALL.finditer(body):
        call = balanced(body, m.end() - 1)
        cols = re.findall(r'\(\s*"(\w+)"\s*,\s*&vec!\[([^\]]*)\]\s*,?\s*\)', call)
        if cols:
            out.append([(n, parse_vec(v)) for n, v in cols])
    return out


def vstack(tables):
    names = [n for n, _ in tables[0]]
    return [(n, sum((dict(t)[n] for t in tables), [])) for n in names]


def snapshot(body):
    """(ordered, header, rows) of the expectation in `body`; an empty `++ ++` snapshot gives (.., [], []).  A test that
    executes the probe partitions one by one holds one snapshot per partition: their rows are concatenated."""
    parts = re.findall(r'assert_snapshot!\((batches_to_sort_string|batches_to_string)\(&batches\), @r"(.*?)"\);', body, re.S)
    if len(parts) > 1:
        tabs = [parse_table([l for l in txt.split("\n") if l.strip().startswith("|")]) for _, txt in parts]
        assert all(t[0] == tabs[0][0] for t in tabs)
        return all(kind == "batches_to_string" for kind, _ in parts), tabs[0][0], sum((t[1] for t in tabs), [])
    m = re.search(r'assert_snapshot!\((batches_to_sort_string|batches_to_string)\(&batches\), @r"(.*?)"\);', body, re.S)
    if m:
        lines = [l for l in m.group(2).split("\n") if l.strip().startswith("|")]
        if not lines:
            return m.group(1) == "batches_to_string", [], []
        return (m.group(1) == "batches_to_string",) + parse_table(lines)
    m = re.search(r"let expected = \[(.*?)\];", body, re.S)
    if m:
        return (False,) + parse_table([s.strip().strip('",') for s in m.group(1).split("\n")])
    raise ValueError("no expectation")


def record(name, line, left, right, on, join_type, null_equality, ordered, header, rows, **extra):
    r = {"name": name, "source": f"datafusion/physical-plan/src/joins/hash_join/exec.rs:{line}",
         "left": {"columns": [n for n, _ in left], "data": [v for _, v in left], "repeat": 1},
         "right": {"columns": [n for n, _ in right], "data": [v for _, v in right], "repeat": 1},
         "on": on, "join_type": join_type, "null_equality": null_equality, "ordered": ordered,
         "expected_columns": header, "expected_rows": rows}
    r.update(extra)
    return r


def on_of(body):
    on = re.findall(r'Column::new_with_schema\("(\w+)", &(left|right)', body)
    lon = [n for n, s in on if s == "left"]
    ron = [n for n, s in on if s == "right"]
    return [list(p) for p in zip(lon, ron)]


def output_columns(join_type, left, right):
    l, r = [n for n, _ in left], [n for n, _ in right]
    return {"LeftSemi": l, "LeftAnti": l, "RightSemi": r, "RightAnti": r, "LeftMark": l + ["mark"], "RightMark": r + ["mark"]}.get(join_type, l + r)


# test name -> (indices of the build_table* calls that make up the left side, ... the right side)
SIMPLE = {
    "join_inner_one_two_parts_left": ([0, 1], [2]),
    "join_inner_one_two_parts_right": ([0], [1, 2]),
    "join_left_empty_right": ([0], [1]),
    "join_full_empty_right": ([0], [1]),
    "join_right_mark": ([0], [1]),
    "partitioned_join_right_mark": ([0], [1]),
    "test_null_aware_anti_join_probe_null": ([0], [1]),
    "test_null_aware_anti_join_build_null": ([0], [1]),
    "test_null_aware_anti_join_no_nulls": ([0], [1]),
    "test_null_aware_right_anti_build_null": ([0], [1]),
    "test_null_aware_right_anti_probe_null": ([0], [1]),
    "test_null_aware_right_anti_no_nulls": ([0], [1]),
    "test_null_aware_right_anti_empty_build": ([0], [1]),
}


def main():
    src = open(SRC).read()
    lines = src.split("\n")
    fn = re.compile(r"\s*(?:async )?fn (\w+)\(")
    starts = [(i, fn.match(l).group(1)) for i, l in enumerate(lines) if fn.match(l)]
    bodies = {}
    for k, (ln, name) in enumerate(starts):
        end = starts[k + 1][0] if k + 1 < len(starts) else len(lines)
        bodies.setdefault(name, (ln + 1, "\n".join(lines[ln:end])))
    records = []

    for name, (li, ri) in SIMPLE.items():
        line, body = bodies[name]
        tabs = tables_in(body)
        left, right = vstack([tabs[i] for i in li]), vstack([tabs[i] for i in ri])
        jts, nes = set(re.findall(r"JoinType::(\w+)", body)), set(re.findall(r"NullEquality::(\w+)", body))
        assert len(jts) == 1 and len(nes) == 1, (name, jts, nes)
        jt = jts.pop()
        ordered, header, rows = snapshot(body)
        extra = {}
        if "null_aware" in name:
            assert re.search(r"true,\s*(//[^\n]*)?\s*\)\?;", body), name   # the `null_aware` argument of try_new
            extra["null_aware"] = True
        if not header:
            header = output_columns(jt, left, right)   # an empty snapshot prints no header
        records.append(record(name, line, left, right, on_of(body), jt, nes.pop(), ordered, header, rows, **extra))

    # one body, every join type: all build-side keys NULL (exec.rs `join_all_null_build_keys`)
    line, body = bodies["join_all_null_build_keys"]
    tabs = tables_in(body)
    arms = re.split(r"\n\s*JoinType::(\w+) => \{", body)
    for jt, arm in zip(arms[1::2], arms[2::2]):
        ordered, header, rows = snapshot(arm)
        records.append(record(f"join_all_null_build_keys/{jt}", line, tabs[0], tabs[1], on_of(body), jt, "NullEqualsNothing", ordered, header, rows))
    for jt in ("Inner", "LeftSemi", "RightSemi"):   # `assert_eq!(num_rows, 0, ...)` arm
        records.append(record(f"join_all_null_build_keys/{jt}", line, tabs[0], tabs[1], on_of(body), jt, "NullEqualsNothing", False,
                              output_columns(jt, tabs[0], tabs[1]), []))

    # Date32 keys
    line, body = bodies["join_date32"]
    dates = [parse_vec(v) for v in re.findall(r"Date32Array::from\(vec!\[([^\]]*)\]\)", body)]
    ns = [parse_vec(v) for v in re.findall(r"Int32Array::from\(vec!\[([^\]]*)\]\)", body)]
    ordered, header, rows_txt = None, None, None
    m = re.search(r'assert_snapshot!\(batches_to_sort_string\(&batches\), @r"(.*?)"\);', body, re.S)
    tl = [l.strip() for l in m.group(1).split("\n") if l.strip().startswith("|")]
    header = [c.strip() for c in tl[0].strip("|").split("|")]
    epoch = datetime.date(1970, 1, 1)
    rows = []
    for l in tl[1:]:
        cells = [c.strip() for c in l.strip("|").split("|")]
        rows.append([(datetime.date.fromisoformat(c) - epoch).days if "-" in c else int(c) for c in cells])
    records.append(record("join_date32", line, [("date", dates[0]), ("n", ns[0])], [("date", dates[1]), ("n", ns[1])], [["date", "date"]],
                          "Inner", "NullEqualsNothing", False, header, rows, types={"date": "date32"}))

    # i64::MIN / i64::MAX keys: the ArrayMap range computation must not overflow; the test asserts the row count only
    line, body = bodies["test_perfect_hash_join_overflow_full_int64_range"]
    assert "Int64Array::from(vec![i64::MIN, i64::MAX])" in body and "assert_eq!(total_rows, 2)" in body
    keys = [-(1 << 63), (1 << 63) - 1]
    records.append(record("test_perfect_hash_join_overflow_full_int64_range", line, [("a", keys)], [("a", keys)], [["a", "a"]], "Inner",
                          "NullEqualsNothing", False, ["a", "a"], [[k, k] for k in keys], types={"a": "int64"}, expected_num_rows=2))

    # every key in one hash chain: only the key re-check keeps the pairs right (the map-level tests assert the
    # (build, probe) index pairs [0, 1] / [0, 1]; here they are turned into the rows those pairs select)
    for name in ("join_with_hash_collisions_64", "join_with_hash_collisions_u32"):
        line, body = bodies[name]
        tabs = tables_in(body)
        l_ids = parse_vec(re.search(r"let left_ids: UInt64Array = vec!\[([^\]]*)\]", body).group(1))
        r_ids = parse_vec(re.search(r"let right_ids: UInt32Array = vec!\[([^\]]*)\]", body).group(1))
        rows = [[v[i] for _, v in tabs[0]] + [v[j] for _, v in tabs[1]] for i, j in zip(l_ids, r_ids)]
        records.append(record(name, line, tabs[0], tabs[1], [["a", "a"]], "Inner", "NullEqualsNothing", True,
                              [n for n, _ in tabs[0]] + [n for n, _ in tabs[1]], rows, force_hash_collisions=True))

    json.dump(records, open(OUT, "w"), indent=1)
    print(f"wrote {len(records)} cases to {OUT}")
    for r in records:
        print(" ", r["name"], r["join_type"], len(r["expected_rows"]), "rows")


if __name__ == "__main__":
    main()
