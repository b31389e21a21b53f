#!/usr/bin/env python3
"""Extract known-answer vectors for the hash-join path from the reference's own tests.

Reads (read-only)  /root/reference/datafusion/physical-plan/src/joins/hash_join/exec.rs
and writes         tests/golden/hash_join_exec.json

Only test functions with the simple, mechanically parseable shape are taken:
  * inputs built with build_table / build_table_two_batches / build_table_two_cols /
    build_semi_anti_{left,right}_table / RecordBatch::try_new(Int32Array::from(vec![..]))
    with build_schema_and_on()
  * exactly one JoinType, one NullEquality, no join filter
  * expectation as insta `assert_snapshot!(.. @r"` table or `assert_batches_sorted_eq!`
Each record carries the source line so the parity tests can cite it.  This script only
runs in the authoring container (the GPU box has no /root/reference); the JSON it emits
is committed.
"""
import json
import os
import re
import sys

SRC = "/root/reference/datafusion/physical-plan/src/joins/hash_join/exec.rs"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hash_join_exec.json")

WANT = [
    "join_inner_one", "partitioned_join_inner_one", "join_inner_one_no_shared_column_names",
    "join_inner_one_randomly_ordered", "join_inner_two", "join_left_multi_batch", "join_full_multi_batch",
    "join_left_one", "partitioned_join_left_one", "join_left_semi", "join_right_semi", "join_left_anti",
    "join_right_anti", "join_right_one", "partitioned_join_right_one", "join_full_one", "join_left_mark",
    "partitioned_join_left_mark", "join_right_mark", "partitioned_join_right_mark",
    "test_perfect_hash_join_with_negative_numbers", "test_phj_null_equals_null_build_no_nulls_probe_has_nulls",
    "test_phj_null_equals_nothing_build_probe_all_have_nulls", "test_phj_null_equals_null_build_have_nulls",
]

SEMI_LEFT = [("a1", [1, 3, 5, 7, 9, 11, 13]), ("b1", [1, 3, 5, 7, 8, 8, 10]), ("c1", [10, 30, 50, 70, 90, 110, 130])]
SEMI_RIGHT = [("a2", [8, 12, 6, 2, 10, 4]), ("b2", [8, 10, 6, 2, 10, 4]), ("c2", [20, 40, 60, 80, 100, 120])]


def parse_vec(txt):
    items = [t.strip() for t in txt.split(",") if t.strip()]
    out = []
    for t in items:
        if t == "None":
            out.append(None)
        elif t.startswith("Some("):
            out.append(int(t[5:-1]))
        else:
            out.append(int(t))
    return out


def parse_build_table(call_txt):
    cols = re.findall(r'\(\s*"(\w+)"\s*,\s*&vec!\[([^\]]*)\]\s*\)', call_txt)
    return [(n, parse_vec(v)) for n, v in cols]


def balanced(txt, start):
    """text of the parenthesised call starting at txt[start] == '('"""
    depth = 0
    for i in range(start, len(txt)):
        if txt[i] == "(":
            depth += 1
        elif txt[i] == ")":
            depth -= 1
            if depth == 0:
                return txt[start:i + 1]
    raise ValueError("unbalanced")


def parse_side(body, side):
    m = re.search(r"let %s = (build_table_two_batches|build_table_two_cols|build_table|build_semi_anti_left_table|build_semi_anti_right_table)\(" % side, body)
    if m:
        fn = m.group(1)
        if fn == "build_semi_anti_left_table":
            return SEMI_LEFT, 1
        if fn == "build_semi_anti_right_table":
            return SEMI_RIGHT, 1
        call = balanced(body, m.end() - 1)
        return parse_build_table(call), (2 if fn == "build_table_two_batches" else 1)
    m = re.search(r"let %s_batch = RecordBatch::try_new\(" % side, body)
    if m:
        call = balanced(body, m.end() - 1)
        vecs = re.findall(r"Int32Array::from\(vec!\[([^\]]*)\]\)", call)
        names = ["a1", "b1"] if side == "left" else ["a2", "b1"]  # build_schema_and_on(), exec.rs:2802-2816
        return [(n, parse_vec(v)) for n, v in zip(names, vecs)], 1
    return None, 0


def parse_table(lines):
    rows = [l.strip() for l in lines if l.strip().startswith("|")]
    header = [c.strip() for c in rows[0].strip("|").split("|")]
    data = []
    for r in rows[1:]:
        cells = [c.strip() for c in r.strip("|").split("|")]
        vals = []
        for c in cells:
            if c == "":
                vals.append(None)
            elif c in ("true", "false"):
                vals.append(c == "true")
            else:
                vals.append(int(c))
        data.append(vals)
    return header, data


def main():
    src = open(SRC).read()
    lines = src.split("\n")
    starts = [(i, re.match(r"\s*async fn (\w+)\(", l).group(1)) for i, l in enumerate(lines) if re.match(r"\s*async fn (\w+)\(", l)]
    records = []
    for k, (ln, name) in enumerate(starts):
        if name not in WANT:
            continue
        end = starts[k + 1][0] if k + 1 < len(starts) else len(lines)
        body = "\n".join(lines[ln:end])
        left, lrep = parse_side(body, "left")
        right, rrep = parse_side(body, "right")
        if not left or not right:
            print("skip (inputs)", name, file=sys.stderr)
            continue
        jts = set(re.findall(r"JoinType::(\w+)", body))
        nes = set(re.findall(r"NullEquality::(\w+)", body))
        if len(jts) != 1 or len(nes) != 1:
            print("skip (join type)", name, jts, nes, file=sys.stderr)
            continue
        on = re.findall(r'Column::new_with_schema\("(\w+)", &(left|right)', body)
        if on:
            lon = [n for n, s in on if s == "left"]
            ron = [n for n, s in on if s == "right"]
        else:
            lon, ron = ["b1"], ["b1"]  # build_schema_and_on()
        m = re.search(r'assert_snapshot!\((batches_to_sort_string|batches_to_string)\(&batches\), @r"(.*?)"\);', body, re.S)
        if m:
            ordered = m.group(1) == "batches_to_string"
            header, data = parse_table(m.group(2).split("\n"))
        else:
            m = re.search(r"assert_batches_sorted_eq!\(\s*\[(.*?)\]", body, re.S)
            if not m:
                print("skip (expectation)", name, file=sys.stderr)
                continue
            ordered = False
            header, data = parse_table([s.strip().strip('",') for s in m.group(1).split("\n")])
        records.append({
            "name": name, "source": f"datafusion/physical-plan/src/joins/hash_join/exec.rs:{ln + 1}",
            "left": {"columns": [n for n, _ in left], "data": [v for _, v in left], "repeat": lrep},
            "right": {"columns": [n for n, _ in right], "data": [v for _, v in right], "repeat": rrep},
            "on": list(zip(lon, ron)), "join_type": jts.pop(), "null_equality": nes.pop(),
            "ordered": ordered, "expected_columns": header, "expected_rows": data,
        })
    json.dump(records, open(OUT, "w"), indent=1)
    print(f"wrote {len(records)} cases to {OUT}")


if __name__ == "__main__":
    main()
