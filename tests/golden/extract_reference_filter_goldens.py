#!/usr/bin/env python3
"""Known-answer vectors for HashJoinExec WITH a JoinFilter, from the reference's own tests.

Reads (read-only)  /root/reference/datafusion/physical-plan/src/joins/hash_join/exec.rs
and writes         tests/golden/hash_join_filter.json

Taken: join_{inner,left,right,full}_with_filter (filter = prepare_join_filter(): left.c > right.c,
exec.rs:5556-5582) and join_{left,right}_{semi,anti}_with_filter (two joins each, filter = one intermediate
column `x` compared with an Int32 literal).  Every record carries the source line.  Runs only in the authoring
container; the JSON is committed.
"""
import json
import os
import re

from extract_reference_goldens import SRC, parse_side, parse_table

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hash_join_filter.json")
WANT = ["join_inner_with_filter", "join_left_with_filter", "join_right_with_filter", "join_full_with_filter",
        "join_left_semi_with_filter", "join_right_semi_with_filter", "join_left_anti_with_filter", "join_right_anti_with_filter"]
OPS = {"Gt": ">", "NotEq": "!=", "Lt": "<", "Eq": "=", "GtEq": ">=", "LtEq": "<="}


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
        left, _ = parse_side(body, "left")
        right, _ = parse_side(body, "right")
        on = re.findall(r'Column::new_with_schema\("(\w+)", &(left|right)', body)
        lon = [n for n, s in on if s == "left"][:1]
        ron = [n for n, s in on if s == "right"][:1]
        # one record per join_with_filter( call: the filter in force = the last definition before the call
        for m in re.finditer(r"join_with_filter\(", body):
            before, after = body[:m.start()], body[m.start():]
            jt = re.search(r"&JoinType::(\w+)", after).group(1)
            ne = re.search(r"NullEquality::(\w+)", after).group(1)
            if "prepare_join_filter()" in before and "ColumnIndex" not in before:
                cols = [[2, "Left"], [2, "Right"]]
                expr = {"op": ">", "left": 0, "right_col": 1}
            else:
                ci = re.findall(r"ColumnIndex \{\s*index: (\d+),\s*side: JoinSide::(\w+)", before)
                cols = [[int(ci[-1][0]), ci[-1][1]]]
                em = re.findall(r'Column::new\("x", 0\)\),\s*Operator::(\w+),\s*Arc::new\(Literal::new\(ScalarValue::Int32\(Some\((-?\d+)\)\)\)\)', before)
                expr = {"op": OPS[em[-1][0]], "left": 0, "right_lit": int(em[-1][1])}
            sm = re.search(r'assert_snapshot!\((batches_to_sort_string|batches_to_string)\(&batches\), @r"(.*?)"\);', after, re.S)
            em2 = re.search(r"let expected = \[(.*?)\];", after, re.S)
            if em2 and (not sm or em2.start() < sm.start()):   # `let expected = [ "...", ]; assert_batches_sorted_eq!` form
                header, data = parse_table([q for q in re.findall(r'"([^"]*)"', em2.group(1))])
            else:
                header, data = parse_table(sm.group(2).split("\n"))
            records.append({"name": f"{name}#{len([r for r in records if r['name'].startswith(name)])}", "source": f"hash_join/exec.rs:{ln + 1 + before.count(chr(10))}",
                            "left": {"columns": [c for c, _ in left], "data": [v for _, v in left]},
                            "right": {"columns": [c for c, _ in right], "data": [v for _, v in right]},
                            "on": [[lon[0], ron[0]]], "join_type": jt, "null_equality": ne,
                            "filter": {"columns": cols, "expr": expr},
                            "expected_columns": header, "expected_rows": data})
    json.dump(records, open(OUT, "w"), indent=1)
    print(f"wrote {len(records)} cases to {OUT}")
    for r in records:
        print(" ", r["name"], r["join_type"], r["filter"], len(r["expected_rows"]), "rows", r["source"])


if __name__ == "__main__":
    main()
