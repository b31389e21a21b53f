#!/usr/bin/env python3
"""The reference's GROUPING SETS known answer.

Reads (read-only) /root/reference/datafusion/physical-plan/src/aggregates/mod.rs: `check_grouping_sets` (:3428-3590) — the grouping sets
(a, NULL), (NULL, b), (a, b) with COUNT(1) over `some_data()` (the input of aggregate_check_aggregates.json), the Partial snapshot of the
non-spilling run and the Final snapshot — and writes tests/golden/aggregate_grouping_sets.json.  Runs only in the authoring container;
the JSON is committed."""
import json
import os
import re

SRC = "/root/reference/datafusion/physical-plan/src/aggregates/mod.rs"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "aggregate_grouping_sets.json")


def table(snapshot: str):
    lines = [l.strip() for l in snapshot.strip().splitlines() if l.strip().startswith("|")]
    header = [c.strip() for c in lines[0].strip("|").split("|")]
    rows = []
    for l in lines[1:]:
        cells = [c.strip() for c in l.strip("|").split("|")]
        rows.append([None if c == "" else (float(c) if "." in c else int(c)) for c in cells])
    return header, rows


def main():
    text = open(SRC).read()
    start = text.index("async fn check_grouping_sets(")
    ln = text[:start].count("\n") + 1
    body = text[start:text.index("/// build the aggregates on the data from some_data()", start)]
    groups = [[x == "true" for x in re.findall(r"true|false", m)] for m in re.findall(r"vec!\[((?:true|false), (?:true|false))\]", body)]
    snaps = re.findall(r'@r"\n(.*?)"\s*\n', body, re.S)
    assert len(snaps) == 3, len(snaps)               # spill partial, plain partial, final
    partial_cols, partial_rows = table(snaps[1])
    final_cols, final_rows = table(snaps[2])
    assert groups == [[False, True], [True, False], [False, False]] and len(partial_rows) == 12 and len(final_rows) == 12
    out = dict(source=f"datafusion/physical-plan/src/aggregates/mod.rs:{ln}", groups=groups, null_types=["UInt32", "Float64"],
               partial=dict(columns=partial_cols, rows=partial_rows), final=dict(columns=final_cols, rows=final_rows))
    json.dump(out, open(OUT, "w"), indent=1)
    print("wrote", OUT, partial_cols, final_cols, len(final_rows))


if __name__ == "__main__":
    main()
