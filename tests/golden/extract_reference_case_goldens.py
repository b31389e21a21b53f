#!/usr/bin/env python3
"""Known answers of CaseExpr from the reference's own unit tests.

Reads (read-only) /root/reference/datafusion/physical-expr/src/expressions/case.rs and writes
tests/golden/case_expr.json.  Inputs (the tests' `case_test_batch*` builders, case.rs:2181-2214) and the expressions
are restated below by test name; the EXPECTED arrays are parsed from the test bodies, so a changed reference shows up
as a changed fixture.  `CASE x WHEN v` forms are recorded as written (`"base"`) — the test lowers them to `x = v`
conditions, the form DFGPU_EXPR_CASE carries.  Runs only in the authoring container; the JSON is committed.
"""
import json
import os
import re

SRC = "/root/reference/datafusion/physical-expr/src/expressions/case.rs"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "case_expr.json")

BATCH = {"a": {"type": "utf8", "values": ["foo", "baz", None, "bar"]}}                                   # case_test_batch
# case_test_batch_nulls: raw values + validity byte 0b00101001 (rows 1, 2 are NULL over a raw 1.77)
NULLS = {"load4": {"type": "f64", "values": [1.77, 1.77, 1.77, 1.78, 0.0, 1.77], "valid": [1, 0, 0, 1, 0, 1]}}
A_EQ = lambda s: ["=", ["col", "a"], ["lit", s, "utf8"]]                                                 # noqa: E731
I32 = lambda v: ["lit", v, "i32"]                                                                        # noqa: E731
TESTS = {
    "case_with_expr": dict(batch=BATCH, base=["col", "a"], whens=[[["lit", "foo", "utf8"], I32(123)], [["lit", "bar", "utf8"], I32(456)]], else_=None),
    "case_with_expr_else": dict(batch=BATCH, base=["col", "a"], whens=[[["lit", "foo", "utf8"], I32(123)], [["lit", "bar", "utf8"], I32(456)]], else_=I32(999)),
    "case_without_expr": dict(batch=BATCH, base=None, whens=[[A_EQ("foo"), I32(123)], [A_EQ("bar"), I32(456)]], else_=None),
    "case_without_expr_else": dict(batch=BATCH, base=None, whens=[[A_EQ("foo"), I32(123)], [A_EQ("bar"), I32(456)]], else_=I32(999)),
    "case_with_expr_when_null": dict(batch=BATCH, base=["col", "a"], whens=[[["lit", None, "utf8"], I32(0)], [["col", "a"], I32(123)]], else_=I32(999)),
    "case_with_matches_and_nulls": dict(batch=NULLS, base=None, whens=[[["=", ["col", "load4"], ["lit", 1.77, "f64"]], ["col", "load4"]]], else_=None),
    "case_with_scalar_predicate": dict(batch=NULLS, base=None, whens=[[["lit", True, "bool"], ["col", "load4"]]], else_=None),
    "case_expr_matches_and_nulls": dict(batch=NULLS, base=["col", "load4"], whens=[[["lit", 1.77, "f64"], ["col", "load4"]]], else_=None),
}


def main():
    lines = open(SRC).read().split("\n")
    starts = [(i, m.group(1)) for i, l in enumerate(lines) for m in [re.match(r"\s*fn (\w+)\(", l)] if m]
    out = []
    for k, (ln, name) in enumerate(starts):
        if name not in TESTS:
            continue
        end = starts[k + 1][0] if k + 1 < len(starts) else len(lines)
        body = "\n".join(lines[ln:end])
        m = re.search(r"(Int32|Float64)Array::from\(vec!\[(.*?)\]\)", body, re.S)
        vals = []
        for tok in re.findall(r"Some\(([-\d.]+)\)|(None)", m.group(2)):
            vals.append(None if tok[1] else (float(tok[0]) if m.group(1) == "Float64" else int(tok[0])))
        rec = dict(name=name, source=f"physical-expr/src/expressions/case.rs:{ln + 1}", expected=vals, expected_type="f64" if m.group(1) == "Float64" else "i32")
        rec.update(TESTS[name])
        out.append(rec)
    # test_when_null_and_some_cond_else_null (case.rs:2155-2179): CASE WHEN (NULL AND a = 'foo') THEN a ELSE NULL END -> all NULL
    ln = [i for i, n in starts if n == "test_when_null_and_some_cond_else_null"][0]
    out.append(dict(name="test_when_null_and_some_cond_else_null", source=f"physical-expr/src/expressions/case.rs:{ln + 1}", batch=BATCH, base=None,
                    whens=[[["and", ["lit", None, "bool"], A_EQ("foo")], ["col", "a"]]], else_=None, expected=[None, None, None, None], expected_type="utf8"))
    assert len(out) == len(TESTS) + 1, [r["name"] for r in out]
    json.dump(out, open(OUT, "w"), indent=1)
    print("wrote", OUT, [(r["name"], r["expected"]) for r in out])


if __name__ == "__main__":
    main()
