#!/usr/bin/env python3
"""Known answers of BinaryExpr's Kleene logic and Int32 arithmetic from the reference's own unit tests.

Reads (read-only) /root/reference/datafusion/physical-expr/src/expressions/binary.rs and writes tests/golden/binary_expr_logic.json:
`and_with_nulls_op` / `or_with_nulls_op` (all nine TRUE / FALSE / NULL combinations) and `plus_op`, `minus_op` (both orders),
`multiply_op` on Int32 columns.  Arrays are parsed from the test bodies.  (The Decimal128 known answers of the same file are in
binary_expr_decimal.json.)  Runs only in the authoring container; the JSON is committed."""
import json
import os
import re

SRC = "/root/reference/datafusion/physical-expr/src/expressions/binary.rs"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "binary_expr_logic.json")


def body_of(lines, name):
    start = next(i for i, l in enumerate(lines) if re.match(rf"\s*fn {name}\(\)", l))
    end = next(i for i in range(start + 1, len(lines)) if re.match(r"\s*(#\[test\]|fn \w+)", lines[i]))
    return start, "\n".join(lines[start:end])


def main():
    lines = open(SRC).read().split("\n")
    out = []
    for name, op in (("and_with_nulls_op", "and"), ("or_with_nulls_op", "or")):
        ln, b = body_of(lines, name)
        arrays = [[None if t == "None" else t == "Some(true)" for t in re.findall(r"Some\(true\)|Some\(false\)|None", m)]
                  for m in re.findall(r"BooleanArray::from\(vec!\[(.*?)\]\)", b, re.S)]
        assert len(arrays) == 3 and all(len(a) == 9 for a in arrays)
        out.append(dict(name=name, source=f"physical-expr/src/expressions/binary.rs:{ln + 1}", op=op, type="bool", a=arrays[0], b=arrays[1], expected=arrays[2]))
    for name, op in (("plus_op", "+"), ("minus_op", "-"), ("multiply_op", "*")):
        ln, b = body_of(lines, name)
        arrays = [[int(x) for x in m.split(",") if x.strip()] for m in re.findall(r"Int32Array::from\(vec!\[([-\d, ]+)\]\)", b)]
        a, bb = arrays[0], arrays[1]
        out.append(dict(name=name, source=f"physical-expr/src/expressions/binary.rs:{ln + 1}", op=op, type="i32", a=a, b=bb, expected=arrays[2]))
        if name == "minus_op":   # second call: operands swapped ("should handle have negative values in result")
            out.append(dict(name="minus_op_swapped", source=f"physical-expr/src/expressions/binary.rs:{ln + 1}", op=op, type="i32", a=bb, b=a, expected=arrays[3]))
    json.dump(out, open(OUT, "w"), indent=1)
    print("wrote", OUT, [(r["name"], r["expected"]) for r in out])


if __name__ == "__main__":
    main()
