#!/usr/bin/env python3
"""AVG known answers and state types from the reference's own unit tests.

Reads (read-only) /root/reference/datafusion/functions-aggregate/src/average.rs (`avg_cases`, :1242-1330) and writes
tests/golden/avg_cases.json: the cases over types the device path carries (Float64, Decimal128) with their input values,
return type, sum-state type and expected value; the Decimal128(34,0) case (sum state Decimal256) is recorded as the case
that must be refused.  Runs only in the authoring container; the JSON is committed."""
import json
import os
import re

SRC = "/root/reference/datafusion/functions-aggregate/src/average.rs"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "avg_cases.json")


def main():
    text = open(SRC).read()
    start = text.index("fn avg_cases()")
    ln = text[:start].count("\n") + 1
    body = text[start:text.index("#[test]", start)]
    consts = {m.group(1): int(m.group(2).replace("_", "")) for m in re.finditer(r"const (\w+): \w+ = ([\d_]+);", body)}
    out = []
    for m in re.finditer(r'AvgCase \{\s*name: "(\w+)",(.*?)\n            \},', body, re.S):
        name, b = m.group(1), m.group(2)
        if name not in ("float64", "decimal128", "decimal128_with_headroom"):
            continue
        rt = re.search(r"return_type: DataType::(\w+)(?:\((\d+), (\d+)\))?", b)
        st = re.search(r"sum_type: DataType::(\w+)(?:\((\d+), (\d+)\))?", b)
        rec = dict(name=name, source=f"functions-aggregate/src/average.rs:{ln}", return_type=[rt.group(1)] + [int(x) for x in rt.groups()[1:] if x],
                   sum_type=[st.group(1)] + [int(x) for x in st.groups()[1:] if x])
        if name == "float64":
            rec.update(input_type=["Float64"], values=[10.0, 20.0], expected=15.0)
        elif name == "decimal128":
            rec.update(input_type=["Decimal128", 34, 0], value_repeated=consts["DECIMAL128_VALUE"], rows=consts["DECIMAL128_ROWS"],
                       expected_unscaled=consts["DECIMAL128_VALUE"] * 10_000)
        else:
            vals = [int(x.replace("_", "")) for x in re.search(r"Decimal128Array::from\(vec!\[([\d_, ]+)\]\)", b).group(1).split(",")]
            ps = re.search(r"with_precision_and_scale\((\d+), (\d+)\)", b)
            exp = int(re.search(r"Decimal128\(Some\(([\d_]+)\)", b).group(1).replace("_", ""))
            rec.update(input_type=["Decimal128", int(ps.group(1)), int(ps.group(2))], values_unscaled=vals, expected_unscaled=exp)
        out.append(rec)
    assert [r["name"] for r in out] == ["float64", "decimal128", "decimal128_with_headroom"]
    json.dump(out, open(OUT, "w"), indent=1)
    print("wrote", OUT, out)


if __name__ == "__main__":
    main()
