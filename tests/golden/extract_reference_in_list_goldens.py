#!/usr/bin/env python3
"""Known answers of InListExpr from the reference's own unit tests.

Reads (read-only) /root/reference/datafusion/physical-expr/src/expressions/in_list.rs and writes tests/golden/in_list.json:
  * `run_test_cases` (in_list.rs:725-855): for every typed case the column [value_in, value_not_in, NULL] against the lists
    (value_in, others...) and (value_in, others..., NULL), plain and negated — the four expected vectors are parsed from the
    function body, the per-type data from `in_list_int_types` / `_string_types` / `_date_types` / `_decimal`;
  * `in_list_float64` (NaN / -NaN / NULL list members) and `in_list_bool`: lists, negation flags and expected vectors parsed
    from the `in_list!` invocations.
Types without a device representation (Int8/16, UInt16, binary, timestamps, Date64, views) are not taken.
Runs only in the authoring container; the JSON is committed."""
import json
import os
import re

SRC = "/root/reference/datafusion/physical-expr/src/expressions/in_list.rs"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "in_list.json")


def fn_body(lines, name):
    start = next(i for i, l in enumerate(lines) if re.match(rf"\s*fn {name}\(", l))
    end = next(i for i in range(start + 1, len(lines)) if re.match(r"\s*(#\[test\]|fn \w+|macro_rules!)", lines[i]))
    return start, "\n".join(lines[start:end])


def vec_opt(text):
    return [None if t == "None" else (t == "Some(true)") for t in re.findall(r"Some\(true\)|Some\(false\)|None", text)]


def main():
    lines = open(SRC).read().split("\n")
    out = []
    ln, body = fn_body(lines, "run_test_cases")
    expected = [vec_opt(m) for m in re.findall(r"vec!\[((?:Some\(\w+\)|None|, )+)\],\s*Arc::clone\(&col_a\)", body)]
    assert expected == [[True, False, None], [False, True, None], [True, None, None], [False, None, None], [True, False], [False, True]], expected
    with_nulls, no_nulls = expected[:4], expected[4:]

    def data(fn, var):
        _, b = fn_body(lines, fn)
        m = re.search(rf"let {var} = PrimitiveTestCaseData \{{\s*value_in: (.+?),\s*value_not_in: (.+?),\s*other_list_values: vec!\[(.*?)\],", b, re.S)
        conv = lambda s: s.strip().strip('"') if '"' in s else int(s)   # noqa: E731
        return conv(m.group(1)), conv(m.group(2)), [conv(x) for x in m.group(3).split(",") if x.strip()]
    ints = data("in_list_int_types", "int_data")
    strs = data("in_list_string_types", "string_data")
    dates = data("in_list_date_types", "date_data")
    _, dec = fn_body(lines, "in_list_decimal")
    dvals = [int(x) for x in re.findall(r"Decimal128\(Some\((\d+)\), 10, 2\)", dec)]
    decs = (dvals[0], dvals[1], dvals[2:])
    typed = [("int32", "i32", ints), ("int64", "i64", ints), ("uint8", "u8", ints), ("uint32", "u32", ints), ("uint64", "u64", ints), ("utf8", "utf8", strs),
             ("date32", "date32", dates), ("decimal128", "decimal128(10,2)", decs)]
    for name, typ, (vin, vnot, others) in typed:
        base = [vin] + list(others)
        for k, (lst, neg) in enumerate([(base, False), (base, True), (base + [None], False), (base + [None], True)]):
            out.append(dict(name=f"{name}#{k}", source=f"physical-expr/src/expressions/in_list.rs:{ln + 1} (run_test_cases)", type=typ,
                            column=[vin, vnot, None], list=lst, negated=neg, expected=with_nulls[k]))
    base = [ints[0]] + ints[2]
    for k, neg in enumerate([False, True]):
        out.append(dict(name=f"int32_no_nulls#{k}", source=f"physical-expr/src/expressions/in_list.rs:{ln + 1} (run_test_cases)", type="i32", column=[ints[0], ints[1]],
                        list=base, negated=neg, expected=no_nulls[k]))
    for fn, typ, column in (("in_list_float64", "f64", [0.0, 0.2, None, "NaN", "-NaN"]), ("in_list_bool", "bool", [True, None])):
        ln2, b = fn_body(lines, fn)
        lists = re.findall(r"let list = vec!\[(.*?)\];", b)
        calls = re.findall(r"in_list!\(\s*batch,\s*list,\s*&(true|false),\s*vec!\[(.*?)\]", b, re.S)
        assert len(lists) == len(calls)

        def lit(tok):
            tok = tok.strip()
            m = re.match(r"lit\((.*)\)$", tok)
            v = m.group(1)
            if v == "ScalarValue::Null":
                return None
            if v in ("true", "false"):
                return v == "true"
            if "NAN" in v:
                return "-NaN" if v.startswith("-") else "NaN"
            return float(v.replace("f64", ""))
        for k, (lst, (neg, exp)) in enumerate(zip(lists, calls)):
            items = [lit(x) for x in re.findall(r"lit\([^()]*(?:\([^()]*\))?[^()]*\)", lst)]
            out.append(dict(name=f"{fn}#{k}", source=f"physical-expr/src/expressions/in_list.rs:{ln2 + 1}", type=typ, column=column, list=items,
                            negated=neg == "true", expected=vec_opt(exp)))
    json.dump(out, open(OUT, "w"), indent=1)
    print("wrote", OUT, len(out), "records;", [(r["name"], r["list"], r["negated"], r["expected"]) for r in out if r["type"] in ("f64", "bool")])


if __name__ == "__main__":
    main()
