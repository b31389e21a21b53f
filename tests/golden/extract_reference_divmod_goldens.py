#!/usr/bin/env python3
"""Known answers of BinaryExpr Divide / Modulo from the reference's own unit tests.

Reads (read-only) /root/reference/datafusion/physical-expr/src/expressions/binary.rs and writes tests/golden/binary_expr_divmod.json:
`divide_op`, `modulus_op`, `divide_op_scalar`, `modulus_op_scalar` (Int32), `divide_op_dict_decimal` / `modulus_op_dict_decimal` (the
dictionary arrays resolved to plain Decimal128 columns: the GPU path sees them after the shim's unpacking), the Divide and Modulo
legs of `arithmetic_decimal_expr_test` (Int32 coerced to Decimal128(10,0) against Decimal128(10,2)) and `arithmetic_divide_zero`.
Arrays are parsed from the test bodies.  Runs only in the authoring container; the JSON is committed."""
import json
import os
import re

SRC = "/root/reference/datafusion/physical-expr/src/expressions/binary.rs"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "binary_expr_divmod.json")


def body_of(lines, name):
    start = next(i for i, l in enumerate(lines) if re.match(rf"\s*fn {name}\(\)", l))
    end = next(i for i in range(start + 1, len(lines)) if re.match(r"\s*(#\[test\]|fn \w+)", lines[i]))
    return start, "\n".join(lines[start:end])


def opt_ints(text, env=None):
    """`Some(value + 2), None, Some(9919), // comment` -> [125, None, 9919]"""
    env = env or {}
    text = re.sub(r"//[^\n]*", "", text)
    out = []
    for tok in re.findall(r"None|Some\(([^)]*)\)|(-?\d+)", text):
        pass
    for m in re.finditer(r"None|Some\(([^)]*)\)|(?<![\w(])(-?\d+)(?![\w)])", text):
        if m.group(0) == "None":
            out.append(None)
        else:
            e = m.group(1) if m.group(1) is not None else m.group(2)
            out.append(int(eval(e, {}, env)))
    return out


def main():
    lines = open(SRC).read().split("\n")
    out = []
    src = lambda ln: f"physical-expr/src/expressions/binary.rs:{ln + 1}"
    for name, op in (("divide_op", "/"), ("modulus_op", "%")):
        ln, b = body_of(lines, name)
        arrays = [[int(x) for x in m.split(",") if x.strip()] for m in re.findall(r"Int32Array::from\(vec!\[([-\d, ]+)\]\)", b)]
        out.append(dict(name=name, source=src(ln), op=op, a=dict(type="int32", values=arrays[0]), b=dict(type="int32", values=arrays[1]),
                        expected_type="int32", expected=arrays[2]))
    for name, op in (("divide_op_scalar", "/"), ("modulus_op_scalar", "%")):
        ln, b = body_of(lines, name)
        arrays = [[int(x) for x in m.split(",") if x.strip()] for m in re.findall(r"Int32Array::from\(vec!\[([-\d, ]+)\]\)", b)]
        scalar = int(re.search(r"ScalarValue::Int32\(Some\((-?\d+)\)\)", b).group(1))
        out.append(dict(name=name, source=src(ln), op=op, a=dict(type="int32", values=arrays[0]), b_scalar=dict(type="int32", value=scalar),
                        expected_type="int32", expected=arrays[1]))
    for name, op in (("divide_op_dict_decimal", "/"), ("modulus_op_dict_decimal", "%")):
        ln, b = body_of(lines, name)
        value = int(re.search(r"let value = (\d+);", b).group(1))
        decs = re.findall(r"create_decimal_array\(\s*&\[(.*?)\],\s*(\d+),\s*(\d+),?\s*\)", b, re.S)
        keys = re.findall(r"Int8Array::from\(vec!\[(.*?)\]\)", b, re.S)
        assert len(decs) == 3 and len(keys) == 2
        va, vb, ve = (opt_ints(d[0], {"value": value}) for d in decs)
        ka, kb = (opt_ints(k) for k in keys)
        ta, tb, te = (f"decimal128({d[1]},{d[2]})" for d in decs)
        res = lambda ks, vs: [None if k is None else vs[k] for k in ks]
        out.append(dict(name=name, source=src(ln), op=op, a=dict(type=ta, unscaled=res(ka, va)), b=dict(type=tb, unscaled=res(kb, vb)),
                        expected_type=te, expected_unscaled=ve))
    # arithmetic_decimal_expr_test: the divide and modulus legs (operands declared at the top of the test)
    ln, b = body_of(lines, "arithmetic_decimal_expr_test")
    value = int(re.search(r"let value: i128 = (\d+);", b).group(1))
    dec = re.search(r"let decimal_array = Arc::new\(create_decimal_array\(\s*&\[(.*?)\],\s*(\d+),\s*(\d+),?\s*\)\)", b, re.S)
    ints = re.search(r"let int32_array = Arc::new\(Int32Array::from\(vec!\[(.*?)\]\)\)", b, re.S)
    dvals, ivals = opt_ints(dec.group(1), {"value": value}), opt_ints(ints.group(1))
    for label, op in (("divide", "/"), ("modulus", "%")):
        seg = b[b.index(f"// {label}: int32 array {label} decimal array"):]
        exp = re.search(r"create_decimal_array\(\s*&\[(.*?)\],\s*(\d+),\s*(\d+),?\s*\)", seg, re.S)
        out.append(dict(name=f"arithmetic_decimal_expr_test/{label}", source=src(ln) + " (Int32 coerced to Decimal128(10,0), expr-common/src/type_coercion/binary.rs:1257-1273)",
                        op=op, a=dict(type="int32", values=ivals), cast_a="decimal128(10,0)", b=dict(type=f"decimal128({dec.group(2)},{dec.group(3)})", unscaled=dvals),
                        expected_type=f"decimal128({exp.group(2)},{exp.group(3)})", expected_unscaled=opt_ints(exp.group(1))))
    ln, b = body_of(lines, "arithmetic_divide_zero")
    assert "Divide by zero" in b
    out.append(dict(name="arithmetic_divide_zero/int32", source=src(ln), op="/", a=dict(type="int32", values=[100]), b=dict(type="int32", values=[0]), error="Divide by zero"))
    out.append(dict(name="arithmetic_divide_zero/decimal", source=src(ln), op="/", a=dict(type="decimal128(25,3)", unscaled=[1234567]),
                    b=dict(type="decimal128(25,3)", unscaled=[0]), error="Divide by zero"))
    json.dump(out, open(OUT, "w"), indent=1)
    for r in out:
        print(r["name"], r.get("expected_type"), r.get("expected", r.get("expected_unscaled", r.get("error"))))


if __name__ == "__main__":
    main()
