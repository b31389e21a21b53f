"""Differential fuzzing against the CPU oracle, in the manner of the reference's fuzz_cases
(datafusion/core/tests/fuzz_cases/join_fuzz.rs, aggregate_fuzz.rs, sort_fuzz.rs): random schemas, sizes, key
cardinalities, NULL fractions and operator settings from a seed; results compared as sorted rows (joins),
first-seen-ordered rows (aggregates) and exactly ordered rows (stable sorts)."""
import numpy as np
import pyarrow as pa
import pytest

from tests.test_gpu_aggregate import assert_agg_equal, oracle_agg
from tests.util import assert_tables_equal, random_table, to_oracle_expr

pytestmark = pytest.mark.gpu

JOIN_TYPES = ["Inner", "Left", "Right", "Full", "LeftSemi", "RightSemi", "LeftAnti", "RightAnti", "LeftMark", "RightMark"]
KEY_TYPES = [pa.int64(), pa.int32(), pa.date32()]


@pytest.mark.parametrize("seed", range(24))
def test_join_fuzz(seed):
    from datafusion_amd import ops
    from datafusion_amd.table import DeviceTable
    from oracle import oracle
    rng = np.random.default_rng(1000 + seed)
    kt = KEY_TYPES[rng.integers(len(KEY_TYPES))]
    nb, npr = int(rng.integers(0, 4000)), int(rng.integers(0, 9000))
    card = int(rng.integers(1, 6000))                       # few distinct keys => long duplicate chains, many => mostly unique
    lo = int(rng.integers(-3000, 3000))
    nulls = float(rng.choice([0.0, 0.0, 0.05, 0.3]))
    two_keys = bool(rng.integers(0, 4) == 0)
    bspec = {"a": (kt, lo, lo + card), "x": (pa.decimal128(15, 2), -10**6, 10**6), "y": (pa.int32(), 0, 50)}
    pspec = {"b": (kt, lo - 20, lo + card + 20), "z": (pa.float64(), -1000, 1000), "w": (pa.int32(), 0, 50)}
    left, right = random_table(rng, nb, bspec, nulls), random_table(rng, npr, pspec, nulls)
    on = [("a", "b")] + ([("y", "w")] if two_keys else [])
    jt = JOIN_TYPES[rng.integers(len(JOIN_TYPES))]
    ne = str(rng.choice(["NullEqualsNothing", "NullEqualsNull"]))
    mode = int(rng.integers(0, 3)) if not two_keys else int(rng.integers(0, 2))   # auto / hash map / array map (single key only)
    if mode == 2 and ne == "NullEqualsNull" and nulls > 0:
        mode = 0                                                                   # direct-address tables cannot hold NULL == NULL
    got = ops.hash_join(DeviceTable.from_arrow(left), DeviceTable.from_arrow(right), on, jt, ne, table_mode=mode).to_arrow()
    exp = oracle.hash_join(left, right, on, jt, ne)
    assert_tables_equal(got, exp)


AGG_POOL = [("sum", "d"), ("avg", "d"), ("min", "d"), ("max", "d"), ("sum", "i"), ("avg", "i"), ("min", "i"), ("max", "dt"),
            ("sum", "f"), ("avg", "f"), ("min", "f"), ("max", "f"), ("count", "i"), ("count", None)]


@pytest.mark.parametrize("evaluator", ["specialised", "interpreted", "column_at_a_time"])
@pytest.mark.parametrize("seed", range(12))
def test_aggregate_fuzz(seed, evaluator):
    import os

    from datafusion_amd import ops
    from datafusion_amd.expr import col, lit
    from datafusion_amd.table import DeviceTable
    from oracle import oracle as O
    rng = np.random.default_rng(2000 + seed)
    n = int(rng.integers(1, 30_000))
    groups = int(rng.choice([1, 3, 40, 2000, 20_000]))
    nulls = float(rng.choice([0.0, 0.1]))
    spec = {"k": (pa.int64(), -groups // 2, groups // 2 + 1), "k2": (pa.int32(), 0, 3), "d": (pa.decimal128(15, 2), -10**9, 10**9), "i": (pa.int32(), -1000, 1000),
            "f": (pa.float64(), -10**6, 10**6), "dt": (pa.date32(), 8000, 10000)}
    t = random_table(rng, n, spec, nulls)
    flag = pa.array(rng.integers(65, 70, size=n).astype(np.uint8))
    t = t.append_column("rf", flag)
    shape = int(rng.integers(0, 4))
    gb = [[(col("k"), "k")], [(col("rf"), "rf")], [(col("k"), "k"), (col("k2"), "k2")], []][shape]
    picks = rng.choice(len(AGG_POOL), size=int(rng.integers(1, 6)), replace=False)
    aggs = [(f, None if c is None else col(c), f"{f}_{c}_{j}") for j, (f, c) in enumerate(AGG_POOL[p] for p in picks)]
    pred = None if rng.integers(0, 3) == 0 else (col("i") > lit(int(rng.integers(-900, 900)), pa.int32()))
    ops.set_fusion(evaluator != "column_at_a_time")
    if evaluator == "specialised":
        ops.set_options(jit="1", jit__min_rows="0", jit__strict="1")
    else:
        ops.set_options(jit="0")
    try:
        got = ops.aggregate(DeviceTable.from_arrow(t), gb, aggs, "Single", predicate=pred).to_arrow()
    finally:
        ops.set_fusion(True)
        ops.reset_options()
    src = t if pred is None else O.filter(t, to_oracle_expr(pred), t.column_names)
    assert_agg_equal(got, oracle_agg(src, gb, aggs, "Single"), ordered=True)


@pytest.mark.parametrize("seed", range(16))
def test_sort_fuzz(seed):
    from datafusion_amd import ops
    from datafusion_amd.table import DeviceTable
    from oracle import oracle
    rng = np.random.default_rng(3000 + seed)
    n = int(rng.integers(0, 40_000))
    spec = {"k": (pa.int64(), -int(rng.choice([3, 500, 10**12])), int(rng.choice([3, 500, 10**12]))), "d": (pa.decimal128(15, 2), -10**6, 10**6),
            "q": (pa.int32(), -5, 5), "f": (pa.float64(), -100, 100), "dt": (pa.date32(), 9000, 9100), "c": (pa.uint8(), 0, 4)}
    t = random_table(rng, n, spec, float(rng.choice([0.0, 0.1])))
    cols = list(rng.choice(list(spec), size=int(rng.integers(1, 4)), replace=False))
    keys = [(c, bool(rng.integers(0, 2)), bool(rng.integers(0, 2))) for c in cols]
    fetch = None if rng.integers(0, 2) == 0 else int(rng.integers(0, max(1, n)))
    got = ops.sort(DeviceTable.from_arrow(t), keys, fetch).to_arrow()
    assert_tables_equal(got, oracle.sort(t, keys, fetch), ordered=True)    # both sides are stable: ties compare position by position


# ------------------------------------------------------------------------------------ expressions
def _gen_expr(rng, typ, depth):
    """random well-typed PhysicalExpr of arrow type `typ` over the columns of EXPR_SPEC (the planner's coercions are
    already applied: both operands of an arithmetic / comparison node have the same type)"""
    from datafusion_amd.expr import case, col, lit
    leaves = {"i32": ["i", "j"], "i64": ["k", "m"], "dec": ["d", "e"], "f64": ["f", "g"], "date": ["dt"]}
    if depth > 0 and rng.integers(0, 6) == 0:   # CaseExpr of this type: one or two WHENs, ELSE optional
        bd = 0 if typ == "dec" else depth - 1          # decimal branches stay leaves: every branch Decimal128(15,2)
        whens = [(_gen_expr(rng, "bool", depth - 1), _gen_expr(rng, typ, bd)) for _ in range(int(rng.integers(1, 3)))]
        return case(whens, None if rng.integers(0, 3) == 0 else _gen_expr(rng, typ, bd))
    if typ == "bool":
        kind = rng.integers(0, 6) if depth > 0 else 0
        if kind <= 2:      # comparison of two same-typed values
            t = str(rng.choice(["i32", "i64", "dec", "f64", "date"]))
            a, b = _gen_expr(rng, t, depth - 1), _gen_expr(rng, t, depth - 1)
            op = str(rng.choice(["=", "!=", "<", "<=", ">", ">="]))
            return {"=": a.eq(b), "!=": a.ne(b), "<": a < b, "<=": a <= b, ">": a > b, ">=": a >= b}[op]
        if kind == 3:
            return _gen_expr(rng, "bool", depth - 1).and_(_gen_expr(rng, "bool", depth - 1))
        if kind == 4:
            return _gen_expr(rng, "bool", depth - 1).or_(_gen_expr(rng, "bool", depth - 1))
        inner = _gen_expr(rng, str(rng.choice(["i32", "dec", "bool"])), depth - 1)
        return inner.is_null() if typ != "bool" or rng.integers(0, 2) else (inner.not_() if _is_bool(inner) else inner.is_not_null())
    if depth <= 0 or rng.integers(0, 3) == 0:
        if typ != "date" and rng.integers(0, 4) == 0:
            return {"i32": lambda: lit(int(rng.integers(-50, 50)), pa.int32()), "i64": lambda: lit(int(rng.integers(-10**6, 10**6))),
                    "dec": lambda: lit(f"{int(rng.integers(-999, 999))}.{int(rng.integers(0, 100)):02d}", pa.decimal128(15, 2)),
                    "f64": lambda: lit(float(rng.integers(-100, 100)) / 8.0)}[typ]()
        return col(str(rng.choice(leaves[typ])))
    if typ == "date":
        return col("dt")
    if typ == "i64" and rng.integers(0, 4) == 0:
        return _gen_expr(rng, "i32", depth - 1).cast(pa.int64())
    if typ == "f64" and rng.integers(0, 4) == 0:
        return _gen_expr(rng, str(rng.choice(["i32", "i64"])), depth - 1).cast(pa.float64())
    a, b = _gen_expr(rng, typ, depth - 1), _gen_expr(rng, typ, depth - 1)
    op = str(rng.choice(["+", "-", "*"])) if typ != "dec" else str(rng.choice(["+", "-"]))   # decimal products: one level, below
    return {"+": a + b, "-": a - b, "*": a * b}[op]


def _is_bool(e):
    from datafusion_amd.expr import BinaryExpr, CaseExpr, IsNotNullExpr, IsNullExpr, NotExpr
    if isinstance(e, CaseExpr):
        return _is_bool(e.when_then[0][1])
    return isinstance(e, (IsNullExpr, IsNotNullExpr, NotExpr)) or (isinstance(e, BinaryExpr) and e.op in ("=", "!=", "<", "<=", ">", ">=", "and", "or"))


EXPR_SPEC = {"i": (pa.int32(), -1000, 1000), "j": (pa.int32(), -40000, 40000), "k": (pa.int64(), -10**9, 10**9), "m": (pa.int64(), -5, 5),
             "d": (pa.decimal128(15, 2), -10**9, 10**9), "e": (pa.decimal128(15, 2), 0, 11), "f": (pa.float64(), -10**6, 10**6), "g": (pa.float64(), -3, 3),
             "dt": (pa.date32(), 8000, 10000)}


@pytest.mark.parametrize("seed", range(20))
def test_expression_fuzz_projection_filter_and_fused_arguments(seed):
    """random expression forests: ProjectionExec / FilterExec column-at-a-time and as aggregate arguments of the fused
    node (register program, specialised HIP source) vs the oracle's evaluator — bit-exact, Float64 sums 1e-6"""
    import os

    from datafusion_amd import ops
    from datafusion_amd.expr import col
    from datafusion_amd.table import DeviceTable
    from oracle import oracle as O
    rng = np.random.default_rng(4000 + seed)
    n = int(rng.integers(1, 20_000))
    t = random_table(rng, n, EXPR_SPEC, float(rng.choice([0.0, 0.15])))
    dev = DeviceTable.from_arrow(t)
    exprs = [(_gen_expr(rng, str(rng.choice(["i32", "i64", "dec", "f64", "bool"])), int(rng.integers(1, 4))), f"x{j}") for j in range(4)]
    exprs.append((col("d") * (_gen_expr(rng, "dec", 1)), "prod"))                 # one decimal product (precision 31+)
    got = ops.project(dev, exprs).to_arrow()
    exp = O.project(t, [(to_oracle_expr(e), nm) for e, nm in exprs])
    assert_tables_equal(got, exp, ordered=True)
    pred = _gen_expr(rng, "bool", int(rng.integers(1, 4)))
    assert_tables_equal(ops.filter(dev, pred).to_arrow(), O.filter(t, to_oracle_expr(pred), t.column_names), ordered=True)
    # the same forest inside the fused aggregate node
    numeric = [(e, nm) for e, nm in exprs if not _is_bool(e)]
    aggs = [("sum", e, f"s_{nm}") for e, nm in numeric] + [("count", e, f"c_{nm}") for e, nm in exprs[:2] if not _is_bool(e)] + [("count", None, "n")]
    src = O.filter(t, to_oracle_expr(pred), t.column_names)
    want = oracle_agg(src, [], aggs, "Single")
    try:
        for opts in (dict(jit=1, jit__min_rows=0, jit__strict=1), dict(jit=0)):
            ops.set_options(**opts)
            assert_agg_equal(ops.aggregate(dev, [], aggs, "Single", predicate=pred).to_arrow(), want)
    finally:
        ops.reset_options()


@pytest.mark.parametrize("seed", range(20))
def test_single_match_probe_flavours_fuzz(seed):
    """the at-most-one-match probes (unique build keys; Inner / RightSemi / RightAnti) under random sizes, build densities, fused
    FilterExec selectivities, key types and probe modes: whichever flavour the library picks — single pass with a cursor, tile counts
    + placed, tile counts + hit words + listed rows — the rows are the oracle's (in probe order whenever the order is promised)"""
    from datafusion_amd import ops
    from datafusion_amd.expr import col, lit
    from datafusion_amd.table import DeviceTable
    from oracle import oracle
    from tests.util import to_oracle_expr
    rng = np.random.default_rng(7000 + seed)
    kt = [pa.int32(), pa.int64()][int(rng.integers(2))]
    span = int(rng.integers(10, 400_000))
    nb = int(rng.integers(1, max(2, int(span * float(rng.choice([0.02, 0.1, 0.5, 1.0]))))))
    npr = int(rng.integers(1, 300_000))
    bkeys = rng.permutation(span)[:nb] if rng.integers(2) else np.sort(rng.permutation(span)[:nb])
    build = pa.table({"a": pa.array(bkeys, type=kt), "x": pa.array(rng.integers(0, 10**6, nb), type=pa.int64()), "y": pa.array(rng.integers(0, 99, nb).astype(np.int32))})
    pk = rng.integers(-5, span + 5, npr)
    if rng.integers(2):
        pk = np.sort(pk)                                   # clustered hits
    null_keys = bool(rng.integers(0, 3) == 0)
    probe = pa.table({"b": pa.array(pk, type=kt, mask=(rng.random(npr) < 0.05) if null_keys else None), "p": pa.array(rng.integers(0, 10**9, npr), type=pa.int64()),
                      "f": pa.array(rng.integers(0, 100, npr).astype(np.int32))})
    pred = None if rng.integers(3) == 0 else col("f") < lit(int(rng.integers(0, 101)), pa.int32())
    jt = ["Inner", "RightSemi", "RightAnti"][int(rng.integers(3))]
    mode = [0, 3, 4][int(rng.integers(3))]
    pcols = ["p", "f"] if null_keys else ["b", "p"]        # (a nullable payload column would take the general path)
    ht = ops.JoinHashTable(DeviceTable.from_arrow(build), ["a"], probe_mode=mode)
    ops.profile_enable(True)
    ops.profile_reset()
    try:
        got = ht.probe(DeviceTable.from_arrow(probe), ["b"], jt, ["y", "x"], pcols, predicate=pred).to_arrow()
        names = set(ops.profile_stats())
    finally:
        ops.profile_enable(False)
    src = probe if pred is None else oracle.filter(probe, to_oracle_expr(pred), probe.column_names)
    exp = oracle.hash_join(build, src, [("a", "b")], jt).select((["y", "x"] if jt == "Inner" else []) + pcols)
    assert_tables_equal(got, exp, ordered="join_probe_fused" not in names)
    ht.free()
