"""Host-side mirror of the PhysicalExpr node kinds the GPU path evaluates
(physical-expr-common/src/physical_expr.rs:76; concrete nodes in physical-expr/src/expressions/:
column.rs `Column`, literal.rs `Literal`, cast.rs `CastExpr`, binary.rs `BinaryExpr`,
is_null.rs `IsNullExpr`, is_not_null.rs `IsNotNullExpr`, not.rs `NotExpr`, case.rs `CaseExpr`).

The trees are lowered to the flat `dfgpu_expr_node[]` IR of include/dfgpu.h; typing (decimal
result precision/scale, operand checks) happens inside the library so a Rust shim can pass
trees through unchanged.
"""
from __future__ import annotations

import ctypes as C
import struct
from decimal import Decimal

import pyarrow as pa

from ._lib import Expr, ExprNode, Field
from .table import field_of

# dfgpu_expr_op
OP_COLUMN, OP_LITERAL, OP_CAST = 1, 2, 3
_BINARY = {"+": 10, "-": 11, "*": 12, "/": 13, "%": 14, "=": 20, "!=": 21, "<": 22, "<=": 23, ">": 24, ">=": 25, "and": 30, "or": 31}
OP_NOT, OP_IS_NULL, OP_IS_NOT_NULL = 32, 33, 34
OP_CASE = 40
OP_LIKE, OP_ILIKE = 41, 42
OP_DATE_PART = 50
OP_SUBSTR = 51
_DATE_PARTS = {"year": 0, "month": 1, "day": 2}


class PhysicalExpr:
    def children(self):
        return []

    # operator sugar so tests can write col("a") + lit(1)
    def __add__(self, o): return BinaryExpr(self, "+", _wrap(o))
    def __sub__(self, o): return BinaryExpr(self, "-", _wrap(o))
    def __mul__(self, o): return BinaryExpr(self, "*", _wrap(o))
    def __truediv__(self, o): return BinaryExpr(self, "/", _wrap(o))
    def __mod__(self, o): return BinaryExpr(self, "%", _wrap(o))
    def __rtruediv__(self, o): return BinaryExpr(_wrap(o), "/", self)
    def __rsub__(self, o): return BinaryExpr(_wrap(o), "-", self)
    def __radd__(self, o): return BinaryExpr(_wrap(o), "+", self)
    def __rmul__(self, o): return BinaryExpr(_wrap(o), "*", self)
    def __lt__(self, o): return BinaryExpr(self, "<", _wrap(o))
    def __le__(self, o): return BinaryExpr(self, "<=", _wrap(o))
    def __gt__(self, o): return BinaryExpr(self, ">", _wrap(o))
    def __ge__(self, o): return BinaryExpr(self, ">=", _wrap(o))
    def eq(self, o): return BinaryExpr(self, "=", _wrap(o))
    def ne(self, o): return BinaryExpr(self, "!=", _wrap(o))
    def and_(self, o): return BinaryExpr(self, "and", _wrap(o))
    def or_(self, o): return BinaryExpr(self, "or", _wrap(o))
    def in_list(self, values, negated=False): return InListExpr(self, values, negated)
    def like(self, pattern, negated=False, case_insensitive=False): return LikeExpr(self, pattern, negated, case_insensitive)
    def is_null(self): return IsNullExpr(self)
    def is_not_null(self): return IsNotNullExpr(self)
    def not_(self): return NotExpr(self)
    def cast(self, t): return CastExpr(self, t)


class Column(PhysicalExpr):
    """Column::new(name, index) — index may be omitted and resolved against a schema"""

    def __init__(self, name: str, index: int | None = None):
        self.name, self.index = name, index

    def __repr__(self):
        return f"{self.name}@{self.index}"


class Literal(PhysicalExpr):
    """Literal::new(ScalarValue)"""

    def __init__(self, value, type_: pa.DataType):
        self.value, self.type = value, type_

    def __repr__(self):
        return f"{self.value}:{self.type}"


class ScalarSubqueryResults(list):
    """the container a ScalarSubqueryExec fills and the ScalarSubqueryExprs of its main plan read by index
    (expr/src/physical_planning_context.rs ScalarSubqueryResults)"""
    PENDING = object()

    def __init__(self, n: int = 1):
        super().__init__([ScalarSubqueryResults.PENDING] * n)


class ScalarSubqueryExpr(Literal):
    """physical-expr/src/scalar_subquery.rs ScalarSubqueryExpr: the value of an uncorrelated scalar subquery — a literal whose
    value arrives when the ScalarSubqueryExec above the plan has run the subquery (reading it earlier is an error, as in the
    reference)."""

    def __init__(self, results: ScalarSubqueryResults, index: int, type_: pa.DataType):
        self.results, self.index, self.type = results, index, type_

    @property
    def value(self):
        v = self.results[self.index]
        if v is ScalarSubqueryResults.PENDING:
            raise RuntimeError(f"scalar subquery {self.index} has not been executed yet (no ScalarSubqueryExec above this plan?)")
        return v

    def __repr__(self):
        return f"scalar_subquery(<{'pending' if self.results[self.index] is ScalarSubqueryResults.PENDING else self.results[self.index]}>)"


class CastExpr(PhysicalExpr):
    def __init__(self, expr: PhysicalExpr, cast_type: pa.DataType):
        self.expr, self.cast_type = expr, cast_type

    def children(self):
        return [self.expr]


class BinaryExpr(PhysicalExpr):
    """BinaryExpr::new(left, op, right)"""

    def __init__(self, left: PhysicalExpr, op: str, right: PhysicalExpr):
        if op not in _BINARY:
            raise ValueError(f"operator {op!r} is not supported on the GPU path")
        self.left, self.op, self.right = left, op, right

    def children(self):
        return [self.left, self.right]

    def __repr__(self):
        return f"({self.left!r} {self.op} {self.right!r})"


class IsNullExpr(PhysicalExpr):
    def __init__(self, arg): self.arg = arg
    def children(self): return [self.arg]


class IsNotNullExpr(PhysicalExpr):
    def __init__(self, arg): self.arg = arg
    def children(self): return [self.arg]


class NotExpr(PhysicalExpr):
    def __init__(self, arg): self.arg = arg
    def children(self): return [self.arg]


class CaseExpr(PhysicalExpr):
    """CaseExpr::try_new(expr = None, when_then_expr, else_expr) (expressions/case.rs:274):
    CASE WHEN c1 THEN v1 [WHEN c2 THEN v2 ...] [ELSE e] END.  The `CASE x WHEN v` form is written with conditions x = v."""

    def __init__(self, when_then, else_expr: PhysicalExpr | None = None):
        if not when_then:
            raise ValueError("CASE needs at least one WHEN")
        self.when_then = [(w, _wrap(t)) for w, t in when_then]
        self.else_expr = None if else_expr is None else _wrap(else_expr)

    def children(self):
        return [x for wt in self.when_then for x in wt] + ([] if self.else_expr is None else [self.else_expr])

    def map_children(self, f) -> "CaseExpr":
        return CaseExpr([(f(w), f(t)) for w, t in self.when_then], None if self.else_expr is None else f(self.else_expr))

    def __repr__(self):
        return "CASE " + " ".join(f"WHEN {w!r} THEN {t!r}" for w, t in self.when_then) + (f" ELSE {self.else_expr!r}" if self.else_expr is not None else "") + " END"


class InListExpr(PhysicalExpr):
    """InListExpr::new(expr, list, negated) (expressions/in_list.rs): `x [NOT] IN (v1, v2, ...)`.  Lowered at the boundary to
    the Kleene OR of `x = v_i` (NOT of it when negated), which is SQL's definition and gives the reference's NULL rules: no
    match against a list holding a NULL is NULL, a NULL `x` is NULL (in_list.rs run_test_cases).  The shim does the same for
    the short lists plans carry; long lists (the reference switches to a hash set) stay on the CPU."""

    def __init__(self, expr: PhysicalExpr, list_, negated: bool = False):
        if not list_:
            raise ValueError("IN list must not be empty")
        self.expr, self.list, self.negated = expr, [_wrap(v) for v in list_], negated

    def children(self):
        return [self.expr] + self.list

    def map_children(self, f) -> "InListExpr":
        return InListExpr(f(self.expr), [f(v) for v in self.list], self.negated)

    def lowered(self) -> PhysicalExpr:
        e = BinaryExpr(self.expr, "=", self.list[0])
        for v in self.list[1:]:
            e = BinaryExpr(e, "or", BinaryExpr(self.expr, "=", v))
        return NotExpr(e) if self.negated else e

    def __repr__(self):
        return f"{self.expr!r} {'NOT ' if self.negated else ''}IN ({', '.join(repr(v) for v in self.list)})"


class DatePartExpr(PhysicalExpr):
    """ScalarFunctionExpr date_part(part, expr) (functions/src/datetime/date_part.rs; `EXTRACT(YEAR FROM d)` plans to it):
    YEAR / MONTH / DAY of a Date32 as Int32"""

    def __init__(self, part: str, arg: PhysicalExpr):
        if part.lower() not in _DATE_PARTS:
            raise ValueError(f"date_part({part!r}) is not supported on the GPU path")
        self.part, self.arg = part.lower(), arg

    def children(self):
        return [self.arg]

    def map_children(self, f) -> "DatePartExpr":
        return DatePartExpr(self.part, f(self.arg))

    def __repr__(self):
        return f"date_part({self.part.upper()}, {self.arg!r})"


class SubstrExpr(PhysicalExpr):
    """ScalarFunctionExpr substr(str, start[, count]) (functions/src/unicode/substr.rs; SQL SUBSTRING): characters from the 1-based
    position `start` for `count` characters (to the end without a count) of a Utf8 or dictionary-encoded string column; the result
    is a string column of the same kind (include/dfgpu.h DFGPU_EXPR_SUBSTR)"""

    def __init__(self, arg: PhysicalExpr, start: int, count: int | None = None):
        self.arg, self.start, self.count = arg, int(start), None if count is None else int(count)

    def children(self):
        return [self.arg]

    def map_children(self, f) -> "SubstrExpr":
        return SubstrExpr(f(self.arg), self.start, self.count)

    def __repr__(self):
        return f"substr({self.arg!r}, {self.start}" + ("" if self.count is None else f", {self.count}") + ")"


class LikeExpr(PhysicalExpr):
    """LikeExpr::new(negated, case_insensitive, expr, pattern) (expressions/like.rs): `col [NOT] LIKE 'pattern'` over a
    dictionary-encoded string column.  Bound at the boundary (bind_string_literals): the pattern is matched against the
    column's dictionary by the library (dfgpu_table_dictionary_like) and the predicate becomes comparisons of the index
    column with the matching indices — one `lo <= index <= hi` per contiguous run (a prefix pattern over an ascending
    dictionary is ONE run); a pattern that matches more than MAX_RUNS separate runs goes to the library as it is, which matches it
    against the dictionary and lets the rows look their index up.  NULL rows stay NULL."""
    MAX_RUNS = 16

    def __init__(self, expr: PhysicalExpr, pattern: str, negated: bool = False, case_insensitive: bool = False):
        self.expr, self.pattern, self.negated, self.case_insensitive = expr, pattern, negated, case_insensitive

    def children(self):
        return [self.expr]

    def __repr__(self):
        return f"{self.expr!r} {'NOT ' if self.negated else ''}{'ILIKE' if self.case_insensitive else 'LIKE'} {self.pattern}"

    @staticmethod
    def runs(codes):
        """[(lo, hi)] of the maximal runs of consecutive integers in ascending `codes`"""
        out = []
        for c in codes:
            if out and out[-1][1] + 1 == c:
                out[-1][1] = c
            else:
                out.append([c, c])
        return [tuple(r) for r in out]

    def bound(self, table) -> PhysicalExpr:
        a = self.expr
        if not isinstance(a, Column):
            raise TypeError("LIKE on the GPU path takes a dictionary-encoded string column")
        idx = table.index_of(a.name if a.index is None else a.index)
        itype = table.schema.field(idx).type
        if pa.types.is_string(itype) or pa.types.is_large_string(itype):
            return self        # a Utf8 column in HBM: matched on its bytes by the device (DFGPU_EXPR_LIKE)
        column = Column(a.name, idx)
        runs = LikeExpr.runs(table.dictionary_like(idx, self.pattern, self.case_insensitive))
        if len(runs) > self.MAX_RUNS:
            # many separate runs (a dictionary with one value per row: p_name LIKE '%green%'): the library matches the pattern against
            # the dictionary and the rows look their index up (DFGPU_EXPR_LIKE with a dictionary-encoded left operand)
            return LikeExpr(column, self.pattern, self.negated, self.case_insensitive)
        e = None
        for lo, hi in runs:
            r = BinaryExpr(column, "=", Literal(lo, itype)) if lo == hi else \
                BinaryExpr(BinaryExpr(column, ">=", Literal(lo, itype)), "and", BinaryExpr(column, "<=", Literal(hi, itype)))
            e = r if e is None else BinaryExpr(e, "or", r)
        if e is None:   # nothing matches: FALSE for every non-NULL row (index -1 is held by no row), NULL rows stay NULL
            wide = column if itype == pa.int64() else CastExpr(column, pa.int64())
            e = BinaryExpr(wide, "=", Literal(-1, pa.int64()))
        return NotExpr(e) if self.negated else e


def date_part(part: str, arg: PhysicalExpr) -> DatePartExpr:
    return DatePartExpr(part, arg)


def substr(arg: PhysicalExpr, start: int, count: int | None = None) -> SubstrExpr:
    return SubstrExpr(arg, start, count)


def case(when_then, else_expr=None) -> CaseExpr:
    return CaseExpr(when_then, else_expr)


def col(name, index=None) -> Column:
    return Column(name, index)


def lit(value, type_: pa.DataType | None = None) -> Literal:
    if type_ is None:
        if isinstance(value, bool):
            type_ = pa.bool_()
        elif isinstance(value, int):
            type_ = pa.int64()
        elif isinstance(value, float):
            type_ = pa.float64()
        else:
            raise TypeError("lit(): give an explicit arrow type")
    return Literal(value, type_)


def _wrap(v):
    return v if isinstance(v, PhysicalExpr) else lit(v)


def _literal_bits(value, t: pa.DataType):
    """(lo, hi) 128-bit two's complement of the literal (f64: IEEE bits in lo)"""
    if pa.types.is_float64(t):
        return struct.unpack("<Q", struct.pack("<d", float(value)))[0], 0
    if pa.types.is_decimal128(t):
        v = int(Decimal(str(value)).scaleb(t.scale).to_integral_value())
    elif pa.types.is_date32(t) and not isinstance(value, int):
        v = pa.scalar(value, type=pa.date32()).cast(pa.int32()).as_py()
    elif pa.types.is_boolean(t):
        v = 1 if value else 0
    else:
        v = int(value)
    v &= (1 << 128) - 1
    return v & 0xFFFFFFFFFFFFFFFF, v >> 64


class LoweredExpr:
    """keeps the node array (and the pool of string literals) alive while the C struct points at them"""

    def __init__(self, nodes, pool: bytes = b""):
        self.nodes = (ExprNode * len(nodes))(*nodes)
        self.pool = pool
        self.c = Expr(C.cast(self.nodes, C.POINTER(ExprNode)), len(nodes), len(nodes) - 1, pool if pool else None)


def bind_string_literals(expr: PhysicalExpr, table) -> PhysicalExpr:
    """`dict_col = 'literal'` / `!=` on a dictionary-encoded string column -> comparison of the indices with the
    literal's dictionary index (DeviceTable.dictionary_code).  A string that is not in the dictionary is compared as
    index -1, which no row holds: `=` is then false and `!=` true for every non-NULL row, NULL rows stay NULL — SQL's answer.
    Ordering comparisons on strings are not translated (they would compare indices)."""
    if isinstance(expr, BinaryExpr):
        if expr.op in ("=", "!="):
            for a, b in ((expr.left, expr.right), (expr.right, expr.left)):
                if isinstance(a, Column) and isinstance(b, Literal) and (pa.types.is_string(b.type) or pa.types.is_large_string(b.type)):
                    idx = table.index_of(a.name if a.index is None else a.index)
                    itype = table.schema.field(idx).type
                    if pa.types.is_string(itype) or pa.types.is_large_string(itype):
                        return expr     # a Utf8 column in HBM: compared on its bytes (DFGPU_UTF8), nothing to bind
                    column = Column(a.name, idx)
                    if b.value is None:
                        return BinaryExpr(column, expr.op, Literal(None, itype))
                    code = table.dictionary_code(idx, b.value)
                    if code is not None:
                        return BinaryExpr(column, expr.op, Literal(code, itype))
                    # not in the dictionary: compare the (widened) index with -1, which no row holds
                    wide = column if itype == pa.int64() else CastExpr(column, pa.int64())
                    return BinaryExpr(wide, expr.op, Literal(-1, pa.int64()))
        return BinaryExpr(bind_string_literals(expr.left, table), expr.op, bind_string_literals(expr.right, table))
    if isinstance(expr, CastExpr):
        return CastExpr(bind_string_literals(expr.expr, table), expr.cast_type)
    if isinstance(expr, IsNullExpr):
        return IsNullExpr(bind_string_literals(expr.arg, table))
    if isinstance(expr, IsNotNullExpr):
        return IsNotNullExpr(bind_string_literals(expr.arg, table))
    if isinstance(expr, NotExpr):
        return NotExpr(bind_string_literals(expr.arg, table))
    if isinstance(expr, CaseExpr):
        return expr.map_children(lambda e: bind_string_literals(e, table))
    if isinstance(expr, InListExpr):
        return bind_string_literals(expr.lowered(), table)
    if isinstance(expr, LikeExpr):
        return expr.bound(table)
    if isinstance(expr, (DatePartExpr, SubstrExpr)):
        return expr.map_children(lambda e: bind_string_literals(e, table))
    return expr


class IntermediateSchema:
    """the schema a JoinFilter's expression is typed against (JoinFilter::schema, joins/join_filter.rs): column k = column
    `index` of the build ("Left") or probe ("Right") table.  Quacks like a table for bind_string_literals, so that string literals
    compared with intermediate columns are bound through the dictionary of the column they come from."""

    def __init__(self, build, probe, column_indices):
        self._src = [(build if side == "Left" else probe, int(i)) for i, side in column_indices]
        self.names = [f"f{k}" for k in range(len(self._src))]

    @property
    def schema(self) -> pa.Schema:
        return pa.schema([pa.field(n, t.schema.field(i).type) for n, (t, i) in zip(self.names, self._src)])

    def index_of(self, c) -> int:
        return c if isinstance(c, int) else self.names.index(c)

    def dictionary_code(self, column, value):
        t, i = self._src[self.index_of(column)]
        return t.dictionary_code(i, value)

    def dictionary_like(self, column, pattern, case_insensitive=False):
        t, i = self._src[self.index_of(column)]
        return t.dictionary_like(i, pattern, case_insensitive)


def _has_string_literal(expr) -> bool:
    if isinstance(expr, LikeExpr):
        return True
    if isinstance(expr, Literal):
        return pa.types.is_string(expr.type) or pa.types.is_large_string(expr.type)
    return any(_has_string_literal(c) for c in expr.children())


def lower(expr: PhysicalExpr, column_names, table=None) -> LoweredExpr:
    """post-order flattening; the root is the last node.  With `table`, string literals compared with dictionary-encoded
    columns are bound to dictionary indices first (bind_string_literals)."""
    if table is not None and _has_string_literal(expr):
        expr = bind_string_literals(expr, table)
    nodes: list[ExprNode] = []
    names = list(column_names)
    pool = bytearray()

    def string_literal(n, value):
        n.op = OP_LITERAL
        n.field = field_of(pa.string())
        if value is None:
            n.is_null = 1
        else:
            b = value.encode()
            n.lit_lo, n.lit_hi = len(pool), len(b)
            pool.extend(b)

    def emit(e) -> int:
        n = ExprNode()
        n.left = n.right = -1
        n.column = -1
        if isinstance(e, Column):
            n.op = OP_COLUMN
            if e.index is not None:
                n.column = e.index
            else:
                if names.count(e.name) != 1:
                    raise KeyError(f"column {e.name!r} not found or ambiguous in {names}")
                n.column = names.index(e.name)
        elif isinstance(e, Literal) and (pa.types.is_string(e.type) or pa.types.is_large_string(e.type)):
            string_literal(n, e.value)
        elif isinstance(e, Literal):
            n.op = OP_LITERAL
            n.field = field_of(e.type)
            if e.value is None:
                n.is_null = 1
            else:
                n.lit_lo, n.lit_hi = _literal_bits(e.value, e.type)
        elif isinstance(e, CastExpr):
            n.left = emit(e.expr)
            n.op = OP_CAST
            n.field = field_of(e.cast_type)
        elif isinstance(e, BinaryExpr):
            n.left = emit(e.left)
            n.right = emit(e.right)
            n.op = _BINARY[e.op]
        elif isinstance(e, (IsNullExpr, IsNotNullExpr, NotExpr)):
            n.left = emit(e.arg)
            n.op = {IsNullExpr: OP_IS_NULL, IsNotNullExpr: OP_IS_NOT_NULL, NotExpr: OP_NOT}[type(e)]
        elif isinstance(e, InListExpr):
            return emit(e.lowered())
        elif isinstance(e, DatePartExpr):
            n.left = emit(e.arg)
            n.op = OP_DATE_PART
            n.column = _DATE_PARTS[e.part]
        elif isinstance(e, SubstrExpr):
            n.left = emit(e.arg)
            n.op = OP_SUBSTR
            n.column = e.start
            if e.count is None:
                n.is_null = 1
            else:
                n.lit_lo = e.count & 0xFFFFFFFFFFFFFFFF
                n.lit_hi = 0xFFFFFFFFFFFFFFFF if e.count < 0 else 0
        elif isinstance(e, LikeExpr):
            # over a Utf8 column in HBM (a dictionary-encoded column was rewritten by LikeExpr.bound before this point)
            n.left = emit(e.expr)
            m = ExprNode()
            m.left = m.right = m.column = -1
            string_literal(m, e.pattern)
            nodes.append(m)
            n.right = len(nodes) - 1
            n.op = OP_ILIKE if e.case_insensitive else OP_LIKE
            if e.negated:
                nodes.append(n)
                n = ExprNode()
                n.right = n.column = -1
                n.left = len(nodes) - 1
                n.op = OP_NOT
        elif isinstance(e, CaseExpr):
            # one DFGPU_EXPR_CASE node per WHEN, later branches nested in ELSE (include/dfgpu.h)
            tail = -1 if e.else_expr is None else emit(e.else_expr)
            for w, t in reversed(e.when_then[1:]):
                m = ExprNode()
                m.op, m.column, m.left, m.right = OP_CASE, emit(w), emit(t), tail
                nodes.append(m)
                tail = len(nodes) - 1
            w, t = e.when_then[0]
            n.op, n.column, n.left, n.right = OP_CASE, emit(w), emit(t), tail
        else:
            raise TypeError(f"{type(e).__name__} is not supported on the GPU path")
        nodes.append(n)
        return len(nodes) - 1

    emit(expr)
    return LoweredExpr(nodes, bytes(pool) + b"\0")
