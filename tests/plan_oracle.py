"""TEST INFRASTRUCTURE: runs a physical plan (datafusion_amd.physical_plan node tree, used here purely as a data
structure) with the CPU oracle's operators, so that ONE statement of a reference plan
(tests/tpch_plans.py) is executed twice: by the product on the GPU and by the oracle here.

Leaves hold pyarrow Tables.  The oracle has no string type: dictionary-encoded string columns are carried as their
index columns plus a {column name: dictionary} side table, string literals are bound to indices exactly as the product
binds them (expr.bind_string_literals: `=` / `!=` only; an absent string compares as index -1), and dictionaries are
re-attached to the same-named columns of the result.  The GPU-specific nodes are interpreted by their definition:
GpuFusedAggregateExec = FilterExec -> AggregateExec over the inlined expressions, GpuHashJoinExec = FilterExec on the
probe side -> HashJoinExec, which checks on the CPU that GpuOffloadRule's rewrites preserve the plan's meaning.
"""
from __future__ import annotations

import pyarrow as pa

from datafusion_amd import expr as X
from datafusion_amd import physical_plan as P
from oracle import oracle


class Rel:
    """a table of oracle-typed columns + the dictionaries of its dictionary-encoded columns"""

    def __init__(self, table: pa.Table, dicts: dict):
        self.table, self.dicts = table, {k: v for k, v in dicts.items() if k in table.column_names}


def _encode(table: pa.Table) -> Rel:
    cols, dicts = [], {}
    for name, c in zip(table.column_names, table.columns):
        if pa.types.is_dictionary(c.type):
            arr = c.combine_chunks()
            dicts[name] = arr.dictionary.to_pylist()
            cols.append(arr.indices)
        else:
            cols.append(c)
    return Rel(pa.Table.from_arrays(cols, names=table.column_names), dicts)


def decode(rel: Rel) -> pa.Table:
    cols = []
    for name, c in zip(rel.table.column_names, rel.table.columns):
        if name in rel.dicts:
            c = pa.DictionaryArray.from_arrays(c.combine_chunks(), pa.array(rel.dicts[name], pa.string()))
        cols.append(c)
    return pa.Table.from_arrays(cols, names=rel.table.column_names)


def _expr(e, rel: Rel):
    """product PhysicalExpr -> oracle tuple AST, string literals bound to dictionary indices"""
    if isinstance(e, X.Column):
        return ("col", e.name)
    if isinstance(e, X.Literal):
        return ("lit", e.value, e.type)
    if isinstance(e, X.CastExpr):
        return ("cast", _expr(e.expr, rel), e.cast_type)
    if isinstance(e, X.BinaryExpr):
        if e.op in ("=", "!="):
            for a, b in ((e.left, e.right), (e.right, e.left)):
                if isinstance(a, X.Column) and isinstance(b, X.Literal) and pa.types.is_string(b.type):
                    itype = rel.table.schema.field(a.name).type
                    values = rel.dicts[a.name]
                    if b.value is None:
                        return ("bin", e.op, ("col", a.name), ("lit", None, itype))
                    if b.value in values:
                        return ("bin", e.op, ("col", a.name), ("lit", values.index(b.value), itype))
                    return ("bin", e.op, ("cast", ("col", a.name), pa.int64()), ("lit", -1, pa.int64()))
        return ("bin", e.op, _expr(e.left, rel), _expr(e.right, rel))
    if isinstance(e, X.IsNullExpr):
        return ("is_null", _expr(e.arg, rel))
    if isinstance(e, X.IsNotNullExpr):
        return ("not", ("is_null", _expr(e.arg, rel)))
    if isinstance(e, X.NotExpr):
        return ("not", _expr(e.arg, rel))
    if isinstance(e, X.InListExpr):
        return _expr(e.lowered(), rel)
    if isinstance(e, X.DatePartExpr):
        return ("date_part", e.part, _expr(e.arg, rel))
    if isinstance(e, X.LikeExpr):
        # the checker's own LIKE: pyarrow's match_like over the dictionary, then membership of the index
        import pyarrow.compute as pc
        values = rel.dicts[e.expr.name]
        hit = pc.match_like(pa.array(values, pa.string()), e.pattern, ignore_case=e.case_insensitive).to_pylist()
        itype = rel.table.schema.field(e.expr.name).type
        codes = [i for i, h in enumerate(hit) if h]
        if codes:
            out = None
            for c in codes:
                t = ("bin", "=", ("col", e.expr.name), ("lit", c, itype))
                out = t if out is None else ("bin", "or", out, t)
        else:
            out = ("bin", "=", ("cast", ("col", e.expr.name), pa.int64()), ("lit", -1, pa.int64()))
        return ("not", out) if e.negated else out
    if isinstance(e, X.CaseExpr):
        tail = None if e.else_expr is None else _expr(e.else_expr, rel)
        for w, t in reversed(e.when_then):
            tail = ("case", _expr(w, rel), _expr(t, rel), tail)
        return tail
    raise TypeError(e)


def _dict_of(e, rel: Rel):
    """the dictionary an expression's values index: a dictionary-encoded column, or a CASE whose branches are such
    columns / NULL literals"""
    if isinstance(e, X.Column):
        return rel.dicts.get(e.name)
    if isinstance(e, X.CaseExpr):
        branches = [t for _, t in e.when_then] + ([] if e.else_expr is None else [e.else_expr])
        found = [d for d in (_dict_of(b, rel) for b in branches) if d is not None]
        return found[0] if found else None
    return None


def _renamed_dicts(rel: Rel, pairs):
    """dictionaries follow column references (and CASE over them) through (expr, name) lists"""
    return {n: d for e, n in pairs for d in [_dict_of(e, rel)] if d is not None}


def _substr(v: str, start: int, count):
    """SQL SUBSTRING over code points (functions/src/unicode/substr.rs): 1-based start, positions below 1 eat into the count"""
    first = max(start - 1, 0)
    if count is None:
        return v[first:]
    assert count >= 0, "negative substring length not allowed"
    return v[first:max(start - 1 + count, first)]


def _lower_substr(rel: Rel, exprs):
    """substr over a dictionary-encoded column, the checker's way: the function runs over the dictionary, the rows get the indices of
    the ascending dictionary of the distinct results as a new column, and the expression becomes a reference to it"""
    table, dicts = rel.table, dict(rel.dicts)

    def lower(e):
        nonlocal table
        if isinstance(e, X.SubstrExpr):
            assert isinstance(e.arg, X.Column) and e.arg.name in dicts, "the oracle's substr takes a dictionary-encoded column"
            name = f"__substr_{e.arg.name}_{e.start}_{e.count}"
            if name not in table.column_names:
                values = dicts[e.arg.name]
                mapped = [None if v is None else _substr(v, e.start, e.count) for v in values]
                order = sorted(set(v for v in mapped if v is not None))
                at = {v: i for i, v in enumerate(order)}
                codes = table.column(e.arg.name).to_pylist()
                itype = table.schema.field(e.arg.name).type
                table = table.append_column(name, pa.array([None if c is None or mapped[c] is None else at[mapped[c]] for c in codes], itype))
                dicts[name] = order
            return X.Column(name)
        if isinstance(e, X.LikeExpr) and isinstance(e.expr, X.Column) and e.expr.name in dicts and len(dicts[e.expr.name]) > 64:
            # a large dictionary (p_name: one value per part): the checker's LIKE runs over the dictionary with pyarrow and the rows
            # get a Boolean column (a chain of `=` per matching index would nest thousands deep)
            import numpy as np
            import pyarrow.compute as pc
            name = f"__like_{e.expr.name}_{len(table.column_names)}"
            hit = np.array(pc.match_like(pa.array(dicts[e.expr.name], pa.string()), e.pattern, ignore_case=e.case_insensitive).fill_null(False).to_pylist(), dtype=bool)
            codes = table.column(e.expr.name).combine_chunks()
            vals = hit[np.asarray(codes.fill_null(0).to_numpy(zero_copy_only=False), dtype=np.int64)]
            if e.negated:
                vals = ~vals
            table = table.append_column(name, pa.array(vals, pa.bool_(), mask=np.asarray(codes.is_null().to_numpy(zero_copy_only=False))))
            return X.Column(name)
        if isinstance(e, (X.Column, X.Literal)):
            return e
        if isinstance(e, X.CastExpr):
            return X.CastExpr(lower(e.expr), e.cast_type)
        if isinstance(e, X.BinaryExpr):
            return X.BinaryExpr(lower(e.left), e.op, lower(e.right))
        if isinstance(e, X.IsNullExpr):
            return X.IsNullExpr(lower(e.arg))
        if isinstance(e, X.IsNotNullExpr):
            return X.IsNotNullExpr(lower(e.arg))
        if isinstance(e, X.NotExpr):
            return X.NotExpr(lower(e.arg))
        if isinstance(e, (X.CaseExpr, X.InListExpr, X.DatePartExpr)):
            return e.map_children(lower)
        return e

    out = [lower(e) for e in exprs]
    rel2 = Rel(table, dicts)
    return rel2, out


def _filter(rel: Rel, predicate, projection) -> Rel:
    keep = projection if projection is not None else rel.table.column_names
    rel, (predicate,) = _lower_substr(rel, [predicate])
    return Rel(oracle.filter(rel.table, _expr(predicate, rel), keep), rel.dicts)


def _avg_return_types(rel: Rel, aggs) -> dict:
    """declared AVG return types over the raw input (what the reference's Final node is planned with)"""
    out = {}
    for f, e, n in aggs:
        if f == "avg" and e is not None:
            t = oracle.evaluate(_expr(e, rel), rel.table.slice(0, 0)).typ
            if pa.types.is_decimal128(t):
                out[n] = oracle.avg_result_type(t)
    return out


def _aggregate(rel: Rel, mode, group_by, aggs, return_types=None) -> Rel:
    final = mode in ("Final", "FinalPartitioned")
    if not final:   # (a fused node carries the projection's expressions: substr over dictionary columns is lowered first)
        rel, low = _lower_substr(rel, [e for e, _ in group_by] + [e for _, e, _ in aggs if e is not None])
        group_by = [(low[i], n) for i, (_, n) in enumerate(group_by)]
        it = iter(low[len(group_by):])
        aggs = [(f, None if e is None else next(it), n) for f, e, n in aggs]
    gb = [(None if final else _expr(e, rel), n) for e, n in group_by]
    ag = [(f, None if (e is None or final) else _expr(e, rel), n) for f, e, n in aggs]
    out = oracle.aggregate(rel.table, gb, ag, mode, return_types=return_types)
    if final:   # group columns are the first columns of the partial state, by position
        dicts = {n: rel.dicts[rel.table.column_names[i]] for i, (_, n) in enumerate(group_by) if rel.table.column_names[i] in rel.dicts}
    else:
        dicts = _renamed_dicts(rel, group_by)
    return Rel(out, dicts)


def _join(node, left: Rel, right: Rel) -> Rel:
    shared_build = node.join_type in P.HashJoinExec._BUILD_EMITTING and P.replicated_build(node.left)
    if shared_build:    # CollectLeft with build-side emission on several ranks: physical_plan.HashJoinExec.execute
        right = Rel(pa.concat_tables(_all_gather_tables(right.table)), right.dicts)
    out = oracle.hash_join(left.table, right.table, node.on, node.join_type, node.null_equality, join_filter=_join_filter(node, left, right), null_aware=node.null_aware)
    if shared_build:
        import torch.distributed as dist
        n, world, rank = out.num_rows, dist.get_world_size(), dist.get_rank()
        out = out.slice(n * rank // world, n * (rank + 1) // world - n * rank // world)
    dicts = dict(right.dicts)
    dicts.update(left.dicts)
    rel = Rel(out, dicts)
    if node.projection:
        bc, pc = node.projection
        names = []
        if node.join_type not in ("RightSemi", "RightAnti", "RightMark"):
            names += list(left.table.column_names if bc is None else bc)
        if node.join_type not in ("LeftSemi", "LeftAnti", "LeftMark"):
            names += list(right.table.column_names if pc is None else pc)
        if node.join_type in ("LeftMark", "RightMark"):
            names.append("mark")
        rel = Rel(out.select(names), dicts)
    return rel


def _join_filter(node, left: Rel, right: Rel):
    """JoinFilter: expression over the intermediate columns f0, f1, ... (column index, side); string literals are bound through
    the dictionaries of the columns the intermediate columns come from"""
    if node.filter is None:
        return None
    e, cols = node.filter
    fields, dicts = [], {}
    for k, (idx, side) in enumerate(cols):
        src = left if side == "Left" else right
        name = src.table.column_names[idx]
        fields.append(pa.field(f"f{k}", src.table.schema.field(idx).type))
        if name in src.dicts:
            dicts[f"f{k}"] = src.dicts[name]
    inter = Rel(pa.Table.from_arrays([pa.array([], f.type) for f in fields], names=[f.name for f in fields]), dicts)
    return (_expr(e, inter), cols)


# --------------------------------------------------------------------------------------------- several ranks (gloo)
# With torch.distributed initialised (world size N > 1) the interpreter runs the plan the way N GPUs do (one process per GPU):
# a leaf is this rank's contiguous row range of its table (what N scans produce), RepartitionExec(Hash) routes rows with the
# oracle's create_hashes (seed 0) % N and exchanges them, CoalescePartitionsExec / SortPreservingMergeExec gather every rank's
# partition to every rank (the result is replicated, as in physical_plan.py).  The product's nodes do the same with the device
# partition kernel + RCCL (exchange.hash_exchange; tests/test_gpu_sort_partition.py pins device routing == oracle routing).
def _world():
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _all_gather_tables(t: pa.Table):
    import torch.distributed as dist
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, t)
    return parts


def _hash_exchange(rel: Rel, keys) -> Rel:
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    parts, _ = oracle.hash_partition(rel.table, keys, world)
    received = [None] * world
    for dst in range(world):                      # one gather per destination: rank dst collects everybody's slice for it
        got = [None] * world if rank == dst else None
        dist.gather_object(parts[dst], got, dst=dst)
        if rank == dst:
            received = got
    return Rel(pa.concat_tables(received), rel.dicts)


def run(plan) -> Rel:
    if _world() > 1:
        import torch.distributed as dist
        world, rank = dist.get_world_size(), dist.get_rank()
        if isinstance(plan, P.MemoryExec):
            t = plan.table if plan.projection is None else plan.table.select(plan.projection)
            lo, hi = t.num_rows * rank // world, t.num_rows * (rank + 1) // world
            return _encode(t.slice(lo, hi - lo))
        if isinstance(plan, P.RepartitionExec):
            return _hash_exchange(run(plan.input), plan.keys)
        if isinstance(plan, P.CoalescePartitionsExec):
            rel = run(plan.input)
            return Rel(pa.concat_tables(_all_gather_tables(rel.table)), rel.dicts)
        if isinstance(plan, P.SortPreservingMergeExec):
            rel = run(plan.input)
            merged = pa.concat_tables(_all_gather_tables(rel.table))
            return Rel(oracle.sort(merged, plan.expr, plan.fetch), rel.dicts)
    if isinstance(plan, P.MemoryExec):
        t = plan.table if plan.projection is None else plan.table.select(plan.projection)
        return _encode(t)
    if isinstance(plan, P.ParquetExec):     # the CPU leg reads the same file with pyarrow's reader
        import pyarrow.parquet as pq
        t = pq.read_table(plan.path, columns=plan.projection)
        strings = [n for n in t.column_names if pa.types.is_string(t.schema.field(n).type)]
        for n in strings:   # as the device scan delivers them: dictionary-encoded, ascending dictionary
            values = sorted(set(v for v in t.column(n).to_pylist() if v is not None))
            idx = pa.array([None if v is None else values.index(v) for v in t.column(n).to_pylist()], pa.int32())
            t = t.set_column(t.column_names.index(n), n, pa.DictionaryArray.from_arrays(idx, pa.array(values, pa.string())))
        return _encode(t)
    if isinstance(plan, (P.CoalesceBatchesExec, P.RepartitionExec, P.CoalescePartitionsExec, P.SortPreservingMergeExec)):
        return run(plan.input)          # one partition: bookkeeping only
    if isinstance(plan, P.FilterExec):
        return _filter(run(plan.input), plan.predicate, plan.projection)
    if isinstance(plan, P.ProjectionExec):
        rel = run(plan.input)
        rel, lowered = _lower_substr(rel, [e for e, _ in plan.exprs])
        pairs = [(e, n) for e, (_, n) in zip(lowered, plan.exprs)]
        return Rel(oracle.project(rel.table, [(_expr(e, rel), n) for e, n in pairs]), _renamed_dicts(rel, pairs))
    if isinstance(plan, P.GpuHashJoinExec):
        probe = _filter(run(plan.right), plan.probe_predicate, None)
        return _join(plan, run(plan.left), probe)
    if isinstance(plan, P.HashJoinExec):
        return _join(plan, run(plan.left), run(plan.right))
    if isinstance(plan, P.GpuFusedAggregateExec):
        rel = run(plan.input)
        if plan.predicate is not None:
            rel = _filter(rel, plan.predicate, None)
        if plan.mode == "Partial":
            plan.return_types = _avg_return_types(rel, plan.aggr_expr)
        return _aggregate(rel, plan.mode, plan.group_by, plan.aggr_expr)
    if isinstance(plan, P.AggregateExec):
        rel = run(plan.input)
        rt = None
        if plan.mode == "Partial":
            plan.return_types = _avg_return_types(rel, plan.aggr_expr)
        elif plan.mode in ("Final", "FinalPartitioned"):
            below = P._partial_below(plan.input)
            rt = below.return_types if below is not None else None
        return _aggregate(rel, plan.mode, plan.group_by, plan.aggr_expr, rt)
    if isinstance(plan, P.SortExec):
        rel = run(plan.input)
        return Rel(oracle.sort(rel.table, plan.expr, plan.fetch), rel.dicts)
    if isinstance(plan, P.ScalarSubqueryExec):   # scalar_subquery.rs:85: the subqueries first, each exactly once
        for sub, index in plan.subqueries:
            t = decode(run(sub))
            assert t.num_rows <= 1, "Scalar subquery returned more than one row"
            plan.results[index] = None if t.num_rows == 0 else t.column(0)[0].as_py()
        return run(plan.input)
    raise TypeError(f"no oracle interpretation of {plan.name()}")


def collect(plan) -> pa.Table:
    return decode(run(plan))
