"""Operator-level calls into libdfgpu.so — thin, typed wrappers of the C ABI entry points.
The ExecutionPlan-shaped layer (physical_plan.py) is built on these.
"""
from __future__ import annotations

import ctypes as C

import pyarrow as pa

from . import _lib
from ._lib import AggSpec, Expr, Field, JoinFilter, JoinInfo, JoinOptions, KernelStat, check
from .expr import LoweredExpr, PhysicalExpr, lower
from .table import DeviceTable, field_of

JOIN_TYPES = {"Inner": 0, "Left": 1, "Right": 2, "Full": 3, "LeftSemi": 4, "RightSemi": 5, "LeftAnti": 6,
              "RightAnti": 7, "LeftMark": 8, "RightMark": 9}
NULL_EQUALITY = {"NullEqualsNothing": 0, "NullEqualsNull": 1}
AGG_MODES = {"Partial": 0, "Final": 1, "FinalPartitioned": 2, "Single": 3, "SinglePartitioned": 4, "PartialReduce": 5}
AGG_FUNCS = {"sum": 0, "min": 1, "max": 2, "count": 3, "avg": 4}
GPU_MIN_KEY_DENSITY = 1.0 / 64.0      # DFGPU_DEFAULT_MIN_KEY_DENSITY (include/dfgpu.h); the reference's CPU default is 0.15
TABLE_MODES = {"auto": 0, "hash_map": 1, "array_map": 2, "rank_map": 3}
PROBE_MODES = {"auto": 0, "two_pass": 1, "single_pass_ordered": 2, "single_pass_unordered": 3, "order_not_needed": 4}


def _ints(values):
    return (C.c_int * max(1, len(values)))(*values)


def filter(table: DeviceTable, predicate: PhysicalExpr, projection=None) -> DeviceTable:
    """FilterExec: predicate + optional embedded projection (filter.rs:85)"""
    lib = _lib.init()
    names = table.column_names
    le = lower(predicate, names, table)
    out = C.c_void_p()
    if projection is None:
        check(lib.dfgpu_filter(table.handle, C.byref(le.c), None, 0, C.byref(out)))
    else:
        idx = [table.index_of(c) for c in projection]
        check(lib.dfgpu_filter(table.handle, C.byref(le.c), _ints(idx), len(idx), C.byref(out)))
    return DeviceTable(out)


class FilterPlan:
    """A PLANNED FilterExec (predicate + embedded projection, filter.rs:85): the predicate is lowered to the C ABI's form and the projected
    columns are resolved once; `execute(table)` runs dfgpu_filter on any table of the planning table's schema (whose dictionaries bound
    the predicate's string literals)."""

    def __init__(self, table: DeviceTable, predicate: PhysicalExpr, projection=None):
        self._pred = lower(predicate, table.column_names, table)
        self._idx = None if projection is None else [table.index_of(c) for c in projection]
        self._arr = None if self._idx is None else _ints(self._idx)

    def execute(self, table: DeviceTable) -> DeviceTable:
        out = C.c_void_p()
        check(_lib.load().dfgpu_filter(table.handle, C.byref(self._pred.c), self._arr, 0 if self._idx is None else len(self._idx), C.byref(out)))
        return DeviceTable(out)


def project(table: DeviceTable, exprs) -> DeviceTable:
    """ProjectionExec: exprs = [(PhysicalExpr, name)] (projection.rs:439)"""
    lib = _lib.init()
    names = table.column_names
    lowered = [lower(e, names, table) for e, _ in exprs]
    arr = (Expr * len(exprs))(*[l.c for l in lowered])
    cnames = (C.c_char_p * len(exprs))(*[n.encode() for _, n in exprs])
    out = C.c_void_p()
    check(lib.dfgpu_project(table.handle, arr, cnames, len(exprs), C.byref(out)))
    return DeviceTable(out)


class JoinHashTable:
    """JoinLeftData (hash_join/exec.rs:195-240): the built side of a hash join"""

    def __init__(self, build: DeviceTable, on_left, null_equality="NullEqualsNothing", table_mode=0,
                 small_build_threshold=1024, min_key_density=GPU_MIN_KEY_DENSITY, force_hash_collisions=False, probe_mode=0,
                 null_aware=False):
        lib = _lib.init()
        self.build = build  # keep alive
        self.key_idx = [build.index_of(k) for k in on_left]
        opts = JoinOptions(small_build_threshold, min_key_density, table_mode, int(force_hash_collisions), probe_mode, int(null_aware))
        self._h = C.c_void_p()
        check(lib.dfgpu_join_build(build.handle, _ints(self.key_idx), len(self.key_idx), NULL_EQUALITY[null_equality],
                                   C.byref(opts), C.byref(self._h)))

    def probe(self, probe: DeviceTable, on_right, join_type="Inner", build_cols=None, probe_cols=None,
              predicate: PhysicalExpr | None = None, join_filter=None) -> DeviceTable:
        """`predicate` = a FilterExec fused below the probe side: with the single-pass probe its row mask is applied
        inside the probe kernel and the filtered probe table is never materialised (dfgpu_join_probe_filtered).
        `join_filter` = JoinFilter (joins/join_filter.rs): (expression over intermediate columns f0, f1, ...,
        [(column index, "Left" | "Right"), ...]); key-equal pairs whose filter value is not TRUE are not matches."""
        lib = _lib.load()
        pk = [probe.index_of(k) for k in on_right]
        bc = list(range(self.build.num_columns)) if build_cols is None else [self.build.index_of(c) for c in build_cols]
        pc = list(range(probe.num_columns)) if probe_cols is None else [probe.index_of(c) for c in probe_cols]
        if join_type in ("LeftSemi", "LeftAnti", "LeftMark"):
            pc = []
        if join_type in ("RightSemi", "RightAnti", "RightMark"):
            bc = []
        out = C.c_void_p()
        if join_filter is not None:
            if predicate is not None:
                raise _lib.DfgpuError("a fused probe-side predicate and a join filter cannot be combined: filter the probe side first")
            fexpr, fcols = join_filter
            from .expr import IntermediateSchema, _has_string_literal
            # string literals compared with intermediate columns: bound through the dictionaries of the columns behind them
            view = IntermediateSchema(self.build, probe, fcols) if _has_string_literal(fexpr) else None
            le = lower(fexpr, [f"f{i}" for i in range(len(fcols))], view)
            idx = (C.c_int32 * max(1, len(fcols)))(*[int(i) for i, _ in fcols])
            side = (C.c_int32 * max(1, len(fcols)))(*[0 if sd == "Left" else 1 for _, sd in fcols])
            jf = JoinFilter(le.c, idx, side, len(fcols))
            check(lib.dfgpu_join_probe_with_filter(self._h, probe.handle, _ints(pk), JOIN_TYPES[join_type], C.byref(jf), _ints(bc), len(bc), _ints(pc),
                                                   len(pc), C.byref(out)))
            return DeviceTable(out)
        if predicate is None:
            check(lib.dfgpu_join_probe(self._h, probe.handle, _ints(pk), JOIN_TYPES[join_type], _ints(bc), len(bc), _ints(pc),
                                       len(pc), C.byref(out)))
        else:
            le = lower(predicate, probe.column_names, probe)
            check(lib.dfgpu_join_probe_filtered(self._h, probe.handle, C.byref(le.c), _ints(pk), JOIN_TYPES[join_type], _ints(bc), len(bc),
                                                _ints(pc), len(pc), C.byref(out)))
        return DeviceTable(out)

    def probe_bounded(self, probe: DeviceTable, on_right, join_type="Inner", build_cols=None, probe_cols=None, max_output_rows: int = 8192):
        """the probe as a stream of pieces of at most `max_output_rows` rows each (dfgpu_join_probe_bounded: HashJoinStream's limit / offset
        resumption, hash_join/stream.rs:396-437): yields one DeviceTable per piece, in probe order of the pieces"""
        lib = _lib.load()
        pk = [probe.index_of(k) for k in on_right]
        bc = list(range(self.build.num_columns)) if build_cols is None else [self.build.index_of(c) for c in build_cols]
        pc = list(range(probe.num_columns)) if probe_cols is None else [probe.index_of(c) for c in probe_cols]
        if join_type in ("LeftSemi", "LeftAnti", "LeftMark"):
            pc = []
        if join_type in ("RightSemi", "RightAnti", "RightMark"):
            bc = []
        offset = 0
        while True:
            out, nxt = C.c_void_p(), C.c_int64()
            check(lib.dfgpu_join_probe_bounded(self._h, probe.handle, _ints(pk), JOIN_TYPES[join_type], _ints(bc), len(bc), _ints(pc), len(pc),
                                               C.c_int64(offset), C.c_int64(int(max_output_rows)), C.byref(out), C.byref(nxt)))
            yield DeviceTable(out)
            offset = nxt.value
            if offset >= probe.num_rows:
                break

    def contains(self, probe: DeviceTable, on_right) -> DeviceTable:
        """HashTableLookupExpr (hash_join/partitioned_hash_eval.rs:278), the Map strategy of the join's dynamic filter: a one-column
        Boolean table `contains` of `probe`'s rows — is the row's key in the build side?  (dfgpu_join_contains)"""
        pk = [probe.index_of(k) for k in on_right]
        out = C.c_void_p()
        check(_lib.load().dfgpu_join_contains(self._h, probe.handle, _ints(pk), C.byref(out)))
        return DeviceTable(out)

    def emit_unmatched(self, join_type, build_cols=None, probe_schema: pa.Schema | None = None) -> DeviceTable:
        """process_unmatched_build_batch (stream.rs:1002-): build rows by visited state"""
        lib = _lib.load()
        bc = list(range(self.build.num_columns)) if build_cols is None else [self.build.index_of(c) for c in build_cols]
        fields = [field_of(f.type) for f in probe_schema] if probe_schema is not None else []
        names = [f.name.encode() for f in probe_schema] if probe_schema is not None else []
        farr = (Field * max(1, len(fields)))(*fields)
        narr = (C.c_char_p * max(1, len(names)))(*names)
        out = C.c_void_p()
        check(lib.dfgpu_join_emit_unmatched(self._h, JOIN_TYPES[join_type], _ints(bc), len(bc), farr, narr, len(fields), C.byref(out)))
        return DeviceTable(out)

    def info(self) -> JoinInfo:
        i = JoinInfo()
        check(_lib.load().dfgpu_join_get_info(self._h, C.byref(i)))
        return i

    def free(self):
        if self._h:
            _lib.load().dfgpu_join_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def hash_join(left: DeviceTable, right: DeviceTable, on, join_type="Inner", null_equality="NullEqualsNothing",
              build_cols=None, probe_cols=None, join_filter=None, **build_opts) -> DeviceTable:
    """HashJoinExec over whole tables: left = build side, right = probe side"""
    ht = JoinHashTable(left, [l for l, _ in on], null_equality, **build_opts)
    out = ht.probe(right, [r for _, r in on], join_type, build_cols, probe_cols, join_filter=join_filter)
    if join_type in ("Left", "Full", "LeftSemi", "LeftAnti", "LeftMark"):
        tail = ht.emit_unmatched(join_type, build_cols, tail_probe_schema(right, join_type, probe_cols))
        if join_type in ("Left", "Full"):
            # concat needs equal nullability handling: go through the generic concat
            out = concat_tables([out, tail])
        else:
            out = tail
    ht.free()
    return out


def tail_probe_schema(right: DeviceTable, join_type, probe_cols=None):
    """the NULL-filled probe columns of the unmatched build rows of Left / Full joins (dfgpu_join_emit_unmatched's probe_fields)"""
    if join_type not in ("Left", "Full"):
        return None
    rs = right.schema
    pnames = rs.names if probe_cols is None else [right.column_names[right.index_of(c)] for c in probe_cols]
    return pa.schema([rs.field(rs.get_field_index(n)) if rs.names.count(n) == 1 else rs.field(right.index_of(n)) for n in pnames])


def concat_tables(parts) -> DeviceTable:
    """concat_batches; falls back to a host round trip when a part carries NULLs (not a hot path)"""
    try:
        return DeviceTable.concat(parts)
    except _lib.DfgpuError:
        tables = [p.to_arrow() for p in parts]
        return DeviceTable.from_arrow(pa.concat_tables([t.cast(tables[0].schema) for t in tables]))


def partition(table: DeviceTable, keys, nparts: int):
    """RepartitionExec Partitioning::Hash(keys, nparts) -> list of DeviceTables"""
    lib = _lib.init()
    idx = [table.index_of(k) for k in keys]
    outs = (C.c_void_p * nparts)()
    check(lib.dfgpu_partition(table.handle, _ints(idx), len(idx), nparts, outs))
    return [DeviceTable(C.c_void_p(h)) for h in outs]


def sort(table: DeviceTable, keys, fetch=None) -> DeviceTable:
    """SortExec: keys = [(column, descending, nulls_first)]; fetch = TopK limit"""
    lib = _lib.init()
    idx = [table.index_of(k) for k, _, _ in keys]
    desc = (C.c_uint8 * len(keys))(*[int(d) for _, d, _ in keys])
    nf = (C.c_uint8 * len(keys))(*[int(f) for _, _, f in keys])
    out = C.c_void_p()
    check(lib.dfgpu_sort(table.handle, _ints(idx), desc, nf, len(keys), C.c_int64(-1 if fetch is None else fetch), C.byref(out)))
    return DeviceTable(out)


class _LoweredAggregate:
    """the C-ABI form of an AggregateExec's expressions (group keys, aggregate arguments, grouping sets): what
    dfgpu_agg_create reads.  Built once per planned node; every execution of the node creates its state from it."""

    def __init__(self, mode, input_names, group_by, aggs, dictionaries=None, return_types=None, grouping_sets=None):
        self._keep = []
        final = mode in ("Final", "FinalPartitioned", "PartialReduce")
        if final:
            # Final modes (and PartialReduce) read the partial-state schema positionally (group columns first); the
            # original argument expressions do not exist in that schema and are not evaluated
            from .expr import Column
            group_by = [(Column(n, i), n) for i, (_, n) in enumerate(group_by)]
            aggs = [(f, None if f == "count" and e is None else Column("state", 0), n) for f, e, n in aggs]
            dictionaries = None
        g_low = [lower(e, input_names, dictionaries) for e, _ in group_by]
        self._keep += g_low
        self.n_group = len(group_by)
        self.garr = (Expr * max(1, len(g_low)))(*[l.c for l in g_low])
        self.gnames = (C.c_char_p * max(1, len(group_by)))(*[n.encode() for _, n in group_by])
        specs = []
        for func, e, name in aggs:
            s = AggSpec()
            s.func = AGG_FUNCS[func]
            s.has_arg = 0 if e is None else 1
            if e is not None:
                l = lower(e, input_names, dictionaries)
                self._keep.append(l)
                s.arg = l.c
            s.name = name.encode()
            if return_types and name in return_types:
                s.return_field = field_of(return_types[name])
            specs.append(s)
        self.n_aggs = len(specs)
        self.sarr = (AggSpec * max(1, len(specs)))(*specs)
        self.grouping = None
        if grouping_sets is not None:
            # PhysicalGroupBy with several groups: grouping_sets = ([typed NULL Literal per key], [[column g is NULL in set s] ...])
            null_exprs, groups = grouping_sets
            n_low = [lower(e, input_names, dictionaries) for e in null_exprs]
            self._keep += n_low
            narr = (Expr * max(1, len(n_low)))(*[l.c for l in n_low])
            flat = (C.c_uint8 * max(1, len(groups) * len(group_by)))(*[int(bool(x)) for g in groups for x in g])
            self.grouping = (narr, flat, len(groups))


class AggregatePlan:
    """A PLANNED AggregateExec (optionally with the FilterExec below it fused in as `predicate`): the expressions are lowered to
    the C ABI's form once, at planning time, and `execute(table)` runs the node — dfgpu_agg_create -> update[_filtered] -> emit —
    any number of times, the way one ExecutionPlan is `execute()`d repeatedly (physical-plan/src/execution_plan.rs:696).  The
    table given here fixes the input schema (and binds string literals to its dictionaries)."""

    def __init__(self, table: DeviceTable, group_by, aggs, mode="Single", predicate: PhysicalExpr | None = None, return_types: dict | None = None):
        self.mode, self.column_names = mode, list(table.column_names)
        self._lowered = _LoweredAggregate(mode, self.column_names, group_by, aggs, table, return_types)
        self._predicate = None if predicate is None else lower(predicate, self.column_names, table)

    def execute(self, table: DeviceTable) -> DeviceTable:
        a = GroupedAggregate(self.mode, None, None, None, _lowered=self._lowered)
        try:
            a.update(table, self._predicate)
            return a.emit()
        finally:
            a.free()


class GroupedAggregate:
    """AggregateExec state: group_by = [(expr, name)], aggs = [(func, expr|None, name)]"""

    def __init__(self, mode, input_names, group_by, aggs, dictionaries: DeviceTable | None = None, return_types: dict | None = None, grouping_sets=None,
                 _lowered: "_LoweredAggregate | None" = None):
        """dictionaries: a table of the input's schema whose dictionary-encoded string columns bind the string
        literals of the argument expressions (`CASE WHEN o_orderpriority = '1-URGENT' ...`).
        return_types {name: arrow type}: the aggregates' declared return types (AggregateFunctionExpr::return_field) —
        required by Final modes for AVG over a Decimal128 state (its Decimal128(38, s) sum does not tell the argument's
        precision); aggregate_return_types() computes them from the raw input."""
        lib = _lib.init()
        low = _lowered if _lowered is not None else _LoweredAggregate(mode, input_names, group_by, aggs, dictionaries, return_types, grouping_sets)
        self._keep = low           # the C structs the library copied from stay alive with the plan that owns them
        self._h = C.c_void_p()
        if low.grouping is not None:
            narr, flat, n_sets = low.grouping
            check(lib.dfgpu_agg_create_grouping_sets(AGG_MODES[mode], low.garr, narr, low.gnames, low.n_group, flat, n_sets, low.sarr, low.n_aggs, C.byref(self._h)))
            return
        check(lib.dfgpu_agg_create(AGG_MODES[mode], low.garr, low.gnames, low.n_group, low.sarr, low.n_aggs, C.byref(self._h)))

    def update(self, table: DeviceTable, predicate: PhysicalExpr | None = None):
        """aggregate_batch_inner over a whole table; `predicate` = a FilterExec fused in front of this node
        (rows whose predicate is false/NULL are skipped) — evaluated in the same pass as the arguments"""
        if predicate is None:
            check(_lib.load().dfgpu_agg_update(self._h, table.handle))
        else:
            le = predicate if isinstance(predicate, LoweredExpr) else lower(predicate, table.column_names, table)
            check(_lib.load().dfgpu_agg_update_filtered(self._h, table.handle, C.byref(le.c)))

    @property
    def fused_updates(self) -> int:
        n = C.c_int64()
        check(_lib.load().dfgpu_agg_fused_updates(self._h, C.byref(n)))
        return n.value

    def emit(self) -> DeviceTable:
        out = C.c_void_p()
        check(_lib.load().dfgpu_agg_emit(self._h, C.byref(out)))
        return DeviceTable(out)

    def free(self):
        if self._h:
            _lib.load().dfgpu_agg_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def expr_type(table: DeviceTable, expr: PhysicalExpr) -> pa.DataType:
    """PhysicalExpr::data_type over a table's schema (dfgpu_expr_type)"""
    from .table import arrow_type_of
    le = lower(expr, table.column_names, table)
    f = _lib.Field()
    check(_lib.load().dfgpu_expr_type(C.byref(le.c), table.handle, C.byref(f)))
    return arrow_type_of(f)


def aggregate_return_types(table: DeviceTable, aggs) -> dict:
    """declared return types of the aggregates that Final modes cannot derive from the partial state: AVG over a
    Decimal128 argument, Decimal128(min(38, p + 4), min(38, s + 4)) (functions-aggregate/src/average.rs:219-252)"""
    out = {}
    for func, e, name in aggs:
        if func == "avg" and e is not None:
            t = expr_type(table, e)
            if pa.types.is_decimal128(t):
                out[name] = pa.decimal128(min(38, t.precision + 4), min(38, t.scale + 4))
    return out


def aggregate_grouping_sets(table: DeviceTable, group_by, null_exprs, groups, aggs, mode="Single", predicate: PhysicalExpr | None = None) -> DeviceTable:
    """AggregateExec over a PhysicalGroupBy with grouping sets (GROUPING SETS / CUBE / ROLLUP, aggregates/mod.rs:400-520): the key
    columns, `__grouping_id`, the aggregates — one row per (grouping set, group)"""
    a = GroupedAggregate(mode, table.column_names, group_by, aggs, dictionaries=table, grouping_sets=(null_exprs, groups))
    a.update(table, predicate)
    out = a.emit()
    a.free()
    return out


def aggregate(table: DeviceTable, group_by, aggs, mode="Single", predicate: PhysicalExpr | None = None, info: dict | None = None,
              return_types: dict | None = None) -> DeviceTable:
    a = GroupedAggregate(mode, table.column_names, group_by, aggs, table, return_types)
    a.update(table, predicate)
    if info is not None:
        info["fused_updates"] = a.fused_updates
    out = a.emit()
    a.free()
    return out


def column_minmax(table: DeviceTable, column):
    """(min, max, non-null count, strictly ascending?) of an integer column; min/max are None when there is no value"""
    lo, hi, n, asc = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int()
    check(_lib.init().dfgpu_column_minmax(table.handle, table.index_of(column), C.byref(lo), C.byref(hi), C.byref(n), C.byref(asc)))
    if n.value == 0:
        return None, None, 0, False
    return lo.value, hi.value, n.value, bool(asc.value)


def column_inlist(table: DeviceTable, column, max_size: int = 128 * 1024, max_distinct_values: int = 150):
    """the distinct non-NULL values (ascending) of a small integer column for the join's `IN (...)` dynamic filter, or None when the
    build side is beyond the reference's limits (optimizer.hash_join_inlist_pushdown_max_size / _max_distinct_values): the Map
    strategy, bounds only (dfgpu_column_inlist)"""
    n = C.c_int64()
    vals = (C.c_int64 * max(1, max_distinct_values))()
    check(_lib.init().dfgpu_column_inlist(table.handle, table.index_of(column), C.c_int64(max_size), C.c_int64(max_distinct_values), vals,
                                          C.c_int64(max_distinct_values), C.byref(n)))
    return None if n.value < 0 else list(vals[:n.value])


def metrics_reset():
    """zero the calling thread's operator metrics (dfgpu_metrics_reset)"""
    check(_lib.load().dfgpu_metrics_reset())


def metrics() -> dict:
    """the calling thread's operator metrics since its last reset (dfgpu_metrics: what a GPU node's ExecutionPlan::metrics() reports)"""
    m = _lib.Metrics()
    check(_lib.load().dfgpu_metrics_get(C.byref(m)))
    return {n: getattr(m, n) for n, _ in _lib.Metrics._fields_}


def jit_stats():
    """(distinct nodes compiled with hiprtc, total compile ms) of this process"""
    n, ms = C.c_int64(), C.c_double()
    check(_lib.init().dfgpu_jit_stats(C.byref(n), C.byref(ms)))
    return n.value, ms.value


def jit_cache_stats() -> dict:
    """the on-disk code-object cache and the per-device module loads of this process (dfgpu_jit_cache_stats)"""
    h, w, m = C.c_int64(), C.c_int64(), C.c_int64()
    check(_lib.init().dfgpu_jit_cache_stats(C.byref(h), C.byref(w), C.byref(m)))
    return dict(disk_hits=h.value, disk_writes=w.value, modules_loaded=m.value)


def set_fusion(on: bool):
    """expression fusion (rowprog) on/off, process-wide; off = column-at-a-time evaluation everywhere"""
    check(_lib.init().dfgpu_set_fusion(int(on)))


# ------------------------------------------------------------------ synthetic workload
def tpch_orders(sf: float, begin=0, end=-1) -> DeviceTable:
    out = C.c_void_p()
    check(_lib.init().dfgpu_tpch_orders(C.c_double(sf), C.c_int64(begin), C.c_int64(end), C.byref(out)))
    return DeviceTable(out)


def tpch_lineitem(sf: float, begin=0, end=-1, float_money=False) -> DeviceTable:
    out = C.c_void_p()
    check(_lib.init().dfgpu_tpch_lineitem(C.c_double(sf), C.c_int64(begin), C.c_int64(end), int(float_money), C.byref(out)))
    return DeviceTable(out)


def tpch_customer(sf: float, begin=0, end=-1) -> DeviceTable:
    out = C.c_void_p()
    check(_lib.init().dfgpu_tpch_customer(C.c_double(sf), C.c_int64(begin), C.c_int64(end), C.byref(out)))
    return DeviceTable(out)


# ----------------------------------------------------------------------------- streaming boundary, admission control
class JoinBuilder:
    """the build side as a stream of batches (dfgpu_join_builder_*: CollectBuildSide over collect_left_input): push every batch,
    finish() -> JoinHashTable.  A push fails with "Resources exhausted" when the device pool cannot admit the build."""

    def __init__(self, on_left_idx, null_equality="NullEqualsNothing", small_build_threshold=1024, min_key_density=1.0 / 64.0, table_mode=0,
                 force_hash_collisions=False, probe_mode=0, null_aware=False):
        lib = _lib.init()
        self.key_idx = list(on_left_idx)
        opts = JoinOptions(small_build_threshold, min_key_density, table_mode, int(force_hash_collisions), probe_mode, int(null_aware))
        self._h = C.c_void_p()
        check(lib.dfgpu_join_builder_create(_ints(self.key_idx), len(self.key_idx), NULL_EQUALITY[null_equality], C.byref(opts), C.byref(self._h)))
        self._first = None

    def push(self, batch: DeviceTable):
        check(_lib.load().dfgpu_join_builder_push(self._h, batch.handle))
        if self._first is None:
            self._first = batch.select(list(range(batch.num_columns)))   # schema carrier for probe()'s column lookups

    def finish(self) -> "JoinHashTable":
        h = C.c_void_p()
        check(_lib.load().dfgpu_join_builder_finish(self._h, C.byref(h)))
        self._h = None
        ht = JoinHashTable.__new__(JoinHashTable)
        ht.build, ht.key_idx, ht._h = self._first, self.key_idx, h
        return ht

    def free(self):
        if self._h:
            _lib.load().dfgpu_join_builder_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def mem_set_limit(nbytes: int):
    check(_lib.init().dfgpu_mem_set_limit(C.c_int64(nbytes)))


def mem_limit():
    """(limit, bytes reserved by operators in flight) of the current device's pool"""
    lim, res = C.c_int64(), C.c_int64()
    check(_lib.init().dfgpu_mem_limit(C.byref(lim), C.byref(res)))
    return lim.value, res.value


class Reservation:
    """MemoryReservation: dfgpu_mem_try_reserve / dfgpu_mem_release"""

    def __init__(self, nbytes: int):
        self._h = C.c_void_p()
        check(_lib.init().dfgpu_mem_try_reserve(C.c_int64(nbytes), C.byref(self._h)))
        self.nbytes = nbytes

    def release(self):
        if self._h:
            _lib.load().dfgpu_mem_release(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.release()

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


def join_estimate_bytes(build_rows, build_row_bytes, probe_rows, output_rows, output_row_bytes) -> int:
    out = C.c_int64()
    check(_lib.load().dfgpu_join_estimate_bytes(C.c_int64(build_rows), C.c_int64(build_row_bytes), C.c_int64(probe_rows), C.c_int64(output_rows),
                                                C.c_int64(output_row_bytes), C.byref(out)))
    return out.value


# ----------------------------------------------------------------------------- metrics
def set_options(**kv):
    """dfgpu_set_option for every keyword (dots as double underscores: jit__min_rows=0 sets "jit.min_rows"); a value of None restores the
    option's default.  What tests use to force a path on a small input and what an embedding engine would map its ConfigOptions to."""
    lib = _lib.init()
    for k, v in kv.items():
        check(lib.dfgpu_set_option(k.replace("__", ".").encode(), None if v is None else str(v).encode()))


def reset_options():
    """every option back to its device-derived default"""
    check(_lib.load().dfgpu_set_option(None, None))


def sync():
    check(_lib.load().dfgpu_sync())


def profile_enable(on=True):
    check(_lib.init().dfgpu_profile_enable(int(on)))


def profile_reset():
    check(_lib.load().dfgpu_profile_reset())


def profile_stats():
    lib = _lib.load()
    n = C.c_int()
    check(lib.dfgpu_profile_count(C.byref(n)))
    out = {}
    for i in range(n.value):
        s = KernelStat()
        check(lib.dfgpu_profile_get(i, C.byref(s)))
        out[s.name.decode()] = {"calls": s.calls, "total_ms": s.total_ms, "bytes": s.algorithmic_bytes}
    return out


def profile_launches(name: str, capacity: int = 4096):
    """[(ms, algorithmic bytes)] of every launch recorded under `name` since the last reset, in launch order (dfgpu_profile_launches)"""
    ms = (C.c_double * capacity)()
    nb = (C.c_int64 * capacity)()
    n = C.c_int64()
    check(_lib.load().dfgpu_profile_launches(name.encode(), C.c_int64(capacity), ms, nb, C.byref(n)))
    k = min(n.value, capacity)
    return [(ms[i], nb[i]) for i in range(k)]


def mem_stats():
    a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
    check(_lib.load().dfgpu_mem_stats(C.byref(a), C.byref(b), C.byref(c)))
    return {"in_use": a.value, "cached": b.value, "peak": c.value}
