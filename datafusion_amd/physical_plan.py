"""Host-side mirror of the reference's operator / plugin surface for the hot path.

The reference plugs new operators in through two traits: `ExecutionPlan`
(physical-plan/src/execution_plan.rs:102 — `name` :110, `children` :268, `with_new_children` :435,
`execute(partition, ctx)` :696-700) and `PhysicalOptimizerRule`
(session/src/physical_optimizer.rs:52-84 — `optimize`, `name`, `schema_check`).  Rust cannot be compiled in
this image (INTEGRATION.md shows the shim a maintainer adds), so this module is the same surface in Python
over the same C ABI: plan nodes carry the reference's names and constructor arguments, `GpuOffloadRule.optimize`
rewrites a reference-shaped physical plan bottom-up exactly as the Rust rule would, and `execute()` runs the
node on device tables.  Parity tests build the plans pinned in the reference's own plan files
(sqllogictest/test_files/tpch/plans/q1.slt.part:50-58, q3.slt.part:61-76) node for node.

What the rule substitutes (everything else is left as it is — on the CPU in the Rust shim, here the plain
per-operator GPU node):
  AggregateExec(ProjectionExec?(FilterExec?(x)))    -> GpuFusedAggregateExec   one pass, dfgpu_agg_update_filtered
  HashJoinExec(build, FilterExec(probe))             -> GpuHashJoinExec         predicate applied inside the probe kernel
  CoalesceBatchesExec / CoalescePartitionsExec       -> removed                 (whole partitions per launch)
  RepartitionExec(Hash) on one GPU                   -> removed                 (one partition per GPU)
  SortPreservingMergeExec on one GPU                 -> removed                 (one sorted partition is the merge)
"""
from __future__ import annotations

from . import _lib, ops
from .expr import BinaryExpr, CaseExpr, CastExpr, Column, DatePartExpr, InListExpr, IsNotNullExpr, IsNullExpr, LikeExpr, Literal, NotExpr, PhysicalExpr, SubstrExpr
from .table import DeviceTable


# --------------------------------------------------------------------------- expression helpers
def substitute(e: PhysicalExpr, mapping: dict) -> PhysicalExpr:
    """replace Column references by the expressions a ProjectionExec computed them from (projection.rs:713-746
    `update_expr`: how the reference pushes expressions through projections)"""
    if isinstance(e, Column):
        return mapping.get(e.name, e)
    if isinstance(e, Literal):
        return e
    if isinstance(e, CastExpr):
        return CastExpr(substitute(e.expr, mapping), e.cast_type)
    if isinstance(e, BinaryExpr):
        return BinaryExpr(substitute(e.left, mapping), e.op, substitute(e.right, mapping))
    if isinstance(e, IsNullExpr):
        return IsNullExpr(substitute(e.arg, mapping))
    if isinstance(e, IsNotNullExpr):
        return IsNotNullExpr(substitute(e.arg, mapping))
    if isinstance(e, NotExpr):
        return NotExpr(substitute(e.arg, mapping))
    if isinstance(e, (CaseExpr, InListExpr, DatePartExpr, SubstrExpr)):
        return e.map_children(lambda x: substitute(x, mapping))
    if isinstance(e, LikeExpr):
        return LikeExpr(substitute(e.expr, mapping), e.pattern, e.negated, e.case_insensitive)
    raise TypeError(f"unsupported expression node {type(e).__name__}")


def columns_of(e: PhysicalExpr, out=None) -> set:
    out = set() if out is None else out
    if isinstance(e, Column):
        out.add(e.name)
    for c in e.children():
        columns_of(c, out)
    return out


# --------------------------------------------------------------------------------- plan nodes
class ExecutionPlan:
    """execution_plan.rs:102.  `execute` returns the node's whole output partition as a device table
    (the pull-stream contract lives at the Rust boundary only)."""

    def name(self) -> str:
        return type(self).__name__

    def children(self) -> list:
        return []

    def with_new_children(self, children: list) -> "ExecutionPlan":
        raise NotImplementedError

    def execute(self, partition: int = 0) -> DeviceTable:
        raise NotImplementedError

    def detail(self) -> str:
        return ""

    def _run_child(self, child: "ExecutionPlan"):
        """(table, owned): leaf tables belong to the caller and are never freed by a plan"""
        t = child.execute()
        return t, not isinstance(child, MemoryExec)

    def _pass_through(self, child: "ExecutionPlan") -> DeviceTable:
        """output = the child's output; a leaf's table is handed on as a zero-copy view the parent may free"""
        t, owned = self._run_child(child)
        return t if owned else t.select(list(range(t.num_columns)))


class MemoryExec(ExecutionPlan):
    """DataSourceExec over an in-memory table (datasource/src/memory.rs:58) — here: a device-resident table.
    `projection` = the scan's column projection (DataSourceExec: projection=[...]); a zero-copy view."""

    def __init__(self, table: DeviceTable, name: str = "", projection=None):
        self.table, self.label, self.projection = table, name, projection
        self._view = None

    def project(self, columns) -> "MemoryExec":
        return MemoryExec(self.table, self.label, list(columns))

    def with_new_children(self, children):
        return self

    def execute(self, partition=0):
        if self.projection is None:
            return self.table
        if self._view is None:
            self._view = self.table.select(self.projection)     # owned by this node, like the table by the caller
        return self._view

    def detail(self):
        return f"{self.label} rows={self.table.num_rows}" + (f", projection={self.projection}" if self.projection else "")


class ParquetExec(ExecutionPlan):
    """DataSourceExec over a ParquetSource (datasource/src/source.rs:366, datasource-parquet) with the scan's column
    projection — here the GPU scan: every projected column chunk is decoded on the device (parquet.ParquetFile.read,
    dfgpu_parquet_decode_chunk).  The decoded table is owned by the node's consumer."""

    def __init__(self, path: str, projection=None, name: str = ""):
        self.path, self.projection, self.label = path, projection, name
        self.dynamic_bounds = {}   # column -> (lo, hi): published by a HashJoinExec above once its build side is known
        self.dynamic_in_lists = {} # column -> ascending distinct build keys of a small build side (PushdownStrategy::InList)
        self.dynamic_membership = {}  # column -> the built join table of a LARGE build side (PushdownStrategy::Map): asked `contains` per row group
        self.metrics = {}          # row_groups_total / row_groups_read of the last execute

    def project(self, columns) -> "ParquetExec":
        return ParquetExec(self.path, list(columns), self.label)

    def with_new_children(self, children):
        return self

    def execute(self, partition=0):
        from .parquet import read_table
        return read_table(self.path, self.projection, bounds=self.dynamic_bounds or None, stats=self.metrics, in_lists=self.dynamic_in_lists or None,
                          membership={k: v for k, v in self.dynamic_membership.items() if v is not None} or None)

    def detail(self):
        return f"{self.label or self.path}" + (f", projection={self.projection}" if self.projection else "")


class ArrowIpcExec(ExecutionPlan):
    """DataSourceExec over an ArrowSource (datasource-arrow/src/source.rs:260): an Arrow IPC file or stream scanned straight into
    HBM (ipc.read_table: no decoding, one H2D copy per buffer, batches concatenated on the device)"""

    def __init__(self, path: str, projection=None, name: str = ""):
        self.path, self.projection, self.label = path, projection, name
        self.metrics = {}

    def project(self, columns) -> "ArrowIpcExec":
        return ArrowIpcExec(self.path, list(columns), self.label)

    def children(self):
        return []

    def with_new_children(self, children):
        return self

    def execute(self, partition=0):
        from .ipc import read_table
        return read_table(self.path, self.projection, stats=self.metrics)

    def detail(self):
        return f"{self.label or self.path}" + (f", projection={self.projection}" if self.projection else "")


class _Unary(ExecutionPlan):
    def children(self):
        return [self.input]


class FilterExec(_Unary):
    """FilterExec::try_new(predicate, input) with an optional embedded projection (filter.rs:85)"""

    def __init__(self, predicate: PhysicalExpr, input: ExecutionPlan, projection=None):
        self.predicate, self.input, self.projection = predicate, input, projection

    def with_new_children(self, c):
        return FilterExec(self.predicate, c[0], self.projection)

    def execute(self, partition=0):
        t, owned = self._run_child(self.input)
        out = ops.filter(t, self.predicate, self.projection)
        if owned:
            t.free()
        return out

    def detail(self):
        return f"{self.predicate!r}" + (f", projection={self.projection}" if self.projection else "")


class ProjectionExec(_Unary):
    """ProjectionExec::try_new(exprs = [(expr, name)], input) (projection.rs:439)"""

    def __init__(self, exprs, input: ExecutionPlan):
        self.exprs, self.input = exprs, input

    def with_new_children(self, c):
        return ProjectionExec(self.exprs, c[0])

    def execute(self, partition=0):
        t, owned = self._run_child(self.input)
        out = ops.project(t, self.exprs)
        if owned:
            t.free()
        return out

    def detail(self):
        return ", ".join(f"{e!r} as {n}" for e, n in self.exprs)


class CoalesceBatchesExec(_Unary):
    """coalesce/mod.rs:63 — batch-size bookkeeping; a device partition is already one batch"""

    def __init__(self, input: ExecutionPlan, target_batch_size: int = 8192):
        self.input, self.target_batch_size = input, target_batch_size

    def with_new_children(self, c):
        return CoalesceBatchesExec(c[0], self.target_batch_size)

    def execute(self, partition=0):
        return self._pass_through(self.input)


class RepartitionExec(_Unary):
    """RepartitionExec::try_new(input, Partitioning::Hash(keys, n)) (repartition/mod.rs:1626).  One process per
    GPU: the exchange is RCCL (exchange.hash_exchange); with a single GPU there is one partition and nothing moves."""

    def __init__(self, input: ExecutionPlan, keys, n_partitions: int, group=None):
        self.input, self.keys, self.n_partitions, self.group = input, keys, n_partitions, group

    def with_new_children(self, c):
        return RepartitionExec(c[0], self.keys, self.n_partitions, self.group)

    def execute(self, partition=0):
        from .queries import _repartition, _world
        if _world(self.group) == 1:
            return self._pass_through(self.input)
        t, owned = self._run_child(self.input)
        out = _repartition(t, self.keys, self.group)
        if owned:
            t.free()
        return out

    def detail(self):
        return f"Hash({self.keys}, {self.n_partitions})"


class CoalescePartitionsExec(_Unary):
    """CoalescePartitionsExec::new(input) (coalesce_partitions.rs:50): all input partitions into one, in arrival
    order.  One partition per GPU: with one GPU nothing happens, with several every rank receives the
    concatenation of the ranks' partitions in rank order — a device all-gather (dfgpu_exchange_broadcast)."""

    def __init__(self, input: ExecutionPlan, group=None):
        self.input, self.group = input, group

    def with_new_children(self, c):
        return CoalescePartitionsExec(c[0], self.group)

    def execute(self, partition=0):
        from .queries import _world
        if _world(self.group) == 1:
            return self._pass_through(self.input)
        from .exchange import broadcast_table
        t, owned = self._run_child(self.input)
        out = broadcast_table(t, self.group)
        if owned and out is not t:
            t.free()
        return out


class SortPreservingMergeExec(_Unary):
    """SortPreservingMergeExec::new(expr, input).with_fetch(fetch) (sorts/sort_preserving_merge.rs:91): k-way merge
    of the sorted partitions.  One partition per GPU: a no-op on one GPU, otherwise queries._merge_sorted."""

    def __init__(self, expr, input: ExecutionPlan, fetch=None, group=None):
        self.expr, self.input, self.fetch, self.group = expr, input, fetch, group

    def with_new_children(self, c):
        return SortPreservingMergeExec(self.expr, c[0], self.fetch, self.group)

    def execute(self, partition=0):
        from .queries import _merge_sorted, _world
        if _world(self.group) == 1:
            return self._pass_through(self.input)
        t, owned = self._run_child(self.input)
        out = _merge_sorted(t, self.expr, self.fetch, self.group)
        if owned and out is not t:
            t.free()
        return out

    def detail(self):
        return str([(c, "DESC" if d else "ASC") for c, d, _ in self.expr]) + (f", fetch={self.fetch}" if self.fetch is not None else "")


class HashJoinExec(ExecutionPlan):
    """HashJoinExec::try_new(left = build, right = probe, on, filter = None, join_type, projection, mode,
    null_equality, null_aware) (joins/hash_join/exec.rs:752).  `projection` = (build columns, probe columns);
    `null_aware` = NOT IN semantics for LeftAnti / RightAnti on one key column (exec.rs:429-455)."""

    def __init__(self, left: ExecutionPlan, right: ExecutionPlan, on, join_type="Inner", projection=None, null_equality="NullEqualsNothing",
                 probe_mode=0, filter=None, null_aware=False):
        self.left, self.right, self.on, self.join_type = left, right, on, join_type
        self.projection, self.null_equality, self.probe_mode = projection, null_equality, probe_mode
        self.filter = filter   # JoinFilter: (expression over f0, f1, ..., [(column index, "Left" | "Right"), ...])
        self.null_aware = null_aware

    def children(self):
        return [self.left, self.right]

    def with_new_children(self, c):
        return HashJoinExec(c[0], c[1], self.on, self.join_type, self.projection, self.null_equality, self.probe_mode, self.filter, self.null_aware)

    # probe rows without a build match are dropped by these join types (JoinType::on_lr_is_preserved, common/src/join_type.rs:115-127:
    # the probe side accepts a pushed-down filter), so the build side's key bounds may prune the probe-side scan
    _DYNAMIC_FILTER_JOINS = ("Inner", "Left", "LeftSemi", "RightSemi", "LeftAnti", "LeftMark")
    _BUILD_EMITTING = ("Left", "Full", "LeftSemi", "LeftAnti", "LeftMark")      # join types that report build rows by their visited marks

    def _publish_membership(self, ht):
        """the Map strategy (hash_join/exec.rs:2727-2751: a build side beyond the IN-list limits pushes the hash table itself):
        the probe-side ParquetExec asks the built table `contains` for every row group's key chunk before it reads anything else.
        Only where the bounds were published (same join types / key restrictions) and no IN list was (that is the small-build case)."""
        node = self.right
        while isinstance(node, (FilterExec, CoalesceBatchesExec, RepartitionExec)):
            node = node.input
        if not isinstance(node, ParquetExec):
            return None
        probe_key = self.on[0][1]
        from .queries import _world
        if probe_key in node.dynamic_bounds and probe_key not in node.dynamic_in_lists and _world() == 1 and self.filter is None:
            node.dynamic_membership[probe_key] = ht
            return node, probe_key        # the caller withdraws the entry before it frees the table (the scan does not own it)
        node.dynamic_membership.pop(probe_key, None)
        return None

    def _publish_dynamic_bounds(self, build_table):
        """the join's dynamic filter (HashJoinExec::create_dynamic_filter, hash_join/exec.rs:869-875; bounds accumulated in
        hash_join/shared_bounds.rs:277-284): [min, max] of the build keys, handed to the probe-side scan, which prunes row groups
        with it before reading them.  Single integer key, no JoinFilter-independent restrictions of the reference apply here
        (one partition); a null-aware anti join never publishes (exec.rs:883-892)."""
        if self.join_type not in self._DYNAMIC_FILTER_JOINS or len(self.on) != 1 or self.null_aware or self.null_equality != "NullEqualsNothing":
            return
        node = self.right
        while isinstance(node, (FilterExec, CoalesceBatchesExec, RepartitionExec)):
            node = node.input
        if not isinstance(node, ParquetExec):
            return
        build_key, probe_key = self.on[0]
        if node.projection is not None and probe_key not in node.projection:
            return
        import pyarrow as pa
        ktype = build_table.schema.field(build_table.index_of(build_key)).type
        if ktype not in (pa.int64(), pa.int32()):
            return
        lo, hi, n, _ = ops.column_minmax(build_table, build_key)
        from .queries import _world
        if _world() > 1:
            # one process per GPU: this rank's build partition holds only the keys routed to it, while its probe-side scan still
            # holds rows bound for every rank — the filter must cover ALL partitions' build keys, as the reference's shared
            # accumulator waits for every partition before it updates the filter (hash_join/shared_bounds.rs: SharedBuildAccumulator)
            import torch.distributed as dist
            parts = [None] * dist.get_world_size()
            dist.all_gather_object(parts, (lo, hi, n))
            seen = [(a, b) for a, b, c in parts if c]
            n = len(seen)
            if n:
                lo, hi = min(a for a, _ in seen), max(b for _, b in seen)
        node.dynamic_bounds[probe_key] = (lo, hi) if n else (1, 0)     # an empty build side prunes the whole scan
        if _world() == 1:
            # the membership half (PushdownStrategy::InList for small build sides, else Map = bounds only); with several ranks the
            # lists of all partitions would have to be merged like the bounds — they stay on bounds
            values = ops.column_inlist(build_table, build_key)
            if values is not None:
                node.dynamic_in_lists[probe_key] = values
            else:
                node.dynamic_in_lists.pop(probe_key, None)     # a larger build side this time (Map strategy): a list of an earlier run must not prune

    def _probe(self, ht, probe_table, predicate=None):
        bc, pc = self.projection if self.projection else (None, None)
        return ht.probe(probe_table, [r for _, r in self.on], self.join_type, bc, pc, predicate=predicate, join_filter=self.filter)

    def execute(self, partition=0, probe_predicate=None):
        if self.filter is not None or self.join_type not in ("Inner", "RightSemi", "RightAnti", "Right", "RightMark"):
            # build-side emission (Left / Full / LeftSemi / LeftAnti / LeftMark): the general path of ops.hash_join
            assert probe_predicate is None
            b, bo = self._run_child(self.left)
            self._publish_dynamic_bounds(b)
            p, po = self._run_child(self.right)
            bc, pc = self.projection if self.projection else (None, None)
            shared_build = self.join_type in self._BUILD_EMITTING and replicated_build(self.left)
            if shared_build:
                # PartitionMode::CollectLeft with build-side emission: the reference's probe partitions mark ONE shared visited bitmap
                # and the last of them reports the build rows (hash_join/exec.rs:1312-1330, stream.rs ProcessUnmatchedBuild).  Here the
                # probe partitions sit on different GPUs and the build side is replicated: every rank probes ITS probe rows, the copies'
                # visited marks (and null-aware flags) are OR-ed across the ranks (dfgpu_exchange_join_visited), and the build rows the
                # marks select — the same on every rank, in build-row order — are split by position.  Nothing is computed twice, and
                # no rank depends on the order another rank's atomics built its chains in.
                import torch.distributed as dist

                from .exchange import comm_for
                ht = ops.JoinHashTable(b, [l for l, _ in self.on], self.null_equality, null_aware=self.null_aware)
                matched = ht.probe(p, [r for _, r in self.on], self.join_type, bc, pc, join_filter=self.filter)
                comm_for().merge_join_visited(ht)
                tail = ht.emit_unmatched(self.join_type, bc, ops.tail_probe_schema(p, self.join_type, pc))
                n, world, rank = tail.num_rows, dist.get_world_size(), dist.get_rank()
                mine = tail.slice(n * rank // world, n * (rank + 1) // world - n * rank // world)
                tail.free()
                if self.join_type in ("Left", "Full"):
                    out = ops.concat_tables([matched, mine])
                    mine.free()
                else:
                    out = mine
                matched.free()
                ht.free()
            else:
                out = ops.hash_join(b, p, self.on, self.join_type, self.null_equality, bc, pc, join_filter=self.filter, null_aware=self.null_aware)
            for t, o in ((b, bo), (p, po)):
                if o:
                    t.free()
            return out
        b, bo = self._run_child(self.left)
        ht = ops.JoinHashTable(b, [l for l, _ in self.on], self.null_equality, probe_mode=self.probe_mode, null_aware=self.null_aware)
        self._publish_dynamic_bounds(b)
        published = self._publish_membership(ht) if self.join_type in ("Inner", "RightSemi") and len(self.on) == 1 else None
        try:
            p, po = self._run_child(self.right)
        finally:
            # the membership entry is the live join table: it is withdrawn (the key stays, as a record that a table was pushed; the
            # scan skips entries without a table) once the probe child has run — a later execution of that scan, on its own or
            # through a re-used node, must not ask a freed table
            if published is not None and published[1] in published[0].dynamic_membership:
                published[0].dynamic_membership[published[1]] = None
        out = self._probe(ht, p, probe_predicate)
        ht.free()
        for t, o in ((b, bo), (p, po)):
            if o:
                t.free()
        return out

    def detail(self):
        return f"join_type={self.join_type}, on={self.on}" + (f", projection={self.projection}" if self.projection else "") + \
            (", null_aware" if self.null_aware else "")


def replicated_build(node: ExecutionPlan) -> bool:
    """several ranks and the build side is a CoalescePartitionsExec (every rank holds ALL build rows: PartitionMode::CollectLeft)"""
    from .queries import _world
    if _world() == 1:
        return False
    while isinstance(node, CoalesceBatchesExec):
        node = node.input
    return isinstance(node, CoalescePartitionsExec)


def _partial_below(node: ExecutionPlan):
    """the Partial aggregate feeding a Final one, through the exchange / bookkeeping nodes between them"""
    while node is not None:
        if isinstance(node, (AggregateExec, GpuFusedAggregateExec)):
            return node if node.mode == "Partial" else None
        kids = node.children()
        node = kids[0] if len(kids) == 1 else None
    return None


class AggregateExec(_Unary):
    """AggregateExec::try_new(mode, group_by = [(expr, name)], aggr_expr = [(func, arg | None, name)], input)
    (aggregates/mod.rs:839)"""

    def __init__(self, mode: str, group_by, aggr_expr, input: ExecutionPlan):
        self.mode, self.group_by, self.aggr_expr, self.input = mode, group_by, aggr_expr, input
        self.return_types = None    # Partial: the aggregates' declared return types, typed over the raw input when it runs

    def with_new_children(self, c):
        return AggregateExec(self.mode, self.group_by, self.aggr_expr, c[0])

    def execute(self, partition=0):
        t, owned = self._run_child(self.input)
        rt = None
        if self.mode == "Partial":
            self.return_types = ops.aggregate_return_types(t, self.aggr_expr)
        elif self.mode in ("Final", "FinalPartitioned"):
            # AggregateFunctionExpr::return_field: the reference's Final node carries the types its Partial twin was planned with
            below = _partial_below(self.input)
            rt = below.return_types if below is not None else None
        out = ops.aggregate(t, self.group_by, self.aggr_expr, self.mode, return_types=rt)
        if owned:
            t.free()
        return out

    def detail(self):
        return f"mode={self.mode}, gby=[{', '.join(n for _, n in self.group_by)}], aggr=[{', '.join(n for _, _, n in self.aggr_expr)}]"


class SortExec(_Unary):
    """SortExec::new(expr = [(column, descending, nulls_first)], input).with_fetch(fetch) (sorts/sort.rs:1366);
    with fetch it is the reference's TopK (topk/mod.rs:397)"""

    def __init__(self, expr, input: ExecutionPlan, fetch=None):
        self.expr, self.input, self.fetch = expr, input, fetch

    def with_new_children(self, c):
        return SortExec(self.expr, c[0], self.fetch)

    def execute(self, partition=0):
        t, owned = self._run_child(self.input)
        out = ops.sort(t, self.expr, self.fetch)
        if owned:
            t.free()
        return out

    def detail(self):
        return ("TopK(fetch=%d), " % self.fetch if self.fetch is not None else "") + str([(c, "DESC" if d else "ASC") for c, d, _ in self.expr])


# ------------------------------------------------------------------------------ fused GPU nodes
class ScalarSubqueryExec(ExecutionPlan):
    """physical-plan/src/scalar_subquery.rs:85 — a pass-through over its main input that first runs every (uncorrelated) scalar
    subquery exactly once and stores its value where the ScalarSubqueryExprs of the main plan read it.  A subquery returns zero
    rows (NULL) or one; more is the reference's "Scalar subquery returned more than one row" error."""

    def __init__(self, input: ExecutionPlan, subqueries, results):
        self.input, self.subqueries, self.results = input, list(subqueries), results   # subqueries: [(plan, index)]

    def children(self):
        return [self.input] + [p for p, _ in self.subqueries]

    def with_new_children(self, c):
        return ScalarSubqueryExec(c[0], [(p, i) for p, (_, i) in zip(c[1:], self.subqueries)], self.results)

    def execute(self, partition=0):
        for plan, index in self.subqueries:
            t, owned = self._run_child(plan)
            if t.num_rows > 1:
                raise _lib.DfgpuError("Scalar subquery returned more than one row")
            self.results[index] = None if t.num_rows == 0 else t.select([0]).to_arrow().column(0)[0].as_py()
            if owned:
                t.free()
        return self._pass_through(self.input)

    def detail(self):
        return f"subqueries={len(self.subqueries)}"


class GpuFusedAggregateExec(_Unary):
    """FilterExec + ProjectionExec + AggregateExec as one node: predicate, inlined argument expressions and
    accumulation in a single pass over the input's referenced columns (dfgpu_agg_update_filtered; the kernel is
    specialised for the forest with hiprtc for large inputs, jit.hip).  Output schema = the AggregateExec's."""

    def __init__(self, mode, group_by, aggr_expr, predicate, input: ExecutionPlan):
        self.mode, self.group_by, self.aggr_expr, self.predicate, self.input = mode, group_by, aggr_expr, predicate, input
        self.return_types = None

    def with_new_children(self, c):
        return GpuFusedAggregateExec(self.mode, self.group_by, self.aggr_expr, self.predicate, c[0])

    def execute(self, partition=0):
        t, owned = self._run_child(self.input)
        if self.mode == "Partial":
            self.return_types = ops.aggregate_return_types(t, self.aggr_expr)
        out = ops.aggregate(t, self.group_by, self.aggr_expr, self.mode, predicate=self.predicate)
        if owned:
            t.free()
        return out

    def detail(self):
        return f"mode={self.mode}, predicate={self.predicate!r}, gby=[{', '.join(n for _, n in self.group_by)}], aggr=[{', '.join(n for _, _, n in self.aggr_expr)}]"


class GpuHashJoinExec(HashJoinExec):
    """HashJoinExec with the FilterExec of its probe side fused below it (dfgpu_join_probe_filtered)"""

    def __init__(self, join: HashJoinExec, probe_predicate: PhysicalExpr, probe_input: ExecutionPlan, probe_mode=None):
        super().__init__(join.left, probe_input, join.on, join.join_type, join.projection, join.null_equality,
                         join.probe_mode if probe_mode is None else probe_mode, null_aware=join.null_aware)
        self.probe_predicate = probe_predicate

    def with_new_children(self, c):
        j = HashJoinExec(c[0], c[1], self.on, self.join_type, self.projection, self.null_equality, self.probe_mode, self.filter, self.null_aware)
        return GpuHashJoinExec(j, self.probe_predicate, c[1])

    def execute(self, partition=0):
        return super().execute(partition, probe_predicate=self.probe_predicate)

    def detail(self):
        return super().detail() + f", probe_predicate={self.probe_predicate!r}"


# --------------------------------------------------------------------------------------- the rule
class GpuOffloadRule:
    """PhysicalOptimizerRule (session/src/physical_optimizer.rs:52-84), appended with
    SessionStateBuilder::with_physical_optimizer_rule (core/src/execution/session_state.rs:1407-1415) so that it
    runs after the built-in rules have fixed distribution, ordering and join sides.  `unordered_probe=True` lets
    joins whose parent does not need the probe-side order (an aggregate or a repartition) take the single-pass
    unordered probe where it applies (probe_mode "order_not_needed": at most one match per probe row, non-nullable
    payload; the library falls back to the general path otherwise) — the rule knows the parent because it rewrites bottom-up and fixes the child when it
    visits the parent."""

    def __init__(self, world_size: int = 1, unordered_probe: bool = True):
        self.world_size, self.unordered_probe = world_size, unordered_probe
        self.declined = []    # (node, reason) of operators the rule left to the CPU

    def _admits(self, node) -> bool:
        """spill-aware fallback: a hash join whose device footprint the pool cannot admit stays the reference's CPU
        operator, which can spill (MemoryReservation::try_grow failing is how the reference itself learns it, hash_join/exec.rs:2608).
        The estimate uses row-count upper bounds of the children (their statistics) and dfgpu_join_estimate_bytes."""
        if _lib._initialised_device is None:
            return True       # planning without a device (tests of the rewrite shapes): nothing to admit against
        try:
            b_rows, b_bytes = _row_bound(node.left)
            p_rows, p_bytes = _row_bound(node.right)
        except (TypeError, AttributeError):
            return True       # no statistics: decided at run time by the build's own reservation
        out_row = (b_bytes + p_bytes)
        need = ops.join_estimate_bytes(b_rows, b_bytes, p_rows, -1, out_row)
        try:
            ops.Reservation(need).release()
            return True
        except _lib.DfgpuError as e:
            self.declined.append((node, str(e)))
            return False

    def name(self) -> str:
        return "gpu_offload_amd"

    def schema_check(self) -> bool:
        return True

    def optimize(self, plan: ExecutionPlan) -> ExecutionPlan:
        return self._rewrite(plan, parent_needs_order=True)

    # transform_up: children first (tree_node.rs), then this node.  `needs_order` = some ancestor observes this
    # node's output order (maintains_input_order / required_input_ordering in the reference's terms)
    def _rewrite(self, node: ExecutionPlan, parent_needs_order: bool) -> ExecutionPlan:
        kids = []
        for i, c in enumerate(node.children()):
            if isinstance(node, (AggregateExec, GpuFusedAggregateExec, RepartitionExec, SortExec)):
                need = False                                   # these consume their input in any order
            elif isinstance(node, HashJoinExec):
                need = parent_needs_order and i == 1           # only the probe side's order shows in the output
            else:
                need = parent_needs_order                      # FilterExec / ProjectionExec / CoalesceBatchesExec keep it
            kids.append(self._rewrite(c, need))
        if kids:
            node = node.with_new_children(kids)
        # bookkeeping nodes that have no meaning for whole-partition device tables
        if isinstance(node, CoalesceBatchesExec):
            return node.input
        if isinstance(node, (RepartitionExec, CoalescePartitionsExec, SortPreservingMergeExec)) and self.world_size == 1:
            return node.input                                  # one partition: nothing to exchange, gather or merge
        if not isinstance(node, (GpuHashJoinExec, GpuFusedAggregateExec)) and not getattr(node, "kept_on_cpu", False):
            try:
                reason = unsupported_reason(node)
            except Exception:  # noqa: BLE001 - a plan this layer cannot type (stand-in leaves of planning-only tests): decided at run time
                reason = None
            if reason is not None:                             # the reference's operator stays (it runs these inputs)
                node.kept_on_cpu = True
                self.declined.append((node, reason))
                return node
        if isinstance(node, HashJoinExec) and not isinstance(node, GpuHashJoinExec) and not self._admits(node):
            node.kept_on_cpu = True
            return node
        if isinstance(node, HashJoinExec) and not isinstance(node, GpuHashJoinExec):
            probe_mode = ops.PROBE_MODES["order_not_needed"] if (self.unordered_probe and not parent_needs_order and
                                                                      node.join_type in ("Inner", "RightSemi", "RightAnti")) else node.probe_mode
            node = HashJoinExec(node.left, node.right, node.on, node.join_type, node.projection, node.null_equality,
                                node.probe_mode if node.filter is not None else probe_mode, node.filter, node.null_aware)
            probe = node.right
            if node.filter is None and isinstance(probe, FilterExec) and node.join_type in ("Inner", "RightSemi", "RightAnti", "Right", "RightMark"):
                needed = set(r for _, r in node.on) | set((node.projection or (None, None))[1] or [])
                # the FilterExec's embedded projection must keep what the join reads (it always does in a valid plan)
                if probe.projection is None or needed <= set(probe.projection):
                    return GpuHashJoinExec(node, probe.predicate, probe.input, probe_mode)
            return node
        if isinstance(node, AggregateExec) and node.mode in ("Single", "SinglePartitioned", "Partial"):
            child, mapping, predicate = node.input, {}, None
            if isinstance(child, ProjectionExec):
                mapping = {n: e for e, n in child.exprs}
                child = child.input
            if isinstance(child, FilterExec):
                predicate = child.predicate
                child = child.input
            if mapping or predicate is not None:
                gb = [(substitute(e, mapping), n) for e, n in node.group_by]
                aggs = [(f, None if e is None else substitute(e, mapping), n) for f, e, n in node.aggr_expr]
                return GpuFusedAggregateExec(node.mode, gb, aggs, predicate, child)
        return node


def _row_bound(node):
    """(upper bound of the output rows, bytes per row) of a subtree from its leaves' statistics"""
    if isinstance(node, MemoryExec):
        t = node.execute()
        return t.num_rows, (t.nbytes() // max(1, t.num_rows))
    if isinstance(node, (FilterExec, ProjectionExec, CoalesceBatchesExec, RepartitionExec, CoalescePartitionsExec, SortExec, SortPreservingMergeExec,
                         AggregateExec, GpuFusedAggregateExec)):
        return _row_bound(node.children()[0])
    if isinstance(node, HashJoinExec):
        (br, bb), (pr, pb) = _row_bound(node.left), _row_bound(node.right)
        return max(br, pr), bb + pb
    raise TypeError(node.name())


# ------------------------------------------------------------------------------ plan-time schemas and declines
def plan_schema(node):
    """Output schema of a plan node (pa.Schema), or None when it cannot be told at plan time (a scan that has not opened its file,
    a node below one that could not be typed).  What ExecutionPlan::schema() is in the reference; here it only serves the rule's
    decisions — expression types come from the library (dfgpu_expr_type over an empty table of the input schema)."""
    import pyarrow as pa
    if isinstance(node, MemoryExec):
        sch = node.table.schema
        if node.projection is None:
            return sch
        idx = [node.table.index_of(c) for c in node.projection]
        return pa.schema([sch.field(i) for i in idx])
    if isinstance(node, GpuHashJoinExec) or isinstance(node, HashJoinExec):
        l, r = plan_schema(node.left), plan_schema(node.right)
        if l is None or r is None:
            return None
        bc, pc = node.projection if node.projection is not None else (None, None)
        lf = list(l) if bc is None else [l.field(l.get_field_index(c)) if isinstance(c, str) else l.field(c) for c in bc]
        rf = list(r) if pc is None else [r.field(r.get_field_index(c)) if isinstance(c, str) else r.field(c) for c in pc]
        jt = node.join_type
        if jt in ("LeftSemi", "LeftAnti"):
            return pa.schema(lf)
        if jt in ("RightSemi", "RightAnti"):
            return pa.schema(rf)
        if jt == "LeftMark":
            return pa.schema(lf + [pa.field("mark", pa.bool_())])
        if jt == "RightMark":
            return pa.schema(rf + [pa.field("mark", pa.bool_())])
        return pa.schema(lf + rf)
    kids = node.children()
    if len(kids) != 1:
        return None
    inp = plan_schema(kids[0])
    if inp is None:
        return None
    if isinstance(node, FilterExec):
        if node.projection is None:
            return inp
        return pa.schema([inp.field(inp.get_field_index(c)) if isinstance(c, str) else inp.field(c) for c in node.projection])
    if isinstance(node, (CoalesceBatchesExec, RepartitionExec, CoalescePartitionsExec, SortPreservingMergeExec, SortExec)):
        return inp
    try:
        empty = DeviceTable.from_arrow(inp.empty_table())
    except Exception:  # noqa: BLE001 - no device (planning-only tests): nothing can be typed
        return None
    try:
        if isinstance(node, ProjectionExec):
            return pa.schema([pa.field(n, ops.expr_type(empty, e)) for e, n in node.exprs])
        if isinstance(node, (AggregateExec, GpuFusedAggregateExec)):
            if node.mode in ("Final", "FinalPartitioned"):
                # a Final node's aggregate expressions are its Partial twin's (over the RAW input): typed there, named here
                below = _partial_below(node.input)
                if below is None:
                    return None
                raw = plan_schema(AggregateExec("Single", below.group_by, below.aggr_expr, below.input))
                if raw is None or len(raw) != len(node.group_by) + len(node.aggr_expr):
                    return None
                names = [n for _, n in node.group_by] + [n for _, _, n in node.aggr_expr]
                return pa.schema([pa.field(n, f.type) for n, f in zip(names, raw)])
            fields = [pa.field(n, ops.expr_type(empty, e)) for e, n in node.group_by]
            for func, e, n in node.aggr_expr:
                t = None if e is None else ops.expr_type(empty, e)
                if node.mode == "Partial":      # state fields (sum.rs:281-301, average.rs:317-360, count.rs): AVG = count + sum
                    if func == "avg":
                        fields.append(pa.field(n + "[count]", pa.uint64()))
                        # avg_sum_data_type (average.rs:131-172): the input precision + 13 digits, never narrower than Decimal128's 38
                        fields.append(pa.field(n + "[sum]", pa.decimal128(38, t.scale) if pa.types.is_decimal128(t) else pa.float64()))
                    else:
                        fields.append(pa.field(n, _agg_type(func, t)))
                else:
                    fields.append(pa.field(n, _agg_type(func, t)))
            return pa.schema(fields)
    except (_lib.DfgpuError, KeyError, TypeError, ValueError):
        return None   # an expression this layer cannot type: the node's schema stays unknown and nothing is decided on it
    finally:
        empty.free()
    return None


def _sum_type(t):
    import pyarrow as pa
    if pa.types.is_decimal128(t):
        return pa.decimal128(min(38, t.precision + 10), t.scale)      # sum.rs:232-260
    if pa.types.is_floating(t):
        return pa.float64()
    return pa.uint64() if pa.types.is_unsigned_integer(t) and t != pa.uint8() else pa.int64()


def _agg_type(func, t):
    import pyarrow as pa
    if func == "count":
        return pa.int64()
    if func == "sum":
        return _sum_type(t)
    if func == "avg":
        return pa.decimal128(min(38, t.precision + 4), min(38, t.scale + 4)) if pa.types.is_decimal128(t) else pa.float64()   # average.rs:219-252
    return t    # min / max


def _key_bits(t):
    """upper bound of the bits a sort key column of this type takes in the packed key (sort.hip packs value ranges; the type is
    what is known at plan time), plus one bit for NULL placement"""
    import pyarrow as pa
    if pa.types.is_boolean(t):
        return 2
    if pa.types.is_string(t) or pa.types.is_large_string(t) or pa.types.is_dictionary(t):
        return 33
    return t.bit_width + 1


def unsupported_reason(node):
    """Why an operator the reference can run has no device form (None = it has one).  The library refuses these inputs when it
    meets them at run time (the DFGPU_CHECKs named below); the rule asks first, so that such an operator STAYS the reference's CPU
    operator instead of failing the query — every entry here has a test that feeds a plan through GpuOffloadRule."""
    import pyarrow as pa
    if isinstance(node, (AggregateExec, GpuFusedAggregateExec)) and node.mode in ("Single", "SinglePartitioned", "Partial"):
        inp = plan_schema(node.input)
        if inp is None:
            return None
        try:
            empty = DeviceTable.from_arrow(inp.empty_table())
        except Exception:  # noqa: BLE001
            return None
        try:
            for func, e, n in node.aggr_expr:
                if e is None:
                    continue
                try:
                    t = ops.expr_type(empty, e)
                except _lib.DfgpuError as err:
                    return f"{func}({e!r}): {err}"
                except (KeyError, TypeError, ValueError):
                    continue   # not typable from this schema (a name the input does not carry): decided at run time
                if func == "avg" and pa.types.is_decimal128(t) and t.precision + 13 > 38:
                    # aggregate.hip avg_sum_type: avg_sum_data_type (average.rs:131-172) widens to Decimal256 beyond 38 digits
                    return f"AVG({n}) over {t} accumulates in Decimal256 in the reference (no device representation)"
                if func in ("min", "max") and pa.types.is_decimal128(t) and t.precision > 18:
                    # aggregate.hip wide_minmax_values_fit: the device compares 64-bit words; values beyond them are only found at run time
                    return f"{func.upper()}({n}) over {t}: values of more than 18 digits do not fit the device's 64-bit comparison"
                if func in ("sum", "avg", "min", "max") and (pa.types.is_boolean(t) or pa.types.is_string(t) or pa.types.is_large_string(t)):
                    return f"{func.upper()}({n}) over {t} is not supported on the GPU path"
        finally:
            empty.free()
        return None
    if isinstance(node, SortExec):
        inp = plan_schema(node.input)
        if inp is None:
            return None
        bits = sum(_key_bits(inp.field(inp.get_field_index(c) if isinstance(c, str) else c).type) for c, _, _ in node.expr)
        if bits > 192:   # sort.hip: "packed sort key longer than 192 bits"
            return f"sort key of up to {bits} bits: the device packs at most 192"
        return None
    if isinstance(node, HashJoinExec):
        try:
            b_rows, _ = _row_bound(node.left)
        except (TypeError, AttributeError):
            return None
        if b_rows >= 0xFFFFFFFF:   # join.hip: the reference switches to JoinHashMapU64 (joins/join_hash_map.rs:224); row ids here are 32 bits
            return f"build side of up to {b_rows} rows: the device addresses build rows with 32 bits (JoinHashMapU64 is not built)"
        return None
    return None


def displayable(plan: ExecutionPlan, indent: int = 0) -> str:
    """DisplayableExecutionPlan::indent (display.rs): one line per node, children indented"""
    line = "  " * indent + plan.name() + (": " + plan.detail() if plan.detail() else "")
    return "\n".join([line] + [displayable(c, indent + 1) for c in plan.children()])


def collect(plan: ExecutionPlan) -> DeviceTable:
    """physical_plan::collect for a single-partition plan; the caller owns the result"""
    out = plan.execute(0)
    return out.select(list(range(out.num_columns))) if isinstance(plan, MemoryExec) else out
