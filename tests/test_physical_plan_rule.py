"""GpuOffloadRule rewrites (no GPU needed: the rule only looks at the plan's shape).  The plans are the reference's
pinned TPC-H Q1 / Q3 physical plans (tpch/plans/q1.slt.part:50-58, q3.slt.part:61-76) as built in
tests/test_gpu_physical_plan.py, over stub leaf tables."""
from tests.test_gpu_physical_plan import names, q1_plan, q3_plan


class StubTable:
    def __init__(self, n):
        self.num_rows = n


def test_q1_plan_rewrite_shape_and_display():
    from datafusion_amd import physical_plan as P
    plan = q1_plan(StubTable(6_001_215))
    assert names(plan) == ["SortPreservingMergeExec", "SortExec", "AggregateExec", "CoalesceBatchesExec", "RepartitionExec", "AggregateExec", "ProjectionExec",
                           "CoalesceBatchesExec", "FilterExec", "MemoryExec"]
    rule = P.GpuOffloadRule()
    opt = rule.optimize(plan)
    assert names(opt) == ["SortExec", "AggregateExec", "GpuFusedAggregateExec", "MemoryExec"]
    fused = opt.children()[0].children()[0]
    assert fused.mode == "Partial" and fused.predicate is not None
    # the ProjectionExec's __common_expr_1 is inlined into the aggregate arguments
    txt = P.displayable(opt)
    assert "__common_expr_1" not in txt.split("GpuFusedAggregateExec")[1].split("\n")[0].split("aggr=")[0]
    assert "GpuFusedAggregateExec: mode=Partial, predicate=(l_shipdate@None <= " in txt
    # with more than one GPU the RepartitionExec stays (it becomes the RCCL exchange)
    multi = names(P.GpuOffloadRule(world_size=8).optimize(plan))
    assert "RepartitionExec" in multi and multi[0] == "SortPreservingMergeExec"


def test_q3_plan_rewrite_shape():
    from datafusion_amd import ops, physical_plan as P
    plan = q3_plan(StubTable(15_000), StubTable(150_000), StubTable(600_000))
    assert names(plan).count("RepartitionExec") == 4 and names(plan).count("CoalesceBatchesExec") == 9
    opt = P.GpuOffloadRule().optimize(plan)
    assert names(opt) == ["ProjectionExec", "SortExec", "AggregateExec", "GpuHashJoinExec", "GpuHashJoinExec", "FilterExec", "MemoryExec", "MemoryExec", "MemoryExec"]
    inner = opt.children()[0].children()[0].children()[0]
    semi = inner.children()[0]
    assert inner.join_type == "Inner" and semi.join_type == "RightSemi"
    assert inner.probe_mode == semi.probe_mode == ops.PROBE_MODES["order_not_needed"]
    assert repr(inner.probe_predicate).startswith("(l_shipdate@None > ") and repr(semi.probe_predicate).startswith("(o_orderdate@None < ")
    # ordered probes when the rule is told not to reorder
    keep = P.GpuOffloadRule(unordered_probe=False).optimize(plan)
    assert all(n.probe_mode == 0 for n in _walk(keep) if isinstance(n, P.HashJoinExec))


def test_join_with_a_join_filter_keeps_its_probe_side_filter_separate():
    from datafusion_amd import physical_plan as P
    from datafusion_amd.expr import col, lit
    jf = (col("f0") > col("f1"), [(1, "Left"), (1, "Right")])
    j = P.HashJoinExec(P.MemoryExec(StubTable(1)), P.FilterExec(col("w") > lit(1), P.MemoryExec(StubTable(1))), [("k", "k2")], "Inner", filter=jf)
    opt = P.GpuOffloadRule().optimize(P.AggregateExec("Single", [], [("count", None, "c")], j))
    assert names(opt) == ["AggregateExec", "HashJoinExec", "MemoryExec", "FilterExec", "MemoryExec"]
    assert opt.children()[0].filter is jf


def _walk(plan):
    yield plan
    for ch in plan.children():
        yield from _walk(ch)


def test_dynamic_bounds_are_published_only_where_the_reference_allows_it(monkeypatch):
    """HashJoinExec._publish_dynamic_bounds (the join's dynamic filter, hash_join/exec.rs:877-925 allow_join_dynamic_filter_pushdown +
    shared_bounds.rs:277-284): only join types whose probe side is preserved, never null-aware anti joins or NullEqualsNull, one
    integer key that the scan projects; an empty build side publishes an empty range"""
    import pyarrow as pa
    from datafusion_amd import ops, physical_plan as P
    from datafusion_amd.expr import col, lit

    class Build:
        def __init__(self, typ=pa.int64()):
            self.schema = pa.schema([pa.field("bk", typ), pa.field("v", pa.int32())])
            self.num_rows = 10

        def index_of(self, c):
            return c if isinstance(c, int) else self.schema.names.index(c)
    calls = []
    monkeypatch.setattr(ops, "column_minmax", lambda t, c: (calls.append(c), (100, 200, 10, True))[1])
    monkeypatch.setattr(ops, "column_inlist", lambda t, c, **kw: None)      # a large build side: the Map strategy, bounds only

    def scan_for(join_type, key_type=pa.int64(), projection=("pk", "w"), **kw):
        scan = P.ParquetExec("/nonexistent.parquet", list(projection) if projection else None, "probe")
        probe = P.CoalesceBatchesExec(P.FilterExec(col("w") > lit(1), scan))
        j = P.HashJoinExec(P.MemoryExec(StubTable(10)), probe, [("bk", "pk")], join_type, **kw)
        j._publish_dynamic_bounds(Build(key_type))
        return scan
    for jt in ("Inner", "Left", "LeftSemi", "RightSemi", "LeftAnti", "LeftMark"):
        assert scan_for(jt).dynamic_bounds == {"pk": (100, 200)}, jt
    for jt in ("Right", "Full", "RightAnti", "RightMark"):                       # unmatched probe rows are output: no pruning
        assert scan_for(jt).dynamic_bounds == {}, jt
    assert scan_for("LeftAnti", null_aware=True).dynamic_bounds == {}
    assert scan_for("Inner", null_equality="NullEqualsNull").dynamic_bounds == {}
    assert scan_for("Inner", key_type=pa.float64()).dynamic_bounds == {}          # bounds of integer keys only
    assert scan_for("Inner", projection=("w",)).dynamic_bounds == {}              # the scan does not read the key column
    assert scan_for("Inner", projection=None).dynamic_bounds == {"pk": (100, 200)}
    monkeypatch.setattr(ops, "column_minmax", lambda t, c: (None, None, 0, False))
    assert scan_for("Inner").dynamic_bounds == {"pk": (1, 0)}                     # empty build side: the empty range
    # two key columns: no single-range filter
    scan = P.ParquetExec("/nonexistent.parquet", ["pk", "pk2"], "probe")
    j = P.HashJoinExec(P.MemoryExec(StubTable(10)), scan, [("bk", "pk"), ("v", "pk2")], "Inner")
    j._publish_dynamic_bounds(Build())
    assert scan.dynamic_bounds == {}
