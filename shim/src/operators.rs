//! The single-input GPU nodes: FilterExec (filter.rs:85), ProjectionExec (projection.rs:721-746), AggregateExec
//! (aggregates/mod.rs:839), SortExec / TopK (sorts/sort.rs:1366) and RepartitionExec with Partitioning::Hash
//! (repartition/mod.rs:1626) over `libdfgpu.so`.  All of them share one shape — `GpuUnaryExec`: the child's partition is uploaded
//! batch by batch and concatenated (a launch wants >= 10^6 rows: an 8192-row batch would be launch-bound), ONE call sequence of
//! the C ABI runs on a blocking thread (HIP waits never block the executor, execution_plan.rs:549-565), the result leaves
//! `batch_size` rows at a time (LimitedBatchCoalescer, coalesce/mod.rs:27-120).  Dropping the stream frees the device tables
//! (drop = cancel, execution_plan.rs:539-547).  Python twins, driven by the parity tests through the same entry points:
//! datafusion_amd/physical_plan.py {FilterExec, ProjectionExec, AggregateExec, GpuFusedAggregateExec, SortExec, RepartitionExec}.
use crate::expr::{field_of, lower, Lowered};
use crate::table::DeviceTable;
use crate::{check, sys};
use arrow::datatypes::SchemaRef;
use datafusion::error::{DataFusionError, Result};
use datafusion::execution::{SendableRecordBatchStream, TaskContext};
use datafusion::physical_expr::expressions::Column;
use datafusion::physical_plan::aggregates::{AggregateExec, AggregateMode};
use datafusion::physical_plan::filter::FilterExec;
use datafusion::physical_plan::projection::ProjectionExec;
use datafusion::physical_plan::repartition::RepartitionExec;
use datafusion::physical_plan::sorts::sort::SortExec;
use datafusion::physical_plan::stream::RecordBatchStreamAdapter;
use datafusion::physical_plan::{DisplayAs, DisplayFormatType, ExecutionPlan, Partitioning, PlanProperties};
use futures::{StreamExt, TryStreamExt};
use std::ffi::CString;
use std::sync::Arc;

/// what one node does with its input table
#[derive(Debug, Clone)]
pub enum GpuOp {
    /// dfgpu_filter: predicate, then the projected columns compacted in input order (NULL predicate rows dropped, filter.rs:1396-1419)
    Filter { predicate: Arc<Lowered>, projection: Vec<i32> },
    /// dfgpu_project: one expression per output column
    Project { exprs: Vec<Arc<Lowered>>, names: Vec<CString> },
    /// dfgpu_agg_create / update(_filtered) / emit.  `predicate` = a FilterExec the rule fused below the aggregate: filter,
    /// projection and accumulation then run as ONE pass over the input columns (aggregate.hip's fused node)
    Aggregate { mode: i32, group_by: Vec<Arc<Lowered>>, group_names: Vec<CString>, aggs: Vec<AggSpec>, predicate: Option<Arc<Lowered>> },
    /// dfgpu_sort; fetch >= 0 = TopK
    Sort { keys: Vec<i32>, descending: Vec<u8>, nulls_first: Vec<u8>, fetch: i64 },
    /// dfgpu_partition: hash(keys; seed 0) % n — output partition p of this node is slice p of every input partition
    HashRepartition { keys: Vec<i32>, n: i32 },
}

#[derive(Debug, Clone)]
pub struct AggSpec {
    pub func: i32, // dfgpu_agg_func: SUM, AVG, COUNT, MIN, MAX
    pub arg: Option<Arc<Lowered>>,
    pub name: CString,
    pub return_field: sys::dfgpu_field,
}

#[derive(Debug)]
pub struct GpuUnaryExec {
    name: &'static str,
    input: Arc<dyn ExecutionPlan>,
    op: GpuOp,
    cache: Arc<PlanProperties>, // copied verbatim from the CPU operator this node replaces: same schema, partitioning, ordering
}

fn lowered(e: &Arc<dyn datafusion::physical_expr::PhysicalExpr>, schema: &arrow::datatypes::Schema) -> Option<Arc<Lowered>> {
    let mut l = Lowered::default();
    lower(e, schema, &mut l)?;
    Some(Arc::new(l))
}
fn cname(s: &str) -> CString {
    CString::new(s.replace('\0', " ")).expect("no interior NUL")
}
fn types_ok(schema: &SchemaRef) -> bool {
    schema.fields().iter().all(|f| field_of(f.data_type()).is_some())
}

impl GpuUnaryExec {
    /// FilterExec -> dfgpu_filter.  None = some expression or type has no device form: the CPU operator stays.
    pub fn try_from_filter(f: &FilterExec) -> Option<Self> {
        let in_schema = f.input().schema();
        if !types_ok(&in_schema) {
            return None;
        }
        let projection = f.projection().cloned().unwrap_or_else(|| (0..in_schema.fields().len()).collect());
        Some(Self { name: "GpuFilterExec", input: Arc::clone(f.input()),
                    op: GpuOp::Filter { predicate: lowered(f.predicate(), &in_schema)?, projection: projection.iter().map(|c| *c as i32).collect() },
                    cache: Arc::clone(f.properties()) })
    }

    pub fn try_from_projection(p: &ProjectionExec) -> Option<Self> {
        let in_schema = p.input().schema();
        if !types_ok(&in_schema) || !types_ok(&p.schema()) {
            return None;
        }
        let mut exprs = vec![];
        let mut names = vec![];
        for pe in p.expr() {
            exprs.push(lowered(&pe.expr, &in_schema)?);
            names.push(cname(&pe.alias));
        }
        Some(Self { name: "GpuProjectionExec", input: Arc::clone(p.input()), op: GpuOp::Project { exprs, names }, cache: Arc::clone(p.properties()) })
    }

    /// AggregateExec with SUM / AVG / COUNT / MIN / MAX, plain (single) grouping sets and no per-aggregate FILTER / DISTINCT /
    /// ORDER BY (the rule leaves everything else on the CPU).  `below` = a GpuFilterExec directly under it, which is absorbed.
    pub fn try_from_aggregate(a: &AggregateExec, below: Option<&GpuUnaryExec>) -> Option<Self> {
        let (input, predicate) = match below.map(|b| (&b.op, &b.input)) {
            // only an unprojected filter keeps the column numbering of the aggregate's expressions
            Some((GpuOp::Filter { predicate, projection }, inp)) if projection.iter().enumerate().all(|(i, c)| i as i32 == *c) && projection.len() == inp.schema().fields().len() =>
                (Arc::clone(inp), Some(Arc::clone(predicate))),
            _ => (Arc::clone(a.input()), None),
        };
        let in_schema = input.schema();
        if !types_ok(&in_schema) || !types_ok(&a.schema()) || !a.group_expr().is_single() || a.filter_expr().iter().any(|f| f.is_some()) {
            return None;
        }
        let mode = match a.mode() {
            AggregateMode::Partial => sys::DFGPU_AGG_PARTIAL,
            AggregateMode::Final => sys::DFGPU_AGG_FINAL,
            AggregateMode::FinalPartitioned => sys::DFGPU_AGG_FINAL_PARTITIONED,
            AggregateMode::Single => sys::DFGPU_AGG_SINGLE,
            AggregateMode::SinglePartitioned => sys::DFGPU_AGG_SINGLE_PARTITIONED,
            _ => return None, // PartialReduce: state in, state out — not offered by the library
        };
        let mut group_by = vec![];
        let mut group_names = vec![];
        for (e, name) in a.group_expr().expr() {
            group_by.push(lowered(e, &in_schema)?);
            group_names.push(cname(name));
        }
        let mut aggs = vec![];
        for f in a.aggr_expr() {
            let func = match f.fun().name().to_ascii_lowercase().as_str() {
                "sum" => sys::DFGPU_AGG_SUM,
                "avg" => sys::DFGPU_AGG_AVG,
                "count" => sys::DFGPU_AGG_COUNT,
                "min" => sys::DFGPU_AGG_MIN,
                "max" => sys::DFGPU_AGG_MAX,
                _ => return None,
            };
            if f.is_distinct() || !f.order_bys().is_empty() || f.expressions().len() > 1 {
                return None;
            }
            // COUNT(*) arrives as count(Int64(1)): a literal argument counts rows
            let arg = f.expressions().first().filter(|e| e.as_any().downcast_ref::<datafusion::physical_expr::expressions::Literal>().is_none());
            let arg = match arg {
                Some(e) => Some(lowered(e, &in_schema)?),
                None => None,
            };
            // Final modes cannot derive AVG(Decimal128)'s declared type from its state: AggregateFunctionExpr::return_field carries it
            aggs.push(AggSpec { func, arg, name: cname(f.name()), return_field: field_of(f.field().data_type())? });
        }
        Some(Self { name: "GpuAggregateExec", input, op: GpuOp::Aggregate { mode, group_by, group_names, aggs, predicate }, cache: Arc::clone(a.properties()) })
    }

    /// SortExec over column keys (expressions are projected below it by the planner); preserve_partitioning = true sorts every
    /// partition by itself, which is what one call per partition does
    pub fn try_from_sort(s: &SortExec) -> Option<Self> {
        if !types_ok(&s.input().schema()) {
            return None;
        }
        let (mut keys, mut descending, mut nulls_first) = (vec![], vec![], vec![]);
        for e in s.expr().iter() {
            keys.push(e.expr.as_any().downcast_ref::<Column>()?.index() as i32);
            descending.push(e.options.descending as u8);
            nulls_first.push(e.options.nulls_first as u8);
        }
        Some(Self { name: "GpuSortExec", input: Arc::clone(s.input()), op: GpuOp::Sort { keys, descending, nulls_first, fetch: s.fetch().map_or(-1, |f| f as i64) },
                    cache: Arc::clone(s.properties()) })
    }

    /// RepartitionExec(Hash(columns, n)) inside ONE process (several processes = dfgpu_exchange_hash over RCCL, driven by the
    /// distributed runtime that owns the ranks).  RoundRobin / UnknownPartitioning stay on the CPU: they only regroup batches.
    pub fn try_from_repartition(r: &RepartitionExec) -> Option<Self> {
        let Partitioning::Hash(exprs, n) = r.partitioning() else { return None };
        if !types_ok(&r.input().schema()) || *n > 64 {
            return None;
        }
        let keys = exprs.iter().map(|e| e.as_any().downcast_ref::<Column>().map(|c| c.index() as i32)).collect::<Option<Vec<_>>>()?;
        Some(Self { name: "GpuRepartitionExec", input: Arc::clone(r.input()), op: GpuOp::HashRepartition { keys, n: *n as i32 }, cache: Arc::clone(r.properties()) })
    }

    pub fn op(&self) -> &GpuOp {
        &self.op
    }
}

/// the call sequence of one node over one device table
fn run(op: &GpuOp, input: &DeviceTable, out_partition: usize) -> Result<DeviceTable> {
    let mut out = std::ptr::null_mut();
    match op {
        GpuOp::Filter { predicate, projection } => {
            let p = predicate.as_c();
            check(unsafe { sys::dfgpu_filter(input.0, &p, projection.as_ptr(), projection.len() as i32, &mut out) })?;
        }
        GpuOp::Project { exprs, names } => {
            let e: Vec<_> = exprs.iter().map(|l| l.as_c()).collect();
            let n: Vec<_> = names.iter().map(|s| s.as_ptr()).collect();
            check(unsafe { sys::dfgpu_project(input.0, e.as_ptr(), n.as_ptr(), e.len() as i32, &mut out) })?;
        }
        GpuOp::Aggregate { mode, group_by, group_names, aggs, predicate } => {
            let g: Vec<_> = group_by.iter().map(|l| l.as_c()).collect();
            let gn: Vec<_> = group_names.iter().map(|s| s.as_ptr()).collect();
            let empty = Lowered::default();
            let specs: Vec<_> = aggs.iter().map(|a| sys::dfgpu_agg_spec { func: a.func, has_arg: a.arg.is_some() as i32, arg: a.arg.as_deref().unwrap_or(&empty).as_c(),
                                                                         name: a.name.as_ptr(), return_field: a.return_field }).collect();
            let mut h = std::ptr::null_mut();
            check(unsafe { sys::dfgpu_agg_create(*mode, g.as_ptr(), gn.as_ptr(), g.len() as i32, specs.as_ptr(), specs.len() as i32, &mut h) })?;
            let rc = match predicate {
                Some(p) => unsafe { sys::dfgpu_agg_update_filtered(h, input.0, &p.as_c()) },
                None => unsafe { sys::dfgpu_agg_update(h, input.0) },
            };
            let rc = if rc == 0 { unsafe { sys::dfgpu_agg_emit(h, &mut out) } } else { rc };
            unsafe { sys::dfgpu_agg_free(h) };
            check(rc)?;
        }
        GpuOp::Sort { keys, descending, nulls_first, fetch } => {
            check(unsafe { sys::dfgpu_sort(input.0, keys.as_ptr(), descending.as_ptr(), nulls_first.as_ptr(), keys.len() as i32, *fetch, &mut out) })?;
        }
        GpuOp::HashRepartition { keys, n } => {
            let mut parts = vec![std::ptr::null_mut(); *n as usize];
            check(unsafe { sys::dfgpu_partition(input.0, keys.as_ptr(), keys.len() as i32, *n, parts.as_mut_ptr()) })?;
            let mut kept = None;
            for (p, h) in parts.into_iter().enumerate() {
                let t = DeviceTable(h); // every slice is owned: the ones this output partition does not take are freed here
                if p == out_partition {
                    kept = Some(t);
                }
            }
            return kept.ok_or_else(|| DataFusionError::Internal("output partition out of range".into()));
        }
    }
    Ok(DeviceTable(out))
}

impl DisplayAs for GpuUnaryExec {
    fn fmt_as(&self, _t: DisplayFormatType, f: &mut std::fmt::Formatter) -> std::fmt::Result {
        match &self.op {
            GpuOp::Aggregate { predicate: Some(_), .. } => write!(f, "{}: FilterExec fused", self.name),
            GpuOp::Sort { fetch, .. } if *fetch >= 0 => write!(f, "{}: TopK(fetch={})", self.name, fetch),
            GpuOp::HashRepartition { keys, n } => write!(f, "{}: Hash({:?}, {})", self.name, keys, n),
            _ => write!(f, "{}", self.name),
        }
    }
}

impl ExecutionPlan for GpuUnaryExec {
    fn name(&self) -> &str { self.name }
    fn as_any(&self) -> &dyn std::any::Any { self }
    fn properties(&self) -> &Arc<PlanProperties> { &self.cache }
    fn children(&self) -> Vec<&Arc<dyn ExecutionPlan>> { vec![&self.input] }
    fn with_new_children(self: Arc<Self>, c: Vec<Arc<dyn ExecutionPlan>>) -> Result<Arc<dyn ExecutionPlan>> {
        Ok(Arc::new(Self { name: self.name, input: Arc::clone(&c[0]), op: self.op.clone(), cache: Arc::clone(&self.cache) }))
    }

    fn execute(&self, partition: usize, ctx: Arc<TaskContext>) -> Result<SendableRecordBatchStream> {
        // which input partitions feed this output partition: all of them for a repartition (and for Final over a coalesced
        // input the planner already put a CoalescePartitionsExec below), the same-numbered one otherwise
        let inputs: Vec<usize> = match &self.op {
            GpuOp::HashRepartition { .. } => (0..self.input.output_partitioning().partition_count()).collect(),
            _ => vec![partition],
        };
        let mut streams = inputs.iter().map(|p| self.input.execute(*p, Arc::clone(&ctx))).collect::<Result<Vec<_>>>()?; // lazy (execution_plan.rs:514-516)
        let op = self.op.clone();
        let schema: SchemaRef = self.schema();
        let out_schema = Arc::clone(&schema);
        let batch_size = ctx.session_config().batch_size() as i64;
        let fut = async move {
            let mut parts = vec![];
            for s in streams.iter_mut() {
                while let Some(batch) = s.next().await {
                    parts.push(DeviceTable::from_batch(&batch?)?);
                }
            }
            let input = DeviceTable::concat(&parts)?;
            drop(parts);
            let out = tokio::task::spawn_blocking(move || run(&op, &input, partition)).await.map_err(|e| DataFusionError::External(Box::new(e)))??;
            let n = out.num_rows()?;
            let batches: Vec<_> = (0..n).step_by(batch_size as usize).map(|off| out.export_batch(off, batch_size.min(n - off), &out_schema)).collect();
            Ok::<_, DataFusionError>(futures::stream::iter(batches))
        };
        Ok(Box::pin(RecordBatchStreamAdapter::new(schema, futures::stream::once(fut).try_flatten())))
    }
}
