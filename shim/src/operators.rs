//! The single-input GPU nodes: FilterExec (filter.rs:85), ProjectionExec (projection.rs:721-746), AggregateExec
//! (aggregates/mod.rs:839), SortExec / TopK (sorts/sort.rs:1366) and RepartitionExec with Partitioning::Hash
//! (repartition/mod.rs:1626) over `libdfgpu.so`.  All of them share one shape — `GpuUnaryExec`: the child's partition arrives as
//! ONE device table (device.rs: handed over by a GPU child, or uploaded batch by batch from a CPU child and concatenated — a launch
//! wants >= 10^6 rows, an 8192-row batch would be launch-bound), ONE call sequence of the C ABI runs on a blocking thread (HIP waits
//! never block the executor, execution_plan.rs:549-565), the result stays on the device for a GPU parent or leaves `batch_size`
//! rows at a time (LimitedBatchCoalescer, coalesce/mod.rs:27-120).  A hash repartition executes every input partition exactly
//! once and shares the slices between its output partitions (`RepartitionState`).  Dropping the stream frees the device tables
//! (drop = cancel, execution_plan.rs:539-547).  Python twins, driven by the parity tests through the same entry points:
//! datafusion_amd/physical_plan.py {FilterExec, ProjectionExec, AggregateExec, GpuFusedAggregateExec, SortExec, RepartitionExec}.
use crate::device::{device_input, host_stream, DeviceFuture, GpuNode};
use crate::expr::{field_of, lower, Lowered};
use crate::table::DeviceTable;
use crate::{blocking, check, sys};
use arrow::datatypes::{DataType, SchemaRef};
use datafusion::common::tree_node::TreeNodeRecursion;
use datafusion::error::{DataFusionError, Result};
use datafusion::execution::{SendableRecordBatchStream, TaskContext};
use datafusion::physical_expr::expressions::{Column, Literal};
use datafusion::physical_expr::PhysicalExpr;
use datafusion::physical_plan::aggregates::{AggregateExec, AggregateMode};
use datafusion::physical_plan::filter::FilterExec;
use datafusion::physical_plan::projection::ProjectionExec;
use datafusion::physical_plan::repartition::RepartitionExec;
use datafusion::physical_plan::sorts::sort::SortExec;
use datafusion::physical_plan::{ChildrenPropertiesMode, DisplayAs, DisplayFormatType, ExecutionPlan, Partitioning, PlanProperties, ReplaceChildrenOptions};
use futures::FutureExt;
use std::ffi::CString;
use std::sync::{Arc, Mutex};

/// what one node does with its input table
#[derive(Debug, Clone)]
pub enum GpuOp {
    /// dfgpu_filter: predicate, then the projected columns compacted in input order (NULL predicate rows dropped, filter.rs:1396-1419)
    Filter { predicate: Arc<Lowered>, projection: Vec<i32> },
    /// dfgpu_project: one expression per output column
    Project { exprs: Vec<Arc<Lowered>>, names: Vec<CString> },
    /// dfgpu_agg_create / update(_filtered) / emit.  `predicate` = a FilterExec the rule fused below the aggregate: filter,
    /// projection and accumulation then run as ONE pass over the input columns (aggregate.hip's fused node)
    /// `grouping_sets` = PhysicalGroupBy with several groups (GROUPING SETS / CUBE / ROLLUP): the typed NULL literal per key and, per set,
    /// which keys are NULLed out (dfgpu_agg_create_grouping_sets; raw modes only — a Final node groups by the keys + __grouping_id)
    Aggregate { mode: i32, group_by: Vec<Arc<Lowered>>, group_names: Vec<CString>, aggs: Vec<AggSpec>, predicate: Option<Arc<Lowered>>,
                grouping_sets: Option<(Vec<Arc<Lowered>>, Vec<u8>, i32)> },
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
    /// HashRepartition only: the slices of ALL input partitions, computed once and shared by the output partitions
    /// (RepartitionExec's shared state: every input partition is executed exactly once, repartition/mod.rs:154-360)
    exchange: Arc<RepartitionState>,
    cache: Arc<PlanProperties>, // copied verbatim from the CPU operator this node replaces: same schema, partitioning, ordering
}

/// output partition p of a hash repartition = slice p of every input partition, concatenated in input-partition order
#[derive(Default)]
pub struct RepartitionState {
    outputs: tokio::sync::OnceCell<Vec<Mutex<Option<DeviceTable>>>>,
}
impl std::fmt::Debug for RepartitionState {
    fn fmt(&self, f: &mut std::fmt::Formatter) -> std::fmt::Result {
        write!(f, "RepartitionState(computed: {})", self.outputs.initialized())
    }
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
        let projection: Vec<usize> = match f.projection() { Some(p) => p.iter().copied().collect(), None => (0..in_schema.fields().len()).collect() };
        Some(Self { name: "GpuFilterExec", input: Arc::clone(f.input()),
                    op: GpuOp::Filter { predicate: lowered(f.predicate(), &in_schema)?, projection: projection.iter().map(|c| *c as i32).collect() },
                    exchange: Default::default(), cache: Arc::clone(f.properties()) })
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
        Some(Self { name: "GpuProjectionExec", input: Arc::clone(p.input()), op: GpuOp::Project { exprs, names }, exchange: Default::default(), cache: Arc::clone(p.properties()) })
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
        if !types_ok(&in_schema) || !types_ok(&a.schema()) || a.filter_expr().iter().any(|f| f.is_some()) {
            return None;
        }
        let mode = match a.mode() {
            AggregateMode::Partial => sys::DFGPU_AGG_PARTIAL,
            AggregateMode::Final => sys::DFGPU_AGG_FINAL,
            AggregateMode::FinalPartitioned => sys::DFGPU_AGG_FINAL_PARTITIONED,
            AggregateMode::Single => sys::DFGPU_AGG_SINGLE,
            AggregateMode::SinglePartitioned => sys::DFGPU_AGG_SINGLE_PARTITIONED,
            AggregateMode::PartialReduce => sys::DFGPU_AGG_PARTIAL_REDUCE, // states in, states out
        };
        // GROUPING SETS: the typed NULL per key + the sets' masks; state-reading modes see __grouping_id as one more key column
        let raw = matches!(a.mode(), AggregateMode::Partial | AggregateMode::Single | AggregateMode::SinglePartitioned);
        let grouping_sets = if a.group_expr().is_single() {
            None
        } else if raw {
            let nulls = a.group_expr().null_expr().iter().map(|(e, _)| lowered(e, &in_schema)).collect::<Option<Vec<_>>>()?;
            let n = a.group_expr().expr().len();
            let masks: Vec<u8> = a.group_expr().groups().iter().flat_map(|g| g.iter().map(|b| *b as u8)).collect();
            // (9..16 grouping columns: __grouping_id is UInt16 in the reference — no device type — so the CPU operator stays)
            Some((nulls, masks, (a.group_expr().groups().len()) as i32)).filter(|_| n >= 1 && n <= 63 && !(9..=16).contains(&n))
        } else {
            return None; // a Final over grouping sets reads (keys, __grouping_id) positionally: planned as single-set by as_final()
        };
        if !a.group_expr().is_single() && grouping_sets.is_none() {
            return None;
        }
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
            // COUNT(*) arrives as count(Int64(1)): a (non-NULL) literal argument counts rows
            let exprs = f.expressions();
            let arg = match exprs.first() {
                Some(e) if e.downcast_ref::<Literal>().is_some_and(|l| !l.value().is_null()) && func == sys::DFGPU_AGG_COUNT => None,
                Some(e) => Some(lowered(e, &in_schema)?),
                None => None,
            };
            // what the device cannot accumulate stays the CPU operator (physical_plan.py unsupported_reason is the tested twin):
            // AVG over Decimal128(p > 25) sums in Decimal256 (avg_sum_data_type, average.rs:131-172); MIN / MAX over
            // Decimal128(p > 18) compares 128-bit values, the device 64-bit words (aggregate.hip wide_minmax_values_fit)
            if raw {
                if let Some(DataType::Decimal128(p, _)) = exprs.first().and_then(|e| e.data_type(&in_schema).ok()) {
                    if (func == sys::DFGPU_AGG_AVG && p + 13 > 38) || ((func == sys::DFGPU_AGG_MIN || func == sys::DFGPU_AGG_MAX) && p > 18) {
                        return None;
                    }
                }
            }
            // Final modes cannot derive AVG(Decimal128)'s declared type from its state: AggregateFunctionExpr::return_field carries it
            aggs.push(AggSpec { func, arg, name: cname(f.name()), return_field: field_of(f.field().data_type())? });
        }
        Some(Self { name: "GpuAggregateExec", input, op: GpuOp::Aggregate { mode, group_by, group_names, aggs, predicate, grouping_sets }, exchange: Default::default(),
                    cache: Arc::clone(a.properties()) })
    }

    /// SortExec over column keys (expressions are projected below it by the planner); preserve_partitioning = true sorts every
    /// partition by itself, which is what one call per partition does
    pub fn try_from_sort(s: &SortExec) -> Option<Self> {
        if !types_ok(&s.input().schema()) {
            return None;
        }
        // sort.hip packs the key columns into at most 192 bits (value ranges are only known at run time: the types bound them here)
        let key_bits: usize = s.expr().iter().filter_map(|e| e.expr.data_type(&s.input().schema()).ok())
            .map(|t| match t { DataType::Boolean => 2, DataType::Utf8 | DataType::LargeUtf8 | DataType::Dictionary(..) => 33, t => t.primitive_width().unwrap_or(16) * 8 + 1 }).sum();
        if key_bits > 192 {
            return None;
        }
        let (mut keys, mut descending, mut nulls_first) = (vec![], vec![], vec![]);
        for e in s.expr().iter() {
            keys.push(e.expr.downcast_ref::<Column>()?.index() as i32);
            descending.push(e.options.descending as u8);
            nulls_first.push(e.options.nulls_first as u8);
        }
        Some(Self { name: "GpuSortExec", input: Arc::clone(s.input()), op: GpuOp::Sort { keys, descending, nulls_first, fetch: s.fetch().map_or(-1, |f| f as i64) },
                    exchange: Default::default(), cache: Arc::clone(s.properties()) })
    }

    /// RepartitionExec(Hash(columns, n)) inside ONE process (several processes = dfgpu_exchange_hash over RCCL, driven by the
    /// distributed runtime that owns the ranks).  RoundRobin / UnknownPartitioning stay on the CPU: they only regroup batches.
    pub fn try_from_repartition(r: &RepartitionExec) -> Option<Self> {
        let Partitioning::Hash(exprs, n) = r.partitioning() else { return None };
        if !types_ok(&r.input().schema()) || *n > 64 || r.preserve_order() {
            return None; // a sort-preserving repartition merges sorted streams: it stays the CPU operator
        }
        let keys = exprs.iter().map(|e| e.downcast_ref::<Column>().map(|c| c.index() as i32)).collect::<Option<Vec<_>>>()?;
        Some(Self { name: "GpuRepartitionExec", input: Arc::clone(r.input()), op: GpuOp::HashRepartition { keys, n: *n as i32 }, exchange: Default::default(), cache: Arc::clone(r.properties()) })
    }

    pub fn op(&self) -> &GpuOp {
        &self.op
    }
}

/// the call sequence of one node over one device table
fn run(op: &GpuOp, input: &DeviceTable) -> Result<DeviceTable> {
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
        GpuOp::Aggregate { mode, group_by, group_names, aggs, predicate, grouping_sets } => {
            let g: Vec<_> = group_by.iter().map(|l| l.as_c()).collect();
            let gn: Vec<_> = group_names.iter().map(|s| s.as_ptr()).collect();
            let empty = Lowered::default();
            let specs: Vec<_> = aggs.iter().map(|a| sys::dfgpu_agg_spec { func: a.func, has_arg: a.arg.is_some() as i32, arg: a.arg.as_deref().unwrap_or(&empty).as_c(),
                                                                         name: a.name.as_ptr(), return_field: a.return_field }).collect();
            let mut h = std::ptr::null_mut();
            match grouping_sets {
                None => check(unsafe { sys::dfgpu_agg_create(*mode, g.as_ptr(), gn.as_ptr(), g.len() as i32, specs.as_ptr(), specs.len() as i32, &mut h) })?,
                Some((nulls, masks, n_sets)) => {
                    let nb: Vec<_> = nulls.iter().map(|l| l.as_c()).collect();
                    check(unsafe { sys::dfgpu_agg_create_grouping_sets(*mode, g.as_ptr(), nb.as_ptr(), gn.as_ptr(), g.len() as i32, masks.as_ptr(), *n_sets, specs.as_ptr(),
                                                                       specs.len() as i32, &mut h) })?
                }
            }
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
        GpuOp::HashRepartition { .. } => {
            return Err(DataFusionError::Internal("a hash repartition runs through RepartitionState, not through run()".into()));
        }
    }
    Ok(DeviceTable(out))
}

/// dfgpu_partition of one input partition: `n` owned slices (partition = hash(keys; seed 0) % n, row order kept inside each)
fn partition_slices(input: &DeviceTable, keys: &[i32], n: i32) -> Result<Vec<DeviceTable>> {
    let mut parts = vec![std::ptr::null_mut(); n as usize];
    check(unsafe { sys::dfgpu_partition(input.0, keys.as_ptr(), keys.len() as i32, n, parts.as_mut_ptr()) })?;
    Ok(parts.into_iter().map(DeviceTable).collect())
}

impl DisplayAs for GpuUnaryExec {
    fn fmt_as(&self, _t: DisplayFormatType, f: &mut std::fmt::Formatter) -> std::fmt::Result {
        match &self.op {
            GpuOp::Aggregate { predicate: Some(_), .. } => write!(f, "{}: FilterExec fused", self.name),
            GpuOp::Aggregate { grouping_sets: Some((_, _, n)), .. } => write!(f, "{}: {} grouping sets", self.name, n),
            GpuOp::Sort { fetch, .. } if *fetch >= 0 => write!(f, "{}: TopK(fetch={})", self.name, fetch),
            GpuOp::HashRepartition { keys, n } => write!(f, "{}: Hash({:?}, {})", self.name, keys, n),
            _ => write!(f, "{}", self.name),
        }
    }
}

impl GpuNode for GpuUnaryExec {
    fn execute_device(&self, partition: usize, ctx: Arc<TaskContext>) -> Result<DeviceFuture> {
        if let GpuOp::HashRepartition { keys, n } = &self.op {
            // every input partition is executed ONCE, by whichever output partition is polled first; the others wait on the cell
            let n_inputs = self.input.output_partitioning().partition_count();
            let inputs = (0..n_inputs).map(|p| device_input(&self.input, p, Arc::clone(&ctx))).collect::<Result<Vec<_>>>()?;
            let (state, keys, n) = (Arc::clone(&self.exchange), keys.clone(), *n);
            return Ok(async move {
                let outputs = state.outputs.get_or_try_init(|| async move {
                    let tables = futures::future::try_join_all(inputs).await?; // the input partitions run concurrently, each on its own stream
                    blocking(move || {
                        let mut per_output: Vec<Vec<DeviceTable>> = (0..n).map(|_| vec![]).collect();
                        for t in &tables {
                            for (p, slice) in partition_slices(t, &keys, n)?.into_iter().enumerate() {
                                per_output[p].push(slice);
                            }
                        }
                        per_output.into_iter().map(|slices| Ok(Mutex::new(Some(if slices.len() == 1 { slices.into_iter().next().unwrap() } else { DeviceTable::concat(&slices)? }))))
                            .collect::<Result<Vec<_>>>()
                    }).await
                }).await?;
                outputs.get(partition).and_then(|m| m.lock().unwrap().take())
                    .ok_or_else(|| DataFusionError::Internal(format!("GpuRepartitionExec: output partition {partition} out of range or executed twice")))
            }
            .boxed());
        }
        let input = device_input(&self.input, partition, ctx)?; // a GPU child hands its device table over: no PCIe in between
        let op = self.op.clone();
        Ok(async move {
            let input = input.await?;
            blocking(move || run(&op, &input)).await
        }
        .boxed())
    }
}

impl ExecutionPlan for GpuUnaryExec {
    fn name(&self) -> &str { self.name }
    fn properties(&self) -> &Arc<PlanProperties> { &self.cache }
    fn children(&self) -> Vec<&Arc<dyn ExecutionPlan>> { vec![&self.input] }
    fn apply_expressions(&self, _f: &mut dyn FnMut(&Arc<dyn PhysicalExpr>) -> Result<TreeNodeRecursion>) -> Result<TreeNodeRecursion> {
        Ok(TreeNodeRecursion::Continue) // every expression was lowered to the flat IR at plan time: no PhysicalExpr is evaluated here
    }
    fn maintains_input_order(&self) -> Vec<bool> {
        vec![matches!(self.op, GpuOp::Filter { .. } | GpuOp::Project { .. })] // FilterExec / ProjectionExec keep their input's order
    }
    fn replace_children(self: Arc<Self>, c: Vec<Arc<dyn ExecutionPlan>>, _options: ReplaceChildrenOptions) -> Result<Arc<dyn ExecutionPlan>> {
        if c.len() != 1 {
            return Err(DataFusionError::Internal(format!("{} takes one child", self.name)));
        }
        Ok(Arc::new(Self { name: self.name, input: Arc::clone(&c[0]), op: self.op.clone(), exchange: Default::default(), cache: Arc::clone(&self.cache) }))
    }
    #[allow(deprecated)]
    fn with_new_children(self: Arc<Self>, c: Vec<Arc<dyn ExecutionPlan>>) -> Result<Arc<dyn ExecutionPlan>> {
        self.replace_children(c, ReplaceChildrenOptions::new(ChildrenPropertiesMode::Recompute))
    }
    fn reset_state(self: Arc<Self>) -> Result<Arc<dyn ExecutionPlan>> {
        let children = vec![Arc::clone(&self.input)];
        self.replace_children(children, ReplaceChildrenOptions::new(ChildrenPropertiesMode::Keep))
    }
    fn execute(&self, partition: usize, ctx: Arc<TaskContext>) -> Result<SendableRecordBatchStream> {
        let batch_size = ctx.session_config().batch_size();
        let schema: SchemaRef = self.schema();
        Ok(host_stream(schema, self.execute_device(partition, ctx)?, batch_size))
    }
}
