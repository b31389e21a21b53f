//! `GpuHashJoinExec`: HashJoinExec (physical-plan/src/joins/hash_join/exec.rs:752) on the device.  The state machine of
//! HashJoinStream (hash_join/stream.rs:127-140, 591-640) maps onto the C ABI one to one — the exact sequence tests/c/plan_driver.c
//! executes from plain C against the reference's snapshot tests:
//!   CollectBuildSide   ONCE per join table (`SharedBuild`, the twin of `OnceAsync<JoinLeftData>`, exec.rs:772,1503-1523): every build
//!                      batch -> dfgpu_join_builder_push (grows a reservation like try_grow, exec.rs:2608), end of the build child ->
//!                      dfgpu_join_builder_finish (= concat_batches + table build).  CollectLeft: one table for all probe
//!                      partitions; Partitioned: one per partition.
//!   FetchProbeBatch /
//!   ProcessProbeBatch  the probe child's partition as one device table (a GPU child hands it over, a CPU child's batches are
//!                      uploaded and concatenated), ONE dfgpu_join_probe / dfgpu_join_probe_with_filter per partition (a launch
//!                      wants >= 10^6 rows; 8192-row batches would be launch-bound); visited build rows are marked in the shared
//!                      table
//!   ExhaustedProbeSide the LAST probe partition to finish (`SharedBuild::remaining`, the reference's probe_threads_counter,
//!                      exec.rs:1312-1330) calls dfgpu_join_emit_unmatched ONCE for Left / Full / LeftSemi / LeftAnti / LeftMark
//!   output             dfgpu_table_export_batch, batch_size rows at a time (LimitedBatchCoalescer) — or the device table itself when
//!                      the parent is a GPU node (device.rs)
//! Python twin: datafusion_amd/physical_plan.py HashJoinExec / GpuHashJoinExec.
use crate::device::{device_input, host_stream, DeviceFuture, GpuNode};
use crate::expr::{field_of, lower, Lowered};
use crate::table::DeviceTable;
use crate::{blocking, check, sys};
use arrow::datatypes::SchemaRef;
use datafusion::common::tree_node::TreeNodeRecursion;
use datafusion::common::{JoinSide, JoinType, NullEquality, ScalarValue};
use datafusion::error::{DataFusionError, Result};
use datafusion::execution::{SendableRecordBatchStream, TaskContext};
use datafusion::logical_expr::Operator;
use datafusion::physical_expr::expressions::{BinaryExpr, Column, DynamicFilterPhysicalExpr, Literal};
use datafusion::physical_expr::PhysicalExpr;
use datafusion::physical_plan::joins::utils::build_join_schema;
use datafusion::physical_plan::joins::{HashJoinExec, PartitionMode};
use datafusion::physical_plan::{ChildrenPropertiesMode, DisplayAs, DisplayFormatType, ExecutionPlan, PlanProperties, ReplaceChildrenOptions};
use futures::{FutureExt, StreamExt};
use std::ffi::CString;
use std::sync::atomic::{AtomicUsize, Ordering};
use std::sync::{Arc, Mutex};

/// the built side: dfgpu_join_t + the device table it was built over (JoinLeftData, exec.rs:195-240).  Immutable after the build
/// and probed by many partitions at once (include/dfgpu.h: "a join table may be probed by many callers at once"); the visited marks
/// are idempotent byte stores.
pub struct GpuJoinTable(sys::dfgpu_join_t);
unsafe impl Send for GpuJoinTable {}
unsafe impl Sync for GpuJoinTable {}
impl Drop for GpuJoinTable {
    fn drop(&mut self) {
        unsafe { sys::dfgpu_join_free(self.0) }; // drop = cancel, as for every DataFusion stream (execution_plan.rs:539-547)
    }
}

/// one join table and the probe partitions that share it
struct SharedBuild {
    table: tokio::sync::OnceCell<Arc<GpuJoinTable>>,
    /// probe partitions that have not finished yet; the one that brings this to 0 reports the unmatched build rows
    remaining: AtomicUsize,
}
impl SharedBuild {
    fn new(probe_partitions: usize) -> Arc<Self> {
        Arc::new(Self { table: tokio::sync::OnceCell::new(), remaining: AtomicUsize::new(probe_partitions) })
    }
}

/// JoinFilter (joins/join_filter.rs:27) lowered for dfgpu_join_probe_with_filter: the expression over the intermediate batch's
/// columns f0, f1, ... and, per column, which side's column it is
struct LoweredJoinFilter {
    expression: Lowered,
    column_index: Vec<i32>,
    column_side: Vec<i32>, // 0 = left (build side), 1 = right (probe side)
}

/// bounds of the build keys accumulated over the build partitions, published to the probe-side scan when all have reported
/// (SharedBuildAccumulator, hash_join/shared_bounds.rs:277-284)
struct BoundsAccumulator {
    filter: Arc<DynamicFilterPhysicalExpr>,
    key_type: arrow::datatypes::DataType,      // the probe-side key expression's type: the bounds are published as scalars of it
    state: Mutex<(usize, Option<(i64, i64)>)>, // (partitions still to report, bounds so far)
}

pub struct GpuHashJoinExec {
    left: Arc<dyn ExecutionPlan>,
    right: Arc<dyn ExecutionPlan>,
    on: Vec<(i32, i32)>,
    filter: Option<Arc<LoweredJoinFilter>>,
    join_type: JoinType,
    null_equality: NullEquality,
    null_aware: bool,
    mode: PartitionMode,
    /// what the library is asked to emit: these build columns, then these probe columns (then `mark` for the mark joins) ...
    build_out: Vec<i32>,
    probe_out: Vec<i32>,
    /// ... and where column k of the node's schema sits in that: the reference's projection may interleave the sides
    reorder: Vec<i32>,
    /// no ancestor observes HashJoinExec's probe-side order (exec.rs:3349): the single-pass unordered probe may be used
    order_insensitive: bool,
    bounds: Option<Arc<BoundsAccumulator>>,
    builds: Vec<Arc<SharedBuild>>, // CollectLeft: one; Partitioned: one per partition
    cache: Arc<PlanProperties>,    // copied verbatim from the HashJoinExec it replaces (exec.rs:1308-1354)
}

impl std::fmt::Debug for GpuHashJoinExec {
    fn fmt(&self, f: &mut std::fmt::Formatter) -> std::fmt::Result {
        write!(f, "GpuHashJoinExec {{ mode: {:?}, join_type: {:?}, on: {:?} }}", self.mode, self.join_type, self.on)
    }
}

fn builds_for(mode: PartitionMode, right: &Arc<dyn ExecutionPlan>) -> Vec<Arc<SharedBuild>> {
    let n = right.output_partitioning().partition_count();
    match mode {
        PartitionMode::CollectLeft => vec![SharedBuild::new(n)],
        _ => (0..n).map(|_| SharedBuild::new(1)).collect(),
    }
}

fn emits_build_side(jt: JoinType) -> bool {
    matches!(jt, JoinType::Left | JoinType::Full | JoinType::LeftSemi | JoinType::LeftAnti | JoinType::LeftMark)
}

impl GpuHashJoinExec {
    /// None = something has no device form (an expression key, an unsupported type, a fetch limit ...): the CPU operator stays
    pub fn try_from_cpu(j: &HashJoinExec, order_insensitive: bool) -> Option<Self> {
        if *j.partition_mode() == PartitionMode::Auto || j.fetch().is_some() {
            return None; // Auto is resolved by JoinSelection before user rules run; a limit inside the join stays on the CPU
        }
        let on = j.on().iter().map(|(l, r)| Some((l.downcast_ref::<Column>()?.index() as i32, r.downcast_ref::<Column>()?.index() as i32)))
            .collect::<Option<Vec<_>>>()?;
        let (left_schema, right_schema) = (j.left().schema(), j.right().schema());
        let filter = match j.filter() {
            None => None,
            Some(f) => {
                let mut expression = Lowered::default();
                lower(f.expression(), f.schema(), &mut expression)?;
                let mut lf = LoweredJoinFilter { expression, column_index: vec![], column_side: vec![] };
                for ci in f.column_indices() {
                    lf.column_index.push(ci.index as i32);
                    lf.column_side.push(match ci.side { JoinSide::Left => 0, JoinSide::Right => 1, JoinSide::None => return None });
                }
                Some(Arc::new(lf))
            }
        };
        // the join's full output schema and which side every column of it comes from (semi / anti joins: one side only; mark
        // joins: one side + the mark column), then the embedded projection on top of it (exec.rs:780)
        let (_, sides) = build_join_schema(&left_schema, &right_schema, j.join_type());
        let projected: Vec<usize> = match &j.projection {
            Some(p) => p.iter().copied().collect(),
            None => (0..sides.len()).collect(),
        };
        let (mut build_out, mut probe_out) = (vec![], vec![]);
        for k in &projected {
            match sides[*k].side {
                JoinSide::Left => build_out.push(sides[*k].index as i32),
                JoinSide::Right => probe_out.push(sides[*k].index as i32),
                JoinSide::None => {}
            }
        }
        let (mut nb, mut np) = (0i32, 0i32);
        let reorder = projected.iter().map(|k| match sides[*k].side {
            JoinSide::Left => { nb += 1; nb - 1 }
            JoinSide::Right => { np += 1; build_out.len() as i32 + np - 1 }
            JoinSide::None => (build_out.len() + probe_out.len()) as i32, // the library appends `mark` last
        }).collect();
        for f in j.schema().fields() {
            field_of(f.data_type())?;
        }
        // the dynamic filter this join feeds (exec.rs:869-875): its bounds are published from the device-side key statistics
        let bounds = j.dynamic_expressions_produced().into_iter().find_map(|e| {
            let e: Arc<dyn std::any::Any + Send + Sync> = e;
            e.downcast::<DynamicFilterPhysicalExpr>().ok()
        }).filter(|_| on.len() == 1 && !j.null_aware).map(|filter| {
            let parts = if *j.partition_mode() == PartitionMode::CollectLeft { 1 } else { j.right().output_partitioning().partition_count() };
            let key_type = j.right().schema().field(on[0].1 as usize).data_type().clone();
            Arc::new(BoundsAccumulator { filter, key_type, state: Mutex::new((parts, None)) })
        });
        Some(Self {
            left: Arc::clone(j.left()),
            right: Arc::clone(j.right()),
            on,
            filter,
            join_type: *j.join_type(),
            null_equality: j.null_equality(),
            null_aware: j.null_aware,
            mode: *j.partition_mode(),
            build_out,
            probe_out,
            reorder,
            order_insensitive,
            bounds,
            builds: builds_for(*j.partition_mode(), j.right()),
            cache: Arc::clone(j.properties()),
        })
    }

    fn options(&self) -> sys::dfgpu_join_options {
        sys::dfgpu_join_options {
            perfect_hash_join_small_build_threshold: 1024,
            perfect_hash_join_min_key_density: sys::DFGPU_DEFAULT_MIN_KEY_DENSITY,
            table_mode: 0,
            force_hash_collisions: 0,
            // 4 = "order not needed": the library takes the single-pass probe where it applies and the ordered path otherwise
            probe_mode: if self.order_insensitive && self.filter.is_none() { 4 } else { 0 },
            null_aware: self.null_aware as i32,
        }
    }
}

/// CollectBuildSide: the build child's partition streams into the builder batch by batch (a GPU child hands ONE table over)
async fn collect_build(left: Arc<dyn ExecutionPlan>, partition: usize, ctx: Arc<TaskContext>, on_l: Vec<i32>, null_equality: i32,
                       opts: sys::dfgpu_join_options, bounds: Option<Arc<BoundsAccumulator>>) -> Result<Arc<GpuJoinTable>> {
    struct Builder(sys::dfgpu_join_builder_t);
    unsafe impl Send for Builder {}
    impl Drop for Builder {
        fn drop(&mut self) {
            if !self.0.is_null() {
                unsafe { sys::dfgpu_join_builder_free(self.0) }; // abandoned (error / cancellation): releases the reservation
            }
        }
    }
    let mut b = Builder(std::ptr::null_mut());
    check(unsafe { sys::dfgpu_join_builder_create(on_l.as_ptr(), on_l.len() as i32, null_equality, &opts, &mut b.0) })?;
    // the dynamic filter's bounds half (PushdownStrategy::Map = bounds only for large build sides, shared_bounds.rs:277-284): the key's
    // min / max over EVERY pushed batch (dfgpu_column_minmax reduces one table and caches the answer on its column), folded as the
    // batches go by — a build partition fed several CPU batches reports like one fed a single device table
    let key = on_l[0];
    // dfgpu_column_minmax reduces integer-like columns only (Int8..Int64, UInt8..UInt32, Date32).  A key of any other type (Utf8,
    // dictionary, Float64, Decimal128, UInt64, Boolean) makes it return non-zero: that is "no bounds for this join" — the fold stops and
    // the filter's bounds half stays unpublished — never a failed query (the reference's bounds are best effort too, shared_bounds.rs:277)
    let mut want_bounds = bounds.is_some();
    let mut key_bounds: Option<(i64, i64)> = None;
    let mut fold = |t: &DeviceTable| -> Result<()> {
        if want_bounds {
            let (mut lo, mut hi, mut valid, mut asc) = (0i64, 0i64, 0i64, 0i32);
            if unsafe { sys::dfgpu_column_minmax(t.0, key, &mut lo, &mut hi, &mut valid, &mut asc) } != 0 {
                want_bounds = false;
                key_bounds = None;
                return Ok(());
            }
            if valid > 0 {
                key_bounds = Some(match key_bounds { Some((a, b)) => (a.min(lo), b.max(hi)), None => (lo, hi) });
            }
        }
        Ok(())
    };
    if crate::device::as_gpu_node(&left).is_some() {
        let t = device_input(&left, partition, ctx)?.await?;
        check(unsafe { sys::dfgpu_join_builder_push(b.0, t.0) })?;
        fold(&t)?;
    } else {
        let schema = left.schema();
        let mut stream = left.execute(partition, ctx)?;
        let mut pushed = 0usize;
        while let Some(batch) = stream.next().await {
            let t = DeviceTable::from_batch(&batch?)?;
            check(unsafe { sys::dfgpu_join_builder_push(b.0, t.0) })?; // "Resources exhausted" here = try_grow failing (exec.rs:2608)
            fold(&t)?;
            pushed += 1;
        }
        if pushed == 0 {
            let t = DeviceTable::empty(&schema)?; // the builder wants at least one (possibly empty) batch
            check(unsafe { sys::dfgpu_join_builder_push(b.0, t.0) })?;
        }
    }
    blocking(move || {
        let mut b = b;
        let mut ht = std::ptr::null_mut();
        let rc = unsafe { sys::dfgpu_join_builder_finish(b.0, &mut ht) }; // consumes the builder whether or not it succeeds
        b.0 = std::ptr::null_mut();
        check(rc)?;
        let ht = Arc::new(GpuJoinTable(ht));
        if let Some(acc) = bounds {
            acc.report(key_bounds)?; // ALWAYS once per build partition: the filter completes when the last one has reported
        }
        Ok(ht)
    }).await
}

impl BoundsAccumulator {
    /// one build partition's key bounds (None = it holds no non-NULL key); the filter is updated once every partition has reported
    fn report(&self, part: Option<(i64, i64)>) -> Result<()> {
        let mut st = self.state.lock().unwrap();
        if let Some((lo, hi)) = part {
            st.1 = Some(match st.1 { Some((a, b)) => (a.min(lo), b.max(hi)), None => (lo, hi) });
        }
        st.0 = st.0.saturating_sub(1);
        if st.0 > 0 {
            return Ok(());
        }
        let key = match self.filter.children().first() {
            Some(k) => Arc::clone(k),
            None => return Ok(()),
        };
        let new_expr: Arc<dyn PhysicalExpr> = match st.1 {
            // an empty build side matches nothing: the probe-side scan may skip everything
            None => Arc::new(Literal::new(ScalarValue::Boolean(Some(false)))),
            Some((lo, hi)) => {
                // scalars of the key column's OWN type, as the reference builds them (shared_bounds.rs: the bounds are ScalarValues
                // taken from the key arrays): `key >= Int64(lo)` over an Int32 / Date32 / UInt8 / UInt32 key would be an ill-typed
                // BinaryExpr — lost pruning, or an error once it runs as a row filter (pushdown_filters = true)
                let lit = |v: i64| -> Result<Arc<dyn PhysicalExpr>> { Ok(Arc::new(Literal::new(ScalarValue::Int64(Some(v)).cast_to(&self.key_type)?))) };
                let ge: Arc<dyn PhysicalExpr> = Arc::new(BinaryExpr::new(Arc::clone(&key), Operator::GtEq, lit(lo)?));
                let le: Arc<dyn PhysicalExpr> = Arc::new(BinaryExpr::new(key, Operator::LtEq, lit(hi)?));
                Arc::new(BinaryExpr::new(ge, Operator::And, le))
            }
        };
        self.filter.update(new_expr)?;
        self.filter.mark_complete();
        Ok(())
    }
}

impl DisplayAs for GpuHashJoinExec {
    fn fmt_as(&self, _t: DisplayFormatType, f: &mut std::fmt::Formatter) -> std::fmt::Result {
        write!(f, "GpuHashJoinExec: mode={:?}, join_type={:?}, on={:?}{}{}", self.mode, self.join_type, self.on,
               if self.filter.is_some() { ", filter" } else { "" }, if self.null_aware { ", null_aware" } else { "" })
    }
}

impl GpuNode for GpuHashJoinExec {
    fn execute_device(&self, partition: usize, ctx: Arc<TaskContext>) -> Result<DeviceFuture> {
        let collect_left = self.mode == PartitionMode::CollectLeft;
        let shared = Arc::clone(&self.builds[if collect_left { 0 } else { partition }]);
        let build_partition = if collect_left { 0 } else { partition };
        let probe_input = device_input(&self.right, partition, Arc::clone(&ctx))?; // lazy: nothing runs until polled
        let (left, bounds) = (Arc::clone(&self.left), self.bounds.clone());
        let (on_l, on_r): (Vec<i32>, Vec<i32>) = self.on.iter().copied().unzip();
        let (bo, po, reorder, filter) = (self.build_out.clone(), self.probe_out.clone(), self.reorder.clone(), self.filter.clone());
        let jt_enum = self.join_type;
        let jt = jt_enum as i32; // same discriminants as dfgpu_join_type (common/src/join_type.rs)
        let ne = matches!(self.null_equality, NullEquality::NullEqualsNull) as i32;
        let opts = self.options();
        // the NULL-filled probe columns of the unmatched build rows of Left / Full joins: types and names of the probe_out columns
        let right_schema = self.right.schema();
        let tail_probe: Vec<(sys::dfgpu_field, CString)> = if matches!(jt_enum, JoinType::Left | JoinType::Full) {
            po.iter().map(|c| {
                let f = right_schema.field(*c as usize);
                (field_of(f.data_type()).expect("checked by try_from_cpu"), CString::new(f.name().replace('\0', " ")).unwrap())
            }).collect()
        } else {
            vec![]
        };
        let out_schema = self.schema();
        Ok(async move {
            // ---- CollectBuildSide: whoever comes first builds; everybody else waits on the same cell (OnceAsync, exec.rs:1503-1523)
            let ht = shared.table.get_or_try_init(|| collect_build(left, build_partition, Arc::clone(&ctx), on_l, ne, opts, bounds)).await?.clone();
            // ---- probe: the whole partition in one launch sequence
            let probe = probe_input.await?;
            let (ht2, bo2, po2) = (Arc::clone(&ht), bo.clone(), po.clone());
            let matched = blocking(move || {
                let mut o = std::ptr::null_mut();
                match &filter {
                    None => check(unsafe { sys::dfgpu_join_probe(ht2.0, probe.0, on_r.as_ptr(), jt, bo2.as_ptr(), bo2.len() as i32, po2.as_ptr(), po2.len() as i32, &mut o) })?,
                    Some(f) => {
                        let jf = sys::dfgpu_join_filter { expression: f.expression.as_c(), column_index: f.column_index.as_ptr(), column_side: f.column_side.as_ptr(),
                                                          n_columns: f.column_index.len() as i32 };
                        check(unsafe { sys::dfgpu_join_probe_with_filter(ht2.0, probe.0, on_r.as_ptr(), jt, &jf, bo2.as_ptr(), bo2.len() as i32, po2.as_ptr(),
                                                                         po2.len() as i32, &mut o) })?
                    }
                }
                Ok(DeviceTable(o))
            }).await?;
            // ---- ExhaustedProbeSide: the build rows are reported ONCE, by the probe partition that finishes last (exec.rs:1312-1330)
            let out = if !emits_build_side(jt_enum) {
                matched
            } else {
                let last = shared.remaining.fetch_sub(1, Ordering::AcqRel) == 1;
                let tail = if last {
                    Some(blocking(move || {
                        let fields: Vec<sys::dfgpu_field> = tail_probe.iter().map(|(f, _)| *f).collect();
                        let names: Vec<*const std::os::raw::c_char> = tail_probe.iter().map(|(_, n)| n.as_ptr()).collect();
                        let mut o = std::ptr::null_mut();
                        check(unsafe { sys::dfgpu_join_emit_unmatched(ht.0, jt, bo.as_ptr(), bo.len() as i32, fields.as_ptr(), names.as_ptr(), fields.len() as i32, &mut o) })?;
                        Ok(DeviceTable(o))
                    }).await?)
                } else {
                    None
                };
                match (jt_enum, tail) {
                    // Left / Full: this partition's matched pairs (+ Full: its unmatched probe rows), then — last partition only —
                    // the unmatched build rows
                    (JoinType::Left | JoinType::Full, Some(t)) => blocking(move || DeviceTable::concat(&[matched, t])).await?,
                    (JoinType::Left | JoinType::Full, None) => matched,
                    // LeftSemi / LeftAnti / LeftMark: the probe call only marks; the rows are the build side's, reported at the end
                    (_, Some(t)) => t,
                    // ... so a partition that is not the last contributes no row: an empty table of the node's schema
                    (_, None) => return DeviceTable::empty(&out_schema),
                }
            };
            // the node's column order (an interleaving projection); zero-copy
            if reorder.iter().enumerate().all(|(i, c)| i as i32 == *c) { Ok(out) } else { out.select(&reorder) }
        }
        .boxed())
    }
}

impl ExecutionPlan for GpuHashJoinExec {
    fn name(&self) -> &str { "GpuHashJoinExec" }
    fn properties(&self) -> &Arc<PlanProperties> { &self.cache }
    fn children(&self) -> Vec<&Arc<dyn ExecutionPlan>> { vec![&self.left, &self.right] }
    fn maintains_input_order(&self) -> Vec<bool> {
        // HashJoinExec::maintains_input_order (exec.rs:1024-1037) unless the rule proved that nobody looks
        vec![false, !self.order_insensitive && matches!(self.join_type, JoinType::Inner | JoinType::Right | JoinType::RightAnti | JoinType::RightSemi | JoinType::RightMark)]
    }
    fn apply_expressions(&self, _f: &mut dyn FnMut(&Arc<dyn PhysicalExpr>) -> Result<TreeNodeRecursion>) -> Result<TreeNodeRecursion> {
        Ok(TreeNodeRecursion::Continue) // keys are column indices, the JoinFilter is lowered: no PhysicalExpr is evaluated by this node
    }
    fn replace_children(self: Arc<Self>, c: Vec<Arc<dyn ExecutionPlan>>, _options: ReplaceChildrenOptions) -> Result<Arc<dyn ExecutionPlan>> {
        if c.len() != 2 {
            return Err(DataFusionError::Internal("GpuHashJoinExec takes two children".into()));
        }
        // fresh build state: a plan with new children is a new execution (as HashJoinExecBuilder::reset_state, exec.rs:410)
        Ok(Arc::new(Self {
            left: Arc::clone(&c[0]), right: Arc::clone(&c[1]), on: self.on.clone(), filter: self.filter.clone(), join_type: self.join_type,
            null_equality: self.null_equality, null_aware: self.null_aware, mode: self.mode, build_out: self.build_out.clone(), probe_out: self.probe_out.clone(),
            reorder: self.reorder.clone(), order_insensitive: self.order_insensitive, bounds: self.bounds.clone(), builds: builds_for(self.mode, &c[1]),
            cache: Arc::clone(&self.cache),
        }))
    }
    #[allow(deprecated)]
    fn with_new_children(self: Arc<Self>, c: Vec<Arc<dyn ExecutionPlan>>) -> Result<Arc<dyn ExecutionPlan>> {
        self.replace_children(c, ReplaceChildrenOptions::new(ChildrenPropertiesMode::Recompute))
    }
    fn reset_state(self: Arc<Self>) -> Result<Arc<dyn ExecutionPlan>> {
        let children = vec![Arc::clone(&self.left), Arc::clone(&self.right)];
        self.replace_children(children, ReplaceChildrenOptions::new(ChildrenPropertiesMode::Keep))
    }
    fn execute(&self, partition: usize, ctx: Arc<TaskContext>) -> Result<SendableRecordBatchStream> {
        let batch_size = ctx.session_config().batch_size();
        let schema: SchemaRef = self.schema();
        Ok(host_stream(schema, self.execute_device(partition, ctx)?, batch_size))
    }
}
