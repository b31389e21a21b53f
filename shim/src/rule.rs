//! `GpuOffloadRule`: PhysicalOptimizerRule (datafusion/session/src/physical_optimizer.rs:52-84), `schema_check() == true`: every
//! replacement node has the identical output schema.  User rules run after the built-in ones (core/src/physical_planner.rs:2909-2926),
//! i.e. after EnsureRequirements and JoinSelection have fixed distribution, ordering, build side and partition mode — the GPU
//! nodes only copy PlanProperties.  Python twin (tested against the reference's pinned TPC-H plans): datafusion_amd/physical_plan.py.
use crate::expr::{field_of, lower, Lowered};
use crate::hash_join::GpuHashJoinExec;
use crate::operators::GpuUnaryExec;
use crate::{check, sys};
use datafusion::common::tree_node::{Transformed, TreeNode};
use datafusion::config::ConfigOptions;
use datafusion::error::Result;
use datafusion::physical_optimizer::PhysicalOptimizerRule;
use datafusion::physical_plan::aggregates::AggregateExec;
use datafusion::physical_plan::filter::FilterExec;
use datafusion::physical_plan::joins::HashJoinExec;
use datafusion::physical_plan::projection::ProjectionExec;
use datafusion::physical_plan::repartition::RepartitionExec;
use datafusion::physical_plan::sorts::sort::SortExec;
use datafusion::physical_plan::ExecutionPlan;
use std::sync::Arc;

#[derive(Debug, Default)]
pub struct GpuOffloadRule;

impl GpuOffloadRule {
    pub fn new(devices: &[i32]) -> Result<Self> {
        crate::init(devices)?;
        Ok(Self)
    }

    /// spill-aware fallback: a join whose device footprint the pool cannot admit stays the CPU operator, which can spill
    fn admits(j: &HashJoinExec) -> bool {
        let rows = |p: &Arc<dyn ExecutionPlan>| p.partition_statistics(None).ok().and_then(|s| s.num_rows.get_value().copied());
        let (Some(b), Some(p)) = (rows(j.left()), rows(j.right())) else { return true }; // no statistics: decided at run time by the builder's reservation
        let row_bytes = |p: &Arc<dyn ExecutionPlan>| p.schema().fields().iter().map(|f| f.data_type().primitive_width().unwrap_or(24) as i64).sum::<i64>();
        let mut need = 0i64;
        let ok = unsafe { sys::dfgpu_join_estimate_bytes(b as i64, row_bytes(j.left()), p as i64, -1, row_bytes(j.left()) + row_bytes(j.right()), &mut need) } == 0;
        ok && crate::table::Reservation::try_new(need).is_ok()
    }
}

impl PhysicalOptimizerRule for GpuOffloadRule {
    fn optimize(&self, plan: Arc<dyn ExecutionPlan>, _cfg: &ConfigOptions) -> Result<Arc<dyn ExecutionPlan>> {
        // bottom-up: children first, so a GPU parent sees that its child already produces device tables
        plan.transform_up(|node| {
            if let Some(j) = node.as_any().downcast_ref::<HashJoinExec>() {
                let types_ok = j.schema().fields().iter().all(|f| field_of(f.data_type()).is_some());
                let filter_ok = j.filter().map_or(true, |f| lower(f.expression(), f.schema(), &mut Lowered::default()).is_some());
                if types_ok && filter_ok && Self::admits(j) {
                    return Ok(Transformed::yes(Arc::new(GpuHashJoinExec::try_from_cpu(j, /*order_insensitive=*/ false)?) as _));
                }
            }
            // the single-input operators (operators.rs).  Children were visited first: an AggregateExec directly over a GpuFilterExec
            // absorbs it (dfgpu_agg_update_filtered: filter, argument expressions and accumulation in ONE pass over the input columns —
            // the shape of TPC-H Q1 and Q6).  The probe-side twin (GpuHashJoinExec over a GpuFilterExec -> dfgpu_join_probe_filtered)
            // is implemented by the Python twin (physical_plan.py GpuHashJoinExec.probe_predicate) and follows the same pattern.
            let any = node.as_any();
            let replaced: Option<GpuUnaryExec> = if let Some(f) = any.downcast_ref::<FilterExec>() {
                GpuUnaryExec::try_from_filter(f)
            } else if let Some(p) = any.downcast_ref::<ProjectionExec>() {
                GpuUnaryExec::try_from_projection(p)
            } else if let Some(a) = any.downcast_ref::<AggregateExec>() {
                GpuUnaryExec::try_from_aggregate(a, a.input().as_any().downcast_ref::<GpuUnaryExec>())
            } else if let Some(so) = any.downcast_ref::<SortExec>() {
                GpuUnaryExec::try_from_sort(so)
            } else if let Some(r) = any.downcast_ref::<RepartitionExec>() {
                GpuUnaryExec::try_from_repartition(r)
            } else {
                None
            };
            if let Some(g) = replaced {
                return Ok(Transformed::yes(Arc::new(g) as _));
            }
            Ok(Transformed::no(node))
        }).map(|t| t.data)
    }
    fn name(&self) -> &str { "gpu_offload_amd" }
    fn schema_check(&self) -> bool { true }
}

#[allow(dead_code)]
fn _abi_guard() -> Result<()> {
    check(if unsafe { sys::dfgpu_abi_version() } == sys::DFGPU_ABI_VERSION { 0 } else { 1 })
}
