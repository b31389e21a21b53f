//! `GpuOffloadRule`: PhysicalOptimizerRule (datafusion/session/src/physical_optimizer.rs:52-84), `schema_check() == true`: every
//! replacement node has the identical output schema.  User rules run after the built-in ones (core/src/physical_planner.rs:2909-2926),
//! i.e. after EnsureRequirements and JoinSelection have fixed distribution, ordering, build side and partition mode — the GPU
//! nodes only copy PlanProperties.  Python twin (tested against the reference's pinned TPC-H plans): datafusion_amd/physical_plan.py.
//!
//! Two plan-wide facts are established before anything is replaced:
//!   * which nodes have an ancestor that OBSERVES their output order (`needs_order`, top-down: a parent passes the need on through
//!     `maintains_input_order()` and creates it with `required_input_ordering()`; the root's order is the query's).  A hash join
//!     nobody looks at may emit its rows in tile order (probe_mode 4; hash_join/exec.rs:3349 is what it gives up).
//!   * whether EVERY `RepartitionExec(Hash)` of the plan can move to the GPU.  The library routes rows with its own hash
//!     (dfgpu_partition), DataFusion with ahash under REPARTITION_RANDOM_STATE (repartition/mod.rs:1097-1150): co-partitioned
//!     inputs of a Partitioned join or a FinalPartitioned aggregate only meet in the same partition when both sides were routed
//!     by the same function — so hash repartitions are replaced all together or not at all.
use crate::expr::{field_of, lower, Lowered};
use crate::hash_join::GpuHashJoinExec;
use crate::operators::GpuUnaryExec;
use crate::sys;
use datafusion::config::ConfigOptions;
use datafusion::error::Result;
use datafusion::physical_optimizer::PhysicalOptimizerRule;
use datafusion::physical_plan::aggregates::AggregateExec;
use datafusion::physical_plan::filter::FilterExec;
use datafusion::physical_plan::joins::HashJoinExec;
use datafusion::physical_plan::projection::ProjectionExec;
use datafusion::physical_plan::repartition::RepartitionExec;
use datafusion::physical_plan::sorts::sort::SortExec;
use datafusion::physical_plan::{ExecutionPlan, Partitioning};
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

    /// every hash repartition of the plan has a device form (see the module comment)
    fn all_hash_repartitions_offloadable(plan: &Arc<dyn ExecutionPlan>) -> bool {
        if let Some(r) = plan.downcast_ref::<RepartitionExec>() {
            if matches!(r.partitioning(), Partitioning::Hash(..)) && GpuUnaryExec::try_from_repartition(r).is_none() {
                return false;
            }
        }
        plan.children().into_iter().all(Self::all_hash_repartitions_offloadable)
    }

    /// children first (a GPU parent sees that its child already produces device tables), then the node itself
    fn rewrite(&self, node: Arc<dyn ExecutionPlan>, needs_order: bool, repartitions: bool) -> Result<Arc<dyn ExecutionPlan>> {
        let maintains = node.maintains_input_order();
        let required = node.required_input_ordering();
        let mut changed = false;
        let mut children = vec![];
        for (i, c) in node.children().into_iter().enumerate() {
            // child i's order is observed if this node demands one of it, or hands it on to somebody who looks
            let child_needs = required.get(i).is_some_and(|r| r.is_some()) || (needs_order && maintains.get(i).copied().unwrap_or(false));
            let new = self.rewrite(Arc::clone(c), child_needs, repartitions)?;
            changed |= !Arc::ptr_eq(&new, c);
            children.push(new);
        }
        let node = if changed { datafusion::physical_plan::execution_plan::replace_children_if_necessary(node, children)? } else { node };

        if let Some(j) = node.downcast_ref::<HashJoinExec>() {
            if Self::admits(j) {
                if let Some(g) = GpuHashJoinExec::try_from_cpu(j, /*order_insensitive=*/ !needs_order) {
                    return Ok(Arc::new(g));
                }
            }
            return Ok(node);
        }
        // the single-input operators (operators.rs).  An AggregateExec directly over a GpuFilterExec absorbs it
        // (dfgpu_agg_update_filtered: filter, argument expressions and accumulation in ONE pass over the input columns — the shape of
        // TPC-H Q1 and Q6).
        let replaced: Option<GpuUnaryExec> = if let Some(f) = node.downcast_ref::<FilterExec>() {
            GpuUnaryExec::try_from_filter(f)
        } else if let Some(p) = node.downcast_ref::<ProjectionExec>() {
            GpuUnaryExec::try_from_projection(p)
        } else if let Some(a) = node.downcast_ref::<AggregateExec>() {
            GpuUnaryExec::try_from_aggregate(a, a.input().downcast_ref::<GpuUnaryExec>())
        } else if let Some(so) = node.downcast_ref::<SortExec>() {
            GpuUnaryExec::try_from_sort(so)
        } else if let Some(r) = node.downcast_ref::<RepartitionExec>() {
            if repartitions { GpuUnaryExec::try_from_repartition(r) } else { None }
        } else {
            None
        };
        if let Some(g) = replaced {
            return Ok(Arc::new(g));
        }
        // a leaf scan of one local Arrow IPC / Parquet file: straight into HBM (scan.rs), the device chunk cache in front
        if let Some(d) = node.downcast_ref::<datafusion::datasource::source::DataSourceExec>() {
            if let Some(scan) = crate::scan::try_from_data_source(d) {
                return Ok(scan);
            }
        }
        Ok(node)
    }
}

impl PhysicalOptimizerRule for GpuOffloadRule {
    fn optimize(&self, plan: Arc<dyn ExecutionPlan>, _cfg: &ConfigOptions) -> Result<Arc<dyn ExecutionPlan>> {
        let repartitions = Self::all_hash_repartitions_offloadable(&plan);
        self.rewrite(plan, /*the root's order is the query's*/ true, repartitions)
    }
    fn name(&self) -> &str { "gpu_offload_amd" }
    fn schema_check(&self) -> bool { true }
}

/// lowering helpers re-exported for embedders that build GPU nodes by hand
pub fn lowerable(e: &Arc<dyn datafusion::physical_expr::PhysicalExpr>, schema: &arrow::datatypes::Schema) -> bool {
    lower(e, schema, &mut Lowered::default()).is_some() && schema.fields().iter().all(|f| field_of(f.data_type()).is_some())
}
