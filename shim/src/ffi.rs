//! The run-time-loadable face of this crate (crate-type cdylib) over datafusion-ffi, for a host that does not link the crate:
//!
//!   * `dfgpu_create_physical_optimizer_rule(devices, n)` -> `FFI_PhysicalOptimizerRule` (ffi/src/physical_optimizer.rs:115-149,
//!     constructor `FFI_PhysicalOptimizerRule::new`, :249-272).  NOTE what the FFI boundary does to a rule: the host's plan arrives
//!     as `FFI_ExecutionPlan` and is rebuilt as `ForeignExecutionPlan` nodes on this side unless host and rule live in the SAME
//!     library image (`library_marker_id`, execution_plan.rs:48-115) — foreign nodes expose name / children / properties /
//!     expressions but not HashJoinExec's join type, keys or partition mode, so a rule that downcasts sees nothing to replace.
//!     This constructor therefore serves hosts that resolve to the same image (the rule handed through a plugin registry of
//!     one binary); for a separately built host use the planner below.
//!   * `dfgpu_create_query_planner(devices, n, logical_codec, physical_codec)` -> `FFI_QueryPlanner` (ffi/src/query_planner.rs:90,
//!     constructor :252).  Planning happens HERE: DefaultPhysicalPlanner builds the plan out of this library's own node types
//!     over the host's (foreign) session, catalog and table providers, the host's physical optimizer rules run as part of it,
//!     then GpuOffloadRule rewrites the result; the plan goes back as an FFI_ExecutionPlan whose GPU nodes execute in this library.
//!   * `datafusion_gpu_amd_get_module()` -> a struct of the two constructors + version, the loading convention of the
//!     reference's own integration tests (`datafusion_ffi_get_module`, ffi/src/tests/mod.rs:344-374; loader ffi/src/tests/utils.rs:65-90).
use crate::GpuOffloadRule;
use async_trait::async_trait;
use datafusion::error::Result;
use datafusion::execution::context::QueryPlanner;
use datafusion::logical_expr::LogicalPlan;
use datafusion::physical_optimizer::PhysicalOptimizerRule;
use datafusion::physical_plan::ExecutionPlan;
use datafusion::physical_planner::{DefaultPhysicalPlanner, PhysicalPlanner};
use datafusion::catalog::Session;
use datafusion_ffi::physical_optimizer::FFI_PhysicalOptimizerRule;
use datafusion_ffi::proto::logical_extension_codec::FFI_LogicalExtensionCodec;
use datafusion_ffi::proto::physical_extension_codec::FFI_PhysicalExtensionCodec;
use datafusion_ffi::query_planner::FFI_QueryPlanner;
use std::sync::Arc;

fn devices_of(devices: *const i32, n: i32) -> Vec<i32> {
    if devices.is_null() || n <= 0 { vec![0] } else { unsafe { std::slice::from_raw_parts(devices, n as usize) }.to_vec() }
}

/// the rule behind the stable ABI.  Panics (= aborts the load, as the reference's constructors do on a version mismatch) when
/// libdfgpu.so cannot take the devices: a host that asked for the GPU rule must not silently run without it.
#[unsafe(no_mangle)]
pub extern "C" fn dfgpu_create_physical_optimizer_rule(devices: *const i32, n_devices: i32) -> FFI_PhysicalOptimizerRule {
    let rule = GpuOffloadRule::new(&devices_of(devices, n_devices)).expect("dfgpu_init failed: no MI355X visible or libdfgpu.so mismatch");
    let rule: Arc<dyn PhysicalOptimizerRule + Send + Sync> = Arc::new(rule);
    FFI_PhysicalOptimizerRule::new(rule, None)
}

/// QueryPlanner (session/src/planner.rs:34-41) that plans inside this library and offloads what it can
#[derive(Debug)]
struct GpuQueryPlanner {
    rule: GpuOffloadRule,
}

#[async_trait]
impl QueryPlanner for GpuQueryPlanner {
    async fn create_physical_plan(&self, logical_plan: &LogicalPlan, session: &dyn Session) -> Result<Arc<dyn ExecutionPlan>> {
        // the default planner runs the session's physical optimizer rules itself (physical_planner.rs:2909-2926), so the join
        // sides, partition modes and distributions are final when the GPU rule looks at the plan — the position a user rule has
        let plan = DefaultPhysicalPlanner::default().create_physical_plan(logical_plan, session).await?;
        self.rule.optimize(plan, session.config_options())
    }
}

#[unsafe(no_mangle)]
pub extern "C" fn dfgpu_create_query_planner(devices: *const i32, n_devices: i32, logical_codec: FFI_LogicalExtensionCodec,
                                             physical_codec: FFI_PhysicalExtensionCodec) -> FFI_QueryPlanner {
    let rule = GpuOffloadRule::new(&devices_of(devices, n_devices)).expect("dfgpu_init failed: no MI355X visible or libdfgpu.so mismatch");
    let planner: Arc<dyn QueryPlanner + Send + Sync> = Arc::new(GpuQueryPlanner { rule });
    FFI_QueryPlanner::new_with_ffi_codecs(planner, logical_codec, physical_codec)
}

/// what a host `dlopen`s: `datafusion_gpu_amd_get_module()`, then checks `version` against its own datafusion_ffi::version()
#[repr(C)]
pub struct GpuAmdModule {
    pub create_physical_optimizer_rule: extern "C" fn(devices: *const i32, n_devices: i32) -> FFI_PhysicalOptimizerRule,
    pub create_query_planner: extern "C" fn(devices: *const i32, n_devices: i32, logical_codec: FFI_LogicalExtensionCodec,
                                            physical_codec: FFI_PhysicalExtensionCodec) -> FFI_QueryPlanner,
    /// major DataFusion version this library was built against (ffi/src/lib.rs `version`)
    pub version: extern "C" fn() -> u64,
    /// DFGPU_ABI_VERSION of the libdfgpu.so this library is linked to
    pub dfgpu_abi_version: extern "C" fn() -> i32,
}

extern "C" fn abi_version() -> i32 {
    unsafe { crate::sys::dfgpu_abi_version() }
}

#[unsafe(no_mangle)]
pub extern "C" fn datafusion_gpu_amd_get_module() -> GpuAmdModule {
    GpuAmdModule {
        create_physical_optimizer_rule: dfgpu_create_physical_optimizer_rule,
        create_query_planner: dfgpu_create_query_planner,
        version: datafusion_ffi::version,
        dfgpu_abi_version: abi_version,
    }
}
