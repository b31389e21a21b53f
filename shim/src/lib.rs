//! DataFusion on AMD MI355X: a `PhysicalOptimizerRule` that replaces the vectorized physical operators of a plan
//! (hash join, hash aggregate, filter / projection, sort / TopK, hash repartition) by `ExecutionPlan` nodes that call
//! `libdfgpu.so` (hand-written HIP for gfx950) through the C ABI of `include/dfgpu.h`.
//!
//! Registration, statically linked: `SessionStateBuilder::with_physical_optimizer_rule(Arc::new(GpuOffloadRule::new(devices)?))`
//! (datafusion/core/src/execution/session_state.rs:1407-1415).  Loaded at run time: `ffi.rs` exports the rule as an
//! `FFI_PhysicalOptimizerRule` and a planner as an `FFI_QueryPlanner`.  Everything the rule does not recognise stays on the CPU.
//!
//! This crate is source only in this repository (no Rust toolchain in the build image).  What keeps it honest:
//!   * `sys.rs` is generated from the header and drift-checked (scripts/gen_shim_sys.py --check, tests/test_abi.py);
//!   * the call sequences of `hash_join.rs` (builder push -> finish once -> probe per partition -> emit_unmatched once ->
//!     export_batch) and of a Filter -> Aggregate -> Sort chain handing device tables on are executed from plain C against
//!     the reference's snapshot tests (tests/c/plan_driver.c, tests/test_gpu_c_driver.py);
//!   * `datafusion_amd/physical_plan.py` is the same rule and the same nodes in Python over the same entry points, run against
//!     the reference's 22 pinned TPC-H plans.
pub mod device;
pub mod expr;
pub mod ffi;
pub mod hash_join;
pub mod operators;
pub mod rule;
pub mod scan;
pub mod sys;
pub mod table;

pub use rule::GpuOffloadRule;

use datafusion::error::{DataFusionError, Result};

/// `int` return code of an entry point -> `Result`, with the library's thread-local message (dfgpu_last_error).  The
/// reference's own error texts are kept by the library ("Resources exhausted: ...", "Arrow error: Divide by zero error").
pub(crate) fn check(rc: std::os::raw::c_int) -> Result<()> {
    if rc == 0 {
        return Ok(());
    }
    let msg = unsafe { std::ffi::CStr::from_ptr(sys::dfgpu_last_error()) }.to_string_lossy().into_owned();
    if msg.starts_with("Resources exhausted") {
        Err(DataFusionError::ResourcesExhausted(msg))
    } else if let Some(arrow) = msg.strip_prefix("Arrow error: ") {
        Err(DataFusionError::ArrowError(Box::new(arrow::error::ArrowError::ComputeError(arrow.to_string())), None))
    } else {
        Err(DataFusionError::Execution(msg))
    }
}

/// once per process: the GPUs this DataFusion process drives (one per output partition when it owns several)
pub fn init(devices: &[i32]) -> Result<()> {
    assert_eq!(unsafe { sys::dfgpu_abi_version() }, sys::DFGPU_ABI_VERSION, "libdfgpu.so and this crate were built from different headers");
    check(unsafe { sys::dfgpu_init(devices.as_ptr(), devices.len() as i32) })
}

/// a HIP wait must never sit on an executor thread (execution_plan.rs:549-565): every call sequence runs on the blocking pool
pub(crate) async fn blocking<T: Send + 'static>(f: impl FnOnce() -> Result<T> + Send + 'static) -> Result<T> {
    tokio::task::spawn_blocking(f).await.map_err(|e| DataFusionError::External(Box::new(e)))?
}
