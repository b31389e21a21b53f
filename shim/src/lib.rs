//! DataFusion on AMD MI355X: a `PhysicalOptimizerRule` that replaces the vectorized physical operators of a plan
//! (hash join, hash aggregate, filter / projection, sort / TopK, hash repartition, Parquet column-chunk decode) by
//! `ExecutionPlan` nodes that call `libdfgpu.so` (hand-written HIP for gfx950) through the C ABI of `include/dfgpu.h`.
//!
//! Registration: `SessionStateBuilder::with_physical_optimizer_rule(Arc::new(GpuOffloadRule::new(devices)))`
//! (datafusion/core/src/execution/session_state.rs:1407-1415).  Everything the rule does not recognise stays on the CPU.
//!
//! This crate is source only in this repository (no Rust toolchain in the build image): `sys.rs` is generated from the header
//! and drift-checked (scripts/gen_shim_sys.py --check, tests/test_abi.py); the other modules (`hash_join.rs`: GpuHashJoinExec,
//! `operators.rs`: filter / projection / aggregate incl. the fused filter / sort and TopK / hash repartition) mirror, call for call,
//! `datafusion_amd/{table,expr,physical_plan}.py`, which the parity tests drive through the same entry points.
pub mod expr;
pub mod hash_join;
pub mod operators;
pub mod rule;
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
    } else {
        Err(DataFusionError::Execution(msg))
    }
}

/// once per process: the GPUs this DataFusion process drives (one per output partition when it owns several)
pub fn init(devices: &[i32]) -> Result<()> {
    assert_eq!(unsafe { sys::dfgpu_abi_version() }, sys::DFGPU_ABI_VERSION, "libdfgpu.so and this crate were built from different headers");
    check(unsafe { sys::dfgpu_init(devices.as_ptr(), devices.len() as i32) })
}
