//! Device-resident hand-off between adjacent GPU nodes (the round-2 review's missing 7): `ExecutionPlan::execute`
//! (execution_plan.rs:696-700) yields host RecordBatches, so two GPU nodes next to each other would export and re-import every
//! partition.  Every GPU node of this crate therefore ALSO implements `GpuNode::execute_device`, which resolves to the node's whole
//! output partition as ONE `DeviceTable`; a GPU parent asks its child for that (`device_input`) and only falls back to uploading
//! RecordBatches when the child is a CPU operator.  `execute` itself is `execute_device` + `host_stream` (export `batch_size` rows per
//! poll).  A Filter -> Aggregate -> Sort chain thus crosses PCIe once on the way in and once on the way out — the sequence
//! tests/c/plan_driver.c runs with `dfgpu_metrics` reading zero PCIe bytes in between.  To a consumer outside this crate a
//! device table can be handed as an Arrow C Device array (`DeviceTable::export_device`, ARROW_DEVICE_ROCM).
use crate::hash_join::GpuHashJoinExec;
use crate::operators::GpuUnaryExec;
use crate::table::DeviceTable;
use arrow::datatypes::SchemaRef;
use datafusion::error::{DataFusionError, Result};
use datafusion::execution::{SendableRecordBatchStream, TaskContext};
use datafusion::physical_plan::stream::RecordBatchStreamAdapter;
use datafusion::physical_plan::ExecutionPlan;
use futures::future::BoxFuture;
use futures::{FutureExt, StreamExt, TryStreamExt};
use std::sync::Arc;

pub type DeviceFuture = BoxFuture<'static, Result<DeviceTable>>;

pub trait GpuNode {
    /// the node's output partition as one device table; nothing runs until the future is polled (execution_plan.rs:514-516)
    fn execute_device(&self, partition: usize, ctx: Arc<TaskContext>) -> Result<DeviceFuture>;
}

pub fn as_gpu_node(plan: &Arc<dyn ExecutionPlan>) -> Option<&dyn GpuNode> {
    if let Some(n) = plan.downcast_ref::<GpuUnaryExec>() {
        return Some(n);
    }
    if let Some(n) = plan.downcast_ref::<GpuHashJoinExec>() {
        return Some(n);
    }
    if let Some(n) = plan.downcast_ref::<crate::scan::GpuIpcScanExec>() {
        return Some(n);
    }
    if let Some(n) = plan.downcast_ref::<crate::scan::GpuParquetScanExec>() {
        return Some(n);
    }
    None
}

/// partition `partition` of `child` as a device table: a GPU child hands its table over, a CPU child's batches are uploaded one by
/// one (dfgpu_table_import: pinned, side stream) and concatenated on the device (a launch wants >= 10^6 rows, not 8192)
pub fn device_input(child: &Arc<dyn ExecutionPlan>, partition: usize, ctx: Arc<TaskContext>) -> Result<DeviceFuture> {
    if let Some(gpu) = as_gpu_node(child) {
        return gpu.execute_device(partition, ctx);
    }
    let mut stream = child.execute(partition, ctx)?;
    let schema = child.schema();
    Ok(async move {
        let mut parts = vec![];
        while let Some(batch) = stream.next().await {
            parts.push(DeviceTable::from_batch(&batch?)?);
        }
        if parts.is_empty() {
            return DeviceTable::empty(&schema); // an exhausted child still has a schema: operators see a 0-row table
        }
        if parts.len() == 1 {
            return Ok(parts.pop().unwrap());
        }
        crate::blocking(move || DeviceTable::concat(&parts)).await
    }
    .boxed())
}

/// the host face of a GPU node: the device table leaves `batch_size` rows at a time in pinned buffers (LimitedBatchCoalescer's
/// fixed target, coalesce/mod.rs:27-120); dropping the stream frees the table (drop = cancel, execution_plan.rs:539-547)
pub fn host_stream(schema: SchemaRef, table: DeviceFuture, batch_size: usize) -> SendableRecordBatchStream {
    let out_schema = Arc::clone(&schema);
    let fut = async move {
        let table = Arc::new(table.await?);
        let n = table.num_rows()?;
        let step = batch_size.max(1) as i64;
        let offsets: Vec<i64> = if n == 0 { vec![] } else { (0..n).step_by(step as usize).collect() };
        let batches = futures::stream::iter(offsets).then(move |off| {
            let (table, schema) = (Arc::clone(&table), Arc::clone(&out_schema));
            async move { crate::blocking(move || table.export_batch(off, step.min(n - off), &schema)).await }
        });
        Ok::<_, DataFusionError>(batches)
    };
    Box::pin(RecordBatchStreamAdapter::new(schema, futures::stream::once(fut).try_flatten()))
}
