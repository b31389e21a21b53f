//! Scan -> device below the C ABI, from the Rust side: the Arrow IPC reader (`dfgpu_ipc_*`, csrc/ipc.hip) and the device-resident scan
//! cache (`dfgpu_cache_*`, csrc/interop.hip) as a leaf `ExecutionPlan` — the GPU twin of `DataSourceExec` over an `ArrowSource`
//! (datasource-arrow/src/source.rs:260-330) whose output partition is ONE device table handed to the GPU node above it (device.rs),
//! or exported `batch_size` rows at a time to a CPU parent.  A file scanned before costs no host read and no PCIe transfer: its
//! batches are views of HBM (`MemorySourceConfig`'s role, datasource/src/memory.rs:58).  Python twin: datafusion_amd/ipc.py.
//! `GpuParquetScanExec` below is the Parquet twin: the footer through the `parquet` crate, the projected column chunks through ONE
//! `dfgpu_parquet_read_chunks` call.
use crate::device::{host_stream, DeviceFuture, GpuNode};
use crate::table::DeviceTable;
use crate::{blocking, check, sys};
use arrow::datatypes::SchemaRef;
use datafusion::common::tree_node::TreeNodeRecursion;
use datafusion::error::{DataFusionError, Result};
use datafusion::execution::{SendableRecordBatchStream, TaskContext};
use datafusion::physical_expr::{EquivalenceProperties, PhysicalExpr};
use datafusion::physical_plan::execution_plan::{Boundedness, EmissionType};
use datafusion::physical_plan::{DisplayAs, DisplayFormatType, ExecutionPlan, Partitioning, PlanProperties, ReplaceChildrenOptions};
use futures::FutureExt;
use std::sync::{Arc, OnceLock};

/// process-wide device cache (budget: DFGPU_TABLE_CACHE_BYTES, default 16 GiB)
pub struct ScanCache(sys::dfgpu_cache_t);
unsafe impl Send for ScanCache {}
unsafe impl Sync for ScanCache {}
impl ScanCache {
    pub fn global() -> &'static ScanCache {
        static CACHE: OnceLock<ScanCache> = OnceLock::new();
        CACHE.get_or_init(|| {
            let budget = std::env::var("DFGPU_TABLE_CACHE_BYTES").ok().and_then(|v| v.parse::<i64>().ok()).unwrap_or(16 << 30);
            let mut h = std::ptr::null_mut();
            check(unsafe { sys::dfgpu_cache_create(budget, &mut h) }).expect("dfgpu_cache_create");
            ScanCache(h)
        })
    }
    pub fn get(&self, key: &[u8]) -> Result<Option<DeviceTable>> {
        let mut out = std::ptr::null_mut();
        check(unsafe { sys::dfgpu_cache_get(self.0, key.as_ptr() as *const _, key.len() as i64, &mut out) })?;
        Ok(if out.is_null() { None } else { Some(DeviceTable(out)) })
    }
    pub fn put(&self, key: &[u8], table: &DeviceTable) -> Result<()> {
        check(unsafe { sys::dfgpu_cache_put(self.0, key.as_ptr() as *const _, key.len() as i64, table.0) })
    }
}

/// one memory-mapped Arrow IPC file; the library has walked its messages (`dfgpu_ipc_open` needs no GPU)
struct IpcFile {
    handle: sys::dfgpu_ipc_t,
    _map: memmap2::Mmap, // the bytes must outlive the handle
    identity: Vec<u8>,   // path + mtime + size: the cache key prefix
    n_batches: i64,
}
unsafe impl Send for IpcFile {}
unsafe impl Sync for IpcFile {}
impl Drop for IpcFile {
    fn drop(&mut self) {
        unsafe { sys::dfgpu_ipc_close(self.handle) };
    }
}

#[derive(Debug)]
pub struct GpuIpcScanExec {
    path: String,
    /// indices into the file's schema (the scan's projection), in output order
    projection: Vec<i32>,
    schema: SchemaRef,
    cache: Arc<PlanProperties>,
}

impl GpuIpcScanExec {
    pub fn new(path: String, projection: Vec<i32>, schema: SchemaRef) -> Self {
        let props = PlanProperties::new(EquivalenceProperties::new(Arc::clone(&schema)), Partitioning::UnknownPartitioning(1), EmissionType::Final, Boundedness::Bounded);
        Self { path, projection, schema, cache: Arc::new(props) }
    }

    fn open(&self) -> Result<IpcFile> {
        let file = std::fs::File::open(&self.path)?;
        let meta = file.metadata()?;
        let map = unsafe { memmap2::Mmap::map(&file) }.map_err(|e| DataFusionError::External(Box::new(e)))?;
        let mut handle = std::ptr::null_mut();
        check(unsafe { sys::dfgpu_ipc_open(map.as_ptr(), map.len() as i64, &mut handle) })?;
        let (mut n_batches, mut n_cols, mut is_file) = (0i64, 0i32, 0i32);
        check(unsafe { sys::dfgpu_ipc_info(handle, &mut n_batches, &mut n_cols, &mut is_file) })?;
        let mtime = meta.modified().ok().and_then(|t| t.duration_since(std::time::UNIX_EPOCH).ok()).map_or(0, |d| d.as_nanos());
        let identity = format!("{}|{}|{}", self.path, mtime, meta.len()).into_bytes();
        Ok(IpcFile { handle, _map: map, identity, n_batches })
    }
}

impl DisplayAs for GpuIpcScanExec {
    fn fmt_as(&self, _t: DisplayFormatType, f: &mut std::fmt::Formatter) -> std::fmt::Result {
        write!(f, "GpuIpcScanExec: {}, projection={:?}", self.path, self.projection)
    }
}

impl GpuNode for GpuIpcScanExec {
    fn execute_device(&self, _partition: usize, _ctx: Arc<TaskContext>) -> Result<DeviceFuture> {
        let file = self.open()?;
        let projection = self.projection.clone();
        Ok(async move {
            blocking(move || {
                let mut parts = vec![];
                for i in 0..file.n_batches {
                    let key = [file.identity.as_slice(), format!("|ipc|{i}|{projection:?}").as_bytes()].concat();
                    if let Some(hit) = ScanCache::global().get(&key)? {
                        parts.push(hit); // a zero-copy view of HBM: no host read, no PCIe
                        continue;
                    }
                    let mut out = std::ptr::null_mut();
                    check(unsafe { sys::dfgpu_ipc_read_batch(file.handle, i, projection.as_ptr(), projection.len() as i32, &mut out) })?;
                    let t = DeviceTable(out);
                    ScanCache::global().put(&key, &t)?;
                    parts.push(t);
                }
                if parts.len() == 1 { Ok(parts.pop().unwrap()) } else { DeviceTable::concat(&parts) }
            }).await
        }
        .boxed())
    }
}

impl ExecutionPlan for GpuIpcScanExec {
    fn name(&self) -> &str { "GpuIpcScanExec" }
    fn properties(&self) -> &Arc<PlanProperties> { &self.cache }
    fn children(&self) -> Vec<&Arc<dyn ExecutionPlan>> { vec![] }
    fn apply_expressions(&self, _f: &mut dyn FnMut(&Arc<dyn PhysicalExpr>) -> Result<TreeNodeRecursion>) -> Result<TreeNodeRecursion> {
        Ok(TreeNodeRecursion::Continue)
    }
    fn replace_children(self: Arc<Self>, _c: Vec<Arc<dyn ExecutionPlan>>, _o: ReplaceChildrenOptions) -> Result<Arc<dyn ExecutionPlan>> { Ok(self) }
    #[allow(deprecated)]
    fn with_new_children(self: Arc<Self>, _c: Vec<Arc<dyn ExecutionPlan>>) -> Result<Arc<dyn ExecutionPlan>> { Ok(self) }
    fn execute(&self, partition: usize, ctx: Arc<TaskContext>) -> Result<SendableRecordBatchStream> {
        let batch_size = ctx.session_config().batch_size();
        Ok(host_stream(Arc::clone(&self.schema), self.execute_device(partition, ctx)?, batch_size))
    }
}

// ----------------------------------------------------------------------------------------------------------- Parquet
/// A local Parquet file scanned straight into HBM: the footer is read with the `parquet` crate (as the CPU reader does,
/// datasource-parquet/src/metadata.rs), every projected column chunk's byte range is described to the library, and ONE call —
/// `dfgpu_parquet_read_chunks` — decodes them on the library's own host threads (host half: page headers, decompression, levels,
/// run headers; device half: value decode), consults and fills the device chunk cache and puts the row groups together on the
/// device.  Python twin: datafusion_amd/parquet.py `ParquetFile.read`.  A chunk the library rejects (an encoding or codec outside
/// include/dfgpu.h's list, nested columns) fails the call with the library's message; the rule keeps the CPU `DataSourceExec` for files
/// whose footer shows such chunks.
#[derive(Debug)]
pub struct GpuParquetScanExec {
    path: String,
    /// leaf-column indices of the file's schema (the scan's projection), in output order
    projection: Vec<usize>,
    /// row groups to read (after statistics pruning by the rule / the dynamic filter), in file order
    row_groups: Vec<usize>,
    schema: SchemaRef,
    cache: Arc<PlanProperties>,
}

fn thrift_codec(c: datafusion::parquet::basic::Compression) -> i32 {
    use datafusion::parquet::basic::Compression::*;
    match c {
        UNCOMPRESSED => 0,
        SNAPPY => 1,
        GZIP(_) => 2,
        LZO => 3,
        BROTLI(_) => 4,
        LZ4 => 5,
        ZSTD(_) => 6,
        LZ4_RAW => 7,
    }
}

impl GpuParquetScanExec {
    pub fn new(path: String, projection: Vec<usize>, row_groups: Vec<usize>, schema: SchemaRef) -> Self {
        let props = PlanProperties::new(EquivalenceProperties::new(Arc::clone(&schema)), Partitioning::UnknownPartitioning(1), EmissionType::Final, Boundedness::Bounded);
        Self { path, projection, row_groups, schema, cache: Arc::new(props) }
    }
}

impl DisplayAs for GpuParquetScanExec {
    fn fmt_as(&self, _t: DisplayFormatType, f: &mut std::fmt::Formatter) -> std::fmt::Result {
        write!(f, "GpuParquetScanExec: {}, projection={:?}, row_groups={:?}", self.path, self.projection, self.row_groups)
    }
}

impl GpuNode for GpuParquetScanExec {
    fn execute_device(&self, _partition: usize, ctx: Arc<TaskContext>) -> Result<DeviceFuture> {
        use datafusion::parquet::file::metadata::ParquetMetaDataReader;
        let file = std::fs::File::open(&self.path)?;
        let meta = file.metadata()?;
        let map = unsafe { memmap2::Mmap::map(&file) }.map_err(|e| DataFusionError::External(Box::new(e)))?;
        let footer = ParquetMetaDataReader::new().parse_and_finish(&file).map_err(|e| DataFusionError::External(Box::new(e)))?;
        let mtime = meta.modified().ok().and_then(|t| t.duration_since(std::time::UNIX_EPOCH).ok()).map_or(0, |d| d.as_nanos());
        let identity = format!("{}|{}|{}", self.path, mtime, meta.len());
        let threads = ctx.session_config().target_partitions().clamp(1, 16) as i32;
        let (projection, row_groups, schema) = (self.projection.clone(), self.row_groups.clone(), Arc::clone(&self.schema));
        Ok(async move {
            blocking(move || {
                if row_groups.is_empty() {
                    return DeviceTable::empty(&schema);
                }
                // everything the chunk descriptors point at (names, cache keys) lives until the call returns
                let names: Vec<std::ffi::CString> = schema.fields().iter().map(|f| std::ffi::CString::new(f.name().as_str()).unwrap()).collect();
                let mut keys: Vec<Vec<u8>> = vec![];
                let mut chunks: Vec<sys::dfgpu_parquet_chunk> = vec![];
                for &g in &row_groups {
                    let rg = footer.row_group(g);
                    for (j, &leaf) in projection.iter().enumerate() {
                        let cc = rg.column(leaf);
                        let (start, len) = cc.byte_range(); // dictionary page first, then the data pages
                        let descr = cc.column_descr();
                        let field = crate::expr::field_of(schema.field(j).data_type())
                            .ok_or_else(|| DataFusionError::NotImplemented(format!("GPU scan of {:?}", schema.field(j).data_type())))?;
                        keys.push(format!("{identity}|{g}|{}", schema.field(j).name()).into_bytes());
                        chunks.push(sys::dfgpu_parquet_chunk {
                            bytes: unsafe { map.as_ptr().add(start as usize) },
                            n_bytes: len as i64,
                            column: sys::dfgpu_parquet_column {
                                physical_type: cc.column_type() as i32,
                                type_length: descr.type_length().max(0),
                                codec: thrift_codec(cc.compression()),
                                max_definition_level: descr.max_def_level() as i32,
                                max_repetition_level: descr.max_rep_level() as i32,
                                _pad: 0,
                                num_values: cc.num_values(),
                                field,
                                name: names[j].as_ptr(),
                            },
                            cache_key: std::ptr::null(),
                            cache_key_bytes: 0,
                        });
                    }
                }
                for (c, k) in chunks.iter_mut().zip(&keys) {
                    c.cache_key = k.as_ptr() as *const _;
                    c.cache_key_bytes = k.len() as i64;
                }
                let (mut out, mut hits) = (std::ptr::null_mut(), 0i64);
                check(unsafe {
                    sys::dfgpu_parquet_read_chunks(chunks.as_ptr(), row_groups.len() as i32, projection.len() as i32, threads, ScanCache::global().0, &mut out, &mut hits)
                })?;
                Ok(DeviceTable(out))
            })
            .await
        }
        .boxed())
    }
}

impl ExecutionPlan for GpuParquetScanExec {
    fn name(&self) -> &str { "GpuParquetScanExec" }
    fn properties(&self) -> &Arc<PlanProperties> { &self.cache }
    fn children(&self) -> Vec<&Arc<dyn ExecutionPlan>> { vec![] }
    fn apply_expressions(&self, _f: &mut dyn FnMut(&Arc<dyn PhysicalExpr>) -> Result<TreeNodeRecursion>) -> Result<TreeNodeRecursion> {
        Ok(TreeNodeRecursion::Continue)
    }
    fn replace_children(self: Arc<Self>, _c: Vec<Arc<dyn ExecutionPlan>>, _o: ReplaceChildrenOptions) -> Result<Arc<dyn ExecutionPlan>> { Ok(self) }
    #[allow(deprecated)]
    fn with_new_children(self: Arc<Self>, _c: Vec<Arc<dyn ExecutionPlan>>) -> Result<Arc<dyn ExecutionPlan>> { Ok(self) }
    fn execute(&self, partition: usize, ctx: Arc<TaskContext>) -> Result<SendableRecordBatchStream> {
        let batch_size = ctx.session_config().batch_size();
        Ok(host_stream(Arc::clone(&self.schema), self.execute_device(partition, ctx)?, batch_size))
    }
}

// ------------------------------------------------------------------------------------------ rule-side conversions
/// `DataSourceExec` over ONE local, unfiltered file -> the GPU scan node that reads it, or None (the CPU scan stays): an Arrow IPC file
/// (`ArrowSource`) or a Parquet file (`ParquetSource` without a pushed-down predicate; every projected column of a type the library
/// decodes).  A scan with several files / partitions, partition columns, a limit or a byte range per file keeps the CPU reader —
/// the node above it then uploads its batches (device.rs `device_input`).
pub fn try_from_data_source(d: &datafusion::datasource::source::DataSourceExec) -> Option<Arc<dyn ExecutionPlan>> {
    use datafusion::datasource::physical_plan::{ArrowSource, FileScanConfig, ParquetSource};
    let conf = d.data_source().downcast_ref::<FileScanConfig>()?;
    if conf.file_groups.len() != 1 || conf.file_groups[0].len() != 1 || conf.limit.is_some() || !conf.table_partition_cols().is_empty() {
        return None;
    }
    let file = &conf.file_groups[0].files()[0];
    if file.range.is_some() || conf.object_store_url.as_str() != "file:///" {
        return None;
    }
    let path = format!("/{}", file.object_meta.location);
    let schema = conf.projected_schema().ok()?;
    if !schema.fields().iter().all(|f| crate::expr::field_of(f.data_type()).is_some()) {
        return None;
    }
    let projection: Vec<usize> = conf.file_column_projection_indices().unwrap_or_else(|| (0..conf.file_schema().fields().len()).collect());
    if conf.file_source().downcast_ref::<ArrowSource>().is_some() {
        return Some(Arc::new(GpuIpcScanExec::new(path, projection.iter().map(|&i| i as i32).collect(), schema)));
    }
    let pq = conf.file_source().downcast_ref::<ParquetSource>()?;
    if pq.predicate().is_some() {
        return None; // row-group / page pruning by a pushed-down predicate stays with the CPU reader (a dynamic join filter arrives this way)
    }
    // all row groups: the footer is read when the node executes (flat schema: leaf index = field index)
    let file_handle = std::fs::File::open(&path).ok()?;
    let footer = datafusion::parquet::file::metadata::ParquetMetaDataReader::new().parse_and_finish(&file_handle).ok()?;
    if footer.file_metadata().schema_descr().num_columns() != conf.file_schema().fields().len() {
        return None; // nested columns: leaves and fields do not line up
    }
    let row_groups: Vec<usize> = (0..footer.num_row_groups()).collect();
    Some(Arc::new(GpuParquetScanExec::new(path, projection, row_groups, schema)))
}
