//! `dfgpu_table_t` as an owning Rust value; RecordBatch <-> device table over the Arrow C Data Interface — the same structs the
//! reference's own FFI streams carry (datafusion/ffi/src/record_batch_stream.rs:105-114) — and device table <-> Arrow C Device
//! array for consumers that stay on the GPU.  Python twin: datafusion_amd/table.py.
use crate::{check, sys};
use arrow::array::{new_empty_array, Array, ArrayRef, RecordBatch, RecordBatchOptions, StructArray};
use arrow::datatypes::SchemaRef;
use arrow::ffi::{from_ffi, to_ffi, FFI_ArrowArray, FFI_ArrowSchema};
use datafusion::error::Result;

pub struct DeviceTable(pub(crate) sys::dfgpu_table_t);
unsafe impl Send for DeviceTable {} // every handle carries its device; entry points switch the calling thread to it
unsafe impl Sync for DeviceTable {} // read-only entry points (probe, export, column views) may run concurrently on one handle

impl Drop for DeviceTable {
    fn drop(&mut self) {
        unsafe { sys::dfgpu_table_free(self.0) };
    }
}

impl DeviceTable {
    /// one RecordBatch -> HBM (dfgpu_table_import pins large host buffers and copies on a side stream: 43.6 GB/s measured).
    /// Utf8 / LargeUtf8 / Utf8View columns arrive as DFGPU_UTF8 (offsets + bytes); dictionary arrays as their indices.
    pub fn from_batch(batch: &RecordBatch) -> Result<Self> {
        let (mut a, mut s): (FFI_ArrowArray, FFI_ArrowSchema) = to_ffi(&StructArray::from(batch.clone()).into_data())?;
        let mut out = std::ptr::null_mut();
        check(unsafe { sys::dfgpu_table_import(&mut a, &mut s, &mut out) })?; // consumes both structs
        Ok(Self(out))
    }

    /// a 0-row table of `schema` (an exhausted input partition)
    pub fn empty(schema: &SchemaRef) -> Result<Self> {
        let columns: Vec<ArrayRef> = schema.fields().iter().map(|f| new_empty_array(f.data_type())).collect();
        Self::from_batch(&RecordBatch::try_new_with_options(schema.clone(), columns, &RecordBatchOptions::new().with_row_count(Some(0)))?)
    }

    /// a stream's batches -> one device table (collect_left_input's concat_batches, hash_join/exec.rs:2705)
    pub fn concat(parts: &[DeviceTable]) -> Result<Self> {
        let handles: Vec<_> = parts.iter().map(|t| t.0).collect();
        let mut out = std::ptr::null_mut();
        check(unsafe { sys::dfgpu_table_concat(handles.as_ptr(), handles.len() as i32, &mut out) })?;
        Ok(Self(out))
    }

    /// a second owner of the same HBM buffers (dfgpu_table_retain): what a node keeps when several consumers take its output
    pub fn retain(&self) -> Result<Self> {
        let mut out = std::ptr::null_mut();
        check(unsafe { sys::dfgpu_table_retain(self.0, &mut out) })?;
        Ok(Self(out))
    }

    /// zero-copy column subset / reorder (RecordBatch::project)
    pub fn select(&self, columns: &[i32]) -> Result<Self> {
        let mut out = std::ptr::null_mut();
        check(unsafe { sys::dfgpu_table_select(self.0, columns.as_ptr(), columns.len() as i32, &mut out) })?;
        Ok(Self(out))
    }

    pub fn num_rows(&self) -> Result<i64> {
        let mut n = 0i64;
        check(unsafe { sys::dfgpu_table_num_rows(self.0, &mut n) })?;
        Ok(n)
    }

    /// rows [offset, offset + length) as one RecordBatch in pinned host memory: what `poll_next` of a GPU node's stream
    /// yields, `batch_size` rows at a time (LimitedBatchCoalescer's fixed target, physical-plan/src/coalesce/mod.rs:27-120)
    pub fn export_batch(&self, offset: i64, length: i64, schema: &SchemaRef) -> Result<RecordBatch> {
        let (mut a, mut s) = (FFI_ArrowArray::empty(), FFI_ArrowSchema::empty());
        check(unsafe { sys::dfgpu_table_export_batch(self.0, offset, length, &mut a, &mut s) })?;
        let data = unsafe { from_ffi(a, &s) }?;
        let batch = RecordBatch::from(StructArray::from(data));
        // the library names columns as the input did and reports every column nullable; the node's declared schema is authoritative
        // for names, nullability and metadata (Utf8 columns come back as Utf8 / LargeUtf8: cast where the declared type is a view)
        let columns = batch.columns().iter().zip(schema.fields()).map(|(c, f)| {
            if c.data_type() == f.data_type() { Ok(c.clone()) } else { arrow::compute::cast(c, f.data_type()) }
        }).collect::<std::result::Result<Vec<_>, _>>()?;
        Ok(RecordBatch::try_new_with_options(schema.clone(), columns, &RecordBatchOptions::new().with_row_count(Some(length as usize)))?)
    }

    /// the table as an Arrow C Device array over its own HBM buffers (ARROW_DEVICE_ROCM): for a consumer that stays on the GPU.
    /// The array owns a reference on the buffers; `self` may be dropped.
    pub fn export_device(&self) -> Result<(sys::ArrowDeviceArray, FFI_ArrowSchema)> {
        let mut a = sys::ArrowDeviceArray { array: FFI_ArrowArray::empty(), device_id: 0, device_type: 0, sync_event: std::ptr::null_mut(), reserved: [0; 3] };
        let mut s = FFI_ArrowSchema::empty();
        check(unsafe { sys::dfgpu_table_export_device(self.0, &mut a, &mut s) })?;
        Ok((a, s))
    }

    /// the inverse, zero-copy (an array of this library comes back as the same buffers; another ROCm producer's array is wrapped)
    pub fn import_device(mut a: sys::ArrowDeviceArray, mut s: FFI_ArrowSchema) -> Result<Self> {
        let mut out = std::ptr::null_mut();
        check(unsafe { sys::dfgpu_table_import_device(&mut a, &mut s, &mut out) })?; // consumes both
        Ok(Self(out))
    }

    /// string key columns -> Int32 dictionary indices interned on the device (joins, GROUP BY, ORDER BY, repartition on strings)
    pub fn dictionary_encode(&self, column: usize, sorted: bool) -> Result<Self> {
        let mut out = std::ptr::null_mut();
        check(unsafe { sys::dfgpu_table_dictionary_encode(self.0, column as i32, sorted as i32, &mut out) })?;
        Ok(Self(out))
    }
}

/// MemoryReservation twin (execution/src/memory_pool/mod.rs:188): admission before an operator allocates
pub struct Reservation(sys::dfgpu_reservation_t);
unsafe impl Send for Reservation {}
impl Reservation {
    pub fn try_new(bytes: i64) -> Result<Self> {
        let mut r = std::ptr::null_mut();
        check(unsafe { sys::dfgpu_mem_try_reserve(bytes, &mut r) })?; // "Resources exhausted: ..." -> the rule keeps the CPU operator
        Ok(Self(r))
    }
}
impl Drop for Reservation {
    fn drop(&mut self) {
        unsafe { sys::dfgpu_mem_release(self.0) };
    }
}
