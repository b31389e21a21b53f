//! `dfgpu_table_t` as an owning Rust value; RecordBatch <-> device table over the Arrow C Data Interface — the same structs the
//! reference's own FFI streams carry (datafusion/ffi/src/record_batch_stream.rs:105-114).  Python twin: datafusion_amd/table.py.
use crate::{check, sys};
use arrow::array::{Array, RecordBatch, StructArray};
use arrow::datatypes::SchemaRef;
use arrow::ffi::{from_ffi, to_ffi, FFI_ArrowArray, FFI_ArrowSchema};
use datafusion::error::Result;

pub struct DeviceTable(pub(crate) sys::dfgpu_table_t);
unsafe impl Send for DeviceTable {} // every handle carries its device; entry points switch the calling thread to it
unsafe impl Sync for DeviceTable {}

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

    /// a stream's batches -> one device table (collect_left_input's concat_batches, hash_join/exec.rs:2705)
    pub fn concat(parts: &[DeviceTable]) -> Result<Self> {
        let handles: Vec<_> = parts.iter().map(|t| t.0).collect();
        let mut out = std::ptr::null_mut();
        check(unsafe { sys::dfgpu_table_concat(handles.as_ptr(), handles.len() as i32, &mut out) })?;
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
        // the library names columns as the input did; the node's declared schema is authoritative for names / metadata
        Ok(batch.with_schema(schema.clone())?)
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
