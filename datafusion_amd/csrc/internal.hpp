// internal.hpp — host-side plumbing shared by every translation unit of libdfgpu.so:
// error channel, HBM pool allocator, stream, launch/profiling helpers, Column/Table.
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/dfgpu.h"

namespace dfgpu {

struct Error : std::runtime_error {
  using std::runtime_error::runtime_error;
};

void set_last_error(const std::string& msg);

#define DFGPU_HIP(expr)                                                                         \
  do {                                                                                          \
    hipError_t _e = (expr);                                                                     \
    if (_e != hipSuccess)                                                                       \
      throw ::dfgpu::Error(std::string("HIP error ") + hipGetErrorString(_e) + " at " #expr);  \
  } while (0)

#define DFGPU_CHECK(cond, msg)                     \
  do {                                             \
    if (!(cond)) throw ::dfgpu::Error(std::string(msg)); \
  } while (0)

// the calling thread's operator metrics (dfgpu_metrics)
dfgpu_metrics& thread_metrics();
struct MetricsTimer {
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  ~MetricsTimer() {
    dfgpu_metrics& m = thread_metrics();
    m.calls++;
    m.elapsed_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
  }
};

// end of every C-ABI call (runtime.hip): when several host threads drive the library, the calling thread's streams are drained
// and the blocks it freed become everybody's
void call_epilogue() noexcept;

// Wraps a C-ABI entry point body: exceptions -> error code + thread-local message.
template <typename F>
int guarded(F&& f) noexcept {
  MetricsTimer timer;
  int rc = 0;
  try {
    f();
  } catch (const std::exception& e) {
    set_last_error(e.what());
    rc = 1;
  } catch (...) {
    set_last_error("unknown error");
    rc = 1;
  }
  call_epilogue();
  return rc;
}

// ---------------------------------------------------------------------------------------
// Options (dfgpu_set_option, include/dfgpu.h): the dispatch policy's numbers and switches in ONE table — the library's twin of the
// reference's ConfigOptions (common/src/config.rs: `datafusion.execution.*`).  Defaults are DERIVED FROM THE DEVICE at dfgpu_init
// (hipDeviceProp: CU count, L2 size of an XCD, LDS per CU), not tuned on a benchmark's shapes; an embedding engine (or a test that
// must force a path on a small input) overrides them by name.  Two environment variables remain, both for debugging:
// DFGPU_OPTIONS="name=value,..." (the same table, read at every lookup so a test may set it) and DFGPU_TRACE="agg,join,scan,dict".
struct Policy {
  int num_cus = 256;
  size_t lds_per_cu = 160 * 1024;
  size_t l2_bytes = 4 << 20;          // of ONE XCD
  int xcds = 8;
  // rows from which a pass that has fixed costs of its own (a specialised kernel's launch and look-up, an extra move of the rows, a
  // second kernel) repays them: every CU gets 16 Ki rows — 64 rows per lane of a 256-lane workgroup, 4 Mi rows on 256 CUs
  int64_t rows_worth_a_pass() const { return (int64_t)num_cus * 16384; }
  // a table / accumulator set beyond this is "beyond the caches": random access into it leaves an XCD's L2 (4 x its size: the share
  // of the 256 MiB Infinity Cache a kernel's other streams leave a table, measured in profiles/r3_random_access.md)
  size_t beyond_cache_bytes() const { return l2_bytes * 4; }
};
const Policy& policy();                                        // of the calling thread's current device
int64_t option_int(const char* name, int64_t dflt);            // the named option, else `dflt`
bool option_on(const char* name, bool dflt);                   // "0" / "off" / "false" = off
std::string option_str(const char* name, const char* dflt);
bool trace_on(const char* what);                               // DFGPU_TRACE names `what` (or "all")

// ---------------------------------------------------------------------------------------
// Runtime: one per initialised device.  A process normally drives ONE GPU (one process per GPU under torchrun); a
// single DataFusion process that owns several GPUs (one per output partition) initialises several and selects the
// calling thread's device with dfgpu_set_device — every entry point that takes a table / join / aggregate handle
// switches the calling thread to the handle's device first (unwrap()).
struct Runtime;
// `r.stream` = the CALLING THREAD's stream on device r (handles usable concurrently from different threads).  The thread
// that initialised the device works on the device's first stream; every other host thread gets a stream of its own on first use
// (taken from / returned to a per-device list when threads come and go: a scan's decode threads, a plan's partition threads), so
// their kernels and copies overlap.  What keeps that safe is the rule of call_epilogue(): once a second thread has appeared,
// every C-ABI call drains its thread's stream before it returns — whatever a handle holds is complete when another thread
// gets to see it — and the blocks a thread frees stay its own until then.  A process with one thread keeps the old behaviour:
// calls return with their work enqueued, freed blocks are reusable at once in stream order.
struct StreamRef {
  Runtime* r = nullptr;
  operator hipStream_t() const;
};
struct Runtime {
  int device = -1;
  StreamRef stream;                 // the calling thread's stream (see above)
  hipStream_t first_stream = nullptr;
  uint64_t generation = 0;          // bumped by dfgpu_init / dfgpu_shutdown: threads re-resolve the streams they remembered
  bool initialised = false;
  int num_cus = 256;
  Policy policy;
  hipStream_t thread_stream();      // what `stream` converts to

  // pool allocator: size-bucketed free lists; blocks are reused in stream order (a block freed by the host after its last kernel was
  // enqueued can be handed to the next kernel of the SAME stream safely; with several threads a freed block waits in its thread's
  // pending list until the end of the call).
  std::mutex mu;
  std::multimap<size_t, void*> free_blocks;
  std::map<void*, size_t> live;  // ptr -> capacity
  int64_t in_use = 0, cached = 0, peak = 0;
  std::atomic<int64_t> driver_allocs{0}, driver_alloc_ns{0};   // pool misses: hipMalloc calls and the host time they took

  void* alloc(size_t bytes);
  void free(void* p);
  void trim();

  // profiling
  bool profiling = false;
  struct Rec {
    std::string name;
    hipEvent_t a, b;
    int64_t bytes;
    std::thread::id thread;  // who launched it (dfgpu_metrics.kernel_ns is per thread)
  };
  std::map<std::thread::id, int64_t> kernel_ns_by_thread;  // filled by collect(), drained by dfgpu_metrics_get
  std::vector<Rec> recs;
  std::vector<dfgpu_kernel_stat> stats;  // aggregated by collect()
  struct Launch { int stat; float ms; int64_t bytes; };
  std::vector<Launch> launches;          // every collected launch, in order (dfgpu_profile_launches); bounded
  void collect();
};

Runtime& rt();                 // the calling thread's current device runtime (hipSetDevice is kept in step, per thread)
void scan_upload_caches_drop(int device);      // parquet.hip: the scan workers' cached upload blocks of `device` (-1: all) go back to the pool
int64_t scan_upload_caches_bytes(int device);  // what they hold right now
Runtime& rt_of(int device);    // an initialised device's runtime
void use_device(int device);   // make `device` the calling thread's current device (must be initialised)
int current_device();          // -1 before dfgpu_init
const std::vector<int>& initialised_devices();
void require_init();

// RAII event pair around a kernel launch when profiling is on.
struct ProfileScope {
  hipEvent_t a = nullptr, b = nullptr;
  const char* name;
  int64_t bytes;
  ProfileScope(const char* name, int64_t algorithmic_bytes);
  ~ProfileScope();
};

// Refcounted device buffer from the pool.
struct DevBuf {
  void* ptr = nullptr;
  size_t bytes = 0;
  Runtime* owner = nullptr;  // the pool the block goes back to (buffers may be released from any thread)
  std::shared_ptr<void> foreign;  // device memory somebody else owns (an imported ArrowDeviceArray): released with the last buffer that refers to it
  // One remembered answer about the (immutable) contents, shared by every column that views this buffer: bits 63..2 = a tag of the
  // question (what was asked, of which rows), bits 1..0 = 2 | answer.  The join's "are these probe keys clustered?" sample lives here:
  // asked once per table instead of once per probe (a kernel and a blocking read-back each time).
  std::atomic<uint64_t> hint{0};
  explicit DevBuf(size_t n);
  DevBuf(void* p, size_t n, std::shared_ptr<void> keep_alive) : ptr(p), bytes(n), foreign(std::move(keep_alive)) {}
  ~DevBuf();
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  template <typename T>
  T* as() const { return reinterpret_cast<T*>(ptr); }
};
using BufPtr = std::shared_ptr<DevBuf>;
inline BufPtr make_buf(size_t bytes) { return std::make_shared<DevBuf>(bytes ? bytes : 1); }
BufPtr make_zero_buf(size_t bytes);

// scratch helper for small host<->device scalars
void d2h(void* dst, const void* src, size_t n);  // synchronous w.r.t. the library stream
// Pinned host memory from the process-wide pool of the export path (table.hip).  A copy between HBM and PAGEABLE memory beyond a
// few hundred KB makes the driver pin the caller's pages for the copy; freeing such memory afterwards invalidates a range the kernel
// driver still tracks, which evicts the process's queues until a worker restores them — measured as 14-23 ms of dead time at the
// start of the NEXT call (profiles/r3_strings.md).  Downloads the library itself consumes go through these blocks instead.
extern std::atomic<int64_t> g_pinned_driver_allocs, g_pinned_driver_ns;   // hipHostMalloc calls of the pinned pool and their host time
void* pinned_alloc(size_t n);
void pinned_release(void* p);
// std::vector over such blocks for host-side staging that is uploaded and dropped (the Parquet chunk plans); plain memory in a
// process without a GPU
void* stage_alloc(size_t n);
template <class T> struct StageAllocator {
  using value_type = T;
  StageAllocator() = default;
  template <class U> StageAllocator(const StageAllocator<U>&) {}
  T* allocate(size_t n) { return static_cast<T*>(stage_alloc(n * sizeof(T))); }
  void deallocate(T* p, size_t) { pinned_release(p); }
  // resize() leaves new elements uninitialised (they are about to be overwritten by a page body; zero-filling a staging buffer
  // first doubles the host's memory traffic)
  template <class U> void construct(U* p) noexcept { ::new ((void*)p) U; }
  template <class U, class... A> void construct(U* p, A&&... a) { ::new ((void*)p) U(std::forward<A>(a)...); }
  template <class U> bool operator==(const StageAllocator<U>&) const { return true; }
  template <class U> bool operator!=(const StageAllocator<U>&) const { return false; }
};
template <class T> using StageVec = std::vector<T, StageAllocator<T>>;
struct PinnedBuf {
  void* ptr = nullptr;
  size_t bytes = 0;
  explicit PinnedBuf(size_t n) : ptr(pinned_alloc(n ? n : 1)), bytes(n) {}
  ~PinnedBuf() { pinned_release(ptr); }
  PinnedBuf(const PinnedBuf&) = delete;
  PinnedBuf& operator=(const PinnedBuf&) = delete;
  template <class T> T* as() const { return reinterpret_cast<T*>(ptr); }
};
void h2d_async(void* dst, const void* src, size_t n);

// ---------------------------------------------------------------------------------------
inline int type_width(int t) {
  switch (t) {
    case DFGPU_INT32: case DFGPU_UINT32: case DFGPU_DATE32: return 4;
    case DFGPU_INT64: case DFGPU_UINT64: case DFGPU_FLOAT64: return 8;
    case DFGPU_DECIMAL128: return 16;
    case DFGPU_UINT8: return 1;
    case DFGPU_BOOL: return 0;  // bit-packed
    case DFGPU_UTF8:
      throw Error("a Utf8 column reached an operator that works on fixed-width columns: dictionary-encode it first (dfgpu_table_dictionary_encode)");
  }
  throw Error("unknown dfgpu_type " + std::to_string(t));
}
inline size_t bitmap_bytes(int64_t n) { return (size_t)((n + 63) / 64) * 8; }  // padded to 64-bit words
inline size_t data_bytes(int t, int64_t n) { return t == DFGPU_BOOL ? bitmap_bytes(n) : (size_t)n * type_width(t); }
inline bool is_integer_like(int t) {
  return t == DFGPU_INT32 || t == DFGPU_INT64 || t == DFGPU_UINT8 || t == DFGPU_UINT32 || t == DFGPU_UINT64 || t == DFGPU_DATE32;
}
inline bool is_signed_type(int t) { return t == DFGPU_INT32 || t == DFGPU_INT64 || t == DFGPU_DATE32 || t == DFGPU_DECIMAL128; }
std::string type_name(const dfgpu_field& f);

// cached integer-column statistics (dfgpu_column_minmax): the device-table twin of the scan statistics the reference
// keeps per column (ColumnStatistics min_value / max_value, common/src/stats.rs).  Shared by zero-copy views of the
// same rows (select / hstack); tables are immutable, so the cache never goes stale.
struct ColStats {
  long long min = 0, max = 0;
  int64_t valid = 0;
  bool ascending = false;      // strictly ascending in row order, no NULLs
  bool nondecreasing = false;  // key[i-1] <= key[i] for every row, no NULLs: equal keys are adjacent (ordered input of an aggregate)
  bool has_present = false;    // UInt8 columns, on request (aggregate.hip u8_presence): which of the 256 values occur
  uint64_t present[4] = {0, 0, 0, 0};
};

// Values of a dictionary-encoded column (Arrow Dictionary(index type, Utf8 | LargeUtf8)): the device column holds the
// indices as a plain integer column — every kernel treats them as integers — and the strings stay on the host,
// attached to the column and handed on to columns derived from it by selecting / reordering rows (filter, take, join
// payload, group keys, partitions).  This is how TPC-H's low-cardinality string columns (l_returnflag, c_mktsegment ...)
// cross the boundary (the dictionary part: group_values/multi_group_by/dictionary.rs).
struct DictValues {
  std::string index_format;         // Arrow format of the indices as imported ("C", "i", "I", "l", "L")
  std::string value_format;         // "u" (Utf8) or "U" (LargeUtf8)
  std::vector<std::string> values;
  std::vector<uint8_t> valid;       // per value: 1 = not NULL
  bool sorted = false;              // values strictly ascending: index order == string order
};

// two dictionary-encoded columns mean the same strings by the same indices
inline bool same_dictionary(const std::shared_ptr<const DictValues>& a, const std::shared_ptr<const DictValues>& b) {
  return a == b || (a && b && a->values == b->values && a->valid == b->valid);
}

struct Column {
  dfgpu_field field{};
  std::string name;
  int64_t length = 0;
  int64_t null_count = 0;  // -1 = unknown (validity present, not counted)
  BufPtr data;             // values (or bit-packed booleans), 64-bit word padded
  size_t data_offset = 0;  // byte offset of row 0 inside `data` (partition outputs share one buffer)
  BufPtr validity;         // optional bitmap, 64-bit word padded; nullptr = all valid
  BufPtr offsets;          // DFGPU_UTF8: int64 [length + 1] byte offsets into `data` (offsets[0] == 0)
  std::shared_ptr<ColStats> stats;  // filled lazily by dfgpu_column_minmax; never set on columns whose rows differ from the source
  std::shared_ptr<const DictValues> dict;  // dictionary-encoded strings: `data` holds the indices

  const void* ptr() const { return data ? (const char*)data->ptr + data_offset : nullptr; }
  const uint64_t* valid_words() const { return validity ? validity->as<uint64_t>() : nullptr; }
  bool has_nulls() const { return validity != nullptr; }
};

struct Table {
  std::vector<Column> cols;
  int64_t nrows = 0;
  int device = current_device();  // the GPU whose HBM holds the columns
};

// every entry point reaches its input tables through unwrap(): the calling thread is switched to the table's device
// (HIP's current device is per thread — a worker thread of the host engine starts on device 0)
inline Table* unwrap(dfgpu_table_t t) {
  DFGPU_CHECK(t != nullptr, "null table handle");
  Table* p = reinterpret_cast<Table*>(t);
  if (p->device >= 0) use_device(p->device);
  thread_metrics().rows_in += p->nrows;
  return p;
}
// the same for entry points that only look at a table (row counts, column views, zero-copy selections): no rows are consumed
inline Table* unwrap_quiet(dfgpu_table_t t) {
  Table* p = unwrap(t);
  thread_metrics().rows_in -= p->nrows;
  return p;
}
inline dfgpu_table_t wrap(Table* t) {
  thread_metrics().rows_out += t->nrows;
  return reinterpret_cast<dfgpu_table_t>(t);
}

Column alloc_column(const dfgpu_field& f, const std::string& name, int64_t n, bool with_validity = false);
// a fresh column for rows taken from `src` (same type and name, dictionary handed on)
inline Column alloc_like(const Column& src, int64_t n, bool with_validity = false) {
  Column c = alloc_column(src.field, src.name, n, with_validity);
  c.dict = src.dict;
  return c;
}

inline dfgpu_table_t wrap_quiet(Table* t) { return reinterpret_cast<dfgpu_table_t>(t); }  // zero-copy views: no rows produced

// ----------------------------------------------------------------- primitives (scan.hip)
// exclusive prefix sum of popcount(mask_word & valid_word) per 64-row word -> u64 offsets
// (ceil(nrows/64) + 1 entries; the last is the total). valid may be null.
void scan_mask_popcounts(const uint64_t* mask, const uint64_t* valid, int64_t nrows, uint64_t* out_prefix);
// exclusive prefix sum of u32 counts -> u64 (n + 1 entries)
void scan_u32(const uint32_t* in, int64_t n, uint64_t* out_prefix);
void scan_bitmap_words_tab(const uint64_t* words, int64_t n_words, uint64_t* out_prefix, void* out_tab);   // + the rank map's interleaved {word, prefix} pairs
uint64_t read_u64(const uint64_t* dev);

// ----------------------------------------------------------------- compaction (filter.hip)
// out = rows of `in` whose mask bit (and mask_valid bit) is set, order preserved.
Table compact_table(const Table& in, const std::vector<int>& cols, const uint64_t* mask, const uint64_t* mask_valid);
// take: out[i] = in[idx[i]] ; idx < 0 -> NULL.  idx is a device array of int64.
Column gather_column(const Column& in, const int64_t* idx, int64_t n, bool idx_may_be_null);
// take of several columns of one table by the same ids: row-major records + ONE random access per row when that pays
// (idx32: the same ids as 32-bit values instead of `idx` — the sort hands its row ids over without widening them)
std::vector<Column> gather_columns(const Table& in, const std::vector<int>& cols, const int64_t* idx, int64_t n, bool idx_may_be_null, const uint32_t* idx32 = nullptr);
// one row-major record per row holding every column of `cols` (records.hpp): layout planning and the take from records laid
// out by the caller (sort.hip's carried sort)
struct PackLayout;
bool plan_record_layout(const Table& in, const std::vector<int>& cols, PackLayout& L, int& R, std::vector<int>& order);
// a Boolean column as one UInt8 per row (validity kept): Boolean key columns of joins, repartitions, sorts and aggregates (aggregate.hip)
Column bool_as_u8(const Column& c, int64_t n);
// out[w] = a[w] & b[w] over nw 64-bit words (aggregate.hip)
void and_bitmaps(const uint64_t* a, const uint64_t* b, int64_t nw, uint64_t* out);
// clear bits beyond n in the last word of a bitmap (keeps padding deterministic)
void count_nulls(Column& c);

// ----------------------------------------------------------------- expressions (expr.hip)
struct Datum {  // ColumnarValue: array or scalar
  Column col;   // for scalars: length-1 column resident on device? no: host literal below
  std::string str;  // scalar of type DFGPU_UTF8: the literal's bytes
  bool scalar = false;
  bool scalar_null = false;
  uint64_t lit_lo = 0, lit_hi = 0;  // scalar bits (sign-extended ints / f64 bits)
};
dfgpu_field expr_type(const dfgpu_expr& e, const Table& input);
Datum evaluate(const dfgpu_expr& e, const Table& input);
Column datum_to_column(const Datum& d, int64_t n, const std::string& name);

// ----------------------------------------------------------------- strings (strings.hip)
inline const int64_t* str_offsets(const Column& c) { return c.offsets ? c.offsets->as<int64_t>() : nullptr; }
// a fresh DFGPU_UTF8 column: offsets allocated (n + 1), bytes allocated by the caller once their total is known
Column alloc_string_column(const Column& like, int64_t n);
// take: out[i] = in[idx[i]]; idx < 0 -> NULL
Column gather_strings(const Column& in, const int64_t* idx, int64_t n, bool idx_may_be_null);
// substr(column, start, count) over a Utf8 column (device) or a dictionary-encoded one (its dictionary values, on the host)
Column substr_column(const Column& in, int64_t start, bool has_count, int64_t count);
// rows of `in` whose mask bit is set (prefix = exclusive popcount prefix per mask word, n_out = total)
Column compact_strings(const Column& in, const uint64_t* mask, const uint64_t* mask_valid, const uint64_t* prefix, int64_t nrows, int64_t n_out);
// rows [offset, offset + length) of a string column (offsets rebased to 0)
Column slice_strings(const Column& in, int64_t offset, int64_t length);
// vertical concatenation of string columns (all DFGPU_UTF8)
Column concat_strings(const std::vector<const Column*>& parts, int64_t total);
// Int32 indices + host dictionary (first-seen order, or ascending when `sorted`)
Column dictionary_encode(const Column& in, bool sorted);
Column dictionary_decode(const Column& in);
Column string_hash_column(const Column& in);
// BinaryExpr comparison / LikeExpr over string operands (a Boolean column; NULL where an operand is NULL)
Datum string_binary(int op, const Datum& a, const Datum& b, int64_t nrows);

// ----------------------------------------------------------------- runtime specialisation (jit.hip)
// compiles `source` with hiprtc for gfx950 (cached per process by source text) and returns `kernel_name`
hipFunction_t jit_get(const std::string& source, const char* kernel_name);
// launches on the library stream; `args` is the kernel's single by-value argument struct
void jit_launch(hipFunction_t fn, int grid, int block, size_t lds_bytes, void* args, size_t args_bytes);

// ----------------------------------------------------------------- runtime specialisation (jit.hip)
// compiles `source` with hiprtc for gfx950 (cached per process by source text) and returns `kernel_name`
hipFunction_t jit_get(const std::string& source, const char* kernel_name);
// launches on the library stream; `args` is the kernel's single by-value argument struct
void jit_launch(hipFunction_t fn, int grid, int block, size_t lds_bytes, void* args, size_t args_bytes);

// ----------------------------------------------------------------- Arrow format strings (table.hip)
dfgpu_field parse_format(const char* fmt, bool nullable);
std::string format_of(const dfgpu_field& f);
// bytes of HBM a table's columns occupy (values, validity, offsets)
int64_t table_device_bytes(const Table& t);

// ----------------------------------------------------------------- dictionaries (table.hip)
// `c`'s indices rewritten so that they index `target`'s values; a value `target` does not hold gets the index
// target->values.size() (equal to no index of a column encoded with `target`) — what joining / comparing two
// dictionary-encoded columns with different dictionaries needs.  Errors when the index type cannot hold that value.
Column remap_to_dictionary(const Column& c, const std::shared_ptr<const DictValues>& target);
Column dictionary_like_column(const Column& c, const std::string& pattern, bool case_insensitive);  // Boolean column: value LIKE pattern

// dst bits [off, off + n) |= src bits [0, n) (src null = all ones); the destination range must start zeroed
void bitmap_place(const uint64_t* src, int64_t off, int64_t n, uint64_t* dst);

// ----------------------------------------------------------------- statistics (join.hip)
// min / max / non-null count / strictly-ascending flag of an integer column, cached on the column
ColStats column_stats(Column& c, int64_t nrows);

// ----------------------------------------------------------------- visited marks of a replicated build side (join.hip)
std::vector<uint8_t> join_visited_export(dfgpu_join_t h);                       // one bit per build row + 8 bytes of null-aware flags
void join_visited_merge(dfgpu_join_t h, const uint8_t* merged, size_t nbytes);  // OR into the table's marks

// ----------------------------------------------------------------- radix passes (sort.hip)
// (key u64, row id u32) pairs stably sorted by bits [lo_bit, lo_bit + nbits) of the key; `idx` null on entry = row id is the position
void radix_sort_pairs(BufPtr& key, BufPtr& idx, int64_t n, int lo_bit, int nbits);
// keys alone, grouped stably by bits [lo_bit, lo_bit + nbits) of their value (no row ids are made)
void radix_group_keys(BufPtr& key, int64_t n, int lo_bit, int nbits);
Table sort_table_ascending(const Table& in, const std::vector<int>& key_cols);   // sort.hip; stable
// partition.hip: the rows of fixed-width columns (no validity) moved into `nparts` (<= 64) contiguous groups by
// ((key - kmin) >> shift) & mask, row order kept inside a group (two calls, low digit first, order the rows by 12 bits);
// bounds[p] .. bounds[p + 1] = group p's rows in every moved column
struct RangePartition {
  std::vector<BufPtr> cols;
  std::vector<uint64_t> bounds;
  int64_t rows = 0;   // rows moved (fewer than the input's under a row mask)
};
RangePartition partition_by_key_range(const void* key, int key_type, int64_t n, long long kmin, int shift, unsigned mask, int nparts,
                                      const std::vector<const void*>& src, const std::vector<int>& widths, bool want_bounds = true,
                                      const uint64_t* row_mask = nullptr, const uint64_t* row_mask_valid = nullptr);   // (mask: one bit per row, 1 = takes part)

// ----------------------------------------------------------------- LDS radix join (radix_join.hip)
struct RadixTable;
std::shared_ptr<RadixTable> radix_join_build(const Table& build, const std::vector<int>& key_cols, bool null_equals_null, bool force_collisions);
int64_t radix_join_table_bytes(const RadixTable& t);
int radix_join_bits(const RadixTable& t);
// (build row, probe row) of every key-equal pair, in partition order; m pairs of int64
void radix_join_pairs(const RadixTable& t, const Table& build, const std::vector<int>& build_keys, const Table& probe, const std::vector<int>& probe_keys,
                      bool null_equals_null, bool force_collisions, BufPtr& out_b, BufPtr& out_p, int64_t& m);

// INNER join with the output columns written by the emit walk (fixed-width, non-nullable payload; exact record keys); false = not applicable
bool radix_join_inner_columns(const RadixTable& t, const Table& build, const std::vector<int>& build_keys, const Table& probe, const std::vector<int>& probe_keys,
                              const std::vector<int>& bout, const std::vector<int>& pout, Table& out);

// ----------------------------------------------------------------- hashing (partition.hip)
// RepartitionExec(Hash): nparts tables, row order kept inside each (slices of one buffer per column when nothing is nullable)
std::vector<Table> partition_table(const Table& in, const std::vector<int>& key_cols, int nparts);
void hash_columns(const std::vector<const Column*>& keys, int64_t n, uint64_t seed, uint64_t* out, bool force_collisions);

}  // namespace dfgpu
