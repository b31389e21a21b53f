/*
 * dfgpu.h — C ABI of libdfgpu.so: MI355X-native (gfx950) execution backend for
 * DataFusion's vectorized physical operators.
 *
 * This is the drop-in boundary.  The reference has no C-friendly ABI for
 * operators (datafusion-ffi is Rust<->Rust: stabby vectors + async-ffi wakers,
 * datafusion/ffi/README.md:61-74), so what a Rust `ExecutionPlan` shim binds is this
 * header; data crosses as Arrow C Data Interface structs, layout-identical to the
 * `FFI_ArrowArray`/`FFI_ArrowSchema` pair the reference's own FFI streams carry
 * (datafusion/ffi/src/arrow_wrappers.rs:31,72-75; record_batch_stream.rs:105-114).
 *
 * Each entry point cites the reference interface it replaces (paths relative to
 * /root/reference/datafusion/).  INTEGRATION.md shows the Rust-side binding.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; the message is in
 *     dfgpu_last_error() (thread-local), mirroring `Result<_, DataFusionError>`.
 *   - devices: dfgpu_init(ids, n) binds the process to one or several GPUs.  One process per GPU is
 *     the usual deployment (torchrun-style launchers, bench.py); a single DataFusion process that owns several
 *     GPUs — one per output partition of the plan — initialises them all and selects the calling thread's device
 *     with dfgpu_set_device (HIP's current device is per thread; the library keeps it in step on every entry).  Every
 *     handle (table, join table, aggregate) lives on the device that was current when it was created, and an entry
 *     point that takes a handle switches the calling thread to that handle's device first.
 *   - streams: every host thread works on a HIP stream of its own per device (the thread that called dfgpu_init on the device's
 *     first stream), so entry points working on different handles from different threads run concurrently on the device —
 *     `execute(partition)` of several partitions, the column chunks of a scan.  With ONE calling thread, calls are synchronous
 *     w.r.t. the results they return (row counts) and asynchronous otherwise; dfgpu_sync() drains the device.  Once a SECOND
 *     thread has called into the library, every call drains its thread's stream before it returns: what a handle holds is
 *     complete when another thread gets to see it, and the HBM blocks a call freed become reusable by other threads only then.
 *   - handles are owned by exactly one caller and freed exactly once (from any thread).  A join table (dfgpu_join_t) is
 *     immutable after build and may be probed by many callers at once (CollectLeft: one build shared by all probe partitions,
 *     hash_join/exec.rs:1503-1523); any other handle is not re-entrant — one caller at a time, as one stream per
 *     `execute(partition)` in the reference.  The allocator, the error channel (thread-local) and the metrics are thread-safe.
 *   - device columns are Arrow-layout buffers in HBM: fixed-width values, optional
 *     validity bitmap (LSB first, 1 = valid), Boolean columns bit-packed.
 */
#ifndef DFGPU_H
#define DFGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DFGPU_ABI_VERSION 13

/* Arrow C Data Interface (https://arrow.apache.org/docs/format/CDataInterface.html) */
#ifndef ARROW_C_DATA_INTERFACE
#define ARROW_C_DATA_INTERFACE
struct ArrowSchema {
  const char* format;
  const char* name;
  const char* metadata;
  int64_t flags;
  int64_t n_children;
  struct ArrowSchema** children;
  struct ArrowSchema* dictionary;
  void (*release)(struct ArrowSchema*);
  void* private_data;
};
struct ArrowArray {
  int64_t length;
  int64_t null_count;
  int64_t offset;
  int64_t n_buffers;
  int64_t n_children;
  const void** buffers;
  struct ArrowArray** children;
  struct ArrowArray* dictionary;
  void (*release)(struct ArrowArray*);
  void* private_data;
};
#endif

/* Arrow C Device Data Interface (https://arrow.apache.org/docs/format/CDeviceDataInterface.html): an ArrowArray whose buffers are
 * DEVICE pointers, tagged with the device that holds them.  This is how a device table crosses to (or from) another component of the
 * process WITHOUT touching the host: dfgpu_table_export_device / dfgpu_table_import_device below. */
#ifndef ARROW_C_DEVICE_DATA_INTERFACE
#define ARROW_C_DEVICE_DATA_INTERFACE
typedef int32_t ArrowDeviceType;
#define ARROW_DEVICE_CPU 1
#define ARROW_DEVICE_CUDA 2
#define ARROW_DEVICE_CUDA_HOST 3
#define ARROW_DEVICE_OPENCL 4
#define ARROW_DEVICE_VULKAN 7
#define ARROW_DEVICE_METAL 8
#define ARROW_DEVICE_VPI 9
#define ARROW_DEVICE_ROCM 10
#define ARROW_DEVICE_ROCM_HOST 11
#define ARROW_DEVICE_EXT_DEV 12
#define ARROW_DEVICE_CUDA_MANAGED 13
#define ARROW_DEVICE_ONEAPI 14
#define ARROW_DEVICE_WEBGPU 15
#define ARROW_DEVICE_HEXAGON 16
struct ArrowDeviceArray {
  struct ArrowArray array;
  int64_t device_id;
  ArrowDeviceType device_type;
  void* sync_event; /* ROCm: hipEvent_t* the consumer waits on before reading, or NULL = the data is ready */
  int64_t reserved[3];
};
#endif

/* ------------------------------------------------------------------ types */

typedef enum dfgpu_type {
  DFGPU_INT32 = 1,
  DFGPU_INT64 = 2,
  DFGPU_DECIMAL128 = 3, /* 16-byte LE two's complement + (precision, scale) */
  DFGPU_FLOAT64 = 4,
  DFGPU_UINT8 = 5, /* dictionary codes / packed 1-byte strings */
  DFGPU_UINT32 = 6,
  DFGPU_UINT64 = 7,
  DFGPU_DATE32 = 8, /* days since epoch, int32 */
  DFGPU_BOOL = 9,   /* bit-packed */
  /* variable-length strings in HBM (Arrow Utf8 / LargeUtf8 / Utf8View on import): `data` = the bytes, 64-bit offsets
   * [length + 1] beside them.  Row-selecting operators (filter, take, join payload, sort output, partitions, concat) move
   * such columns; `=`, `!=`, `<` ..., LIKE / ILIKE against a literal or another string column are evaluated on the bytes;
   * operators that hash or order by a string key run on interned indices (hash + byte comparison on the device — ArrowBytesMap,
   * physical-expr-common/src/binary_map.rs; group_values/{single,multi}_group_by/bytes*.rs): GROUP BY (dfgpu_agg_update; one
   * update per aggregate), ORDER BY (dfgpu_sort) and the joins (build and probe side, each against the other's dictionary) intern a
   * DFGPU_UTF8 key themselves and hand DFGPU_UTF8 back; dfgpu_table_dictionary_encode does it once for a column used many times.
   * Hash repartitioning (dfgpu_partition, dfgpu_exchange_hash) routes string keys — DFGPU_UTF8 or dictionary-encoded — on a hash
   * of their BYTES, so equal strings meet in one partition whatever table, dictionary, call or rank they come from. */
  DFGPU_UTF8 = 10
} dfgpu_type;

typedef struct dfgpu_field {
  int32_t type; /* dfgpu_type */
  int32_t precision;
  int32_t scale;
  int32_t nullable;
} dfgpu_field;

/* read-only view of one device column */
typedef struct dfgpu_column_view {
  dfgpu_field field;
  int64_t length;
  int64_t null_count;
  const void* data;        /* device pointer */
  const uint8_t* validity; /* device pointer or NULL */
  const char* name;        /* owned by the table */
  const int64_t* offsets;  /* DFGPU_UTF8: device pointer to length + 1 byte offsets into `data`; NULL otherwise */
} dfgpu_column_view;

typedef struct dfgpu_table_s* dfgpu_table_t;
typedef struct dfgpu_join_s* dfgpu_join_t;
typedef struct dfgpu_agg_s* dfgpu_agg_t;

/* JoinType (common/src/join_type.rs) — same order as the reference enum */
typedef enum dfgpu_join_type {
  DFGPU_JOIN_INNER = 0,
  DFGPU_JOIN_LEFT = 1,
  DFGPU_JOIN_RIGHT = 2,
  DFGPU_JOIN_FULL = 3,
  DFGPU_JOIN_LEFT_SEMI = 4,
  DFGPU_JOIN_RIGHT_SEMI = 5,
  DFGPU_JOIN_LEFT_ANTI = 6,
  DFGPU_JOIN_RIGHT_ANTI = 7,
  DFGPU_JOIN_LEFT_MARK = 8,
  DFGPU_JOIN_RIGHT_MARK = 9
} dfgpu_join_type;

/* NullEquality (common/src/null_equality.rs) */
typedef enum dfgpu_null_equality { DFGPU_NULL_EQUALS_NOTHING = 0, DFGPU_NULL_EQUALS_NULL = 1 } dfgpu_null_equality;

/* ------------------------------------------------------- lifecycle / errors */

int dfgpu_abi_version(void);
/* bind this process to the listed devices (idempotent per device) and create a library stream + memory pool on
 * each; the first device of the first call is the default device of every host thread.  Peer access between the
 * listed devices is enabled where the hardware allows it (xGMI). */
int dfgpu_init(const int* device_ids, int n_devices);
/* the calling thread's current device: entry points without a handle argument (generators, dfgpu_table_import,
 * dfgpu_table_alloc, dfgpu_parquet_decode_chunk, dfgpu_sync, dfgpu_mem_*) act on it */
int dfgpu_set_device(int device);
int dfgpu_get_device(int* out);
int dfgpu_shutdown(void);
int dfgpu_device_count(int* out);
/* thread-local message of the last failing call ("" if none) */
const char* dfgpu_last_error(void);
/* drain the library stream */
int dfgpu_sync(void);
/* the hipStream_t all kernels are launched on (as void*) */
void* dfgpu_stream(void);

/* pool statistics: bytes currently handed out / cached / high-water mark.
 * Counterpart of MemoryReservation accounting (execution/src/memory_pool/mod.rs:188). */
int dfgpu_mem_stats(int64_t* in_use, int64_t* cached, int64_t* peak);
int dfgpu_mem_trim(void);
/* Admission control = MemoryPool::try_grow / MemoryReservation (execution/src/memory_pool/mod.rs:188; the hash join's
 * build side charges one per batch, hash_join/exec.rs:2608).  An operator reserves what it is about to hold on the current
 * device; the call fails with the reference's "Resources exhausted: ..." message when tables + reservations + the request
 * exceed the limit (default 92 % of the device's memory) — the optimizer rule then keeps the CPU operator, which can spill
 * (the spill-aware fallback of SURVEY §8f N4).  A reservation does not allocate; release it when the operator is done. */
typedef struct dfgpu_reservation_s* dfgpu_reservation_t;
int dfgpu_mem_set_limit(int64_t bytes); /* 0 = back to the default */
int dfgpu_mem_limit(int64_t* limit, int64_t* reserved);
int dfgpu_mem_try_reserve(int64_t bytes, dfgpu_reservation_t* out);
int dfgpu_mem_reservation_size(dfgpu_reservation_t r, int64_t* out);
int dfgpu_mem_release(dfgpu_reservation_t r);

/* ------------------------------------------------------------------ tables */

/* Import a RecordBatch (a struct array, as exported by arrow-rs `to_ffi` /
 * pyarrow `_export_to_c`): host buffers are copied to HBM with async H2D copies.
 * Consumes `array` and `schema` (calls their release callbacks).
 * Replaces: the input side of ExecutionPlan::execute (physical-plan/src/
 * execution_plan.rs:696-700) for batches produced by a CPU child. */
int dfgpu_table_import(struct ArrowArray* array, struct ArrowSchema* schema, dfgpu_table_t* out);
/* Export to host memory as a struct array + schema owned by the caller (release
 * callbacks set).  Replaces: the RecordBatch items a SendableRecordBatchStream yields. */
int dfgpu_table_export(dfgpu_table_t t, struct ArrowArray* out_array, struct ArrowSchema* out_schema);
/* Rows [offset, offset + length) as one RecordBatch: the output batching of the stream contract (LimitedBatchCoalescer's
 * fixed target batch size, physical-plan/src/coalesce/mod.rs:27-120; batch_size 8192 by default) — the shim's poll_next
 * exports the next slice.  Buffers are pinned host memory from a cached pool (one DMA at PCIe rate; they return to the pool
 * when the consumer releases the batch); a child whose bitmaps do not start at a word boundary carries an Arrow `offset`. */
int dfgpu_table_export_batch(dfgpu_table_t t, int64_t offset, int64_t length, struct ArrowArray* out_array, struct ArrowSchema* out_schema);
/* The same into buffers the caller owns (e.g. arrow-rs MutableBuffers it registered once with dfgpu_host_register =
 * hipHostRegister): data_buffers[i] / validity_buffers[i] per column, NULL entries are skipped; bitmaps are copied as whole
 * 64-bit words, so `offset` must be a multiple of 64 when a validity or Boolean buffer is requested. */
int dfgpu_host_register(void* ptr, size_t bytes);
int dfgpu_host_unregister(void* ptr);
int dfgpu_table_export_into(dfgpu_table_t t, int64_t offset, int64_t length, void* const* data_buffers, void* const* validity_buffers);
/* Dictionary-encoded string columns (Arrow Dictionary(UInt8 | Int32 | UInt32 | Int64 | UInt64, Utf8 | LargeUtf8)) are
 * imported as their index column; the dictionary stays on the host, is handed on to every column derived from it by
 * selecting or reordering rows (filter, take, join payload, group keys, partitions, sort output) and re-attached on
 * export.  Dictionary values must be unique (grouping / joining on indices must mean grouping / joining on strings);
 * ORDER BY on such a column needs a sorted dictionary.  `col = 'literal'` is lowered by the caller to a comparison with
 * the literal's index: this returns it (-1 = the string is not in the dictionary, the predicate is constant false).
 * Reference: group_values/multi_group_by/dictionary.rs, hash_utils.rs:401-640 (the dictionary part). */
int dfgpu_table_dictionary_lookup(dfgpu_table_t table, int column, const char* utf8, int64_t len, int64_t* out_code);
/* `col LIKE 'pattern'` / ILIKE on a dictionary-encoded string column (BinaryExpr LikeMatch / ILikeMatch, binary.rs:640-650 ->
 * arrow-string like.rs: `%`, `_`, backslash escape): the indices of the dictionary values that match, ascending.  The caller
 * lowers the predicate to comparisons of the index column with them — index ranges when the dictionary is in ascending
 * order, where a prefix pattern matches one contiguous range.  *out_n = number of matches (may exceed `capacity`: call again
 * with a larger buffer); NOT LIKE is NOT(...) of the same, NULL rows stay NULL. */
/* A DFGPU_UTF8 column -> the same rows dictionary-encoded (Int32 indices in HBM, dictionary on the host), interned on the
 * device: one pass hashes every string and claims a slot of an open-addressing table, comparing bytes with the slot's
 * representative; representatives converge to each string's first row, so dictionary order = first-seen order exactly as
 * ArrowBytesMap::insert_if_new hands out payloads (binary_map.rs:877-960); NULL rows keep a NULL index.  `sorted` != 0
 * re-numbers the dictionary in ascending string order (what ORDER BY and range predicates over the indices need).  `out` = a
 * copy of `table` with column `column` replaced (the other columns are shared, not copied). */
int dfgpu_table_dictionary_encode(dfgpu_table_t table, int column, int sorted, dfgpu_table_t* out);
/* the inverse: a dictionary-encoded string column becomes a DFGPU_UTF8 column (its dictionary's values go to HBM once, every row
 * takes its value; NULL rows and NULL dictionary values are NULL) — e.g. to unify a Parquet column whose row groups are partly
 * dictionary-encoded and partly PLAIN strings */
int dfgpu_table_dictionary_decode(dfgpu_table_t table, int column, dfgpu_table_t* out);
/* number of values in the column's dictionary, -1 when the column is not dictionary-encoded (introspection: a dictionary-encoded
 * column's dfgpu_field is that of its indices) */
int dfgpu_table_dictionary_size(dfgpu_table_t table, int column, int64_t* out_n);
int dfgpu_table_dictionary_like(dfgpu_table_t table, int column, const char* pattern, int64_t len, int case_insensitive, int64_t* out_codes,
                                int64_t capacity, int64_t* out_n);

/* allocate a table of uninitialised device columns (filled by generators / exchange) */
int dfgpu_table_alloc(int ncols, const dfgpu_field* fields, const char* const* names, int64_t nrows, dfgpu_table_t* out);
int dfgpu_table_free(dfgpu_table_t t);
int dfgpu_table_num_rows(dfgpu_table_t t, int64_t* out);
int dfgpu_table_num_columns(dfgpu_table_t t, int* out);
int dfgpu_table_column(dfgpu_table_t t, int i, dfgpu_column_view* out);
/* zero-copy column subset / reorder (RecordBatch::project) */
int dfgpu_table_select(dfgpu_table_t t, const int* cols, int ncols, dfgpu_table_t* out);
/* zero-copy horizontal concatenation of two tables with equal row counts */
int dfgpu_table_hstack(dfgpu_table_t a, dfgpu_table_t b, dfgpu_table_t* out);
/* vertical concatenation (arrow-select concat_batches, hash_join/exec.rs:2705) */
int dfgpu_table_concat(const dfgpu_table_t* parts, int nparts, dfgpu_table_t* out);
/* contiguous row range copy (RecordBatch::slice) */
int dfgpu_table_slice(dfgpu_table_t t, int64_t offset, int64_t length, dfgpu_table_t* out);

/* A second owner of the same device table: *out shares every buffer of `t` (no copy) and is freed on its own with
 * dfgpu_table_free.  What a plan node keeps when it hands its output to more than one consumer (CollectLeft's build side shared
 * by all probe partitions, hash_join/exec.rs:772,1503; a cached scan handed to several queries). */
int dfgpu_table_retain(dfgpu_table_t t, dfgpu_table_t* out);

/* Device-resident hand-off between adjacent GPU nodes (and to / from any other ROCm component of the process) as an
 * ArrowDeviceArray: the RecordBatch a GPU node's stream yields when its consumer is another GPU node
 * (physical-plan/src/execution_plan.rs:696-700 `execute` -> SendableRecordBatchStream; the reference's FFI streams carry the same
 * ArrowArray structs, ffi/src/record_batch_stream.rs:105-114).  No byte crosses PCIe:
 *   - dfgpu_table_export_device: a struct array whose children point INTO the table's HBM buffers (device_type ARROW_DEVICE_ROCM,
 *     device_id = the HIP device; validity / Boolean bitmaps as stored; Utf8 columns as LargeUtf8 = the 64-bit offsets as stored;
 *     dictionary-encoded columns as Arrow dictionaries whose values are uploaded once).  The array holds a reference on the
 *     buffers: `t` may be freed at once, the memory lives until the consumer calls array.release.  The calling thread's stream is
 *     drained first, so sync_event is NULL.
 *   - dfgpu_table_import_device: the inverse.  An array this library exported is recognised by its release callback and comes
 *     back as the SAME buffers with dictionaries, cached column statistics and names intact; any other producer's array
 *     (ARROW_DEVICE_ROCM on an initialised device; Int32 / Int64 / UInt8 / UInt32 / UInt64 / Float64 / Date32 / Decimal128 /
 *     Boolean / LargeUtf8 children with offset 0) is wrapped zero-copy — its release callback runs when the last column that
 *     refers to it is freed — after the calling thread's stream has been made to wait on sync_event.  Consumes `array` and
 *     `schema` like dfgpu_table_import. */
int dfgpu_table_export_device(dfgpu_table_t t, struct ArrowDeviceArray* out_array, struct ArrowSchema* out_schema);
int dfgpu_table_import_device(struct ArrowDeviceArray* array, struct ArrowSchema* schema, dfgpu_table_t* out);

/* Device-resident scan cache = the HBM twin of the reference keeping hot inputs in memory (MemorySourceConfig over cached
 * RecordBatches, datasource/src/memory.rs:58; the file-metadata / statistics caches of execution/src/cache): decoded column
 * chunks / record batches stay in HBM under a caller-chosen key (file identity + row group + column ...), least recently used
 * entries leave when `budget_bytes` is exceeded.  A hit is a zero-copy view: no host read, no decompression, no PCIe.  One cache may
 * be used from many threads (the column chunks of a scan are decoded in parallel).  Entries live on the device their table
 * lives on; the budget is per cache. */
typedef struct dfgpu_cache_s* dfgpu_cache_t;
typedef struct dfgpu_cache_stats {
  int64_t entries, bytes, budget_bytes;
  int64_t hits, misses, insertions, evictions;
} dfgpu_cache_stats;
int dfgpu_cache_create(int64_t budget_bytes, dfgpu_cache_t* out);
int dfgpu_cache_free(dfgpu_cache_t cache);
/* *out = a view of the cached table (the caller frees it), or NULL when the key is not cached */
int dfgpu_cache_get(dfgpu_cache_t cache, const void* key, int64_t key_bytes, dfgpu_table_t* out);
/* keeps a view of `table` (the caller still owns its handle); a table larger than the whole budget is not kept; an existing key
 * keeps its first table */
int dfgpu_cache_put(dfgpu_cache_t cache, const void* key, int64_t key_bytes, dfgpu_table_t table);
int dfgpu_cache_clear(dfgpu_cache_t cache);
int dfgpu_cache_get_stats(dfgpu_cache_t cache, dfgpu_cache_stats* out);

/* ------------------------------------------------------------- expressions */

/* PhysicalExpr trees (physical-expr-common/src/physical_expr.rs:76) cross the ABI as a
 * flat node array; the Rust shim lowers Column / Literal / CastExpr / BinaryExpr /
 * IsNullExpr / NotExpr / CaseExpr into it (anything else keeps the CPU operator). */
typedef enum dfgpu_expr_op {
  DFGPU_EXPR_COLUMN = 1,  /* expressions/column.rs:121 */
  DFGPU_EXPR_LITERAL = 2, /* expressions/literal.rs */
  DFGPU_EXPR_CAST = 3,    /* expressions/cast.rs */
  DFGPU_EXPR_ADD = 10,    /* BinaryExpr, expressions/binary.rs:536-656 */
  DFGPU_EXPR_SUB = 11,
  DFGPU_EXPR_MUL = 12,
  /* arrow-arith `div` / `rem` (binary.rs:636-637): integers truncate toward zero; a zero divisor in a non-NULL row is the
   * error "Arrow error: Divide by zero error" (Float64 follows IEEE: inf / NaN); Decimal128(p1,s1) / Decimal128(p2,s2) has
   * scale min(38, s1 + 4) and precision min(38, p1 - s1 + s2 + scale), the quotient truncated (arrow-arith numeric.rs
   * decimal_op Op::Div, pinned by binary.rs:3042-3075,4800-4817); % has scale max(s1, s2) and precision
   * min(p1 - s1, p2 - s2) + scale (binary.rs:4819-4836) */
  DFGPU_EXPR_DIV = 13,
  DFGPU_EXPR_MOD = 14,
  DFGPU_EXPR_EQ = 20,
  DFGPU_EXPR_NE = 21,
  DFGPU_EXPR_LT = 22,
  DFGPU_EXPR_LE = 23,
  DFGPU_EXPR_GT = 24,
  DFGPU_EXPR_GE = 25,
  DFGPU_EXPR_AND = 30,
  DFGPU_EXPR_OR = 31,
  DFGPU_EXPR_NOT = 32,
  DFGPU_EXPR_IS_NULL = 33,
  DFGPU_EXPR_IS_NOT_NULL = 34,
  /* CaseExpr without a base expression (expressions/case.rs: `CASE WHEN c THEN a ELSE b END`), one WHEN per node:
   * `column` = node index of the WHEN condition (Boolean), `left` = THEN, `right` = ELSE (-1 = no ELSE: NULL).
   * The THEN value is taken where the condition is TRUE, the ELSE value where it is FALSE or NULL.  Further WHEN
   * branches nest in ELSE; `CASE x WHEN v ...` is lowered by the caller to conditions `x = v`.  THEN and ELSE have
   * the same type (the planner's coercion). */
  DFGPU_EXPR_CASE = 40,
  /* LikeExpr (expressions/like.rs -> arrow-string like / ilike): left = a DFGPU_UTF8 column or a dictionary-encoded string column
   * (the pattern is then matched against the dictionary's values once and the rows look their index up), right = the pattern (a
   * DFGPU_UTF8 literal): `%` any run of characters, `_` one character, backslash escapes; NOT LIKE = NOT of it */
  DFGPU_EXPR_LIKE = 41,
  DFGPU_EXPR_ILIKE = 42,
  /* date_part(part, Date32) -> Int32 (functions/src/datetime/date_part.rs:165-187; the form `EXTRACT(YEAR FROM d)` plans
   * to): `column` = dfgpu_date_part, `left` = the Date32 argument */
  DFGPU_EXPR_DATE_PART = 50,
  /* substr(string, start[, count]) (functions/src/unicode/substr.rs; SQL SUBSTRING): `left` = a DFGPU_UTF8 column or a
   * dictionary-encoded string column, `column` = start (1-based, in characters; values below 1 eat into count as in SQL),
   * `lit_lo` = count in characters, `is_null` != 0 = no count (to the end).  The result is a string column of the argument's
   * kind: Utf8 bytes in HBM, or the argument's indices re-pointed at the dictionary of the substrings (ascending, distinct) —
   * TPC-H Q22's `substr(c_phone, 1, 2)` over 15 M phone numbers becomes a 25-entry dictionary.  A negative count is an error. */
  DFGPU_EXPR_SUBSTR = 51
} dfgpu_expr_op;
typedef enum dfgpu_date_part { DFGPU_DATE_PART_YEAR = 0, DFGPU_DATE_PART_MONTH = 1, DFGPU_DATE_PART_DAY = 2 } dfgpu_date_part;

typedef struct dfgpu_expr_node {
  int32_t op;          /* dfgpu_expr_op */
  int32_t column;      /* COLUMN: index into the input table; CASE: node index of the WHEN condition; DATE_PART: the part; SUBSTR: start */
  int32_t left, right; /* child node indices, -1 = none */
  dfgpu_field field;   /* LITERAL: literal type; CAST: target type; else ignored */
  int32_t is_null;     /* LITERAL: SQL NULL */
  int32_t _pad;
  uint64_t lit_lo, lit_hi; /* LITERAL bits: int sign-extended to 128 / f64 bits in lit_lo */
} dfgpu_expr_node;

typedef struct dfgpu_expr {
  const dfgpu_expr_node* nodes;
  int32_t n_nodes;
  int32_t root;
  /* bytes of the string literals: a LITERAL node of type DFGPU_UTF8 holds (lit_lo = byte offset, lit_hi = byte length) into
   * this pool; NULL when the expression has none */
  const char* string_pool;
} dfgpu_expr;

/* result type of an expression over a table (PhysicalExpr::data_type, :80) */
int dfgpu_expr_type(const dfgpu_expr* e, dfgpu_table_t input, dfgpu_field* out);

/* --------------------------------------------------------------- operators */

/* FilterExec (physical-plan/src/filter.rs:85; FilterExecStream::poll_next :1367-1444):
 * evaluate `predicate` -> boolean mask, optional embedded projection (column indices,
 * NULL = all), compaction of every projected column preserving row order; rows whose
 * predicate is NULL are dropped (arrow-select filter_record_batch). */
int dfgpu_filter(dfgpu_table_t input, const dfgpu_expr* predicate, const int* projection, int nproj,
                 dfgpu_table_t* out);

/* ProjectionExec (physical-plan/src/projection.rs:439,713-740): evaluate n expressions */
int dfgpu_project(dfgpu_table_t input, const dfgpu_expr* exprs, const char* const* names, int n,
                  dfgpu_table_t* out);

/* Default of perfect_hash_join_min_key_density when `opts` is NULL.  The reference ships 0.15
 * (common/src/config.rs:923), tuned for CPU caches; on MI355X the direct-address table stays the
 * better structure far below that (its memset + one sequential-ish 4 B read per probe row cost less
 * than the 1.5-2 random 64 B sectors per probe row of a chained hash table: 7.1 ms vs < 2 ms for
 * the 324 M-row probe of TPC-H Q3 at SF100, density 0.024).  The knob is the reference's own; only
 * its default differs.  Results do not depend on the table kind. */
#define DFGPU_DEFAULT_MIN_KEY_DENSITY (1.0 / 64.0)

typedef struct dfgpu_join_options {
  /* execution.perfect_hash_join_small_build_threshold (common/src/config.rs:913), default 1024 */
  int64_t perfect_hash_join_small_build_threshold;
  /* execution.perfect_hash_join_min_key_density (config.rs:923), default DFGPU_DEFAULT_MIN_KEY_DENSITY */
  double perfect_hash_join_min_key_density;
  /* 0 = auto: rank map (GPU-native compressed direct addressing: 1 bit per value of the key range +
   *     popcount directory; unique integer keys, range/rows <= 256) -> else ArrayMap by the reference's
   *     gating (hash_join/exec.rs:111-191) with the two knobs above -> else chained hash table;
   * 1 = force chained hash table; 2 = force ArrayMap; 3 = force rank map (2/3: error if not applicable);
   * 4 = LDS-staged radix-partitioned table: both sides are radix partitioned on the top bits of a mixed key until a build
   *     partition fits the LDS of one workgroup, every partition pair is joined in LDS (any key set, duplicate keys, NULL ==
   *     NULL; all join types; JoinFilter).  The output is in partition order, not probe order: for plans in which no ancestor
   *     observes HashJoinExec's probe-side ordering (what probe_mode 4 declares);
   * 5 = force the flat table: open addressing with the KEYS INLINE — the packed key columns (<= 16 bytes: any mix of integer /
   *     date / Float64 / Decimal128 columns) and the first build row per slot, rows with equal keys chained behind it; one
   *     access per probe row and no key re-check.  Error if the key columns do not pack.  `auto` takes it wherever it used
   *     to take the chained table and the keys pack. */
  int32_t table_mode;
  /* test hook = cargo feature `force_hash_collisions` (common/src/hash_utils.rs:1186-1197):
   * every key hashes to 0 so only the key re-check (K4) keeps results right */
  int32_t force_hash_collisions;
  /* probe strategy when a probe row has at most one match (unique build keys, RightSemi/RightAnti)
   * and the payload is non-nullable:
   *   0 = auto: two passes (lookup -> scan -> materialise), output in probe order, exact allocation;
   *   1 = two passes (same as auto today);
   *   2 = single pass, output in probe order (decoupled look-back; slower than two passes on MI355X);
   *   3 = single pass, UNORDERED: probe order inside 2048-row tiles, tiles in arbitrary order.  For
   *       plans where no ancestor needs the probe-side ordering (HashJoinExec::maintains_input_order,
   *       joins/hash_join/exec.rs:1024-1040,1352, would have to report false for the shim node).
   *   4 = order not needed (a planner's hint): 3 when it is applicable, else the ordered / general path.
   * Modes 2/3 allocate the output for the probe-row upper bound and fail if they are not applicable. */
  int32_t probe_mode;
  /* HashJoinExec::null_aware (hash_join/exec.rs:429-455,786): NOT IN semantics for anti joins on a single key column.
   * LeftAnti  (stream.rs:762-808,1016-1076): a NULL probe key in any probe table empties the whole output
   *           (dfgpu_join_emit_unmatched returns no rows); otherwise, if any probe row was seen, build rows with a NULL
   *           key are not emitted.
   * RightAnti (stream.rs:762-768,937-955): a NULL build key empties the output; an empty build side emits every probe
   *           row (NULL keys included); otherwise probe rows with a NULL key are not emitted.  No JoinFilter allowed.
   * Any other join type, or more than one key column, is an error (the reference's try_new messages). */
  int32_t null_aware;
} dfgpu_join_options;

/* collect_left_input (physical-plan/src/joins/hash_join/exec.rs:2569-2776): build the
 * join table over the whole build side.  `build` must stay alive until dfgpu_join_free
 * (the handle holds a reference).  key_cols index into `build`. */
int dfgpu_join_build(dfgpu_table_t build, const int* key_cols, int nkeys, int null_equality,
                     const dfgpu_join_options* opts, dfgpu_join_t* out);
/* The build side as a stream of batches = HashJoinStream's CollectBuildSide state (hash_join/stream.rs:127-140,591-640) over
 * collect_left_input (exec.rs:2569-2705): push every batch of the build child (each push grows a memory reservation and fails
 * with "Resources exhausted" like try_grow, exec.rs:2608), then finish = concat_batches + table build.  finish consumes the
 * builder; free abandons it.  The probe side is a stream already: dfgpu_join_probe per probe batch, then
 * dfgpu_join_emit_unmatched (ExhaustedProbeSide). */
typedef struct dfgpu_join_builder_s* dfgpu_join_builder_t;
int dfgpu_join_builder_create(const int* key_cols, int nkeys, int null_equality, const dfgpu_join_options* opts, dfgpu_join_builder_t* out);
int dfgpu_join_builder_push(dfgpu_join_builder_t b, dfgpu_table_t batch);
int dfgpu_join_builder_finish(dfgpu_join_builder_t b, dfgpu_join_t* out);
int dfgpu_join_builder_free(dfgpu_join_builder_t b);
/* peak device bytes of a hash join of these sizes (output_rows < 0: assume one output row per probe row), for
 * dfgpu_mem_try_reserve: what the optimizer rule checks before it substitutes the GPU operator */
int dfgpu_join_estimate_bytes(int64_t build_rows, int64_t build_row_bytes, int64_t probe_rows, int64_t output_rows, int64_t output_row_bytes, int64_t* out);

/* HashJoinStream::process_probe_batch (hash_join/stream.rs:740-1000): probe one probe
 * table (any size — the whole partition, not 8192-row batches) and materialise the output
 * for `join_type`.  Output columns = build_out_cols of the build table followed by
 * probe_out_cols of the probe table (the `projection` of HashJoinExec, exec.rs:752);
 * Semi / Anti joins emit one side only; Mark joins append a Boolean `mark` column.
 * For Left/Full/LeftSemi/LeftAnti/LeftMark the probe call emits only what is known per
 * probe batch (matched pairs) and records visited build rows; call
 * dfgpu_join_emit_unmatched once all probe tables are done (stream.rs:1002-). */
int dfgpu_join_probe(dfgpu_join_t ht, dfgpu_table_t probe, const int* probe_key_cols, int join_type,
                     const int* build_out_cols, int n_build_out, const int* probe_out_cols, int n_probe_out,
                     dfgpu_table_t* out);
/* (ABI 13) The probe in bounded pieces = HashJoinStream's resumable lookup (get_matched_indices_with_limit_offset + MapOffset,
 * joins/join_hash_map.rs:389-484; hash_join/stream.rs:396-437: at most batch_size output rows per poll, the next poll resumes where
 * the last one stopped).  Takes the probe rows from `probe_offset` (0, or what the previous call returned) up to the last 64-row
 * boundary at which the output stays within `max_output_rows`, materialises exactly those rows' output like dfgpu_join_probe, and
 * returns where to resume in *next_offset (= the probe's row count when it is exhausted).  The cut falls on whole 64-row words — one
 * word is always taken, so a single word whose matches exceed the bound comes out whole (the reference can stop inside one probe
 * row's chain).  With the LDS radix table (table kind radix_lds) and for null-aware anti joins the bound applies to the probe rows
 * taken per call.  An M:N join whose full output would not fit HBM is consumed piece by piece this way; dfgpu_join_emit_unmatched
 * follows the last piece as usual. */
int dfgpu_join_probe_bounded(dfgpu_join_t ht, dfgpu_table_t probe, const int* probe_key_cols, int join_type,
                             const int* build_out_cols, int n_build_out, const int* probe_out_cols, int n_probe_out,
                             int64_t probe_offset, int64_t max_output_rows, dfgpu_table_t* out, int64_t* next_offset);
/* JoinFilter (physical-plan/src/joins/join_filter.rs): a residual predicate over columns of both sides.  The
 * expression's Column i is the i-th (column_index, column_side) entry — the reference's intermediate batch
 * (apply_join_filter_to_indices, joins/utils.rs:1248-1318).  Key-equal pairs whose filter value is not TRUE are
 * not matches: they mark no build row visited and leave outer / anti / mark rows unmatched. */
typedef struct dfgpu_join_filter {
  dfgpu_expr expression;
  const int32_t* column_index; /* index into its side's table */
  const int32_t* column_side;  /* 0 = left (build side), 1 = right (probe side) */
  int32_t n_columns;
} dfgpu_join_filter;
int dfgpu_join_probe_with_filter(dfgpu_join_t ht, dfgpu_table_t probe, const int* probe_key_cols, int join_type, const dfgpu_join_filter* filter,
                                 const int* build_out_cols, int n_build_out, const int* probe_out_cols, int n_probe_out, dfgpu_table_t* out);

/* min / max / non-null count of an integer column and whether it is strictly ascending (the statistics
 * ArrayMap::try_new takes from the build keys, joins/array_map.rs:175-203; also the probe-key bounds the multi-GPU
 * exchange prunes the build-side broadcast with — hash_join/shared_bounds.rs:277-284 turned around).  One pass, one
 * device-to-host copy.  *out_valid == 0: no non-null value, min / max are INT64_MAX / INT64_MIN. */
int dfgpu_column_minmax(dfgpu_table_t table, int column, int64_t* out_min, int64_t* out_max, int64_t* out_valid, int* out_ascending);
/* The membership half of the join's dynamic filter (PushdownStrategy, hash_join/shared_bounds.rs:275-284, chosen in
 * collect_left_input, hash_join/exec.rs:2727-2751): a build-side key column that is small — at most `max_size` bytes
 * (optimizer.hash_join_inlist_pushdown_max_size, default 128 KiB) and at most `max_distinct_values` distinct values
 * (hash_join_inlist_pushdown_max_distinct_values, default 150; 0 = never) — is pushed to the probe-side scan as
 * `key IN (v1, v2, ...)`: *out_n = the number of distinct non-NULL values, written ascending to out_values (up to `capacity`).
 * Larger build sides use the Map strategy (the hash table itself is the membership test — here: the probe kernel), and only
 * their bounds are pushed: *out_n = -1.  An empty column gives *out_n = 0 (PushdownStrategy::Empty: nothing can match). */
int dfgpu_column_inlist(dfgpu_table_t table, int column, int64_t max_size, int64_t max_distinct_values, int64_t* out_values, int64_t capacity, int64_t* out_n);
/* The Map strategy of the join's dynamic filter (PushdownStrategy::Map, hash_join/shared_bounds.rs:275-284; HashTableLookupExpr,
 * hash_join/partitioned_hash_eval.rs:278): a build side too large for an IN list pushes the TABLE ITSELF to the probe-side scan as the
 * membership test.  *out = a one-column Boolean table `contains` of `probe`'s rows: TRUE where the row's key is in the build side
 * (never NULL: a NULL key is FALSE under NullEqualsNothing).  Needs no build row — over a rank map of keys in no particular order the
 * bitmap alone answers.  The scan evaluates it on the key column of a row group FIRST and decodes the other columns only when a row
 * passes (datafusion_amd/parquet.py read_table(membership=...)). */
int dfgpu_join_contains(dfgpu_join_t ht, dfgpu_table_t probe, const int* probe_key_cols, dfgpu_table_t* out);
/* The same with a FilterExec fused below the probe side (filter.rs:1396-1419 -> hash_join/stream.rs:687-1000):
 * probe rows whose predicate is false or NULL do not exist for the join.  With the single-pass probe the predicate's
 * row mask is applied inside the probe kernel and the filtered probe table is never materialised; every other
 * probe flavour filters first.  What the optimizer rule substitutes for FilterExec -> HashJoinExec(probe side). */
int dfgpu_join_probe_filtered(dfgpu_join_t ht, dfgpu_table_t probe, const dfgpu_expr* probe_predicate, const int* probe_key_cols, int join_type,
                              const int* build_out_cols, int n_build_out, const int* probe_out_cols, int n_probe_out, dfgpu_table_t* out);
int dfgpu_join_emit_unmatched(dfgpu_join_t ht, int join_type, const int* build_out_cols, int n_build_out,
                              const dfgpu_field* probe_fields, const char* const* probe_names, int n_probe_out,
                              dfgpu_table_t* out);
/* BuildProbeJoinMetrics (joins/utils.rs:1756-1800) + which table was built */
typedef struct dfgpu_join_info {
  int64_t build_rows;
  int64_t table_bytes;
  int32_t used_array_map; /* 1 = direct-address table (ArrayMap, joins/array_map.rs:103) */
  int32_t build_keys_unique;
  int64_t probe_rows;  /* accumulated over probe calls */
  int64_t output_rows; /* accumulated */
  int32_t table_kind;  /* 0 = chained hash table (JoinHashMap), 1 = ArrayMap, 2 = rank map (bitmap + popcount directory), 3 = LDS radix partitions,
                        * 4 / 5 = flat hash table with inline keys of <= 8 / <= 16 bytes */
  int32_t build_keys_ascending; /* 1 = single integer key, strictly ascending in row order */
} dfgpu_join_info;
int dfgpu_join_get_info(dfgpu_join_t ht, dfgpu_join_info* out);
int dfgpu_join_free(dfgpu_join_t ht);

/* AggregateMode (physical-plan/src/aggregates/mod.rs:289-400) */
typedef enum dfgpu_agg_mode {
  DFGPU_AGG_PARTIAL = 0,          /* raw input -> partial state (state_fields) */
  DFGPU_AGG_FINAL = 1,            /* partial state -> final values */
  DFGPU_AGG_FINAL_PARTITIONED = 2,
  DFGPU_AGG_SINGLE = 3,           /* raw input -> final values */
  DFGPU_AGG_SINGLE_PARTITIONED = 4,
  DFGPU_AGG_PARTIAL_REDUCE = 5    /* partial state -> partial state: merges like Final, emits like Partial (mod.rs:340-361) */
} dfgpu_agg_mode;
typedef enum dfgpu_agg_func { DFGPU_AGG_SUM = 0, DFGPU_AGG_MIN = 1, DFGPU_AGG_MAX = 2, DFGPU_AGG_COUNT = 3, DFGPU_AGG_AVG = 4 } dfgpu_agg_func;
typedef struct dfgpu_agg_spec {
  int32_t func;          /* dfgpu_agg_func */
  int32_t has_arg;       /* 0 = COUNT(*) */
  dfgpu_expr arg;        /* argument expression over the input (raw modes) */
  const char* name;      /* output column name */
  /* planner-declared return type (AggregateFunctionExpr::field); type 0 = derive from the
   * argument type.  Give it in FINAL modes for AVG(Decimal128), whose precision cannot be
   * recovered from the clamped sum-state type. */
  dfgpu_field return_field;
} dfgpu_agg_spec;

/* AggregateExec (aggregates/mod.rs:839): group keys + accumulators
 * (GroupValues group_values/mod.rs:93, GroupsAccumulator expr-common/src/
 * groups_accumulator.rs:105).  In FINAL modes the input is the partial-state schema the
 * reference uses (group cols, then per aggregate: SUM -> [sum]; COUNT -> [count];
 * MIN/MAX -> [value]; AVG -> [count u64, sum]; sum.rs:281-301, average.rs:317-360) and
 * `arg`/`group_by` expressions are ignored beyond their count. */
int dfgpu_agg_create(int mode, const dfgpu_expr* group_by, const char* const* group_names, int n_group,
                     const dfgpu_agg_spec* aggs, int n_aggs, dfgpu_agg_t* out);
/* GROUPING SETS / CUBE / ROLLUP = PhysicalGroupBy with several groups (aggregates/mod.rs:400-520: expr, null_expr, groups; pinned by
 * check_grouping_sets, :3428-3590).  group_by[g] / null_by[g] = the key expression and its typed NULL literal, groups[s * n_group + g]
 * != 0 = column g is NULL in grouping set s.  Output: the n_group key columns, `__grouping_id` (bit n_group-1-g set = column g is
 * NULLed out; UInt8 up to 8 columns, UInt32 up to 32, else UInt64), then the aggregates (Partial: their state columns).  Raw modes
 * only (Partial / Single / SinglePartitioned): the Final node of such a plan groups by the partial state's n_group + 1 key columns
 * and is a plain dfgpu_agg_create.  update / update_filtered / emit / free as for any aggregate. */
int dfgpu_agg_create_grouping_sets(int mode, const dfgpu_expr* group_by, const dfgpu_expr* null_by, const char* const* group_names, int n_group,
                                   const uint8_t* groups, int n_sets, const dfgpu_agg_spec* aggs, int n_aggs, dfgpu_agg_t* out);
/* aggregate_batch_inner (aggregate_hash_table/common.rs:205-236) over a whole table */
int dfgpu_agg_update(dfgpu_agg_t h, dfgpu_table_t input);
/* The same with a FilterExec predicate fused in front (filter.rs:1396-1419 -> common.rs:205-236): rows whose
 * predicate is false or NULL neither create groups nor accumulate.  What the optimizer rule substitutes for
 * AggregateExec(ProjectionExec(FilterExec(x))) — e.g. TPC-H Q1, tpch/plans/q1.slt.part:50-58: the projection's
 * expressions are inlined into the group/argument expressions, and predicate + expressions + accumulation run
 * as ONE pass over the referenced input columns (rowprog: per-row register program, no intermediate columns).
 * predicate == NULL is dfgpu_agg_update. */
int dfgpu_agg_update_filtered(dfgpu_agg_t h, dfgpu_table_t input, const dfgpu_expr* predicate);
/* number of updates of `h` that ran fused (0 when the expression forest did not fit the register program and
 * the column-at-a-time evaluator was used; results are identical either way) */
int dfgpu_agg_fused_updates(dfgpu_agg_t h, int64_t* out);
/* Options (ABI 12): the dispatch policy's thresholds and switches by name — the library's twin of the reference's ConfigOptions
 * (common/src/config.rs).  Every default is derived from the device at dfgpu_init (CU count, L2 size of an XCD, LDS per CU), e.g.
 * "rows worth a pass" = 16 Ki rows per CU; an embedding engine, or a test that must force a path on a small input, overrides by name.
 * value == NULL restores the option's default; name == NULL restores all of them.  Unknown names are kept and ignored.
 *   jit = 0|1 (1)                       plan-time specialisation of the fused aggregate node (hiprtc)
 *   jit.min_rows (rows worth a pass)    smallest input a node is specialised for
 *   jit.strict = 0|1 (0)                a failed specialisation is an error instead of a fall-back to the interpreter
 *   jit.cache = 0|1 (1), jit.cache_dir  on-disk cache of compiled nodes; jit.dump_dir: generated sources are written there
 *   agg.partitioned = 0|1 (1), agg.partitioned_min_rows (2 x rows worth a pass), agg.grouped_move, agg.direct_table, agg.runs = 0|1 (1)
 *   join.grouped_probe = 0|1 (1), join.grouped_min_rows (rows worth a pass), join.beyond_cache_bytes (4 x the L2 of an XCD),
 *   join.grouped_bits, join.near_window (test hooks: 0 = derived)
 *   join.radix_onesweep = 0|1 (1: the LDS radix join partitions by two or three 8-bit passes straight off the key column; 0: round 5's
 *   record kernel + 6-bit passes), join.radix_tile_threads = 256|512 (512) and join.radix_tile_items = 8|16 (with 256 threads; 16): the partitioning tile, 512 x 8 = 4096 rows by default, join.radix_fused_emit = 0|1 (1),
 *   join.radix_partition_rows (2400: build rows per LDS partition on average; test hook: small values give several passes on small inputs)
 *   sort.carried = o|i|p|0 (o), sort.carried_min_rows (rows worth a pass), sort.lsd = 0|1 (1: narrow keys sorted by record passes alone),
 *   sort.lsd_ahead = 0|1 (1: two-pass narrow-key sorts take their offsets from the digit-totals pass instead of a look-back)
 *   parquet.device_decode = 0|1 (1: page headers on the host, levels / runs / values decoded by kernels), parquet.snappy = host|device (host),
 *   parquet.in_flight (4: chunks a scan worker keeps in flight)
 * The same table is read from the environment variable DFGPU_OPTIONS="name=value,name=value" (lower priority than this call).
 * DFGPU_TRACE="agg,join,scan,dict,rowprog" (or "all") prints the named subsystems' decisions on stderr; no other environment
 * variable changes what the library does. */
int dfgpu_set_option(const char* name, const char* value);
/* process-wide switch for expression fusion (default on); off = always column-at-a-time (A/B measurements, tests) */
int dfgpu_set_fusion(int on);
/* Plan-time specialisation (jit.hip): for inputs of at least `jit.min_rows` rows (dfgpu_set_option; default 4 Mi on 256 CUs) the fused
 * aggregate node is compiled for its expression forest with hiprtc (cached per process by source text); this reports
 * how many distinct nodes were compiled and the total compile time.  Option jit = 0 keeps the interpreter. */
int dfgpu_jit_stats(int64_t* compiles, double* compile_ms);
/* Compiled nodes are kept as code objects on disk (option jit.cache_dir, else $XDG_CACHE_HOME/dfgpu/jit, else ~/.cache/dfgpu/jit;
 * option jit.cache = 0: off) keyed by target + hiprtc version + source: a plan seen before by ANY process of the machine costs a file
 * read instead of a 150 ms compile.  Modules are loaded once per (device, source) — a process that drives several GPUs loads the
 * same code object on each.  Any out pointer may be NULL. */
int dfgpu_jit_cache_stats(int64_t* disk_hits, int64_t* disk_writes, int64_t* modules_loaded);
/* next_output_batch_inner (common.rs:247-300): emit all groups */
int dfgpu_agg_emit(dfgpu_agg_t h, dfgpu_table_t* out);
int dfgpu_agg_free(dfgpu_agg_t h);

/* SortExec (physical-plan/src/sorts/sort.rs:1366; sort_batch :894-914) and TopK
 * (topk/mod.rs:397) when fetch >= 0. */
int dfgpu_sort(dfgpu_table_t input, const int* key_cols, const uint8_t* descending, const uint8_t* nulls_first,
               int nkeys, int64_t fetch, dfgpu_table_t* out);

/* RepartitionExec, Partitioning::Hash (physical-plan/src/repartition/mod.rs:1097-1150):
 * partition = create_hashes(keys; seed 0) % nparts; row order preserved inside each
 * partition.  outs[nparts].  Also the per-GPU routing step of the multi-GPU exchange. */
int dfgpu_partition(dfgpu_table_t input, const int* key_cols, int nkeys, int nparts, dfgpu_table_t* outs);
/* create_hashes (common/src/hash_utils.rs:1239) for tests / routing checks */
int dfgpu_hash_columns(dfgpu_table_t input, const int* key_cols, int nkeys, uint64_t seed, uint64_t* out_device);

/* -------------------------------------------------------- multi-GPU exchange */

/* The exchange step of a distributed plan, below the C ABI (SURVEY §8b dfgpu_exchange, §8e):
 *   RepartitionExec(Partitioning::Hash(keys, n)) across GPUs (physical-plan/src/repartition/mod.rs:1097-1150, channels
 *   :154-360) = partition kernel + all-to-all(v); PartitionMode::CollectLeft's collected build side
 *   (joins/hash_join/exec.rs:1325-1328) = all-gather, optionally pruned by the destinations' probe-key bounds
 *   (hash_join/shared_bounds.rs:277-284 turned around).  In the reference these are in-process channels between partition
 *   tasks; here a partition is a GPU and the channel is RCCL over xGMI (grouped ncclSend / ncclRecv, one slice per link).
 * A communicator spans `world` ranks, one GPU each:
 *   - one process per GPU: rank 0 calls dfgpu_comm_unique_id, the host engine hands the 128 bytes to the other processes,
 *     every process calls dfgpu_comm_init_rank on its current device (a collective: returns when all ranks arrived);
 *   - one process, several GPUs: dfgpu_comm_init_all makes the devices given to dfgpu_init ranks 0..n-1;
 *   - dfgpu_comm_init_host: the same protocol over collectives the embedding engine supplies on host memory (an engine
 *     that spans nodes with its own transport; tests that run two ranks on one GPU).
 * Every exchange call is a collective over the communicator and takes / returns ONE table per local rank (arrays of
 * dfgpu_comm_info.n_local entries: 1 unless dfgpu_comm_init_all made the communicator).  Value buffers, validity bitmaps,
 * Boolean columns and dictionary-encoded strings cross: ranks whose dictionaries differ merge them (ascending, identical
 * on every rank) and rewrite their indices first, so routing / joining / grouping on the indices means the strings
 * everywhere.  Results hold the rows received from rank 0, 1, ... in that order, each sender's rows in its own order. */
typedef struct dfgpu_comm_s* dfgpu_comm_t;
#define DFGPU_COMM_ID_BYTES 128
int dfgpu_comm_unique_id(uint8_t* out_id /* DFGPU_COMM_ID_BYTES */);
int dfgpu_comm_init_rank(const uint8_t* id /* DFGPU_COMM_ID_BYTES */, int world, int rank, dfgpu_comm_t* out);
int dfgpu_comm_init_all(dfgpu_comm_t* out);
typedef struct dfgpu_host_transport {
  void* ctx;
  /* send[p] (send_bytes[p] bytes, host memory) goes to rank p, recv[p] receives recv_bytes[p] bytes from it; entries
   * of the calling rank itself are empty.  Returns 0 on success. */
  int (*alltoallv)(void* ctx, const void* const* send, const int64_t* send_bytes, void* const* recv, const int64_t* recv_bytes);
  /* every rank contributes `bytes` bytes; `all` receives world * bytes in rank order */
  int (*allgather)(void* ctx, const void* mine, int64_t bytes, void* all);
} dfgpu_host_transport;
int dfgpu_comm_init_host(const dfgpu_host_transport* transport, int world, int rank, dfgpu_comm_t* out);
int dfgpu_comm_free(dfgpu_comm_t comm);
int dfgpu_comm_info(dfgpu_comm_t comm, int* world, int* first_rank, int* n_local);
/* What the transport itself reports: is_rccl = 1 (RCCL) / 0 (host transport); rccl_ranks = ncclCommCount and rccl_rank =
 * ncclCommUserRank of the first local rank's communicator, -1 under the host transport.  A first run on a multi-GPU node checks
 * rccl_ranks == world before it trusts a scaling number (bench.py prints it as n_ranks_seen_by_rccl). */
int dfgpu_comm_transport_info(dfgpu_comm_t comm, int* is_rccl, int* rccl_ranks, int* rccl_rank);
/* RepartitionExec(Hash): outs[l] = every row (of all ranks) with hash(keys; seed 0) % world == rank of local l —
 * routing is dfgpu_partition's, so co-partitioned inputs meet on one GPU (hash_join/exec.rs:1312-1324) */
int dfgpu_exchange_hash(dfgpu_comm_t comm, const dfgpu_table_t* inputs, const int* key_cols, int nkeys, dfgpu_table_t* outs);
/* The same exchange as a STREAM (ABI 12): RepartitionExec hands batches to its output partitions' channels while their consumers run
 * (physical-plan/src/repartition/mod.rs:154-360, 1097-1150).  Every rank cuts its input into `n_chunks` row ranges (the same number on
 * every rank: the chunks' all-to-all(v)s pair up; a rank with fewer rows sends empty pieces); _next hands over chunk k — the rows of
 * every rank's k-th range that route here — while chunk k + 1 crosses the links and chunk k + 2 is being partitioned, on threads and
 * streams of the library's own.  The union of the chunks is dfgpu_exchange_hash's result (row order differs: chunk by chunk, rank order
 * inside a chunk).  *done = 1 (and no table) after the last chunk.  Tables with Utf8 columns arrive as ONE chunk.  While a stream is
 * open its communicator belongs to it: no other collective of the same communicator may be called, and every rank must take (or
 * _free) the same chunks.  _free joins the workers; it may be called before the stream is drained once every rank does so. */
typedef struct dfgpu_exchange_stream_s* dfgpu_exchange_stream_t;
int dfgpu_exchange_hash_stream_open(dfgpu_comm_t comm, const dfgpu_table_t* inputs, const int* key_cols, int nkeys, int n_chunks, dfgpu_exchange_stream_t* out);
int dfgpu_exchange_hash_stream_next(dfgpu_exchange_stream_t stream, dfgpu_table_t* outs, int* done);
int dfgpu_exchange_hash_stream_free(dfgpu_exchange_stream_t stream);
/* all-gather: outs[l] = the rows of all ranks in rank order (CollectLeft's build side; CoalescePartitionsExec to every rank) */
int dfgpu_exchange_broadcast(dfgpu_comm_t comm, const dfgpu_table_t* inputs, dfgpu_table_t* outs);
/* the all-gather of a build side pruned by bounds: rank r receives only build rows whose key lies inside [min, max] of
 * r's own probe keys — a superset of what it can match, so the local join is unchanged.  Inputs clustered by key (scans
 * of range-partitioned data) move almost nothing; uniformly spread keys degrade to the full all-gather.  Integer keys. */
int dfgpu_exchange_broadcast_pruned(dfgpu_comm_t comm, const dfgpu_table_t* builds, int build_key, const dfgpu_table_t* probes, int probe_key,
                                    dfgpu_table_t* outs);
/* The exchange of a distributed ORDER BY (SortPreservingMergeExec over partitions on different GPUs, sorts/sort_preserving_merge.rs:91;
 * SURVEY §8e "all-to-all for sample-sort"): every rank samples its rows' first sort key (`key_col`, integer-like), the samples of all
 * ranks give world - 1 splitters, a row goes to the rank whose key range holds its key — rank 0 the smallest keys (the largest when
 * `descending`), NULL keys to the first or the last rank as `nulls_first` says.  Rows with equal first keys land on one rank, so after a
 * local dfgpu_sort by the full key list the ranks' outputs read in rank order are the globally sorted result: no rank holds or
 * re-sorts everything.  outs[l] = the rows rank l owns (in no particular order yet). */
int dfgpu_exchange_range(dfgpu_comm_t comm, const dfgpu_table_t* inputs, int key_col, int descending, int nulls_first, dfgpu_table_t* outs);
/* PartitionMode::CollectLeft with build-side emission (Left / Full / LeftSemi / LeftAnti / LeftMark) when the probe partitions sit on
 * different GPUs: in the reference they all mark ONE visited bitmap and the last of them reports the build rows
 * (hash_join/exec.rs:1312-1330).  Here every rank probes ITS probe partition against its copy of the replicated build side
 * (dfgpu_exchange_broadcast), then this collective ORs the copies' visited marks (and the null-aware flags) so that
 * dfgpu_join_emit_unmatched sees the union — the same rows on every rank, of which rank r keeps its share by build-row position.
 * joins[l] = the table of local rank l; all build sides must hold the same rows. */
int dfgpu_exchange_join_visited(dfgpu_comm_t comm, const dfgpu_join_t* joins);
/* what crossed GPU boundaries since the communicator was created / last reset (summed over this process's local ranks) */
typedef struct dfgpu_exchange_stats {
  int64_t bytes_sent_to_peers, bytes_received_from_peers;
  int64_t rows_sent_to_peers, rows_received_from_peers;
  int64_t messages;     /* point-to-point sends issued (a slice above 1 GiB is cut) */
  int64_t collectives;  /* grouped all-to-all(v) rounds */
} dfgpu_exchange_stats;
int dfgpu_comm_stats(dfgpu_comm_t comm, dfgpu_exchange_stats* out, int reset);

/* ----------------------------------------------------------- scan -> device */

/* One Parquet column chunk decoded straight into a device column.  The reference's scan
 * (datasource-parquet, DataSourceExec datasource/src/source.rs:366) reads each projected column chunk's byte range
 * (ColumnChunkMetaData: dictionary_page_offset | data_page_offset, total_compressed_size) and decodes pages on the CPU;
 * a GPU scan node hands the same bytes to dfgpu_parquet_decode_chunk instead and assembles the row group with
 * dfgpu_table_hstack / dfgpu_table_concat.  Enum values are parquet.thrift's. */
typedef enum dfgpu_parquet_type {
  DFGPU_PARQUET_BOOLEAN = 0, DFGPU_PARQUET_INT32 = 1, DFGPU_PARQUET_INT64 = 2, DFGPU_PARQUET_INT96 = 3, DFGPU_PARQUET_FLOAT = 4,
  DFGPU_PARQUET_DOUBLE = 5, DFGPU_PARQUET_BYTE_ARRAY = 6, DFGPU_PARQUET_FIXED_LEN_BYTE_ARRAY = 7
} dfgpu_parquet_type;
typedef enum dfgpu_parquet_codec {
  DFGPU_PARQUET_UNCOMPRESSED = 0, DFGPU_PARQUET_SNAPPY = 1, DFGPU_PARQUET_GZIP = 2, DFGPU_PARQUET_LZO = 3, DFGPU_PARQUET_BROTLI = 4,
  DFGPU_PARQUET_LZ4 = 5, DFGPU_PARQUET_ZSTD = 6, DFGPU_PARQUET_LZ4_RAW = 7
} dfgpu_parquet_codec;
typedef struct dfgpu_parquet_column {
  int32_t physical_type;         /* dfgpu_parquet_type (SchemaElement.type) */
  int32_t type_length;           /* FIXED_LEN_BYTE_ARRAY: bytes per value */
  int32_t codec;                 /* dfgpu_parquet_codec (ColumnMetaData.codec) */
  int32_t max_definition_level;  /* 0 = required, 1 = optional; deeper nesting is not supported */
  int32_t max_repetition_level;  /* must be 0 */
  int32_t _pad;
  int64_t num_values;            /* ColumnMetaData.num_values = rows of the row group for a flat column */
  dfgpu_field field;             /* Arrow type to produce.  INT32 -> Int32 / Date32 / UInt8 / UInt32 / Decimal128, INT64 -> Int64 /
                                  * UInt64 / Decimal128, DOUBLE -> Float64, FIXED_LEN_BYTE_ARRAY -> Decimal128, BYTE_ARRAY (all pages
                                  * dictionary-encoded) -> Int32 indices of a dictionary-encoded string column (ascending dictionary) */
  const char* name;
} dfgpu_parquet_column;
/* Supported: data pages v1 / v2, PLAIN / PLAIN_DICTIONARY / RLE_DICTIONARY values, RLE definition levels, UNCOMPRESSED /
 * SNAPPY / ZSTD (libzstd.so.1 on the host).  Anything else returns an error and the caller keeps the CPU scan.
 * `out` = a one-column device table of num_values rows. */
int dfgpu_parquet_decode_chunk(const uint8_t* chunk, int64_t chunk_bytes, const dfgpu_parquet_column* column, dfgpu_table_t* out);
/* the host half alone (page headers, decompression, levels, run headers): needs no GPU and no dfgpu_init */
typedef struct dfgpu_parquet_chunk_info {
  int32_t n_pages, n_dictionary_pages, n_data_pages_v1, n_data_pages_v2, n_plain_pages, n_dictionary_encoded_pages;
  int64_t n_runs_rle, n_runs_bitpacked;
  int64_t values, nulls;                       /* rows of the chunk, NULLs among them */
  int64_t uncompressed_bytes, compressed_bytes; /* page bodies */
  int64_t dictionary_values;
} dfgpu_parquet_chunk_info;
int dfgpu_parquet_inspect_chunk(const uint8_t* chunk, int64_t chunk_bytes, const dfgpu_parquet_column* column, dfgpu_parquet_chunk_info* out);

/* The projected column chunks of a scan in ONE call: chunks[g * n_columns + j] = column j of the g-th row group read (file order).
 * `threads` host threads inside the library each take chunks off the list — a chunk's host half on that thread, its device half on
 * that thread's stream — as the reference's scan decodes row groups on its partition threads (DataSourceExec,
 * datasource/src/source.rs:366); the row groups are then put side by side and below each other on the device.  A BYTE_ARRAY
 * column asked for as dictionary indices comes back as Utf8 when any of its chunks holds PLAIN pages.  `cache` (may be NULL) +
 * per-chunk keys: decoded chunks are taken from / left in the device chunk cache (dfgpu_cache_*), `chunks_from_cache` (may be NULL)
 * counts the hits.  `out` = n_columns columns, the row groups' rows in order. */
typedef struct dfgpu_parquet_chunk {
  const uint8_t* bytes;          /* the chunk's byte range (dictionary page first), e.g. where the page cache maps the file */
  int64_t n_bytes;
  dfgpu_parquet_column column;
  const void* cache_key;         /* NULL = not cached */
  int64_t cache_key_bytes;
} dfgpu_parquet_chunk;
int dfgpu_parquet_read_chunks(const dfgpu_parquet_chunk* chunks, int32_t n_row_groups, int32_t n_columns, int32_t threads, dfgpu_cache_t cache,
                              dfgpu_table_t* out, int64_t* chunks_from_cache);

/* Arrow IPC files and streams (DataSourceExec over an ArrowSource, datasource-arrow/src/source.rs:260-330: arrow-ipc FileReader /
 * StreamReader) scanned straight into HBM.  The file already holds Arrow buffers: dfgpu_ipc_open walks the encapsulated messages of
 * the bytes it is given (a memory-mapped file; they must stay valid until dfgpu_ipc_close) — schema, dictionary batches, record
 * batches — without copying anything and without a GPU; dfgpu_ipc_read_batch lays Arrow C Data structs over the bytes of batch `i`,
 * the projected columns only (columns == NULL: all), undoes per-buffer compression (ZSTD; LZ4_FRAME when liblz4 is present) and
 * imports them like dfgpu_table_import (pinned side-stream copies).  Flat columns of the device's types, Utf8 / LargeUtf8 /
 * Utf8View and dictionary-encoded strings; nested columns, delta dictionaries, big-endian files are errors (the rule keeps the CPU
 * scan).  dfgpu_ipc_column: `format` is the Arrow C Data format string of the column's VALUE type ("l", "d:15,2", "u" ...). */
typedef struct dfgpu_ipc_s* dfgpu_ipc_t;
int dfgpu_ipc_open(const uint8_t* data, int64_t nbytes, dfgpu_ipc_t* out);
int dfgpu_ipc_close(dfgpu_ipc_t f);
int dfgpu_ipc_info(dfgpu_ipc_t f, int64_t* n_batches, int32_t* n_columns, int32_t* is_file_format);
int dfgpu_ipc_column(dfgpu_ipc_t f, int32_t i, const char** name, const char** format, int32_t* nullable, int32_t* dictionary_encoded);
int dfgpu_ipc_batch_rows(dfgpu_ipc_t f, int64_t i, int64_t* rows);
int dfgpu_ipc_read_batch(dfgpu_ipc_t f, int64_t i, const int* columns, int ncols, dfgpu_table_t* out);

/* ------------------------------------------------------ synthetic workload */

/* Deterministic TPC-H-shaped generator (counter-based PRNG, any slice
 * reproducible; bit-identical numpy mirror in datafusion_amd/tpch.py).  Rows
 * [order_begin, order_end) of `orders`, and the lineitem rows of exactly those orders.
 * Schemas follow benchmarks/src/tpch/mod.rs:93-122 restricted to the columns Q1/Q3 read. */
int dfgpu_tpch_orders(double scale_factor, int64_t order_begin, int64_t order_end, dfgpu_table_t* out);
int dfgpu_tpch_lineitem(double scale_factor, int64_t order_begin, int64_t order_end, int32_t float_money,
                        dfgpu_table_t* out);
int dfgpu_tpch_customer(double scale_factor, int64_t begin, int64_t end, dfgpu_table_t* out);

/* ----------------------------------------------------------------- metrics */

/* Operator metrics (MetricsSet / BaselineMetrics: output_rows, elapsed_compute — physical-expr-common/src/metrics/baseline.rs:53-75;
 * what `ExecutionPlan::metrics()` of a GPU node reports): counters of the CALLING THREAD since its last dfgpu_metrics_reset.
 * A shim resets them on the blocking thread that runs `execute(partition)`'s device work and reads them when the partition's
 * stream ends; nothing is shared between threads, so concurrent partitions do not mix. */
typedef struct dfgpu_metrics {
  int64_t calls;                 /* entry points entered */
  int64_t elapsed_ns;            /* host time spent inside entry points (launches, waits for results, copies) = elapsed_compute */
  int64_t kernel_ns;             /* device time of the launches this thread profiled (0 unless dfgpu_profile_enable(1)) */
  int64_t h2d_bytes, d2h_bytes;  /* bytes that crossed PCIe in dfgpu_table_import / dfgpu_table_export* */
  int64_t hbm_bytes_algorithmic; /* SURVEY 8(d) bytes of the launches: inputs read once + outputs written once */
  int64_t rows_in, rows_out;     /* rows of the table handles passed in / handed out */
} dfgpu_metrics;
int dfgpu_metrics_reset(void);
int dfgpu_metrics_get(dfgpu_metrics* out);

/* per-kernel HIP-event timing on the library stream (BaselineMetrics.elapsed_compute
 * analogue, physical-expr-common/src/metrics/baseline.rs:53-75) */
int dfgpu_profile_enable(int on);
int dfgpu_profile_reset(void);
/* number of distinct kernel names recorded since the last reset */
int dfgpu_profile_count(int* out);
typedef struct dfgpu_kernel_stat {
  char name[64];
  int64_t calls;
  double total_ms;
  int64_t algorithmic_bytes; /* bytes the launch had to move (inputs read once + outputs written once) */
} dfgpu_kernel_stat;
int dfgpu_profile_get(int i, dfgpu_kernel_stat* out);
/* (ABI 13) the launches recorded under one name since the last reset, in launch order: duration and algorithmic bytes of EACH — a plan
 * that runs the same kernel over unlike inputs (Q3: the counts pass over 450 M orders rows, then over 1.8 G lineitem rows) reports a
 * roofline per launch instead of the average of two different things.  Fills at most `capacity` entries of `ms` / `bytes`;
 * *out_n = how many launches there were (the 65536 most recent are kept). */
int dfgpu_profile_launches(const char* name, int64_t capacity, double* ms, int64_t* bytes, int64_t* out_n);

#ifdef __cplusplus
}
#endif
#endif /* DFGPU_H */
