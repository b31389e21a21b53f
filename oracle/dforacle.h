/*
 * dforacle.h — CPU restatement of DataFusion's vectorized physical operators.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and
 * there only as the checker / the timed CPU baseline.  The product path is
 * libdfgpu.so (include/dfgpu.h) and never calls into this file.
 *
 * Every function cites the reference file:line (relative to /root/reference/) whose
 * algorithm it restates.  The reference is Rust and cannot be built in this image
 * (no cargo/rustc) so this is a restatement pinned against the golden vectors the
 * reference's own tests hold (tests/golden/, extracted by
 * tests/golden/extract_reference_goldens.py).
 *
 * Parity notes:
 *   - hashing: the reference hashes with foldhash 0.2.0 (Cargo.lock:3080), an
 *     un-vendored dependency.  No reference test pins a concrete hash value; hash
 *     choice only affects partition routing and emission order, both unordered by
 *     contract.  We use our own 64-bit mixer (orc_hash_*), the SAME
 *     one the HIP kernels use, so partition routing can be compared bit-exact.
 *     "parity unpinned" applies to hash VALUES only.
 */
#ifndef DFORACLE_H
#define DFORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* physical value types (Arrow fixed-width layouts, little endian) */
enum {
  ORC_I32 = 1,  /* Int32, Date32 */
  ORC_I64 = 2,  /* Int64, Timestamp, Decimal64 */
  ORC_I128 = 3, /* Decimal128 (16-byte LE two's complement) */
  ORC_F64 = 4,  /* Float64 */
  ORC_U8 = 5,   /* UInt8 (dictionary codes / packed 1-byte strings) */
  ORC_U32 = 6,
  ORC_U64 = 7,
};

/* column view: values + optional Arrow validity bitmap (LSB first, 1 = valid) */
typedef struct {
  int32_t type;
  int32_t _pad;
  int64_t n;
  const void* data;
  const uint8_t* valid; /* NULL = no nulls */
} orc_col;

/* JoinType (datafusion/common/src/join_type.rs) */
enum {
  ORC_JOIN_INNER = 0,
  ORC_JOIN_LEFT = 1,
  ORC_JOIN_RIGHT = 2,
  ORC_JOIN_FULL = 3,
  ORC_JOIN_LEFT_SEMI = 4,
  ORC_JOIN_RIGHT_SEMI = 5,
  ORC_JOIN_LEFT_ANTI = 6,
  ORC_JOIN_RIGHT_ANTI = 7,
  ORC_JOIN_LEFT_MARK = 8,
  ORC_JOIN_RIGHT_MARK = 9,
};
/* NullEquality (datafusion/common/src/null_equality.rs) */
enum { ORC_NULL_EQUALS_NOTHING = 0, ORC_NULL_EQUALS_NULL = 1 };

/* seeds: the reference deliberately uses different seeds for join tables
 * (hash_join/exec.rs:105-106), aggregation (aggregates/mod.rs:236-238) and
 * repartition (repartition/mod.rs:650) */
#define ORC_SEED_JOIN 0xA98409FE2C1A0E6CULL
#define ORC_SEED_AGG 0x51D7348D9B2F63A5ULL
#define ORC_SEED_REPARTITION 0ULL

/* K1: create_hashes (common/src/hash_utils.rs:1239-1252, 306-345) */
void orc_create_hashes(const orc_col* cols, int ncols, int64_t n, uint64_t seed, uint64_t* out);

/* K2..K5: hash join.  Result = index pairs in the reference's emission order;
 * -1 = NULL side.  *out_build / *out_probe are malloc'd (orc_free).
 * For *_MARK joins *out_mark (malloc'd, one byte per emitted row) holds the mark.
 * mode: 0 = follow the reference's gating (try_create_array_map, exec.rs:111-191),
 *       1 = force JoinHashMap, 2 = force ArrayMap (fails with -2 if not applicable).
 * returns 0 ok / <0 error; *used_array_map reports which structure was used. */
int orc_hash_join(const orc_col* build_keys, const orc_col* probe_keys, int nkeys,
                  int join_type, int null_equality, int mode,
                  int64_t phj_small_build_threshold, double phj_min_key_density,
                  int64_t** out_build, int64_t** out_probe, uint8_t** out_mark,
                  int64_t* out_n, int* used_array_map);
void orc_free(void* p);

/* partitioned (multi-threaded) inner join used for the CPU baseline:
 * PartitionMode::Partitioned (hash_join/exec.rs:1314-1324) with nparts = nthreads,
 * both sides hash-repartitioned (repartition/mod.rs:1111-1150) then one build +
 * probe per partition, 8192-row probe batches.  Returns matched pair count and a
 * checksum (sum of build_idx ^ probe_idx) instead of materialising pairs when
 * out arrays are NULL. */
int orc_partitioned_inner_join_i64(const int64_t* build_keys, int64_t nb,
                                   const int64_t* probe_keys, int64_t np, int nthreads,
                                   int64_t* out_pairs, uint64_t* out_checksum);
/* the same plan carrying TPC-H Q3's payload through RepartitionExec and build_batch_from_indices (bench.py cpu_baseline) */
int orc_partitioned_q3_join(const int64_t* bkeys, const int32_t* bdate, const int32_t* bprio, int64_t nb, const int64_t* pkeys,
                            const __int128* pprice, const __int128* pdisc, int64_t np, int nthreads, int64_t* out_rows, uint64_t* out_checksum);


/* K8: filter (physical-plan/src/filter.rs:1339-1362; arrow `filter` treats a NULL
 * predicate as false).  mask/mask_valid are bit-packed; writes selected row ids. */
int64_t orc_filter_indices(const uint8_t* mask, const uint8_t* mask_valid, int64_t n, int64_t* out_idx);

/* K9: arrow-arith wrapping kernels as called from
 * physical-expr/src/expressions/binary.rs:625-637.  op: 0 add 1 sub 2 mul.
 * a_scalar/b_scalar: treat operand as a length-1 broadcast datum (datum.rs:36-57). */
int orc_arith(int op, int type, const void* a, int a_scalar, const void* b, int b_scalar,
              int64_t n, void* out);
/* comparison (physical-expr-common/src/datum.rs:60-100): op 0 eq 1 ne 2 lt 3 le 4 gt 5 ge.
 * Output bit-packed booleans (value bits only; validity = AND of input validities
 * is handled by the caller). */
int orc_cmp(int op, int type, const void* a, int a_scalar, const void* b, int b_scalar,
            int64_t n, uint8_t* out_bits);
/* Decimal128 rescale by 10^k (k>=0) with wrapping multiply — the cast inserted by
 * type coercion for decimal add/sub (expr-common/src/type_coercion/binary.rs:400-470). */
int orc_decimal_rescale_up(const void* a, int64_t n, int k, void* out);
/* widen ints to i128 (CastExpr Int64 -> Decimal128(20,0), binary.rs:1257-1273) */
int orc_cast_to_i128(int type, const void* a, int64_t n, void* out);

/* K6/K6': group interning (group_values/single_group_by/primitive.rs:138-179,
 * multi_group_by/mod.rs:455-520): dense group ids in first-seen order.  NULL is a
 * group of its own.  first_row[g] = first input row of group g. */
int64_t orc_group_intern(const orc_col* keys, int nkeys, int64_t n, int64_t* gids, int64_t* first_row);

/* K7: accumulate (functions-aggregate-common/.../accumulate.rs:373-470, prim_op.rs:89-118).
 * op: 0 SUM (add_wrapping, sum.rs:308-320) 1 MIN 2 MAX 3 COUNT(non-null)
 * out: ngroups values of the value type (COUNT: int64); out_seen[g]=1 if any
 * non-null value was accumulated (NullState, accumulate.rs:164-190).
 * sel: optional byte-per-row filter (opt_filter). */
int orc_accumulate(int op, const orc_col* values, const int64_t* gids, int64_t ngroups,
                   const uint8_t* sel, void* out, uint8_t* out_seen);
/* AVG(Decimal128) finalisation: DecimalAverager::avg
 * (functions-aggregate-common/src/utils.rs:157-176): (sum * 10^(ts - ss)) / count,
 * truncating i128 division; returns -1 on overflow of the multiply. */
int orc_decimal_avg(const void* sums, const int64_t* counts, int64_t n, int sum_scale,
                    int target_scale, void* out);

/* K10: hash repartition (repartition/mod.rs:1111-1150): part = hash % nparts */
void orc_hash_partition(const orc_col* keys, int nkeys, int64_t n, int nparts, uint32_t* out_part);

/* K11: lexsort_to_indices (sorts/sort.rs:894-914).  Stable (ties keep input order;
 * the reference's tie order is unpinned, compare ties as sets). */
int orc_lexsort(const orc_col* keys, const uint8_t* descending, const uint8_t* nulls_first,
                int nkeys, int64_t n, int64_t* out_idx);

#ifdef __cplusplus
}
#endif
#endif
