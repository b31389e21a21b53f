/*
 * dforacle.c — CPU restatement of DataFusion's hot-path operators (see dforacle.h).
 * TEST INFRASTRUCTURE ONLY — never linked into or called by the product path.
 *
 * Plain C11 + __int128 (gcc).  Each function cites the reference file:line it follows
 * (paths relative to /root/reference/datafusion/).
 */
#define _GNU_SOURCE
#include "dforacle.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef __int128 i128;
typedef unsigned __int128 u128;

static inline int bit_get(const uint8_t* bits, int64_t i) { return (bits[i >> 3] >> (i & 7)) & 1; }
static inline void bit_set(uint8_t* bits, int64_t i) { bits[i >> 3] |= (uint8_t)(1u << (i & 7)); }
static inline int col_valid(const orc_col* c, int64_t i) { return c->valid == NULL || bit_get(c->valid, i); }

static inline int type_width(int type) {
  switch (type) {
    case ORC_I32: case ORC_U32: return 4;
    case ORC_I64: case ORC_U64: case ORC_F64: return 8;
    case ORC_I128: return 16;
    case ORC_U8: return 1;
  }
  return 0;
}

/* ---------------------------------------------------------------- hashing (K1) */

static inline uint64_t fmix64(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
  x ^= x >> 33;
  return x;
}
static inline uint64_t hash_u64(uint64_t v, uint64_t seed) { return fmix64(v ^ seed ^ 0x9E3779B97F4A7C15ULL); }

/* value of row i widened to (lo, hi); hash_utils.rs:258-276: floats hash by bit
 * pattern with -0.0 normalised to +0.0 */
static inline void load_words(const orc_col* c, int64_t i, uint64_t* lo, uint64_t* hi) {
  *hi = 0;
  switch (c->type) {
    case ORC_I32: *lo = (uint64_t)(int64_t)((const int32_t*)c->data)[i]; break;
    case ORC_U32: *lo = ((const uint32_t*)c->data)[i]; break;
    case ORC_I64: case ORC_U64: *lo = ((const uint64_t*)c->data)[i]; break;
    case ORC_F64: { uint64_t b = ((const uint64_t*)c->data)[i]; *lo = (b << 1) == 0 ? 0 : b; break; }
    case ORC_U8: *lo = ((const uint8_t*)c->data)[i]; break;
    case ORC_I128: *lo = ((const uint64_t*)c->data)[2 * i]; *hi = ((const uint64_t*)c->data)[2 * i + 1]; break;
    default: *lo = 0;
  }
}
static inline uint64_t hash_value(const orc_col* c, int64_t i, uint64_t seed) {
  uint64_t lo, hi; load_words(c, i, &lo, &hi);
  uint64_t h = hash_u64(lo, seed);
  if (c->type == ORC_I128) h = fmix64(hi ^ h);
  return h;
}

/* create_hashes (common/src/hash_utils.rs:1239-1252): column 0 hashes the value with
 * the operator's seed, column i>=1 re-seeds the hasher with the previous hash
 * (hash_array_primitive rehash=true, :322-327); NULLs leave the buffer untouched
 * (:331-343), the buffer starts at 0. */
void orc_create_hashes(const orc_col* cols, int ncols, int64_t n, uint64_t seed, uint64_t* out) {
  for (int64_t i = 0; i < n; i++) out[i] = 0;
  for (int c = 0; c < ncols; c++) {
    for (int64_t i = 0; i < n; i++) {
      if (!col_valid(&cols[c], i)) continue;
      out[i] = hash_value(&cols[c], i, c == 0 ? seed : out[i]);
    }
  }
}

/* -------------------------------------------------- key equality (K4) */

/* equal_rows_arr (physical-plan/src/joins/utils.rs:2191-2260): real key comparison
 * for a candidate pair; NULL==NULL only under NullEqualsNull (eq vs not_distinct). */
static inline int keys_equal(const orc_col* a, int64_t ia, const orc_col* b, int64_t ib, int nkeys, int null_equals_null) {
  for (int k = 0; k < nkeys; k++) {
    int va = col_valid(&a[k], ia), vb = col_valid(&b[k], ib);
    if (!va || !vb) {
      if (null_equals_null && !va && !vb) continue;
      return 0;
    }
    uint64_t alo, ahi, blo, bhi;
    load_words(&a[k], ia, &alo, &ahi);
    load_words(&b[k], ib, &blo, &bhi);
    if (a[k].type == ORC_F64) { /* arrow-ord eq on floats = total order equality on bits; keep raw bits */
      alo = ((const uint64_t*)a[k].data)[ia]; blo = ((const uint64_t*)b[k].data)[ib];
    }
    if (alo != blo || ahi != bhi) return 0;
  }
  return 1;
}
static inline int any_null_key(const orc_col* k, int nkeys, int64_t i) {
  for (int c = 0; c < nkeys; c++) if (!col_valid(&k[c], i)) return 1;
  return 0;
}

/* ------------------------------------------------------------- join maps (K2) */

/* JoinHashMap (joins/join_hash_map.rs:144-338): hashbrown HashTable<(hash, head)>
 * keyed by the 64-bit hash + `next` chain vector.  head/next hold row+1, 0 = end.
 * We restate the table as open addressing over distinct hash values. */
typedef struct {
  uint64_t* hashes; /* slot -> hash */
  uint64_t* heads;  /* slot -> row+1 (0 = empty slot) */
  uint64_t mask;
  uint64_t* next;   /* row -> next row+1 */
} join_hash_map;

static int jhm_init(join_hash_map* m, int64_t nrows) {
  uint64_t cap = 16;
  while (cap < (uint64_t)nrows * 2) cap <<= 1;
  m->mask = cap - 1;
  m->hashes = (uint64_t*)malloc(cap * 8);
  m->heads = (uint64_t*)calloc(cap, 8);
  m->next = (uint64_t*)calloc(nrows > 0 ? nrows : 1, 8);
  return (m->hashes && m->heads && m->next) ? 0 : -1;
}
static void jhm_free(join_hash_map* m) { free(m->hashes); free(m->heads); free(m->next); }

/* update_from_iter (join_hash_map.rs:307-338): insert (row, hash); an existing
 * entry's head becomes the new row and next[row] = old head.  The driver feeds rows
 * in REVERSE (joins/utils.rs:2127-2164, hash_join/exec.rs:638-676) so a chain walks
 * build rows in ascending order. */
static inline void jhm_insert(join_hash_map* m, uint64_t hash, int64_t row) {
  uint64_t s = hash & m->mask;
  for (;;) {
    if (m->heads[s] == 0) { m->hashes[s] = hash; m->heads[s] = (uint64_t)row + 1; return; }
    if (m->hashes[s] == hash) { m->next[row] = m->heads[s]; m->heads[s] = (uint64_t)row + 1; return; }
    s = (s + 1) & m->mask;
  }
}
static inline uint64_t jhm_find(const join_hash_map* m, uint64_t hash) {
  uint64_t s = hash & m->mask;
  for (;;) {
    if (m->heads[s] == 0) return 0;
    if (m->hashes[s] == hash) return m->heads[s];
    s = (s + 1) & m->mask;
  }
}

/* ArrayMap (joins/array_map.rs:103-236): data[key - min] = row+1, duplicates chained
 * through next[]; fill iterates in reverse to keep ascending chains. */
typedef struct {
  uint32_t* data;
  uint32_t* next; /* may be NULL when no duplicates */
  uint64_t offset; /* min key as u64 (wrapping) */
  uint64_t size;
} array_map;

static inline uint64_t key_as_u64(const orc_col* c, int64_t i) { /* ArrayMap::key_to_u64: `as u64` */
  switch (c->type) {
    case ORC_I32: return (uint64_t)(int64_t)((const int32_t*)c->data)[i];
    case ORC_U32: return ((const uint32_t*)c->data)[i];
    case ORC_U8: return ((const uint8_t*)c->data)[i];
    default: return ((const uint64_t*)c->data)[i];
  }
}
static inline int key_less(const orc_col* c, int64_t i, int64_t j) {
  switch (c->type) {
    case ORC_I32: return ((const int32_t*)c->data)[i] < ((const int32_t*)c->data)[j];
    case ORC_I64: return ((const int64_t*)c->data)[i] < ((const int64_t*)c->data)[j];
    case ORC_U32: return ((const uint32_t*)c->data)[i] < ((const uint32_t*)c->data)[j];
    case ORC_U8: return ((const uint8_t*)c->data)[i] < ((const uint8_t*)c->data)[j];
    default: return ((const uint64_t*)c->data)[i] < ((const uint64_t*)c->data)[j];
  }
}

/* try_create_array_map (hash_join/exec.rs:111-191): single integer key, no NULL keys
 * under NullEqualsNull, rows < u32::MAX, and (range < small_build_threshold or
 * rows/(range+1) > min_key_density).  Returns 1 if built. */
static int try_create_array_map(const orc_col* key, int nkeys, int null_equality,
                                int64_t small_build_threshold, double min_key_density, int force,
                                array_map* am) {
  if (nkeys != 1) return 0;
  if (!(key->type == ORC_I32 || key->type == ORC_I64 || key->type == ORC_U32 || key->type == ORC_U64 || key->type == ORC_U8)) return 0;
  int64_t n = key->n, nvalid = 0, imin = -1, imax = -1;
  for (int64_t i = 0; i < n; i++) {
    if (!col_valid(key, i)) { if (null_equality == ORC_NULL_EQUALS_NULL) return 0; continue; }
    if (imin < 0 || key_less(key, i, imin)) imin = i;
    if (imax < 0 || key_less(key, imax, i)) imax = i;
    nvalid++;
  }
  if (imin < 0) return 0; /* bounds are NULL (empty or all-NULL build side) */
  uint64_t mn = key_as_u64(key, imin), mx = key_as_u64(key, imax);
  uint64_t range = mx - mn; /* ArrayMap::calculate_range: wrapping_sub */
  if ((uint64_t)n >= 0xFFFFFFFFULL) return 0;
  if (range == UINT64_MAX) return 0;
  double dense_ratio = (double)n / (double)(range + 1);
  if (!force && range >= (uint64_t)small_build_threshold && dense_ratio <= min_key_density) return 0;
  if (range >= (1ULL << 33)) return 0; /* oracle allocation guard (not in the reference) */
  am->size = range + 1; am->offset = mn;
  am->data = (uint32_t*)calloc(am->size, 4);
  am->next = NULL;
  if (!am->data) return 0;
  for (int64_t i = n - 1; i >= 0; i--) { /* fill_data (array_map.rs:205-236) */
    if (!col_valid(key, i)) continue;
    uint64_t idx = key_as_u64(key, i) - mn;
    if (am->data[idx] != 0) {
      if (!am->next) am->next = (uint32_t*)calloc(n, 4);
      am->next[i] = am->data[idx];
    }
    am->data[idx] = (uint32_t)i + 1;
  }
  return 1;
}

/* ----------------------------------------------------------- hash join (K2-K5) */

typedef struct { int64_t* b; int64_t* p; uint8_t* m; int64_t n, cap; int want_mark; } pair_vec;
static int pv_push(pair_vec* v, int64_t b, int64_t p, int mark) {
  if (v->n == v->cap) {
    int64_t nc = v->cap ? v->cap * 2 : 1024;
    v->b = (int64_t*)realloc(v->b, nc * 8); v->p = (int64_t*)realloc(v->p, nc * 8);
    if (v->want_mark) v->m = (uint8_t*)realloc(v->m, nc);
    if (!v->b || !v->p) return -1;
    v->cap = nc;
  }
  v->b[v->n] = b; v->p[v->n] = p; if (v->want_mark) v->m[v->n] = (uint8_t)mark; v->n++;
  return 0;
}

/*
 * HashJoinExec semantics (physical-plan/src/joins/hash_join/exec.rs:560-751 doc,
 * collect_left_input :2569-2776, stream.rs:740-1000):
 *   build: NULL keys are not inserted under NullEqualsNothing; rows inserted in reverse.
 *   probe: per probe row (ascending), walk the chain (ascending build rows), keep pairs
 *          whose real keys are equal (equal_rows_arr).
 *   adjust_indices_by_join_type (joins/utils.rs:1432-1488): Inner/Left = matched pairs;
 *          Right/Full append unmatched probe rows (NULL build side); RightSemi = distinct
 *          matched probe rows; RightAnti = unmatched probe rows; RightMark = every probe
 *          row + mark; Left/Full/LeftSemi/LeftAnti/LeftMark emit from the visited bitmap
 *          after the probe side is exhausted (stream.rs:1002-).
 * The probe side is treated as one batch, so "per batch" orderings (Right/Full append
 * unmatched rows after the matched rows of the same batch) collapse to "matched pairs
 * first, then unmatched probe rows"; compare as multisets (join_fuzz.rs:914-925).
 */
int orc_hash_join(const orc_col* bk, const orc_col* pk, int nkeys, int join_type, int null_equality,
                  int mode, int64_t small_build_threshold, double min_key_density,
                  int64_t** out_build, int64_t** out_probe, uint8_t** out_mark, int64_t* out_n,
                  int* used_array_map) {
  int64_t nb = bk[0].n, np = pk[0].n;
  int nen = null_equality == ORC_NULL_EQUALS_NULL;
  array_map am; memset(&am, 0, sizeof am);
  join_hash_map hm; memset(&hm, 0, sizeof hm);
  int use_am = 0;
  if (mode != 1) use_am = try_create_array_map(&bk[0], nkeys, null_equality, small_build_threshold, min_key_density, mode == 2, &am);
  if (mode == 2 && !use_am) return -2;
  uint64_t* bh = NULL;
  if (!use_am) {
    if (jhm_init(&hm, nb)) return -1;
    bh = (uint64_t*)malloc((nb > 0 ? nb : 1) * 8);
    orc_create_hashes(bk, nkeys, nb, ORC_SEED_JOIN, bh);
    for (int64_t i = nb - 1; i >= 0; i--) {
      if (!nen && any_null_key(bk, nkeys, i)) continue; /* update_hash skips NULL keys: joins/utils.rs:2127-2164 */
      jhm_insert(&hm, bh[i], i);
    }
  }
  if (used_array_map) *used_array_map = use_am;

  uint8_t* visited = (uint8_t*)calloc(nb > 0 ? nb : 1, 1); /* visited bitmap, exec.rs:2713-2724 */
  pair_vec out; memset(&out, 0, sizeof out);
  out.want_mark = (join_type == ORC_JOIN_LEFT_MARK || join_type == ORC_JOIN_RIGHT_MARK);
  uint64_t* ph = NULL;
  if (!use_am) { ph = (uint64_t*)malloc((np > 0 ? np : 1) * 8); orc_create_hashes(pk, nkeys, np, ORC_SEED_JOIN, ph); }

  int probe_emits_pairs = (join_type == ORC_JOIN_INNER || join_type == ORC_JOIN_LEFT || join_type == ORC_JOIN_RIGHT || join_type == ORC_JOIN_FULL);
  int64_t n_unmatched_probe = 0; int64_t* unmatched_probe = NULL;
  if (join_type == ORC_JOIN_RIGHT || join_type == ORC_JOIN_FULL) unmatched_probe = (int64_t*)malloc((np > 0 ? np : 1) * 8);

  for (int64_t p = 0; p < np; p++) {
    int64_t nmatch = 0;
    int probe_null = any_null_key(pk, nkeys, p);
    if (!(probe_null && !nen)) {
      uint64_t cur = 0;
      if (use_am) { /* ArrayMap lookup, array_map.rs:247-380 */
        if (!probe_null) {
          uint64_t idx = key_as_u64(&pk[0], p) - am.offset;
          /* probe value must also be inside the build type's key range */
          if (idx < am.size) cur = am.data[idx];
        }
      } else {
        cur = jhm_find(&hm, ph[p]);
      }
      while (cur) { /* traverse_chain, joins/chain.rs:29-69 */
        int64_t b = (int64_t)cur - 1;
        if (keys_equal(bk, b, pk, p, nkeys, nen)) {
          nmatch++;
          visited[b] = 1;
          if (probe_emits_pairs) { if (pv_push(&out, b, p, 0)) return -1; }
        }
        cur = use_am ? (am.next ? am.next[b] : 0) : hm.next[b];
      }
    }
    switch (join_type) {
      case ORC_JOIN_RIGHT: case ORC_JOIN_FULL: if (!nmatch) unmatched_probe[n_unmatched_probe++] = p; break;
      case ORC_JOIN_RIGHT_SEMI: if (nmatch) pv_push(&out, -1, p, 0); break;  /* get_semi_indices :1628 */
      case ORC_JOIN_RIGHT_ANTI: if (!nmatch) pv_push(&out, -1, p, 0); break; /* get_anti_indices :1576 */
      case ORC_JOIN_RIGHT_MARK: pv_push(&out, -1, p, nmatch > 0); break;     /* get_mark_indices */
      default: break;
    }
  }
  for (int64_t i = 0; i < n_unmatched_probe; i++) pv_push(&out, -1, unmatched_probe[i], 0); /* append_right_indices :1506 */
  /* process_unmatched_build_batch (stream.rs:1002-): emit from the visited bitmap */
  for (int64_t b = 0; b < nb; b++) {
    switch (join_type) {
      case ORC_JOIN_LEFT: case ORC_JOIN_FULL: if (!visited[b]) pv_push(&out, b, -1, 0); break;
      case ORC_JOIN_LEFT_SEMI: if (visited[b]) pv_push(&out, b, -1, 0); break;
      case ORC_JOIN_LEFT_ANTI: if (!visited[b]) pv_push(&out, b, -1, 0); break;
      case ORC_JOIN_LEFT_MARK: pv_push(&out, b, -1, visited[b]); break;
      default: break;
    }
  }
  free(visited); free(unmatched_probe); free(bh); free(ph);
  if (use_am) { free(am.data); free(am.next); } else jhm_free(&hm);
  *out_build = out.b; *out_probe = out.p; *out_n = out.n;
  if (out_mark) *out_mark = out.m; else free(out.m);
  return 0;
}
void orc_free(void* p) { free(p); }

/* ------------------------------------------ CPU baseline: partitioned inner join */

/*
 * The plan DataFusion runs for an inner equi-join with target_partitions = T
 * (physical-optimizer EnsureRequirements + JoinSelection; plan shape pinned in
 * sqllogictest/test_files/tpch/plans/q3.slt.part:60-76):
 *   RepartitionExec Hash([key], T) on both inputs -> HashJoinExec mode=Partitioned.
 * Restated: (1) BatchPartitioner::partition_iter (repartition/mod.rs:1111-1150) splits
 * every 8192-row batch by hash % T with seed 0 and copies the rows (`take`);
 * (2) per partition: collect_left_input builds JoinHashMap over the whole build
 * partition (ArrayMap is not chosen: density drops below 0.15 once T >= 2 for the
 * TPC-H key layout, hash_join/exec.rs:590-605); (3) probe in 8192-row batches:
 * create_hashes, lookup, equal_rows check, emit (build_idx, probe_idx) pairs.
 * One thread per partition, like one tokio task per partition.
 */
int orc_partitioned_inner_join_i64(const int64_t* bkeys, int64_t nb, const int64_t* pkeys, int64_t np,
                                   int nthreads, int64_t* out_pairs, uint64_t* out_checksum) {
  if (nthreads < 1) nthreads = 1;
  const int T = nthreads;
  const int64_t BATCH = 8192;
  /* phase 1: repartition both sides (parallel over input ranges, then concatenated per partition) */
  int64_t* bcount = (int64_t*)calloc((size_t)T * T + T, 8);
  int64_t* pcount = (int64_t*)calloc((size_t)T * T + T, 8);
  /* count pass: thread t handles input slice t */
#pragma omp parallel for num_threads(T) schedule(static, 1)
  for (int t = 0; t < T; t++) {
    int64_t lo = nb * t / T, hi = nb * (t + 1) / T;
    for (int64_t i = lo; i < hi; i++) bcount[(size_t)t * T + hash_u64((uint64_t)bkeys[i], ORC_SEED_REPARTITION) % (uint64_t)T]++;
    lo = np * t / T; hi = np * (t + 1) / T;
    for (int64_t i = lo; i < hi; i++) pcount[(size_t)t * T + hash_u64((uint64_t)pkeys[i], ORC_SEED_REPARTITION) % (uint64_t)T]++;
  }
  /* offsets: partition-major, slice-minor (keeps input order inside a partition) */
  int64_t* boff = (int64_t*)malloc((size_t)T * T * 8), *poff = (int64_t*)malloc((size_t)T * T * 8);
  int64_t* bstart = (int64_t*)malloc((T + 1) * 8), *pstart = (int64_t*)malloc((T + 1) * 8);
  int64_t accb = 0, accp = 0;
  for (int part = 0; part < T; part++) {
    bstart[part] = accb; pstart[part] = accp;
    for (int t = 0; t < T; t++) { boff[(size_t)t * T + part] = accb; accb += bcount[(size_t)t * T + part]; poff[(size_t)t * T + part] = accp; accp += pcount[(size_t)t * T + part]; }
  }
  bstart[T] = accb; pstart[T] = accp;
  int64_t* bk2 = (int64_t*)malloc((nb > 0 ? nb : 1) * 8), *bi2 = (int64_t*)malloc((nb > 0 ? nb : 1) * 8);
  int64_t* pk2 = (int64_t*)malloc((np > 0 ? np : 1) * 8), *pi2 = (int64_t*)malloc((np > 0 ? np : 1) * 8);
#pragma omp parallel for num_threads(T) schedule(static, 1)
  for (int t = 0; t < T; t++) {
    int64_t lo = nb * t / T, hi = nb * (t + 1) / T;
    for (int64_t i = lo; i < hi; i++) { int part = (int)(hash_u64((uint64_t)bkeys[i], ORC_SEED_REPARTITION) % (uint64_t)T); int64_t o = boff[(size_t)t * T + part]++; bk2[o] = bkeys[i]; bi2[o] = i; }
    lo = np * t / T; hi = np * (t + 1) / T;
    for (int64_t i = lo; i < hi; i++) { int part = (int)(hash_u64((uint64_t)pkeys[i], ORC_SEED_REPARTITION) % (uint64_t)T); int64_t o = poff[(size_t)t * T + part]++; pk2[o] = pkeys[i]; pi2[o] = i; }
  }
  int64_t total = 0; uint64_t checksum = 0;
  /* phase 2: one build + probe per partition */
#pragma omp parallel for num_threads(T) schedule(static, 1) reduction(+ : total, checksum)
  for (int part = 0; part < T; part++) {
    int64_t b0 = bstart[part], nbp = bstart[part + 1] - b0;
    int64_t p0 = pstart[part], npp = pstart[part + 1] - p0;
    join_hash_map hm; jhm_init(&hm, nbp);
    for (int64_t i = nbp - 1; i >= 0; i--) jhm_insert(&hm, hash_u64((uint64_t)bk2[b0 + i], ORC_SEED_JOIN), i);
    uint64_t hashes[8192];
    for (int64_t s = 0; s < npp; s += BATCH) {
      int64_t e = s + BATCH < npp ? s + BATCH : npp;
      for (int64_t i = s; i < e; i++) hashes[i - s] = hash_u64((uint64_t)pk2[p0 + i], ORC_SEED_JOIN);
      for (int64_t i = s; i < e; i++) {
        uint64_t cur = jhm_find(&hm, hashes[i - s]);
        while (cur) {
          int64_t b = (int64_t)cur - 1;
          if (bk2[b0 + b] == pk2[p0 + i]) { total++; checksum += (uint64_t)bi2[b0 + b] * 0x9E3779B97F4A7C15ULL ^ (uint64_t)pi2[p0 + i]; }
          cur = hm.next[b];
        }
      }
    }
    jhm_free(&hm);
  }
  free(bcount); free(pcount); free(boff); free(poff); free(bstart); free(pstart);
  free(bk2); free(bi2); free(pk2); free(pi2);
  if (out_pairs) *out_pairs = total;
  if (out_checksum) *out_checksum = checksum;
  return 0;
}

/* The same plan with TPC-H Q3's payload (bench.py's cpu_baseline leg, SURVEY 8d "CPU path timed beside it"): what the GPU leg
 * of the benchmark does, as the reference does it.
 *   RepartitionExec(Hash) moves EVERY column of both inputs (BatchPartitioner::partition_iter -> take of each column,
 *   repartition/mod.rs:1111-1150,1237): build {o_orderkey i64, o_orderdate i32, o_shippriority i32}, probe {l_orderkey i64,
 *   l_extendedprice i128, l_discount i128};
 *   HashJoinExec(Partitioned) per partition: JoinHashMap over the partition's build keys (update_from_iter), probe in
 *   8192-row batches (lookup_join_hashmap -> equal_rows_arr), and build_batch_from_indices (joins/utils.rs:1332-1386) = arrow
 *   `take` of the two build payload columns by the matched build indices and of the three probe columns by the probe indices
 *   into a fresh output batch (48 bytes per row).  The reference hands each output batch to the next operator and drops it; here
 *   the batch is folded into a checksum (one pass over a cache-resident batch) and its buffers are reused.
 * out_rows = joined rows; out_checksum = wrapping sum over output rows of
 *   (o_orderdate + 3 * o_shippriority + 5 * l_orderkey + 7 * low64(l_extendedprice) + 11 * low64(l_discount)) — order independent. */
int orc_partitioned_q3_join(const int64_t* bkeys, const int32_t* bdate, const int32_t* bprio, int64_t nb, const int64_t* pkeys,
                            const i128* pprice, const i128* pdisc, int64_t np, int nthreads, int64_t* out_rows, uint64_t* out_checksum) {
  if (nthreads < 1) nthreads = 1;
  const int T = nthreads;
  const int64_t BATCH = 8192;
  int64_t* bcount = (int64_t*)calloc((size_t)T * T + T, 8);
  int64_t* pcount = (int64_t*)calloc((size_t)T * T + T, 8);
#pragma omp parallel for num_threads(T) schedule(static, 1)
  for (int t = 0; t < T; t++) {
    int64_t lo = nb * t / T, hi = nb * (t + 1) / T;
    for (int64_t i = lo; i < hi; i++) bcount[(size_t)t * T + hash_u64((uint64_t)bkeys[i], ORC_SEED_REPARTITION) % (uint64_t)T]++;
    lo = np * t / T; hi = np * (t + 1) / T;
    for (int64_t i = lo; i < hi; i++) pcount[(size_t)t * T + hash_u64((uint64_t)pkeys[i], ORC_SEED_REPARTITION) % (uint64_t)T]++;
  }
  int64_t* boff = (int64_t*)malloc((size_t)T * T * 8), *poff = (int64_t*)malloc((size_t)T * T * 8);
  int64_t* bstart = (int64_t*)malloc((T + 1) * 8), *pstart = (int64_t*)malloc((T + 1) * 8);
  int64_t accb = 0, accp = 0;
  for (int part = 0; part < T; part++) {
    bstart[part] = accb; pstart[part] = accp;
    for (int t = 0; t < T; t++) { boff[(size_t)t * T + part] = accb; accb += bcount[(size_t)t * T + part]; poff[(size_t)t * T + part] = accp; accp += pcount[(size_t)t * T + part]; }
  }
  bstart[T] = accb; pstart[T] = accp;
  const size_t nb1 = (size_t)(nb > 0 ? nb : 1), np1 = (size_t)(np > 0 ? np : 1);
  int64_t* bk2 = (int64_t*)malloc(nb1 * 8); int32_t* bd2 = (int32_t*)malloc(nb1 * 4); int32_t* bp2 = (int32_t*)malloc(nb1 * 4);
  int64_t* pk2 = (int64_t*)malloc(np1 * 8); i128* pp2 = (i128*)malloc(np1 * 16); i128* pd2 = (i128*)malloc(np1 * 16);
  if (!bk2 || !bd2 || !bp2 || !pk2 || !pp2 || !pd2) return -1;
#pragma omp parallel for num_threads(T) schedule(static, 1)
  for (int t = 0; t < T; t++) {
    int64_t lo = nb * t / T, hi = nb * (t + 1) / T;
    for (int64_t i = lo; i < hi; i++) {
      int part = (int)(hash_u64((uint64_t)bkeys[i], ORC_SEED_REPARTITION) % (uint64_t)T);
      int64_t o = boff[(size_t)t * T + part]++;
      bk2[o] = bkeys[i]; bd2[o] = bdate[i]; bp2[o] = bprio[i];
    }
    lo = np * t / T; hi = np * (t + 1) / T;
    for (int64_t i = lo; i < hi; i++) {
      int part = (int)(hash_u64((uint64_t)pkeys[i], ORC_SEED_REPARTITION) % (uint64_t)T);
      int64_t o = poff[(size_t)t * T + part]++;
      pk2[o] = pkeys[i]; pp2[o] = pprice[i]; pd2[o] = pdisc[i];
    }
  }
  int64_t total = 0; uint64_t checksum = 0;
#pragma omp parallel for num_threads(T) schedule(static, 1) reduction(+ : total, checksum)
  for (int part = 0; part < T; part++) {
    int64_t b0 = bstart[part], nbp = bstart[part + 1] - b0;
    int64_t p0 = pstart[part], npp = pstart[part + 1] - p0;
    join_hash_map hm; jhm_init(&hm, nbp);
    for (int64_t i = nbp - 1; i >= 0; i--) jhm_insert(&hm, hash_u64((uint64_t)bk2[b0 + i], ORC_SEED_JOIN), i);
    uint64_t* hashes = (uint64_t*)malloc(BATCH * 8);
    /* pairs of one probe batch; an FK -> PK probe has one match per row, M:N batches grow */
    int64_t cap = BATCH * 2, *ib = (int64_t*)malloc(cap * 8), *ip = (int64_t*)malloc(cap * 8);
    int32_t* o_date = (int32_t*)malloc(cap * 4), *o_prio = (int32_t*)malloc(cap * 4);
    int64_t* o_key = (int64_t*)malloc(cap * 8); i128* o_price = (i128*)malloc(cap * 16), *o_disc = (i128*)malloc(cap * 16);
    for (int64_t s0 = 0; s0 < npp; s0 += BATCH) {
      int64_t e = s0 + BATCH < npp ? s0 + BATCH : npp, m = 0;
      for (int64_t i = s0; i < e; i++) hashes[i - s0] = hash_u64((uint64_t)pk2[p0 + i], ORC_SEED_JOIN);
      for (int64_t i = s0; i < e; i++) {
        uint64_t cur = jhm_find(&hm, hashes[i - s0]);
        while (cur) {
          int64_t b = (int64_t)cur - 1;
          if (bk2[b0 + b] == pk2[p0 + i]) {
            if (m == cap) {
              cap *= 2;
              ib = (int64_t*)realloc(ib, cap * 8); ip = (int64_t*)realloc(ip, cap * 8);
              o_date = (int32_t*)realloc(o_date, cap * 4); o_prio = (int32_t*)realloc(o_prio, cap * 4);
              o_key = (int64_t*)realloc(o_key, cap * 8); o_price = (i128*)realloc(o_price, cap * 16); o_disc = (i128*)realloc(o_disc, cap * 16);
            }
            ib[m] = b; ip[m] = i; m++;
          }
          cur = hm.next[b];
        }
      }
      /* build_batch_from_indices: one take per output column */
      for (int64_t j = 0; j < m; j++) o_date[j] = bd2[b0 + ib[j]];
      for (int64_t j = 0; j < m; j++) o_prio[j] = bp2[b0 + ib[j]];
      for (int64_t j = 0; j < m; j++) o_key[j] = pk2[p0 + ip[j]];
      for (int64_t j = 0; j < m; j++) o_price[j] = pp2[p0 + ip[j]];
      for (int64_t j = 0; j < m; j++) o_disc[j] = pd2[p0 + ip[j]];
      uint64_t c = 0;
      for (int64_t j = 0; j < m; j++)
        c += (uint64_t)(int64_t)o_date[j] + 3u * (uint64_t)(int64_t)o_prio[j] + 5u * (uint64_t)o_key[j] + 7u * (uint64_t)o_price[j] + 11u * (uint64_t)o_disc[j];
      checksum += c;
      total += m;
    }
    free(hashes); free(ib); free(ip); free(o_date); free(o_prio); free(o_key); free(o_price); free(o_disc);
    jhm_free(&hm);
  }
  free(bcount); free(pcount); free(boff); free(poff); free(bstart); free(pstart);
  free(bk2); free(bd2); free(bp2); free(pk2); free(pp2); free(pd2);
  if (out_rows) *out_rows = total;
  if (out_checksum) *out_checksum = checksum;
  return 0;
}

/* ------------------------------------------------------------------ filter (K8) */

/* filter_record_batch with a BooleanArray predicate (filter.rs:1339-1362; arrow-select
 * `filter`: rows whose predicate is NULL are dropped).  Order preserved. */
int64_t orc_filter_indices(const uint8_t* mask, const uint8_t* mask_valid, int64_t n, int64_t* out_idx) {
  int64_t k = 0;
  for (int64_t i = 0; i < n; i++)
    if (bit_get(mask, i) && (mask_valid == NULL || bit_get(mask_valid, i))) out_idx[k++] = i;
  return k;
}

/* ------------------------------------------------------------ expressions (K9) */

#define ARITH_LOOP(T, UT)                                                                     \
  {                                                                                            \
    const T* pa = (const T*)a; const T* pb = (const T*)b; T* po = (T*)out;                     \
    for (int64_t i = 0; i < n; i++) {                                                          \
      UT x = (UT)pa[a_scalar ? 0 : i], y = (UT)pb[b_scalar ? 0 : i];                           \
      po[i] = (T)(op == 0 ? x + y : op == 1 ? x - y : x * y);                                  \
    }                                                                                          \
  }
/* arrow-arith add_wrapping / sub_wrapping / mul_wrapping as dispatched from
 * BinaryExpr::evaluate (physical-expr/src/expressions/binary.rs:625-637).  Integer and
 * Decimal128 arithmetic wraps (two's complement); Float64 is IEEE. */
int orc_arith(int op, int type, const void* a, int a_scalar, const void* b, int b_scalar, int64_t n, void* out) {
  switch (type) {
    case ORC_I32: ARITH_LOOP(int32_t, uint32_t); return 0;
    case ORC_I64: ARITH_LOOP(int64_t, uint64_t); return 0;
    case ORC_I128: ARITH_LOOP(i128, u128); return 0;
    case ORC_F64: {
      const double* pa = (const double*)a; const double* pb = (const double*)b; double* po = (double*)out;
      for (int64_t i = 0; i < n; i++) { double x = pa[a_scalar ? 0 : i], y = pb[b_scalar ? 0 : i]; po[i] = op == 0 ? x + y : op == 1 ? x - y : x * y; }
      return 0;
    }
  }
  return -1;
}

#define CMP_LOOP(T)                                                                            \
  {                                                                                            \
    const T* pa = (const T*)a; const T* pb = (const T*)b;                                      \
    for (int64_t i = 0; i < n; i++) {                                                          \
      T x = pa[a_scalar ? 0 : i], y = pb[b_scalar ? 0 : i];                                    \
      int r = op == 0 ? x == y : op == 1 ? x != y : op == 2 ? x < y : op == 3 ? x <= y : op == 4 ? x > y : x >= y; \
      if (r) bit_set(out_bits, i);                                                             \
    }                                                                                          \
  }
/* f64 total order key (arrow-ord cmp uses total_cmp for floats) */
static inline int64_t f64_total_key(double d) { int64_t b; memcpy(&b, &d, 8); return b ^ (int64_t)((uint64_t)(b >> 63) >> 1); }
/* apply_cmp -> arrow-ord cmp::{eq,neq,lt,lt_eq,gt,gt_eq} (physical-expr-common/src/datum.rs:60-100) */
int orc_cmp(int op, int type, const void* a, int a_scalar, const void* b, int b_scalar, int64_t n, uint8_t* out_bits) {
  memset(out_bits, 0, (size_t)((n + 7) / 8));
  switch (type) {
    case ORC_I32: CMP_LOOP(int32_t); return 0;
    case ORC_I64: CMP_LOOP(int64_t); return 0;
    case ORC_I128: CMP_LOOP(i128); return 0;
    case ORC_U8: CMP_LOOP(uint8_t); return 0;
    case ORC_U32: CMP_LOOP(uint32_t); return 0;
    case ORC_U64: CMP_LOOP(uint64_t); return 0;
    case ORC_F64: {
      const double* pa = (const double*)a; const double* pb = (const double*)b;
      for (int64_t i = 0; i < n; i++) {
        int64_t x = f64_total_key(pa[a_scalar ? 0 : i]), y = f64_total_key(pb[b_scalar ? 0 : i]);
        int r = op == 0 ? x == y : op == 1 ? x != y : op == 2 ? x < y : op == 3 ? x <= y : op == 4 ? x > y : x >= y;
        if (r) bit_set(out_bits, i);
      }
      return 0;
    }
  }
  return -1;
}

int orc_decimal_rescale_up(const void* a, int64_t n, int k, void* out) {
  u128 m = 1; for (int i = 0; i < k; i++) m *= 10;
  const i128* pa = (const i128*)a; i128* po = (i128*)out;
  for (int64_t i = 0; i < n; i++) po[i] = (i128)((u128)pa[i] * m);
  return 0;
}
int orc_cast_to_i128(int type, const void* a, int64_t n, void* out) {
  i128* po = (i128*)out;
  switch (type) {
    case ORC_I32: for (int64_t i = 0; i < n; i++) po[i] = ((const int32_t*)a)[i]; return 0;
    case ORC_I64: for (int64_t i = 0; i < n; i++) po[i] = ((const int64_t*)a)[i]; return 0;
    case ORC_U8: for (int64_t i = 0; i < n; i++) po[i] = ((const uint8_t*)a)[i]; return 0;
    case ORC_I128: memcpy(out, a, (size_t)n * 16); return 0;
  }
  return -1;
}

/* ------------------------------------------------------------ group by (K6/K7) */

/* GroupValuesPrimitive::intern (aggregates/group_values/single_group_by/primitive.rs:138-179)
 * and GroupValuesColumn::vectorized_intern (multi_group_by/mod.rs:455-520): hash the
 * key, probe HashTable<(gid, hash)>, compare real key values, append new group.
 * Group ids are dense, in first-seen order (group_values/mod.rs:88-92); NULL is a
 * group (primitive.rs: `null_group`). */
int64_t orc_group_intern(const orc_col* keys, int nkeys, int64_t n, int64_t* gids, int64_t* first_row) {
  uint64_t cap = 16; while (cap < (uint64_t)n * 2) cap <<= 1;
  uint64_t mask = cap - 1;
  int64_t* slots = (int64_t*)malloc(cap * 8); /* slot -> gid, -1 empty */
  uint64_t* shash = (uint64_t*)malloc(cap * 8);
  for (uint64_t i = 0; i < cap; i++) slots[i] = -1;
  int64_t ngroups = 0;
  for (int64_t i = 0; i < n; i++) {
    uint64_t h = 0;
    for (int c = 0; c < nkeys; c++) {
      if (col_valid(&keys[c], i)) h = hash_value(&keys[c], i, c == 0 ? ORC_SEED_AGG : h);
      else h = fmix64(h ^ 0x6E756C6CULL); /* stable hash for NULL: hash_null, hash_utils.rs:214-229 */
    }
    uint64_t s = h & mask;
    for (;;) {
      if (slots[s] < 0) { slots[s] = ngroups; shash[s] = h; first_row[ngroups] = i; gids[i] = ngroups++; break; }
      if (shash[s] == h && keys_equal(keys, first_row[slots[s]], keys, i, nkeys, 1)) { gids[i] = slots[s]; break; }
      s = (s + 1) & mask;
    }
  }
  free(slots); free(shash);
  return ngroups;
}

#define ACC_LOOP(T, UT, INIT_MIN, INIT_MAX)                                                    \
  {                                                                                            \
    const T* v = (const T*)values->data; T* o = (T*)out;                                       \
    for (int64_t g = 0; g < ngroups; g++) o[g] = op == 0 ? (T)0 : op == 1 ? INIT_MAX : INIT_MIN; \
    for (int64_t i = 0; i < n; i++) {                                                          \
      if (sel && !sel[i]) continue;                                                            \
      if (!col_valid(values, i)) continue;                                                     \
      int64_t g = gids[i]; out_seen[g] = 1;                                                    \
      if (op == 0) o[g] = (T)((UT)o[g] + (UT)v[i]);                                            \
      else if (op == 1) { if (v[i] < o[g]) o[g] = v[i]; }                                      \
      else { if (v[i] > o[g]) o[g] = v[i]; }                                                   \
    }                                                                                          \
  }
/* PrimitiveGroupsAccumulator::update_batch (functions-aggregate-common/src/aggregate/
 * groups_accumulator/prim_op.rs:89-118) over NullState::accumulate (accumulate.rs:164-190,
 * 373-470): for each non-null, filter-passing row: values[gid] = op(values[gid], x).
 * SUM = add_wrapping (functions-aggregate/src/sum.rs:308-320); Float64 SUM adds in row
 * order; COUNT counts non-null rows (count.rs:631-639). */
int orc_accumulate(int op, const orc_col* values, const int64_t* gids, int64_t ngroups,
                   const uint8_t* sel, void* out, uint8_t* out_seen) {
  int64_t n = values->n;
  memset(out_seen, 0, (size_t)ngroups);
  if (op == 3) {
    int64_t* o = (int64_t*)out; memset(o, 0, (size_t)ngroups * 8);
    for (int64_t i = 0; i < n; i++) { if (sel && !sel[i]) continue; if (!col_valid(values, i)) continue; o[gids[i]]++; out_seen[gids[i]] = 1; }
    for (int64_t g = 0; g < ngroups; g++) out_seen[g] = 1; /* COUNT is never NULL */
    return 0;
  }
  switch (values->type) {
    case ORC_I32: ACC_LOOP(int32_t, uint32_t, INT32_MIN, INT32_MAX); return 0;
    case ORC_I64: ACC_LOOP(int64_t, uint64_t, INT64_MIN, INT64_MAX); return 0;
    case ORC_I128: {
      const i128 mn = (i128)((u128)1 << 127), mx = ~mn;
      ACC_LOOP(i128, u128, mn, mx); return 0;
    }
    case ORC_F64: {
      const double* v = (const double*)values->data; double* o = (double*)out;
      for (int64_t g = 0; g < ngroups; g++) o[g] = op == 0 ? 0.0 : op == 1 ? INFINITY : -INFINITY;
      for (int64_t i = 0; i < n; i++) {
        if (sel && !sel[i]) continue;
        if (!col_valid(values, i)) continue;
        int64_t g = gids[i]; out_seen[g] = 1;
        if (op == 0) o[g] += v[i];
        else if (op == 1) { if (v[i] < o[g]) o[g] = v[i]; }
        else { if (v[i] > o[g]) o[g] = v[i]; }
      }
      return 0;
    }
  }
  return -1;
}

/* DecimalAverager::avg (functions-aggregate-common/src/utils.rs:157-176):
 *   sum.mul_checked(target_mul.div_wrapping(sum_mul)) then div_wrapping(count),
 * target_mul = 10^target_scale, sum_mul = 10^sum_scale; truncating division. */
int orc_decimal_avg(const void* sums, const int64_t* counts, int64_t n, int sum_scale, int target_scale, void* out) {
  i128 mul = 1; for (int i = sum_scale; i < target_scale; i++) mul *= 10;
  const i128* s = (const i128*)sums; i128* o = (i128*)out;
  for (int64_t i = 0; i < n; i++) {
    i128 r;
    if (__builtin_mul_overflow(s[i], mul, &r)) return -1;
    o[i] = counts[i] ? r / (i128)counts[i] : 0;
  }
  return 0;
}

/* ------------------------------------------------------------- repartition (K10) */

/* BatchPartitioner::partition_iter, Hash arm (physical-plan/src/repartition/mod.rs:
 * 1111-1150): create_hashes with REPARTITION_RANDOM_STATE (seed 0, :650) then
 * hash % num_partitions (StrengthReducedU64 is only a faster modulo, :875-914). */
void orc_hash_partition(const orc_col* keys, int nkeys, int64_t n, int nparts, uint32_t* out_part) {
  uint64_t* h = (uint64_t*)malloc((n > 0 ? n : 1) * 8);
  orc_create_hashes(keys, nkeys, n, ORC_SEED_REPARTITION, h);
  for (int64_t i = 0; i < n; i++) out_part[i] = (uint32_t)(h[i] % (uint64_t)nparts);
  free(h);
}

/* -------------------------------------------------------------------- sort (K11) */

typedef struct { const orc_col* keys; const uint8_t* desc; const uint8_t* nulls_first; int nkeys; } sort_ctx;

static inline int cmp_values(const orc_col* c, int64_t i, int64_t j) {
  switch (c->type) {
    case ORC_I32: { int32_t x = ((const int32_t*)c->data)[i], y = ((const int32_t*)c->data)[j]; return (x > y) - (x < y); }
    case ORC_I64: { int64_t x = ((const int64_t*)c->data)[i], y = ((const int64_t*)c->data)[j]; return (x > y) - (x < y); }
    case ORC_I128: { i128 x = ((const i128*)c->data)[i], y = ((const i128*)c->data)[j]; return (x > y) - (x < y); }
    case ORC_F64: { int64_t x = f64_total_key(((const double*)c->data)[i]), y = f64_total_key(((const double*)c->data)[j]); return (x > y) - (x < y); }
    case ORC_U8: { uint8_t x = ((const uint8_t*)c->data)[i], y = ((const uint8_t*)c->data)[j]; return (x > y) - (x < y); }
    case ORC_U32: { uint32_t x = ((const uint32_t*)c->data)[i], y = ((const uint32_t*)c->data)[j]; return (x > y) - (x < y); }
    case ORC_U64: { uint64_t x = ((const uint64_t*)c->data)[i], y = ((const uint64_t*)c->data)[j]; return (x > y) - (x < y); }
  }
  return 0;
}
static int sort_cmp(const void* pa, const void* pb, void* vctx) {
  const sort_ctx* ctx = (const sort_ctx*)vctx;
  int64_t i = *(const int64_t*)pa, j = *(const int64_t*)pb;
  for (int k = 0; k < ctx->nkeys; k++) {
    const orc_col* c = &ctx->keys[k];
    int vi = col_valid(c, i), vj = col_valid(c, j);
    if (!vi || !vj) { /* SortOptions.nulls_first places NULLs independent of `descending` */
      if (!vi && !vj) continue;
      int r = !vi ? -1 : 1; /* NULL first */
      return ctx->nulls_first[k] ? r : -r;
    }
    int r = cmp_values(c, i, j);
    if (r) return ctx->desc[k] ? -r : r;
  }
  return (i > j) - (i < j); /* stable */
}
/* sort_batch -> arrow-ord lexsort_to_indices (physical-plan/src/sorts/sort.rs:894-914) */
int orc_lexsort(const orc_col* keys, const uint8_t* descending, const uint8_t* nulls_first, int nkeys, int64_t n, int64_t* out_idx) {
  sort_ctx ctx = {keys, descending, nulls_first, nkeys};
  for (int64_t i = 0; i < n; i++) out_idx[i] = i;
  qsort_r(out_idx, (size_t)n, 8, sort_cmp, &ctx);
  return 0;
}
