// aggregate.hip — K6/K6'/K7/K7': AggregateExec (hash group-by) on device.
//
// Reference: AggregateHashTable::aggregate_batch_inner (physical-plan/src/aggregates/
// aggregate_hash_table/common.rs:205-236) = evaluate keys/args -> GroupValues::intern
// (group_values/single_group_by/primitive.rs:138-179, multi_group_by/mod.rs:455-520) ->
// GroupsAccumulator::update_batch / merge_batch per aggregate (functions-aggregate-common/src/
// aggregate/groups_accumulator/prim_op.rs:89-118, accumulate.rs:373-470).
//
// Device design (whole partition per update, not 8192-row batches):
//   intern   : open-addressing table in HBM whose slots hold a REPRESENTATIVE ROW id (row+1).
//              A slot is claimed with one agent-scope atomicCAS and lowered with atomicMin, so
//              the representative of a key is its first row and key comparison only ever reads
//              the immutable input columns (no inter-workgroup publish/consume hazards).
//              Representatives -> row bitmask -> popcount prefix (scan.hip) -> dense group ids in
//              FIRST-SEEN ORDER, exactly the reference's numbering (group_values/mod.rs:88-92).
//              The table starts small and is regrown on overflow (probe length guard).
//   update   : each row re-probes the (now read-only, cache-resident for low cardinality) table
//              to find its group id, then accumulates.  Low cardinality: LDS-privatised
//              accumulators per workgroup, replicated across lanes to spread same-address LDS
//              atomics, flushed once with global atomics.  High cardinality: global atomics.
//              SUM(Decimal128) = wrapping i128 add as two u64 atomics with carry (order-
//              independent => bit-exact, sum.rs:308-320); Float64 sums are atomics in arbitrary
//              order (1e-6 relative tolerance).
//   emit     : group key columns (gather of representative rows) + state or final columns;
//              AVG(Decimal128) = (sum * 10^(ts-ss)) / count truncating (DecimalAverager::avg,
//              functions-aggregate-common/src/utils.rs:157-176).
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "device.hpp"
#include "internal.hpp"
#include "rowprog_host.hpp"
#include "grouped.hpp"

namespace dfgpu {

void pack_bytes_to_bitmap(const uint8_t* bytes, int64_t n, uint64_t* words);

constexpr int MAX_AGGS = 16;
constexpr uint32_t PROBE_LIMIT = 256;  // longer probe sequence => table too full => regrow

// accumulator kinds
enum AccKind : int { ACC_SUM_I64 = 0, ACC_SUM_I128 = 1, ACC_SUM_F64 = 2, ACC_MIN_I64 = 3, ACC_MAX_I64 = 4, ACC_COUNT = 5, ACC_COUNT_STAR = 6 };
// how a value is loaded & widened
enum ValKind : int { VAL_I32 = 0, VAL_I64 = 1, VAL_I128 = 2, VAL_F64 = 3, VAL_U8 = 4, VAL_F64_ORDERED = 5, VAL_I32_TO_F64 = 6, VAL_I64_TO_F64 = 7, VAL_U32 = 8, VAL_U64 = 9 };

struct AccDesc {
  const void* values;          // input value column (null for COUNT(*))
  const uint64_t* valid;       // optional validity
  unsigned long long* acc_lo;  // [ngroups] low word / i64 / f64 bits / count
  unsigned long long* acc_hi;  // [ngroups] high word (i128 only)
  uint32_t* seen;              // [ngroups] non-null value seen (NullState)
  int kind;                    // AccKind
  int val;                     // ValKind
  int narrow;                  // the values fit 64 bits (their high word is the sign): a partial 128-bit sum may keep a 32-bit high word
};
struct AccSet {
  AccDesc a[MAX_AGGS];
  int n;
};

__device__ __forceinline__ int64_t f64_ordered(double d) {
  int64_t b = __double_as_longlong(d);
  return b ^ (int64_t)((uint64_t)(b >> 63) >> 1);
}
__host__ __device__ __forceinline__ double f64_from_ordered(int64_t k) {
  int64_t b = k ^ (int64_t)((uint64_t)(k >> 63) >> 1);
  double d;
#ifdef __HIP_DEVICE_COMPILE__
  d = __longlong_as_double(b);
#else
  std::memcpy(&d, &b, 8);
#endif
  return d;
}

// value of row i as (lo, hi) according to ValKind
__device__ __forceinline__ void load_value(const AccDesc& d, int64_t i, uint64_t& lo, uint64_t& hi) {
  hi = 0;
  switch (d.val) {
    case VAL_I32: { int64_t v = ((const int32_t*)d.values)[i]; lo = (uint64_t)v; hi = (uint64_t)(v >> 63); break; }
    case VAL_U32: lo = ((const uint32_t*)d.values)[i]; break;
    case VAL_I64: { int64_t v = ((const int64_t*)d.values)[i]; lo = (uint64_t)v; hi = (uint64_t)(v >> 63); break; }
    case VAL_U64: lo = ((const uint64_t*)d.values)[i]; break;
    case VAL_U8: lo = ((const uint8_t*)d.values)[i]; break;
    case VAL_I128: { const uint64_t* p = (const uint64_t*)d.values + 2 * i; lo = p[0]; hi = p[1]; break; }
    case VAL_F64: lo = ((const uint64_t*)d.values)[i]; break;
    case VAL_F64_ORDERED: lo = (uint64_t)f64_ordered(((const double*)d.values)[i]); break;
    case VAL_I32_TO_F64: lo = (uint64_t)__double_as_longlong((double)((const int32_t*)d.values)[i]); break;
    case VAL_I64_TO_F64: lo = (uint64_t)__double_as_longlong((double)((const int64_t*)d.values)[i]); break;
  }
}

// the values of U rows of one argument column at once: the ValKind is asked ONCE, outside the unrolled loop, so the U loads leave back to
// back (a switch per load puts every load in a basic block of its own with a wait behind it)
template <int U>
__device__ __forceinline__ void load_values_batch(int val, const void* values, const int64_t (&ii)[U], uint32_t live, uint64_t (&lo)[U], uint64_t (&hi)[U], int estride = 1) {
#pragma unroll
  for (int u = 0; u < U; u++) lo[u] = hi[u] = 0;
  switch (val) {
    case VAL_I32:
    case VAL_I32_TO_F64: {
      int32_t v[U];
#pragma unroll
      for (int u = 0; u < U; u++) v[u] = (live >> u) & 1u ? ((const int32_t*)values)[ii[u]] : 0;
#pragma unroll
      for (int u = 0; u < U; u++) {
        if (val == VAL_I32) { lo[u] = (uint64_t)(int64_t)v[u]; hi[u] = (uint64_t)((int64_t)v[u] >> 63); }
        else lo[u] = (uint64_t)__double_as_longlong((double)v[u]);
      }
      break;
    }
    case VAL_U32:
#pragma unroll
      for (int u = 0; u < U; u++) lo[u] = (live >> u) & 1u ? ((const uint32_t*)values)[ii[u]] : 0u;
      break;
    case VAL_U8:
#pragma unroll
      for (int u = 0; u < U; u++) lo[u] = (live >> u) & 1u ? ((const uint8_t*)values)[ii[u]] : 0u;
      break;
    case VAL_I128: {
      ulonglong2 v[U];
#pragma unroll
      for (int u = 0; u < U; u++) v[u] = (live >> u) & 1u ? ((const ulonglong2*)values)[ii[u]] : make_ulonglong2(0ull, 0ull);
#pragma unroll
      for (int u = 0; u < U; u++) { lo[u] = v[u].x; hi[u] = v[u].y; }
      break;
    }
    default: {   // the 64-bit kinds
      uint64_t v[U];
      if (estride > 1) {   // (records: `ii` counts 32-bit words, the value is two of them — 4-byte aligned)
        uint32_t a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
          a[u] = (live >> u) & 1u ? ((const uint32_t*)values)[ii[u]] : 0u;
          b[u] = (live >> u) & 1u ? ((const uint32_t*)values)[ii[u] + 1] : 0u;
        }
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = (uint64_t)a[u] | ((uint64_t)b[u] << 32);
      } else {
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = (live >> u) & 1u ? ((const uint64_t*)values)[ii[u]] : 0ull;
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        switch (val) {
          case VAL_I64: lo[u] = v[u]; hi[u] = (uint64_t)((int64_t)v[u] >> 63); break;
          case VAL_F64_ORDERED: lo[u] = (uint64_t)f64_ordered(__longlong_as_double((long long)v[u])); break;
          case VAL_I64_TO_F64: lo[u] = (uint64_t)__double_as_longlong((double)(int64_t)v[u]); break;
          default: lo[u] = v[u]; break;   // VAL_U64, VAL_F64
        }
      }
      break;
    }
  }
}

// one accumulation into (lo, hi) cells that may live in LDS or HBM
__device__ __forceinline__ void accumulate_cell(int kind, unsigned long long* lo_cell, unsigned long long* hi_cell, uint64_t lo, uint64_t hi) {
  switch (kind) {
    case ACC_SUM_I64: atomicAdd(lo_cell, (unsigned long long)lo); break;
    case ACC_SUM_I128: {
      unsigned long long old = atomicAdd(lo_cell, (unsigned long long)lo);
      unsigned long long carry = (old + lo) < old ? 1ull : 0ull;  // this add wrapped the low word
      if ((unsigned long long)hi + carry) atomicAdd(hi_cell, (unsigned long long)hi + carry);   // (non-negative values: nearly never)
      break;
    }
    case ACC_SUM_F64: atomicAdd(reinterpret_cast<double*>(lo_cell), __longlong_as_double((long long)lo)); break;
    case ACC_MIN_I64: atomicMin(reinterpret_cast<long long*>(lo_cell), (long long)lo); break;
    case ACC_MAX_I64: atomicMax(reinterpret_cast<long long*>(lo_cell), (long long)lo); break;
    default: atomicAdd(lo_cell, 1ull); break;  // COUNT / COUNT(*)
  }
}
__host__ __device__ __forceinline__ unsigned long long acc_identity(int kind) {
  if (kind == ACC_MIN_I64) return (unsigned long long)INT64_MAX;
  if (kind == ACC_MAX_I64) return (unsigned long long)INT64_MIN;
  return 0ull;
}

// ------------------------------------------------------------------------------ intern
struct InternCtx {
  KeySet keys;          // concatenated [existing group keys ; input keys]
  uint32_t* slots;      // representative row + 1, 0 = empty
  uint64_t mask;        // capacity - 1
  struct KeyedSlot* keyed;  // keyed table (below): packed key and smallest row of every slot; null = slots are compared through their rows
  // direct table (round 4): the key columns' value ranges multiply to a few thousand — the slot of a row IS its mixed-radix number
  // sum((value_c - dmin[c]) * dstride[c]); nothing is hashed, compared or probed (k_intern_claim_direct)
  int direct;           // number of slots of the direct table, 0 = a hash table
  long long dmin[MAX_KEYS];
  uint32_t dstride[MAX_KEYS];
  const uint8_t* dcode[MAX_KEYS];   // UInt8 columns: value -> its rank among the values that occur (256 bytes; null: value - dmin)
};
struct KeyedSlot {   // 16 bytes: one L2 request brings a slot's key and its row
  uint64_t key;      // KEY_EMPTY = none
  uint32_t row;      // smallest row + 1 so far, 0 = none yet
  uint32_t pad;
};

// Keyed table: key columns without NULLs that are <= 64 bits wide TOGETHER are interned as one word — their raw bits side by side,
// put together in registers — and the table keeps that word in the slot: a row hashes one word and compares it with the slot's,
// without going to the representative row's columns (one dependent random load less per row, and per key column).  Equal rows <=>
// equal words.  The one word that cannot be told from an empty slot (all ones: possible when the columns fill all 64 bits) has
// a slot of its own past the table's end (index capacity).
constexpr uint64_t KEY_EMPTY = ~0ull;
// (the columns here are 1, 4 or 8 bytes wide: their raw bits are read by width, R rows at a time so that a thread has R loads in
// flight per column)
template <int R>
__device__ __forceinline__ void packed_keys(const KeySet& ks, const int64_t (&i)[R], uint32_t live, uint64_t (&v)[R]) {
#pragma unroll
  for (int r = 0; r < R; r++) v[r] = 0;
  int shift = 0;
  for (int c = 0; c < ks.n; c++) {
    const void* p = ks.c[c].data;
    uint64_t x[R];
    if (ks.c[c].width == 1) {
#pragma unroll
      for (int r = 0; r < R; r++) x[r] = (live >> r) & 1u ? (uint64_t)((const uint8_t*)p)[i[r]] : 0ull;
    } else if (ks.c[c].width == 4) {
#pragma unroll
      for (int r = 0; r < R; r++) x[r] = (live >> r) & 1u ? (uint64_t)((const uint32_t*)p)[i[r]] : 0ull;
    } else {
#pragma unroll
      for (int r = 0; r < R; r++) x[r] = (live >> r) & 1u ? ((const uint64_t*)p)[i[r]] : 0ull;
    }
#pragma unroll
    for (int r = 0; r < R; r++) v[r] |= x[r] << shift;
    shift += ks.c[c].width * 8;
  }
}
__device__ __forceinline__ uint64_t packed_key(const KeySet& ks, int64_t i) {
  const int64_t ii[1] = {i};
  uint64_t v[1];
  packed_keys<1>(ks, ii, 1u, v);
  return v[0];
}
__device__ __forceinline__ uint64_t packed_key_slot(uint64_t k, uint64_t mask) { return fmix64(k ^ SEED_AGG) & mask; }

__device__ __forceinline__ uint64_t group_hash(const KeySet& ks, int64_t i) {
  uint64_t h = SEED_AGG;
  for (int c = 0; c < ks.n; c++) {
    if (ks.c[c].valid && !bit_at(ks.c[c].valid, i)) h = fmix64(h ^ 0x6E756C6CULL);  // NULL is a group value
    else h = hash_value(ks.c[c], i, h);
  }
  return h;
}

// GroupValues::intern, claim phase: every row finds or claims the slot of its key; the slot ends
// up holding the smallest row id of the key (first-seen representative).
// `row_mask` (optional): bit r covers concatenated row mask_offset + r; rows below mask_offset (the existing
// groups) are always interned.
// `row_slot` (optional): the slot every row ended at (0xFFFFFFFF: masked out, or skipped by the run compression below) — a later
// row -> group pass then reads slot_gid[row_slot[i]] instead of hashing and comparing the keys again (k_row_gids).
__global__ __launch_bounds__(BLOCK) void k_intern_claim(InternCtx c, int64_t n, int* overflow, const uint64_t* __restrict__ row_mask, int64_t mask_offset,
                                                       uint32_t* __restrict__ row_slot = nullptr) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    if (__hip_atomic_load(overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    if (row_slot) row_slot[i] = 0xFFFFFFFFu;
    if (row_mask && i >= mask_offset && !bit_at(row_mask, i - mask_offset)) continue;
    // Run compression: an input row whose key equals its (live) predecessor's never needs to claim — the predecessor
    // or, by induction, the first row of the run does, and that row is the group's smaller row id anyway.  Clustered
    // input (a join's output in probe order, a table stored by key) skips most of the random CAS traffic.
    if (i > mask_offset && (!row_mask || bit_at(row_mask, i - 1 - mask_offset)) && keys_equal(c.keys, i - 1, c.keys, i, true)) continue;
    uint64_t s = group_hash(c.keys, i) & c.mask;
    const uint32_t me = (uint32_t)i + 1u;
    uint32_t steps = 0;
    for (;;) {
      uint32_t cur = __hip_atomic_load(&c.slots[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (cur == 0u) {
        cur = atomicCAS(&c.slots[s], 0u, me);
        if (cur == 0u) break;  // claimed
      }
      if (cur == me) break;
      if (keys_equal(c.keys, (int64_t)cur - 1, c.keys, i, true)) {
        if (me < cur) atomicMin(&c.slots[s], me);
        break;
      }
      s = (s + 1) & c.mask;
      if (++steps > PROBE_LIMIT) {
        __hip_atomic_store(overflow, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
      }
    }
    if (row_slot) row_slot[i] = (uint32_t)s;
  }
}
// the same over a keyed table: find or claim the slot of the row's packed key, leave the smallest row there; every live row
// gets its slot into row_slot.  What bounds this pass is the number of random L2 requests and, per thread, the chain key -> slot:
// a slot is 16 bytes read at once (PLAIN loads — a stale copy can only show an emptier slot or a larger row than the true ones, and
// both are settled by the compare-and-swap that follows), and a thread works on U rows whose keys, then whose slots, are in flight
// together.  Measured over 600 M rows x (u8, u8, date32), 3817 groups: key and row in separate arrays read by atomic loads, one row
// at a time 7.1 ms; the same four rows at a time with their loads one after the other 9.2 ms; a table of the met keys in every
// workgroup's LDS (a row then costs LDS probes only) 7.1 - 8.1 ms — slow for its structure (one 1024-thread workgroup per CU, three
// barriers per 8 K rows), not for LDS: a random LDS read costs 10 - 27 clocks per wave and CU, a random L2 hit ~120
// (scripts/microbench/random_access.hip, profiles/r3_random_access.md).  By those numbers this pass's random read is worth 2.3 ms of
// its 6.5: the rest is the chain of dependent round trips a thread waits for (the key columns one after the other, then the slot).
__device__ __forceinline__ void keyed_load(const KeyedSlot* t, uint64_t s, uint64_t& key, uint32_t& row) {
  const uint4 e = *reinterpret_cast<const uint4*>(t + s);
  key = (uint64_t)e.x | ((uint64_t)e.y << 32);
  row = e.z;
}
// continues from the slot's loaded (key, row); false: the table is too full
__device__ __forceinline__ bool keyed_find_or_claim(const InternCtx& c, uint64_t k, uint64_t& s, uint64_t key, uint32_t& row) {
  if (k == KEY_EMPTY) return true;   // (the all-ones key owns the slot past the end: nothing to find)
  uint32_t steps = 0;
  for (;;) {
    if (key == KEY_EMPTY) {
      key = atomicCAS(reinterpret_cast<unsigned long long*>(&c.keyed[s].key), (unsigned long long)KEY_EMPTY, (unsigned long long)k);
      if (key == KEY_EMPTY) return true;   // claimed
    }
    if (key == k) return true;
    s = (s + 1) & c.mask;
    if (++steps > PROBE_LIMIT) return false;
    keyed_load(c.keyed, s, key, row);
  }
}
__device__ __forceinline__ void keyed_offer_row(const InternCtx& c, uint64_t s, uint32_t me, uint32_t old) {   // me = row + 1; old = the slot's row as loaded
  while (old == 0u || me < old) {
    const uint32_t prev = atomicCAS(&c.keyed[s].row, old, me);
    if (prev == old) break;
    old = prev;
  }
}
constexpr int KEYED_ROWS = 4;
__global__ __launch_bounds__(BLOCK) void k_intern_claim_keyed(InternCtx c, int64_t n, int* overflow, const uint64_t* __restrict__ row_mask, int64_t mask_offset,
                                                             uint32_t* __restrict__ row_slot) {
  constexpr int U = KEYED_ROWS;
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  for (int64_t i0 = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i0 < n; i0 += U * stride) {
    if (__hip_atomic_load(overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    int64_t rows[U];
    uint64_t k[U], s[U], key[U];
    uint32_t row[U];
    uint32_t live = 0;
#pragma unroll
    for (int u = 0; u < U; u++) {
      rows[u] = i0 + u * stride;
      if (rows[u] < n && !(row_mask && rows[u] >= mask_offset && !bit_at(row_mask, rows[u] - mask_offset))) live |= 1u << u;
    }
    packed_keys<U>(c.keys, rows, live, k);
#pragma unroll
    for (int u = 0; u < U; u++) s[u] = k[u] == KEY_EMPTY ? c.mask + 1 : packed_key_slot(k[u], c.mask);
#pragma unroll
    for (int u = 0; u < U; u++) {
      key[u] = 0;
      row[u] = 0;
      if ((live >> u) & 1u) keyed_load(c.keyed, s[u], key[u], row[u]);
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (rows[u] >= n) continue;
      if (!((live >> u) & 1u)) {
        if (row_slot) row_slot[rows[u]] = 0xFFFFFFFFu;
        continue;
      }
      if (!keyed_find_or_claim(c, k[u], s[u], key[u], row[u])) {
        __hip_atomic_store(overflow, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
      }
      keyed_offer_row(c, s[u], (uint32_t)rows[u] + 1u, row[u]);
      if (row_slot) row_slot[rows[u]] = (uint32_t)s[u];
    }
  }
}
__global__ __launch_bounds__(BLOCK) void k_keyed_clear(KeyedSlot* t, uint64_t n_slots) {
  for (uint64_t s = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; s < n_slots; s += (uint64_t)gridDim.x * BLOCK)
    *reinterpret_cast<uint4*>(t + s) = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u);
}
// the slots' rows as the array the column-by-column table keeps (what numbers the groups reads that)
__global__ __launch_bounds__(BLOCK) void k_keyed_rows(const KeyedSlot* __restrict__ t, uint64_t n_slots, uint32_t* __restrict__ slots) {
  for (uint64_t s = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; s < n_slots; s += (uint64_t)gridDim.x * BLOCK) slots[s] = t[s].row;
}
// table sizing from a sample: out[0] = occupied slots, out[1] = those whose representative row lies before `early`
__global__ __launch_bounds__(BLOCK) void k_count_slots(const uint32_t* __restrict__ slots, uint64_t capacity, uint32_t early, unsigned long long* __restrict__ out) {
  unsigned long long a = 0, b = 0;
  for (uint64_t s = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; s < capacity; s += (uint64_t)gridDim.x * BLOCK) {
    const uint32_t v = slots[s];
    a += v != 0u;
    b += v != 0u && v - 1u < early;
  }
  a = wave_sum(a);
  b = wave_sum(b);
  if (lane_id() == 0) {
    if (a) atomicAdd(&out[0], a);
    if (b) atomicAdd(&out[1], b);
  }
}
// representatives -> row bitmask
__global__ __launch_bounds__(BLOCK) void k_mark_reps(const uint32_t* __restrict__ slots, uint64_t capacity, unsigned long long* __restrict__ rep_mask) {
  for (uint64_t s = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; s < capacity; s += (uint64_t)gridDim.x * BLOCK) {
    uint32_t v = slots[s];
    if (v) atomicOr(&rep_mask[(v - 1) >> 6], 1ull << ((v - 1) & 63));
  }
}
// slot -> dense group id (rank of its representative among all representatives), gid -> rep row
__global__ __launch_bounds__(BLOCK) void k_slot_gids(const uint32_t* __restrict__ slots, uint64_t capacity, const uint64_t* __restrict__ rep_mask,
                                                     const uint64_t* __restrict__ prefix, uint32_t* __restrict__ slot_gid, int64_t* __restrict__ rep_row) {
  for (uint64_t s = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; s < capacity; s += (uint64_t)gridDim.x * BLOCK) {
    uint32_t v = slots[s];
    if (!v) continue;
    uint64_t r = v - 1;
    uint64_t below = rep_mask[r >> 6] & ((1ull << (r & 63)) - 1ull);
    uint32_t gid = (uint32_t)(prefix[r >> 6] + __popcll(below));
    slot_gid[s] = gid;
    rep_row[gid] = (int64_t)r;
  }
}

// ---- direct table: mixed-radix slot numbers from the key columns' value ranges
template <int R>
__device__ __forceinline__ void direct_slots(const InternCtx& c, const int64_t (&i)[R], uint32_t live, uint32_t (&v)[R]) {
#pragma unroll
  for (int r = 0; r < R; r++) v[r] = 0;
  for (int k = 0; k < c.keys.n; k++) {
    const void* p = c.keys.c[k].data;
    uint64_t x[R];
    switch (c.keys.c[k].type) {   // (all R loads of a column in flight together)
      case DFGPU_UINT8:
#pragma unroll
        for (int r = 0; r < R; r++) x[r] = (live >> r) & 1u ? (uint64_t)((const uint8_t*)p)[i[r]] : (uint64_t)c.dmin[k];
        if (c.dcode[k]) {
#pragma unroll
          for (int r = 0; r < R; r++) x[r] = (uint64_t)c.dmin[k] + c.dcode[k][x[r]];
        }
        break;
      case DFGPU_UINT32:
#pragma unroll
        for (int r = 0; r < R; r++) x[r] = (live >> r) & 1u ? (uint64_t)((const uint32_t*)p)[i[r]] : (uint64_t)c.dmin[k];
        break;
      case DFGPU_INT64:
#pragma unroll
        for (int r = 0; r < R; r++) x[r] = (live >> r) & 1u ? ((const uint64_t*)p)[i[r]] : (uint64_t)c.dmin[k];
        break;
      default:   // Int32, Date32
#pragma unroll
        for (int r = 0; r < R; r++) x[r] = (live >> r) & 1u ? (uint64_t)(int64_t)((const int32_t*)p)[i[r]] : (uint64_t)c.dmin[k];
        break;
    }
#pragma unroll
    for (int r = 0; r < R; r++) v[r] += (uint32_t)(x[r] - (uint64_t)c.dmin[k]) * c.dstride[k];
  }
}
__device__ __forceinline__ uint32_t direct_slot(const InternCtx& c, int64_t i) {
  const int64_t ii[1] = {i};
  uint32_t v[1];
  direct_slots<1>(c, ii, 1u, v);
  return v[0];
}
// claim pass over a direct table: a workgroup takes a contiguous slice of the rows and keeps the smallest row of every slot it
// meets in LDS (a read per row, an atomic only when the row is smaller than what is there — rows arrive in ascending order, so
// after a slot's first rows hardly ever); what it met goes to the device-wide array at the end, one atomic per slot and workgroup.
// first_row: u32 [n_slots], 0xFFFFFFFF = no row.  Reads the key columns, writes the rows' slots: no random access beyond LDS.
// A thread takes FOUR CONSECUTIVE rows: a UInt8 column's four values are one 32-bit load, an Int32 column's one 16-byte load, the
// four slots one 16-byte store (VEC: every key column starts on a 16-byte boundary) — a wave-level load of one byte per lane costs
// what one of 16 bytes per lane does (the first version, a row per lane and load: 4.2 ms for 600 M rows x (u8, u8, date32); this one
// 1.14 ms).  The UInt8 columns' value -> code tables sit in LDS behind the first rows.
constexpr int DIRECT_BLOCK = 1024;
constexpr int DIRECT_ROWS = 4;
constexpr uint32_t DIRECT_MAX_SLOTS = 32768;   // x 4 bytes of LDS
template <bool VEC>
__device__ __forceinline__ void direct_load4(const KeyCol& k, int64_t r0, int64_t hi, bool full, uint64_t fill, uint64_t (&x)[4]) {
  const void* p = k.data;
  if (VEC && full) {
    switch (k.type) {
      case DFGPU_UINT8: {
        const uint32_t w = *reinterpret_cast<const uint32_t*>((const uint8_t*)p + r0);
#pragma unroll
        for (int j = 0; j < 4; j++) x[j] = (w >> (8 * j)) & 255u;
        return;
      }
      case DFGPU_UINT32: {
        const uint4 w = *reinterpret_cast<const uint4*>((const uint32_t*)p + r0);
        x[0] = w.x; x[1] = w.y; x[2] = w.z; x[3] = w.w;
        return;
      }
      case DFGPU_INT64: {
        const ulonglong2 a = *reinterpret_cast<const ulonglong2*>((const uint64_t*)p + r0), b = *reinterpret_cast<const ulonglong2*>((const uint64_t*)p + r0 + 2);
        x[0] = a.x; x[1] = a.y; x[2] = b.x; x[3] = b.y;
        return;
      }
      default: {
        const int4 w = *reinterpret_cast<const int4*>((const int32_t*)p + r0);
        x[0] = (uint64_t)(int64_t)w.x; x[1] = (uint64_t)(int64_t)w.y; x[2] = (uint64_t)(int64_t)w.z; x[3] = (uint64_t)(int64_t)w.w;
        return;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; j++) {
    x[j] = fill;
    if (r0 + j >= hi) continue;
    switch (k.type) {
      case DFGPU_UINT8: x[j] = ((const uint8_t*)p)[r0 + j]; break;
      case DFGPU_UINT32: x[j] = ((const uint32_t*)p)[r0 + j]; break;
      case DFGPU_INT64: x[j] = ((const uint64_t*)p)[r0 + j]; break;
      default: x[j] = (uint64_t)(int64_t)((const int32_t*)p)[r0 + j]; break;
    }
  }
}
template <bool VEC>
__global__ __launch_bounds__(DIRECT_BLOCK) void k_intern_claim_direct(InternCtx c, int64_t n, uint32_t n_slots, const uint64_t* __restrict__ row_mask,
                                                                     uint32_t* __restrict__ row_slot, uint32_t* __restrict__ first_row) {
  extern __shared__ uint32_t s_first[];
  uint8_t* s_code = reinterpret_cast<uint8_t*>(s_first + n_slots);   // [keys.n][256]
  for (uint32_t x = threadIdx.x; x < n_slots; x += DIRECT_BLOCK) s_first[x] = 0xFFFFFFFFu;
  for (int k = 0; k < c.keys.n; k++)
    if (c.dcode[k] && threadIdx.x < 256) s_code[k * 256 + threadIdx.x] = c.dcode[k][threadIdx.x];
  __syncthreads();
  constexpr int U = DIRECT_ROWS;
  const int64_t per = ((n + gridDim.x - 1) / gridDim.x + 63) & ~(int64_t)63;
  const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
  for (int64_t r0 = lo + (int64_t)threadIdx.x * U; r0 < hi; r0 += (int64_t)U * DIRECT_BLOCK) {
    const bool full = r0 + U <= hi;
    uint32_t slot[U] = {0, 0, 0, 0};
    for (int k = 0; k < c.keys.n; k++) {
      uint64_t x[U];
      direct_load4<VEC>(c.keys.c[k], r0, hi, full, (uint64_t)c.dmin[k], x);
      if (c.dcode[k]) {
#pragma unroll
        for (int j = 0; j < U; j++) x[j] = s_code[k * 256 + (int)(x[j] & 255u)];   // (dmin is 0 for a coded column)
      }
#pragma unroll
      for (int j = 0; j < U; j++) slot[j] += (uint32_t)(x[j] - (uint64_t)c.dmin[k]) * c.dstride[k];
    }
    uint32_t live = 0xFu;
    if (row_mask) live = (uint32_t)(row_mask[r0 >> 6] >> (r0 & 63)) & 0xFu;   // (r0 is a multiple of 4: the four bits lie in one word)
#pragma unroll
    for (int j = 0; j < U; j++) {
      if (r0 + j >= hi) live &= ~(1u << j);
      if (!((live >> j) & 1u)) slot[j] = 0xFFFFFFFFu;
      else if ((uint32_t)(r0 + j) < s_first[slot[j]]) atomicMin(&s_first[slot[j]], (uint32_t)(r0 + j));
    }
    if (row_slot) {
      if (full) {
        *reinterpret_cast<uint4*>(row_slot + r0) = make_uint4(slot[0], slot[1], slot[2], slot[3]);
      } else {
#pragma unroll
        for (int j = 0; j < U; j++)
          if (r0 + j < hi) row_slot[r0 + j] = slot[j];
      }
    }
  }
  __syncthreads();
  for (uint32_t x = threadIdx.x; x < n_slots; x += DIRECT_BLOCK) {
    const uint32_t v = s_first[x];
    if (v != 0xFFFFFFFFu) atomicMin(&first_row[x], v);
  }
}
// first rows (0xFFFFFFFF = none) -> the slot array's convention (row + 1, 0 = empty)
__global__ __launch_bounds__(BLOCK) void k_direct_slots_from_rows(const uint32_t* __restrict__ first_row, uint64_t n_slots, uint32_t* __restrict__ slots) {
  for (uint64_t x = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; x < n_slots; x += (uint64_t)gridDim.x * BLOCK) slots[x] = first_row[x] + 1u;
}
// which of the 256 values a UInt8 column takes (a key column like l_returnflag holds 'A', 'N', 'R': 3 codes instead of a span of 18)
__global__ __launch_bounds__(BLOCK) void k_u8_presence(const uint8_t* __restrict__ col, int64_t n, unsigned long long* __restrict__ out) {
  __shared__ unsigned s_bits[8];
  if (threadIdx.x < 8) s_bits[threadIdx.x] = 0;
  __syncthreads();
  unsigned mine[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  auto mark = [&](unsigned v) {   // (constant indices: the eight words stay in registers)
#pragma unroll
    for (int q = 0; q < 8; q++) mine[q] |= (v >> 5) == (unsigned)q ? 1u << (v & 31) : 0u;
  };
  // 16 bytes per lane and load; the column's start is 16-byte aligned or the head is done bytewise.  (0.43 ms per 600 M rows either way,
  // with 4-byte loads as well: the pass is bound by its ~26 VALU operations per byte — eight compare / select / or per value —, not by
  // its loads.  It runs once per column and table: the set is kept with the column's statistics.)
  const int64_t head = std::min<int64_t>(n, (16 - ((uintptr_t)col & 15)) & 15);
  const int64_t n16 = (n - head) / 16;
  const uint4* v16 = reinterpret_cast<const uint4*>(col + head);
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n16; i += (int64_t)gridDim.x * BLOCK) {
    const uint4 w = v16[i];
    const unsigned ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int k = 0; k < 4; k++) {
#pragma unroll
      for (int b = 0; b < 4; b++) mark((ws[k] >> (8 * b)) & 255u);
    }
  }
  if (blockIdx.x == 0) {   // the unaligned head and the tail
    for (int64_t i = threadIdx.x; i < head; i += BLOCK) mark(col[i]);
    for (int64_t i = head + n16 * 16 + threadIdx.x; i < n; i += BLOCK) mark(col[i]);
  }
#pragma unroll
  for (int q = 0; q < 8; q++)
    if (mine[q]) atomicOr(&s_bits[q], mine[q]);
  __syncthreads();
  if (threadIdx.x < 4) {
    const unsigned long long w = (unsigned long long)s_bits[2 * threadIdx.x] | ((unsigned long long)s_bits[2 * threadIdx.x + 1] << 32);
    if (w) atomicOr(&out[threadIdx.x], w);
  }
}
// min / max of 4096 evenly spaced rows: tells a key column whose values span too much for a direct table without a pass over it
template <typename T>
__global__ void k_sample_minmax(const T* __restrict__ col, int64_t n, long long* __restrict__ out) {
  const int64_t step = n / 4096 > 0 ? n / 4096 : 1;
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * step;
  if (i >= n) return;
  const long long v = (long long)col[i];
  atomicMin(&out[0], v);
  atomicMax(&out[1], v);
}

__device__ __forceinline__ uint32_t lookup_gid(const InternCtx& c, const uint32_t* __restrict__ slot_gid, int64_t i) {
  if (c.direct) return slot_gid[direct_slot(c, i)];
  if (c.keyed) {
    const uint64_t k = packed_key(c.keys, i);
    if (k == KEY_EMPTY) return slot_gid[c.mask + 1];
    uint64_t s = packed_key_slot(k, c.mask);
    while (c.keyed[s].key != k) s = (s + 1) & c.mask;
    return slot_gid[s];
  }
  uint64_t s = group_hash(c.keys, i) & c.mask;
  for (;;) {
    uint32_t cur = c.slots[s];
    if (cur == (uint32_t)i + 1u || keys_equal(c.keys, (int64_t)cur - 1, c.keys, i, true)) return slot_gid[s];
    s = (s + 1) & c.mask;
  }
}

// ------------------------------------------------------------------------------ update
// HBM accumulators, one global atomic per (row, aggregate)
__global__ __launch_bounds__(BLOCK) void k_accumulate_global(InternCtx c, const uint32_t* __restrict__ slot_gid, int has_groups, int64_t row_offset,
                                                            int64_t n, AccSet accs) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    uint32_t gid = has_groups ? lookup_gid(c, slot_gid, row_offset + i) : 0u;
    for (int k = 0; k < accs.n; k++) {
      const AccDesc& d = accs.a[k];
      if (d.kind != ACC_COUNT_STAR && d.valid && !bit_at(d.valid, i)) continue;
      uint64_t lo = 0, hi = 0;
      if (d.kind != ACC_COUNT_STAR && d.kind != ACC_COUNT) load_value(d, i, lo, hi);
      accumulate_cell(d.kind, d.acc_lo + gid, d.acc_hi ? d.acc_hi + gid : nullptr, lo, hi);
      if (d.seen) d.seen[gid] = 1u;
    }
  }
}

// LDS-privatised accumulators: cell(rep, agg, gid) in shared memory, flushed once per workgroup
constexpr int LDS_CELLS = 2048;  // 2 x 8 B x 2048 = 32 KiB of accumulators + 8 KiB of flags per workgroup
__global__ __launch_bounds__(BLOCK) void k_accumulate_lds(InternCtx c, const uint32_t* __restrict__ slot_gid, int has_groups, int64_t row_offset,
                                                         int64_t n, AccSet accs, int ngroups, int nrep) {
  __shared__ unsigned long long s_lo[LDS_CELLS];
  __shared__ unsigned long long s_hi[LDS_CELLS];
  __shared__ uint32_t s_seen[LDS_CELLS];
  const int per_rep = accs.n * ngroups;
  const int cells = per_rep * nrep;
  for (int x = threadIdx.x; x < cells; x += BLOCK) {
    int k = (x % per_rep) / ngroups;
    s_lo[x] = acc_identity(accs.a[k].kind);
    s_hi[x] = 0ull;
    s_seen[x] = 0u;
  }
  __syncthreads();
  const int rep = (int)(threadIdx.x % (unsigned)nrep);
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    uint32_t gid = has_groups ? lookup_gid(c, slot_gid, row_offset + i) : 0u;
    for (int k = 0; k < accs.n; k++) {
      const AccDesc& d = accs.a[k];
      if (d.kind != ACC_COUNT_STAR && d.valid && !bit_at(d.valid, i)) continue;
      uint64_t lo = 0, hi = 0;
      if (d.kind != ACC_COUNT_STAR && d.kind != ACC_COUNT) load_value(d, i, lo, hi);
      int cell = rep * per_rep + k * ngroups + (int)gid;
      accumulate_cell(d.kind, &s_lo[cell], &s_hi[cell], lo, hi);
      s_seen[cell] = 1u;
    }
  }
  __syncthreads();
  for (int x = threadIdx.x; x < cells; x += BLOCK) {
    if (!s_seen[x]) continue;
    int within = x % per_rep;
    int k = within / ngroups, gid = within % ngroups;
    const AccDesc& d = accs.a[k];
    int kind = d.kind;
    if (kind == ACC_COUNT || kind == ACC_COUNT_STAR) kind = ACC_SUM_I64;  // merge counts by adding
    accumulate_cell(kind, d.acc_lo + gid, d.acc_hi ? d.acc_hi + gid : nullptr, s_lo[x], s_hi[x]);
    if (d.seen) d.seen[gid] = 1u;
  }
}

__global__ __launch_bounds__(BLOCK) void k_fill_u64(unsigned long long v, int64_t n, unsigned long long* out) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) out[i] = v;
}

// -------------------------------------------------------------------------------- emit
// mode: 0 copy i64 (lo), 1 i128 (lo,hi), 2 ordered-i64 -> f64, 3 i64 -> i32 narrowing, 4 i64 -> u8
__global__ __launch_bounds__(BLOCK) void k_emit_values(int mode, const unsigned long long* lo, const unsigned long long* hi, const uint32_t* seen,
                                                       int64_t n, void* out, uint8_t* valid_bytes) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    bool ok = !seen || seen[i] != 0;
    switch (mode) {
      case 0: ((unsigned long long*)out)[i] = ok ? lo[i] : 0ull; break;
      case 1: ((unsigned long long*)out)[2 * i] = ok ? lo[i] : 0ull; ((unsigned long long*)out)[2 * i + 1] = ok ? hi[i] : 0ull; break;
      case 2: ((double*)out)[i] = ok ? f64_from_ordered((int64_t)lo[i]) : 0.0; break;
      case 3: ((int32_t*)out)[i] = ok ? (int32_t)(int64_t)lo[i] : 0; break;
      case 4: ((uint8_t*)out)[i] = ok ? (uint8_t)lo[i] : 0; break;
    }
    if (valid_bytes) valid_bytes[i] = ok ? 1 : 0;
  }
}
__global__ __launch_bounds__(BLOCK) void k_seen_bytes(const uint32_t* __restrict__ seen, int64_t n, uint8_t* __restrict__ valid_bytes) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) valid_bytes[i] = seen[i] != 0 ? 1 : 0;
}
// AVG finalisation.  decimal: (sum * mul) / count, truncating; f64: sum / count
__global__ __launch_bounds__(BLOCK) void k_emit_avg(int is_decimal, const unsigned long long* lo, const unsigned long long* hi, const unsigned long long* cnt,
                                                    i128 mul, int64_t n, void* out, uint8_t* valid_bytes, int* overflow) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    unsigned long long c = cnt[i];
    bool ok = c != 0;
    if (is_decimal) {
      i128 r = 0;
      if (ok) {
        i128 s = (i128)(((u128)hi[i] << 64) | (u128)lo[i]);
        if (__builtin_mul_overflow(s, mul, &r)) *overflow = 1;  // sum.mul_checked
        r = r / (i128)c;
      }
      ((i128*)out)[i] = r;
    } else {
      ((double*)out)[i] = ok ? __longlong_as_double((long long)lo[i]) / (double)c : 0.0;
    }
    valid_bytes[i] = ok ? 1 : 0;
  }
}

// Every output column of an aggregate in ONE launch (blockIdx.y = column): the value (k_emit_values / k_emit_avg), its validity
// WORD (a wave's ballot: no byte-per-row detour) and the column's count of valid rows — read back once for all columns, together
// with the AVG overflow flags.  stats[2 e] = valid rows of entry e, stats[2 e + 1] = overflow seen.
struct EmitEntry {
  int kind;                         // 0 = accumulator value (mode as k_emit_values), 1 = AVG (mode = is_decimal)
  int mode;
  const unsigned long long *lo, *hi, *cnt;
  const uint32_t* seen;             // null: every row is valid
  unsigned long long mul_lo, mul_hi;   // AVG over decimals: 10^(return scale - sum scale)
  void* dst;
  uint64_t* valid_words;            // null: the column is not nullable
};
constexpr int EMIT_MAX = 2 * MAX_AGGS;
struct EmitSet {
  int n;
  EmitEntry e[EMIT_MAX];
};
__global__ __launch_bounds__(BLOCK) void k_emit_set(EmitSet s, int64_t n, unsigned long long* __restrict__ stats) {
  const EmitEntry& e = s.e[blockIdx.y];
  // a wave takes U consecutive 64-group words per round: everything it reads of them is requested before the first value is made (round 6:
  // one group per thread and round was a chain of load -> store round trips, 0.23 ms for 2 x 10 M groups whose bytes are worth 0.06)
  constexpr int U = 4;
  const unsigned lane = lane_id();
  const int64_t n_words = (n + WAVE - 1) / WAVE;
  const int64_t wv = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6, n_waves = ((int64_t)gridDim.x * BLOCK) >> 6;
  const bool avg = e.kind != 0;
  const bool need_lo = avg || e.mode != 5, need_hi = avg ? e.mode != 0 : e.mode == 1;
  unsigned long long valid_rows = 0;
  bool overflow = false;
  for (int64_t w0 = wv * U; w0 < n_words; w0 += n_waves * U) {
    bool in[U];
    uint32_t sn[U];
    unsigned long long lo[U], hi[U], cn[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t i = ((w0 + u) << 6) + lane;
      in[u] = i < n;
      sn[u] = in[u] && !avg && e.seen ? e.seen[i] : 1u;
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t i = ((w0 + u) << 6) + lane;
      lo[u] = in[u] && need_lo ? e.lo[i] : 0ull;
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t i = ((w0 + u) << 6) + lane;
      hi[u] = in[u] && need_hi ? e.hi[i] : 0ull;
      cn[u] = in[u] && avg ? e.cnt[i] : 0ull;
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t i = ((w0 + u) << 6) + lane;
      bool ok = false;
      if (in[u]) {
        if (!avg) {
          ok = sn[u] != 0;
          switch (e.mode) {
            case 0: ((unsigned long long*)e.dst)[i] = ok ? lo[u] : 0ull; break;
            case 1: reinterpret_cast<ulonglong2*>(e.dst)[i] = ok ? make_ulonglong2(lo[u], hi[u]) : make_ulonglong2(0ull, 0ull); break;
            case 2: ((double*)e.dst)[i] = ok ? f64_from_ordered((int64_t)lo[u]) : 0.0; break;
            case 3: ((int32_t*)e.dst)[i] = ok ? (int32_t)(int64_t)lo[u] : 0; break;
            case 4: ((uint8_t*)e.dst)[i] = ok ? (uint8_t)lo[u] : 0; break;
            default: break;   // 5: the values are in place already (the runs node's interleaved cells), only the validity is made
          }
        } else {
          const unsigned long long c = cn[u];
          ok = c != 0;
          if (e.mode) {
            i128 r = 0;
            if (ok) {
              const i128 sum = (i128)(((u128)hi[u] << 64) | (u128)lo[u]);
              const i128 mul = (i128)(((u128)e.mul_hi << 64) | (u128)e.mul_lo);
              if (__builtin_mul_overflow(sum, mul, &r)) overflow = true;  // sum.mul_checked
              r = r / (i128)c;
            }
            ((i128*)e.dst)[i] = r;
          } else {
            ((double*)e.dst)[i] = ok ? __longlong_as_double((long long)lo[u]) / (double)c : 0.0;
          }
        }
      }
      const uint64_t word = ballot64(ok);
      if (e.valid_words && lane == 0 && w0 + u < n_words) e.valid_words[w0 + u] = word;
      if (lane == 0) valid_rows += (unsigned long long)__popcll(word);
    }
  }
  // the column's count of valid rows: one atomic per workgroup, none for a column that is not nullable (the host does not look at it)
  __shared__ unsigned long long s_valid[BLOCK / WAVE];
  if (lane == 0) s_valid[threadIdx.x >> 6] = valid_rows;
  __syncthreads();
  if (threadIdx.x == 0 && e.valid_words) {
    unsigned long long t = 0;
#pragma unroll
    for (int w = 0; w < BLOCK / WAVE; w++) t += s_valid[w];
    if (t) atomicAdd(&stats[2 * blockIdx.y], t);
  }
  if (overflow) stats[2 * blockIdx.y + 1] = 1ull;
}

// ------------------------------------------------------------------------------- host
struct AggState {
  int func;               // dfgpu_agg_func
  bool has_arg;
  std::vector<dfgpu_expr_node> nodes;
  int root = 0;
  std::string name;
  dfgpu_field in_type{};  // argument type (raw modes) / state value type (final modes)
  dfgpu_field ret{};      // planner-declared return type (type 0 = derive from in_type)
  bool typed = false;
  // accumulator storage
  BufPtr lo, hi, seen;    // primary accumulator (sum / min / max / count)
  BufPtr cnt;             // AVG: row count
  // SUM over Decimal128 written by the ordered-input runs node as the first update: {lo, hi} side by side, 16 bytes per group —
  // the Decimal128 column itself, which emit hands out as it is (the lo / hi arrays would cost a 2 x 8 -> 16 byte pass over
  // every group: 1.0 of the 7.2 ms of GROUP BY l_orderkey at SF100).  lo / hi are stale while this is set; the next update
  // splits it back (split_interleaved).
  BufPtr inter;
  // set by the ordered-input runs node for an accumulator whose argument cannot be NULL: every group has a value, `seen` was not written
  // (4 bytes per group the kernel does not store and emit does not read back — 150 M groups: 1.2 GB).  The next update fills it in
  // (split_interleaved); emit hands the column out without a validity buffer.
  bool seen_all = false;
};

struct Aggregate {
  int mode;
  std::vector<std::vector<dfgpu_expr_node>> group_nodes;
  std::vector<int> group_roots;
  std::vector<std::string> group_names;
  std::vector<AggState> aggs;
  Table group_keys;  // dense, gid order
  int64_t ngroups = 0;
  int64_t capacity_hint = 1 << 16;
  std::vector<uint16_t> small_keys;  // small-domain interning: host mirror of the group keys (byte g of key i = column g)
  int64_t fused_updates = 0;         // updates that took the fused (rowprog) path
  std::vector<uint8_t> utf8_key;     // group key g arrived as a Utf8 column: interned on entry, decoded again on emit
  std::vector<uint8_t> bool_key;     // group key g arrived as a Boolean column: one byte per row on entry, bit-packed again on emit
  bool touched = false;              // an update ran: the state is bound to its device
  // grouping sets (PhysicalGroupBy::groups, aggregates/mod.rs:400-520): one aggregate per set — the set's NULLed-out key
  // expressions replaced by the typed NULL literals, `__grouping_id` appended as a literal key — updated together, emitted
  // one after the other.  Empty for a plain GROUP BY.
  std::vector<std::unique_ptr<Aggregate>> sets;
  // PartialReduce (aggregates/mod.rs:340-361): partial states in, partial states out — the merge of Final with the output of Partial
  bool final_mode() const { return mode == DFGPU_AGG_FINAL || mode == DFGPU_AGG_FINAL_PARTITIONED || mode == DFGPU_AGG_PARTIAL_REDUCE; }
  bool partial_out() const { return mode == DFGPU_AGG_PARTIAL || mode == DFGPU_AGG_PARTIAL_REDUCE; }
};

// the calling thread moves to the device the aggregate's state lives on
static Aggregate* unwrap_agg(dfgpu_agg_t h) {
  DFGPU_CHECK(h != nullptr, "null aggregate handle");
  Aggregate* a = reinterpret_cast<Aggregate*>(h);
  if (a->group_keys.device >= 0) use_device(a->group_keys.device);
  return a;
}

// an aggregate binds to the device of its first input (it is created before any input exists)
static const Table& agg_input(Aggregate& a, dfgpu_table_t input) {
  const Table& in = *unwrap(input);
  if (!a.touched) {
    a.group_keys.device = in.device;
    a.touched = true;
  }
  DFGPU_CHECK(in.device == a.group_keys.device, "the input table lives on another device than the aggregate's state");
  return in;
}

static dfgpu_field fld(int type, int p = 0, int s = 0) {
  dfgpu_field f{};
  f.type = type;
  f.precision = p;
  f.scale = s;
  f.nullable = 1;
  return f;
}
// SUM result type (functions-aggregate/src/sum.rs:232-260)
static dfgpu_field sum_type(const dfgpu_field& t) {
  switch (t.type) {
    case DFGPU_DECIMAL128: return fld(DFGPU_DECIMAL128, std::min(38, t.precision + 10), t.scale);
    case DFGPU_INT32: case DFGPU_INT64: case DFGPU_UINT8: return fld(DFGPU_INT64);
    case DFGPU_UINT32: case DFGPU_UINT64: return fld(DFGPU_UINT64);
    case DFGPU_FLOAT64: return fld(DFGPU_FLOAT64);
  }
  throw Error("SUM over " + type_name(t) + " is not supported on the GPU path");
}
// AVG result type (functions-aggregate/src/average.rs:219-252)
static dfgpu_field avg_type(const dfgpu_field& t) {
  if (t.type == DFGPU_DECIMAL128) return fld(DFGPU_DECIMAL128, std::min(38, t.precision + 4), std::min(38, t.scale + 4));
  return fld(DFGPU_FLOAT64);
}
// the type the AVG sum state carries: avg_sum_data_type (functions-aggregate/src/average.rs:131-172) — the input precision
// plus 13 digits of headroom, never narrower than Decimal128's maximum; beyond 38 digits the reference accumulates in
// Decimal256, which has no device representation (the sums here are i128)
static dfgpu_field avg_sum_type(const dfgpu_field& t) {
  if (t.type == DFGPU_DECIMAL128) {
    DFGPU_CHECK(t.precision + 13 <= 38, "AVG over " + type_name(t) + " accumulates in Decimal256 in the reference: not supported on the GPU path");
    return fld(DFGPU_DECIMAL128, 38, t.scale);
  }
  return fld(DFGPU_FLOAT64);
}

static BufPtr filled(int64_t n, unsigned long long v) {
  BufPtr b = make_buf((size_t)(n ? n : 1) * 8);
  if (n) k_fill_u64<<<grid_for(n, BLOCK), BLOCK, 0, rt().stream>>>(v, n, b->as<unsigned long long>());
  return b;
}
static BufPtr grown(const BufPtr& old, int64_t old_n, int64_t new_n, unsigned long long fill, int elem = 8, bool init = true) {
  if (old && old_n == new_n && old_n > 0 && old->bytes >= (size_t)new_n * elem) return old;   // no new group: nothing to allocate, fill or copy
  BufPtr b;
  if (!init) b = make_buf((size_t)(new_n ? new_n : 1) * elem);  // the caller overwrites every element
  else if (elem == 8) b = filled(new_n, fill);
  else b = make_zero_buf((size_t)(new_n ? new_n : 1) * elem);
  if (old && old_n) DFGPU_HIP(hipMemcpyAsync(b->ptr, old->ptr, (size_t)old_n * elem, hipMemcpyDeviceToDevice, rt().stream));
  return b;
}

// what one aggregate accumulates, given its input column type
struct AccPlan {
  int kind, val;
  bool needs_hi;
};
static AccPlan plan_for(int func, const dfgpu_field& t, bool merging_counts) {
  switch (func) {
    case DFGPU_AGG_COUNT:
      // Final modes merge partial counts by summing them (count.rs merge_batch)
      if (merging_counts) return {ACC_SUM_I64, t.type == DFGPU_UINT64 ? VAL_U64 : VAL_I64, false};
      return {ACC_COUNT, VAL_I64, false};
    case DFGPU_AGG_SUM:
    case DFGPU_AGG_AVG:
      switch (t.type) {
        case DFGPU_DECIMAL128: return {ACC_SUM_I128, VAL_I128, true};
        case DFGPU_INT32: return func == DFGPU_AGG_AVG ? AccPlan{ACC_SUM_F64, VAL_I32_TO_F64, false} : AccPlan{ACC_SUM_I64, VAL_I32, false};
        case DFGPU_INT64: return func == DFGPU_AGG_AVG ? AccPlan{ACC_SUM_F64, VAL_I64_TO_F64, false} : AccPlan{ACC_SUM_I64, VAL_I64, false};
        case DFGPU_UINT8: return {ACC_SUM_I64, VAL_U8, false};
        case DFGPU_UINT32: return {ACC_SUM_I64, VAL_U32, false};
        case DFGPU_UINT64: return {ACC_SUM_I64, VAL_U64, false};
        case DFGPU_FLOAT64: return {ACC_SUM_F64, VAL_F64, false};
      }
      break;
    case DFGPU_AGG_MIN:
    case DFGPU_AGG_MAX: {
      int k = func == DFGPU_AGG_MIN ? ACC_MIN_I64 : ACC_MAX_I64;
      switch (t.type) {
        case DFGPU_INT32: case DFGPU_DATE32: return {k, VAL_I32, false};
        case DFGPU_INT64: return {k, VAL_I64, false};
        case DFGPU_UINT8: return {k, VAL_U8, false};
        case DFGPU_UINT32: return {k, VAL_U32, false};
        case DFGPU_FLOAT64: return {k, VAL_F64_ORDERED, false};
        case DFGPU_DECIMAL128:
          // 64-bit atomics: exact whenever the values fit in 64 bits — by type for precision <= 18 (every TPC-H money column is
          // Decimal128(15,2)); for wider types (MAX over a SUM's Decimal128(38,4), TPC-H Q15) the update checks the values
          // themselves on the device first (wide_minmax_values_fit) and is an error otherwise
          return {k, VAL_I128, false};  // low word sign-carrying: values fit in i64
      }
      break;
    }
  }
  throw Error("aggregate over " + type_name(t) + " is not supported on the GPU path");
}

// MIN / MAX over Decimal128 wider than 18 digits: is every (valid) value representable in 64 bits?
static bool wide_minmax(int func, const dfgpu_field& t) {
  return (func == DFGPU_AGG_MIN || func == DFGPU_AGG_MAX) && t.type == DFGPU_DECIMAL128 && t.precision > 18;
}
__global__ __launch_bounds__(BLOCK) void k_i128_fits_i64(const uint64_t* __restrict__ words, const uint64_t* __restrict__ valid, int64_t n, uint32_t* __restrict__ flag) {
  bool bad = false;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    if (valid && !bit_at(valid, i)) continue;
    const int64_t lo = (int64_t)words[2 * i], hi = (int64_t)words[2 * i + 1];
    bad |= hi != (lo >> 63);
  }
  if (__any(bad) && lane_id() == 0) atomicOr(flag, 1u);
}
static void wide_minmax_values_fit(const Column& v, const std::string& name) {
  if (v.length == 0) return;
  BufPtr flag = make_zero_buf(4);
  k_i128_fits_i64<<<grid_for(v.length, BLOCK), BLOCK, 0, rt().stream>>>((const uint64_t*)v.ptr(), v.valid_words(), v.length, flag->as<uint32_t>());
  DFGPU_HIP(hipGetLastError());
  uint32_t bad = 0;
  d2h(&bad, flag->ptr, 4);
  DFGPU_CHECK(!bad, "MIN/MAX(" + name + ") over Decimal128 with precision > 18: a value does not fit in 64 bits (not supported on the GPU path)");
}

// ------------------------------------------------------------------ fused update (rowprog)
// FilterExec predicate + ProjectionExec expressions + aggregate arguments evaluated per row in
// registers (rowprog.hpp), then accumulated — one pass over the referenced input columns.
struct FusedAcc {
  unsigned long long* acc_lo;
  unsigned long long* acc_hi;
  uint32_t* seen;
  int16_t kind;  // AccKind
  int16_t reg;   // value register, -1 = none (COUNT(*))
};
struct FusedAccSet {
  FusedAcc a[MAX_AGGS];
  int n;
};
enum GidMode : int { GID_NONE = 0, GID_SMALL = 1, GID_HASH = 2 };
constexpr int SMALL_DOMAIN = 1 << 16;  // up to two 1-byte group keys address a gid table directly
struct GidSpec {
  int mode;
  int key_reg0, key_reg1;     // GID_SMALL: registers holding the key bytes (key_reg1 = -1: one key)
  const uint32_t* gid_table;  // GID_SMALL: [SMALL_DOMAIN] key bytes -> dense group id
  InternCtx ictx;             // GID_HASH
  const uint32_t* slot_gid;
  int64_t row_offset;         // GID_HASH: index of input row 0 in the concatenated key columns
};

__device__ __forceinline__ uint32_t small_key(const RpRegs& r, int k0, int k1) {
  uint32_t k = r.w0[k0] & 0xFFu;
  if (k1 >= 0) k |= (r.w0[k1] & 0xFFu) << 8;
  return k;
}

// GroupValues::intern for 1-byte keys: first[key] = smallest passing row holding that key.  A wave
// elects one lane per distinct key (its lowest lane = its smallest row), and a cached read of the
// current minimum filters almost every atomic once the first rows have been seen.
template <int NREG>
__global__ __launch_bounds__(BLOCK) void k_small_first_rows(RowProgram p, int n_prologue, int n_pred_end, int pred_reg, int key_reg0, int key_reg1,
                                                           int64_t n, uint32_t* __restrict__ first) {
  RP_DECLARE_REGS(r, NREG);
  rp_exec(p, 0, n_prologue, r);
  const int64_t n_round = (n + BLOCK - 1) / BLOCK * BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n_round; i += (int64_t)gridDim.x * BLOCK) {
    bool active = i < n;
    if (active) {
      rp_load_row(p, i, r);
      rp_exec(p, n_prologue, n_pred_end, r);
      if (pred_reg >= 0) active = rp_true(r, pred_reg);
      if (active) rp_exec(p, n_pred_end, p.n_ins, r);
    }
    const uint32_t key = active ? small_key(r, key_reg0, key_reg1) : 0u;
    uint64_t todo = ballot64(active);
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const uint32_t k = (uint32_t)__shfl((int)key, leader, 64);
      const uint64_t same = ballot64(active && key == k);
      if ((int)lane_id() == leader) {
        const uint32_t row = (uint32_t)i;
        if (row < __hip_atomic_load(&first[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&first[k], row);
      }
      todo &= ~same;
    }
  }
}

template <bool USE_LDS, int NREG>
__global__ __launch_bounds__(BLOCK) void k_agg_fused(RowProgram p, int n_prologue, int n_pred_end, int pred_reg, GidSpec g, FusedAccSet accs, int64_t n,
                                                    int ngroups, int nrep) {
  __shared__ unsigned long long s_lo[USE_LDS ? LDS_CELLS : 1];
  __shared__ unsigned long long s_hi[USE_LDS ? LDS_CELLS : 1];
  __shared__ uint32_t s_seen[USE_LDS ? LDS_CELLS : 1];
  // LDS cell of (aggregate k, group gid, replica rep) = (k * ngroups + gid) * nrep + rep: the replicas of one
  // accumulator are adjacent, so the lanes of a wave (rep = lane % nrep) spread over the LDS banks
  const int cells = accs.n * ngroups * nrep;
  if (USE_LDS) {
    for (int x = threadIdx.x; x < cells; x += BLOCK) {
      int k = x / (ngroups * nrep);
      s_lo[x] = acc_identity(accs.a[k].kind);
      s_hi[x] = 0ull;
      s_seen[x] = 0u;
    }
    __syncthreads();
  }
  const int rep = (int)(threadIdx.x & (unsigned)(nrep - 1));
  RP_DECLARE_REGS(r, NREG);
  rp_exec(p, 0, n_prologue, r);
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    rp_load_row(p, i, r);
    if (pred_reg >= 0) {
      rp_exec(p, n_prologue, n_pred_end, r);
      if (!rp_true(r, pred_reg)) continue;
    }
    rp_exec(p, n_pred_end, p.n_ins, r);
    uint32_t gid = 0;
    if (g.mode == GID_SMALL) gid = g.gid_table[small_key(r, g.key_reg0, g.key_reg1)];
    else if (g.mode == GID_HASH) gid = lookup_gid(g.ictx, g.slot_gid, g.row_offset + i);
    for (int k = 0; k < accs.n; k++) {
      const FusedAcc& d = accs.a[k];
      uint64_t lo = 0, hi = 0;
      if (d.reg >= 0) {
        if (rp_is_null(r, d.reg)) continue;
        lo = r.lo(d.reg);
        hi = r.hi(d.reg);
      }
      if (USE_LDS) {
        const int cell = (k * ngroups + (int)gid) * nrep + rep;
        accumulate_cell(d.kind, &s_lo[cell], &s_hi[cell], lo, hi);
        s_seen[cell] = 1u;
      } else {
        accumulate_cell(d.kind, d.acc_lo + gid, d.acc_hi ? d.acc_hi + gid : nullptr, lo, hi);
        if (d.seen) d.seen[gid] = 1u;
      }
    }
  }
  if (USE_LDS) {
    __syncthreads();
    // fold the replicas of each (aggregate, group) cell, one thread per cell, then one global update
    const int per = accs.n * ngroups;
    for (int x = threadIdx.x; x < per; x += BLOCK) {
      const int k = x / ngroups, gid = x % ngroups;
      const FusedAcc& d = accs.a[k];
      int kind = d.kind;
      bool any = false;
      unsigned long long lo = acc_identity(kind), hi = 0ull;
      for (int q = 0; q < nrep; q++) {
        const int cell = x * nrep + q;
        if (!s_seen[cell]) continue;
        any = true;
        const unsigned long long vlo = s_lo[cell], vhi = s_hi[cell];
        switch (kind) {
          case ACC_SUM_I128: { unsigned long long o = lo; lo += vlo; hi += vhi + (lo < o ? 1ull : 0ull); break; }
          case ACC_SUM_F64: lo = (unsigned long long)__double_as_longlong(__longlong_as_double((long long)lo) + __longlong_as_double((long long)vlo)); break;
          case ACC_MIN_I64: lo = (unsigned long long)min((long long)lo, (long long)vlo); break;
          case ACC_MAX_I64: lo = (unsigned long long)max((long long)lo, (long long)vlo); break;
          default: lo += vlo; break;  // SUM_I64 / COUNT / COUNT(*)
        }
      }
      if (!any) continue;
      if (kind == ACC_COUNT || kind == ACC_COUNT_STAR) kind = ACC_SUM_I64;  // merge counts by adding
      accumulate_cell(kind, d.acc_lo + gid, d.acc_hi ? d.acc_hi + gid : nullptr, lo, hi);
      if (d.seen) d.seen[gid] = 1u;
    }
  }
}

// ------------------------------------------------------- single-pass small-domain fused node
// Low-cardinality groups keyed by up to two 1-byte columns (TPC-H Q1: l_returnflag, l_linestatus), or no
// GROUP BY at all: ONE pass over the input does FilterExec + ProjectionExec + intern + accumulate.
//   * no interning pre-pass: a workgroup maps key bytes -> local slot through a tiny LDS table (LDS CAS);
//     accumulators are flushed into HBM arrays indexed by the KEY itself (domain 1 / 256 / 65536) and the
//     first passing row of every key is tracked with min(), so the host can number the groups in first-seen
//     order afterwards (group_values/mod.rs:88-92) and a tiny merge kernel folds key-indexed partials into
//     the dense accumulators.
//   * SUM(Decimal128) without carries: the 128-bit value is split into limbs of 43 + 43 + 42 bits that are
//     added with fire-and-forget ds_add_u64 (no returned value, no dependent second atomic).  A limb cell
//     holds < 2^43 per add and a workgroup adds < 2^20 rows, so no limb sum can wrap 64 bits; the limbs are
//     recombined mod 2^128 at flush time — bit-identical to wrapping i128 addition in any order.
//   * row i+stride's column loads are issued before row i is interpreted (rp_issue_row / rp_commit_row).
constexpr int SM_MAX_CELLS = 48;    // LDS cells per group (an i128 sum takes 3)
constexpr int SM_MAX_L = 256;       // local slots per workgroup
constexpr uint64_t LIMB_MASK = (1ull << 43) - 1ull;
struct SmallAcc {
  unsigned long long* tmp_lo;  // [domain] key-indexed partials
  unsigned long long* tmp_hi;  // [domain], i128 sums only
  int16_t kind;                // AccKind
  int16_t reg;                 // value register, -1 = none (COUNT(*))
  int16_t cell0;               // first LDS cell of this accumulator
  int16_t pad;
};
struct SmallAccSet {
  SmallAcc a[MAX_AGGS];
  int n;
  int ncell;
  uint8_t cell_kind[SM_MAX_CELLS];  // AccKind whose identity initialises the cell
};

__device__ __forceinline__ int small_local_slot(uint32_t* s_key, int L, uint32_t key) {
  const uint32_t want = key + 1u;
  int slot = (int)((key * 0x9E3779B1u) >> 20) & (L - 1);
  for (int probe = 0; probe < L; probe++) {
    uint32_t cur = s_key[slot];
    if (cur == want) return slot;
    if (cur == 0u) {
      cur = atomicCAS(&s_key[slot], 0u, want);
      if (cur == 0u || cur == want) return slot;
    }
    slot = (slot + 1) & (L - 1);
  }
  return -1;
}

// LDS limb cells += the 128-bit value (lo, hi): three fire-and-forget ds_add_u64
__device__ __forceinline__ void lds_add_limbs(unsigned long long* c, int plane, uint64_t lo, uint64_t hi) {
  atomicAdd(c, (unsigned long long)(lo & LIMB_MASK));
  atomicAdd(c + plane, (unsigned long long)(((lo >> 43) | (hi << 21)) & LIMB_MASK));
  atomicAdd(c + 2 * plane, (unsigned long long)((int64_t)hi >> 22));
}

// rows [begin, end) of the input.  LDS: register file (TileProgram) | accumulator cells | seen | keys | first rows.
template <bool PREFETCH>
__global__ __launch_bounds__(BLOCK) void k_agg_fused_tile(TileProgram p, int pred_opnd, int key_opnd0, int key_opnd1, SmallAccSet accs, int64_t begin,
                                                         int64_t end, int L, int nrep, uint32_t* __restrict__ g_first, uint32_t* __restrict__ g_seen) {
  extern __shared__ __align__(16) unsigned char s_raw[];
  TileRegs t;
  t.wide = reinterpret_cast<char*>(s_raw) + threadIdx.x * 16;
  t.narrow = reinterpret_cast<char*>(s_raw) + (size_t)p.n_wide * (BLOCK * 16) + threadIdx.x * 8;
  t.nulls = 0u;
  unsigned long long* s_cell = reinterpret_cast<unsigned long long*>(s_raw + tile_regfile_bytes(p.n_wide, p.n_narrow));
  const int plane = L * nrep;  // cell(c, slot, rep) = (c * L + slot) * nrep + rep
  const int ncells = accs.ncell * plane;
  uint32_t* s_seen = (uint32_t*)(s_cell + ncells);  // [slot * nrep + rep] bit k = accumulator k saw a value
  uint32_t* s_key = s_seen + plane;                 // [L] key + 1, 0 = free
  uint32_t* s_first = s_key + L;                    // [L] smallest passing row of the slot's key
  for (int x = threadIdx.x; x < ncells; x += BLOCK) s_cell[x] = acc_identity(accs.cell_kind[x / plane]);
  for (int x = threadIdx.x; x < plane; x += BLOCK) s_seen[x] = 0u;
  for (int x = threadIdx.x; x < L; x += BLOCK) {
    s_key[x] = 0u;
    s_first[x] = 0xFFFFFFFFu;
  }
  __syncthreads();
  const int rep = (int)(threadIdx.x & (unsigned)(nrep - 1));
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  int64_t i = begin + (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  RpRaw raw;
  if (PREFETCH && i < end) tp_issue_row(p, i, raw);
  for (; i < end; i += stride) {
    if (!PREFETCH) tp_issue_row(p, i, raw);
    tp_commit_row(p, raw, t);
    if (PREFETCH && i + stride < end) tp_issue_row(p, i + stride, raw);
    if (pred_opnd >= 0) {
      tp_exec(p, 0, p.n_pred_end, t);
      if (!tp_true(p, t, (uint32_t)pred_opnd)) continue;
    }
    tp_exec(p, p.n_pred_end, p.n_ins, t);
    uint32_t key = 0u;
    if (key_opnd0 >= 0) {
      uint64_t klo, khi;
      bool kn;
      tp_fetch(p, t, (uint32_t)key_opnd0, klo, khi, kn);
      key = (uint32_t)klo & 0xFFu;
      if (key_opnd1 >= 0) {
        tp_fetch(p, t, (uint32_t)key_opnd1, klo, khi, kn);
        key |= ((uint32_t)klo & 0xFFu) << 8;
      }
    }
    const int slot = small_local_slot(s_key, L, key);
    if (slot >= 0) {
      if ((uint32_t)i < s_first[slot]) atomicMin(&s_first[slot], (uint32_t)i);
      const int base = slot * nrep + rep;
      uint32_t seen = 0u;
      for (int k = 0; k < accs.n; k++) {
        const SmallAcc& d = accs.a[k];
        uint64_t lo = 0, hi = 0;
        if (d.reg >= 0) {
          bool isnull;
          tp_fetch(p, t, (uint32_t)d.reg, lo, hi, isnull);
          if (isnull) continue;
        }
        seen |= 1u << k;
        unsigned long long* c = s_cell + d.cell0 * plane + base;
        switch (d.kind) {
          case ACC_SUM_I128: lds_add_limbs(c, plane, lo, hi); break;
          case ACC_SUM_I64: atomicAdd(c, (unsigned long long)lo); break;
          case ACC_SUM_F64: atomicAdd(reinterpret_cast<double*>(c), __longlong_as_double((long long)lo)); break;
          case ACC_MIN_I64: atomicMin(reinterpret_cast<long long*>(c), (long long)lo); break;
          case ACC_MAX_I64: atomicMax(reinterpret_cast<long long*>(c), (long long)lo); break;
          default: atomicAdd(c, 1ull); break;  // COUNT / COUNT(*)
        }
      }
      if (seen & ~s_seen[base]) atomicOr(&s_seen[base], seen);
    } else {
      // more distinct keys in this workgroup than local slots: straight to the key-indexed HBM partials
      if ((uint32_t)i < __hip_atomic_load(&g_first[key], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&g_first[key], (uint32_t)i);
      uint32_t seen = 0u;
      for (int k = 0; k < accs.n; k++) {
        const SmallAcc& d = accs.a[k];
        uint64_t lo = 0, hi = 0;
        if (d.reg >= 0) {
          bool isnull;
          tp_fetch(p, t, (uint32_t)d.reg, lo, hi, isnull);
          if (isnull) continue;
        }
        seen |= 1u << k;
        accumulate_cell(d.kind, d.tmp_lo + key, d.tmp_hi ? d.tmp_hi + key : nullptr, lo, hi);
      }
      if (seen) atomicOr(&g_seen[key], seen);
    }
  }
  __syncthreads();
  // fold the replicas of each (slot, accumulator), then one update of the key-indexed partials
  for (int x = threadIdx.x; x < L * accs.n; x += BLOCK) {
    const int slot = x / accs.n, k = x % accs.n;
    const uint32_t kv = s_key[slot];
    if (!kv) continue;
    const uint32_t key = kv - 1u;
    uint32_t seen = 0u;
    for (int q = 0; q < nrep; q++) seen |= s_seen[slot * nrep + q];
    if (!((seen >> k) & 1u)) continue;
    const SmallAcc& d = accs.a[k];
    const unsigned long long* c = s_cell + d.cell0 * plane + slot * nrep;
    int kind = d.kind;
    unsigned long long lo = acc_identity(kind), hi = 0ull;
    if (kind == ACC_SUM_I128) {
      unsigned long long s0 = 0, s1 = 0, s2 = 0;
      for (int q = 0; q < nrep; q++) {
        s0 += c[q];
        s1 += c[plane + q];
        s2 += c[2 * plane + q];
      }
      u128 v = (u128)s0 + ((u128)s1 << 43) + (u128)((i128)(int64_t)s2 << 86);
      lo = (unsigned long long)v;
      hi = (unsigned long long)(v >> 64);
    } else {
      for (int q = 0; q < nrep; q++) {
        const unsigned long long v = c[q];
        switch (kind) {
          case ACC_SUM_F64: lo = (unsigned long long)__double_as_longlong(__longlong_as_double((long long)lo) + __longlong_as_double((long long)v)); break;
          case ACC_MIN_I64: lo = (unsigned long long)min((long long)lo, (long long)v); break;
          case ACC_MAX_I64: lo = (unsigned long long)max((long long)lo, (long long)v); break;
          default: lo += v; break;  // SUM_I64 / COUNT / COUNT(*)
        }
      }
      if (kind == ACC_COUNT || kind == ACC_COUNT_STAR) kind = ACC_SUM_I64;  // merge counts by adding
    }
    accumulate_cell(kind, d.tmp_lo + key, d.tmp_hi ? d.tmp_hi + key : nullptr, lo, hi);
    atomicOr(&g_seen[key], 1u << k);
  }
  for (int slot = threadIdx.x; slot < L; slot += BLOCK)
    if (s_key[slot] && s_first[slot] != 0xFFFFFFFFu) atomicMin(&g_first[s_key[slot] - 1u], s_first[slot]);
}

// key-indexed partials of the touched keys -> dense accumulators (one thread per (key, destination)).
// Several destinations may share one source: SUM(x) and AVG(x) accumulate the same sum once.
struct MergeDst {
  unsigned long long* lo[MAX_AGGS];
  unsigned long long* hi[MAX_AGGS];
  uint32_t* seen[MAX_AGGS];
  int src[MAX_AGGS];  // index into SmallAccSet::a
  int n;
};
__device__ __forceinline__ void small_merge_one(int e, uint32_t key, uint32_t gid, const SmallAccSet& accs, const MergeDst& dst, const uint32_t* __restrict__ g_seen) {
  const int k = dst.src[e];
  if (!((g_seen[key] >> k) & 1u)) return;
  const SmallAcc& d = accs.a[k];
  int kind = d.kind;
  if (kind == ACC_COUNT || kind == ACC_COUNT_STAR) kind = ACC_SUM_I64;
  accumulate_cell(kind, dst.lo[e] + gid, dst.hi[e] ? dst.hi[e] + gid : nullptr, d.tmp_lo[key], d.tmp_hi ? d.tmp_hi[key] : 0ull);
  if (dst.seen[e]) dst.seen[e][gid] = 1u;
}
__global__ __launch_bounds__(BLOCK) void k_small_merge(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ gids, int n_keys, SmallAccSet accs,
                                                      MergeDst dst, const uint32_t* __restrict__ g_seen) {
  const int x = blockIdx.x * BLOCK + threadIdx.x;
  if (x >= n_keys * dst.n) return;
  const int t = x / dst.n, e = x % dst.n;
  small_merge_one(e, keys[t], gids[t], accs, dst, g_seen);
}
// the same with the touched keys and their group numbers in the kernel's ARGUMENTS (a handful of groups: no upload, no host
// buffer to keep alive, nothing to wait for)
constexpr int SMALL_KG_MAX = 64;
struct SmallKG {
  uint32_t key[SMALL_KG_MAX], gid[SMALL_KG_MAX];
};
__global__ __launch_bounds__(BLOCK) void k_small_merge_v(SmallKG kg, int n_keys, SmallAccSet accs, MergeDst dst, const uint32_t* __restrict__ g_seen) {
  const int x = blockIdx.x * BLOCK + threadIdx.x;
  if (x >= n_keys * dst.n) return;
  const int t = x / dst.n, e = x % dst.n;
  small_merge_one(e, kg.key[t], kg.gid[t], accs, dst, g_seen);
}
struct SmallBytes {
  uint8_t v[SMALL_KG_MAX];
};
__global__ void k_set_bytes_v(SmallBytes b, int n, uint8_t* __restrict__ dst) {
  if ((int)threadIdx.x < n) dst[threadIdx.x] = b.v[threadIdx.x];
}

__global__ __launch_bounds__(BLOCK) void k_and_words(const uint64_t* __restrict__ a, const uint64_t* __restrict__ b, int64_t nw, uint64_t* __restrict__ out) {
  for (int64_t w = (int64_t)blockIdx.x * BLOCK + threadIdx.x; w < nw; w += (int64_t)gridDim.x * BLOCK) out[w] = a[w] & b[w];
}
void and_bitmaps(const uint64_t* a, const uint64_t* b, int64_t nw, uint64_t* out) {
  if (nw) k_and_words<<<grid_for(nw, BLOCK), BLOCK, 0, rt().stream>>>(a, b, nw, out);
}

static bool g_fusion_enabled = true;
void set_fusion_enabled(bool on) { g_fusion_enabled = on; }
bool fusion_enabled() { return g_fusion_enabled; }

// ------------------------------------------------------------------------------ host
// GroupValues::intern over materialised key columns (hash table of representative rows)
struct InternResult {
  InternCtx ictx{};
  BufPtr slots, slot_gid;
  BufPtr row_slot;               // (on request) the slot of every concatenated row, 0xFFFFFFFF where the claim pass skipped it
  BufPtr keyed;                  // keyed table: packed key and row of every slot
  std::vector<BufPtr> direct_codes;   // direct table: the UInt8 key columns' value -> code tables
  std::vector<Column> cat_keys;  // [existing group keys ; input keys] — referenced by ictx
  int64_t G1 = 0;
};
// Do the key columns' value ranges multiply to a table of <= DIRECT_MAX_SLOTS slots?  Fills ictx.dmin / dstride and returns the number
// of slots (0: no).  Ranges come from the columns' cached statistics (column_stats: a pass over a column that has none yet — a
// 4096-row sample first says whether it can be worth it).
static ColStats u8_presence(Column& c, int64_t n) {
  ColStats st = column_stats(c, n);
  if (st.has_present) return st;
  Runtime& r = rt();
  BufPtr d = make_zero_buf(32);
  {
    ProfileScope ps("column_u8_presence", n);
    k_u8_presence<<<std::min(grid_for(n / 16 + 1, BLOCK * 4), 2048), BLOCK, 0, r.stream>>>((const uint8_t*)c.ptr(), n, d->as<unsigned long long>());
    DFGPU_HIP(hipGetLastError());
  }
  d2h(st.present, d->ptr, 32);
  st.has_present = true;
  std::atomic_store(&c.stats, std::make_shared<ColStats>(st));
  return st;
}
static uint32_t direct_table_spec(std::vector<Column>& keys, const std::vector<Column*>* homes, int64_t n, InternCtx& ictx, std::vector<BufPtr>& keep) {
  Runtime& r = rt();
  const int ngk = (int)keys.size();
  std::vector<int> order((size_t)ngk);
  for (int g = 0; g < ngk; g++) order[(size_t)g] = g;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return type_width(keys[(size_t)a].field.type) < type_width(keys[(size_t)b].field.type); });
  uint64_t ranges[MAX_KEYS];
  uint64_t prod = 1;
  for (int g : order) {   // (narrow columns first: their statistics are the cheap ones)
    // (the statistics live on the table's own column when the key is one: `keys` holds copies that die with this update)
    Column& c = homes && (*homes)[(size_t)g] ? *(*homes)[(size_t)g] : keys[(size_t)g];
    const int t = c.field.type;
    if (c.validity || !(t == DFGPU_UINT8 || t == DFGPU_INT32 || t == DFGPU_DATE32 || t == DFGPU_UINT32 || t == DFGPU_INT64)) return 0;
    if (!std::atomic_load(&c.stats) && t != DFGPU_UINT8) {
      long long mm[2] = {INT64_MAX, INT64_MIN};
      BufPtr d = make_buf(16);
      h2d_async(d->ptr, mm, 16);
      switch (t) {
        case DFGPU_INT64: k_sample_minmax<int64_t><<<16, 256, 0, r.stream>>>((const int64_t*)c.ptr(), n, d->as<long long>()); break;
        case DFGPU_UINT32: k_sample_minmax<uint32_t><<<16, 256, 0, r.stream>>>((const uint32_t*)c.ptr(), n, d->as<long long>()); break;
        default: k_sample_minmax<int32_t><<<16, 256, 0, r.stream>>>((const int32_t*)c.ptr(), n, d->as<long long>()); break;
      }
      d2h(mm, d->ptr, 16);
      if ((unsigned long long)(mm[1] - mm[0]) >= DIRECT_MAX_SLOTS / prod) return 0;
    }
    const ColStats st = t == DFGPU_UINT8 ? u8_presence(c, n) : column_stats(c, n);
    if (st.valid != n) return 0;
    unsigned long long span = (unsigned long long)(st.max - st.min);
    ictx.dmin[g] = st.min;
    ictx.dcode[g] = nullptr;
    if (t == DFGPU_UINT8) {   // the values that occur, numbered: 'A', 'N', 'R' -> 0, 1, 2
      uint8_t code[256];
      unsigned next = 0;
      for (int v = 0; v < 256; v++) code[v] = (uint8_t)((st.present[v >> 6] >> (v & 63)) & 1ull ? next++ : 0u);
      if (next >= 1 && next - 1 < span) {
        span = next - 1;
        BufPtr d = make_buf(256);
        h2d_async(d->ptr, code, 256);
        DFGPU_HIP(hipStreamSynchronize(r.stream));   // (`code` is a local)
        ictx.dcode[g] = d->as<uint8_t>();
        ictx.dmin[g] = 0;
        keep.push_back(d);
      }
    }
    if (span >= DIRECT_MAX_SLOTS / prod) return 0;
    ranges[g] = span + 1;
    prod *= span + 1;
  }
  uint32_t stride = 1;
  for (int g = ngk - 1; g >= 0; g--) {
    ictx.dstride[g] = stride;
    stride *= (uint32_t)ranges[g];
  }
  return (uint32_t)prod;
}
static InternResult intern_keys(Aggregate& A, const std::vector<Column>& key_cols, int64_t n, const uint64_t* row_mask, bool want_row_slots = false,
                                const std::vector<Column*>* stat_homes = nullptr) {
  Runtime& r = rt();
  InternResult R;
  const int ngk = (int)key_cols.size();
  const int64_t G0 = A.ngroups;
  const int64_t total = G0 + n;
  DFGPU_CHECK(total < 0xFFFFFFFFll, "aggregate input exceeds u32 row ids");
  for (int g = 0; g < ngk; g++) {
    if (G0 == 0) {
      R.cat_keys.push_back(key_cols[g]);
    } else {
      const Column& gk = A.group_keys.cols[g];
      const Column& ik = key_cols[g];
      DFGPU_CHECK(gk.field.type == ik.field.type, "group key type changed between batches");
      DFGPU_CHECK(same_dictionary(gk.dict, ik.dict), "group key " + gk.name + ": the dictionary changed between updates (unify the inputs' dictionaries first, e.g. dfgpu_table_concat)");
      Column cc = alloc_column(gk.field, gk.name, total, gk.validity || ik.validity);
      cc.dict = gk.dict ? gk.dict : ik.dict;
      int w = type_width(gk.field.type);
      DFGPU_HIP(hipMemcpyAsync(cc.data->ptr, gk.ptr(), (size_t)G0 * w, hipMemcpyDeviceToDevice, r.stream));
      if (n) DFGPU_HIP(hipMemcpyAsync((char*)cc.data->ptr + (size_t)G0 * w, ik.ptr(), (size_t)n * w, hipMemcpyDeviceToDevice, r.stream));
      if (cc.validity) {
        // nullable group keys across updates (group_values/multi_group_by/mod.rs:595-745: NULL is a key value like any other): the
        // validity of [existing groups ; input rows] — a side without a bitmap is all valid
        DFGPU_HIP(hipMemsetAsync(cc.validity->ptr, 0, bitmap_bytes(total), r.stream));
        bitmap_place(gk.valid_words(), 0, G0, cc.validity->as<uint64_t>());
        if (n) bitmap_place(ik.valid_words(), G0, n, cc.validity->as<uint64_t>());
        cc.null_count = -1;
      }
      R.cat_keys.push_back(std::move(cc));
    }
  }
  InternCtx& ictx = R.ictx;
  ictx.keys.n = ngk;
  DFGPU_CHECK(ngk <= MAX_KEYS, "too many group-by columns");
  for (int g = 0; g < ngk; g++) {
    DFGPU_CHECK(R.cat_keys[g].field.type != DFGPU_BOOL, "Boolean group keys are not supported on the GPU path");
    ictx.keys.c[g] = KeyCol{R.cat_keys[g].ptr(), R.cat_keys[g].valid_words(), R.cat_keys[g].field.type, type_width(R.cat_keys[g].field.type)};
  }
  int64_t key_bytes = 0;
  for (int g = 0; g < ngk; g++) key_bytes += total * type_width(R.cat_keys[g].field.type);
  // key columns without NULLs that fit 64 bits together: a keyed table (k_intern_claim_keyed)
  bool keyed = total >= (1 << 16);
  {
    int bits = 0;
    for (int g = 0; g < ngk && keyed; g++) {
      const Column& c = R.cat_keys[g];
      keyed = !c.validity && c.field.type != DFGPU_DECIMAL128 && c.field.type != DFGPU_FLOAT64 && c.field.type != DFGPU_UTF8;
      bits += type_width(c.field.type) * 8;
    }
    keyed = keyed && bits <= 64;
  }
  // key columns whose value ranges multiply to a few thousand: a direct table (first batch of a large input only: the concatenation
  // with earlier groups' keys has no statistics)
  uint32_t direct_n = 0;
  if (G0 == 0 && total >= policy().rows_worth_a_pass() && ngk >= 1 && option_on("agg.direct_table", true))   // (A/B switch)
    direct_n = direct_table_spec(R.cat_keys, stat_homes, total, ictx, R.direct_codes);
  if (direct_n) keyed = false;
  BufPtr flag = make_zero_buf(4);
  uint64_t cap = (uint64_t)A.capacity_hint;
  const uint64_t cap_max = [&] { uint64_t c = 64; while (c < (uint64_t)total * 2) c <<= 1; return c; }();
  // First batch: size the table before the first attempt.  An attempt at too small a capacity is the expensive way to find
  // out — every row walks PROBE_LIMIT slots before the overflow flag stops the launch, and the table then grows 16x at a time
  // (SF100 Q3's 3 M joined rows / 1.1 M groups: 64 K -> 1 M -> 8 M slots, 1.6 ms; sized first: one attempt).  Small inputs
  // take the 2-slots-per-row bound (its zero-fill is microseconds); large ones intern a 1 M-row prefix into a table of their
  // own and extrapolate: distinct keys still growing with the sample => groups ~ rows x (distinct / sample); flat between
  // the first quarter and the whole sample => the sample has seen them all.
  constexpr int64_t SAMPLE = 1 << 20;
  if (direct_n) {
    cap = 64;
    while (cap < direct_n) cap <<= 1;
    R.slots = make_zero_buf((cap + 1) * 4);
    ictx.slots = R.slots->as<uint32_t>();
    ictx.mask = cap - 1;
    ictx.direct = (int)direct_n;
    BufPtr first = make_buf(cap * 4);
    DFGPU_HIP(hipMemsetAsync(first->ptr, 0xFF, cap * 4, r.stream));
    if (want_row_slots) R.row_slot = make_buf((size_t)total * 4);
    const size_t lds = (size_t)direct_n * 4 + (size_t)ngk * 256;
    bool vec = true;   // 16-byte loads of four consecutive rows: every key column on a 16-byte boundary (views of partitions may not be)
    for (int g = 0; g < ngk; g++) vec = vec && ((uintptr_t)ictx.keys.c[g].data & 15u) == 0;
    auto kern = vec ? k_intern_claim_direct<true> : k_intern_claim_direct<false>;
    DFGPU_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int per_cu = lds * 2 <= ((size_t)150 << 10) ? 2 : 1;   // (1024-thread workgroups: two per CU at most)
    const int grid = (int)std::min<int64_t>((int64_t)r.num_cus * per_cu, (total + 8191) / 8192);
    {
      ProfileScope ps("agg_intern_claim_direct", key_bytes + (R.row_slot ? total * 4 : 0));
      kern<<<grid, DIRECT_BLOCK, lds, r.stream>>>(ictx, total, direct_n, row_mask, R.row_slot ? R.row_slot->as<uint32_t>() : nullptr, first->as<uint32_t>());
      k_direct_slots_from_rows<<<grid_for((int64_t)cap, BLOCK), BLOCK, 0, r.stream>>>(first->as<uint32_t>(), cap, ictx.slots);
      DFGPU_HIP(hipGetLastError());
    }
    DFGPU_HIP(hipStreamSynchronize(r.stream));   // (`first` is a local)
  }
  if (G0 == 0 && !direct_n) {
    if (total <= 4 * SAMPLE) {
      cap = cap_max;
    } else if (!row_mask) {
      BufPtr sslots = make_zero_buf((size_t)(2 * SAMPLE) * 4);
      BufPtr cnt = make_zero_buf(16);
      InternCtx sc = ictx;   // (the sample goes through the column-by-column table whatever the real attempts use)
      sc.slots = sslots->as<uint32_t>();
      sc.mask = 2 * SAMPLE - 1;
      {
        ProfileScope ps("agg_intern_sample", key_bytes / total * SAMPLE);
        k_intern_claim<<<grid_for(SAMPLE, BLOCK), BLOCK, 0, r.stream>>>(sc, SAMPLE, flag->as<int>(), nullptr, 0);
        k_count_slots<<<grid_for(2 * SAMPLE, BLOCK), BLOCK, 0, r.stream>>>(sc.slots, (uint64_t)(2 * SAMPLE), (uint32_t)(SAMPLE / 4), cnt->as<unsigned long long>());
        DFGPU_HIP(hipGetLastError());
      }
      unsigned long long c2[2] = {0, 0};
      d2h(c2, cnt->ptr, 16);
      DFGPU_HIP(hipMemsetAsync(flag->ptr, 0, 4, r.stream));  // (a 2x table cannot overflow; the flag is shared with the real attempts)
      const double d_all = (double)c2[0], d_early = (double)c2[1];
      const double est = d_all < 1.25 * d_early ? 2.0 * d_all : d_all / (double)SAMPLE * (double)total;
      // (round 3 left two opt-in variants here — a table sized down to ~4 slots per key, and a barrier-free claim pass that kept the met
      // keys in LDS — for the three-key aggregate over 600 M rows; measured in round 4 (profiles/r4_agg_knobs.md): 15.1 ms either way
      // against 13.6 ms without them, so they are gone)
      uint64_t want = 1 << 16;
      while ((double)want < 3.0 * est && want < cap_max) want <<= 1;
      cap = std::max<uint64_t>(cap, want);
    }
  }
  if (cap > cap_max && !direct_n) cap = cap_max;
  while (!direct_n) {
    R.slots = make_zero_buf((cap + 1) * 4);   // (+ 1: the keyed table's slot for the all-ones key)
    ictx.slots = R.slots->as<uint32_t>();
    ictx.mask = cap - 1;
    if (keyed) {
      R.keyed = make_buf((cap + 1) * sizeof(KeyedSlot));
      ictx.keyed = R.keyed->as<KeyedSlot>();
      k_keyed_clear<<<grid_for((int64_t)cap + 1, BLOCK), BLOCK, 0, r.stream>>>(ictx.keyed, cap + 1);
    }
    if (total) {
      ProfileScope ps(keyed ? "agg_intern_claim_keyed" : "agg_intern_claim", key_bytes);
      if (want_row_slots && !R.row_slot) R.row_slot = make_buf((size_t)total * 4);
      uint32_t* rs = R.row_slot ? R.row_slot->as<uint32_t>() : nullptr;
      if (keyed) {
        k_intern_claim_keyed<<<grid_for((total + KEYED_ROWS - 1) / KEYED_ROWS, BLOCK), BLOCK, 0, r.stream>>>(ictx, total, flag->as<int>(), row_mask, G0, rs);
        k_keyed_rows<<<grid_for((int64_t)cap + 1, BLOCK), BLOCK, 0, r.stream>>>(ictx.keyed, cap + 1, ictx.slots);
      }
      else k_intern_claim<<<grid_for(total, BLOCK), BLOCK, 0, r.stream>>>(ictx, total, flag->as<int>(), row_mask, G0, rs);
      DFGPU_HIP(hipGetLastError());
    }
    int ovf = 0;
    d2h(&ovf, flag->ptr, 4);
    if (!ovf) break;
    DFGPU_CHECK(cap < cap_max, "group table overflow at maximum capacity");
    cap = std::min<uint64_t>(cap_max, cap * 16);
    DFGPU_HIP(hipMemsetAsync(flag->ptr, 0, 4, r.stream));
  }
  A.capacity_hint = (int64_t)cap;
  const int64_t n_words = (total + 63) / 64;
  BufPtr rep_mask = make_zero_buf((size_t)(n_words ? n_words : 1) * 8);
  BufPtr prefix = make_buf((size_t)(n_words + 1) * 8);
  const uint64_t n_slots = cap + (keyed ? 1 : 0);
  k_mark_reps<<<grid_for((int64_t)n_slots, BLOCK), BLOCK, 0, r.stream>>>(R.slots->as<uint32_t>(), n_slots, rep_mask->as<unsigned long long>());
  scan_mask_popcounts(rep_mask->as<uint64_t>(), nullptr, total, prefix->as<uint64_t>());
  R.G1 = (int64_t)read_u64(prefix->as<uint64_t>() + n_words);
  R.slot_gid = make_buf((cap + 1) * 4);
  BufPtr rep_row = make_buf((size_t)(R.G1 ? R.G1 : 1) * 8);
  k_slot_gids<<<grid_for((int64_t)n_slots, BLOCK), BLOCK, 0, r.stream>>>(R.slots->as<uint32_t>(), n_slots, rep_mask->as<uint64_t>(), prefix->as<uint64_t>(),
                                                                      R.slot_gid->as<uint32_t>(), rep_row->as<int64_t>());
  DFGPU_HIP(hipGetLastError());
  // new dense group key columns = representative rows (first-seen order)
  Table gk;
  gk.nrows = R.G1;
  for (int g = 0; g < ngk; g++) gk.cols.push_back(gather_column(R.cat_keys[g], rep_row->as<int64_t>(), R.G1, false));
  A.group_keys = std::move(gk);
  return R;
}

// grow every accumulator to G1 groups; returns the accumulation plan per aggregate
// init = false: the new cells are left uninitialised (a caller that writes all G1 of them itself)
__global__ __launch_bounds__(BLOCK) void k_split128(const unsigned long long* __restrict__ cells, int64_t n, unsigned long long* __restrict__ lo, unsigned long long* __restrict__ hi) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    lo[i] = cells[2 * i];
    hi[i] = cells[2 * i + 1];
  }
}
// the seen words the runs node left unwritten (every group has a value) exist again: before any other update reads or extends them
static void materialize_seen(Aggregate& A, int64_t G) {
  for (AggState& a : A.aggs) {
    if (!a.seen_all) continue;
    if (G > 0 && a.seen) DFGPU_HIP(hipMemsetD32Async((hipDeviceptr_t)a.seen->ptr, 1, (size_t)G, rt().stream));
    a.seen_all = false;
  }
}
static void split_interleaved(Aggregate& A, int64_t G) {
  materialize_seen(A, G);
  for (AggState& a : A.aggs) {
    if (!a.inter) continue;
    if (G > 0) {
      k_split128<<<grid_for(G, BLOCK), BLOCK, 0, rt().stream>>>(a.inter->as<unsigned long long>(), G, a.lo->as<unsigned long long>(), a.hi->as<unsigned long long>());
      DFGPU_HIP(hipGetLastError());
    }
    a.inter.reset();  // (stream-ordered pool: the kernel above runs before the buffer is reused)
  }
}

// the fresh accumulators of a FEW groups (a first batch's: nothing to copy) filled by ONE launch — blockIdx.y is the array.  Q1's eight
// aggregates were 28 fill launches a step, 0.13 ms behind a kernel of 7.4
constexpr int FM_MAX = 4 * MAX_AGGS;
struct FillMany {
  void* p[FM_MAX];
  unsigned long long v[FM_MAX];
  int elem[FM_MAX];
};
__global__ __launch_bounds__(BLOCK) void k_fill_many(FillMany f, int64_t n) {
  const int c = blockIdx.y;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    if (f.elem[c] == 8) reinterpret_cast<unsigned long long*>(f.p[c])[i] = f.v[c];
    else reinterpret_cast<uint32_t*>(f.p[c])[i] = (uint32_t)f.v[c];
  }
}
static std::vector<AccPlan> grow_accumulators(Aggregate& A, int64_t G0, int64_t G1, bool init = true) {
  std::vector<AccPlan> plans;
  const bool final_mode = A.final_mode();
  split_interleaved(A, G0);
  if (init && G0 == 0 && G1 > 0 && G1 <= 65536 && (int)A.aggs.size() <= MAX_AGGS) {
    FillMany fm{};
    int m = 0;
    auto fresh = [&](BufPtr& b, unsigned long long fill, int elem) {
      b = make_buf((size_t)G1 * elem);
      fm.p[m] = b->ptr;
      fm.v[m] = fill;
      fm.elem[m] = elem;
      m++;
    };
    for (AggState& a : A.aggs) {
      AccPlan p = plan_for(a.func, a.in_type, final_mode);
      fresh(a.lo, acc_identity(p.kind), 8);
      if (p.needs_hi) fresh(a.hi, 0ull, 8);
      fresh(a.seen, 0ull, 4);
      if (a.func == DFGPU_AGG_AVG) fresh(a.cnt, 0ull, 8);
      plans.push_back(p);
    }
    if (m) {
      k_fill_many<<<dim3((unsigned)grid_for(G1, BLOCK), (unsigned)m), BLOCK, 0, rt().stream>>>(fm, G1);
      DFGPU_HIP(hipGetLastError());
    }
    return plans;
  }
  for (AggState& a : A.aggs) {
    AccPlan p = plan_for(a.func, a.in_type, final_mode);
    a.lo = grown(a.lo, G0, G1, acc_identity(p.kind), 8, init);
    if (p.needs_hi) a.hi = grown(a.hi, G0, G1, 0ull, 8, init);
    a.seen = grown(a.seen, G0, G1, 0, 4, init);
    if (a.func == DFGPU_AGG_AVG) a.cnt = grown(a.cnt, G0, G1, 0ull, 8, init);
    plans.push_back(p);
  }
  return plans;
}

static bool is_plain_column(const std::vector<dfgpu_expr_node>& nodes, int root, int* col) {
  if (root < 0 || root >= (int)nodes.size() || nodes[root].op != DFGPU_EXPR_COLUMN) return false;
  *col = nodes[root].column;
  return true;
}

// Small-domain interning (every group key is a non-null 1-byte column): returns false when not applicable.
static bool small_domain_applicable(const Aggregate& A, const Table& in, std::vector<int>& key_cols) {
  const int ngk = (int)A.group_roots.size();
  if (ngk < 1 || ngk > 2) return false;
  key_cols.clear();
  for (int g = 0; g < ngk; g++) {
    int c = -1;
    if (!is_plain_column(A.group_nodes[g], A.group_roots[g], &c)) return false;
    if (c < 0 || c >= (int)in.cols.size()) return false;
    if (in.cols[c].field.type != DFGPU_UINT8 || in.cols[c].validity) return false;
    key_cols.push_back(c);
  }
  if (A.ngroups > 0) {
    for (int g = 0; g < ngk; g++)
      if (A.group_keys.cols[g].field.type != DFGPU_UINT8 || A.group_keys.cols[g].validity) return false;
  }
  return true;
}



// rebuilds A.group_keys (dense, gid order) from the host mirror of the 1-byte keys
static void small_rebuild_group_keys(Aggregate& A, const Table& in, const std::vector<int>& small_cols) {
  Runtime& r = rt();
  const int ngk = (int)small_cols.size();
  const int64_t G1 = (int64_t)A.small_keys.size();
  Table gk;
  gk.nrows = G1;
  for (int g = 0; g < ngk; g++) {
    Column c = alloc_column(in.cols[small_cols[g]].field, A.group_names[g], G1);
    c.dict = in.cols[small_cols[g]].dict;
    if (G1 > 0 && G1 <= SMALL_KG_MAX) {   // a handful of groups: the key bytes travel as kernel arguments (no upload to wait for)
      SmallBytes sb{};
      for (int64_t i = 0; i < G1; i++) sb.v[(size_t)i] = (uint8_t)(A.small_keys[(size_t)i] >> (8 * g));
      k_set_bytes_v<<<1, SMALL_KG_MAX, 0, r.stream>>>(sb, (int)G1, c.data->as<uint8_t>());
      DFGPU_HIP(hipGetLastError());
      gk.cols.push_back(std::move(c));
      continue;
    }
    std::vector<uint8_t> b((size_t)(G1 ? G1 : 1));
    for (int64_t i = 0; i < G1; i++) b[(size_t)i] = (uint8_t)(A.small_keys[(size_t)i] >> (8 * g));
    if (G1) h2d_async(c.data->ptr, b.data(), (size_t)G1);
    DFGPU_HIP(hipStreamSynchronize(r.stream));  // b is a host temporary
    gk.cols.push_back(std::move(c));
  }
  A.group_keys = std::move(gk);
}
static void small_sync_host_keys(Aggregate& A, int ngk) {
  const int64_t G0 = A.ngroups;
  if ((int64_t)A.small_keys.size() == G0) return;
  A.small_keys.assign((size_t)G0, 0);
  for (int g = 0; g < ngk && G0; g++) {
    std::vector<uint8_t> b((size_t)G0);
    d2h(b.data(), A.group_keys.cols[g].ptr(), (size_t)G0);
    for (int64_t i = 0; i < G0; i++) A.small_keys[(size_t)i] |= (uint16_t)(b[(size_t)i] << (8 * g));
  }
}

// ---------------------------------------------------------------- runtime-specialised node (jit.hip)
// Argument block of the generated kernel: the layout below is repeated textually in the generated source.
struct AggNodeArgs {
  const void* col[RP_MAX_COLS];
  const uint64_t* valid[RP_MAX_COLS];
  unsigned long long* tmp_lo[MAX_AGGS];
  unsigned long long* tmp_hi[MAX_AGGS];
  uint32_t* g_first;
  uint32_t* g_seen;
  int64_t begin, end;
  int L, nrep;
};
static_assert(RP_MAX_COLS == 10 && MAX_AGGS == 16, "update the Args struct in agg_node_source");

// HIP source of the single-pass small-domain node for ONE expression forest and accumulator set: the skeleton
// is k_agg_fused_tile's (local key slots, limb cells, replica fold, key-indexed partials); the interpreter is
// replaced by the forest's straight-line code and every per-accumulator switch by its one live arm.
static std::string agg_node_source(const CompiledProgram& cp, const SmallAccSet& accs, const std::vector<int>& acc_val, int key_val0, int key_val1) {
  auto S = [](long long v) { return std::to_string(v); };
  std::string src;
  src += R"SRC(
typedef __int128 i128;
typedef unsigned __int128 u128;
typedef unsigned long long U64;
typedef long long I64;
typedef unsigned int U32;
typedef int I32;
typedef unsigned char U8;
#define BLOCK 256
#define LIMB_MASK ((1ull << 43) - 1ull)
struct Args {
  const void* col[10];
  const U64* valid[10];
  U64* tmp_lo[16];
  U64* tmp_hi[16];
  unsigned int* g_first;
  unsigned int* g_seen;
  long long begin, end;
  int L, nrep;
};
enum { SUM_I64 = 0, SUM_I128 = 1, SUM_F64 = 2, MIN_I64 = 3, MAX_I64 = 4, COUNT = 5, COUNT_STAR = 6 };
__device__ __forceinline__ double v2f(i128 x) { return __longlong_as_double((long long)(U64)x); }
__device__ __forceinline__ i128 f2v(double d) { return (i128)(u128)(U64)__double_as_longlong(d); }
__device__ __forceinline__ long long f64ord(U64 bits) { long long b = (long long)bits; return b ^ (long long)((U64)(b >> 63) >> 1); }
__device__ __forceinline__ I32 date32_part(I32 days, int part) {  // device.hpp date32_part
  const I64 z = (I64)days + 719468, era = (z >= 0 ? z : z - 146096) / 146097, doe = z - era * 146097;
  const I64 yoe = (doe - doe / 1460 + doe / 36524 - doe / 146096) / 365, doy = doe - (365 * yoe + yoe / 4 - yoe / 100), mp = (5 * doy + 2) / 153;
  const I64 d = doy - (153 * mp + 2) / 5 + 1, m = mp < 10 ? mp + 3 : mp - 9, y = yoe + era * 400 + (m <= 2 ? 1 : 0);
  return (I32)(part == 0 ? y : part == 1 ? m : d);
}
__device__ __forceinline__ bool kt(i128 v, bool n) { return !n && ((int)v & 1); }
__device__ __forceinline__ bool kf(i128 v, bool n) { return !n && !((int)v & 1); }
__device__ __forceinline__ U64 identity_of(int kind) {
  return kind == MIN_I64 ? 0x7fffffffffffffffull : kind == MAX_I64 ? 0x8000000000000000ull : 0ull;
}
__device__ __forceinline__ int local_slot(unsigned int* s_key, int L, unsigned int key) {
  const unsigned int want = key + 1u;
  int slot = (int)((key * 0x9E3779B1u) >> 20) & (L - 1);
  for (int probe = 0; probe < L; probe++) {
    unsigned int cur = s_key[slot];
    if (cur == want) return slot;
    if (cur == 0u) {
      cur = atomicCAS(&s_key[slot], 0u, want);
      if (cur == 0u || cur == want) return slot;
    }
    slot = (slot + 1) & (L - 1);
  }
  return -1;
}
__device__ __forceinline__ void global_accumulate(int kind, U64* lo_cell, U64* hi_cell, U64 lo, U64 hi) {
  switch (kind) {
    case SUM_I64: atomicAdd(lo_cell, lo); break;
    case SUM_I128: {
      U64 old = atomicAdd(lo_cell, lo);
      atomicAdd(hi_cell, hi + ((old + lo) < old ? 1ull : 0ull));
      break;
    }
    case SUM_F64: atomicAdd(reinterpret_cast<double*>(lo_cell), __longlong_as_double((long long)lo)); break;
    case MIN_I64: atomicMin(reinterpret_cast<long long*>(lo_cell), (long long)lo); break;
    case MAX_I64: atomicMax(reinterpret_cast<long long*>(lo_cell), (long long)lo); break;
    default: atomicAdd(lo_cell, 1ull); break;
  }
}
)SRC";
  src += "#define NACC " + S(accs.n) + "\n#define NCELL " + S(accs.ncell) + "\n";
  src += "__device__ __forceinline__ int acc_kind(int k) { switch (k) {";
  for (int k = 0; k < accs.n; k++) src += " case " + S(k) + ": return " + S(accs.a[k].kind) + ";";
  src += " default: return 0; } }\n";
  src += "__device__ __forceinline__ int acc_cell0(int k) { switch (k) {";
  for (int k = 0; k < accs.n; k++) src += " case " + S(k) + ": return " + S(accs.a[k].cell0) + ";";
  src += " default: return 0; } }\n";
  src += "__device__ __forceinline__ int cell_kind(int c) { switch (c) {";
  for (int c = 0; c < accs.ncell; c++) src += " case " + S(c) + ": return " + S(accs.cell_kind[c]) + ";";
  src += " default: return 0; } }\n";
  src += R"SRC(
extern "C" __global__ __launch_bounds__(BLOCK) void agg_node(Args a) {
  extern __shared__ U64 s_cell[];  // cell(c, slot, rep) = (c * L + slot) * nrep + rep
  const int L = a.L, nrep = a.nrep, plane = L * nrep, ncells = NCELL * plane;
  unsigned int* s_seen = (unsigned int*)(s_cell + ncells);
  unsigned int* s_key = s_seen + plane;
  unsigned int* s_first = s_key + L;
  for (int x = threadIdx.x; x < ncells; x += BLOCK) s_cell[x] = identity_of(cell_kind(x / plane));
  for (int x = threadIdx.x; x < plane; x += BLOCK) s_seen[x] = 0u;
  for (int x = threadIdx.x; x < L; x += BLOCK) { s_key[x] = 0u; s_first[x] = 0xFFFFFFFFu; }
  __syncthreads();
  const int rep = (int)(threadIdx.x & (unsigned)(nrep - 1));
  const long long stride = (long long)gridDim.x * BLOCK;
  for (long long i = a.begin + (long long)blockIdx.x * BLOCK + threadIdx.x; i < a.end; i += stride) {
)SRC";
  src += cp.src_loads;
  src += cp.src_pred;
  if (cp.src_pred_val >= 0) src += "    if (N" + S(cp.src_pred_val) + " || !((int)V" + S(cp.src_pred_val) + " & 1)) continue;\n";
  src += cp.src_outs;
  src += "    unsigned int key = 0u;\n";
  if (key_val0 >= 0) src += "    key = (unsigned int)V" + S(key_val0) + " & 0xFFu;\n";
  if (key_val1 >= 0) src += "    key |= ((unsigned int)V" + S(key_val1) + " & 0xFFu) << 8;\n";
  src += R"SRC(    const int slot = local_slot(s_key, L, key);
    unsigned int seen = 0u;
    if (slot >= 0) {
      if ((unsigned int)i < s_first[slot]) atomicMin(&s_first[slot], (unsigned int)i);
      const int base = slot * nrep + rep;
)SRC";
  auto value_of = [&](int k, std::string& vlo, std::string& vhi, std::string& guard) {
    if (acc_val[k] >= 0) {
      vlo = "(U64)V" + S(acc_val[k]);
      vhi = "(U64)((u128)V" + S(acc_val[k]) + " >> 64)";
      guard = "if (!N" + S(acc_val[k]) + ") ";
    } else {
      vlo = vhi = "0ull";
      guard = "";
    }
  };
  for (int k = 0; k < accs.n; k++) {
    std::string vlo, vhi, guard;
    value_of(k, vlo, vhi, guard);
    const std::string c = "(s_cell + " + S(accs.a[k].cell0) + " * plane + base)";
    src += "      " + guard + "{ seen |= " + S(1u << k) + "u; ";
    switch (accs.a[k].kind) {
      case ACC_SUM_I128:
        src += "const U64 lo = " + vlo + ", hi = " + vhi + "; U64* c = " + c + "; atomicAdd(c, lo & LIMB_MASK); atomicAdd(c + plane, ((lo >> 43) | (hi << 21)) & LIMB_MASK); "
               "atomicAdd(c + 2 * plane, (U64)((long long)hi >> 22));";
        break;
      case ACC_SUM_I64: src += "atomicAdd(" + c + ", " + vlo + ");"; break;
      case ACC_SUM_F64: src += "atomicAdd(reinterpret_cast<double*>" + c + ", __longlong_as_double((long long)" + vlo + "));"; break;
      case ACC_MIN_I64: src += "atomicMin(reinterpret_cast<long long*>" + c + ", (long long)" + vlo + ");"; break;
      case ACC_MAX_I64: src += "atomicMax(reinterpret_cast<long long*>" + c + ", (long long)" + vlo + ");"; break;
      default: src += "atomicAdd(" + c + ", 1ull);"; break;
    }
    src += " }\n";
  }
  src += "      if (seen & ~s_seen[base]) atomicOr(&s_seen[base], seen);\n    } else {\n";
  src += "      if ((unsigned int)i < a.g_first[key]) atomicMin(&a.g_first[key], (unsigned int)i);\n";
  for (int k = 0; k < accs.n; k++) {
    std::string vlo, vhi, guard;
    value_of(k, vlo, vhi, guard);
    src += "      " + guard + "{ seen |= " + S(1u << k) + "u; global_accumulate(" + S(accs.a[k].kind) + ", a.tmp_lo[" + S(k) + "] + key, a.tmp_hi[" + S(k) +
           "] ? a.tmp_hi[" + S(k) + "] + key : nullptr, " + vlo + ", " + vhi + "); }\n";
  }
  src += R"SRC(      if (seen) atomicOr(&a.g_seen[key], seen);
    }
  }
  __syncthreads();
  for (int x = threadIdx.x; x < L * NACC; x += BLOCK) {
    const int slot = x / NACC, k = x % NACC;
    const unsigned int kv = s_key[slot];
    if (!kv) continue;
    const unsigned int key = kv - 1u;
    unsigned int seen = 0u;
    for (int q = 0; q < nrep; q++) seen |= s_seen[slot * nrep + q];
    if (!((seen >> k) & 1u)) continue;
    const U64* c = s_cell + acc_cell0(k) * plane + slot * nrep;
    int kind = acc_kind(k);
    U64 lo = identity_of(kind), hi = 0ull;
    if (kind == SUM_I128) {
      U64 s0 = 0, s1 = 0, s2 = 0;
      for (int q = 0; q < nrep; q++) { s0 += c[q]; s1 += c[plane + q]; s2 += c[2 * plane + q]; }
      u128 v = (u128)s0 + ((u128)s1 << 43) + (u128)((i128)(long long)s2 << 86);
      lo = (U64)v;
      hi = (U64)(v >> 64);
    } else {
      for (int q = 0; q < nrep; q++) {
        const U64 v = c[q];
        switch (kind) {
          case SUM_F64: lo = (U64)__double_as_longlong(__longlong_as_double((long long)lo) + __longlong_as_double((long long)v)); break;
          case MIN_I64: lo = (U64)min((long long)lo, (long long)v); break;
          case MAX_I64: lo = (U64)max((long long)lo, (long long)v); break;
          default: lo += v; break;
        }
      }
      if (kind == COUNT || kind == COUNT_STAR) kind = SUM_I64;
    }
    global_accumulate(kind, a.tmp_lo[k] + key, a.tmp_hi[k] ? a.tmp_hi[k] + key : nullptr, lo, hi);
    atomicOr(&a.g_seen[key], 1u << k);
  }
  for (int slot = threadIdx.x; slot < L; slot += BLOCK)
    if (s_key[slot] && s_first[slot] != 0xFFFFFFFFu) atomicMin(&a.g_first[s_key[slot] - 1u], s_first[slot]);
}
)SRC";
  return src;
}

// ---------------------------------------------------------------- specialised node: one dense integer group key
// GROUP BY <integer expression> with many groups (TPC-H: l_orderkey, o_custkey, l_partkey ...).  Hash interning
// costs several random HBM accesses per row (slot, representative row's key, group id, accumulator): 125 ms for
// 600 M rows / 150 M groups.  Integer keys that fill at least 1/64 of their value range are interned by RANK
// instead, the structure of the join's rank map (join.hip): one bit per value of the range, group number =
// number of set bits below the key's bit (popcount directory) — a 600 M-value range is a 75 MB bitmap that stays
// in MALL / L2, and group numbers ascend with the key, so clustered input accumulates into neighbouring cells.
//   minmax     : range of the key over the passing rows                      (generated, reads predicate + key)
//   setbits    : bitmap[key - min] = 1                                        (generated)
//   scan       : exclusive popcount prefix per bitmap word                    (scan.hip)
//   accumulate : g = prefix[w] + popc(bits[w] & below); first_row[g] = min(row); cells[.][g] op= value  (generated)
//   emit       : groups renumbered by first row (the reference's first-seen order), keys rebuilt from bit positions
struct DenseNodeArgs {
  const void* col[RP_MAX_COLS];
  const uint64_t* valid[RP_MAX_COLS];
  long long* minmax;             // [0] min, [1] max of the key
  uint32_t* flags;               // [0] any NULL key among the passing rows, [1] passing rows with a key
  unsigned long long* bits;      // bitmap over [kmin, kmax]
  const uint64_t* prefix;        // exclusive popcount prefix per bitmap word
  uint32_t* first_row;           // [G]
  unsigned long long* cells;     // [ncellwords][G]
  uint32_t* seen;                // [G]
  int64_t kmin;
  int64_t G;                     // groups incl. the NULL group (last) when present
  int64_t null_group;            // group number of the NULL key, -1 = none
  int64_t begin, end;
};
struct DenseAcc {
  int kind;  // AccKind
  int val;   // value id in the generated source, -1 = none (COUNT(*))
  int cell;  // cell word (i128 sums: lo at cell, hi at cell + 1)
};

static std::string agg_dense_node_source(const CompiledProgram& cp, int key_val, const std::vector<DenseAcc>& accs) {
  auto S = [](long long v) { return std::to_string(v); };
  std::string row;  // per-row prologue shared by the three kernels (unused values are dead code in minmax / setbits)
  row += cp.src_loads;
  row += cp.src_pred;
  if (cp.src_pred_val >= 0) row += "    if (N" + S(cp.src_pred_val) + " || !((int)V" + S(cp.src_pred_val) + " & 1)) continue;\n";
  row += cp.src_outs;
  row += "    const bool knull = N" + S(key_val) + ";\n    const long long key = (long long)(U64)V" + S(key_val) + ";\n";
  std::string src = R"SRC(
typedef __int128 i128;
typedef unsigned __int128 u128;
typedef unsigned long long U64;
typedef long long I64;
typedef unsigned int U32;
typedef int I32;
typedef unsigned char U8;
#define BLOCK 256
struct Args {
  const void* col[10];
  const U64* valid[10];
  long long* minmax;
  U32* flags;
  U64* bits;
  const U64* prefix;
  U32* first_row;
  U64* cells;
  U32* seen;
  long long kmin;
  long long G;
  long long null_group;
  long long begin, end;
};
__device__ __forceinline__ double v2f(i128 x) { return __longlong_as_double((long long)(U64)x); }
__device__ __forceinline__ i128 f2v(double d) { return (i128)(u128)(U64)__double_as_longlong(d); }
__device__ __forceinline__ long long f64ord(U64 bits) { long long b = (long long)bits; return b ^ (long long)((U64)(b >> 63) >> 1); }
__device__ __forceinline__ I32 date32_part(I32 days, int part) {  // device.hpp date32_part
  const I64 z = (I64)days + 719468, era = (z >= 0 ? z : z - 146096) / 146097, doe = z - era * 146097;
  const I64 yoe = (doe - doe / 1460 + doe / 36524 - doe / 146096) / 365, doy = doe - (365 * yoe + yoe / 4 - yoe / 100), mp = (5 * doy + 2) / 153;
  const I64 d = doy - (153 * mp + 2) / 5 + 1, m = mp < 10 ? mp + 3 : mp - 9, y = yoe + era * 400 + (m <= 2 ? 1 : 0);
  return (I32)(part == 0 ? y : part == 1 ? m : d);
}
__device__ __forceinline__ bool kt(i128 v, bool n) { return !n && ((int)v & 1); }
__device__ __forceinline__ bool kf(i128 v, bool n) { return !n && !((int)v & 1); }
// Segmented inclusive scans over the lanes of a wave: a run = adjacent lanes holding the same group (`head` marks
// its first lane).  The run's last lane ends up with the run's total, so clustered input issues ONE atomic per run
// and accumulator instead of one per row.  Every lane of the wave must be executing.
#define SEG_SCAN(STEP)                                        \
  bool f = head;                                              \
  const int lane = threadIdx.x & 63;                          \
  _Pragma("unroll") for (int d = 1; d < 64; d <<= 1) {        \
    const int fo = __shfl_up((int)f, d, 64);                  \
    STEP                                                      \
    if (lane >= d && !f) f = fo != 0;                         \
  }
__device__ __forceinline__ U64 seg_add_u64(U64 v, bool head) {
  SEG_SCAN(const U64 o = __shfl_up(v, d, 64); if (lane >= d && !f) v += o;)
  return v;
}
__device__ __forceinline__ u128 seg_add_u128(u128 v, bool head) {
  SEG_SCAN(const U64 ol = __shfl_up((U64)v, d, 64); const U64 oh = __shfl_up((U64)(v >> 64), d, 64); if (lane >= d && !f) v += ((u128)oh << 64) | ol;)
  return v;
}
__device__ __forceinline__ double seg_add_f64(double v, bool head) {
  SEG_SCAN(const double o = __shfl_up(v, d, 64); if (lane >= d && !f) v += o;)
  return v;
}
__device__ __forceinline__ long long seg_min_i64(long long v, bool head) {
  SEG_SCAN(const long long o = __shfl_up(v, d, 64); if (lane >= d && !f) v = o < v ? o : v;)
  return v;
}
__device__ __forceinline__ long long seg_max_i64(long long v, bool head) {
  SEG_SCAN(const long long o = __shfl_up(v, d, 64); if (lane >= d && !f) v = o > v ? o : v;)
  return v;
}

extern "C" __global__ __launch_bounds__(BLOCK) void dense_minmax(Args a) {
  long long mn = 0x7fffffffffffffffll, mx = -0x7fffffffffffffffll - 1;
  unsigned any_null = 0u, rows = 0u;
  const long long stride = (long long)gridDim.x * BLOCK;
  for (long long i = a.begin + (long long)blockIdx.x * BLOCK + threadIdx.x; i < a.end; i += stride) {
)SRC";
  src += row;
  src += R"SRC(    if (knull) { any_null = 1u; continue; }
    rows = 1u;
    mn = key < mn ? key : mn;
    mx = key > mx ? key : mx;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const long long omn = __shfl_xor(mn, d, 64), omx = __shfl_xor(mx, d, 64);
    mn = omn < mn ? omn : mn;
    mx = omx > mx ? omx : mx;
    any_null |= __shfl_xor(any_null, d, 64);
    rows |= __shfl_xor(rows, d, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    if (rows) {
      atomicMin(a.minmax, mn);
      atomicMax(a.minmax + 1, mx);
      atomicOr(a.flags + 1, 1u);
    }
    if (any_null) atomicOr(a.flags, 1u);
  }
}

// Adjacent lanes that target the same bitmap word are merged (segmented OR) into ONE atomic per word and wave;
// bits only ever go 0 -> 1, so the plain (possibly stale) read is a safe filter: it can only cause a redundant atomic.
__device__ __forceinline__ U64 seg_or_u64(U64 v, bool head) {
  SEG_SCAN(const U64 o = __shfl_up(v, d, 64); if (lane >= d && !f) v |= o;)
  return v;
}
extern "C" __global__ __launch_bounds__(BLOCK) void dense_setbits(Args a) {
  const long long stride = (long long)gridDim.x * BLOCK;
  const long long n_round = a.begin + (a.end - a.begin + BLOCK - 1) / BLOCK * BLOCK;
  for (long long i = a.begin + (long long)blockIdx.x * BLOCK + threadIdx.x; i < n_round; i += stride) {
    long long idx = -1;
    if (i < a.end) do {
)SRC";
  src += row;
  src += R"SRC(      if (!knull) idx = key - a.kmin;
    } while (0);
    const long long w = idx >= 0 ? (idx >> 6) : -1 - (long long)(threadIdx.x & 63);
    const long long wprev = __shfl_up(w, 1, 64), wnext = __shfl_down(w, 1, 64);
    const bool head = (threadIdx.x & 63) == 0 || wprev != w;
    const bool tail = (threadIdx.x & 63) == 63 || wnext != w;
    const U64 acc = seg_or_u64(idx >= 0 ? 1ull << (idx & 63) : 0ull, head);
    if (idx >= 0 && tail) {
      U64* p = a.bits + w;
      if ((*p & acc) != acc) atomicOr(p, acc);
    }
  }
}

extern "C" __global__ __launch_bounds__(BLOCK) void dense_accumulate(Args a) {
  const long long stride = (long long)gridDim.x * BLOCK;
  const long long n_round = a.begin + (a.end - a.begin + BLOCK - 1) / BLOCK * BLOCK;
  const int lane_ = threadIdx.x & 63;
  for (long long i = a.begin + (long long)blockIdx.x * BLOCK + threadIdx.x; i < n_round; i += stride) {
    long long g = -1 - (long long)lane_;  // dropped rows: a run of their own, nothing to add
)SRC";
  for (size_t k = 0; k < accs.size(); k++) {
    src += "    U64 X" + S((long long)k) + "lo = 0ull, X" + S((long long)k) + "hi = 0ull; bool X" + S((long long)k) + "ok = false;\n";
  }
  src += "    if (i < a.end) do {\n";
  src += row;
  src += R"SRC(      g = a.null_group;
      if (!knull) {
        const U64 idx = (U64)(key - a.kmin);
        g = (long long)(a.prefix[idx >> 6] + __popcll(a.bits[idx >> 6] & ((1ull << (idx & 63)) - 1ull)));
      }
)SRC";
  for (size_t k = 0; k < accs.size(); k++) {
    const DenseAcc& c = accs[k];
    const std::string X = "X" + S((long long)k);
    if (c.val >= 0) {
      src += "      " + X + "ok = !N" + S(c.val) + "; " + X + "lo = (U64)V" + S(c.val) + "; " + X + "hi = (U64)((u128)V" + S(c.val) + " >> 64);\n";
    } else {
      src += "      " + X + "ok = true;\n";
    }
  }
  src += R"SRC(    } while (0);
    const long long gprev = __shfl_up(g, 1, 64), gnext = __shfl_down(g, 1, 64);
    const bool head = lane_ == 0 || gprev != g;
    const bool tail = (lane_ == 63 || gnext != g) && g >= 0;
    const U64 run_len = seg_add_u64(1ull, head);
    if (tail) {
      const unsigned first = (unsigned)(i - (long long)run_len + 1);  // the run's rows are consecutive
      unsigned* fr = a.first_row + g;
      if (first < *fr) atomicMin(fr, first);
    }
    unsigned seen = 0u;
)SRC";
  for (size_t k = 0; k < accs.size(); k++) {
    const DenseAcc& c = accs[k];
    const std::string X = "X" + S((long long)k);
    const bool maybe_null = c.val >= 0 && cp.src_maybe_null[(size_t)c.val];
    const std::string cell = "(a.cells + " + S(c.cell) + " * a.G + g)";
    // number of contributing rows of the run: the run length unless the value can be NULL
    const std::string cnt = maybe_null ? "seg_add_u64(" + X + "ok ? 1ull : 0ull, head)" : "run_len";
    src += "    {\n      const U64 cnt = " + cnt + ";\n";
    switch (c.kind) {
      case ACC_SUM_I128:
        src += "      const u128 t = seg_add_u128(" + X + "ok ? (((u128)" + X + "hi << 64) | " + X + "lo) : (u128)0, head);\n"
               "      if (tail && cnt) { seen |= " + S(1ll << k) + "u; const U64 lo = (U64)t, hi = (U64)(t >> 64); const U64 old = atomicAdd(" + cell +
               ", lo); atomicAdd(" + cell + " + a.G, hi + ((old + lo) < old ? 1ull : 0ull)); }\n";
        break;
      case ACC_SUM_I64:
        src += "      const U64 t = seg_add_u64(" + X + "ok ? " + X + "lo : 0ull, head);\n      if (tail && cnt) { seen |= " + S(1ll << k) + "u; atomicAdd(" + cell + ", t); }\n";
        break;
      case ACC_SUM_F64:
        src += "      const double t = seg_add_f64(" + X + "ok ? __longlong_as_double((long long)" + X + "lo) : 0.0, head);\n      if (tail && cnt) { seen |= " +
               S(1ll << k) + "u; atomicAdd(reinterpret_cast<double*>" + cell + ", t); }\n";
        break;
      case ACC_MIN_I64:
        src += "      const long long t = seg_min_i64(" + X + "ok ? (long long)" + X + "lo : 0x7fffffffffffffffll, head);\n      if (tail && cnt) { seen |= " +
               S(1ll << k) + "u; atomicMin(reinterpret_cast<long long*>" + cell + ", t); }\n";
        break;
      case ACC_MAX_I64:
        src += "      const long long t = seg_max_i64(" + X + "ok ? (long long)" + X + "lo : (-0x7fffffffffffffffll - 1), head);\n      if (tail && cnt) { seen |= " +
               S(1ll << k) + "u; atomicMax(reinterpret_cast<long long*>" + cell + ", t); }\n";
        break;
      default:
        src += "      if (tail && cnt) { seen |= " + S(1ll << k) + "u; atomicAdd(" + cell + ", cnt); }\n";
        break;
    }
    src += "    }\n";
  }
  src += "    if (tail && (seen & ~a.seen[g])) atomicOr(a.seen + g, seen);\n  }\n}\n";
  return src;
}

// first rows -> bitmap over row numbers.  Neighbouring groups of clustered input have neighbouring first rows:
// adjacent lanes targeting the same bitmap word are merged into one atomicOr (device-scope atomics are the cost).
__global__ __launch_bounds__(BLOCK) void k_mark_first_rows(const uint32_t* __restrict__ first_row, int64_t G, unsigned long long* __restrict__ rep_mask) {
  const int64_t g_round = (G + BLOCK - 1) / BLOCK * BLOCK;
  for (int64_t g = (int64_t)blockIdx.x * BLOCK + threadIdx.x; g < g_round; g += (int64_t)gridDim.x * BLOCK) {
    const bool in = g < G;
    const uint32_t fr = in ? first_row[g] : 0u;
    const int64_t w = in ? (int64_t)(fr >> 6) : -1 - (int64_t)lane_id();
    const int64_t wprev = __shfl_up(w, 1, 64), wnext = __shfl_down(w, 1, 64);
    const bool head = lane_id() == 0 || wprev != w;
    const bool tail = lane_id() == 63 || wnext != w;
    const uint64_t acc = wave_seg_or(in ? 1ull << (fr & 63) : 0ull, head);
    if (in && tail) atomicOr(&rep_mask[w], (unsigned long long)acc);
  }
}
struct DenseEmit {
  unsigned long long* dst[2 * MAX_AGGS];  // destination arrays [G], first-seen order
  int src_word[2 * MAX_AGGS];             // cell word copied to dst
  int n_dst;
  uint32_t* seen_dst[MAX_AGGS];
  int seen_bit[MAX_AGGS];
  int n_seen;
};
// key-order group g -> first-seen number; accumulators and seen flags permuted
__global__ __launch_bounds__(BLOCK) void k_dense_permute(const uint32_t* __restrict__ first_row, const unsigned long long* __restrict__ cells,
                                                        const uint32_t* __restrict__ seen, int64_t G, const uint64_t* __restrict__ rep_mask,
                                                        const uint64_t* __restrict__ prefix, uint32_t* __restrict__ new_gid, DenseEmit e) {
  for (int64_t g = (int64_t)blockIdx.x * BLOCK + threadIdx.x; g < G; g += (int64_t)gridDim.x * BLOCK) {
    const uint32_t fr = first_row[g];
    const uint64_t below = rep_mask[fr >> 6] & ((1ull << (fr & 63)) - 1ull);
    const int64_t ng = (int64_t)(prefix[fr >> 6] + __popcll(below));
    new_gid[g] = (uint32_t)ng;
    for (int d = 0; d < e.n_dst; d++) e.dst[d][ng] = cells[(int64_t)e.src_word[d] * G + g];
    const uint32_t sb = seen[g];
    for (int d = 0; d < e.n_seen; d++) e.seen_dst[d][ng] = (sb >> e.seen_bit[d]) & 1u;
  }
}
// set bits of the key bitmap -> key values at their groups' first-seen numbers
__global__ __launch_bounds__(BLOCK) void k_dense_keys(const uint64_t* __restrict__ bits, const uint64_t* __restrict__ prefix, int64_t n_words, int64_t kmin,
                                                     const uint32_t* __restrict__ new_gid, unsigned long long* __restrict__ key_out) {
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * BLOCK) >> 6;
  for (int64_t w = wave; w < n_words; w += n_waves) {
    const uint64_t m = bits[w];
    if ((m >> lane_id()) & 1ull) {
      const int64_t g = (int64_t)(prefix[w] + mbcnt(m));
      key_out[new_gid[g]] = (unsigned long long)(kmin + (w << 6) + (int64_t)lane_id());
    }
  }
}

// The partitioned accumulation's emit (round 6): the first row of every group is marked in `rep_mask` (one bit per INPUT row), so the
// groups in first-seen order are the set bits in row order — group number = popcount prefix.  One thread per 64-row word walks its set
// bits: the row's key comes from the key column where it lies (neighbouring bits share lines: all but a stream of the column), the
// group's totals from the value-major cells (one line per group), and keys, accumulators and seen flags leave at consecutive group
// numbers.  No per-value presence bitmap, no per-group first rows, no renumbering scatter (k_values_to_groups + k_mark_first_rows +
// k_dense_permute + k_dense_keys + k_emit_values: 1.64 ms for 10 M groups of 150 M orders; this: one pass).
// 64-row words per tile: 4096 rows, whose first-row marks become a list of at most that many 16-bit row numbers.  (16384-row tiles take
// the same time.  Of 0.47 ms for 10 M groups of 150 M orders, 0.28 are the groups' cell reads — a random 64-byte line each, 36 G/s, the
// chip's random-line rate — measured by reading the cells in order instead; the stores are 0.06.)
constexpr int GE_TILE_WORDS = 64;
template <typename KT>
__global__ __launch_bounds__(BLOCK) void k_dense_gather_emit(const uint64_t* __restrict__ rep_mask, const uint64_t* __restrict__ rprefix, int64_t row_words,
                                                            const KT* __restrict__ key, long long kmin, const unsigned long long* __restrict__ cells_v, int ncw,
                                                            DenseEmit e, KT* __restrict__ key_out) {
  // (a thread per word that wrote its own groups kept one open line per lane and array — 2 M partly written lines on the chip, evicted
  // before they filled: 0.94 ms.  The tile's marked rows go through a list in LDS instead: the groups of a tile leave in group order,
  // whole lines per wave, four groups per thread in flight)
  __shared__ uint16_t s_list[GE_TILE_WORDS * 64];
  const int64_t n_tiles = (row_words + GE_TILE_WORDS - 1) / GE_TILE_WORDS;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t w0 = tile * GE_TILE_WORDS, w = w0 + threadIdx.x;
    const int64_t w_end = w0 + GE_TILE_WORDS < row_words ? w0 + GE_TILE_WORDS : row_words;
    const int64_t g0 = (int64_t)rprefix[w0], g1 = (int64_t)rprefix[w_end];
    if (g1 == g0) continue;   // (uniform: no group's first row lies in this tile)
    if (threadIdx.x < GE_TILE_WORDS && w < row_words) {
      uint64_t m = rep_mask[w];
      unsigned off = (unsigned)((int64_t)rprefix[w] - g0);
      while (m) {
        s_list[off++] = (uint16_t)((threadIdx.x << 6) | (unsigned)__builtin_ctzll(m));
        m &= m - 1;
      }
    }
    __syncthreads();
    const unsigned cnt = (unsigned)(g1 - g0);
    const int64_t row0 = w0 << 6;
    constexpr int U = 4;
    for (unsigned q0 = threadIdx.x; q0 < cnt; q0 += BLOCK * U) {
      KT k[U];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const unsigned q = q0 + (unsigned)u * BLOCK;
        ok[u] = q < cnt;
        k[u] = key[row0 + s_list[ok[u] ? q : q0]];
      }
      const unsigned long long* c[U];
#pragma unroll
      for (int u = 0; u < U; u++) c[u] = cells_v + (int64_t)((long long)k[u] - kmin) * ncw;
      for (int d = 0; d < e.n_dst; d++) {
        const int sw = e.src_word[d];
        unsigned long long* dst = e.dst[d] + g0;
        unsigned long long a[U];
#pragma unroll
        for (int u = 0; u < U; u++) a[u] = c[u][sw];
#pragma unroll
        for (int u = 0; u < U; u++)
          if (ok[u]) dst[q0 + (unsigned)u * BLOCK] = a[u];
      }
      for (int d = 0; d < e.n_seen; d++) {
        uint32_t* dst = e.seen_dst[d] + g0;
#pragma unroll
        for (int u = 0; u < U; u++)
          if (ok[u]) dst[q0 + (unsigned)u * BLOCK] = 1u;
      }
#pragma unroll
      for (int u = 0; u < U; u++)
        if (ok[u]) key_out[g0 + q0 + (unsigned)u * BLOCK] = k[u];
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------- dense-key node, medium cardinality: partition, then LDS
// One global atomic per (row, aggregate) is what dense_accumulate costs when neighbouring rows carry different keys — ~27 G
// atomics/s on this part, 30 M rows x SUM = 1.36 ms where the bytes are worth 0.07 ms.  When key and arguments are columns as they
// stand, the rows are first moved into <= 64 groups by the RANGE their key falls in (partition.hip: count + one stable scatter at
// copy rate); a group's distinct values then fit one workgroup's LDS, where the rows are accumulated with LDS atomics, and only
// the per-workgroup totals go to the global cells (groups x workgroups-per-partition atomics instead of rows).
constexpr int PART_ACC_MAX = 16;
struct PartAcc {
  int kind;          // AccKind
  int cell;          // first cell word among the node's global cells
  int lcell;         // first 64-bit cell word among this launch's LDS cells
  int val;           // ValKind of the moved argument column (unused for the counts)
  const void* data;  // the moved argument column (partition-major), null for COUNT / COUNT(*)
  int lcell32;       // 32-bit LDS word: a count (a workgroup sees < 2^32 rows), the high word of a 128-bit sum of `narrow` values
  int narrow;        // 128-bit sum of values that fit 64 bits: LDS keeps {64-bit low word, 32-bit high word = negative rows and carries}
};
struct PartAccSet {   // the accumulators of ONE launch: as many as fit LDS beside the window's first rows
  PartAcc a[PART_ACC_MAX];
  int n;
  int ncw;            // 64-bit LDS cell words per value in this launch
  int n32;            // 32-bit LDS words per value
  int track_first;    // this launch also tracks first rows / seen flags (the first one does)
};
__host__ __device__ __forceinline__ bool part_acc_is_count(int kind) { return kind == ACC_COUNT || kind == ACC_COUNT_STAR; }
struct PartBlock {
  int64_t begin, end;  // rows of the window order
  int32_t part;        // window number
  int32_t alone;       // no other workgroup works on this window
};
constexpr int PART_PRELOAD = 3;   // accumulators whose arguments are loaded together with the keys
constexpr int PART_BLOCK = 1024;   // threads per workgroup: the windows' LDS leaves room for one or two workgroups per CU, so they are big
// U rows per thread and iteration, the first NPRE accumulators' arguments loaded with the keys.  (<8, 1> for moved rows with one argument
// column — twice the loads in flight per wave — was measured in round 6 and is not instantiated: 0.94 against 0.96 ms, nothing.)
template <typename KT, int U = 4, int NPRE = PART_PRELOAD>
__global__ __launch_bounds__(PART_BLOCK) void k_dense_accumulate_parts(const PartBlock* __restrict__ blocks, const KT* __restrict__ key, const uint32_t* __restrict__ row_id,
                                                                 PartAccSet accs, long long kmin, int wshift, unsigned long long* __restrict__ cells_v,
                                                                 int64_t vstride, uint64_t vrange, uint32_t* __restrict__ first_row_v,
                                                                 const uint64_t* __restrict__ row_mask, const uint64_t* __restrict__ row_mask_valid, int rows_in_place,
                                                                 const uint32_t* __restrict__ key_map, int key_map_lds, int64_t istride,
                                                                 unsigned long long* __restrict__ rep_mask, int estride) {
  // cell word w of value v lives at cells_v[w * vstride + v * istride]: word-major (vstride = values, istride = 1) or value-major
  // (vstride = 1, istride = words: what the gathering emit reads — one line per group).  rep_mask (every window has ONE workgroup):
  // the first row of every value that has one is marked here, one bit per input row — the first-seen order falls out of its popcounts
  extern __shared__ unsigned long long s_mem[];
  const int W = 1 << wshift;
  unsigned long long* s_cell = s_mem;                                   // [ncw][W]
  uint32_t* s_c32 = reinterpret_cast<uint32_t*>(s_mem + (size_t)accs.ncw * W);   // [n32][W]
  uint32_t* s_first = s_c32 + (size_t)accs.n32 * W;                      // [W]
  // key_map_lds > 0: the first key_map_lds entries of key_map (all that occur) as 16-bit words behind the cells — a random LDS read
  // per row instead of a random L1 / L2 one (0.2 against 1.2 - 2 ms per 600 M rows, profiles/r3_random_access.md)
  uint16_t* s_map = key_map_lds > 0 ? reinterpret_cast<uint16_t*>(s_first + W) : nullptr;
  const PartBlock b = blocks[blockIdx.x];
  for (int x = threadIdx.x; x < key_map_lds; x += PART_BLOCK) s_map[x] = (uint16_t)key_map[x];
  for (int x = threadIdx.x; x < W; x += PART_BLOCK) s_first[x] = 0xFFFFFFFFu;
  for (int k = 0; k < accs.n; k++) {
    const PartAcc& a = accs.a[k];
    const unsigned long long id = acc_identity(a.kind);
    for (int x = threadIdx.x; x < W; x += PART_BLOCK) {
      if (!part_acc_is_count(a.kind)) s_cell[(size_t)a.lcell * W + x] = id;
      if (part_acc_is_count(a.kind) || (a.kind == ACC_SUM_I128 && a.narrow)) s_c32[(size_t)a.lcell32 * W + x] = 0u;
      else if (a.kind == ACC_SUM_I128) s_cell[(size_t)(a.lcell + 1) * W + x] = 0ull;
    }
  }
  __syncthreads();
  const unsigned long long base = (unsigned long long)b.part << wshift;
  // FOUR rows per thread at a time (round 4): their keys, then their group numbers (key_map), then each accumulator's four arguments
  // are in flight together — one row at a time, a thread waited for key -> key_map -> LDS in turn at four waves per SIMD
  for (int64_t i0 = b.begin + threadIdx.x; i0 < b.end; i0 += (int64_t)U * PART_BLOCK) {
    int64_t ii[U];
    uint32_t live = 0;
#pragma unroll
    for (int u = 0; u < U; u++) {
      ii[u] = i0 + (int64_t)u * PART_BLOCK;
      bool ok = ii[u] < b.end;
      // (rows in place — the one-window form: the predicate's mask is looked at here instead of on the move)
      if (ok && row_mask) ok = (row_mask[ii[u] >> 6] >> (ii[u] & 63)) & 1ull;
      if (ok && row_mask_valid) ok = (row_mask_valid[ii[u] >> 6] >> (ii[u] & 63)) & 1ull;
      if (ok) live |= 1u << u;
    }
    if (!live) continue;
    // EVERYTHING the iteration reads from HBM leaves now: the keys, the rows' numbers and the first NPRE accumulators' arguments
    // (round 6: keys -> row numbers -> one accumulator's arguments after the other were three and more memory round trips per iteration
    // at four waves per SIMD — 7.5 us per 4096 rows, 1.08 ms for 150 M orders whose bytes are worth 0.35)
    // (estride > 1: the moved rows are RECORDS of estride 32-bit words — key, arguments and row number side by side, grouped.hip's record
    // form — and `key`, `row_id` and the arguments' `data` point at their word of the first record)
    int64_t ix[U];
#pragma unroll
    for (int u = 0; u < U; u++) ix[u] = ii[u] * estride;
    KT kraw[U];
#pragma unroll
    for (int u = 0; u < U; u++) kraw[u] = (live >> u) & 1u ? key[ix[u]] : KT(0);
    uint32_t rid[U];
    if (row_id) {
#pragma unroll
      for (int u = 0; u < U; u++) rid[u] = (live >> u) & 1u ? row_id[ix[u]] : 0u;
    } else {
#pragma unroll
      for (int u = 0; u < U; u++) rid[u] = (uint32_t)ii[u];
    }
    uint64_t plo[NPRE][U], phi[NPRE][U];
#pragma unroll
    for (int k = 0; k < NPRE; k++)
      if (k < accs.n && accs.a[k].data && !part_acc_is_count(accs.a[k].kind)) load_values_batch<U>(accs.a[k].val, accs.a[k].data, ix, live, plo[k], phi[k], estride);
    int x[U];
    // (key_map: the key column holds table slots, the value is the slot's group number — hash-interned groups in place; its 16-bit
    // copy in LDS when the launch had room for one)
#pragma unroll
    for (int u = 0; u < U; u++) {
      long long kv = (long long)kraw[u];
      if ((live >> u) & 1u) {
        if (s_map) kv = (long long)s_map[(size_t)kraw[u]];
        else if (key_map) kv = (long long)key_map[(size_t)kraw[u]];
      }
      x[u] = (int)((unsigned long long)(kv - kmin) - base);   // value index inside the window
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (!((live >> u) & 1u)) continue;
      if (row_id || rows_in_place) atomicMin(&s_first[x[u]], rid[u]);
      else s_first[x[u]] = 0u;   // (no first rows wanted: a mark that the value has a row — a plain store, every writer's the same)
    }
    auto accumulate_rows = [&](const PartAcc& a, const uint64_t (&lo)[U], const uint64_t (&hi)[U]) {
#pragma unroll
      for (int u = 0; u < U; u++) {
        if (!((live >> u) & 1u)) continue;
        unsigned long long* c = s_cell + (size_t)a.lcell * W + x[u];
        if (a.kind == ACC_SUM_I128 && a.narrow) {   // the high word is 0 or -1: what the LDS keeps of it is 32 bits wide
          const unsigned long long old = atomicAdd(c, (unsigned long long)lo[u]);
          const int32_t h = (int32_t)(uint32_t)hi[u] + ((old + lo[u]) < old ? 1 : 0);
          if (h) atomicAdd(&s_c32[(size_t)a.lcell32 * W + x[u]], (uint32_t)h);
          continue;
        }
        accumulate_cell(a.kind, c, c + W, lo[u], hi[u]);   // (LDS cells)
      }
    };
#pragma unroll
    for (int k = 0; k < NPRE; k++) {
      if (k >= accs.n) continue;
      const PartAcc& a = accs.a[k];
      if (part_acc_is_count(a.kind)) {
#pragma unroll
        for (int u = 0; u < U; u++)
          if ((live >> u) & 1u) atomicAdd(&s_c32[(size_t)a.lcell32 * W + x[u]], 1u);
        continue;
      }
      if (!a.data) {
#pragma unroll
        for (int u = 0; u < U; u++) plo[k][u] = phi[k][u] = 0;
      }
      accumulate_rows(a, plo[k], phi[k]);
    }
    for (int k = NPRE; k < accs.n; k++) {   // further accumulators: one batch of loads each
      const PartAcc& a = accs.a[k];
      if (part_acc_is_count(a.kind)) {
#pragma unroll
        for (int u = 0; u < U; u++)
          if ((live >> u) & 1u) atomicAdd(&s_c32[(size_t)a.lcell32 * W + x[u]], 1u);
        continue;
      }
      uint64_t lo[U], hi[U];
      if (a.data) load_values_batch<U>(a.val, a.data, ix, live, lo, hi, estride);
      else {
#pragma unroll
        for (int u = 0; u < U; u++) lo[u] = hi[u] = 0;
      }
      accumulate_rows(a, lo, hi);
    }
  }
  __syncthreads();
  // the workgroup's totals -> the per-VALUE arrays (value index = key - kmin); which values exist, and their group numbers in
  // first-seen order, are worked out afterwards from the first rows (k_presence_bits, k_values_to_groups)
  for (int x = threadIdx.x; x < W; x += PART_BLOCK) {
    const unsigned long long idx = base + (unsigned)x;
    if (idx >= vrange) continue;
    const uint32_t fr = s_first[x];
    // the workgroup's total of accumulator k for this value as 64-bit words (the 32-bit LDS words widened)
    auto total_lo = [&](const PartAcc& a) -> unsigned long long {
      return part_acc_is_count(a.kind) ? (unsigned long long)s_c32[(size_t)a.lcell32 * W + x] : s_cell[(size_t)a.lcell * W + x];
    };
    auto total_hi = [&](const PartAcc& a) -> unsigned long long {
      return a.narrow ? (unsigned long long)(long long)(int32_t)s_c32[(size_t)a.lcell32 * W + x] : s_cell[(size_t)(a.lcell + 1) * W + x];
    };
    if (b.alone) {   // the window's only workgroup: its values belong to nobody else — plain, coalesced stores
      if (accs.track_first) {
        if (rep_mask) {
          if (fr != 0xFFFFFFFFu) atomicOr(&rep_mask[fr >> 6], 1ull << (fr & 63));
        } else {
          first_row_v[idx] = fr;
        }
      }
      if (fr != 0xFFFFFFFFu)
        for (int k = 0; k < accs.n; k++) {
          const PartAcc& a = accs.a[k];
          cells_v[(int64_t)a.cell * vstride + (int64_t)idx * istride] = total_lo(a);
          if (a.kind == ACC_SUM_I128) cells_v[(int64_t)(a.cell + 1) * vstride + (int64_t)idx * istride] = total_hi(a);
        }
      continue;
    }
    if (fr == 0xFFFFFFFFu) continue;
    if (accs.track_first && fr < first_row_v[idx]) atomicMin(first_row_v + idx, fr);
    for (int k = 0; k < accs.n; k++) {
      const PartAcc& a = accs.a[k];
      const unsigned long long v = total_lo(a);
      unsigned long long* c = cells_v + (int64_t)a.cell * vstride + (int64_t)idx * istride;
      switch (a.kind) {
        case ACC_SUM_I128: {
          const unsigned long long old = atomicAdd(c, v);
          const unsigned long long h = total_hi(a) + ((old + v) < old ? 1ull : 0ull);
          if (h) atomicAdd(c + vstride, h);
          break;
        }
        case ACC_SUM_F64: atomicAdd(reinterpret_cast<double*>(c), __longlong_as_double((long long)v)); break;
        case ACC_MIN_I64: atomicMin(reinterpret_cast<long long*>(c), (long long)v); break;
        case ACC_MAX_I64: atomicMax(reinterpret_cast<long long*>(c), (long long)v); break;
        default: atomicAdd(c, v); break;
      }
    }
  }
}

// which values have a row: one bit per value of the key range, from the first rows the partitioned accumulation left
__global__ __launch_bounds__(BLOCK) void k_presence_bits(const uint32_t* __restrict__ first_row_v, uint64_t vrange, int64_t n_words, uint64_t* __restrict__ bits) {
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * BLOCK) >> 6;
  for (int64_t w = wave; w < n_words; w += n_waves) {
    const uint64_t v = ((uint64_t)w << 6) + lane_id();
    const uint64_t m = ballot64(v < vrange && first_row_v[v] != 0xFFFFFFFFu);
    if (lane_id() == 0) bits[w] = m;
  }
}
// per-value totals -> per-group cells (group = rank of the value among the present ones), first rows and seen flags
__global__ __launch_bounds__(BLOCK) void k_values_to_groups(const uint64_t* __restrict__ bits, const uint64_t* __restrict__ prefix, const uint32_t* __restrict__ first_row_v,
                                                           const unsigned long long* __restrict__ cells_v, int64_t vstride, uint64_t vrange, int ncw, int64_t G,
                                                           uint32_t* __restrict__ first_row, unsigned long long* __restrict__ cells, uint32_t* __restrict__ seen,
                                                           uint32_t seen_mask) {
  for (uint64_t v = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; v < vrange; v += (uint64_t)gridDim.x * BLOCK) {
    const uint64_t m = bits[v >> 6];
    if (!((m >> (v & 63)) & 1ull)) continue;
    const int64_t g = (int64_t)(prefix[v >> 6] + __popcll(m & ((1ull << (v & 63)) - 1ull)));
    first_row[g] = first_row_v[v];
    seen[g] = seen_mask;
    for (int w = 0; w < ncw; w++) cells[(int64_t)w * G + g] = cells_v[(int64_t)w * vstride + (int64_t)v];
  }
}

// first row of every window in the window-ordered keys (-1 stays where a window has no row)
template <typename KT>
__global__ __launch_bounds__(BLOCK) void k_window_begins(const KT* __restrict__ key, int64_t n, long long kmin, int wshift, long long* __restrict__ begins) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    const long long w = (long long)(((unsigned long long)((long long)key[i] - kmin)) >> wshift);
    const long long wp = i ? (long long)(((unsigned long long)((long long)key[i - 1] - kmin)) >> wshift) : -1;
    if (w != wp) begins[w] = i;
  }
}
__global__ __launch_bounds__(BLOCK) void k_row_ids(int64_t n, uint32_t* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) out[i] = (uint32_t)i;
}
// applies when: no predicate, the key and every aggregate argument are columns as they stand without NULLs, the value range splits
// into <= 64 windows that fit LDS, enough rows to pay for the move.  Fills the same cells / first_row / seen as dense_accumulate.
// acc_col[u]: input column of accumulator u's argument (-1 = none: the counts; -2 = an expression), acc_val[u]: its ValKind.
struct PartValues {     // what the partitioned accumulation leaves: totals and first rows per VALUE of the key range
  BufPtr first_row_v;   // u32 [vstride], 0xFFFFFFFF = no row has this value
  BufPtr cells_v;       // u64 [ncw][vstride] — or, value_major, [vstride][ncw]
  int64_t vstride = 0;
  // asked for by the caller (want_rep_mask) and granted when every window had ONE workgroup: the first rows were marked straight into
  // a bitmap over the input rows (no first_row_v), the cells lie value-major
  bool want_rep_mask = false;
  BufPtr rep_mask;      // u64 [(n_in + 63) / 64]
  bool value_major = false;
};
static int part_val_width(int val) {
  switch (val) {
    case VAL_I32: case VAL_U32: case VAL_I32_TO_F64: return 4;
    case VAL_U8: return 1;
    case VAL_I128: return 16;
    default: return 8;
  }
}
// does a range of this many values fit ONE workgroup's LDS beside at least one of these accumulators (partitioned_accumulate then
// moves nothing)?
constexpr size_t PART_LDS_BUDGET = (size_t)128 << 10;   // of the CU's 160 KB: one workgroup per CU at the widest windows
constexpr size_t PART_LDS_MAX = (size_t)160 << 10;      // what a launch may ask for: the windows' cells + (in place, over table slots) the slot -> group words
static int part_window_cap(const std::vector<PartAcc>& all) {
  int min_words = 1;
  for (const PartAcc& a : all)
    if (a.kind == ACC_SUM_I128) min_words = 2;
  int wcap = 0;
  while (((size_t)2 << wcap) * (4 + 8 * (size_t)min_words) <= PART_LDS_BUDGET) wcap++;
  return wcap;
}
static bool partitioned_in_place(uint64_t range, const std::vector<PartAcc>& all) {
  return range > 0 && ((range - 1) >> part_window_cap(all)) == 0;
}
// The core: `key` (type kt, no NULLs) takes values in [kmin, kmin + range); accumulator u reads its argument from all[u].data (a
// source column without NULLs, null for the counts) and owns the cell words all[u].cell (.. + 1 for a 128-bit sum) of `ncw`.
// Moves key, arguments (and row numbers, when first rows are wanted) into window order and leaves totals per value in `out`.
static bool partitioned_accumulate(const void* key, int kt, int64_t n_in, long long kmin, uint64_t range, std::vector<PartAcc> all, int ncw, bool want_first_rows,
                                   PartValues& out, const uint64_t* row_mask = nullptr, const uint64_t* row_mask_valid = nullptr, const uint32_t* key_map = nullptr,
                                   int64_t key_map_n = 0) {
  const bool off = !option_on("agg.partitioned", true);
  const int64_t min_rows = option_int("agg.partitioned_min_rows", 2 * policy().rows_worth_a_pass());
  int64_t n = n_in;
  if (off || n < min_rows || range < 256 || all.empty() || all.size() > (size_t)PART_ACC_MAX) return false;
  if (!(kt == DFGPU_INT32 || kt == DFGPU_DATE32 || kt == DFGPU_INT64 || kt == DFGPU_UINT32 || kt == DFGPU_UINT8)) return false;
  // windows of 2^wshift values whose first rows (4 B) and at least one accumulator (8 B, 16 for a 128-bit sum) fit the LDS budget:
  // at most 64 of them after ONE move of the rows, at most 4096 after two (low 6 bits of the window number first, then the high 6)
  constexpr size_t LDS_BUDGET = PART_LDS_BUDGET;
  const int wcap = part_window_cap(all);
  int wshift = 0;
  while (((range - 1) >> wshift) >= 64) wshift++;
  int levels = 1;
  bool grouped_move = false;
  // the whole range fits ONE workgroup's LDS: nothing is moved.  Every workgroup takes a slice of the rows where they lie (the
  // predicate's mask looked at row by row) and accumulates into its own copy of the one window; the copies merge through atomics
  // — range x workgroups of them, against the 2 x (key + arguments) bytes per row the move costs
  const bool in_place = partitioned_in_place(range, all);
  if (key_map && !in_place) return false;   // (a mapped key cannot be moved by range: the caller maps it first)
  if (in_place) {
    wshift = 6;
    while (((range - 1) >> wshift) >= 1) wshift++;
  } else if (wshift > wcap) {
    wshift = wcap;
    if (((range - 1) >> wshift) >= 4096) return false;
    // up to 2048 windows: ONE move by grouped.hip's pass (round 4: the two 64-way moves cost 4.6 ms for 150 M orders, this one 1.2)
    const bool grouped_off = !option_on("agg.grouped_move", true);   // (A/B switch)
    grouped_move = !grouped_off && ((range - 1) >> wshift) < (uint64_t)GP_MAX_GROUPS && !row_mask_valid && n < 0xFFFFFFFFll;
    if (!grouped_move) {
      levels = 2;
      if (n < 4 * min_rows) return false;   // two moves: only for inputs where the atomics are long
    }
  }
  const int64_t n_windows = (int64_t)((range - 1) >> wshift) + 1;
  const size_t W = (size_t)1 << wshift;
  const size_t bytes_per_value = LDS_BUDGET / W - 4;   // of LDS, beside the first row
  // more than 64 KB of dynamic LDS has to be asked for, per kernel (and per device: not cached) — on EVERY key-type instantiation:
  // which one is launched is only final further down (grouped_move can still fall back to the two-level move of the original key type)
  {
    const bool granted = [] {
      const void* fns[] = {(const void*)k_dense_accumulate_parts<int64_t>, (const void*)k_dense_accumulate_parts<uint32_t>,
                           (const void*)k_dense_accumulate_parts<uint8_t>, (const void*)k_dense_accumulate_parts<int32_t>};
      bool ok = true;
      for (const void* fn : fns)
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PART_LDS_MAX) != hipSuccess) {
          (void)hipGetLastError();
          ok = false;
        }
      return ok;
    }();
    // the windows and the accumulators per value above were sized for LDS_BUDGET: a device that does not grant it (none of the
    // CDNA parts this is built for) takes the global-atomic path instead of launching with more LDS than it has
    if (!granted) return false;
  }
  Runtime& r = rt();
  // what moves: the key, every distinct argument column, the row numbers
  std::vector<const void*> src;
  std::vector<int> widths;
  src.push_back(key);
  widths.push_back(type_width(kt));
  std::vector<int> acc_src(all.size(), -1);
  for (size_t u = 0; u < all.size(); u++) {
    if (!all[u].data || all[u].kind == ACC_COUNT || all[u].kind == ACC_COUNT_STAR) {
      all[u].data = nullptr;
      continue;
    }
    for (size_t q = 1; q < src.size(); q++)
      if (src[q] == all[u].data) acc_src[u] = (int)q;
    if (acc_src[u] < 0) {
      acc_src[u] = (int)src.size();
      src.push_back(all[u].data);
      widths.push_back(part_val_width(all[u].val));
    }
  }
  if (grouped_move) {   // what grouped.hip's pass carries: <= GP_MAX_COLS columns, and its LDS holds a tile of the widest one beside the groups' words
    bool wide = false;
    for (size_t q = 1; q < widths.size(); q++) wide |= widths[q] == 16;
    if ((int)src.size() - 1 + (want_first_rows ? 1 : 0) > GP_MAX_COLS || (wide && n_windows > 1024)) {
      grouped_move = false;
      levels = 2;
      if (n < 4 * min_rows) return false;
    }
  }
  BufPtr ids;
  int ids_at = -1;
  if (want_first_rows && !in_place) {
    ids_at = (int)src.size();
    if (grouped_move) {
      src.push_back(nullptr);   // (grouped.hip carries the rows' numbers without a column of them)
    } else {
      ids = make_buf((size_t)n * 4);
      k_row_ids<<<grid_for(n, BLOCK), BLOCK, 0, r.stream>>>(n, ids->as<uint32_t>());
      src.push_back(ids->ptr);
    }
    widths.push_back(4);
  }
  std::vector<PartBlock> blocks;
  RangePartition rp;
  BufPtr records;       // the grouped move's record form: column q of the moved rows is word q of every rec_dwords-word record
  int rec_dwords = 0;
  std::vector<int> rec_off;   // word of the record where carried column q - 1 starts (the key is word 0)
  auto moved_col = [&](size_t q) -> const void* { return records ? (const void*)(records->as<uint32_t>() + (q == 0 ? 0 : rec_off[q - 1])) : rp.cols[q]->ptr; };
  if (in_place) {
    const int64_t nb = std::min<int64_t>(std::max<int64_t>((n + (1 << 16) - 1) >> 16, 1), 2048);
    for (int64_t b = 0; b < nb; b++) blocks.push_back(PartBlock{n * b / nb, n * (b + 1) / nb, 0, nb == 1 ? 1 : 0});
  } else {
  // (under a predicate only the rows it lets through are moved: `n` is their number from here on)
  std::vector<uint64_t> group_bounds;
  if (grouped_move) {
    // group = window: floor(idx * 2^(64 - wshift) / 2^64) = idx >> wshift.  The keys arrive widened to 64 bits (cols[0]), the
    // arguments and row numbers as carried columns; where every window's rows begin comes with them
    int nbits = 1;
    while (((int64_t)1 << nbits) < n_windows) nbits++;
    const KeyCol kc{key, nullptr, kt, type_width(kt)};
    const GroupSpec gs{(uint64_t)kmin, range, 1ull << (64 - wshift)};
    // (the keys leave as key - kmin in 32 bits when the range allows: half the key bytes of the widened form)
    const bool narrow = range <= (1ull << 32);
    // (records: 32-bit keys and one or two 4-byte carried columns leave side by side — a third of the move's partial-line stores)
    GroupedRows gr = group_rows_by_key(kc, n, gs, nbits, row_mask, /*want_keys=*/true, /*want_dest=*/false, std::vector<const void*>(src.begin() + 1, src.end()),
                                       std::vector<int>(widths.begin() + 1, widths.end()), "agg_group_rows", narrow, /*records=*/narrow);
    rp.rows = gr.rows;
    if (gr.records) {
      records = gr.records;
      rec_dwords = gr.rec_dwords;
      rec_off = gr.rec_off;
    } else {
      rp.cols.push_back(gr.keys);
      for (BufPtr& b : gr.cols) rp.cols.push_back(b);
    }
    group_bounds.resize((size_t)gr.P + 1);
    d2h(group_bounds.data(), gr.bounds->ptr, group_bounds.size() * 8);
    widths[0] = gr.key_width;
    kt = narrow ? DFGPU_UINT32 : DFGPU_INT64;
    if (narrow) kmin = 0;   // (the moved keys are offsets from the range's start)
  } else {
    rp = partition_by_key_range(key, kt, n, kmin, wshift, 63u, (int)std::min<int64_t>(n_windows, 64), src, widths, /*want_bounds=*/false, row_mask, row_mask_valid);
  }
  n = rp.rows;
  if (n == 0) {   // the predicate dropped every row
    out.vstride = ((int64_t)range + 63) / 64 * 64;
    out.first_row_v = make_buf((size_t)out.vstride * 4);
    DFGPU_HIP(hipMemsetAsync(out.first_row_v->ptr, 0xFF, (size_t)out.vstride * 4, r.stream));
    out.cells_v = make_buf((size_t)std::max(1, ncw) * (size_t)out.vstride * 8);
    DFGPU_HIP(hipStreamSynchronize(r.stream));
    return true;
  }
  if (levels == 2) {   // stable second move by the high digit: the rows end up in window order
    std::vector<const void*> src2;
    for (const BufPtr& b : rp.cols) src2.push_back(b->ptr);
    RangePartition rp2 = partition_by_key_range(rp.cols[0]->ptr, kt, n, kmin, wshift + 6, 63u, (int)((n_windows + 63) / 64), src2, widths, /*want_bounds=*/false);
    rp = std::move(rp2);
  }
  for (size_t u = 0; u < all.size(); u++)
    if (acc_src[u] >= 0) all[u].data = moved_col((size_t)acc_src[u]);
  // where every window's rows begin (read off the moved keys), then the workgroups: one per window while its rows are few (its
  // totals then leave as plain stores); else chunks, merged by atomics
  std::vector<long long> begins((size_t)n_windows);
  if (grouped_move) {
    for (int64_t w = 0; w < n_windows; w++) begins[(size_t)w] = group_bounds[(size_t)w + 1] > group_bounds[(size_t)w] ? (long long)group_bounds[(size_t)w] : -1ll;
  } else {
  BufPtr d_begins = make_buf((size_t)n_windows * 8);
  DFGPU_HIP(hipMemsetAsync(d_begins->ptr, 0xFF, (size_t)n_windows * 8, r.stream));
  switch (kt) {
    case DFGPU_INT64: k_window_begins<int64_t><<<grid_for(n, BLOCK), BLOCK, 0, r.stream>>>(rp.cols[0]->as<int64_t>(), n, kmin, wshift, d_begins->as<long long>()); break;
    case DFGPU_UINT32: k_window_begins<uint32_t><<<grid_for(n, BLOCK), BLOCK, 0, r.stream>>>(rp.cols[0]->as<uint32_t>(), n, kmin, wshift, d_begins->as<long long>()); break;
    case DFGPU_UINT8: k_window_begins<uint8_t><<<grid_for(n, BLOCK), BLOCK, 0, r.stream>>>(rp.cols[0]->as<uint8_t>(), n, kmin, wshift, d_begins->as<long long>()); break;
    default: k_window_begins<int32_t><<<grid_for(n, BLOCK), BLOCK, 0, r.stream>>>(rp.cols[0]->as<int32_t>(), n, kmin, wshift, d_begins->as<long long>()); break;
  }
  d2h(begins.data(), d_begins->ptr, (size_t)n_windows * 8);
  }
  {
    int64_t end = n;
    std::vector<PartBlock> rev;
    for (int64_t w = n_windows - 1; w >= 0; w--) {
      const int64_t b0 = begins[(size_t)w];
      if (b0 < 0) continue;
      const int64_t rows = end - b0;
      const int64_t chunk = rows <= ((int64_t)1 << 18) ? rows : std::max<int64_t>((int64_t)1 << 17, (rows + 15) / 16);
      for (int64_t at = b0; at < end; at += chunk) rev.push_back(PartBlock{at, std::min(at + chunk, end), (int32_t)w, chunk >= rows ? 1 : 0});
      end = b0;
    }
    blocks.assign(rev.rbegin(), rev.rend());
  }
  }
  out.vstride = ((int64_t)range + 63) / 64 * 64;
  bool any_chunked = false;
  for (const PartBlock& b : blocks) any_chunked |= !b.alone;
  const bool mark = out.want_rep_mask && want_first_rows && !in_place && !any_chunked;
  if (mark) {
    out.rep_mask = make_zero_buf((size_t)((n_in + 63) / 64 + 1) * 8);
    out.value_major = true;
  } else {
    out.first_row_v = make_buf((size_t)out.vstride * 4);
    DFGPU_HIP(hipMemsetAsync(out.first_row_v->ptr, 0xFF, (size_t)out.vstride * 4, r.stream));   // (windows without a row have no workgroup)
  }
  out.cells_v = make_buf((size_t)std::max(1, ncw) * (size_t)out.vstride * 8);
  if (any_chunked)   // chunks of one window merge through atomics: their cells start from the identities
    for (size_t u = 0; u < all.size(); u++) {
      k_fill_u64<<<grid_for(out.vstride, BLOCK), BLOCK, 0, r.stream>>>(acc_identity(all[u].kind), out.vstride, out.cells_v->as<unsigned long long>() + (int64_t)all[u].cell * out.vstride);
      if (all[u].kind == ACC_SUM_I128)
        k_fill_u64<<<grid_for(out.vstride, BLOCK), BLOCK, 0, r.stream>>>(0ull, out.vstride, out.cells_v->as<unsigned long long>() + (int64_t)(all[u].cell + 1) * out.vstride);
    }
  BufPtr d_blocks = make_buf(blocks.size() * sizeof(PartBlock) + 16);
  DFGPU_HIP(hipMemcpyAsync(d_blocks->ptr, blocks.data(), blocks.size() * sizeof(PartBlock), hipMemcpyHostToDevice, r.stream));
  {
    int64_t bytes = 0;
    for (int w : widths) bytes += n * w;
    ProfileScope psc("agg_dense_accumulate_partitioned", bytes);
    const PartBlock* db = d_blocks->as<PartBlock>();
    const uint32_t* rid = ids_at >= 0 ? (const uint32_t*)moved_col((size_t)ids_at) : nullptr;
    const void* mk = in_place ? key : moved_col(0);
    const int estride = records ? rec_dwords : 1;
    const uint64_t* km = in_place ? row_mask : nullptr;
    const uint64_t* kmv = in_place ? row_mask_valid : nullptr;
    const int ip = in_place && want_first_rows ? 1 : 0;
    const int nb = (int)blocks.size();
    unsigned long long* cv = out.cells_v->as<unsigned long long>();
    uint32_t* fv = mark ? nullptr : out.first_row_v->as<uint32_t>();
    unsigned long long* rm = mark ? out.rep_mask->as<unsigned long long>() : nullptr;
    const int64_t wstride = out.value_major ? 1 : out.vstride, istride = out.value_major ? (int64_t)std::max(1, ncw) : 1;
    size_t u = 0;
    bool first_launch = true;
    while (u < all.size()) {
      PartAccSet ps{};
      ps.track_first = first_launch ? 1 : 0;
      while (u < all.size() && ps.n < PART_ACC_MAX) {
        const bool count = part_acc_is_count(all[u].kind), wide = all[u].kind == ACC_SUM_I128;
        const int need64 = count ? 0 : (wide && !all[u].narrow) ? 2 : 1;
        const int need32 = count || (wide && all[u].narrow) ? 1 : 0;
        if (8 * (size_t)(ps.ncw + need64) + 4 * (size_t)(ps.n32 + need32) > bytes_per_value) break;
        ps.a[ps.n] = all[u];
        ps.a[ps.n].lcell = ps.ncw;
        ps.a[ps.n].lcell32 = ps.n32;
        ps.ncw += need64;
        ps.n32 += need32;
        ps.n++;
        u++;
      }
      DFGPU_CHECK(ps.n > 0, "partitioned aggregation: an accumulator does not fit the window");
      // the accumulators that read an argument column first: they are the ones whose loads leave with the keys (NPRE of the kernel)
      std::stable_partition(ps.a, ps.a + ps.n, [](const PartAcc& a) { return a.data && !part_acc_is_count(a.kind); });
      size_t lds_bytes = W * (8 * (size_t)ps.ncw + 4 * (size_t)ps.n32 + 4);
      // the slot -> group table as 16-bit words in LDS behind the cells when both fit (in place over table slots: every row looks it up)
      int map_lds = 0;
      if (in_place && key_map && key_map_n > 0 && range <= 65536 && lds_bytes + (size_t)key_map_n * 2 + 8 <= PART_LDS_MAX) {
        map_lds = (int)key_map_n;
        lds_bytes += (size_t)key_map_n * 2 + 8;
      }
      switch (kt) {
        case DFGPU_INT64: k_dense_accumulate_parts<int64_t><<<nb, PART_BLOCK, lds_bytes, r.stream>>>(db, (const int64_t*)mk, rid, ps, kmin, wshift, cv, wstride, range, fv, km, kmv, ip, key_map, map_lds, istride, rm, estride); break;
        case DFGPU_UINT32: k_dense_accumulate_parts<uint32_t><<<nb, PART_BLOCK, lds_bytes, r.stream>>>(db, (const uint32_t*)mk, rid, ps, kmin, wshift, cv, wstride, range, fv, km, kmv, ip, key_map, map_lds, istride, rm, estride); break;
        case DFGPU_UINT8: k_dense_accumulate_parts<uint8_t><<<nb, PART_BLOCK, lds_bytes, r.stream>>>(db, (const uint8_t*)mk, rid, ps, kmin, wshift, cv, wstride, range, fv, km, kmv, ip, key_map, map_lds, istride, rm, estride); break;
        default: k_dense_accumulate_parts<int32_t><<<nb, PART_BLOCK, lds_bytes, r.stream>>>(db, (const int32_t*)mk, rid, ps, kmin, wshift, cv, wstride, range, fv, km, kmv, ip, key_map, map_lds, istride, rm, estride); break;
      }
      DFGPU_HIP(hipGetLastError());
      first_launch = false;
    }
  }
  DFGPU_HIP(hipStreamSynchronize(r.stream));   // `blocks` and the moved columns are locals
  return true;
}
// the dense-key node's face of it: applies when there is no predicate and the key and every aggregate argument are columns as they
// stand, without NULLs.  acc_col[u]: input column of accumulator u's argument (-1 = none: the counts; -2 = an expression),
// acc_val[u]: its ValKind.
static bool dense_accumulate_partitioned(const Aggregate& A, const Table& in, const dfgpu_expr* pred, const std::vector<DenseAcc>& accs, const std::vector<int>& acc_col,
                                         const std::vector<int>& acc_val, const std::vector<int>& acc_agg, int ncw, long long kmin, uint64_t range, PartValues& out) {
  // cheap refusals before any argument expression is evaluated
  if (in.nrows < option_int("agg.partitioned_min_rows", 2 * policy().rows_worth_a_pass()) || range < 4096) return false;
  int kc = -1;
  if (!is_plain_column(A.group_nodes[0], A.group_roots[0], &kc) || kc < 0 || kc >= (int)in.cols.size()) return false;
  const Column& key = in.cols[(size_t)kc];
  if (key.validity) return false;
  std::vector<PartAcc> all(accs.size());
  std::vector<Column> evaluated;   // argument expressions evaluated for this call (alive until the rows have been moved)
  for (size_t u = 0; u < accs.size(); u++) {
    const int kind = accs[u].kind;
    all[u] = PartAcc{kind, accs[u].cell, 0, acc_val[u], nullptr, 0, 0};
    if (kind == ACC_COUNT_STAR) continue;
    const int c = acc_col[u];
    auto narrow = [&](const Column& v) { return acc_val[u] != VAL_I128 || (v.field.type == DFGPU_DECIMAL128 && v.field.precision <= 18) ? 1 : 0; };
    if (c >= 0 && c < (int)in.cols.size()) {
      const Column& col = in.cols[(size_t)c];
      if (col.validity || col.dict) return false;
      all[u].data = col.ptr();
      all[u].narrow = narrow(col);
      continue;
    }
    // an expression (Q15's l_extendedprice * (1 - l_discount)): evaluated column-at-a-time first — a streaming pass, where the
    // specialised kernel would pay a global atomic per row for it.  Not under a predicate: the reference evaluates arguments on the
    // rows the FilterExec lets through, and an expression may fail (divide by zero) on the others
    const AggState& a = A.aggs[(size_t)acc_agg[u]];
    if (!a.has_arg || pred) return false;
    int found = -1;
    for (size_t q = 0; q < u; q++)
      if (acc_agg[q] == acc_agg[u] && acc_col[q] == c && all[q].data) found = (int)q;   // (AVG: sum and count share the argument)
    if (found >= 0) {
      all[u].data = all[(size_t)found].data;
      all[u].narrow = all[(size_t)found].narrow;
      continue;
    }
    dfgpu_expr e{a.nodes.data(), (int)a.nodes.size(), a.root};
    Column v = datum_to_column(evaluate(e, in), in.nrows, a.name);
    if (v.validity || v.dict || v.field.type == DFGPU_BOOL || v.field.type == DFGPU_UTF8) return false;
    all[u].data = v.ptr();
    all[u].narrow = narrow(v);
    evaluated.push_back(std::move(v));
  }
  // a FilterExec fused below the aggregate: its mask (false and NULL drop the row) decides which rows are moved at all
  Column mask;
  if (pred) {
    Datum m = evaluate(*pred, in);
    DFGPU_CHECK(m.col.field.type == DFGPU_BOOL, "Cannot create filter with non-boolean predicate");
    mask = datum_to_column(m, in.nrows, "");
  }
  return partitioned_accumulate(key.ptr(), key.field.type, in.nrows, kmin, range, std::move(all), ncw, /*want_first_rows=*/true, out,
                                pred ? (const uint64_t*)mask.ptr() : nullptr, pred ? mask.valid_words() : nullptr);
}

// ---- the same for groups that were interned by hash (agg_update_unfused: evaluated key / argument columns, Final-mode merges of
// partial states): the key that is moved is the row's group number, the per-value totals ARE per-group totals
__global__ __launch_bounds__(BLOCK) void k_row_gids(InternCtx c, const uint32_t* __restrict__ slot_gid, int64_t row_offset, int64_t n, const uint64_t* __restrict__ row_mask,
                                                   const uint32_t* __restrict__ row_slot, uint32_t* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    if (row_mask && !bit_at(row_mask, i)) {   // (rows a predicate dropped were never interned)
      out[i] = 0u;
      continue;
    }
    const uint32_t rs = row_slot ? row_slot[row_offset + i] : 0xFFFFFFFFu;
    out[i] = rs != 0xFFFFFFFFu ? slot_gid[rs] : lookup_gid(c, slot_gid, row_offset + i);
  }
}
struct MergeAcc {
  unsigned long long* acc_lo;
  unsigned long long* acc_hi;
  uint32_t* seen;
  int kind, cell;
};
struct MergeSet {
  MergeAcc a[MAX_AGGS];
  int n;
};
// per-group totals of this batch -> the node's accumulators (which may hold earlier batches); one thread per group, no atomics
__global__ __launch_bounds__(BLOCK) void k_merge_group_totals(const uint32_t* __restrict__ first_row_v, const unsigned long long* __restrict__ cells_v, int64_t vstride, int64_t G,
                                                             MergeSet m) {
  for (int64_t g = (int64_t)blockIdx.x * BLOCK + threadIdx.x; g < G; g += (int64_t)gridDim.x * BLOCK) {
    if (first_row_v[g] == 0xFFFFFFFFu) continue;   // no row of this batch in the group
    for (int k = 0; k < m.n; k++) {
      const MergeAcc& a = m.a[k];
      const unsigned long long v = cells_v[(int64_t)a.cell * vstride + g];
      switch (a.kind) {
        case ACC_SUM_I128: {
          const unsigned long long hi = cells_v[(int64_t)(a.cell + 1) * vstride + g];
          const unsigned long long old = a.acc_lo[g];
          a.acc_lo[g] = old + v;
          a.acc_hi[g] += hi + ((old + v) < old ? 1ull : 0ull);
          break;
        }
        case ACC_SUM_F64: a.acc_lo[g] = (unsigned long long)__double_as_longlong(__longlong_as_double((long long)a.acc_lo[g]) + __longlong_as_double((long long)v)); break;
        case ACC_MIN_I64: if ((long long)v < (long long)a.acc_lo[g]) a.acc_lo[g] = v; break;
        case ACC_MAX_I64: if ((long long)v > (long long)a.acc_lo[g]) a.acc_lo[g] = v; break;
        default: a.acc_lo[g] += v; break;   // SUM_I64 and the counts
      }
      if (a.seen) a.seen[g] = 1u;
    }
  }
}
static bool general_accumulate_partitioned(const InternCtx& ictx, const uint32_t* slot_gid, int64_t G0, int64_t n, const AccSet& accs, int64_t G1,
                                           const uint64_t* row_mask = nullptr, const uint32_t* row_slot = nullptr) {
  if (accs.n <= 0 || accs.n > PART_ACC_MAX || G1 >= 0xFFFFFFFFll) return false;
  std::vector<PartAcc> all((size_t)accs.n);
  MergeSet m{};
  int ncw = 0;
  for (int k = 0; k < accs.n; k++) {
    const AccDesc& d = accs.a[k];
    if (d.valid && d.kind != ACC_COUNT_STAR) return false;   // NULL arguments: the row-at-a-time kernels
    all[(size_t)k] = PartAcc{d.kind, ncw, 0, d.val, d.kind == ACC_COUNT_STAR ? nullptr : d.values, 0, d.narrow};
    m.a[m.n++] = MergeAcc{d.acc_lo, d.acc_hi, d.seen, d.kind, ncw};
    ncw += d.kind == ACC_SUM_I128 ? 2 : 1;
  }
  // cheap refusals first (the row -> group pass below is a random lookup per row)
  const bool off = !option_on("agg.partitioned", true);
  if (off || n < option_int("agg.partitioned_min_rows", 2 * policy().rows_worth_a_pass()) || G1 < 256) return false;
  Runtime& r = rt();
  PartValues pv;
  if (row_slot && (ictx.keyed || ictx.direct) && partitioned_in_place((uint64_t)G1, all)) {
    // few enough groups to accumulate the rows where they lie: the slot the claim pass left for every row (a keyed table leaves one for
    // every live row) stands in for the key, its group number is looked up on the way (the slot -> group table is cache-sized)
    if (!partitioned_accumulate(row_slot + G0, DFGPU_UINT32, n, 0, (uint64_t)G1, std::move(all), ncw, /*want_first_rows=*/false, pv, row_mask, nullptr, slot_gid, ictx.direct ? (int64_t)ictx.direct : (int64_t)ictx.mask + 2)) return false;
  } else {
    BufPtr gids = make_buf((size_t)n * 4);
    {
      ProfileScope ps("agg_row_gids", n * 4);
      k_row_gids<<<grid_for(n, BLOCK), BLOCK, 0, r.stream>>>(ictx, slot_gid, G0, n, row_mask, row_slot, gids->as<uint32_t>());
      DFGPU_HIP(hipGetLastError());
    }
    if (!partitioned_accumulate(gids->ptr, DFGPU_UINT32, n, 0, (uint64_t)G1, std::move(all), ncw, /*want_first_rows=*/false, pv, row_mask, nullptr)) return false;
  }
  k_merge_group_totals<<<grid_for(G1, BLOCK), BLOCK, 0, r.stream>>>(pv.first_row_v->as<uint32_t>(), pv.cells_v->as<unsigned long long>(), pv.vstride, G1, m);
  DFGPU_HIP(hipGetLastError());
  DFGPU_HIP(hipStreamSynchronize(r.stream));
  return true;
}

// the fused node's face of it (hash-interned groups, the keys are interned already): the aggregate arguments are evaluated
// column-at-a-time — plain columns in place; expressions only without a predicate (see dense_accumulate_partitioned) — and the rows
// are moved by group number.  `plans`: plan_for of every aggregate; `row_mask`: the fused FilterExec's mask (null: none).
static bool fused_general_partitioned(Aggregate& A, const Table& in, const std::vector<AccPlan>& plans, const InternCtx& ictx, const uint32_t* slot_gid, int64_t G0,
                                      int64_t G1, const uint64_t* row_mask, const uint32_t* row_slot) {
  const int64_t n = in.nrows;
  const bool off = !option_on("agg.partitioned", true);
  if (trace_on("agg")) fprintf(stderr, "[agg] fused_general_partitioned: n %lld, G0 %lld, G1 %lld, off %d\n", (long long)n, (long long)G0, (long long)G1, (int)off);
  if (off || n < option_int("agg.partitioned_min_rows", 2 * policy().rows_worth_a_pass()) || G1 < 256) return false;   // (fewer groups: the LDS-replicated cells of the fused kernel)
  AccSet accs{};
  std::vector<Column> keep;
  for (size_t k = 0; k < A.aggs.size(); k++) {
    AggState& a = A.aggs[k];
    AccDesc d{};
    d.kind = (a.func == DFGPU_AGG_COUNT && !a.has_arg) ? ACC_COUNT_STAR : plans[k].kind;
    d.val = plans[k].val;
    if (a.has_arg) {
      int c = -1;
      Column v;
      if (is_plain_column(a.nodes, a.root, &c) && c >= 0 && c < (int)in.cols.size()) {
        v = in.cols[(size_t)c];
      } else {
        if (row_mask) return false;
        dfgpu_expr e{a.nodes.data(), (int)a.nodes.size(), a.root};
        v = datum_to_column(evaluate(e, in), n, a.name);
      }
      if (v.validity || v.dict || v.field.type == DFGPU_BOOL || v.field.type == DFGPU_UTF8) return false;
      d.values = v.ptr();
      d.narrow = d.val != VAL_I128 || (v.field.type == DFGPU_DECIMAL128 && v.field.precision <= 18) ? 1 : 0;
      keep.push_back(std::move(v));
    }
    d.acc_lo = a.lo->as<unsigned long long>();
    d.acc_hi = a.hi ? a.hi->as<unsigned long long>() : nullptr;
    d.seen = a.seen->as<uint32_t>();
    if (accs.n >= MAX_AGGS) return false;
    accs.a[accs.n++] = d;
    if (a.func == DFGPU_AGG_AVG) {   // companion count of the non-null arguments (there are no NULLs here: the rows)
      AccDesc c2{};
      c2.kind = ACC_COUNT;
      c2.val = VAL_I64;
      c2.values = d.values;
      c2.acc_lo = a.cnt->as<unsigned long long>();
      if (accs.n >= MAX_AGGS) return false;
      accs.a[accs.n++] = c2;
    }
  }
  return general_accumulate_partitioned(ictx, slot_gid, G0, n, accs, G1, row_mask, row_slot);
}

// The specialised dense-key node.  Returns false (state untouched) when it does not apply.
static bool agg_update_dense_key_jit(Aggregate& A, const Table& in, const dfgpu_expr* pred) {
  Runtime& r = rt();
  const int64_t n = in.nrows;
  if (!option_on("jit", true) || A.group_roots.size() != 1 || A.ngroups != 0) return false;
  if (n < option_int("jit.min_rows", policy().rows_worth_a_pass()) || n >= 0xFFFFFFFFll) return false;
  // ---- compile: predicate, key expression, aggregate arguments
  std::string why;
  RowProgramCompiler comp(in);
  if (pred) comp.set_predicate(*pred);
  dfgpu_expr ke{A.group_nodes[0].data(), (int)A.group_nodes[0].size(), A.group_roots[0]};
  const int key_out = comp.add_output(ke);
  const dfgpu_field kf = comp.output_type(key_out);
  if (!(kf.type == DFGPU_INT32 || kf.type == DFGPU_INT64 || kf.type == DFGPU_DATE32 || kf.type == DFGPU_UINT32 || kf.type == DFGPU_UINT8)) return false;
  std::vector<int> arg_out(A.aggs.size(), -1);
  for (size_t k = 0; k < A.aggs.size(); k++) {
    AggState& a = A.aggs[k];
    if (!a.has_arg) continue;
    dfgpu_expr e{a.nodes.data(), (int)a.nodes.size(), a.root};
    arg_out[k] = comp.add_output(e);
    dfgpu_field t = comp.output_type(arg_out[k]);
    if (a.typed) DFGPU_CHECK(a.in_type.type == t.type, "aggregate argument type changed between batches");
    AccPlan p = plan_for(a.func, t, false);
    if (p.val == VAL_I32_TO_F64 || p.val == VAL_I64_TO_F64) comp.convert_output(arg_out[k], RP_I2F, t);
    else if (p.val == VAL_F64_ORDERED) comp.convert_output(arg_out[k], RP_F64ORD, t);
  }
  CompiledProgram cp;
  if (!comp.finish(cp, why)) return false;
  // ---- accumulator cells
  struct Ent { int agg; bool is_avg_count; int kind; int cell; int seen_bit; };
  std::vector<Ent> entries;
  std::vector<DenseAcc> accs;
  std::vector<int> cell_kind, acc_col, acc_val, acc_agg;   // per accumulator: its argument's input column (-1 none, -2 an expression), ValKind, aggregate
  for (size_t k = 0; k < A.aggs.size(); k++) {
    AggState& a = A.aggs[k];
    dfgpu_field t = a.typed ? a.in_type : (a.has_arg ? cp.out_types[(size_t)arg_out[k]] : fld(DFGPU_INT64));
    AccPlan pl = plan_for(a.func, t, false);
    const int kind = (a.func == DFGPU_AGG_COUNT && !a.has_arg) ? ACC_COUNT_STAR : pl.kind;
    const int val = a.has_arg ? cp.src_out_vals[(size_t)arg_out[k]] : -1;
    int arg_col = -1;
    const int acc_column = !a.has_arg ? -1 : (is_plain_column(a.nodes, a.root, &arg_col) ? arg_col : -2);
    auto unique = [&](int kd) {
      for (size_t u = 0; u < accs.size(); u++)
        if (accs[u].kind == kd && accs[u].val == val) return (int)u;
      accs.push_back({kd, val, (int)cell_kind.size()});
      acc_col.push_back(acc_column);
      acc_val.push_back(pl.val);
      acc_agg.push_back((int)k);
      cell_kind.push_back(kd == ACC_SUM_I128 ? ACC_SUM_I64 : kd);
      if (kd == ACC_SUM_I128) cell_kind.push_back(ACC_SUM_I64);
      return (int)accs.size() - 1;
    };
    int u = unique(kind);
    entries.push_back({(int)k, false, kind, accs[(size_t)u].cell, u});
    if (a.func == DFGPU_AGG_AVG) {
      int uc = unique(ACC_COUNT);
      entries.push_back({(int)k, true, ACC_COUNT, accs[(size_t)uc].cell, uc});
    }
  }
  if (accs.size() > 32 || entries.size() > (size_t)MAX_AGGS) return false;
  const std::string source = agg_dense_node_source(cp, cp.src_out_vals[(size_t)key_out], accs);
  hipFunction_t f_minmax = nullptr, f_setbits = nullptr, f_acc = nullptr;
  try {
    f_minmax = jit_get(source, "dense_minmax");
    f_setbits = jit_get(source, "dense_setbits");
    f_acc = jit_get(source, "dense_accumulate");
  } catch (const Error& e) {
    if (option_on("jit.strict", false)) throw;
    fprintf(stderr, "[dfgpu] node specialisation failed, using the interpreter: %s\n", e.what());
    return false;
  }
  DenseNodeArgs args{};
  for (int c = 0; c < cp.prog.n_cols; c++) {
    args.col[c] = cp.prog.col_data[c];
    args.valid[c] = cp.prog.col_valid[c];
  }
  args.begin = 0;
  args.end = n;
  const int grid = grid_for(n, BLOCK);
  // ---- key range (no state is modified until the node is known to apply)
  BufPtr mm = make_buf(16), flags = make_zero_buf(8);
  const long long init_mm[2] = {INT64_MAX, INT64_MIN};
  h2d_async(mm->ptr, init_mm, 16);
  args.minmax = mm->as<long long>();
  args.flags = flags->as<uint32_t>();
  int64_t key_bytes = n * (kf.type == DFGPU_INT64 ? 8 : kf.type == DFGPU_UINT8 ? 1 : 4);
  long long hmm[2];
  uint32_t hflags[2];
  // the key's range is known without a pass when the key is a column as it stands, without NULLs, under no predicate, and the column
  // is dictionary-encoded (its codes lie in [0, dictionary size)) or carries cached statistics (internal.hpp ColStats)
  bool range_known = false;
  int key_col = -1;
  if (!pred && n > 0 && is_plain_column(A.group_nodes[0], A.group_roots[0], &key_col) && key_col >= 0 && key_col < (int)in.cols.size() && !in.cols[(size_t)key_col].validity) {
    const Column& kc = in.cols[(size_t)key_col];
    if (kc.dict && !kc.dict->values.empty()) {
      hmm[0] = 0;
      hmm[1] = (long long)kc.dict->values.size() - 1;
      range_known = true;
    } else if (auto cached = std::atomic_load(&kc.stats)) {
      if (cached->valid == n) {
        hmm[0] = cached->min;
        hmm[1] = cached->max;
        range_known = true;
      }
    }
  }
  if (range_known) {
    hflags[0] = 0;   // no NULL group
    hflags[1] = 1;   // rows exist
  } else {
    {
      ProfileScope ps("agg_dense_key_range", key_bytes);
      jit_launch(f_minmax, grid, BLOCK, 0, &args, sizeof(args));
    }
    d2h(hmm, mm->ptr, 16);
    d2h(hflags, flags->ptr, 8);
  }
  const bool any_rows = hflags[1] != 0, null_group = hflags[0] != 0;
  uint64_t range = any_rows ? (uint64_t)hmm[1] - (uint64_t)hmm[0] + 1 : 0;
  // dense enough: at most 64 bitmap bits per input row (the join's rank-map gate), and a bounded bitmap
  if (range > (1ull << 36) || range > (uint64_t)n * 64) return false;
  // ---- from here on state is modified
  for (size_t k = 0; k < A.aggs.size(); k++) {
    AggState& a = A.aggs[k];
    if (!a.typed) {
      a.in_type = a.has_arg ? cp.out_types[(size_t)arg_out[k]] : fld(DFGPU_INT64);
      a.typed = true;
    }
  }
  const int64_t n_words = (int64_t)((range + 63) / 64) + 1;
  BufPtr bits = make_zero_buf((size_t)n_words * 8);
  BufPtr prefix = make_buf((size_t)(n_words + 1) * 8);
  args.bits = bits->as<unsigned long long>();
  args.kmin = any_rows ? hmm[0] : 0;
  const int ncw = (int)cell_kind.size();
  // medium cardinalities over plain columns: the rows are moved into LDS-sized key windows and accumulated there FIRST
  // (dense_accumulate_partitioned: totals and first rows per value of the range); which values exist then falls out of the first
  // rows, and the per-value totals are compacted into the groups' cells — no pass that sets bits, no accumulation by global atomics
  PartValues pv;
  pv.want_rep_mask = option_on("agg.gather_emit", true);   // (A/B switch)
  const bool by_parts = any_rows && !null_group && dense_accumulate_partitioned(A, in, pred, accs, acc_col, acc_val, acc_agg, ncw, args.kmin, range, pv);
  // the emit of both forms: where every aggregate's words go (the accumulators exist once the group count is known)
  auto dense_emit_of = [&]() {
    DenseEmit e{};
    for (const Ent& en : entries) {
      AggState& a = A.aggs[(size_t)en.agg];
      if (en.is_avg_count) {
        e.dst[e.n_dst] = a.cnt->as<unsigned long long>();
        e.src_word[e.n_dst++] = en.cell;
      } else {
        e.dst[e.n_dst] = a.lo->as<unsigned long long>();
        e.src_word[e.n_dst++] = en.cell;
        if (en.kind == ACC_SUM_I128) {
          e.dst[e.n_dst] = a.hi->as<unsigned long long>();
          e.src_word[e.n_dst++] = en.cell + 1;
        }
        e.seen_dst[e.n_seen] = a.seen->as<uint32_t>();
        e.seen_bit[e.n_seen++] = en.seen_bit;
      }
    }
    return e;
  };
  if (by_parts && pv.rep_mask) {
    // every window had one workgroup: the first rows are marked over the input rows, the totals lie value-major — the groups in
    // first-seen order (group_values/mod.rs:88-92) are gathered in one pass (k_dense_gather_emit)
    const int64_t row_words = (n + 63) / 64;
    BufPtr rprefix = make_buf((size_t)(row_words + 1) * 8);
    scan_mask_popcounts(pv.rep_mask->as<uint64_t>(), nullptr, n, rprefix->as<uint64_t>());
    const int64_t G = (int64_t)read_u64(rprefix->as<uint64_t>() + row_words);
    grow_accumulators(A, 0, G, /*init=*/false);   // every cell of every group is written by the gather
    const DenseEmit e = dense_emit_of();
    int kci = -1;   // (dense_accumulate_partitioned: the key is a column as it stands, without NULLs)
    DFGPU_CHECK(is_plain_column(A.group_nodes[0], A.group_roots[0], &kci) && kci >= 0 && kci < (int)in.cols.size(), "internal: the partitioned node's key is not a column");
    const Column& kin = in.cols[(size_t)kci];
    Column kc = alloc_column(kf, A.group_names[0], G);
    kc.dict = kin.dict;
    if (G) {
      ProfileScope ps("agg_dense_gather_emit", n * type_width(kf.type) + G * (type_width(kf.type) + 16 * (int64_t)e.n_dst + 4 * (int64_t)e.n_seen));
      const int g = (int)std::min<int64_t>((row_words + GE_TILE_WORDS - 1) / GE_TILE_WORDS, 2048);
      const uint64_t* rm = pv.rep_mask->as<uint64_t>();
      const uint64_t* rp = rprefix->as<uint64_t>();
      const unsigned long long* cv = pv.cells_v->as<unsigned long long>();
      switch (type_width(kf.type)) {
        case 8: k_dense_gather_emit<int64_t><<<g, BLOCK, 0, r.stream>>>(rm, rp, row_words, (const int64_t*)kin.ptr(), args.kmin, cv, std::max(1, ncw), e, (int64_t*)kc.data->ptr); break;
        case 1: k_dense_gather_emit<uint8_t><<<g, BLOCK, 0, r.stream>>>(rm, rp, row_words, (const uint8_t*)kin.ptr(), args.kmin, cv, std::max(1, ncw), e, (uint8_t*)kc.data->ptr); break;
        default:
          if (kf.type == DFGPU_UINT32) k_dense_gather_emit<uint32_t><<<g, BLOCK, 0, r.stream>>>(rm, rp, row_words, (const uint32_t*)kin.ptr(), args.kmin, cv, std::max(1, ncw), e, (uint32_t*)kc.data->ptr);
          else k_dense_gather_emit<int32_t><<<g, BLOCK, 0, r.stream>>>(rm, rp, row_words, (const int32_t*)kin.ptr(), args.kmin, cv, std::max(1, ncw), e, (int32_t*)kc.data->ptr);
          break;
      }
      DFGPU_HIP(hipGetLastError());
    }
    Table gk;
    gk.nrows = G;
    gk.cols.push_back(std::move(kc));
    A.group_keys = std::move(gk);
    A.ngroups = G;
    DFGPU_HIP(hipStreamSynchronize(r.stream));   // (the moved cells and the marks are locals)
    return true;
  }
  if (by_parts) {
    k_presence_bits<<<grid_for(n_words, BLOCK / WAVE), BLOCK, 0, r.stream>>>(pv.first_row_v->as<uint32_t>(), range, n_words, bits->as<uint64_t>());
    DFGPU_HIP(hipGetLastError());
  } else if (any_rows) {
    ProfileScope ps("agg_dense_setbits", key_bytes);
    jit_launch(f_setbits, grid, BLOCK, 0, &args, sizeof(args));
  }
  scan_mask_popcounts(bits->as<uint64_t>(), nullptr, n_words * 64, prefix->as<uint64_t>());
  const int64_t Gk = (int64_t)read_u64(prefix->as<uint64_t>() + n_words);
  const int64_t G = Gk + (null_group ? 1 : 0);
  BufPtr first_row = make_buf((size_t)std::max<int64_t>(G, 1) * 4);
  BufPtr seen = by_parts ? make_buf((size_t)std::max<int64_t>(G, 1) * 4) : make_zero_buf((size_t)std::max<int64_t>(G, 1) * 4);
  BufPtr cells = make_buf((size_t)std::max(1, ncw) * std::max<int64_t>(G, 1) * 8);
  args.prefix = prefix->as<uint64_t>();
  args.first_row = first_row->as<uint32_t>();
  args.cells = cells->as<unsigned long long>();
  args.seen = seen->as<uint32_t>();
  args.G = G;
  args.null_group = null_group ? Gk : -1;
  if (by_parts) {
    if (G) {
      ProfileScope ps("agg_dense_values_to_groups", (int64_t)range * (4 + 8 * ncw) + G * (8 + 8 * ncw));
      k_values_to_groups<<<grid_for((int64_t)range, BLOCK), BLOCK, 0, r.stream>>>(bits->as<uint64_t>(), prefix->as<uint64_t>(), pv.first_row_v->as<uint32_t>(),
                                                                                 pv.cells_v->as<unsigned long long>(), pv.vstride, range, ncw, G, first_row->as<uint32_t>(),
                                                                                 cells->as<unsigned long long>(), seen->as<uint32_t>(), (uint32_t)((1ull << accs.size()) - 1ull));
      DFGPU_HIP(hipGetLastError());
    }
  } else {
    DFGPU_HIP(hipMemsetAsync(first_row->ptr, 0xFF, (size_t)std::max<int64_t>(G, 1) * 4, r.stream));
    for (int w = 0; w < ncw && G; w++)
      k_fill_u64<<<grid_for(G, BLOCK), BLOCK, 0, r.stream>>>(acc_identity(cell_kind[(size_t)w]), G, cells->as<unsigned long long>() + (int64_t)w * G);
    if (G) {
      ProfileScope ps("agg_dense_accumulate", n * cp.input_bytes_per_row);
      jit_launch(f_acc, grid, BLOCK, 0, &args, sizeof(args));
    }
  }
  // ---- first-seen order (group_values/mod.rs:88-92), keys rebuilt from the bit positions
  const int64_t row_words = (n + 63) / 64;
  BufPtr rep_mask = make_zero_buf((size_t)row_words * 8);
  BufPtr rprefix = make_buf((size_t)(row_words + 1) * 8);
  if (G) k_mark_first_rows<<<grid_for(G, BLOCK), BLOCK, 0, r.stream>>>(first_row->as<uint32_t>(), G, rep_mask->as<unsigned long long>());
  scan_mask_popcounts(rep_mask->as<uint64_t>(), nullptr, n, rprefix->as<uint64_t>());
  DFGPU_CHECK((int64_t)read_u64(rprefix->as<uint64_t>() + row_words) == G, "dense-key node: first-row marks do not match the group count");
  grow_accumulators(A, 0, G, /*init=*/false);  // new_gid is a permutation of 0..G-1: k_dense_permute writes every cell
  const DenseEmit e = dense_emit_of();
  BufPtr new_gid = make_buf((size_t)std::max<int64_t>(G, 1) * 4);
  BufPtr key_i64 = make_zero_buf((size_t)std::max<int64_t>(G, 1) * 8);
  if (G) {
    ProfileScope ps("agg_dense_emit", G * (4 + 8 * (int64_t)e.n_dst));
    k_dense_permute<<<grid_for(G, BLOCK), BLOCK, 0, r.stream>>>(first_row->as<uint32_t>(), cells->as<unsigned long long>(), seen->as<uint32_t>(), G,
                                                                rep_mask->as<uint64_t>(), rprefix->as<uint64_t>(), new_gid->as<uint32_t>(), e);
    k_dense_keys<<<grid_for(n_words, BLOCK / WAVE), BLOCK, 0, r.stream>>>(bits->as<uint64_t>(), prefix->as<uint64_t>(), n_words, args.kmin, new_gid->as<uint32_t>(),
                                                                          key_i64->as<unsigned long long>());
    DFGPU_HIP(hipGetLastError());
  }
  Column kc = alloc_column(kf, A.group_names[0], G);
  {
    int key_col = -1;
    if (is_plain_column(A.group_nodes[0], A.group_roots[0], &key_col) && key_col >= 0 && key_col < (int)in.cols.size()) kc.dict = in.cols[key_col].dict;
  }
  if (G) {
    int mode = 0;
    if (kf.type == DFGPU_INT32 || kf.type == DFGPU_DATE32 || kf.type == DFGPU_UINT32) mode = 3;
    else if (kf.type == DFGPU_UINT8) mode = 4;
    k_emit_values<<<grid_for(G, BLOCK), BLOCK, 0, r.stream>>>(mode, key_i64->as<unsigned long long>(), nullptr, nullptr, G, kc.data->ptr, nullptr);
    DFGPU_HIP(hipGetLastError());
    if (null_group) {
      // the NULL group's first-seen number
      uint32_t ng = 0;
      d2h(&ng, new_gid->as<uint32_t>() + Gk, 4);
      std::vector<uint8_t> vb((size_t)G, 1);
      vb[ng] = 0;
      BufPtr dvb = make_buf((size_t)G + 64);
      h2d_async(dvb->ptr, vb.data(), (size_t)G);
      kc.validity = make_buf(bitmap_bytes(G));
      pack_bytes_to_bitmap(dvb->as<uint8_t>(), G, kc.validity->as<uint64_t>());
      kc.null_count = 1;
      DFGPU_HIP(hipStreamSynchronize(r.stream));
    }
  }
  Table gk;
  gk.nrows = G;
  gk.cols.push_back(std::move(kc));
  A.group_keys = std::move(gk);
  A.ngroups = G;
  DFGPU_HIP(hipStreamSynchronize(r.stream));
  return true;
}

// ---------------------------------------------------------------- specialised node: group key arrives in order
// GROUP BY <integer column> whose values are non-decreasing in row order (TPC-H: lineitem by l_orderkey, orders by
// o_orderkey — the clustering dbgen produces and a sorted scan declares): equal keys are adjacent, so groups are RUNS of
// rows and the group number of a row is the number of run heads before it — the reference's fully-ordered aggregation
// (aggregates/order/full.rs GroupOrderingFull: the current group is the last one, earlier groups are complete), without
// any table: no bitmap over the key range, no hash, no renumbering (run order IS first-seen order).
//   heads      : head[i] = key[i] != key[i-1]                 (one pass over the key column; wave ballot = one word)
//   scan       : exclusive popcount prefix per 64-row word     (scan.hip)
//   accumulate : one wave per word; a segmented wave scan gives every run's total at its last lane.  A run that starts in
//                this word and ends before the end of the NEXT word is finished by this wave (it evaluates the run's first
//                rows of the next word itself) and leaves as plain stores — no atomics, no initialised accumulators; only
//                pieces of longer runs (> 64 rows) fall back to atomics on identity-filled cells.
// Whether the column is ordered comes from its cached statistics (column_stats: min / max / ascending / non-decreasing,
// one pass, shared with the join's map gating) — the scan-statistics twin of the reference's `output_ordering`.
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_run_heads(const T* __restrict__ key, int64_t n, uint64_t* __restrict__ heads) {
  const int64_t n_words = (n + 63) >> 6;
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * BLOCK) >> 6;
  const unsigned lane = lane_id();
  constexpr int U = 4;
  for (int64_t w0 = wave * U; w0 < n_words; w0 += n_waves * U) {
    T k[U], kp[U];
#pragma unroll
    for (int j = 0; j < U; j++) {  // a wave's U words are consecutive: lane 0 of word j takes its predecessor from lane 63 of word j - 1
      const int64_t i = ((w0 + j) << 6) + lane;
      const int64_t ic = i < n ? i : n - 1;
      k[j] = key[ic];
    }
    const int64_t first = w0 << 6;
    const T before = key[first > 0 ? (first < n ? first : n) - 1 : 0];  // one extra element per wave iteration (same line as k[0])
#pragma unroll
    for (int j = 0; j < U; j++) {
      const T up = (T)__shfl_up((unsigned long long)k[j], 1, 64);
      const T carry = j == 0 ? before : (T)__shfl((unsigned long long)k[j > 0 ? j - 1 : 0], 63, 64);
      kp[j] = lane == 0 ? carry : up;
    }
#pragma unroll
    for (int j = 0; j < U; j++) {
      const int64_t i = ((w0 + j) << 6) + lane;
      const uint64_t h = ballot64(i < n && (i == 0 || k[j] != kp[j]));
      if (lane == 0 && w0 + j < n_words) heads[w0 + j] = h;
    }
  }
}
// Further group keys of an ordered-input aggregation: flag[0] |= 1 when the column changes INSIDE a run of the first key —
// then the first key does not determine it and the runs are not the groups.  (GROUP BY l_orderkey, o_orderdate, o_shippriority
// over a join's output in probe order: the order key determines the other two, one run = one group, no hash table.)
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_run_dependent(const T* __restrict__ col, const uint64_t* __restrict__ heads, int64_t n, uint32_t* __restrict__ flag) {
  bool bad = false;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    if (i == 0) continue;
    const T a = col[i - 1], b = col[i];
    bool differ;
    if constexpr (sizeof(T) == 16) differ = a.x != b.x || a.y != b.y || a.z != b.z || a.w != b.w;
    else differ = a != b;
    bad |= differ && !((heads[i >> 6] >> (i & 63)) & 1ull);
  }
  if (__any(bad) && lane_id() == 0) atomicOr(flag, 1u);
}
// flag[0] |= 1 when some run spans more than "the rest of its first word + the leading rows of the next word"
__global__ __launch_bounds__(BLOCK) void k_run_long_flag(const uint64_t* __restrict__ heads, int64_t n_words, uint32_t* __restrict__ flag) {
  bool any = false;
  for (int64_t w = (int64_t)blockIdx.x * BLOCK + threadIdx.x + 1; w < n_words; w += (int64_t)gridDim.x * BLOCK) {
    const uint64_t hw = heads[w];
    if (hw & 1ull) continue;                                  // a run starts at the word's first row: nothing is carried in
    const bool short_run = heads[w - 1] != 0ull && (hw != 0ull || w == n_words - 1);
    any |= !short_run;
  }
  if (ballot64(any) != 0 && lane_id() == 0) atomicOr(flag, 1u);
}

constexpr int RUNS_MAX_ACCS = 16;
struct RunsNodeArgs {
  const void* col[RP_MAX_COLS];
  const uint64_t* valid[RP_MAX_COLS];
  const uint64_t* heads;
  const uint64_t* prefix;
  unsigned long long* cell[2 * RUNS_MAX_ACCS];  // per accumulator: [2k] = value / low word, [2k + 1] = high word of an i128 sum
  uint32_t* seen[RUNS_MAX_ACCS];                // per accumulator, null = none kept (the count of an AVG)
  void* key_out;
  long long n;
};
struct RunsAcc {
  int kind;     // AccKind
  int val;      // value id in the generated source, -1 = none (COUNT(*))
  bool narrow;  // ACC_SUM_I128 over values of at most 16 decimal digits: the sum of 128 rows fits 64 bits
  bool inter;   // ACC_SUM_I128 cells interleaved {lo, hi}: cell = the Decimal128 column, hi cell pointer = lo cell pointer + 1
};

static std::string agg_runs_node_source(const CompiledProgram& cp, int key_val, int key_type, const std::vector<RunsAcc>& accs) {
  auto S = [](long long v) { return std::to_string(v); };
  const int K = (int)accs.size();
  std::string src = R"SRC(
typedef __int128 i128;
typedef unsigned __int128 u128;
typedef unsigned long long U64;
typedef long long I64;
typedef unsigned int U32;
typedef int I32;
typedef unsigned char U8;
#define BLOCK 256
struct Args {
  const void* col[10];
  const U64* valid[10];
  const U64* heads;
  const U64* prefix;
  U64* cell[32];
  U32* seen[16];
  void* key_out;
  long long n;
};
__device__ __forceinline__ double v2f(i128 x) { return __longlong_as_double((long long)(U64)x); }
__device__ __forceinline__ i128 f2v(double d) { return (i128)(u128)(U64)__double_as_longlong(d); }
__device__ __forceinline__ long long f64ord(U64 bits) { long long b = (long long)bits; return b ^ (long long)((U64)(b >> 63) >> 1); }
__device__ __forceinline__ I32 date32_part(I32 days, int part) {  // device.hpp date32_part
  const I64 z = (I64)days + 719468, era = (z >= 0 ? z : z - 146096) / 146097, doe = z - era * 146097;
  const I64 yoe = (doe - doe / 1460 + doe / 36524 - doe / 146096) / 365, doy = doe - (365 * yoe + yoe / 4 - yoe / 100), mp = (5 * doy + 2) / 153;
  const I64 d = doy - (153 * mp + 2) / 5 + 1, m = mp < 10 ? mp + 3 : mp - 9, y = yoe + era * 400 + (m <= 2 ? 1 : 0);
  return (I32)(part == 0 ? y : part == 1 ? m : d);
}
__device__ __forceinline__ bool kt(i128 v, bool n) { return !n && ((int)v & 1); }
__device__ __forceinline__ bool kf(i128 v, bool n) { return !n && !((int)v & 1); }
#define SEG_SCAN(STEP)                                        \
  bool f = head;                                              \
  const int lane = threadIdx.x & 63;                          \
  _Pragma("unroll") for (int d = 1; d < 64; d <<= 1) {        \
    const int fo = __shfl_up((int)f, d, 64);                  \
    STEP                                                      \
    if (lane >= d && !f) f = fo != 0;                         \
  }
__device__ __forceinline__ U64 seg_add_u64(U64 v, bool head) {
  SEG_SCAN(const U64 o = __shfl_up(v, d, 64); if (lane >= d && !f) v += o;)
  return v;
}
__device__ __forceinline__ u128 seg_add_u128(u128 v, bool head) {
  SEG_SCAN(const U64 ol = __shfl_up((U64)v, d, 64); const U64 oh = __shfl_up((U64)(v >> 64), d, 64); if (lane >= d && !f) v += ((u128)oh << 64) | ol;)
  return v;
}
__device__ __forceinline__ double seg_add_f64(double v, bool head) {
  SEG_SCAN(const double o = __shfl_up(v, d, 64); if (lane >= d && !f) v += o;)
  return v;
}
__device__ __forceinline__ long long seg_min_i64(long long v, bool head) {
  SEG_SCAN(const long long o = __shfl_up(v, d, 64); if (lane >= d && !f) v = o < v ? o : v;)
  return v;
}
__device__ __forceinline__ long long seg_max_i64(long long v, bool head) {
  SEG_SCAN(const long long o = __shfl_up(v, d, 64); if (lane >= d && !f) v = o > v ? o : v;)
  return v;
}
// Unsegmented inclusive wave prefix sums in 6 DPP steps (VALU only — the segmented scans above cost 3-5 LDS-crossbar
// permutes per step): a run's total is P[last lane] - P[lane before its first lane], exact in wrapping integer arithmetic.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ U64 dpp_u64(U64 v) {
  const U32 lo = (U32)__builtin_amdgcn_update_dpp(0, (int)(U32)v, CTRL, ROW_MASK, 0xF, true);
  const U32 hi = (U32)__builtin_amdgcn_update_dpp(0, (int)(U32)(v >> 32), CTRL, ROW_MASK, 0xF, true);
  return ((U64)hi << 32) | lo;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ u128 dpp_u128(u128 v) { return ((u128)dpp_u64<CTRL, ROW_MASK>((U64)(v >> 64)) << 64) | dpp_u64<CTRL, ROW_MASK>((U64)v); }
__device__ __forceinline__ U64 scan_u64(U64 v) {
  v += dpp_u64<0x111, 0xF>(v); v += dpp_u64<0x112, 0xF>(v); v += dpp_u64<0x114, 0xF>(v); v += dpp_u64<0x118, 0xF>(v);
  v += dpp_u64<0x142, 0xA>(v); v += dpp_u64<0x143, 0xC>(v);
  return v;
}
__device__ __forceinline__ u128 scan_u128(u128 v) {
  v += dpp_u128<0x111, 0xF>(v); v += dpp_u128<0x112, 0xF>(v); v += dpp_u128<0x114, 0xF>(v); v += dpp_u128<0x118, 0xF>(v);
  v += dpp_u128<0x142, 0xA>(v); v += dpp_u128<0x143, 0xC>(v);
  return v;
}
// value of lane `src` (all lanes execute)
__device__ __forceinline__ U64 lane_u64(U64 v, int src) { return __shfl(v, src, 64); }
__device__ __forceinline__ u128 lane_u128(u128 v, int src) { return ((u128)__shfl((U64)(v >> 64), src, 64) << 64) | __shfl((U64)v, src, 64); }
// whole-wave reductions (the rows of the next word that finish this word's last run)
__device__ __forceinline__ double all_add_f64(double v) {
  _Pragma("unroll") for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}
__device__ __forceinline__ long long all_min_i64(long long v) {
  _Pragma("unroll") for (int d = 32; d >= 1; d >>= 1) { const long long o = __shfl_xor(v, d, 64); v = o < v ? o : v; }
  return v;
}
__device__ __forceinline__ long long all_max_i64(long long v) {
  _Pragma("unroll") for (int d = 32; d >= 1; d >>= 1) { const long long o = __shfl_xor(v, d, 64); v = o > v ? o : v; }
  return v;
}
)SRC";
  const int KK = std::max(K, 1);
  src += "#define NACC " + S(KK) + "\n";
  src += "__device__ __forceinline__ void eval_row(const Args& a, const long long i, long long& key, U64 (&Xlo)[NACC], U64 (&Xhi)[NACC], bool (&Xok)[NACC]) {\n";
  src += cp.src_loads;
  src += cp.src_pred;
  src += cp.src_outs;
  src += "    key = (long long)(U64)V" + S(key_val) + ";\n";
  for (int k = 0; k < K; k++) {
    const RunsAcc& c = accs[(size_t)k];
    if (c.val >= 0)
      src += "    Xok[" + S(k) + "] = !N" + S(c.val) + "; Xlo[" + S(k) + "] = (U64)V" + S(c.val) + "; Xhi[" + S(k) + "] = (U64)((u128)V" + S(c.val) + " >> 64);\n";
    else
      src += "    Xok[" + S(k) + "] = true;\n";
  }
  src += "}\n";
  std::string key_store;
  switch (key_type) {
    case DFGPU_INT64: key_store = "((long long*)a.key_out)[g] = key;"; break;
    case DFGPU_UINT8: key_store = "((U8*)a.key_out)[g] = (U8)key;"; break;
    default: key_store = "((U32*)a.key_out)[g] = (U32)key;"; break;  // INT32 / DATE32 / UINT32
  }
  src += R"SRC(
extern "C" __global__ __launch_bounds__(BLOCK) void runs_accumulate(Args a) {
  const int lane_ = threadIdx.x & 63;
  const long long n_words = (a.n + 63) >> 6;
  const long long n_waves = ((long long)gridDim.x * BLOCK) >> 6;
  for (long long w = ((long long)blockIdx.x * BLOCK + threadIdx.x) >> 6; w < n_words; w += n_waves) {
    const U64 hw = a.heads[w];
    const bool has_next = w + 1 < n_words;
    const U64 hprev = w > 0 ? a.heads[w - 1] : 1ull;
    const U64 hnext = has_next ? a.heads[w + 1] : 1ull;
    const int lead = hw ? __builtin_ctzll(hw) : 64;                  // leading rows that continue a run of an earlier word
    const bool lead_short = hprev != 0ull && (hw != 0ull || !has_next);  // ... which the previous word's wave finishes itself
    int ext = 0;                                                      // rows of the next word that finish this word's last run
    if (has_next && !(hnext & 1ull) && hw != 0ull && (hnext != 0ull || w + 2 == n_words)) {
      const long long rem = a.n - ((w + 1) << 6);
      ext = hnext ? __builtin_ctzll(hnext) : (int)(rem < 64 ? rem : 64);
    }
    const long long i = (w << 6) + lane_;
    const bool active = i < a.n && (lane_ >= lead || !lead_short);
    U64 Xlo[NACC], Xhi[NACC], Elo[NACC], Ehi[NACC];
    bool Xok[NACC], Eok[NACC];
#pragma unroll
    for (int k = 0; k < NACC; k++) { Xlo[k] = Xhi[k] = Elo[k] = Ehi[k] = 0ull; Xok[k] = Eok[k] = false; }
    long long key = 0, key2 = 0;
    if (active) eval_row(a, i, key, Xlo, Xhi, Xok);
    if (lane_ < ext) eval_row(a, ((w + 1) << 6) + lane_, key2, Elo, Ehi, Eok);
    const long long g = (long long)a.prefix[w] + __popcll(hw & ((2ull << lane_) - 1ull)) - 1;
    const bool is_head = (hw >> lane_) & 1ull;
    const bool head = lane_ == 0 || is_head;
    const bool tail = lane_ == 63 || ((hw >> (lane_ + 1)) & 1ull);
    const bool cont = lane_ < lead;
    const bool ends = lane_ < 63 || !has_next || (hnext & 1ull) || ext > 0;
    const bool plain = tail && !cont && ends;                    // the whole run is in this wave's hands: plain stores
    const bool atom = tail && !plain && !(cont && lead_short);   // a piece of a long run: atomics on identity-filled cells
    if (is_head) { )SRC" + key_store + R"SRC( }
    // the segment (piece of a run) this lane belongs to starts at lane h; rows and non-NULL values are counted on ballots
    const int h = 63 - __builtin_clzll((hw | 1ull) & ((2ull << lane_) - 1ull));
    const U64 seg_mask = ((2ull << lane_) - 1ull) & ~((1ull << h) - 1ull);
    const U64 ext_mask = ext >= 64 ? ~0ull : ((1ull << ext) - 1ull);
    const int hp = (h + 63) & 63;  // the lane before the segment (its prefix is subtracted; none for h == 0)
    const U64 run_rows = (U64)__popcll(__ballot(active) & seg_mask) + (lane_ == 63 ? (U64)ext : 0ull);
)SRC";
  for (int k = 0; k < K; k++) {
    const RunsAcc& c = accs[(size_t)k];
    const std::string ks = "[" + S(k) + "]";
    const bool maybe_null = c.val >= 0 && cp.src_maybe_null[(size_t)c.val];
    const std::string cell = "a.cell[" + S(2 * k) + "]", cellhi = "a.cell[" + S(2 * k + 1) + "]", seen = "a.seen[" + S(k) + "]";
    src += "    {\n";
    if (maybe_null)  // the ballots are taken by the whole wave, outside any lane-dependent expression
      src += "      const U64 okm = __ballot(Xok" + ks + "), eokm = __ballot(Eok" + ks + ");\n"
             "      const U64 cnt = (U64)__popcll(okm & seg_mask) + (lane_ == 63 ? (U64)__popcll(eokm & ext_mask) : 0ull);\n";
    else src += "      const U64 cnt = run_rows;\n";
    const std::string seen_plain = "if (" + seen + ") " + seen + "[g] = cnt ? 1u : 0u;";
    const std::string seen_atom = "if (" + seen + " && !" + seen + "[g]) atomicOr(" + seen + " + g, 1u);";
    const std::string gi = c.inter ? "(g << 1)" : "g";  // interleaved cells: group g at words 2g (lo) and 2g + 1 (hi)
    const std::string add128 = "const U64 lo = (U64)t, hi = (U64)(t >> 64); const U64 old = atomicAdd(" + cell + " + " + gi + ", lo); atomicAdd(" + cellhi +
                               " + " + gi + ", hi + ((old + lo) < old ? 1ull : 0ull));";
    switch (c.kind) {
      case ACC_SUM_I128:
        if (c.narrow) {
          src += "      const U64 P = scan_u64(Xok" + ks + " ? Xlo" + ks + " : 0ull);\n"
                 "      const U64 Pp = lane_u64(P, hp);\n"
                 "      U64 t64 = P - (h ? Pp : 0ull);\n"
                 "      if (ext) { const U64 E = scan_u64(Eok" + ks + " ? Elo" + ks + " : 0ull); if (lane_ == 63) t64 += E; }\n"
                 "      const u128 t = (u128)(i128)(long long)t64;\n";
        } else {
          src += "      const u128 P = scan_u128(Xok" + ks + " ? (((u128)Xhi" + ks + " << 64) | Xlo" + ks + ") : (u128)0);\n"
                 "      const u128 Pp = lane_u128(P, hp);\n"
                 "      u128 t = P - (h ? Pp : (u128)0);\n"
                 "      if (ext) { const u128 E = scan_u128(Eok" + ks + " ? (((u128)Ehi" + ks + " << 64) | Elo" + ks + ") : (u128)0); if (lane_ == 63) t += E; }\n";
        }
        // (interleaved cells: both words of a sum leave in one 16-byte store)
        src += std::string("      if (plain) { ") +
               (c.inter ? "*reinterpret_cast<ulonglong2*>(" + cell + " + " + gi + ") = make_ulonglong2((U64)t, (U64)(t >> 64)); "
                        : cell + "[" + gi + "] = (U64)t; " + cellhi + "[" + gi + "] = (U64)(t >> 64); ") +
               seen_plain + " }\n"
               "      else if (atom && cnt) { " + add128 + " " + seen_atom + " }\n";
        break;
      case ACC_SUM_I64:
        src += "      const U64 P = scan_u64(Xok" + ks + " ? Xlo" + ks + " : 0ull);\n"
               "      const U64 Pp = lane_u64(P, hp);\n"
               "      U64 t = P - (h ? Pp : 0ull);\n"
               "      if (ext) { const U64 E = scan_u64(Eok" + ks + " ? Elo" + ks + " : 0ull); if (lane_ == 63) t += E; }\n"
               "      if (plain) { " + cell + "[g] = t; " + seen_plain + " }\n"
               "      else if (atom && cnt) { atomicAdd(" + cell + " + g, t); " + seen_atom + " }\n";
        break;
      case ACC_SUM_F64:  // floating point: a difference of prefixes would cancel, so the run is summed by a segmented scan
        src += "      double t = seg_add_f64(Xok" + ks + " ? __longlong_as_double((long long)Xlo" + ks + ") : 0.0, head);\n"
               "      if (ext) { const double e = all_add_f64(Eok" + ks + " ? __longlong_as_double((long long)Elo" + ks + ") : 0.0); if (lane_ == 63) t += e; }\n"
               "      if (plain) { " + cell + "[g] = (U64)__double_as_longlong(t); " + seen_plain + " }\n"
               "      else if (atom && cnt) { atomicAdd(reinterpret_cast<double*>(" + cell + " + g), t); " + seen_atom + " }\n";
        break;
      case ACC_MIN_I64:
        src += "      long long t = seg_min_i64(Xok" + ks + " ? (long long)Xlo" + ks + " : 0x7fffffffffffffffll, head);\n"
               "      if (ext) { const long long e = all_min_i64(Eok" + ks + " ? (long long)Elo" + ks + " : 0x7fffffffffffffffll); if (lane_ == 63) t = e < t ? e : t; }\n"
               "      if (plain) { " + cell + "[g] = (U64)t; " + seen_plain + " }\n"
               "      else if (atom && cnt) { atomicMin(reinterpret_cast<long long*>(" + cell + " + g), t); " + seen_atom + " }\n";
        break;
      case ACC_MAX_I64:
        src += "      long long t = seg_max_i64(Xok" + ks + " ? (long long)Xlo" + ks + " : (-0x7fffffffffffffffll - 1), head);\n"
               "      if (ext) { const long long e = all_max_i64(Eok" + ks + " ? (long long)Elo" + ks + " : (-0x7fffffffffffffffll - 1)); if (lane_ == 63) t = e > t ? e : t; }\n"
               "      if (plain) { " + cell + "[g] = (U64)t; " + seen_plain + " }\n"
               "      else if (atom && cnt) { atomicMax(reinterpret_cast<long long*>(" + cell + " + g), t); " + seen_atom + " }\n";
        break;
      default:  // ACC_COUNT / ACC_COUNT_STAR
        src += "      if (plain) { " + cell + "[g] = cnt; " + seen_plain + " }\n"
               "      else if (atom && cnt) { atomicAdd(" + cell + " + g, cnt); " + seen_atom + " }\n";
        break;
    }
    src += "    }\n";
  }
  src += "  }\n}\n";
  return src;
}

// The ordered-input node.  Returns false (state untouched) when it does not apply.
static bool agg_update_sorted_runs_jit(Aggregate& A, const Table& in, const dfgpu_expr* pred) {
  Runtime& r = rt();
  const int64_t n = in.nrows;
  if (pred || !option_on("jit", true) || !option_on("agg.runs", true) || A.group_roots.empty() || A.ngroups != 0) return false;
  if (n < option_int("jit.min_rows", policy().rows_worth_a_pass()) || n >= 0xFFFFFFFFll) return false;
  int key_col = -1;
  if (!is_plain_column(A.group_nodes[0], A.group_roots[0], &key_col) || key_col < 0 || key_col >= (int)in.cols.size()) return false;
  const Column& kcol = in.cols[(size_t)key_col];
  const dfgpu_field kf = kcol.field;
  if (kcol.validity || !(kf.type == DFGPU_INT32 || kf.type == DFGPU_INT64 || kf.type == DFGPU_DATE32 || kf.type == DFGPU_UINT32 || kf.type == DFGPU_UINT8)) return false;
  // further group keys: plain non-NULL fixed-width columns that the first key determines (checked against the run heads below)
  std::vector<int> more_keys;
  for (size_t g = 1; g < A.group_roots.size(); g++) {
    int c = -1;
    if (!is_plain_column(A.group_nodes[g], A.group_roots[g], &c) || c < 0 || c >= (int)in.cols.size()) return false;
    const Column& mc = in.cols[(size_t)c];
    if (mc.validity || mc.field.type == DFGPU_BOOL || mc.field.type == DFGPU_UTF8) return false;
    more_keys.push_back(c);
  }
  if (!column_stats(const_cast<Column&>(kcol), n).nondecreasing) return false;  // cached on the (immutable) column
  // ---- compile: key + aggregate arguments
  std::string why;
  RowProgramCompiler comp(in);
  dfgpu_expr ke{A.group_nodes[0].data(), (int)A.group_nodes[0].size(), A.group_roots[0]};
  const int key_out = comp.add_output(ke);
  std::vector<int> arg_out(A.aggs.size(), -1);
  for (size_t k = 0; k < A.aggs.size(); k++) {
    AggState& a = A.aggs[k];
    if (!a.has_arg) continue;
    dfgpu_expr e{a.nodes.data(), (int)a.nodes.size(), a.root};
    arg_out[k] = comp.add_output(e);
    dfgpu_field t = comp.output_type(arg_out[k]);
    if (a.typed) DFGPU_CHECK(a.in_type.type == t.type, "aggregate argument type changed between batches");
    AccPlan p = plan_for(a.func, t, false);
    if (p.val == VAL_I32_TO_F64 || p.val == VAL_I64_TO_F64) comp.convert_output(arg_out[k], RP_I2F, t);
    else if (p.val == VAL_F64_ORDERED) comp.convert_output(arg_out[k], RP_F64ORD, t);
  }
  CompiledProgram cp;
  if (!comp.finish(cp, why)) return false;
  // ---- accumulators: one per aggregate (+ the row count of an AVG); every run writes its own cells, nothing is shared
  struct Ent { int agg; bool is_avg_count; int kind; };
  std::vector<Ent> entries;
  std::vector<RunsAcc> accs;
  for (size_t k = 0; k < A.aggs.size(); k++) {
    AggState& a = A.aggs[k];
    dfgpu_field t = a.typed ? a.in_type : (a.has_arg ? cp.out_types[(size_t)arg_out[k]] : fld(DFGPU_INT64));
    AccPlan pl = plan_for(a.func, t, false);
    const int kind = (a.func == DFGPU_AGG_COUNT && !a.has_arg) ? ACC_COUNT_STAR : pl.kind;
    const int val = a.has_arg ? cp.src_out_vals[(size_t)arg_out[k]] : -1;
    entries.push_back({(int)k, false, kind});
    accs.push_back({kind, val, kind == ACC_SUM_I128 && t.type == DFGPU_DECIMAL128 && t.precision > 0 && t.precision <= 16,
                    kind == ACC_SUM_I128 && a.func == DFGPU_AGG_SUM && t.type == DFGPU_DECIMAL128});
    if (a.func == DFGPU_AGG_AVG) {
      entries.push_back({(int)k, true, ACC_COUNT});
      accs.push_back({ACC_COUNT, val, false, false});
    }
  }
  if (accs.size() > (size_t)RUNS_MAX_ACCS) return false;
  const std::string source = agg_runs_node_source(cp, cp.src_out_vals[(size_t)key_out], kf.type, accs);
  hipFunction_t f_acc = nullptr;
  try {
    f_acc = jit_get(source, "runs_accumulate");
  } catch (const Error& e) {
    if (option_on("jit.strict", false)) throw;
    fprintf(stderr, "[dfgpu] node specialisation failed, using the interpreter: %s\n", e.what());
    return false;
  }
  // ---- run heads -> group numbers
  const int64_t n_words = (n + 63) / 64;
  BufPtr heads = make_buf((size_t)n_words * 8);
  {
    ProfileScope ps("agg_runs_heads", n * type_width(kf.type));
    const int g = grid_for(n_words, (BLOCK / WAVE) * 4);
    switch (type_width(kf.type)) {
      case 8: k_run_heads<uint64_t><<<g, BLOCK, 0, r.stream>>>((const uint64_t*)kcol.ptr(), n, heads->as<uint64_t>()); break;
      case 4: k_run_heads<uint32_t><<<g, BLOCK, 0, r.stream>>>((const uint32_t*)kcol.ptr(), n, heads->as<uint64_t>()); break;
      default: k_run_heads<uint8_t><<<g, BLOCK, 0, r.stream>>>((const uint8_t*)kcol.ptr(), n, heads->as<uint64_t>()); break;
    }
    DFGPU_HIP(hipGetLastError());
  }
  BufPtr prefix = make_buf((size_t)(n_words + 1) * 8);
  scan_mask_popcounts(heads->as<uint64_t>(), nullptr, n, prefix->as<uint64_t>());
  BufPtr long_flag = make_zero_buf(8);  // [0] some run needs the atomics, [1] a further key changes inside a run
  k_run_long_flag<<<grid_for(n_words, BLOCK), BLOCK, 0, r.stream>>>(heads->as<uint64_t>(), n_words, long_flag->as<uint32_t>());
  if (!more_keys.empty()) {
    int64_t bytes = 0;
    for (int c : more_keys) bytes += n * type_width(in.cols[(size_t)c].field.type);
    ProfileScope ps("agg_runs_dependent_keys", bytes);
    const int g = grid_for(n, BLOCK * 4);
    for (int c : more_keys) {
      const Column& mc = in.cols[(size_t)c];
      uint32_t* fl = long_flag->as<uint32_t>() + 1;
      switch (type_width(mc.field.type)) {
        case 16: k_run_dependent<uint4><<<g, BLOCK, 0, r.stream>>>((const uint4*)mc.ptr(), heads->as<uint64_t>(), n, fl); break;
        case 8: k_run_dependent<uint64_t><<<g, BLOCK, 0, r.stream>>>((const uint64_t*)mc.ptr(), heads->as<uint64_t>(), n, fl); break;
        case 4: k_run_dependent<uint32_t><<<g, BLOCK, 0, r.stream>>>((const uint32_t*)mc.ptr(), heads->as<uint64_t>(), n, fl); break;
        default: k_run_dependent<uint8_t><<<g, BLOCK, 0, r.stream>>>((const uint8_t*)mc.ptr(), heads->as<uint64_t>(), n, fl); break;
      }
    }
    DFGPU_HIP(hipGetLastError());
  }
  const int64_t G = (int64_t)read_u64(prefix->as<uint64_t>() + n_words);
  uint32_t flags[2] = {0, 0};
  d2h(flags, long_flag->ptr, 8);
  if (flags[1]) return false;  // the runs of the first key are not the groups: the hash path
  const uint32_t has_long = flags[0];
  // ---- from here on state is modified
  for (size_t k = 0; k < A.aggs.size(); k++) {
    AggState& a = A.aggs[k];
    if (!a.typed) {
      a.in_type = a.has_arg ? cp.out_types[(size_t)arg_out[k]] : fld(DFGPU_INT64);
      a.typed = true;
    }
  }
  grow_accumulators(A, 0, G, /*init=*/has_long != 0);  // plain stores cover every cell unless some run needs the atomics
  Column kc = alloc_like(kcol, G);
  kc.name = A.group_names[0];
  RunsNodeArgs args{};
  for (int c = 0; c < cp.prog.n_cols; c++) {
    args.col[c] = cp.prog.col_data[c];
    args.valid[c] = cp.prog.col_valid[c];
  }
  args.heads = heads->as<uint64_t>();
  args.prefix = prefix->as<uint64_t>();
  args.key_out = kc.data->ptr;
  args.n = n;
  for (size_t e = 0; e < entries.size(); e++) {
    AggState& a = A.aggs[(size_t)entries[e].agg];
    if (entries[e].is_avg_count) {
      args.cell[2 * e] = a.cnt->as<unsigned long long>();
    } else if (accs[e].inter) {
      a.inter = has_long ? make_zero_buf((size_t)(G ? G : 1) * 16) : make_buf((size_t)(G ? G : 1) * 16);
      args.cell[2 * e] = a.inter->as<unsigned long long>();
      args.cell[2 * e + 1] = a.inter->as<unsigned long long>() + 1;
      args.seen[e] = a.seen->as<uint32_t>();
    } else {
      args.cell[2 * e] = a.lo->as<unsigned long long>();
      if (entries[e].kind == ACC_SUM_I128) args.cell[2 * e + 1] = a.hi->as<unsigned long long>();
      args.seen[e] = a.seen->as<uint32_t>();
    }
    // an argument that cannot be NULL (no validity anywhere under it): every run has a value — the seen words stay unwritten
    if (!entries[e].is_avg_count && accs[e].val >= 0 && !cp.src_maybe_null[(size_t)accs[e].val] && option_on("agg.runs_seen_all", true)) {
      args.seen[e] = nullptr;
      a.seen_all = true;
    }
  }
  {
    ProfileScope ps("agg_runs_accumulate", n * cp.input_bytes_per_row);
    // (agg.runs_max_blocks: a test's way to make every wave walk many words over a small table.  Measured and dropped in round 6: the next
    // word's head bits and rows loaded one iteration ahead — 4.4 -> 5.2 ms for 600 M rows; the head bits alone a word ahead — 4.3 -> 4.7 on
    // the same kind of box: whatever bounds this kernel at 4.3 TB/s, it is not the head bits -> rows round trip)
    const int max_blocks = (int)option_int("agg.runs_max_blocks", 1 << 30);
    jit_launch(f_acc, std::max(1, std::min(grid_for(n_words, BLOCK / WAVE), max_blocks)), BLOCK, 0, &args, sizeof(args));
  }
  Table gk;
  gk.nrows = G;
  gk.cols.push_back(std::move(kc));
  if (!more_keys.empty()) {  // the further keys of a group = their values at its run head
    Table heads_rows = compact_table(in, more_keys, heads->as<uint64_t>(), nullptr);
    for (size_t g = 0; g < more_keys.size(); g++) {
      heads_rows.cols[g].name = A.group_names[g + 1];
      gk.cols.push_back(std::move(heads_rows.cols[g]));
    }
  }
  A.group_keys = std::move(gk);
  A.ngroups = G;
  DFGPU_HIP(hipStreamSynchronize(r.stream));
  return true;
}

// identities of every key-indexed partial of a range, its first-row words (no row yet) and seen words: one launch
__global__ __launch_bounds__(BLOCK) void k_small_reset(SmallAccSet accs, int D, uint32_t* __restrict__ g_first, uint32_t* __restrict__ g_seen) {
  const int64_t total = (int64_t)D * (accs.n + 1);
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * BLOCK) {
    const int k = (int)(i / D), j = (int)(i % D);
    if (k == accs.n) {
      g_first[j] = 0xFFFFFFFFu;
      g_seen[j] = 0u;
    } else {
      accs.a[k].tmp_lo[j] = acc_identity(accs.a[k].kind);
      if (accs.a[k].tmp_hi) accs.a[k].tmp_hi[j] = 0ull;
    }
  }
}

// The single-pass small-domain node (k_agg_fused_tile).  `cp` = predicate + key bytes + arguments.
// Returns false (state untouched) when the forest has no tile form or the LDS budget does not fit.
constexpr size_t TILE_LDS_BUDGET = 64 * 1024;  // per workgroup: at least two workgroups per CU (160 KiB LDS)
static bool agg_update_small_single_pass(Aggregate& A, const Table& in, const CompiledProgram& cp, const std::vector<int>& small_cols, const int* key_out,
                                         const std::vector<int>& arg_out) {
  Runtime& r = rt();
  const int64_t n = in.nrows;
  const int ngk = (int)small_cols.size();
  const int D = ngk == 0 ? 1 : ngk == 1 ? 256 : SMALL_DOMAIN;
  DFGPU_CHECK(n < 0xFFFFFFFFll, "aggregate input exceeds u32 row ids");
  const TileProgram& T = cp.tile;
  if (T.n_wide < 0) return false;
  const size_t regfile = tile_regfile_bytes(T.n_wide, T.n_narrow);
  if (regfile + 4096 > TILE_LDS_BUDGET) return false;
  // ---- accumulator entries and their LDS cells
  struct Entry { int agg; bool is_avg_count; int kind; int opnd; int val; };
  std::vector<Entry> entries;
  for (size_t k = 0; k < A.aggs.size(); k++) {
    AggState& a = A.aggs[k];
    dfgpu_field t = a.typed ? a.in_type : (a.has_arg ? cp.out_types[arg_out[k]] : fld(DFGPU_INT64));
    AccPlan pl = plan_for(a.func, t, false);
    int kind = (a.func == DFGPU_AGG_COUNT && !a.has_arg) ? ACC_COUNT_STAR : pl.kind;
    int opnd = a.has_arg ? cp.tile_outs[arg_out[k]] : -1;
    int val = a.has_arg ? cp.src_out_vals[arg_out[k]] : -1;
    entries.push_back({(int)k, false, kind, opnd, val});
    if (a.func == DFGPU_AGG_AVG) entries.push_back({(int)k, true, ACC_COUNT, opnd, val});
  }
  DFGPU_CHECK((int)entries.size() <= MAX_AGGS, "too many aggregates for one GPU aggregate node");
  SmallAccSet accs{};
  std::vector<int> src_of(entries.size(), -1);  // entry -> unique accumulator: SUM(x) and AVG(x) share one sum
  std::vector<int> acc_val;                     // unique accumulator -> value id in the generated source
  int ncell = 0;
  for (size_t ei = 0; ei < entries.size(); ei++) {
    const Entry& e = entries[ei];
    for (int u = 0; u < accs.n && src_of[ei] < 0; u++)
      if (accs.a[u].kind == e.kind && accs.a[u].reg == e.opnd) src_of[ei] = u;
    if (src_of[ei] >= 0) continue;
    const int w = e.kind == ACC_SUM_I128 ? 3 : 1;
    if (ncell + w > SM_MAX_CELLS) return false;
    src_of[ei] = accs.n;
    acc_val.push_back(e.val);
    SmallAcc& d = accs.a[accs.n++];
    d.kind = (int16_t)e.kind;
    d.reg = (int16_t)e.opnd;
    d.cell0 = (int16_t)ncell;
    for (int j = 0; j < w; j++) accs.cell_kind[ncell + j] = (uint8_t)(e.kind == ACC_SUM_I128 ? ACC_SUM_I64 : e.kind);
    ncell += w;
  }
  accs.ncell = ncell;
  if (accs.n == 0) return false;
  // LDS left for accumulator planes: (slot, replica) entries of ncell cells + a seen word
  const size_t plane_bytes = (size_t)ncell * 8 + 4;
  const int plane_max = (int)std::min<size_t>((TILE_LDS_BUDGET - regfile - 2048) / plane_bytes, 4096);
  auto slots_for = [&](int64_t groups) {
    int L = 1;
    while (L < groups && L < SM_MAX_L && L < D) L *= 2;
    return L;
  };
  if (D > 1 && slots_for(std::max<int64_t>(A.ngroups, 1)) > plane_max) return false;  // too many groups for LDS: two-pass node

  // ---- key-indexed partials: one set of buffers per range, so that the second range can be launched before the first range's
  // bookkeeping (group numbering, merge) is done on the host
  struct RangeState {
    SmallAccSet accs;
    std::vector<BufPtr> keep;
    BufPtr g_first, g_seen;
    int64_t begin = 0, end = 0;
  };
  auto make_range = [&](int64_t begin, int64_t end) {
    RangeState st;
    st.accs = accs;
    st.begin = begin;
    st.end = end;
    for (int k = 0; k < accs.n; k++) {
      SmallAcc& d = st.accs.a[k];
      BufPtr lo = make_buf((size_t)D * 8);
      st.keep.push_back(lo);
      d.tmp_lo = lo->as<unsigned long long>();
      if (d.kind == ACC_SUM_I128) {
        BufPtr hi = make_buf((size_t)D * 8);
        st.keep.push_back(hi);
        d.tmp_hi = hi->as<unsigned long long>();
      }
    }
    st.g_first = make_buf((size_t)D * 4);
    st.g_seen = make_buf((size_t)D * 4);
    return st;
  };
  const int ko0 = ngk > 0 ? cp.tile_outs[key_out[0]] : -1, ko1 = ngk > 1 ? cp.tile_outs[key_out[1]] : -1;
  const bool prefetch = true;
  DFGPU_HIP(hipFuncSetAttribute((const void*)k_agg_fused_tile<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TILE_LDS_BUDGET));
  DFGPU_HIP(hipFuncSetAttribute((const void*)k_agg_fused_tile<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TILE_LDS_BUDGET));

  // large inputs run the node specialised for this forest (jit.hip); DFGPU_JIT=0 keeps the interpreter
  const int64_t jit_min_rows = option_int("jit.min_rows", policy().rows_worth_a_pass());
  const int plane_max_jit = (int)std::min<size_t>((TILE_LDS_BUDGET - 2048) / plane_bytes, 4096);
  hipFunction_t jit_fn = nullptr;
  if (option_on("jit", true) && n >= jit_min_rows) {
    try {
      jit_fn = jit_get(agg_node_source(cp, accs, acc_val, ngk > 0 ? cp.src_out_vals[key_out[0]] : -1, ngk > 1 ? cp.src_out_vals[key_out[1]] : -1), "agg_node");
    } catch (const Error& e) {
      if (option_on("jit.strict", false)) throw;
      fprintf(stderr, "[dfgpu] node specialisation failed, using the interpreter: %s\n", e.what());
    }
  }

  // launch half: ONE reset kernel (identities of every partial, first-row and seen words) and the node's kernel over [begin, end).
  // `groups_known` = the number of groups the local tables are sized for (0: unknown, 16 slots)
  auto launch_range = [&](RangeState& st, int64_t groups_known) {
    const int64_t begin = st.begin, end = st.end, m = end - begin;
    k_small_reset<<<grid_for((int64_t)D * (st.accs.n + 1), BLOCK), BLOCK, 0, r.stream>>>(st.accs, D, st.g_first->as<uint32_t>(), st.g_seen->as<uint32_t>());
    DFGPU_HIP(hipGetLastError());
    // local slots: the groups known so far (unknown: 16), replicas: what the LDS budget leaves, up to one per lane
    const bool use_jit = jit_fn != nullptr && m >= jit_min_rows;
    const int pmax = use_jit ? plane_max_jit : plane_max;
    int L = D == 1 ? 1 : slots_for(groups_known > 0 ? groups_known : 16);
    while (L > 1 && L > pmax) L /= 2;
    int nrep = 1;
    while (nrep * 2 <= WAVE && L * nrep * 2 <= pmax) nrep *= 2;
    if (m <= 0) return;
    const int plane = L * nrep;
    const size_t cell_bytes = (size_t)ncell * plane * 8 + (size_t)plane * 4 + (size_t)L * 8;
    int grid = grid_for(m, BLOCK * 4);
    const int64_t min_grid = (m >> 20) + 1;  // < 2^20 rows per workgroup: limb sums cannot wrap (see kernel header)
    if (grid < min_grid) grid = (int)min_grid;
    if (use_jit) {
      ProfileScope ps("agg_fused_jit", m * cp.input_bytes_per_row);
      AggNodeArgs args{};
      for (int c = 0; c < T.n_cols; c++) {
        args.col[c] = T.col_data[c];
        args.valid[c] = T.col_valid[c];
      }
      for (int k = 0; k < st.accs.n; k++) {
        args.tmp_lo[k] = st.accs.a[k].tmp_lo;
        args.tmp_hi[k] = st.accs.a[k].tmp_hi;
      }
      args.g_first = st.g_first->as<uint32_t>();
      args.g_seen = st.g_seen->as<uint32_t>();
      args.begin = begin;
      args.end = end;
      args.L = L;
      args.nrep = nrep;
      jit_launch(jit_fn, grid, BLOCK, cell_bytes, &args, sizeof(args));
    } else {
      ProfileScope ps("agg_fused_tile", m * cp.input_bytes_per_row);
      const size_t lds = regfile + cell_bytes;
      if (prefetch) k_agg_fused_tile<true><<<grid, BLOCK, lds, r.stream>>>(T, cp.tile_pred, ko0, ko1, st.accs, begin, end, L, nrep, st.g_first->as<uint32_t>(), st.g_seen->as<uint32_t>());
      else k_agg_fused_tile<false><<<grid, BLOCK, lds, r.stream>>>(T, cp.tile_pred, ko0, ko1, st.accs, begin, end, L, nrep, st.g_first->as<uint32_t>(), st.g_seen->as<uint32_t>());
      DFGPU_HIP(hipGetLastError());
    }
  };
  // the keys a range touched, as the host sees them (waits for the range's kernel)
  auto read_first = [&](RangeState& st) {
    std::vector<uint32_t> hfirst((size_t)D);
    d2h(hfirst.data(), st.g_first->ptr, (size_t)D * 4);
    return hfirst;
  };
  // finish half: number the new groups in first-seen order, grow the accumulators, merge the range's key-indexed partials
  auto finish_range = [&](RangeState& st, const std::vector<uint32_t>& hfirst) {
    const int64_t G0 = A.ngroups;
    std::vector<uint32_t> touched_keys, touched_gids;
    if (ngk == 0) {
      touched_keys.push_back(0);
      touched_gids.push_back(0);
      A.ngroups = 1;  // AggregateStream: one output row even for empty input
    } else {
      small_sync_host_keys(A, ngk);
      std::vector<int64_t> gid_of((size_t)D, -1);
      for (int64_t g = 0; g < G0; g++) gid_of[A.small_keys[(size_t)g]] = g;
      std::vector<std::pair<uint32_t, uint32_t>> fresh;  // (first row, key)
      for (uint32_t k = 0; k < (uint32_t)D; k++)
        if (hfirst[k] != 0xFFFFFFFFu && gid_of[k] < 0) fresh.push_back({hfirst[k], k});
      std::sort(fresh.begin(), fresh.end());  // first-seen order (group_values/mod.rs:88-92)
      for (auto& fk : fresh) {
        gid_of[fk.second] = (int64_t)A.small_keys.size();
        A.small_keys.push_back((uint16_t)fk.second);
      }
      for (uint32_t k = 0; k < (uint32_t)D; k++)
        if (hfirst[k] != 0xFFFFFFFFu) {
          touched_keys.push_back(k);
          touched_gids.push_back((uint32_t)gid_of[k]);
        }
      if (!fresh.empty() || (int)A.group_keys.cols.size() != ngk) small_rebuild_group_keys(A, in, small_cols);
      A.ngroups = (int64_t)A.small_keys.size();
    }
    const int64_t G1 = A.ngroups;
    grow_accumulators(A, G0, G1);
    // ---- merge the key-indexed partials
    const int K = (int)touched_keys.size();
    if (K > 0) {
      MergeDst dst{};
      dst.n = (int)entries.size();
      for (int ei = 0; ei < dst.n; ei++) {
        const Entry& e = entries[(size_t)ei];
        AggState& a = A.aggs[(size_t)e.agg];
        dst.src[ei] = src_of[(size_t)ei];
        if (e.is_avg_count) {
          dst.lo[ei] = a.cnt->as<unsigned long long>();
        } else {
          dst.lo[ei] = a.lo->as<unsigned long long>();
          dst.hi[ei] = a.hi ? a.hi->as<unsigned long long>() : nullptr;
          dst.seen[ei] = a.seen->as<uint32_t>();
        }
      }
      if (K <= SMALL_KG_MAX) {
        SmallKG kgv{};
        for (int i = 0; i < K; i++) {
          kgv.key[i] = touched_keys[(size_t)i];
          kgv.gid[i] = touched_gids[(size_t)i];
        }
        k_small_merge_v<<<grid_for((int64_t)K * dst.n, BLOCK), BLOCK, 0, r.stream>>>(kgv, K, st.accs, dst, st.g_seen->as<uint32_t>());
        DFGPU_HIP(hipGetLastError());
        return;   // (asynchronous: the range's buffers stay alive until the update's last synchronise)
      }
      // (keys and group numbers in ONE upload)
      std::vector<uint32_t> kg(touched_keys);
      kg.insert(kg.end(), touched_gids.begin(), touched_gids.end());
      BufPtr dkg = make_buf((size_t)K * 8);
      h2d_async(dkg->ptr, kg.data(), (size_t)K * 8);
      k_small_merge<<<grid_for((int64_t)K * dst.n, BLOCK), BLOCK, 0, r.stream>>>(dkg->as<uint32_t>(), dkg->as<uint32_t>() + K, K, st.accs, dst, st.g_seen->as<uint32_t>());
      DFGPU_HIP(hipGetLastError());
      DFGPU_HIP(hipStreamSynchronize(r.stream));  // host vectors are released on return
    }
  };
  // The first update of a large input learns the number of groups from a short prefix, so the bulk runs with
  // exactly-sized local tables and as many accumulator replicas as LDS allows.  The bulk's kernel is LAUNCHED as soon as the
  // prefix's keys are on the host (their count sizes its local tables); the prefix's bookkeeping — numbering the groups,
  // growing the accumulators, merging its partials — is issued behind it and overlaps the bulk's run instead of delaying it.
  const int64_t prefix = 1 << 18;
  if (A.ngroups == 0 && D > 1 && n > 4 * prefix) {
    RangeState head = make_range(0, prefix), bulk = make_range(prefix, n);
    launch_range(head, 0);
    const std::vector<uint32_t> first_head = read_first(head);
    int64_t seen_keys = 0;
    for (uint32_t v : first_head) seen_keys += v != 0xFFFFFFFFu;
    launch_range(bulk, seen_keys);
    finish_range(head, first_head);
    finish_range(bulk, read_first(bulk));
    DFGPU_HIP(hipStreamSynchronize(r.stream));   // (the ranges' buffers are still referenced by queued kernels until here)
  } else {
    RangeState all = make_range(0, n);
    launch_range(all, A.ngroups);
    finish_range(all, read_first(all));
    DFGPU_HIP(hipStreamSynchronize(r.stream));
  }
  return true;
}

// Returns false (nothing changed) when the forest cannot be fused; the caller then takes the
// column-at-a-time path.
static bool agg_update_fused(Aggregate& A, const Table& in, const dfgpu_expr* pred, std::string& why) {
  Runtime& r = rt();
  const int64_t n = in.nrows;
  const int ngk = (int)A.group_roots.size();
  if (A.final_mode() || n == 0 || !g_fusion_enabled) {
    why = "not a raw-input update";
    return false;
  }
  if (A.aggs.empty()) {   // gby=[...], aggr=[] (the inner level of COUNT(DISTINCT), q16.slt.part:75-77): the keys are interned, nothing accumulates
    why = "no aggregates";
    return false;
  }
  for (const AggState& a : A.aggs)
    if (a.has_arg && (a.func == DFGPU_AGG_MIN || a.func == DFGPU_AGG_MAX)) {
      dfgpu_expr e{a.nodes.data(), (int)a.nodes.size(), a.root};
      if (wide_minmax(a.func, expr_type(e, in))) {
        why = "MIN/MAX over a Decimal128 wider than 18 digits checks its values column-at-a-time";
        return false;
      }
    }
  std::vector<int> small_cols;
  const bool small = small_domain_applicable(A, in, small_cols);
  const int gid_mode = ngk == 0 ? GID_NONE : small ? GID_SMALL : GID_HASH;
  if (gid_mode == GID_HASH && A.ngroups == 0 && agg_update_sorted_runs_jit(A, in, pred)) return true;
  if (gid_mode == GID_HASH && A.ngroups == 0 && agg_update_dense_key_jit(A, in, pred)) return true;

  // ---- compile: predicate, (small mode) key bytes, aggregate arguments
  RowProgramCompiler comp(in);
  if (pred) comp.set_predicate(*pred);
  int key_out[2] = {-1, -1};
  if (gid_mode == GID_SMALL)
    for (int g = 0; g < ngk; g++) {
      dfgpu_expr e{A.group_nodes[g].data(), (int)A.group_nodes[g].size(), A.group_roots[g]};
      key_out[g] = comp.add_output(e);
    }
  std::vector<int> arg_out(A.aggs.size(), -1);
  for (size_t k = 0; k < A.aggs.size(); k++) {
    AggState& a = A.aggs[k];
    if (!a.has_arg) continue;
    dfgpu_expr e{a.nodes.data(), (int)a.nodes.size(), a.root};
    arg_out[k] = comp.add_output(e);
    dfgpu_field t = comp.output_type(arg_out[k]);
    if (a.typed) DFGPU_CHECK(a.in_type.type == t.type, "aggregate argument type changed between batches");
    // conversions the accumulator expects (plan_for): AVG over ints sums f64; MIN/MAX(f64) on the ordered key
    AccPlan p = plan_for(a.func, t, false);
    if (p.val == VAL_I32_TO_F64 || p.val == VAL_I64_TO_F64) comp.convert_output(arg_out[k], RP_I2F, t);
    else if (p.val == VAL_F64_ORDERED) comp.convert_output(arg_out[k], RP_F64ORD, t);
  }
  CompiledProgram cp;
  if (!comp.finish(cp, why)) return false;

  // key-only program for the small-domain intern pass (touches the predicate and key columns only)
  CompiledProgram kp;
  if (gid_mode == GID_SMALL) {
    RowProgramCompiler kc(in);
    if (pred) kc.set_predicate(*pred);
    for (int g = 0; g < ngk; g++) {
      dfgpu_expr e{A.group_nodes[g].data(), (int)A.group_nodes[g].size(), A.group_roots[g]};
      kc.add_output(e);
    }
    if (!kc.finish(kp, why)) return false;
  }

  // ---- from here on state is modified
  for (size_t k = 0; k < A.aggs.size(); k++) {
    AggState& a = A.aggs[k];
    if (!a.typed) {
      a.in_type = a.has_arg ? cp.out_types[arg_out[k]] : fld(DFGPU_INT64);
      a.typed = true;
    }
  }
  if (gid_mode != GID_HASH &&
      agg_update_small_single_pass(A, in, cp, small_cols, key_out, arg_out))
    return true;
  const int64_t G0 = A.ngroups;
  int64_t G1 = G0;
  GidSpec gs{};
  gs.mode = gid_mode;
  gs.key_reg0 = gs.key_reg1 = -1;
  InternResult IR;
  BufPtr gid_table, pred_mask_keepalive;
  std::vector<Column> key_cols_keepalive;
  if (gid_mode == GID_NONE) {
    G1 = 1;
  } else if (gid_mode == GID_SMALL) {
    // host mirror of the existing groups' keys
    if ((int64_t)A.small_keys.size() != G0) {
      A.small_keys.assign((size_t)G0, 0);
      for (int g = 0; g < ngk && G0; g++) {
        std::vector<uint8_t> b((size_t)G0);
        d2h(b.data(), A.group_keys.cols[g].ptr(), (size_t)G0);
        for (int64_t i = 0; i < G0; i++) A.small_keys[(size_t)i] |= (uint16_t)(b[(size_t)i] << (8 * g));
      }
    }
    BufPtr first = make_buf((size_t)SMALL_DOMAIN * 4);
    DFGPU_HIP(hipMemsetAsync(first->ptr, 0xFF, (size_t)SMALL_DOMAIN * 4, r.stream));
    {
      ProfileScope ps("agg_small_domain_intern", n * kp.input_bytes_per_row);
      auto kern = kp.n_regs <= 16 ? k_small_first_rows<16> : k_small_first_rows<32>;
      kern<<<grid_for(n, BLOCK * 8), BLOCK, 0, r.stream>>>(kp.prog, kp.n_prologue, kp.n_pred_end, kp.pred_reg, kp.out_regs[0],
                                                           ngk > 1 ? kp.out_regs[1] : -1, n, first->as<uint32_t>());
      DFGPU_HIP(hipGetLastError());
    }
    std::vector<uint32_t> hfirst((size_t)SMALL_DOMAIN);
    d2h(hfirst.data(), first->ptr, (size_t)SMALL_DOMAIN * 4);
    std::vector<uint32_t> table((size_t)SMALL_DOMAIN, 0u);
    std::vector<bool> known((size_t)SMALL_DOMAIN, false);
    for (int64_t gidx = 0; gidx < G0; gidx++) {
      table[A.small_keys[(size_t)gidx]] = (uint32_t)gidx;
      known[A.small_keys[(size_t)gidx]] = true;
    }
    std::vector<std::pair<uint32_t, uint32_t>> fresh;  // (first row, key)
    for (uint32_t k = 0; k < (uint32_t)SMALL_DOMAIN; k++)
      if (hfirst[k] != 0xFFFFFFFFu && !known[k]) fresh.push_back({hfirst[k], k});
    std::sort(fresh.begin(), fresh.end());  // first-seen order (group_values/mod.rs:88-92)
    for (auto& fk : fresh) {
      table[fk.second] = (uint32_t)A.small_keys.size();
      A.small_keys.push_back((uint16_t)fk.second);
    }
    G1 = (int64_t)A.small_keys.size();
    gid_table = make_buf((size_t)SMALL_DOMAIN * 4);
    h2d_async(gid_table->ptr, table.data(), (size_t)SMALL_DOMAIN * 4);
    // dense group key columns rebuilt from the host mirror
    Table gk;
    gk.nrows = G1;
    for (int g = 0; g < ngk; g++) {
      Column c = alloc_column(in.cols[small_cols[g]].field, A.group_names[g], G1);
      c.dict = in.cols[small_cols[g]].dict;
      std::vector<uint8_t> b((size_t)(G1 ? G1 : 1));
      for (int64_t i = 0; i < G1; i++) b[(size_t)i] = (uint8_t)(A.small_keys[(size_t)i] >> (8 * g));
      if (G1) h2d_async(c.data->ptr, b.data(), (size_t)G1);
      DFGPU_HIP(hipStreamSynchronize(r.stream));  // b and table are host temporaries
      gk.cols.push_back(std::move(c));
    }
    A.group_keys = std::move(gk);
    gs.key_reg0 = cp.out_regs[key_out[0]];
    gs.key_reg1 = ngk > 1 ? cp.out_regs[key_out[1]] : -1;
    gs.gid_table = gid_table->as<uint32_t>();
  } else {
    // hash interning needs the key columns in memory: plain column references are used in place
    for (int g = 0; g < ngk; g++) {
      int c = -1;
      if (is_plain_column(A.group_nodes[g], A.group_roots[g], &c)) {
        DFGPU_CHECK(c >= 0 && c < (int)in.cols.size(), "Column index out of range");
        key_cols_keepalive.push_back(in.cols[c]);
      } else {
        dfgpu_expr e{A.group_nodes[g].data(), (int)A.group_nodes[g].size(), A.group_roots[g]};
        key_cols_keepalive.push_back(datum_to_column(evaluate(e, in), n, A.group_names[g]));
      }
      key_cols_keepalive.back().name = A.group_names[g];
    }
    const uint64_t* row_mask = nullptr;
    Column mask_col;
    if (pred) {
      // rows failing the predicate must not create groups: intern under the predicate's mask
      mask_col = datum_to_column(evaluate(*pred, in), n, "");
      DFGPU_CHECK(mask_col.field.type == DFGPU_BOOL, "Cannot create filter with non-boolean predicate");
      if (mask_col.validity) {
        BufPtr m = make_buf(bitmap_bytes(n));
        int64_t nw = (n + 63) / 64;
        and_bitmaps(mask_col.data->as<uint64_t>(), mask_col.valid_words(), nw, m->as<uint64_t>());
        pred_mask_keepalive = m;
      } else {
        pred_mask_keepalive = mask_col.data;
      }
      row_mask = pred_mask_keepalive->as<uint64_t>();
    }
    // (large inputs may take the partitioned accumulation below: it wants every row's slot from the claim pass)
    const bool maybe_partitioned = n >= option_int("agg.partitioned_min_rows", 2 * policy().rows_worth_a_pass()) && option_on("agg.partitioned", true);
    // (statistics the interning takes of plain key columns are kept on the table's own columns — the same rows: the next query finds them)
    std::vector<Column*> homes((size_t)ngk, nullptr);
    for (int g = 0; g < ngk; g++) {
      int c = -1;
      if (is_plain_column(A.group_nodes[g], A.group_roots[g], &c) && c >= 0 && c < (int)in.cols.size()) homes[(size_t)g] = const_cast<Column*>(&in.cols[(size_t)c]);
    }
    IR = intern_keys(A, key_cols_keepalive, n, row_mask, maybe_partitioned, &homes);
    G1 = IR.G1;
    gs.ictx = IR.ictx;
    gs.slot_gid = IR.slot_gid->as<uint32_t>();
    gs.row_offset = G0;
  }

  // ---- accumulators
  std::vector<AccPlan> plans = grow_accumulators(A, G0, G1);
  FusedAccSet accs{};
  for (size_t k = 0; k < A.aggs.size(); k++) {
    AggState& a = A.aggs[k];
    FusedAcc d{};
    d.kind = (int16_t)((a.func == DFGPU_AGG_COUNT && !a.has_arg) ? ACC_COUNT_STAR : plans[k].kind);
    d.reg = (int16_t)(a.has_arg ? cp.out_regs[arg_out[k]] : -1);
    d.acc_lo = a.lo->as<unsigned long long>();
    d.acc_hi = a.hi ? a.hi->as<unsigned long long>() : nullptr;
    d.seen = a.seen->as<uint32_t>();
    DFGPU_CHECK(accs.n < MAX_AGGS, "too many aggregates for one GPU aggregate node");
    accs.a[accs.n++] = d;
    if (a.func == DFGPU_AGG_AVG) {
      FusedAcc c{};
      c.kind = ACC_COUNT;  // counts the non-null arguments
      c.reg = d.reg;
      c.acc_lo = a.cnt->as<unsigned long long>();
      DFGPU_CHECK(accs.n < MAX_AGGS, "too many aggregates for one GPU aggregate node");
      accs.a[accs.n++] = c;
    }
  }
  if (accs.n > 0) {
    const int64_t bytes = n * cp.input_bytes_per_row;
    if (G1 * accs.n <= LDS_CELLS) {
      int nrep = 1;
      while (nrep * 2 <= 32 && (int64_t)nrep * 2 * G1 * accs.n <= LDS_CELLS) nrep *= 2;
      ProfileScope ps("agg_fused_lds", bytes);
      auto kern = cp.n_regs <= 16 ? k_agg_fused<true, 16> : k_agg_fused<true, 32>;
      kern<<<grid_for(n, BLOCK * 8), BLOCK, 0, r.stream>>>(cp.prog, cp.n_prologue, cp.n_pred_end, cp.pred_reg, gs, accs, n, (int)G1, nrep);
    } else if (gid_mode == GID_HASH && fused_general_partitioned(A, in, plans, gs.ictx, gs.slot_gid, G0, G1, pred ? pred_mask_keepalive->as<uint64_t>() : nullptr, IR.row_slot ? IR.row_slot->as<uint32_t>() : nullptr)) {
      // (medium cardinalities: rows moved by group number into LDS-sized windows instead of one global atomic per row and aggregate)
    } else {
      ProfileScope ps("agg_fused_global", bytes);
      auto kern = cp.n_regs <= 16 ? k_agg_fused<false, 16> : k_agg_fused<false, 32>;
      kern<<<grid_for(n, BLOCK), BLOCK, 0, r.stream>>>(cp.prog, cp.n_prologue, cp.n_pred_end, cp.pred_reg, gs, accs, n, (int)std::min<int64_t>(G1, INT32_MAX), 1);
    }
    DFGPU_HIP(hipGetLastError());
  }
  A.ngroups = G1;
  DFGPU_HIP(hipStreamSynchronize(r.stream));  // temporaries are released on return
  return true;
}

static void agg_update_unfused(Aggregate& A, const Table& in) {
  Runtime& r = rt();
  const int64_t n = in.nrows;
  const int ngk = (int)A.group_roots.size();
  const bool final_mode = A.final_mode();

  // ---- evaluate group keys and aggregate arguments (evaluate_batch, common.rs:169-197)
  std::vector<Column> key_cols;
  for (int g = 0; g < ngk; g++) {
    if (final_mode) {
      DFGPU_CHECK(g < (int)in.cols.size(), "final aggregate input has too few columns");
      key_cols.push_back(in.cols[g]);
    } else {
      dfgpu_expr e{A.group_nodes[g].data(), (int)A.group_nodes[g].size(), A.group_roots[g]};
      key_cols.push_back(datum_to_column(evaluate(e, in), n, A.group_names[g]));
    }
    key_cols.back().name = A.group_names[g];
  }
  // argument / state columns per aggregate
  struct Inputs { Column v; Column c; bool has_v = false, has_c = false; };
  std::vector<Inputs> inputs(A.aggs.size());
  int state_col = ngk;
  for (size_t k = 0; k < A.aggs.size(); k++) {
    AggState& a = A.aggs[k];
    if (final_mode) {
      // partial-state schema: AVG -> [count, sum]; others -> one column (average.rs:317-360, sum.rs:281-301)
      if (a.func == DFGPU_AGG_AVG) {
        DFGPU_CHECK(state_col + 1 < (int)in.cols.size(), "final aggregate input has too few state columns");
        inputs[k].c = in.cols[state_col++];
        inputs[k].has_c = true;
      }
      DFGPU_CHECK(state_col < (int)in.cols.size(), "final aggregate input has too few state columns");
      inputs[k].v = in.cols[state_col++];
      inputs[k].has_v = true;
    } else if (a.has_arg) {
      dfgpu_expr e{a.nodes.data(), (int)a.nodes.size(), a.root};
      inputs[k].v = datum_to_column(evaluate(e, in), n, a.name);
      inputs[k].has_v = true;
    }
    if (!a.typed) {
      if (inputs[k].has_v) a.in_type = inputs[k].v.field;
      else a.in_type = fld(DFGPU_INT64);
      a.typed = true;
    }
    if (inputs[k].has_v && wide_minmax(a.func, inputs[k].v.field)) wide_minmax_values_fit(inputs[k].v, a.name);
  }

  // ---- intern (GroupValues::intern)
  const int64_t G0 = A.ngroups;
  int64_t G1 = G0;
  InternResult IR;
  if (ngk > 0) {
    IR = intern_keys(A, key_cols, n, nullptr);
    G1 = IR.G1;
  } else {
    G1 = 1;  // no GROUP BY: AggregateStream, one output row even for empty input
  }
  InternCtx& ictx = IR.ictx;

  // ---- grow accumulators to G1 groups
  std::vector<AccPlan> plans = grow_accumulators(A, G0, G1);
  AccSet accs{};
  for (size_t k = 0; k < A.aggs.size(); k++) {
    AggState& a = A.aggs[k];
    const AccPlan& p = plans[k];
    AccDesc d{};
    d.kind = (a.func == DFGPU_AGG_COUNT && !a.has_arg && !final_mode) ? ACC_COUNT_STAR : p.kind;
    d.val = p.val;
    d.values = inputs[k].has_v ? inputs[k].v.ptr() : nullptr;
    d.valid = inputs[k].has_v ? inputs[k].v.valid_words() : nullptr;
    d.narrow = inputs[k].has_v && (p.val != VAL_I128 || (inputs[k].v.field.type == DFGPU_DECIMAL128 && inputs[k].v.field.precision <= 18)) ? 1 : 0;
    d.acc_lo = a.lo->as<unsigned long long>();
    d.acc_hi = a.hi ? a.hi->as<unsigned long long>() : nullptr;
    d.seen = a.seen->as<uint32_t>();
    DFGPU_CHECK(accs.n < MAX_AGGS, "too many aggregates for one GPU aggregate node");
    accs.a[accs.n++] = d;
    if (a.func == DFGPU_AGG_AVG) {
      // companion count accumulator: raw modes count non-null args, final modes add partial counts
      AccDesc c{};
      c.acc_lo = a.cnt->as<unsigned long long>();
      if (final_mode) {
        c.kind = ACC_SUM_I64;
        c.val = VAL_U64;
        c.values = inputs[k].c.ptr();
        c.valid = inputs[k].c.valid_words();
      } else {
        c.kind = ACC_COUNT;
        c.val = VAL_I64;
        c.values = d.values;
        c.valid = d.valid;
      }
      DFGPU_CHECK(accs.n < MAX_AGGS, "too many aggregates for one GPU aggregate node");
      accs.a[accs.n++] = c;
    }
  }
  if (n == 0 || accs.n == 0) {
    A.ngroups = G1;
    return;
  }
  // ---- accumulate (update_batch / merge_batch)
  int64_t bytes = 0;
  for (int k = 0; k < accs.n; k++)
    if (accs.a[k].values) bytes += n * 8;
  const int per_rep = accs.n * (int)std::min<int64_t>(G1, LDS_CELLS + 1);
  const uint32_t* sg = IR.slot_gid ? IR.slot_gid->as<uint32_t>() : nullptr;
  if (G1 * accs.n <= LDS_CELLS) {
    int nrep = std::max(1, std::min(64, LDS_CELLS / per_rep));
    // power of two so that consecutive lanes spread over the replicas
    int p2 = 1;
    while (p2 * 2 <= nrep) p2 *= 2;
    ProfileScope ps("agg_accumulate_lds", bytes);
    k_accumulate_lds<<<grid_for(n, BLOCK * 8), BLOCK, 0, r.stream>>>(ictx, sg, ngk > 0, G0, n, accs, (int)G1, p2);
  } else if (ngk > 0 && general_accumulate_partitioned(ictx, sg, G0, n, accs, G1)) {
    // (rows moved into LDS-sized windows of group numbers, accumulated there, merged per group)
  } else {
    ProfileScope ps("agg_accumulate_global", bytes);
    k_accumulate_global<<<grid_for(n, BLOCK), BLOCK, 0, r.stream>>>(ictx, sg, ngk > 0, G0, n, accs);
  }
  DFGPU_HIP(hipGetLastError());
  A.ngroups = G1;
  DFGPU_HIP(hipStreamSynchronize(r.stream));  // temporaries (evaluated columns, tables) are released on return
}

// aggregate_batch_inner over a whole table, optionally under a FilterExec predicate fused in front
__global__ __launch_bounds__(BLOCK) void k_bits_to_u8(const uint64_t* __restrict__ bits, int64_t n, uint8_t* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) out[i] = (uint8_t)((bits[i >> 6] >> (i & 63)) & 1ull);
}
// a Boolean column (bit-packed) as one UInt8 per row, validity kept: how Boolean KEY columns enter the hashing / ordering kernels
// (group keys here; join, repartition and sort keys in their files) — hash_utils.rs:306-345 hashes a BooleanArray value by value too
Column bool_as_u8(const Column& c, int64_t n) {
  dfgpu_field f{};
  f.type = DFGPU_UINT8;
  f.nullable = c.field.nullable;
  Column u = alloc_column(f, c.name, n);
  if (n) k_bits_to_u8<<<grid_for(n, BLOCK), BLOCK, 0, rt().stream>>>((const uint64_t*)c.ptr(), n, u.data->as<uint8_t>());
  DFGPU_HIP(hipGetLastError());
  u.validity = c.validity;
  u.null_count = c.null_count;
  return u;
}
static void agg_update_keys_fixed(Aggregate& A, const Table& in, const dfgpu_expr* pred);
// Utf8 group keys (plain column references; in Final modes the leading columns): interned on entry with an ascending dictionary —
// grouping on the indices is grouping on the strings — and decoded again when the groups are emitted, so the node's schema keeps
// Utf8.  One update per aggregate: a second batch would arrive with another dictionary (the check below says so).
static void agg_update(Aggregate& A, const Table& in, const dfgpu_expr* pred = nullptr) {
  const int ngk = (int)A.group_roots.size();
  Table coded;
  bool any = false;
  for (int g = 0; g < ngk; g++) {
    int kc = A.final_mode() ? g : -1;
    if (!A.final_mode() && !is_plain_column(A.group_nodes[(size_t)g], A.group_roots[(size_t)g], &kc)) continue;
    if (kc >= 0 && kc < (int)in.cols.size() && in.cols[(size_t)kc].field.type == DFGPU_BOOL) {
      // Boolean group keys (plain column references): one byte per row on entry, a Boolean column again on emit
      if (!any) coded = in;
      any = true;
      const Column& bc = in.cols[(size_t)kc];
      if (coded.cols[(size_t)kc].field.type == DFGPU_BOOL) coded.cols[(size_t)kc] = bool_as_u8(bc, in.nrows);
      A.bool_key.resize((size_t)ngk, 0);
      A.bool_key[(size_t)g] = 1;
      continue;
    }
    if (kc < 0 || kc >= (int)in.cols.size() || in.cols[(size_t)kc].field.type != DFGPU_UTF8 || in.cols[(size_t)kc].dict) continue;
    if (!any) coded = in;
    any = true;
    if (coded.cols[(size_t)kc].field.type == DFGPU_UTF8) coded.cols[(size_t)kc] = dictionary_encode(in.cols[(size_t)kc], true);
    A.utf8_key.resize((size_t)ngk, 0);
    A.utf8_key[(size_t)g] = 1;
  }
  agg_update_keys_fixed(A, any ? coded : in, pred);
}
static void agg_update_keys_fixed(Aggregate& A, const Table& in, const dfgpu_expr* pred) {
  if (A.ngroups > 0) materialize_seen(A, A.ngroups);   // (a batch after the runs node's)
  // group keys that are dictionary-encoded columns are interned by their indices: every update must use the dictionary of
  // the groups that exist already (all interning paths — LDS cells, dense ranks, hash — rely on this)
  if (A.ngroups > 0)
    for (size_t g = 0; g < A.group_roots.size() && g < A.group_keys.cols.size(); g++) {
      int kc = A.final_mode() ? (int)g : -1;
      if (!A.final_mode() && !is_plain_column(A.group_nodes[g], A.group_roots[g], &kc)) continue;
      if (kc < 0 || kc >= (int)in.cols.size()) continue;
      DFGPU_CHECK(same_dictionary(A.group_keys.cols[g].dict, in.cols[kc].dict),
                  "group key " + A.group_names[g] + ": the dictionary changed between updates (unify the inputs' dictionaries first, e.g. dfgpu_table_concat)");
    }
  std::string why;
  if (agg_update_fused(A, in, pred, why)) {
    A.fused_updates++;
    return;
  }
  if (!pred) {
    agg_update_unfused(A, in);
    return;
  }
  // unfused fallback of the fused node: FilterExec, then the aggregate over its output
  Datum m = evaluate(*pred, in);
  DFGPU_CHECK(m.col.field.type == DFGPU_BOOL, "Cannot create filter with non-boolean predicate");
  Column mc = datum_to_column(m, in.nrows, "");
  std::vector<int> all(in.cols.size());
  for (size_t i = 0; i < all.size(); i++) all[i] = (int)i;
  Table filtered = compact_table(in, all, mc.data->as<uint64_t>(), mc.valid_words());
  agg_update_unfused(A, filtered);
}


static Column emit_column(const dfgpu_field& f, const std::string& name, int mode, const BufPtr& lo, const BufPtr& hi, const BufPtr& seen, int64_t n,
                          bool nullable) {
  Column c = alloc_column(f, name, n);
  if (n == 0) return c;
  BufPtr vb = nullable ? make_buf((size_t)n + 64) : nullptr;
  k_emit_values<<<grid_for(n, BLOCK), BLOCK, 0, rt().stream>>>(mode, lo->as<unsigned long long>(), hi ? hi->as<unsigned long long>() : nullptr,
                                                               nullable ? seen->as<uint32_t>() : nullptr, n, c.data->ptr, vb ? vb->as<uint8_t>() : nullptr);
  DFGPU_HIP(hipGetLastError());
  if (nullable) {
    c.validity = make_buf(bitmap_bytes(n));
    pack_bytes_to_bitmap(vb->as<uint8_t>(), n, c.validity->as<uint64_t>());
    c.null_count = -1;
    count_nulls(c);
  }
  return c;
}

static Table agg_emit(Aggregate& A) {
  Runtime& r = rt();
  const int ngk = (int)A.group_roots.size();
  if (ngk == 0 && A.ngroups == 0) {
    // no input at all: still one row (accumulators at identity)
    Table empty;
    empty.nrows = 0;
    agg_update(A, empty);
  }
  const int64_t G = A.ngroups;
  Table out;
  out.nrows = G;
  // the aggregates' columns are written by ONE kernel (k_emit_set) launched after this loop; `pending[e]` = the column entry e fills
  EmitSet eset{};
  std::vector<size_t> pending;
  auto emit_later = [&](const dfgpu_field& f, const std::string& name, int kind, int mode, const BufPtr& lo, const BufPtr& hi, const BufPtr& seen,
                        const BufPtr& cnt, i128 mul, bool nullable) {
    Column c = alloc_column(f, name, G);
    if (G > 0) {
      DFGPU_CHECK(eset.n < EMIT_MAX, "too many aggregate output columns for one GPU aggregate node");
      EmitEntry& e = eset.e[eset.n++];
      e.kind = kind;
      e.mode = mode;
      e.lo = lo ? lo->as<unsigned long long>() : nullptr;
      e.hi = hi ? hi->as<unsigned long long>() : nullptr;
      e.cnt = cnt ? cnt->as<unsigned long long>() : nullptr;
      e.seen = (nullable && seen) ? seen->as<uint32_t>() : nullptr;
      e.mul_lo = (unsigned long long)(u128)mul;
      e.mul_hi = (unsigned long long)((u128)mul >> 64);
      e.dst = c.data->ptr;
      if (nullable) {
        c.validity = make_buf(bitmap_bytes(G));
        c.null_count = -1;
        e.valid_words = c.validity->as<uint64_t>();
      }
      pending.push_back(out.cols.size());
    }
    out.cols.push_back(std::move(c));
  };
  for (int g = 0; g < ngk; g++) {
    if (G == 0 && (int)A.group_keys.cols.size() <= g) throw Error("aggregate emitted before any input: group key types unknown");
    out.cols.push_back(A.group_keys.cols[g]);
    if ((size_t)g < A.utf8_key.size() && A.utf8_key[(size_t)g] && out.cols.back().dict) {
      out.cols.back() = dictionary_decode(out.cols.back());
      out.cols.back().name = A.group_names[(size_t)g];
    }
    if ((size_t)g < A.bool_key.size() && A.bool_key[(size_t)g] && out.cols.back().field.type == DFGPU_UINT8) {
      const Column& u = out.cols.back();
      Column b = alloc_column(fld(DFGPU_BOOL), A.group_names[(size_t)g], G);
      if (G) pack_bytes_to_bitmap((const uint8_t*)u.ptr(), G, b.data->as<uint64_t>());
      b.validity = u.validity;
      b.null_count = u.null_count;
      out.cols.back() = std::move(b);
    }
  }
  for (AggState& a : A.aggs) {
    DFGPU_CHECK(a.typed || G == 0, "aggregate emitted before any input");
    if (!a.typed) {
      a.in_type = fld(DFGPU_INT64);
      a.typed = true;
    }
    const bool fin = A.final_mode();
    AccPlan p = plan_for(a.func, a.in_type, fin);
    // value type of the primary accumulator when emitted
    auto value_field = [&]() -> dfgpu_field {
      switch (a.func) {
        case DFGPU_AGG_COUNT: return fld(DFGPU_INT64);
        case DFGPU_AGG_SUM: return fin ? a.in_type : sum_type(a.in_type);
        case DFGPU_AGG_AVG: return fin ? a.in_type : avg_sum_type(a.in_type);
        default: return a.in_type;
      }
    };
    dfgpu_field vf = value_field();
    int mode = 0;
    if (p.kind == ACC_SUM_I128) mode = 1;
    else if (p.val == VAL_F64_ORDERED) mode = 2;
    else if (vf.type == DFGPU_INT32 || vf.type == DFGPU_DATE32) mode = 3;
    else if (vf.type == DFGPU_UINT8) mode = 4;
    if (a.func == DFGPU_AGG_AVG) {
      if (A.partial_out()) {
        // state_fields of AVG: [count: UInt64, sum] (average.rs:317-360)
        emit_later(fld(DFGPU_UINT64), a.name + "[count]", 0, 0, a.cnt, nullptr, nullptr, nullptr, 1, false);
        emit_later(vf, a.name + "[sum]", 0, mode, a.lo, a.hi, a.seen, nullptr, 1, !a.seen_all);
      } else {
        // raw modes: in_type = argument type Decimal(p,s) -> AVG type Decimal(min(38,p+4), min(38,s+4))
        // (average.rs:219-252).  final modes: in_type = the sum state Decimal(38, s), which no longer tells p: the
        // planner-declared return type (AggregateFunctionExpr::return_field) is required.
        dfgpu_field base = a.in_type;
        dfgpu_field rt_;
        i128 mul = 1;
        bool dec = base.type == DFGPU_DECIMAL128;
        if (dec) {
          int s = base.scale;
          if (!fin) (void)avg_sum_type(base);   // precision check: the i128 sum must have the reference's headroom
          DFGPU_CHECK(!fin || a.ret.type == DFGPU_DECIMAL128,
                      "Final AVG over a Decimal128 state needs the aggregate's declared return type (dfgpu_agg_spec.return_field)");
          rt_ = a.ret.type == DFGPU_DECIMAL128 ? a.ret : fld(DFGPU_DECIMAL128, std::min(38, base.precision + 4), std::min(38, s + 4));
          DFGPU_CHECK(rt_.scale >= s, "AVG return scale smaller than the sum scale");
          for (int i = s; i < rt_.scale; i++) mul *= 10;
        } else {
          rt_ = fld(DFGPU_FLOAT64);
        }
        emit_later(rt_, a.name, 1, dec ? 1 : 0, a.lo, a.hi, nullptr, a.cnt, mul, true);
      }
      continue;
    }
    bool nullable = a.func != DFGPU_AGG_COUNT && !a.seen_all;
    // partial state field names: format_state_name (expr/src/utils.rs:1416) — `name[sum]` (sum.rs:293-299), `name[count]`
    // (count.rs:317-323), `name[value]` for MIN / MAX (the default AggregateUDFImpl::state_fields, expr/src/udaf.rs:579-585)
    const std::string out_name = !A.partial_out() ? a.name : a.name + (a.func == DFGPU_AGG_SUM ? "[sum]" : a.func == DFGPU_AGG_COUNT ? "[count]" : "[value]");
    if (vf.type == DFGPU_DECIMAL128 && p.kind != ACC_SUM_I128) {
      // widen the i64 MIN/MAX accumulator to i128: hi = sign(lo)
      Column c = emit_column(fld(DFGPU_INT64), out_name, 0, a.lo, nullptr, a.seen, G, nullable);
      Column w = alloc_column(vf, out_name, G);
      if (G) {
        dfgpu_expr_node nodes[2]{};
        nodes[0].op = DFGPU_EXPR_COLUMN; nodes[0].column = 0; nodes[0].left = nodes[0].right = -1;
        nodes[1].op = DFGPU_EXPR_CAST; nodes[1].left = 0; nodes[1].right = -1; nodes[1].field = fld(DFGPU_DECIMAL128, vf.precision, 0);
        Table t1;
        t1.nrows = G;
        t1.cols.push_back(c);
        dfgpu_expr e{nodes, 2, 1};
        Column casted = datum_to_column(evaluate(e, t1), G, out_name);
        casted.field = vf;
        w = casted;
      }
      out.cols.push_back(std::move(w));
      continue;
    }
    if (a.inter && p.kind == ACC_SUM_I128 && vf.type == DFGPU_DECIMAL128 && G > 0) {
      // the runs node's interleaved cells ARE the column (a group without a non-NULL value holds 0, as emit would write)
      Column c;
      c.field = vf;
      c.name = out_name;
      c.length = G;
      c.data = a.inter;
      if (a.seen_all) {   // every group has a value: no validity buffer at all
        c.null_count = 0;
        out.cols.push_back(std::move(c));
        continue;
      }
      // the validity words and the count of valid groups come from the seen flags in k_emit_set's pass (mode 5: no value is written) —
      // a byte per group, the packing of the bytes and a count over the words were three passes over 150 M groups (0.40 of 5.8 ms)
      DFGPU_CHECK(eset.n < EMIT_MAX, "too many aggregate output columns for one GPU aggregate node");
      c.validity = make_buf(bitmap_bytes(G));
      c.null_count = -1;
      EmitEntry& e = eset.e[eset.n++];
      e = EmitEntry{};
      e.kind = 0;
      e.mode = 5;
      e.seen = a.seen->as<uint32_t>();
      e.valid_words = c.validity->as<uint64_t>();
      pending.push_back(out.cols.size());
      out.cols.push_back(std::move(c));
      continue;
    }
    emit_later(vf, out_name, 0, mode, a.lo, a.hi, a.seen, nullptr, 1, nullable);
  }
  if (eset.n > 0) {
    BufPtr stats = make_zero_buf((size_t)eset.n * 16);
    k_emit_set<<<dim3((unsigned)grid_for(G, BLOCK), (unsigned)eset.n), BLOCK, 0, r.stream>>>(eset, G, stats->as<unsigned long long>());
    DFGPU_HIP(hipGetLastError());
    std::vector<unsigned long long> h((size_t)eset.n * 2);
    d2h(h.data(), stats->ptr, h.size() * 8);   // (waits for the stream: the columns are complete)
    for (int e = 0; e < eset.n; e++) {
      DFGPU_CHECK(!h[(size_t)e * 2 + 1], "Arithmetic Overflow in AvgAccumulator");
      Column& c = out.cols[pending[(size_t)e]];
      if (!c.validity) continue;
      c.null_count = G - (int64_t)h[(size_t)e * 2];
      if (c.null_count == 0) c.validity.reset();
    }
  } else {
    DFGPU_HIP(hipStreamSynchronize(r.stream));
  }
  return out;
}

}  // namespace dfgpu

using namespace dfgpu;

extern "C" {

int dfgpu_agg_create(int mode, const dfgpu_expr* group_by, const char* const* group_names, int n_group, const dfgpu_agg_spec* aggs, int n_aggs,
                     dfgpu_agg_t* out) {
  return guarded([&] {
    require_init();
    DFGPU_CHECK(mode >= DFGPU_AGG_PARTIAL && mode <= DFGPU_AGG_PARTIAL_REDUCE, "bad aggregate mode");
    auto A = std::make_unique<Aggregate>();
    A->mode = mode;
    for (int g = 0; g < n_group; g++) {
      A->group_nodes.emplace_back(group_by[g].nodes, group_by[g].nodes + group_by[g].n_nodes);
      A->group_roots.push_back(group_by[g].root);
      A->group_names.push_back(group_names && group_names[g] ? group_names[g] : "");
    }
    for (int k = 0; k < n_aggs; k++) {
      AggState a;
      a.func = aggs[k].func;
      DFGPU_CHECK(a.func >= DFGPU_AGG_SUM && a.func <= DFGPU_AGG_AVG, "unsupported aggregate function");
      a.has_arg = aggs[k].has_arg != 0;
      DFGPU_CHECK(a.has_arg || a.func == DFGPU_AGG_COUNT, "only COUNT may omit its argument");
      if (a.has_arg && aggs[k].arg.nodes) {
        a.nodes.assign(aggs[k].arg.nodes, aggs[k].arg.nodes + aggs[k].arg.n_nodes);
        a.root = aggs[k].arg.root;
      }
      a.name = aggs[k].name ? aggs[k].name : "";
      a.ret = aggs[k].return_field;
      A->aggs.push_back(std::move(a));
    }
    *out = reinterpret_cast<dfgpu_agg_t>(A.release());
  });
}

// GROUPING SETS / CUBE / ROLLUP (PhysicalGroupBy with several groups, aggregates/mod.rs:400-520; merge_expressions / group_schema
// :700-760): the reference evaluates every grouping set's keys — expr where the set keeps the column, the typed NULL of `null_expr`
// where it does not, plus `__grouping_id` (bit n-1-i set = column i is NULLed out; UInt8 / 16 / 32 / 64 by the number of
// columns) — and interns them into ONE table.  Here every set is an aggregate of its own over the same input (their group spaces are
// disjoint: the grouping id differs), updated together and emitted set after set; output columns and types are the reference's.
int dfgpu_agg_create_grouping_sets(int mode, const dfgpu_expr* group_by, const dfgpu_expr* null_by, const char* const* group_names, int n_group,
                                   const uint8_t* groups, int n_sets, const dfgpu_agg_spec* aggs, int n_aggs, dfgpu_agg_t* out) {
  return guarded([&] {
    require_init();
    DFGPU_CHECK(mode == DFGPU_AGG_PARTIAL || mode == DFGPU_AGG_SINGLE || mode == DFGPU_AGG_SINGLE_PARTITIONED,
                "grouping sets: Final modes group by the partial state's key columns (the n_group keys and __grouping_id): use dfgpu_agg_create");
    DFGPU_CHECK(group_by && null_by && groups && n_group >= 1 && n_group <= 63 && n_sets >= 1 && out, "grouping sets: bad argument");
    auto top = std::make_unique<Aggregate>();
    top->mode = mode;
    // Aggregate::grouping_id_type: UInt8 / UInt16 / UInt32 / UInt64 by the number of grouping columns.  The device has no UInt16
    // column: 9..16 grouping columns would come out as UInt32 and a CPU Final node fed by this Partial one would see another schema
    // than the reference's — refused, the planner keeps the CPU operator (shim/src/operators.rs try_from_aggregate)
    DFGPU_CHECK(!(n_group >= 9 && n_group <= 16), "grouping sets over 9..16 columns: __grouping_id is UInt16 in the reference, which has no device type");
    const int id_type = n_group <= 8 ? DFGPU_UINT8 : n_group <= 32 ? DFGPU_UINT32 : DFGPU_UINT64;
    for (int s = 0; s < n_sets; s++) {
      std::vector<dfgpu_expr> keys((size_t)n_group + 1);
      std::vector<const char*> names((size_t)n_group + 1);
      uint64_t id = 0;
      for (int g = 0; g < n_group; g++) {
        const bool nulled = groups[(size_t)s * n_group + g] != 0;
        keys[(size_t)g] = nulled ? null_by[g] : group_by[g];
        if (nulled) {
          DFGPU_CHECK(null_by[g].n_nodes >= 1 && null_by[g].nodes[null_by[g].root].op == DFGPU_EXPR_LITERAL && null_by[g].nodes[null_by[g].root].is_null,
                      "grouping sets: null_by must hold typed NULL literals");
          id |= 1ull << (n_group - 1 - g);
        }
        names[(size_t)g] = group_names && group_names[g] ? group_names[g] : "";
      }
      dfgpu_expr_node idn{};
      idn.op = DFGPU_EXPR_LITERAL;
      idn.column = -1;
      idn.left = idn.right = -1;
      idn.field.type = id_type;
      idn.lit_lo = id;
      keys[(size_t)n_group] = dfgpu_expr{&idn, 1, 0, nullptr};
      names[(size_t)n_group] = "__grouping_id";
      dfgpu_agg_t sub = nullptr;
      if (dfgpu_agg_create(mode, keys.data(), names.data(), n_group + 1, aggs, n_aggs, &sub) != 0) throw Error(dfgpu_last_error());
      top->sets.emplace_back(reinterpret_cast<Aggregate*>(sub));
    }
    *out = reinterpret_cast<dfgpu_agg_t>(top.release());
  });
}

int dfgpu_agg_update(dfgpu_agg_t h, dfgpu_table_t input) {
  return guarded([&] {
    require_init();
    Aggregate* a = unwrap_agg(h);
    if (!a->sets.empty()) {
      for (auto& s : a->sets) agg_update(*s, agg_input(*s, input));
      return;
    }
    agg_update(*a, agg_input(*a, input));
  });
}

int dfgpu_agg_update_filtered(dfgpu_agg_t h, dfgpu_table_t input, const dfgpu_expr* predicate) {
  return guarded([&] {
    require_init();
    Aggregate* a = unwrap_agg(h);
    if (!a->sets.empty()) {
      for (auto& s : a->sets) agg_update(*s, agg_input(*s, input), predicate);
      return;
    }
    agg_update(*a, agg_input(*a, input), predicate);
  });
}

int dfgpu_agg_fused_updates(dfgpu_agg_t h, int64_t* out) {
  return guarded([&] { *out = reinterpret_cast<Aggregate*>(h)->fused_updates; });
}

int dfgpu_set_fusion(int on) {
  return guarded([&] { set_fusion_enabled(on != 0); });
}

int dfgpu_agg_emit(dfgpu_agg_t h, dfgpu_table_t* out) {
  return guarded([&] {
    require_init();
    Aggregate* a = unwrap_agg(h);
    if (!a->sets.empty()) {
      std::vector<std::unique_ptr<Table>> parts;
      std::vector<dfgpu_table_t> hs;
      for (auto& s : a->sets) {
        if (s->group_keys.device >= 0) use_device(s->group_keys.device);
        parts.push_back(std::make_unique<Table>(agg_emit(*s)));
        hs.push_back(wrap_quiet(parts.back().get()));
      }
      if (hs.size() == 1) {
        *out = wrap(parts[0].release());
        return;
      }
      if (dfgpu_table_concat(hs.data(), (int)hs.size(), out) != 0) throw Error(dfgpu_last_error());
      return;
    }
    auto t = std::make_unique<Table>(agg_emit(*a));
    *out = wrap(t.release());
  });
}

int dfgpu_agg_free(dfgpu_agg_t h) {
  return guarded([&] { if (h) delete unwrap_agg(h); });
}

}  // extern "C"
