// join.hip — K1..K5: HashJoinExec build + probe on device.
//
// Build (collect_left_input, physical-plan/src/joins/hash_join/exec.rs:2569-2776):
//   * direct-address table  = ArrayMap (joins/array_map.rs:103-236), chosen by the reference's
//     own gating try_create_array_map (exec.rs:111-191): one integer key, range < threshold or
//     rows/(range+1) > min density.  data[key-min] = row+1, duplicates chained through next[].
//   * rank map (GPU-native, no reference counterpart) = direct addressing compressed 16x: one bit per
//     key value of the range + an exclusive popcount directory per 64-bit word.  For UNIQUE integer
//     keys the rank of a present key is its position in key order; row id = rank when the build keys
//     arrive in ascending order (TPC-H primary keys, and anything filtered from them), else
//     perm[rank].  A 600 M-value range costs 150 MB (MALL-resident) instead of ArrayMap's 2.4 GB, so
//     the memset, the build atomics and the probe lookups stop streaming the table through HBM.
//     Falls back to ArrayMap / JoinHashMap when the build side has duplicate keys.
//   * chained hash table    = JoinHashMap (joins/join_hash_map.rs:144-338): head[hash & mask] =
//     row+1, next[row] = previous head.  Built with one atomicExch per row (no locks).
//   NULL keys are not inserted under NullEqualsNothing (joins/utils.rs:2127-2164).
// Probe (HashJoinStream::process_probe_batch, hash_join/stream.rs:740-1000), whole partition
//   per call instead of 8192-row batches:
//   pass 1: per probe row walk the chain, compare REAL keys (equal_rows_arr, joins/utils.rs:
//           2191-2260 — K4 fused), produce a per-row output count; one wave64 = 64 rows, its
//           total goes to a per-word count (ballot/popcount when counts are 0/1);
//   scan  : exclusive prefix of the per-word counts (scan.hip);
//   pass 2: emit.  Fast path (<=1 match per probe row: unique build keys, or semi/anti):
//           pass 1 stored the matched build row per probe row; pass 2 is a fused
//           compaction of the probe columns + gather of the build columns (K5) that writes
//           output rows in probe order.  General M:N path: (build_idx, probe_idx) pairs, then
//           arrow-`take`-style gathers (build_batch_from_indices, joins/utils.rs:1332-1386).
// Output order = probe order, then chain order (unordered by contract for duplicates).
#include <thread>
#include <functional>
#include <cstdlib>

#include "device.hpp"
#include "internal.hpp"
#include "grouped.hpp"

namespace dfgpu {

Table compact_table(const Table& in, const std::vector<int>& cols, const uint64_t* mask, const uint64_t* mask_valid);
void pack_bytes_to_bitmap(const uint8_t* bytes, int64_t n, uint64_t* words);

enum TableKind : int { KIND_HASH = 0, KIND_ARRAY = 1, KIND_RANK = 2, KIND_RADIX = 3, KIND_FLAT = 4, KIND_FLAT16 = 5,
                       KIND_RETURNED = 6 };   // (6 is not a table: the probe reads what a grouped lookup found for every probe row — join_probe)
template <int KIND> constexpr bool kind_is_flat() { return KIND == KIND_FLAT || KIND == KIND_FLAT16; }
template <int KIND> constexpr bool kind_is_direct() { return KIND == KIND_ARRAY || KIND == KIND_RANK; }

// Flat table (round 4): the general hash table with the KEYS INLINE.  JoinHashMap (joins/join_hash_map.rs:144-338) keeps
// (hash, first row) per slot and re-checks candidates on the key columns (equal_rows_arr, joins/utils.rs:2191-2260): on the GPU
// that is a chain of dependent random accesses per probe row — slot, next[], then one line per key column of the build side
// (profiles/r3_join_shapes_v2.md: two-column key, 1 M x 60 M rows: 5.5 ms per pass, 1.6 % of peak).  When the key columns of a row
// pack into 16 bytes (any mix of integer / date / Float64 / Decimal128 columns, plus one bit per nullable column under NULL == NULL)
// a slot holds the packed key itself and the first build row with that key: ONE 16- or 32-byte access answers "is the key there,
// and which row", equal packed keys ARE equal keys (no re-check), rows with the same key chain through next[] exactly as in
// JoinHashMap.  Open addressing, linear probing, load <= 0.5, slot = top bits of the hash (so that a range of hashes is a range
// of slots: what a grouped probe wants).
struct FlatLayout {
  int n;                       // key columns
  uint8_t off[MAX_KEYS];       // byte offset of column i inside the packed key
  uint8_t width[MAX_KEYS];     // its bytes
  int8_t null_bit[MAX_KEYS];   // bit of the packed key that says "column i is NULL" (NULL == NULL over a nullable build column), -1 = none
};
constexpr double RANK_MAP_MIN_KEY_DENSITY = 1.0 / 256.0;  // 64 B of bitmap + directory per build row at the limit

struct JoinTable {
  Table build;
  std::vector<int> key_cols;
  int null_equality = 0;
  bool array_map = false;
  int kind = 0;  // TableKind
  bool keys_unique = true;
  bool force_collisions = false;
  int probe_mode = 0;  // dfgpu_join_options.probe_mode
  BufPtr heads;  // u32: ArrayMap data[] or hash heads[]
  BufPtr next;   // u32 per build row (null when array_map && unique)
  uint64_t am_offset = 0, am_size = 0;
  uint64_t hash_mask = 0;
  BufPtr rank_bits, rank_prefix, rank_perm;  // rank map: u64 bitmap, u64 exclusive popcount prefix per word, optional u32 perm
  BufPtr rank_tab;  // the probe's view of the rank map: {bitmap word, prefix} interleaved, ONE 16-byte load per lookup; made by the
                    // first probe that looks ranks up row by row (ensure_rank_tab): membership passes and the listed emit read
                    // rank_bits / rank_prefix
  std::mutex tab_mu;
  // build keys not in ascending row order: rank != row id, a probe that needs BUILD ROWS (payload columns, visited marks, pairs)
  // goes through the rank -> row permutation — built on the first such probe (ensure_rank_perm), because a probe that only asks
  // "is the key there" (no build column in the output: SELECT l.k ... JOIN, semi / anti joins) never touches it
  bool rank_needs_perm = false;
  std::shared_ptr<RadixTable> radix;  // KIND_RADIX: build records partitioned for the LDS join (radix_join.hip)
  std::map<int, BufPtr> rank_payload;  // rank map over keys in no order: build column -> its copy in RANK order (ensure_rank_payload)
  std::map<std::vector<int>, BufPtr> rank_records;  // ... and, for a payload of <= 12 bytes, those columns as ONE 16-byte record per rank
  BufPtr flat;                        // KIND_FLAT / KIND_FLAT16: uint4 {key, first row + 1, 0} per slot (two uint4 for 16-byte keys)
  FlatLayout flat_layout{};
  int flat_shift = 0;                 // slot = hash >> flat_shift
  uint64_t flat_mask = 0;
  BufPtr visited;  // u8 per build row, lazily allocated
  std::mutex mu;   // a join table is probed by several threads at once (CollectLeft): `visited` is made once, `info` counts under it
  // HashJoinExec::null_aware (NOT IN semantics, single key column): what JoinLeftData shares between the probe
  // partitions in the reference (probe_side_has_null / probe_side_non_empty / build_side_has_null, exec.rs:195-240)
  bool null_aware = false;
  bool probe_side_has_null = false, probe_side_non_empty = false, build_side_has_null = false;
  dfgpu_join_info info{};
};

struct ProbeCtx {
  KeySet bkeys, pkeys;
  const uint32_t* heads;
  const uint32_t* next;
  const ulonglong2* rank_tab;   // KIND_RANK: .x = bitmap word (one bit per key value), .y = set bits in earlier words
  const uint64_t* rank_bits;    // KIND_RANK: the bitmap words alone (what "is the key there" asks: half the bytes of rank_tab per line)
  const uint64_t* rank_prefix;  // KIND_RANK: set bits in earlier words, one u64 per word (what rank_tab interleaves with the bits)
  const uint32_t* rank_perm;    // null when the build keys are in ascending order (row id == rank)
  uint64_t am_offset, am_size, hash_mask;
  int null_equals_null;
  int force_collisions;
  const uint64_t* row_mask;     // optional: probe rows whose bit is 0 do not exist (FilterExec fused below the probe side)
  const uint4* flat;            // KIND_FLAT / KIND_FLAT16
  FlatLayout flat_layout;
  int flat_shift;
  uint64_t flat_mask;
  // KIND_RETURNED: probe row p -> ret_dest[p] (~0: nothing looked up) -> a record of ret_R bytes whose first word is the match id
  const uint32_t* ret_dest;
  const uint8_t* ret_rec;
  int ret_R;
};

// A FilterExec fused below the probe side whose predicate is an AND of `column <op> literal` over fixed-width integer-like columns
// (Int32 / Date32 / Int64 / UInt8 / UInt32 — TPC-H's date and code filters): the counts pass of the selective probe evaluates it
// per row from the column itself instead of reading a mask that k_cmp wrote a moment ago (one launch, one 1-bit-per-row write and
// one read less; filter.rs:1339-1444 semantics: a NULL operand drops the row).
constexpr int ROWPRED_MAX = 2;
struct RowPred {
  const void* data[ROWPRED_MAX];
  const uint64_t* valid[ROWPRED_MAX];
  long long lit[ROWPRED_MAX];
  uint8_t width[ROWPRED_MAX], is_unsigned[ROWPRED_MAX], op[ROWPRED_MAX];   // op: rowpred_sel(DFGPU_EXPR_EQ .. DFGPU_EXPR_GE)
  int n;
};
// All W rows of one conjunct at once: the width is asked ONCE, outside the unrolled loop, so the W loads leave back to back (a switch
// per load put every load in a basic block of its own and an s_waitcnt vmcnt(0) behind it: the fused counts pass ran slower than k_cmp
// plus the mask-reading counts pass, 4.82 vs 4.66 ms on SF300's lineitem).  The comparison is branch-free: `sel` holds what the
// predicate says for v < lit, v == lit, v > lit (bits 0, 1, 2), made on the host from the operator.
template <bool NT, int W>
__device__ __forceinline__ void rowpred_load_words(const RowPred& q, int i, int64_t w0, int64_t np, unsigned lane, long long (&v)[W]) {
  const int width = q.width[i];
  if (width == 8) {
    const long long* d = reinterpret_cast<const long long*>(q.data[i]);
#pragma unroll
    for (int j = 0; j < W; j++) {
      const int64_t p = ((w0 + j) << 6) + lane;
      v[j] = NT ? __builtin_nontemporal_load(d + (p < np ? p : np - 1)) : d[p < np ? p : np - 1];
    }
  } else if (width == 4) {
    const uint32_t* d = reinterpret_cast<const uint32_t*>(q.data[i]);
    uint32_t u[W];
#pragma unroll
    for (int j = 0; j < W; j++) {
      const int64_t p = ((w0 + j) << 6) + lane;
      u[j] = NT ? __builtin_nontemporal_load(d + (p < np ? p : np - 1)) : d[p < np ? p : np - 1];
    }
    const long long keep = q.is_unsigned[i] ? 0xFFFFFFFFll : -1ll;   // uniform: zero- or sign-extension
#pragma unroll
    for (int j = 0; j < W; j++) v[j] = (long long)(int32_t)u[j] & keep;
  } else {
    const uint8_t* d = reinterpret_cast<const uint8_t*>(q.data[i]);
#pragma unroll
    for (int j = 0; j < W; j++) {
      const int64_t p = ((w0 + j) << 6) + lane;
      v[j] = (long long)(NT ? __builtin_nontemporal_load(d + (p < np ? p : np - 1)) : d[p < np ? p : np - 1]);
    }
  }
}
__device__ __forceinline__ bool rowpred_cmp(unsigned sel, long long v, long long lit) {
  return ((v < lit ? sel : v == lit ? sel >> 1 : sel >> 2) & 1u) != 0;
}
static unsigned rowpred_sel(int op) {   // bit 0: holds for v < lit, bit 1: v == lit, bit 2: v > lit
  switch (op) {
    case DFGPU_EXPR_EQ: return 2u;
    case DFGPU_EXPR_NE: return 5u;
    case DFGPU_EXPR_LT: return 1u;
    case DFGPU_EXPR_LE: return 3u;
    case DFGPU_EXPR_GT: return 4u;
    default: return 6u;   // GE
  }
}

static KeySet make_keyset(const Table& t, const std::vector<int>& cols) {
  KeySet ks{};
  DFGPU_CHECK((int)cols.size() <= MAX_KEYS, "too many join key columns");
  ks.n = (int)cols.size();
  for (int i = 0; i < ks.n; i++) {
    DFGPU_CHECK(cols[i] >= 0 && cols[i] < (int)t.cols.size(), "join key column index out of range");
    const Column& c = t.cols[cols[i]];
    DFGPU_CHECK(c.field.type != DFGPU_BOOL, "Boolean join keys are not supported on the GPU path");
    ks.c[i] = KeyCol{c.ptr(), c.valid_words(), c.field.type, type_width(c.field.type)};
  }
  return ks;
}


// ------------------------------------------------------------------------ build kernels
struct MinMax {
  long long smin, smax;
  unsigned long long valid;
  unsigned unsorted;  // set when some key is <= its predecessor (or a NULL key exists): keys are not strictly ascending
  unsigned descends;  // set when some key is < its predecessor (or a NULL key exists): keys are not in non-decreasing order
};
constexpr int BUILD_UNROLL = 4;  // independent key loads in flight per thread
// min / max / valid count of the build key (ArrayMap::try_new bounds, array_map.rs:175-203) and whether
// the keys are strictly ascending in row order (=> unique, and rank == row id for the rank map).
// One workgroup reduces through LDS and issues ONE set of atomics: per-wave atomics on a single
// line serialise at ~12 ns each and tripled this kernel's time.
template <int KT, bool HASV>
__global__ __launch_bounds__(BLOCK) void k_key_minmax(KeyCol k, int64_t n, MinMax* out) {
  long long mn = INT64_MAX, mx = INT64_MIN;
  unsigned long long cnt = 0;
  bool unsorted = false, descends = false;
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  for (int64_t i0 = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i0 < n; i0 += stride * BUILD_UNROLL) {
    uint64_t lo[BUILD_UNROLL], plo[BUILD_UNROLL];
    bool ok[BUILD_UNROLL], pok[BUILD_UNROLL];
#pragma unroll
    for (int j = 0; j < BUILD_UNROLL; j++) {  // unconditional (clamped) loads: all 2 x BUILD_UNROLL in flight together
      int64_t i = i0 + j * stride;
      int64_t ic = i < n ? i : n - 1;
      lo[j] = load_key<KT>(k, ic);
      plo[j] = load_key<KT>(k, ic > 0 ? ic - 1 : 0);  // neighbour's line: an L1/L2 hit
    }
#pragma unroll
    for (int j = 0; j < BUILD_UNROLL; j++) {
      int64_t i = i0 + j * stride;
      ok[j] = i < n;
      pok[j] = i < n && i > 0;
      if (HASV) {
        ok[j] = ok[j] && bit_at(k.valid, i < n ? i : n - 1);
        pok[j] = pok[j] && bit_at(k.valid, (i < n ? i : n - 1) > 0 ? (i < n ? i : n - 1) - 1 : 0);
      }
    }
#pragma unroll
    for (int j = 0; j < BUILD_UNROLL; j++) {
      int64_t i = i0 + j * stride;
      if (i < n && !ok[j]) unsorted = descends = true;  // NULL key
      if (!ok[j]) continue;
      long long v = (long long)lo[j];
      mn = v < mn ? v : mn;
      mx = v > mx ? v : mx;
      cnt++;
      if (pok[j] && (long long)plo[j] >= v) unsorted = true;
      if (pok[j] && (long long)plo[j] > v) descends = true;
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    long long omn = __shfl_xor(mn, d, 64), omx = __shfl_xor(mx, d, 64);
    mn = omn < mn ? omn : mn;
    mx = omx > mx ? omx : mx;
    cnt += __shfl_xor(cnt, d, 64);
  }
  const unsigned wave_unsorted = (ballot64(unsorted) != 0 ? 1u : 0u) | (ballot64(descends) != 0 ? 2u : 0u);
  __shared__ long long s_mn[BLOCK / WAVE], s_mx[BLOCK / WAVE];
  __shared__ unsigned long long s_cnt[BLOCK / WAVE];
  __shared__ unsigned s_uns[BLOCK / WAVE];
  const int wv = threadIdx.x >> 6;
  if (lane_id() == 0) { s_mn[wv] = mn; s_mx[wv] = mx; s_cnt[wv] = cnt; s_uns[wv] = wave_unsorted; }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned uns = 0;
    for (int i = 1; i < BLOCK / WAVE; i++) {
      mn = s_mn[i] < mn ? s_mn[i] : mn;
      mx = s_mx[i] > mx ? s_mx[i] : mx;
      cnt += s_cnt[i];
    }
    for (int i = 0; i < BLOCK / WAVE; i++) uns |= s_uns[i];
    if (cnt) {
      atomicMin(&out->smin, mn);
      atomicMax(&out->smax, mx);
      atomicAdd(&out->valid, cnt);
    }
    if (uns & 1u) atomicOr(&out->unsorted, 1u);
    if (uns & 2u) atomicOr(&out->descends, 1u);
  }
}

// first key, last key, and whether `samples` evenly spaced neighbour pairs all ascend strictly: what the build speculates on
template <int KT>
__global__ void k_key_ends(KeyCol k, int64_t n, int64_t every, int samples, long long* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) {
    out[0] = (long long)load_key<KT>(k, 0);
    out[1] = (long long)load_key<KT>(k, n - 1);
  }
  if (i >= samples) return;
  const int64_t r = (int64_t)i * every;
  if (r + 1 >= n) return;
  if ((long long)load_key<KT>(k, r + 1) <= (long long)load_key<KT>(k, r)) atomicOr((unsigned long long*)&out[2], 1ull);
}

// rank map build, step 1: one bit per present key value.  Strictly ascending keys are unique by
// construction; otherwise the returned old word detects duplicates.
// VERIFY (ASCENDING only): min / max / order are the caller's GUESS (first key, last key, a sample of neighbours) — every key is
// checked against its predecessor and the guessed range on the way, a violation raises dup_flag (and sets no bit out of range);
// the caller then throws the table away and builds from measured statistics
template <int KT, bool HASV, bool ASCENDING, bool VERIFY = false>
__global__ __launch_bounds__(BLOCK) void k_rank_setbits(KeyCol k, int64_t n, uint64_t offset, unsigned long long* __restrict__ bits, int* dup_flag, uint64_t range = 0) {
  if (ASCENDING) {
    // a wave holds 64 consecutive rows => ascending keys: the lanes that share a bitmap word are
    // contiguous and carry distinct bits, so the word a segment sets is the SUM of its lanes' bits =
    // S[last lane] - S[lane before the segment] of one wave prefix sum (mod 2^64, done with DPP).
    // The last lane of each segment issues ONE fire-and-forget atomic (TPC-H orderkeys: 4 per wave
    // instead of 64; 150 M -> 9.4 M atomics at SF100).
    const unsigned lane = lane_id();
    const int64_t stride = (int64_t)gridDim.x * BLOCK;
    for (int64_t base = (int64_t)blockIdx.x * BLOCK + (threadIdx.x & ~63u); base < n; base += stride * BUILD_UNROLL) {
      uint64_t idx[BUILD_UNROLL], pidx[BUILD_UNROLL];
#pragma unroll
      for (int j = 0; j < BUILD_UNROLL; j++) {
        int64_t i = base + j * stride + lane;
        idx[j] = load_key<KT, true>(k, i < n ? i : n - 1) - offset;   // (the keys stream by once: non-temporal)
        if (VERIFY) pidx[j] = load_key<KT>(k, i < n ? (i > 0 ? i - 1 : 0) : n - 1) - offset;   // the neighbour's line: a cache hit, in flight with the rest
      }
#pragma unroll
      for (int j = 0; j < BUILD_UNROLL; j++) {
        const int64_t i = base + j * stride + lane;
        bool ok = i < n;
        if (VERIFY) {
          const bool bad = ok && (idx[j] > range || (i > 0 && idx[j] <= pidx[j]));
          if (ballot64(bad) != 0 && lane == 0) atomicOr(dup_flag, 1);
          ok = ok && idx[j] <= range;
        }
        const uint64_t w = ok ? (idx[j] >> 6) : ~0ull;  // the ragged tail forms its own (ignored) segment
        const uint64_t v = ok ? 1ull << (idx[j] & 63) : 0ull;
        const uint64_t inc = wave_inclusive_sum_dpp(v);
        const uint64_t w_next = __shfl_down(w, 1, 64);
        const uint64_t tails = ballot64(lane == 63 || w_next != w);
        const uint64_t heads = (tails << 1) | 1ull;
        const int head_lane = 63 - __builtin_clzll(heads & ((2ull << lane) - 1ull));  // start of this lane's segment
        const uint64_t before = __shfl(inc - v, head_lane, 64);                          // prefix sum ahead of the segment
        // (plain stores for the segments that touch neither end of the wave — their words belong to this wave alone — were measured:
        // 0.40 -> 1.41 ms; scattered 8-byte stores cost more than the fire-and-forget atomics they replace.  So was a tile flavour —
        // 1024 consecutive keys per workgroup, their words assembled in LDS and written with coalesced stores, 2 global atomics per
        // tile —: 0.58 ms; the LDS atomics and the two barriers per tile cost more than the 62 global atomics they save)
        if (ok && ((tails >> lane) & 1ull)) atomicOr(&bits[w], (unsigned long long)(inc - before));
      }
    }
    return;
  }
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  for (int64_t i0 = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i0 < n; i0 += stride * BUILD_UNROLL) {
    uint64_t lo[BUILD_UNROLL];
    bool ok[BUILD_UNROLL];
#pragma unroll
    for (int j = 0; j < BUILD_UNROLL; j++) {
      int64_t i = i0 + j * stride;
      lo[j] = load_key<KT>(k, i < n ? i : n - 1);
    }
#pragma unroll
    for (int j = 0; j < BUILD_UNROLL; j++) {
      int64_t i = i0 + j * stride;
      ok[j] = i < n;
      if (HASV) ok[j] = ok[j] && bit_at(k.valid, i < n ? i : n - 1);
    }
#pragma unroll
    for (int j = 0; j < BUILD_UNROLL; j++) {
      if (!ok[j]) continue;
      const uint64_t idx = lo[j] - offset;
      const unsigned long long bit = 1ull << (idx & 63);
      // fire-and-forget: nothing waits for the old word (a returning atomic per key was what 150 M keys in no order spent their
      // 5.7 ms on); duplicate keys show afterwards as fewer set bits than non-NULL keys (join_build_fixed_keys)
      (void)__hip_atomic_fetch_or(&bits[idx >> 6], bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
// rank map build over keys in no order, without atomics: one BYTE per value of the range is set by a plain store (racing stores all
// write 1; an agent-scope atomic per key — even fire-and-forget — retires at ~27 G/s on this part: 150 M keys = 5.6 ms), then the
// bytes are packed to the bitmap (pack_bytes_to_bitmap: one ballot per 64 values)
template <int KT, bool HASV>
__global__ __launch_bounds__(BLOCK) void k_rank_setbytes(KeyCol k, int64_t n, uint64_t offset, uint8_t* __restrict__ bytes) {
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  for (int64_t i0 = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i0 < n; i0 += stride * BUILD_UNROLL) {
    uint64_t lo[BUILD_UNROLL];
#pragma unroll
    for (int j = 0; j < BUILD_UNROLL; j++) {
      const int64_t i = i0 + j * stride;
      lo[j] = load_key<KT>(k, i < n ? i : n - 1);
    }
#pragma unroll
    for (int j = 0; j < BUILD_UNROLL; j++) {
      const int64_t i = i0 + j * stride;
      if (i >= n) continue;
      if (HASV && !bit_at(k.valid, i)) continue;
      bytes[lo[j] - offset] = 1;
    }
  }
}
// rank map build: {bitmap word, prefix} side by side for the probe
// Rank map over MANY keys in no order (round 4): the build keys are grouped by their position in the key range first
// (grouped.hip: 1024 groups), then ONE workgroup per group sets the group's bits in an LDS bitmap — a group spans at most 2^20
// key values = 128 KB of bits — and writes the words out: interior words with plain coalesced stores, the two words it may share
// with its neighbour groups with an atomicOr.  No byte map (one byte per VALUE of the range), no random global atomics; a bit
// that was already set is a duplicate key (150 M shuffled keys: 5.4 + 0.7 ms of byte map + pack -> see profiles/r4_join_shapes.md).
constexpr int RB_THREADS = 1024;
__global__ __launch_bounds__(RB_THREADS) void k_rank_bits_grouped(const uint64_t* __restrict__ gkeys, const uint64_t* __restrict__ bounds, const uint64_t* __restrict__ gfirst,
                                                                  uint64_t offset, unsigned long long* __restrict__ bits, int* __restrict__ dup_flag) {
  extern __shared__ unsigned long long rb_words[];
  const int g = blockIdx.x;
  const uint64_t v0 = gfirst[g], v1 = gfirst[g + 1];   // the group's values [v0, v1) of the key range
  if (v1 <= v0) return;
  const uint64_t w0 = v0 >> 6, w1 = (v1 - 1) >> 6;      // its bitmap words [w0, w1]
  const int nw = (int)(w1 - w0 + 1);
  for (int i = threadIdx.x; i < nw; i += RB_THREADS) rb_words[i] = 0ull;
  __syncthreads();
  const int64_t r0 = (int64_t)bounds[g], r1 = (int64_t)bounds[g + 1];
  bool dup = false;
  for (int64_t i = r0 + threadIdx.x; i < r1; i += RB_THREADS) {
    const uint64_t idx = gkeys[i] - offset;
    const unsigned long long bit = 1ull << (idx & 63);
    const unsigned long long old = atomicOr(&rb_words[(idx >> 6) - w0], bit);
    dup |= (old & bit) != 0;
  }
  if (dup) *dup_flag = 1;
  __syncthreads();
  for (int i = threadIdx.x; i < nw; i += RB_THREADS) {
    const unsigned long long w = rb_words[i];
    if (i == 0 || i == nw - 1) {
      if (w) atomicOr(&bits[w0 + i], w);   // may be shared with the neighbouring group
    } else {
      bits[w0 + i] = w;
    }
  }
}
__global__ __launch_bounds__(BLOCK) void k_rank_interleave(const uint64_t* __restrict__ bits, const uint64_t* __restrict__ prefix, int64_t n_words, ulonglong2* __restrict__ tab) {
  for (int64_t w = (int64_t)blockIdx.x * BLOCK + threadIdx.x; w < n_words; w += (int64_t)gridDim.x * BLOCK) tab[w] = make_ulonglong2(bits[w], prefix[w]);
}
// rank map build, step 2 (keys not in ascending row order): perm[rank(key_i)] = i, read off the interleaved table
__global__ __launch_bounds__(BLOCK) void k_rank_perm_tab(KeyCol k, int64_t n, uint64_t offset, const ulonglong2* __restrict__ tab, uint32_t* __restrict__ perm) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    if (k.valid && !bit_at(k.valid, i)) continue;
    uint64_t lo, hi;
    load_words(k, i, lo, hi);
    const uint64_t idx = lo - offset;
    const ulonglong2 e = tab[idx >> 6];
    perm[(uint32_t)e.y + (uint32_t)__popcll(e.x & ((1ull << (idx & 63)) - 1ull))] = (uint32_t)i;
  }
}

// ArrayMap::fill_data (array_map.rs:205-236), lock-free: data[key-min] <- row+1, the previous
// occupant becomes next[row] (chain order is arbitrary; the reference's is ascending).
template <int KT, bool HASV>
__global__ __launch_bounds__(BLOCK) void k_am_build(KeyCol k, int64_t n, uint64_t offset, uint32_t* data, uint32_t* next, int* dup_flag) {
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  for (int64_t i0 = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i0 < n; i0 += stride * BUILD_UNROLL) {
    uint64_t lo[BUILD_UNROLL];
    bool ok[BUILD_UNROLL];
#pragma unroll
    for (int j = 0; j < BUILD_UNROLL; j++) {
      int64_t i = i0 + j * stride;
      lo[j] = load_key<KT>(k, i < n ? i : n - 1);
    }
#pragma unroll
    for (int j = 0; j < BUILD_UNROLL; j++) {
      int64_t i = i0 + j * stride;
      ok[j] = i < n;
      if (HASV) ok[j] = ok[j] && bit_at(k.valid, i < n ? i : n - 1);
    }
#pragma unroll
    for (int j = 0; j < BUILD_UNROLL; j++) {
      if (!ok[j]) continue;
      int64_t i = i0 + j * stride;
      uint32_t old = atomicExch(&data[lo[j] - offset], (uint32_t)i + 1u);
      if (old) {
        if (next) next[i] = old;
        else *dup_flag = 1;
      }
    }
  }
}

// JoinHashMap::update_from_iter (join_hash_map.rs:307-338), lock-free
__global__ __launch_bounds__(BLOCK) void k_hm_build(KeySet ks, int64_t n, uint64_t mask, int null_equals_null, int force_collisions,
                                                    uint32_t* heads, uint32_t* next) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    bool any_null;
    uint64_t h = hash_row(ks, i, SEED_JOIN, any_null);
    if (any_null && !null_equals_null) continue;
    if (force_collisions) h = 0;
    uint32_t old = atomicExch(&heads[h & mask], (uint32_t)i + 1u);
    next[i] = old;
  }
}
// are the build KEYS unique?  (lets the probe stop at the first match)
__global__ __launch_bounds__(BLOCK) void k_hm_check_unique(KeySet ks, int64_t n, int null_equals_null, const uint32_t* next, int* dup_flag) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    uint32_t cur = next[i];
    while (cur) {
      if (keys_equal(ks, i, ks, (int64_t)cur - 1, null_equals_null, true)) { *dup_flag = 1; break; }
      cur = next[cur - 1];
    }
  }
}

// Are there duplicate keys?  A sample answers "yes" cheaply: m evenly spaced rows AND their successors go into an open-addressing table in
// HBM; a key met twice raises the flag.  (A key that appears k times among n rows is met twice with probability ~ m^2 (k - 1) / 2n per key
// population: 256 K samples of 150 M rows holding every key three times meet several hundred pairs; runs of equal neighbours are met
// through the successors.)  "No" proves nothing: the caller then finds out the usual way.
template <int KT>
__global__ __launch_bounds__(BLOCK) void k_sample_duplicates(KeyCol k, int64_t n, int64_t every, int64_t m, unsigned long long* __restrict__ table, uint64_t mask,
                                                             int* __restrict__ found) {
  for (int64_t j = (int64_t)blockIdx.x * BLOCK + threadIdx.x; j < 2 * m; j += (int64_t)gridDim.x * BLOCK) {
    const int64_t i = (j >> 1) * every + (j & 1);
    if (i >= n || (k.valid && !bit_at(k.valid, i))) continue;
    const unsigned long long key = load_key<KT>(k, i) + 1ull;   // (0 = empty slot)
    if (key == 0ull) continue;
    uint64_t s = fmix64(key) & mask;
    for (int step = 0; step < 64; step++) {
      const unsigned long long old = atomicCAS(&table[s], 0ull, key);
      if (old == 0ull) break;
      if (old == key) {
        *found = 1;
        break;
      }
      s = (s + 1) & mask;
    }
  }
}

// ------------------------------------------------------------------------ flat table (keys inline)
// the key columns of row i as one packed key of <= 16 bytes; false = the row has a NULL key that matches nothing
__device__ __forceinline__ bool flat_pack(const KeySet& ks, const FlatLayout& L, int64_t i, bool null_equals_null, uint64_t& k0, uint64_t& k1) {
  if (ks.n == 1 && ks.c[0].width == 8 && !ks.c[0].valid) {   // the common single Int64 / Float64 key: its bits are the packed key
    k0 = ((const uint64_t*)ks.c[0].data)[i];
    k1 = 0;
    return true;
  }
  u128 k = 0;
  for (int c = 0; c < ks.n; c++) {
    const KeyCol& kc = ks.c[c];
    if (kc.valid && !bit_at(kc.valid, i)) {
      if (!null_equals_null || L.null_bit[c] < 0) return false;  // (no NULL on the build side of this column: nothing to equal)
      k |= (u128)1 << L.null_bit[c];
      continue;
    }
    uint64_t lo, hi;
    load_words(kc, i, lo, hi);
    if (kc.type == 4) lo = ((const uint64_t*)kc.data)[i];   // Float64 keys are equal when their bits are (arrow-ord eq = totalOrder: -0.0 != +0.0)
    u128 v = ((u128)hi << 64) | lo;
    if (L.width[c] < 16) v &= ((u128)1 << (8 * L.width[c])) - 1;   // sign extension of the narrow types: both sides drop it alike
    k |= v << (8 * L.off[c]);
  }
  k0 = (uint64_t)k;
  k1 = (uint64_t)(k >> 64);
  return true;
}
template <bool WIDE>
__device__ __forceinline__ uint64_t flat_hash(uint64_t k0, uint64_t k1, bool force_collisions) {
  if (force_collisions) return 0;
  uint64_t h = hash_u64(k0, SEED_JOIN);
  if (WIDE) h = fmix64(k1 ^ h);
  return h;
}
template <bool WIDE>
__device__ __forceinline__ bool flat_slot_is(const uint4* __restrict__ tab, uint64_t s, uint64_t k0, uint64_t k1, uint32_t& head, uint32_t* rows = nullptr) {
  if (!WIDE) {
    const uint4 e = tab[s];
    head = e.z;
    if (rows) *rows = e.w;
    return e.x == (uint32_t)k0 && e.y == (uint32_t)(k0 >> 32);
  }
  const uint4 a = tab[2 * s], b = tab[2 * s + 1];
  head = b.x;
  if (rows) *rows = b.y;
  return a.x == (uint32_t)k0 && a.y == (uint32_t)(k0 >> 32) && a.z == (uint32_t)k1 && a.w == (uint32_t)(k1 >> 32);
}
// first build row + 1 with this key, 0 = none; `rows` (optional) = how many build rows carry the key (the slot keeps the count,
// so a pass that only COUNTS matches never walks next[])
template <bool WIDE>
__device__ __forceinline__ uint32_t flat_find(const uint4* __restrict__ tab, int shift, uint64_t mask, uint64_t k0, uint64_t k1, bool force_collisions, uint32_t* rows = nullptr) {
  uint64_t s = flat_hash<WIDE>(k0, k1, force_collisions) >> shift;
  for (;;) {
    uint32_t head;
    const bool same = flat_slot_is<WIDE>(tab, s, k0, k1, head, rows);
    if (head == 0) {   // an empty slot ends the run of its hash neighbourhood
      if (rows) *rows = 0;
      return 0u;
    }
    if (same) return head;
    s = (s + 1) & mask;
  }
}
// build, pass 1: every key claims ONE slot with the row that gets there first (a 4-byte CAS; the slot's key is read from that
// row's key columns — immutable input — so nobody ever waits for another thread's stores); later rows with the same key push
// themselves behind the owner: next[owner] -> newest -> ... -> 0 (LIFO like JoinHashMap's chains; order among equal keys is
// unobserved by contract)
template <bool WIDE>
__global__ __launch_bounds__(BLOCK) void k_flat_claim(KeySet ks, FlatLayout L, int64_t n, int null_equals_null, int force_collisions, int shift, uint64_t mask,
                                                      uint32_t* __restrict__ owner, uint32_t* __restrict__ next, int* __restrict__ dup_flag, uint32_t* __restrict__ more) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    uint64_t k0, k1;
    if (!flat_pack(ks, L, i, null_equals_null != 0, k0, k1)) continue;
    uint64_t s = flat_hash<WIDE>(k0, k1, force_collisions != 0) >> shift;
    for (;;) {
      uint32_t o = __hip_atomic_load(&owner[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (o == 0) {
        o = atomicCAS(&owner[s], 0u, (uint32_t)i + 1u);
        if (o == 0) break;  // ours
      }
      uint64_t q0, q1;
      flat_pack(ks, L, (int64_t)o - 1, null_equals_null != 0, q0, q1);
      if (q0 == k0 && (!WIDE || q1 == k1)) {
        next[i] = atomicExch(&next[o - 1], (uint32_t)i + 1u);
        atomicAdd(&more[o - 1], 1u);   // rows behind the owner
        *dup_flag = 1;
        break;
      }
      s = (s + 1) & mask;
    }
  }
}
// build, pass 2: the owners' keys move into the slots
template <bool WIDE>
__global__ __launch_bounds__(BLOCK) void k_flat_fill(KeySet ks, FlatLayout L, int64_t cap, int null_equals_null, const uint32_t* __restrict__ owner,
                                                     const uint32_t* __restrict__ more, uint4* __restrict__ tab) {
  for (int64_t s = (int64_t)blockIdx.x * BLOCK + threadIdx.x; s < cap; s += (int64_t)gridDim.x * BLOCK) {
    const uint32_t o = owner[s];
    uint64_t k0 = 0, k1 = 0;
    if (o) flat_pack(ks, L, (int64_t)o - 1, null_equals_null != 0, k0, k1);
    const uint32_t rows = o ? more[o - 1] + 1u : 0u;   // build rows with this key
    if (!WIDE) {
      tab[s] = make_uint4((uint32_t)k0, (uint32_t)(k0 >> 32), o, rows);
    } else {
      tab[2 * s] = make_uint4((uint32_t)k0, (uint32_t)(k0 >> 32), (uint32_t)k1, (uint32_t)(k1 >> 32));
      tab[2 * s + 1] = make_uint4(o, rows, 0u, 0u);
    }
  }
}

// ------------------------------------------------------------------------ probe kernels
template <int KIND>
__device__ __forceinline__ uint32_t chain_head(const ProbeCtx& c, int64_t p) {
  if (kind_is_flat<KIND>()) {
    uint64_t k0, k1;
    if (!flat_pack(c.pkeys, c.flat_layout, p, c.null_equals_null != 0, k0, k1)) return 0u;
    return flat_find<KIND == KIND_FLAT16>(c.flat, c.flat_shift, c.flat_mask, k0, k1, c.force_collisions != 0);
  } else if (KIND != KIND_HASH) {
    const KeyCol& k = c.pkeys.c[0];
    if (k.valid && !bit_at(k.valid, p)) return 0;  // direct-address tables are never built with NULL==NULL + NULL build keys
    uint64_t lo, hi;
    load_words(k, p, lo, hi);
    uint64_t idx = lo - c.am_offset;
    if (idx >= c.am_size) return 0u;
    if (KIND == KIND_ARRAY) return c.heads[idx];
    const ulonglong2 e = c.rank_tab[idx >> 6];
    const uint64_t bits = e.x;
    if (!((bits >> (idx & 63)) & 1ull)) return 0u;
    const uint32_t rank = (uint32_t)e.y + (uint32_t)__popcll(bits & ((1ull << (idx & 63)) - 1ull));
    return (c.rank_perm ? c.rank_perm[rank] : rank) + 1u;
  } else {
    bool any_null;
    uint64_t h = hash_row(c.pkeys, p, SEED_JOIN, any_null);
    if (any_null && !c.null_equals_null) return 0;
    if (c.force_collisions) h = 0;
    return c.heads[h & c.hash_mask];
  }
}
template <int KIND>
__device__ __forceinline__ bool chain_match(const ProbeCtx& c, int64_t b, int64_t p) {
  if (KIND != KIND_HASH) return true;  // direct addressing / inline keys: same slot <=> same key
  return keys_equal(c.bkeys, b, c.pkeys, p, c.null_equals_null, true);
}

// match ids (build row + 1, 0 = none) of N consecutive 64-row probe words for this lane.  Every
// stage (keys -> validity -> table words -> perm) issues its N loads together; out-of-range lanes
// load a clamped address and are masked afterwards, so there is no branch between the loads.
template <int KIND, int KT, int N>
__device__ __forceinline__ void lookup_words(const ProbeCtx& c, int64_t w0, int64_t np, uint32_t (&m)[N], uint64_t* raw_keys = nullptr, uint4* rec16 = nullptr,
                                             uint32_t* key_rows = nullptr /* flat kinds: build rows per matched key */) {
  const unsigned lane = lane_id();
  if (KIND == KIND_RETURNED) {
    uint32_t d[N];
    bool ok[N];
#pragma unroll
    for (int j = 0; j < N; j++) {
      const int64_t p = ((w0 + j) << 6) + lane;
      ok[j] = p < np;
      d[j] = c.ret_dest[ok[j] ? p : np - 1];
      if (c.row_mask) ok[j] = ok[j] && ((c.row_mask[w0 + j] >> lane) & 1ull);
    }
    if (rec16 && c.ret_R == 16) {  // 16-byte records: match id and build columns in ONE access, kept in registers until the row is written
#pragma unroll
      for (int j = 0; j < N; j++) {
        ok[j] = ok[j] && d[j] != 0xFFFFFFFFu;
        rec16[j] = reinterpret_cast<const uint4*>(c.ret_rec)[ok[j] ? d[j] : 0u];
      }
#pragma unroll
      for (int j = 0; j < N; j++) m[j] = ok[j] ? rec16[j].x : 0u;
      return;
    }
#pragma unroll
    for (int j = 0; j < N; j++) {
      ok[j] = ok[j] && d[j] != 0xFFFFFFFFu;
      m[j] = *reinterpret_cast<const uint32_t*>(c.ret_rec + (uint64_t)(ok[j] ? d[j] : 0u) * (uint64_t)c.ret_R);
    }
#pragma unroll
    for (int j = 0; j < N; j++) m[j] = ok[j] ? m[j] : 0u;
    return;
  }
  if (kind_is_flat<KIND>()) {
    // the packed keys of all N words first (their column loads in flight together), then the N first slots, then whoever did not
    // settle on its first slot walks on
    constexpr bool WIDE = KIND == KIND_FLAT16;
    uint64_t k0[N], k1[N], s[N];
    bool live[N];
#pragma unroll
    for (int j = 0; j < N; j++) {
      const int64_t p = ((w0 + j) << 6) + lane;
      live[j] = p < np && (!c.row_mask || ((c.row_mask[w0 + j] >> lane) & 1ull));
      k0[j] = k1[j] = 0;
      live[j] = live[j] && flat_pack(c.pkeys, c.flat_layout, p, c.null_equals_null != 0, k0[j], k1[j]);
      s[j] = flat_hash<WIDE>(k0[j], k1[j], c.force_collisions != 0) >> c.flat_shift;
    }
    uint32_t head[N], rws[N];
    bool same[N];
#pragma unroll
    for (int j = 0; j < N; j++) same[j] = flat_slot_is<WIDE>(c.flat, s[j], k0[j], k1[j], head[j], &rws[j]);
#pragma unroll
    for (int j = 0; j < N; j++) {
      uint64_t ss = s[j];
      uint32_t hd = head[j], rw = rws[j];
      bool sm = same[j];
      while (live[j] && hd != 0 && !sm) {
        ss = (ss + 1) & c.flat_mask;
        sm = flat_slot_is<WIDE>(c.flat, ss, k0[j], k1[j], hd, &rw);
      }
      m[j] = live[j] && sm ? hd : 0u;   // (hd == 0 with sm: an empty slot compared equal to an all-zero key)
      if (key_rows) key_rows[j] = m[j] ? rw : 0u;
    }
    return;
  }
  if (KIND == KIND_HASH) {
#pragma unroll
    for (int j = 0; j < N; j++) {
      int64_t p = ((w0 + j) << 6) + lane;
      const bool live = p < np && (!c.row_mask || ((c.row_mask[w0 + j] >> lane) & 1ull));
      m[j] = live ? chain_head<KIND>(c, p) : 0u;
    }
#pragma unroll
    for (int j = 0; j < N; j++) {
      int64_t p = ((w0 + j) << 6) + lane;
      while (m[j] && !chain_match<KIND>(c, (int64_t)m[j] - 1, p)) m[j] = c.next[m[j] - 1];
    }
    return;
  }
  const KeyCol& k = c.pkeys.c[0];
  uint64_t idx[N];
  bool ok[N];
#pragma unroll
  for (int j = 0; j < N; j++) {
    int64_t p = ((w0 + j) << 6) + lane;
    ok[j] = p < np;
    idx[j] = load_key<KT>(k, ok[j] ? p : np - 1);
    if (raw_keys) raw_keys[j] = idx[j];  // the caller writes the key column of the output from registers (no second read)
    idx[j] -= c.am_offset;
    if (c.row_mask) ok[j] = ok[j] && ((c.row_mask[w0 + j] >> lane) & 1ull);  // filtered-out rows skip the table lookups
  }
  if (k.valid) {  // direct-address tables are never built with NULL==NULL + NULL build keys: a NULL probe key matches nothing
    uint64_t vw[N];
#pragma unroll
    for (int j = 0; j < N; j++) {
      int64_t p = ((w0 + j) << 6) + lane;
      vw[j] = k.valid[(p < np ? p : np - 1) >> 6];
    }
#pragma unroll
    for (int j = 0; j < N; j++) ok[j] = ok[j] && ((vw[j] >> lane) & 1ull);  // p & 63 == lane
  }
#pragma unroll
  for (int j = 0; j < N; j++) {
    ok[j] = ok[j] && idx[j] < c.am_size;
    if (!ok[j]) idx[j] = 0;
  }
  if (KIND == KIND_ARRAY) {
#pragma unroll
    for (int j = 0; j < N; j++) m[j] = c.heads[idx[j]];
#pragma unroll
    for (int j = 0; j < N; j++) m[j] = ok[j] ? m[j] : 0u;
  } else {
    uint64_t bits[N], pre[N];
#pragma unroll
    for (int j = 0; j < N; j++) {
      const ulonglong2 e = c.rank_tab[idx[j] >> 6];  // a random lookup costs one line, not two
      bits[j] = e.x;
      pre[j] = e.y;
    }
#pragma unroll
    for (int j = 0; j < N; j++) {
      const bool hit = ok[j] && ((bits[j] >> (idx[j] & 63)) & 1ull);
      const uint32_t rank = (uint32_t)pre[j] + (uint32_t)__popcll(bits[j] & ((1ull << (idx[j] & 63)) - 1ull));
      m[j] = hit ? rank + 1u : 0u;
    }
    if (c.rank_perm) {
      uint32_t row[N];
#pragma unroll
      for (int j = 0; j < N; j++) row[j] = c.rank_perm[m[j] ? m[j] - 1 : 0u];  // clamped, unconditional: N loads in flight
#pragma unroll
      for (int j = 0; j < N; j++) m[j] = m[j] ? row[j] + 1u : 0u;
    }
  }
}

// Which rows of a wave's N probe words have a match (bit `lane` of word[j]) — all that the tile counts ask.  The rank map answers
// from the bitmap half of its entries alone (8 of 16 bytes, no prefix, no rank -> row step): two dependent loads instead of three
// and 50 VGPRs instead of 76, and occupancy is what a latency-bound lookup is paid in (both SF100 Q3 probes: 1.48 ms against
// 2.95 ms through lookup_words; profiles/r2_selective_probe.md).
// `pre` (optional): lane-private "this row exists" flags of the N words (a predicate evaluated by the caller); rows without it skip the
// table like rows outside the row mask do.
// `keys_in` (optional, KIND_RANK): the N raw key values, loaded by the caller ahead of its own work.
template <int KIND, int KT, int N, bool NT = false>
__device__ __forceinline__ void hit_words(const ProbeCtx& c, int64_t w0, int64_t np, uint64_t (&word)[N], const bool* pre = nullptr, const uint64_t* keys_in = nullptr) {
  if (KIND != KIND_RANK) {
    uint32_t m[N];
    lookup_words<KIND, KT, N>(c, w0, np, m);
#pragma unroll
    for (int j = 0; j < N; j++) word[j] = ballot64(m[j] != 0);
    return;
  }
  const unsigned lane = lane_id();
  const KeyCol& k = c.pkeys.c[0];
  uint64_t idx[N];
  bool ok[N];
#pragma unroll
  for (int j = 0; j < N; j++) {
    const int64_t p = ((w0 + j) << 6) + lane;
    ok[j] = p < np;
    idx[j] = (keys_in ? keys_in[j] : load_key<KT, NT>(k, ok[j] ? p : np - 1)) - c.am_offset;
    if (pre) ok[j] = ok[j] && pre[j];
    else if (c.row_mask) ok[j] = ok[j] && ((c.row_mask[w0 + j] >> lane) & 1ull);  // filtered-out rows skip the table
  }
  if (k.valid) {
    uint64_t vw[N];
#pragma unroll
    for (int j = 0; j < N; j++) {
      const int64_t p = ((w0 + j) << 6) + lane;
      vw[j] = k.valid[(p < np ? p : np - 1) >> 6];
    }
#pragma unroll
    for (int j = 0; j < N; j++) ok[j] = ok[j] && ((vw[j] >> lane) & 1ull);
  }
  // the bitmap words alone: 8 useful bytes per 8-byte stride (through rank_tab's {bits, prefix} pairs the same question touched
  // twice the lines — SF300's orders: 450 MB, outside the 256 MB Infinity Cache, against 225 MB inside it)
  const uint64_t* tab = c.rank_bits;
  uint64_t bits[N];
#pragma unroll
  for (int j = 0; j < N; j++) {
    ok[j] = ok[j] && idx[j] < c.am_size;
    bits[j] = tab[ok[j] ? (idx[j] >> 6) : 0];
  }
#pragma unroll
  for (int j = 0; j < N; j++) word[j] = ballot64(ok[j] && ((bits[j] >> (idx[j] & 63)) & 1ull));
}

// per-row output multiplicity for the probe-side part of each JoinType
// (adjust_indices_by_join_type, joins/utils.rs:1432-1488)
__device__ __forceinline__ uint32_t out_count(int join_type, uint32_t nmatch) {
  switch (join_type) {
    case DFGPU_JOIN_INNER: case DFGPU_JOIN_LEFT: return nmatch;
    case DFGPU_JOIN_RIGHT: case DFGPU_JOIN_FULL: return nmatch ? nmatch : 1u;
    case DFGPU_JOIN_RIGHT_SEMI: return nmatch ? 1u : 0u;
    case DFGPU_JOIN_RIGHT_ANTI: return nmatch ? 0u : 1u;
    case DFGPU_JOIN_RIGHT_MARK: return 1u;
    default: return 0u;  // LeftSemi / LeftAnti / LeftMark: emitted from the visited bitmap
  }
}

constexpr int PROBE_UNROLL = 4;

// pass 1, at-most-one-match flavour: writes first_match[p] = build row + 1 (0 = none) and one
// ballot word per 64 probe rows (bit = row produces an output row for this join type).
template <int KIND, int KT>
__global__ __launch_bounds__(BLOCK) void k_probe_first(ProbeCtx c, int64_t np, int invert, uint32_t* __restrict__ first_match,
                                                       uint64_t* __restrict__ mask, uint8_t* __restrict__ visited) {
  const int64_t n_words = (np + 63) >> 6;
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * BLOCK) >> 6;
  for (int64_t w0 = wave * PROBE_UNROLL; w0 < n_words; w0 += n_waves * PROBE_UNROLL) {
    uint32_t m[PROBE_UNROLL];
    lookup_words<KIND, KT, PROBE_UNROLL>(c, w0, np, m);
#pragma unroll
    for (int j = 0; j < PROBE_UNROLL; j++) {
      int64_t p = ((w0 + j) << 6) + lane_id();
      if (p < np) {
        if (first_match) first_match[p] = m[j];
        if (visited && m[j]) visited[m[j] - 1] = 1;
      }
      uint64_t word = ballot64(p < np && ((m[j] != 0) != (invert != 0)));
      if (c.row_mask && w0 + j < n_words) word &= c.row_mask[w0 + j];  // filtered-out probe rows do not exist (anti joins included)
      if (lane_id() == 0 && w0 + j < n_words) mask[w0 + j] = word;
    }
  }
}

// pass 1, general M:N flavour: row_counts[p] = output rows of probe row p, word_counts[w] = sum
// `row_first` (optional): the first MATCHING build row + 1 of every probe row (0 = none) — pass 2 starts its walk there instead of
// looking the key up a second time (the lookup is the expensive part: a random line of the table per probe row)
template <int KIND>
__global__ __launch_bounds__(BLOCK) void k_probe_count(ProbeCtx c, int64_t np, int join_type, uint32_t* __restrict__ row_counts,
                                                       uint32_t* __restrict__ word_counts, uint8_t* __restrict__ visited, uint32_t* __restrict__ row_first = nullptr) {
  const int64_t n_words = (np + 63) >> 6;
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * BLOCK) >> 6;
  if (kind_is_flat<KIND>() && !visited) {
    // the flat table answers "which row first, and how many" from the slot: PROBE_UNROLL words per step, their slot loads in flight
    // together (one word at a time this pass waited on every load: 1.8 ms for 60 M probe rows against a 64 MB table, now the
    // table's random-access rate sets the pace)
    for (int64_t w0 = wave * PROBE_UNROLL; w0 < n_words; w0 += n_waves * PROBE_UNROLL) {
      uint32_t m[PROBE_UNROLL], rws[PROBE_UNROLL];
      lookup_words<KIND, KT_ANY, PROBE_UNROLL>(c, w0, np, m, nullptr, nullptr, rws);
#pragma unroll
      for (int j = 0; j < PROBE_UNROLL; j++) {
        const int64_t p = ((w0 + j) << 6) + lane_id();
        uint32_t cnt = 0;
        if (p < np) {
          cnt = out_count(join_type, rws[j]);
          row_counts[p] = cnt;
          if (row_first) row_first[p] = m[j];
        }
        const uint32_t tot = wave_sum(cnt);
        if (lane_id() == 0 && w0 + j < n_words) word_counts[w0 + j] = tot;
      }
    }
    return;
  }
  for (int64_t w = wave; w < n_words; w += n_waves) {
    int64_t p = (w << 6) + lane_id();
    uint32_t cnt = 0;
    if (p < np) {
      uint32_t nmatch = 0, first = 0;
      if (kind_is_flat<KIND>() && !visited) {
        // the slot holds how many build rows carry the key: counting needs no walk (next[] is read by the pass that emits pairs)
        uint64_t k0, k1;
        if (flat_pack(c.pkeys, c.flat_layout, p, c.null_equals_null != 0, k0, k1))
          first = flat_find<KIND == KIND_FLAT16>(c.flat, c.flat_shift, c.flat_mask, k0, k1, c.force_collisions != 0, &nmatch);
      } else {
      uint32_t cur = chain_head<KIND>(c, p);
      while (cur) {
        int64_t b = (int64_t)cur - 1;
        if (chain_match<KIND>(c, b, p)) {
          if (!nmatch) first = cur;
          nmatch++;
          if (visited) visited[b] = 1;
        }
        cur = c.next ? c.next[b] : 0u;
      }
      }
      cnt = out_count(join_type, nmatch);
      row_counts[p] = cnt;
      if (row_first) row_first[p] = first;
    }
    uint32_t tot = wave_sum(cnt);
    if (lane_id() == 0) word_counts[w] = tot;
  }
}

// pass 2, general flavour: emit (build_idx, probe_idx) pairs; -1 = NULL side
template <int KIND>
__global__ __launch_bounds__(BLOCK) void k_probe_emit(ProbeCtx c, int64_t np, int join_type, const uint32_t* __restrict__ row_counts,
                                                      const uint64_t* __restrict__ prefix, int64_t* __restrict__ out_build,
                                                      int64_t* __restrict__ out_probe, uint8_t* __restrict__ out_mark, const uint32_t* __restrict__ row_first = nullptr) {
  const int64_t n_words = (np + 63) >> 6;
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * BLOCK) >> 6;
  const bool emit_pairs = join_type == DFGPU_JOIN_INNER || join_type == DFGPU_JOIN_LEFT || join_type == DFGPU_JOIN_RIGHT || join_type == DFGPU_JOIN_FULL;
  for (int64_t w = wave; w < n_words; w += n_waves) {
    int64_t p = (w << 6) + lane_id();
    uint32_t cnt = p < np ? row_counts[p] : 0u;
    uint32_t inc = wave_inclusive_sum(cnt);
    int64_t o = (int64_t)prefix[w] + (inc - cnt);
    if (cnt == 0) continue;
    uint32_t nmatch = 0;
    // Inner / Left: a row's output rows ARE its matches, so the walk ends with the last one — a probe row with one match (all of
    // them, over unique build keys) never reads next[]
    const uint32_t stop_at = (join_type == DFGPU_JOIN_INNER || join_type == DFGPU_JOIN_LEFT) ? cnt : 0xFFFFFFFFu;
    if (emit_pairs || join_type == DFGPU_JOIN_RIGHT_MARK) {
      uint32_t cur = row_first ? row_first[p] : chain_head<KIND>(c, p);
      while (cur) {
        int64_t b = (int64_t)cur - 1;
        if (chain_match<KIND>(c, b, p)) {
          if (emit_pairs) { out_build[o] = b; out_probe[o] = p; o++; }
          nmatch++;
          if (nmatch == stop_at) break;
        }
        cur = c.next ? c.next[b] : 0u;
      }
    }
    if (emit_pairs) {
      if (nmatch == 0) { out_build[o] = -1; out_probe[o] = p; }  // Right / Full: unmatched probe row
    } else {
      out_probe[o] = p;
      if (out_build) out_build[o] = -1;
      if (out_mark) out_mark[o] = nmatch ? 1 : 0;
    }
  }
}

// The general M:N probe in ONE pass (round 4): flat tables answer "which build row first, and how many" from the slot, so a tile of
// probe rows knows its number of pairs after one lookup — it reserves that many output positions with ONE atomic on a device-wide
// cursor and writes its pairs there (probe order inside the tile, tiles in the order they got there: for INNER joins whose order no
// ancestor observes, probe_mode 4).  No counts array, no scan, no second pass that re-reads what the first one found: 60 M probe
// rows x 1 M two-column build rows: counts 1.77 + pairs 0.91 ms -> 2.04 ms.  The pair buffer is sized from a sample of the probe rows with
// slack; a tile that would write past it sets nothing and only counts — the cursor then holds the exact size for the second try.
template <int KIND>
__global__ __launch_bounds__(BLOCK) void k_probe_pairs_single(ProbeCtx c, int64_t np, unsigned long long* __restrict__ cursor, unsigned long long capacity,
                                                             int64_t* __restrict__ out_build, int64_t* __restrict__ out_probe) {
  constexpr int W = PROBE_UNROLL;
  constexpr int TILE_WORDS = W * (BLOCK / WAVE);
  __shared__ unsigned long long s_wtot[BLOCK / WAVE];
  __shared__ unsigned long long s_base;
  const int64_t n_words = (np + 63) >> 6;
  const int64_t n_tiles = (n_words + TILE_WORDS - 1) / TILE_WORDS;
  const unsigned lane = lane_id();
  const int wv = threadIdx.x >> 6;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t w0 = tile * TILE_WORDS + (int64_t)wv * W;
    uint32_t m[W], rws[W];
    lookup_words<KIND, KT_ANY, W>(c, w0, np, m, nullptr, nullptr, rws);
    // pair counts are summed in 64 bits: one 64-row word over heavily duplicated build keys can hold >= 2^31 pairs
    uint64_t inc[W];
    unsigned long long wave_total = 0;
#pragma unroll
    for (int j = 0; j < W; j++) {
      inc[j] = wave_inclusive_sum_dpp((uint64_t)rws[j]);
      wave_total += (unsigned long long)__shfl(inc[j], 63, 64);
    }
    if (lane == 0) s_wtot[wv] = wave_total;
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned long long t = 0;
#pragma unroll
      for (int i = 0; i < BLOCK / WAVE; i++) t += s_wtot[i];
      s_base = t ? atomicAdd(cursor, t) : 0ull;
    }
    __syncthreads();
    unsigned long long off = s_base, tile_total = 0;
#pragma unroll
    for (int i = 0; i < BLOCK / WAVE; i++) {
      if (i < wv) off += s_wtot[i];
      tile_total += s_wtot[i];
    }
    if (s_base + tile_total <= capacity) {
#pragma unroll
      for (int j = 0; j < W; j++) {
        const int64_t pr = ((w0 + j) << 6) + lane;
        unsigned long long o = off + inc[j] - rws[j];
        uint32_t cur = m[j];
        for (uint32_t q = 0; q < rws[j]; q++) {   // the key's rows: the slot's head, then next[] (flat tables chain equal keys only)
          out_build[o] = (int64_t)cur - 1;
          out_probe[o] = pr;
          o++;
          if (q + 1 < rws[j]) cur = c.next[cur - 1];
        }
        off += (unsigned long long)__shfl(inc[j], 63, 64);
      }
    }
    __syncthreads();   // s_wtot / s_base are reused by the next tile
  }
}
// pairs of `every`-th probe rows' keys (the sample that sizes the single-pass pair buffer)
template <int KIND>
__global__ __launch_bounds__(BLOCK) void k_probe_pairs_sample(ProbeCtx c, int64_t np, int64_t every, int64_t n_sample, unsigned long long* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  uint32_t cnt = 0;
  if (i < n_sample && i * every < np) {
    uint64_t k0, k1;
    if (flat_pack(c.pkeys, c.flat_layout, i * every, c.null_equals_null != 0, k0, k1)) {
      uint32_t rows = 0;
      if (flat_find<KIND == KIND_FLAT16>(c.flat, c.flat_shift, c.flat_mask, k0, k1, c.force_collisions != 0, &rows)) cnt = rows;
    }
  }
  const uint32_t tot = wave_sum(cnt);
  if (lane_id() == 0 && tot) atomicAdd(out, (unsigned long long)tot);
}

// pass 2, fast flavour: fused compaction of the probe columns + gather of the build columns
constexpr int MAX_JOIN_COLS = 12;
struct JoinCopyCols {
  const void* src[MAX_JOIN_COLS];
  void* dst[MAX_JOIN_COLS];
  int width[MAX_JOIN_COLS];
  int n_build;  // first n_build entries gather from the build side, the rest stream the probe side
  int n;
  int key_col;  // entry that is the probe key column itself (single integer key, direct-address kinds): written from the
                // registers the lookup loaded it into instead of being read a second time; -1 = none
  // KIND_RETURNED: a build column is a field of the returned records — element ret_dest[p] * ret_mul + ret_add of `src` (= the records)
  int ret_mul[MAX_JOIN_COLS], ret_add[MAX_JOIN_COLS];
};
template <typename T>
__device__ __forceinline__ void jcopy(const void* src, void* dst, int64_t s, int64_t d) {
  reinterpret_cast<T*>(dst)[d] = reinterpret_cast<const T*>(src)[s];
}
// A tile of the fused probe whose rows ALL come out writes whole, aligned lines and reads its probe columns exactly once:
// non-temporal loads and stores (device.hpp stream_load / stream_store; bench.py's SF100 probe 9.93 -> 9.69 ms).  A tile that
// emits part of its rows shares output lines with its neighbours — L2 merges those pieces, a non-temporal store would send
// each piece to HBM on its own (the same stores made FilterExec's compaction 10-15 % slower) — and keeps the plain forms.
template <typename T>
__device__ __forceinline__ void jcopy_stream(const void* src, void* dst, int64_t s, int64_t d, bool stream_ld, bool stream_st) {
  T v;
  if (stream_ld) v = stream_load(reinterpret_cast<const T*>(src) + s);
  else v = reinterpret_cast<const T*>(src)[s];
  if (stream_st) stream_store(reinterpret_cast<T*>(dst) + d, v);
  else reinterpret_cast<T*>(dst)[d] = v;
}
__global__ __launch_bounds__(BLOCK) void k_join_materialize(JoinCopyCols cols, const uint64_t* __restrict__ mask,
                                                            const uint64_t* __restrict__ prefix,
                                                            const uint32_t* __restrict__ first_match, int64_t np) {
  const int64_t n_words = (np + 63) >> 6;
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * BLOCK) >> 6;
  const unsigned lane = lane_id();
  for (int64_t w0 = wave * PROBE_UNROLL; w0 < n_words; w0 += n_waves * PROBE_UNROLL) {
    bool sel[PROBE_UNROLL];
    int64_t dst[PROBE_UNROLL], brow[PROBE_UNROLL];
#pragma unroll
    for (int j = 0; j < PROBE_UNROLL; j++) {
      int64_t w = w0 + j;
      uint64_t m = w < n_words ? mask[w] : 0ull;
      sel[j] = (m >> lane) & 1ull;
      dst[j] = 0;
      brow[j] = 0;
      if (sel[j]) {
        dst[j] = (int64_t)(prefix[w] + mbcnt(m));
        brow[j] = (int64_t)first_match[(w << 6) + lane] - 1;
      }
    }
    for (int c = 0; c < cols.n; c++) {
      const int width = cols.width[c];
      const bool from_build = c < cols.n_build;
#pragma unroll
      for (int j = 0; j < PROBE_UNROLL; j++) {
        if (!sel[j]) continue;
        int64_t s = from_build ? brow[j] : ((w0 + j) << 6) + lane;
        switch (width) {
          case 16: jcopy<uint4>(cols.src[c], cols.dst[c], s, dst[j]); break;
          case 8: jcopy<uint64_t>(cols.src[c], cols.dst[c], s, dst[j]); break;
          case 4: jcopy<uint32_t>(cols.src[c], cols.dst[c], s, dst[j]); break;
          case 1: jcopy<uint8_t>(cols.src[c], cols.dst[c], s, dst[j]); break;
        }
      }
    }
  }
}

// ---------------------------------------------------------------- single-pass probe (K3+K4+K5 fused)
// At most one match per probe row (unique build keys / probe-side semi+anti) and non-nullable
// payload: lookup, output-offset computation and materialisation happen in ONE kernel, so the
// probe keys are read once and neither match ids nor masks ever reach HBM.  A tile = one
// workgroup pass over 256*W probe rows.  Two ways to place a tile's rows in the output:
//
//  UNORDERED (probe_mode 3) — for plans that do not need the probe order (the join feeds an
//    aggregate / repartition, as in TPC-H Q3): wave 0 claims the tile's output range with ONE
//    returning atomicAdd on a cursor.  Rows stay in probe order inside a tile; tiles land in
//    claim order.  No tile waits on another tile; one workgroup per tile, no persistence.
//  ORDERED (probe_mode 2) — output in probe order like the reference (exec.rs:3349): the global
//    offset of a tile comes from a decoupled look-back over per-tile {status, count} words: a
//    tile publishes its count (AGG), wave 0 walks back 256 predecessors per round until it meets
//    an inclusive prefix (PFX), then publishes its own.  Tiles are handed out in order by an
//    atomic ticket, so a tile only ever waits on tiles whose workgroup is already resident =>
//    forward progress without a cooperative launch or any residency assumption.  Cross-
//    workgroup traffic is one 8-byte agent-scope atomic granule per tile — the form
//    MI355X_MICROARCH.md lists as valid without fences (per-XCD L2s are not coherent).
//    Measured on the SF100 Q3 join this is SLOWER than the two-pass path (15.4 vs 12.5 ms: each
//    look-back round is an agent-scope load queued behind the CU's own streaming loads, 3-5 us,
//    with the whole workgroup parked on it), so `auto` keeps two passes for ordered output.
constexpr int FUSED_W = 8;  // 64-row words per wave per tile: 2048-row tiles (4 and 8 measured equal, 16 slower: 142 VGPRs)
// (tile states and the look-back itself: device.hpp — scan.hip's single-pass scan walks the same chain)
struct alignas(128) FusedCtl {
  unsigned long long total;  // out: number of output rows (ordered: last tile's inclusive prefix; unordered: the cursor)
  char _pad0[120];           // the ticket lives on its own cache line: both words are hot atomics
  unsigned ticket;           // ordered mode: next tile to hand out
  char _pad1[124];
};

// KIND_RETURNED: consecutive tiles read neighbouring records (the rows of one (grouping tile, group) run lie in a few fused tiles),
// and workgroup b runs on XCD b % 8 (MI355X_MICROARCH.md) — so every XCD takes a CONTIGUOUS eighth of the tiles, in order, and the
// lines one of its tiles fetched serve the next ones out of the same L2.  A bijection of [0, n_tiles) for any n_tiles; speed only.
__device__ __forceinline__ int64_t xcd_contiguous_tile(int64_t b, int64_t n_tiles) {
  const int64_t q = n_tiles >> 3, r = n_tiles & 7, x = b & 7, j = b >> 3;
  return x * q + (x < r ? x : r) + j;
}
// per-tile output row counts of the same tiling: pass 1 of the PLACED flavour.  Reads the probe keys (and the row mask)
// only; the table words it touches (rank-map bitmap + directory, MALL-resident) are warm for pass 2.
// `out_words` (optional): the output rows themselves, one bit per probe row — what k_join_emit_listed materialises from.
// PRED: a FilterExec's predicate (RowPred) evaluated here, per row, instead of a row mask: its columns' loads go out first, all W of
// them, then the keys'; rows it drops skip the table.
template <int KIND, int KT, int W, bool PRED = false, bool NT = false>
__global__ __launch_bounds__(BLOCK) void k_join_tile_counts(ProbeCtx c, int64_t np, int invert, const uint64_t* __restrict__ row_mask,
                                                            uint32_t* __restrict__ tile_counts, uint64_t* __restrict__ out_words, RowPred pred = RowPred{}) {
  __shared__ uint32_t s_wcount[BLOCK / WAVE];
  constexpr int TILE_WORDS = W * (BLOCK / WAVE);
  static_assert(TILE_WORDS <= BLOCK, "one thread per output word of the tile");
  __shared__ uint64_t s_words[TILE_WORDS];
  const int64_t n_words = (np + 63) >> 6;
  const unsigned lane = lane_id();
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // uniform: the row mask words load as scalars
  const int64_t tile = KIND == KIND_RETURNED ? xcd_contiguous_tile(blockIdx.x, gridDim.x) : (int64_t)blockIdx.x;
  const int64_t w0 = tile * TILE_WORDS + (int64_t)wv * W;
  uint64_t hit[W];
  bool pre[PRED ? W : 1];
  if (PRED) {
    // the keys' loads first, then the predicate columns': everything is in flight before the first comparison waits
    uint64_t keyv[KIND == KIND_RANK ? W : 1];
    if (KIND == KIND_RANK) {
#pragma unroll
      for (int j = 0; j < W; j++) {
        const int64_t p = ((w0 + j) << 6) + lane;
        keyv[KIND == KIND_RANK ? j : 0] = load_key<KT, NT>(c.pkeys.c[0], p < np ? p : np - 1);
      }
    }
#pragma unroll
    for (int j = 0; j < W; j++) pre[PRED ? j : 0] = (((w0 + j) << 6) + lane) < np;
#pragma unroll
    for (int i = 0; i < ROWPRED_MAX; i++) {
      if (i >= pred.n) continue;     // (uniform: one branch per conjunct, none per load)
      long long v[W];
      rowpred_load_words<NT, W>(pred, i, w0, np, lane, v);
      const unsigned sel = pred.op[i];
      const long long lit = pred.lit[i];
#pragma unroll
      for (int j = 0; j < W; j++) pre[PRED ? j : 0] = pre[PRED ? j : 0] & rowpred_cmp(sel, v[j], lit);
      if (pred.valid[i]) {
        uint64_t vw[W];
#pragma unroll
        for (int j = 0; j < W; j++) {
          const int64_t p = ((w0 + j) << 6) + lane;
          vw[j] = pred.valid[i][(p < np ? p : np - 1) >> 6];
        }
#pragma unroll
        for (int j = 0; j < W; j++) pre[PRED ? j : 0] = pre[PRED ? j : 0] & (bool)((vw[j] >> lane) & 1ull);
      }
    }
    hit_words<KIND, KT, W, NT>(c, w0, np, hit, pre, KIND == KIND_RANK ? keyv : nullptr);
  } else {
    hit_words<KIND, KT, W, NT>(c, w0, np, hit);
  }
  uint32_t wave_cnt = 0;
#pragma unroll
  for (int j = 0; j < W; j++) {
    const int64_t p = ((w0 + j) << 6) + lane;
    uint64_t word = ballot64(p < np) & (invert ? ~hit[j] : hit[j]);
    if (PRED) word &= ballot64(pre[PRED ? j : 0]);
    else if (row_mask) word &= (w0 + j < n_words) ? row_mask[w0 + j] : 0ull;
    // the tile's output words leave together, below (one 8-byte store per wave and word — 28 M partial-line stores for SF300's
    // lineitem — cost 0.7 ms of 4.0: scripts/microbench/stream_width.hip G vs E)
    if (lane == 0) s_words[wv * W + j] = word;
    wave_cnt += (uint32_t)__popcll(word);
  }
  if (lane == 0) s_wcount[wv] = wave_cnt;
  __syncthreads();
  if (out_words && threadIdx.x < TILE_WORDS) {
    const int64_t w = tile * TILE_WORDS + threadIdx.x;
    if (w < n_words) out_words[w] = s_words[threadIdx.x];   // TILE_WORDS x 8 bytes, contiguous: whole lines
  }
  if (threadIdx.x == 0) {
    uint32_t agg = 0;
#pragma unroll
    for (int i = 0; i < BLOCK / WAVE; i++) agg += s_wcount[i];
    tile_counts[tile] = agg;
  }
}

enum FusedMode : int { FUSED_UNORDERED = 0, FUSED_LOOKBACK = 1, FUSED_PLACED = 2 };
template <int KIND, int KT, int W, int MODE, bool KEYREG>
// (flat tables: the lookup is a random line per row, a fourth wave per SIMD buys 0.2 ms of 2.0 on the Decimal128 shape.  The other kinds
// keep the compiler's own choice: the same bound took the rank-map instance of the SF100 probe from 9.1 to 11.4 ms)
__global__ __launch_bounds__(BLOCK, ((KIND == KIND_FLAT || KIND == KIND_FLAT16) ? 4 : 1)) void k_join_probe_fused(ProbeCtx c, JoinCopyCols cols, int64_t np, int invert, uint64_t* __restrict__ tile_state,
                                                            FusedCtl* __restrict__ ctl, const uint64_t* __restrict__ row_mask) {
  constexpr bool ORDERED = MODE == FUSED_LOOKBACK;
  __shared__ unsigned s_tile;
  __shared__ uint32_t s_wcount[BLOCK / WAVE];
  __shared__ uint64_t s_prefix;
  __shared__ bool s_full;  // every row of the tile comes out
  constexpr int TILE_WORDS = W * (BLOCK / WAVE);
  const int64_t n_words = (np + 63) >> 6;
  const int64_t n_tiles = (n_words + TILE_WORDS - 1) / TILE_WORDS;
  const unsigned lane = lane_id();
  const int wv = threadIdx.x >> 6;
  for (;;) {
    int64_t tile = KIND == KIND_RETURNED && !ORDERED ? xcd_contiguous_tile(blockIdx.x, n_tiles) : (int64_t)blockIdx.x;
    if (ORDERED) {
      // every tile takes its OWN ticket: handing one workgroup several consecutive tiles would make
      // tile 4k wait on the AGG of tile 4k-1, which its owner only reaches after finishing
      // 4k-4..4k-2 => a fully serial chain
      if (threadIdx.x == 0) s_tile = atomicAdd(&ctl->ticket, 1u);
      __syncthreads();
      tile = s_tile;
      if (tile >= n_tiles) return;
    }
    const int64_t w0 = tile * TILE_WORDS + (int64_t)wv * W;

    // ---- lookup (every stage issues its W loads back-to-back: memory-level parallelism)
    uint32_t m[W];
    uint64_t key[KEYREG ? W : 1];
    uint4 rec16[KIND == KIND_RETURNED ? W : 1];
    const bool rec_in_regs = KIND == KIND_RETURNED && c.ret_R == 16;
    // (the array is handed over whenever the kind has records — a pointer chosen at run time kept it in scratch memory: 144 bytes per
    // lane written and read back in the kernel that moves 65 GB; whether the records are 16 bytes wide is asked inside)
    lookup_words<KIND, KT, W>(c, w0, np, m, KEYREG ? key : nullptr, KIND == KIND_RETURNED ? rec16 : nullptr);
    uint64_t word[W];  // wave-uniform (SGPR pairs)
    uint32_t wave_cnt = 0;
#pragma unroll
    for (int j = 0; j < W; j++) {
      int64_t p = ((w0 + j) << 6) + lane;
      word[j] = ballot64(p < np && ((m[j] != 0) != (invert != 0)));
      // a FilterExec fused below the probe side: rows whose predicate is false / NULL do not exist for the join
      if (row_mask) word[j] &= (w0 + j < n_words) ? row_mask[w0 + j] : 0ull;
      wave_cnt += (uint32_t)__popcll(word[j]);
    }
    if (lane == 0) s_wcount[wv] = wave_cnt;
    __syncthreads();

    // ---- where the tile's rows go (wave 0)
    if (wv == 0) {
      uint64_t agg = 0;
#pragma unroll
      for (int i = 0; i < BLOCK / WAVE; i++) agg += s_wcount[i];
      uint64_t excl = 0;
      if (ORDERED) {
        excl = lookback_exclusive(tile_state, tile, agg);
        if (lane == 0 && tile == n_tiles - 1) ctl->total = excl + agg;
      } else if (MODE == FUSED_PLACED) {
        if (tile_state) {
          excl = tile_state[tile];  // exclusive prefix of k_join_tile_counts: the output is in probe order, allocation exact
        } else {
          // SPECULATION (join_probe): every probe row finds its key — a foreign key against its primary key — so row i of the
          // output is probe row i and no counts pass is needed.  A tile that disagrees raises the flag; its rows stay inside its own
          // range of the output, and the host starts over with the counts.
          excl = (uint64_t)tile * (uint64_t)(TILE_WORDS * 64);
          const int64_t left = np - tile * (int64_t)(TILE_WORDS * 64);
          const uint64_t rows_here = (uint64_t)(left < (int64_t)(TILE_WORDS * 64) ? left : (int64_t)(TILE_WORDS * 64));
          if (lane == 0 && agg != rows_here) ctl->ticket = 1u;
        }
      } else if (lane == 0 && agg) {
        excl = atomicAdd(&ctl->total, (unsigned long long)agg);
      }
      if (lane == 0) {
        s_prefix = excl;
        s_full = agg == (uint64_t)TILE_WORDS * 64;
      }
    }
    __syncthreads();
    uint64_t wave_base = s_prefix;
#pragma unroll
    for (int i = 0; i < BLOCK / WAVE; i++)
      if (i < wv) wave_base += s_wcount[i];

    // ---- materialise: probe columns stream, build columns gather; probe order inside the tile
    const bool full = s_full;  // uniform: the two forms of a copy differ in their cache-policy bits only
    for (int cidx = 0; cidx < cols.n; cidx++) {
      const int width = cols.width[cidx];
      const bool from_build = cidx < cols.n_build;
      const bool stream = full && !from_build;
      uint64_t ob = wave_base;
#pragma unroll
      for (int j = 0; j < W; j++) {
        const int64_t d = (int64_t)(ob + mbcnt(word[j]));
        ob += (uint32_t)__popcll(word[j]);
        if (!((word[j] >> lane) & 1ull)) continue;
        if (KEYREG && cidx == cols.key_col) {  // a tile's key lines have left the L2 by now (256 tiles x ~100 KB are in flight)
          if (KT == KT_I64) reinterpret_cast<uint64_t*>(cols.dst[cidx])[d] = key[KEYREG ? j : 0];
          else if (KT == KT_U8) reinterpret_cast<uint8_t*>(cols.dst[cidx])[d] = (uint8_t)key[KEYREG ? j : 0];
          else reinterpret_cast<uint32_t*>(cols.dst[cidx])[d] = (uint32_t)key[KEYREG ? j : 0];
          continue;
        }
        int64_t s = from_build ? (int64_t)m[j] - 1 : ((w0 + j) << 6) + lane;
        if (KIND == KIND_RETURNED && from_build) {
          if (rec_in_regs) {  // a 4- or 8-byte field of the record the lookup loaded
            const uint4 rv = rec16[KIND == KIND_RETURNED ? j : 0];
            const int bo = cols.ret_add[cidx] * width;   // byte offset inside the record (word 0 is the match id)
            const uint32_t w32 = (bo >> 2) == 1 ? rv.y : (bo >> 2) == 2 ? rv.z : rv.w;
            if (width == 4) reinterpret_cast<uint32_t*>(cols.dst[cidx])[d] = w32;
            else if (width == 8) reinterpret_cast<uint64_t*>(cols.dst[cidx])[d] = ((uint64_t)rv.w << 32) | rv.z;   // an 8-byte field is the upper half
            else reinterpret_cast<uint8_t*>(cols.dst[cidx])[d] = (uint8_t)(w32 >> ((bo & 3) * 8));
            continue;
          }
          s = (int64_t)c.ret_dest[((w0 + j) << 6) + lane] * cols.ret_mul[cidx] + cols.ret_add[cidx];
        }
        switch (width) {
          case 16: jcopy_stream<uint4>(cols.src[cidx], cols.dst[cidx], s, d, stream, full); break;
          case 8: jcopy_stream<uint64_t>(cols.src[cidx], cols.dst[cidx], s, d, stream, full); break;
          case 4: jcopy_stream<uint32_t>(cols.src[cidx], cols.dst[cidx], s, d, stream, full); break;
          case 1: jcopy_stream<uint8_t>(cols.src[cidx], cols.dst[cidx], s, d, stream, full); break;
        }
      }
    }
    if (!ORDERED) return;
    __syncthreads();  // s_tile / s_wcount / s_prefix are reused by the next tile
  }
}

// ---------------------------------------------------------------- the selective probe (direct-address kinds)
// A probe that emits few of its rows is made of latency, not of bytes: in the fused kernel every tile walks key -> table ->
// (rank -> row) -> barrier -> offset -> a load/store round trip per column with a few lanes alive, at the 5 waves per SIMD
// that the materialising half of the kernel leaves (SF100 Q3 under its row masks: 5.4 ms for 6 GB of keys).  Split where
// the work changes shape: k_join_tile_counts looks every key up at 50 VGPRs (key -> bitmap word) and leaves the output rows
// as one bit per probe row next to its tile counts; after the scan, this kernel lists the set bits of 8192 probe rows in
// LDS and gives every OUTPUT row one thread that takes the whole chain (key, rank, build row, all columns' loads in flight
// together) and writes consecutive output rows.  The output is in probe order and allocated exactly.
constexpr int EL_WORDS = 128;  // probe words per workgroup: 4 tiles of the counts pass
// A very selective probe (SF300's lineitem under Q3's filters: 9 M of 1.8 G rows) leaves ~40 listed rows per 8192-row group: 40 of 256
// threads walk the key -> rank -> build row -> columns chain and the workgroup is gone, 220 K times over (1.05 ms).  EL_WORDS_SPARSE
// gives a workgroup 65536 probe rows (the most a 16-bit row index addresses): ~330 rows per chain at that density, an eighth of
// the workgroups.  The listed rows are taken EL_CAP at a time, so any density stays correct.
constexpr int EL_WORDS_SPARSE = 1024;
constexpr int EL_CAP = 8192;   // listed rows per round (16 KB of LDS)
template <int KIND, int KT, int EW>
__global__ __launch_bounds__(BLOCK) void k_join_emit_listed(ProbeCtx c, JoinCopyCols cols, int64_t np, const uint64_t* __restrict__ out_words,
                                                            const uint64_t* __restrict__ tile_prefix, int tiles_per_group) {
  static_assert(EW % BLOCK == 0 || BLOCK % EW == 0, "whole words per thread");
  static_assert(EW * 64 <= 65536, "a listed row is a 16-bit index into the group");
  constexpr int WPT = EW >= BLOCK ? EW / BLOCK : 1;   // words per thread (contiguous)
  __shared__ uint64_t s_word[EW];
  __shared__ uint32_t s_off[EW];
  __shared__ uint32_t s_wsum[BLOCK / WAVE];
  __shared__ uint16_t s_row[EL_CAP];
  const int64_t n_words = (np + 63) >> 6;
  const int64_t w_base = (int64_t)blockIdx.x * EW;
  const unsigned lane = lane_id();
  const int wv = threadIdx.x >> 6;
  uint64_t w[WPT];
  uint32_t cnt = 0;
#pragma unroll
  for (int t = 0; t < WPT; t++) {
    const int i = (int)threadIdx.x * WPT + t;
    w[t] = (i < EW && w_base + i < n_words) ? out_words[w_base + i] : 0ull;
    cnt += (uint32_t)__popcll(w[t]);
  }
  const uint32_t inc = wave_inclusive_sum(cnt);
  if (lane == 63) s_wsum[wv] = inc;
#pragma unroll
  for (int t = 0; t < WPT; t++) {
    const int i = (int)threadIdx.x * WPT + t;
    if (i < EW) s_word[i] = w[t];
  }
  __syncthreads();
  uint32_t before = 0, agg = 0;
#pragma unroll
  for (int i = 0; i < BLOCK / WAVE; i++) {
    if (i < wv) before += s_wsum[i];
    agg += s_wsum[i];
  }
  if (agg == 0) return;
  {
    uint32_t off = before + inc - cnt;
#pragma unroll
    for (int t = 0; t < WPT; t++) {
      const int i = (int)threadIdx.x * WPT + t;
      if (i < EW) s_off[i] = off;
      off += (uint32_t)__popcll(w[t]);
    }
  }
  __syncthreads();
  const uint64_t out0 = tile_prefix[(int64_t)blockIdx.x * tiles_per_group];
  const KeyCol& k = c.pkeys.c[0];
  for (uint32_t base = 0; base < agg; base += EL_CAP) {
  if (base) __syncthreads();   // the previous round's readers of s_row are done
  for (int i = wv; i < EW; i += BLOCK / WAVE) {  // the listed rows of this round, in probe order
    const uint64_t ww = s_word[i];
    const uint32_t pos = s_off[i] + mbcnt(ww) - base;
    if (((ww >> lane) & 1ull) && pos < (uint32_t)EL_CAP) s_row[pos] = (uint16_t)((i << 6) | lane);   // (pos wraps for rows of earlier rounds)
  }
  __syncthreads();
  const uint32_t round_rows = agg - base < (uint32_t)EL_CAP ? agg - base : (uint32_t)EL_CAP;
  for (uint32_t q0 = threadIdx.x; q0 < round_rows; q0 += BLOCK) {
    const uint32_t q = base + q0;
    const int64_t prow = (w_base << 6) + s_row[q0];
    int64_t brow = 0;
    uint64_t key = 0;
    if (cols.n_build > 0 || cols.key_col >= 0) {  // (a RightAnti probe lists the rows WITHOUT a match and has no build columns)
      key = load_key<KT>(k, prow);
      if (cols.n_build > 0) {
        const uint64_t idx = key - c.am_offset;
        if (KIND == KIND_ARRAY) {
          brow = (int64_t)c.heads[idx] - 1;
        } else {
          uint64_t bw, pre;
          if (c.rank_tab) {   // (uniform)
            const ulonglong2 e = c.rank_tab[idx >> 6];
            bw = e.x;
            pre = e.y;
          } else {
            bw = c.rank_bits[idx >> 6];
            pre = c.rank_prefix[idx >> 6];
          }
          const uint32_t rank = (uint32_t)pre + (uint32_t)__popcll(bw & ((1ull << (idx & 63)) - 1ull));
          brow = c.rank_perm ? (int64_t)c.rank_perm[rank] : (int64_t)rank;
        }
      }
    }
    const int64_t d = (int64_t)(out0 + q);
#pragma unroll
    for (int c0 = 0; c0 < MAX_JOIN_COLS; c0 += 4) {
      if (c0 >= cols.n) continue;
      uint4 v[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        if (c0 + u >= cols.n) continue;
        int ci = c0 + u + cols.n_build;  // the probe side's columns first: their loads do not wait for the build row
        if (ci >= cols.n) ci -= cols.n;
        if (ci == cols.key_col) {
          v[u].x = (uint32_t)key;
          v[u].y = (uint32_t)(key >> 32);
          continue;
        }
        const int64_t sr = ci < cols.n_build ? brow : prow;
        switch (cols.width[ci]) {
          case 16: v[u] = reinterpret_cast<const uint4*>(cols.src[ci])[sr]; break;
          case 8: {
            const uint2 t = reinterpret_cast<const uint2*>(cols.src[ci])[sr];
            v[u].x = t.x;
            v[u].y = t.y;
          } break;
          case 4: v[u].x = reinterpret_cast<const uint32_t*>(cols.src[ci])[sr]; break;
          case 1: v[u].x = reinterpret_cast<const uint8_t*>(cols.src[ci])[sr]; break;
        }
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        if (c0 + u >= cols.n) continue;
        int ci = c0 + u + cols.n_build;
        if (ci >= cols.n) ci -= cols.n;
        switch (cols.width[ci]) {
          case 16: reinterpret_cast<uint4*>(cols.dst[ci])[d] = v[u]; break;
          case 8: reinterpret_cast<uint2*>(cols.dst[ci])[d] = make_uint2(v[u].x, v[u].y); break;
          case 4: reinterpret_cast<uint32_t*>(cols.dst[ci])[d] = v[u].x; break;
          case 1: reinterpret_cast<uint8_t*>(cols.dst[ci])[d] = (uint8_t)v[u].x; break;
        }
      }
    }
  }
  }
}

// ---------------------------------------------------------------- grouped probe with positional return (round 4)
// Probe keys in NO order against a rank map beyond the caches: every lookup is a 64-byte unit of HBM / Infinity Cache of its own
// (~50 G/s), and so is every build-payload gather behind it (150 M shuffled x 600 M rows with the Q3 payload: ~24 ms).  Moving the
// probe ROWS to the table (radix partitioning keys and payload together, the textbook GPU join) costs two passes over 40 bytes
// per row; moving only the KEYS costs 8:
//   1. group_rows_by_key (grouped.hip): the probe keys grouped by the top bits of their position in the table's key range, 2^9
//      groups or so, every probe row told where its key went (`dest`);
//   2. k_gp_lookup: group by group — a group's slice of the table and of the build payload is a few MB, and all workgroups of
//      an XCD walk the groups in the same order, so the slice is fetched into that XCD's L2 once — every key is looked up and what
//      it found (match id + the build columns of the output) is left as one record per key, in group order;
//   3. the ordinary fused probe in PROBE order with KIND_RETURNED as its "table": row p reads record dest[p].  Rows of one
//      (tile, group) run share lines, so these reads are L2 hits, not HBM lines.  Output in probe order: every probe_mode is served.
// The build payload has to be read at rank positions: as it is when the build keys ascend (row == rank), else from a rank-ordered
// copy made once per join table (ensure_rank_payload: the build rows grouped the same way, then placed inside L2-sized ranges).
constexpr int GP_U = 4;  // keys per thread and step
struct RetLayout {
  const void* src[MAX_JOIN_COLS];   // the build columns, readable at RANK positions
  int width[MAX_JOIN_COLS], off[MAX_JOIN_COLS];
  int n, R;
  const uint4* packed16;            // R == 16 only: the build columns of rank r already laid out as the record (word 0 free): one access
};
// Which rows a workgroup takes next.  All workgroups of an XCD (blockIdx % 8, MI355X_MICROARCH.md: a placement that only speed
// depends on) walk that XCD's groups — g = xcd, xcd + 8, ... — in order, GP_CHUNK rows at a time, handed out by ONE counter per XCD:
// whatever pace the workgroups keep, the chunks in flight on an XCD are consecutive — 256 workgroups x 2048 rows = less than one
// group — so its L2 holds the slice of the table of one or two groups at a time.  (Round 4's first form gave every workgroup a
// fixed share of every group: the workgroups drifted apart over the 128 groups, the XCD worked on many slices at once and 44 %
// of the lookups missed the L2, profiles/r4_join_grouped_pmc.md.)
constexpr int GP_CHUNK = 2048;
struct GpWalk {
  uint32_t pref[GP_MAX_GROUPS / 8 + 2];   // chunks of this XCD's groups before its k-th group
  unsigned ticket;
  int ng;
};
__device__ __forceinline__ void gp_walk_init(GpWalk& w, const uint64_t* __restrict__ bounds, int P) {
  const int xcd = blockIdx.x & 7;
  const int ng = xcd < P ? (P - xcd + 7) / 8 : 0;
  for (int k = threadIdx.x; k < ng; k += BLOCK) {
    const int g = xcd + 8 * k;
    w.pref[k + 1] = (uint32_t)((bounds[g + 1] - bounds[g] + GP_CHUNK - 1) / GP_CHUNK);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    w.pref[0] = 0;
    for (int k = 0; k < ng; k++) w.pref[k + 1] += w.pref[k];
    w.ng = ng;
  }
  __syncthreads();
}
// false = no chunk left.  Every thread of the workgroup calls it (barriers inside).
__device__ __forceinline__ bool gp_walk_next(GpWalk& w, const uint64_t* __restrict__ bounds, unsigned* __restrict__ tickets, int64_t& lo, int64_t& hi) {
  const int xcd = blockIdx.x & 7;
  __syncthreads();   // the previous chunk's readers of w.ticket are done
  if (threadIdx.x == 0) w.ticket = atomicAdd(&tickets[xcd * 32], 1u);
  __syncthreads();
  const unsigned t = w.ticket;
  if (t >= w.pref[w.ng]) return false;
  int a = 0, b = w.ng - 1;   // the last k with pref[k] <= t
  while (a < b) {
    const int m = (a + b + 1) >> 1;
    if (w.pref[m] <= t) a = m;
    else b = m - 1;
  }
  const int g = xcd + 8 * a;
  lo = (int64_t)bounds[g] + (int64_t)(t - w.pref[a]) * GP_CHUNK;
  const int64_t end = (int64_t)bounds[g + 1];
  hi = lo + GP_CHUNK < end ? lo + GP_CHUNK : end;
  return true;
}
__global__ __launch_bounds__(BLOCK) void k_gp_lookup(const ulonglong2* __restrict__ rank_tab, uint64_t am_offset, uint64_t am_size, const uint64_t* __restrict__ gkeys,
                                                     const uint64_t* __restrict__ bounds, int P, RetLayout L, uint8_t* __restrict__ rec,
                                                     unsigned long long* __restrict__ total_hits, unsigned* __restrict__ tickets) {
  __shared__ unsigned long long s_hits[BLOCK / WAVE];
  __shared__ GpWalk walk;
  unsigned long long hits = 0;
  gp_walk_init(walk, bounds, P);
  for (int64_t lo, hi; gp_walk_next(walk, bounds, tickets, lo, hi);) {
    for (int64_t base = lo; base < hi; base += BLOCK * GP_U) {
      uint64_t idx[GP_U];
      bool in[GP_U];
#pragma unroll
      for (int u = 0; u < GP_U; u++) {
        const int64_t i = base + u * BLOCK + threadIdx.x;
        in[u] = i < hi;
        idx[u] = stream_load(gkeys + (in[u] ? i : hi - 1)) - am_offset;   // in range by construction (group_rows_by_key drops the others)
      }
      ulonglong2 e[GP_U];
#pragma unroll
      for (int u = 0; u < GP_U; u++) e[u] = rank_tab[idx[u] >> 6];
      uint32_t m[GP_U];
#pragma unroll
      for (int u = 0; u < GP_U; u++) {
        const bool hit = in[u] && ((e[u].x >> (idx[u] & 63)) & 1ull);
        const uint32_t rank = (uint32_t)e[u].y + (uint32_t)__popcll(e[u].x & ((1ull << (idx[u] & 63)) - 1ull));
        m[u] = hit ? rank + 1u : 0u;
        hits += hit ? 1u : 0u;
      }
      if (L.R == 16) {
        // the whole record in registers, ONE 16-byte store per row (streamed: the records are read back much later, and must not
        // push the group's slice of the table out of the L2)
        uint4 rv[GP_U];
        if (L.packed16) {   // the payload of a rank is one 16-byte record already: one L2 access per row beside the table's
#pragma unroll
          for (int u = 0; u < GP_U; u++) rv[u] = L.packed16[m[u] ? m[u] - 1u : 0u];
#pragma unroll
          for (int u = 0; u < GP_U; u++) rv[u].x = m[u];
        } else {
#pragma unroll
        for (int u = 0; u < GP_U; u++) rv[u] = make_uint4(m[u], 0u, 0u, 0u);
        for (int c = 0; c < L.n; c++) {
          const int w = L.width[c], bo = L.off[c];
#pragma unroll
          for (int u = 0; u < GP_U; u++) {
            const int64_t r = m[u] ? (int64_t)m[u] - 1 : 0;   // (a miss reads rank 0: harmless, nobody looks at its fields)
            if (w == 8) {
              const uint64_t v = reinterpret_cast<const uint64_t*>(L.src[c])[r];
              rv[u].z = (uint32_t)v;
              rv[u].w = (uint32_t)(v >> 32);
            } else {
              const uint32_t v = w == 4 ? reinterpret_cast<const uint32_t*>(L.src[c])[r] : (uint32_t)reinterpret_cast<const uint8_t*>(L.src[c])[r] << ((bo & 3) * 8);
              rv[u].y |= (bo >> 2) == 1 ? v : 0u;
              rv[u].z |= (bo >> 2) == 2 ? v : 0u;
              rv[u].w |= (bo >> 2) == 3 ? v : 0u;
            }
          }
        }
        }
#pragma unroll
        for (int u = 0; u < GP_U; u++) {
          const int64_t i = base + u * BLOCK + threadIdx.x;
          if (in[u]) stream_store(reinterpret_cast<uint4*>(rec) + i, rv[u]);
        }
        continue;
      }
#pragma unroll
      for (int u = 0; u < GP_U; u++) {
        const int64_t i = base + u * BLOCK + threadIdx.x;
        if (in[u]) *reinterpret_cast<uint32_t*>(rec + (uint64_t)i * L.R) = m[u];
      }
      for (int c = 0; c < L.n; c++) {
        const int w = L.width[c];
#pragma unroll
        for (int u = 0; u < GP_U; u++) {
          if (!m[u]) continue;
          const int64_t i = base + u * BLOCK + threadIdx.x;
          const int64_t r = (int64_t)m[u] - 1;
          uint8_t* d = rec + (uint64_t)i * L.R + L.off[c];
          switch (w) {
            case 16: *reinterpret_cast<uint4*>(d) = reinterpret_cast<const uint4*>(L.src[c])[r]; break;
            case 8: *reinterpret_cast<uint64_t*>(d) = reinterpret_cast<const uint64_t*>(L.src[c])[r]; break;
            case 4: *reinterpret_cast<uint32_t*>(d) = reinterpret_cast<const uint32_t*>(L.src[c])[r]; break;
            default: *d = reinterpret_cast<const uint8_t*>(L.src[c])[r]; break;
          }
        }
      }
    }
  }
  hits = wave_sum<unsigned long long>(hits);
  if (lane_id() == 0) s_hits[threadIdx.x >> 6] = hits;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
    for (int w = 0; w < BLOCK / WAVE; w++) t += s_hits[w];
    if (t) atomicAdd(total_hits, t);
  }
}
// build rows in group order (every key is in the table): their columns to the RANK position of their key
// `out16` (optional): the columns of a row as ONE 16-byte record at its rank — field c at byte rec_off[c], word 0 left free — instead
// of one store per column (a store of a few bytes leaves the L2 as a whole fabric write: 4-byte stores made this kernel write
// 13 GB for 1.2 GB of payload, profiles/r4_join_grouped_pmc.md)
struct PlaceRec {
  uint4* out16;
  int off[GP_MAX_COLS];
};
__global__ __launch_bounds__(BLOCK) void k_gp_place(const ulonglong2* __restrict__ rank_tab, uint64_t am_offset, const uint64_t* __restrict__ gkeys,
                                                    const uint64_t* __restrict__ bounds, int P, GroupCols cols, unsigned* __restrict__ tickets, PlaceRec pr) {
  __shared__ GpWalk walk;
  gp_walk_init(walk, bounds, P);
  for (int64_t lo, hi; gp_walk_next(walk, bounds, tickets, lo, hi);) {
    for (int64_t base = lo; base < hi; base += BLOCK * GP_U) {
      uint64_t idx[GP_U];
      bool in[GP_U];
#pragma unroll
      for (int u = 0; u < GP_U; u++) {
        const int64_t i = base + u * BLOCK + threadIdx.x;
        in[u] = i < hi;
        idx[u] = stream_load(gkeys + (in[u] ? i : hi - 1)) - am_offset;
      }
      ulonglong2 e[GP_U];
#pragma unroll
      for (int u = 0; u < GP_U; u++) e[u] = rank_tab[idx[u] >> 6];
      if (pr.out16) {
        uint4 rv[GP_U];
#pragma unroll
        for (int u = 0; u < GP_U; u++) rv[u] = make_uint4(0u, 0u, 0u, 0u);
        for (int c = 0; c < cols.n; c++) {
          const int w = cols.width[c], bo = pr.off[c];
#pragma unroll
          for (int u = 0; u < GP_U; u++) {
            const int64_t i = in[u] ? base + u * BLOCK + threadIdx.x : lo;
            if (w == 8) {
              const uint64_t v = reinterpret_cast<const uint64_t*>(cols.src[c])[i];
              rv[u].z = (uint32_t)v;
              rv[u].w = (uint32_t)(v >> 32);
            } else {
              const uint32_t v = w == 4 ? reinterpret_cast<const uint32_t*>(cols.src[c])[i] : (uint32_t)reinterpret_cast<const uint8_t*>(cols.src[c])[i] << ((bo & 3) * 8);
              rv[u].y |= (bo >> 2) == 1 ? v : 0u;
              rv[u].z |= (bo >> 2) == 2 ? v : 0u;
              rv[u].w |= (bo >> 2) == 3 ? v : 0u;
            }
          }
        }
#pragma unroll
        for (int u = 0; u < GP_U; u++) {
          if (!in[u]) continue;
          const int64_t r = (int64_t)((uint32_t)e[u].y + (uint32_t)__popcll(e[u].x & ((1ull << (idx[u] & 63)) - 1ull)));
          pr.out16[r] = rv[u];
        }
        continue;
      }
      for (int c = 0; c < cols.n; c++) {
#pragma unroll
        for (int u = 0; u < GP_U; u++) {
          if (!in[u]) continue;
          const int64_t i = base + u * BLOCK + threadIdx.x;
          const int64_t r = (int64_t)((uint32_t)e[u].y + (uint32_t)__popcll(e[u].x & ((1ull << (idx[u] & 63)) - 1ull)));
          switch (cols.width[c]) {
            case 16: reinterpret_cast<uint4*>(cols.dst[c])[r] = reinterpret_cast<const uint4*>(cols.src[c])[i]; break;
            case 8: reinterpret_cast<uint64_t*>(cols.dst[c])[r] = reinterpret_cast<const uint64_t*>(cols.src[c])[i]; break;
            case 4: reinterpret_cast<uint32_t*>(cols.dst[c])[r] = reinterpret_cast<const uint32_t*>(cols.src[c])[i]; break;
            default: reinterpret_cast<uint8_t*>(cols.dst[c])[r] = reinterpret_cast<const uint8_t*>(cols.src[c])[i]; break;
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------- JoinFilter (joins/join_filter.rs)
// apply_join_filter_to_indices (joins/utils.rs:1248-1318): key-equal pairs -> intermediate batch -> filter
// expression -> pairs whose value is TRUE.  Only passing pairs count as matches (visited bits, probe hits).
__global__ __launch_bounds__(BLOCK) void k_pair_tally(const int64_t* __restrict__ ob, const int64_t* __restrict__ op, const uint64_t* __restrict__ pass, int64_t m,
                                                     uint32_t* __restrict__ probe_hits, uint8_t* __restrict__ visited) {
  for (int64_t j = (int64_t)blockIdx.x * BLOCK + threadIdx.x; j < m; j += (int64_t)gridDim.x * BLOCK) {
    if (!bit_at(pass, j)) continue;
    atomicAdd(&probe_hits[op[j]], 1u);
    if (visited) visited[ob[j]] = 1;
  }
}
__global__ __launch_bounds__(BLOCK) void k_pairs_compact(const int64_t* __restrict__ ob, const int64_t* __restrict__ op, const uint64_t* __restrict__ pass,
                                                        const uint64_t* __restrict__ prefix, int64_t m, int64_t* __restrict__ ob2, int64_t* __restrict__ op2) {
  const int64_t n_words = (m + 63) >> 6;
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * BLOCK) >> 6;
  for (int64_t w = wave; w < n_words; w += n_waves) {
    uint64_t mk = pass[w];
    const int64_t rem = m - (w << 6);
    if (rem < 64) mk &= (~0ull) >> (64 - rem);
    if ((mk >> lane_id()) & 1ull) {
      const int64_t d = (int64_t)(prefix[w] + mbcnt(mk)), j = (w << 6) + lane_id();
      ob2[d] = ob[j];
      op2[d] = op[j];
    }
  }
}
// mask[p] = (probe row p has a passing pair) == want; bytes[p] (optional) = has a passing pair
__global__ __launch_bounds__(BLOCK) void k_hits_mask(const uint32_t* __restrict__ hits, int64_t np, int want, uint64_t* __restrict__ mask, uint8_t* __restrict__ bytes) {
  const int64_t n_words = (np + 63) >> 6;
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * BLOCK) >> 6;
  for (int64_t w = wave; w < n_words; w += n_waves) {
    const int64_t p = (w << 6) + lane_id();
    const bool hit = p < np && hits[p] != 0;
    if (bytes && p < np) bytes[p] = hit ? 1 : 0;
    const uint64_t word = ballot64(p < np && (hit == (want != 0)));
    if (lane_id() == 0) mask[w] = word;
  }
}
// row ids of the set bits -> idx[base + rank]; build[base + rank] = -1 (unmatched probe rows of Right / Full joins)
__global__ __launch_bounds__(BLOCK) void k_append_unmatched(const uint64_t* __restrict__ mask, const uint64_t* __restrict__ prefix, int64_t np, int64_t base,
                                                           int64_t* __restrict__ ob2, int64_t* __restrict__ op2) {
  const int64_t n_words = (np + 63) >> 6;
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * BLOCK) >> 6;
  for (int64_t w = wave; w < n_words; w += n_waves) {
    const uint64_t mk = mask[w];
    if ((mk >> lane_id()) & 1ull) {
      const int64_t d = base + (int64_t)(prefix[w] + mbcnt(mk));
      ob2[d] = -1;
      op2[d] = (w << 6) + lane_id();
    }
  }
}

// unmatched / matched build rows from the visited bytes (process_unmatched_build_batch, stream.rs:1002-)
__global__ __launch_bounds__(BLOCK) void k_visited_mask(const uint8_t* __restrict__ visited, int64_t n, int want_visited, uint64_t* __restrict__ mask) {
  const int64_t n_words = (n + 63) >> 6;
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * BLOCK) >> 6;
  for (int64_t w = wave; w < n_words; w += n_waves) {
    int64_t i = (w << 6) + lane_id();
    uint64_t word = ballot64(i < n && ((visited[i] != 0) == (want_visited != 0)));
    if (lane_id() == 0) mask[w] = word;
  }
}

// --------------------------------------------------------------------------------- host
// call f(std::integral_constant<int, KIND>) for the table kind chosen at build time
template <typename F>
static void with_kind(int kind, F&& f) {
  switch (kind) {
    case KIND_ARRAY: f(std::integral_constant<int, KIND_ARRAY>{}); break;
    case KIND_RANK: f(std::integral_constant<int, KIND_RANK>{}); break;
    case KIND_FLAT: f(std::integral_constant<int, KIND_FLAT>{}); break;
    case KIND_FLAT16: f(std::integral_constant<int, KIND_FLAT16>{}); break;
    default: f(std::integral_constant<int, KIND_HASH>{}); break;
  }
}
template <typename F>
static void with_key_type(int dfgpu_type, F&& f) {
  switch (dfgpu_type) {
    case DFGPU_INT32: case DFGPU_DATE32: f(std::integral_constant<int, KT_I32>{}); break;
    case DFGPU_UINT32: f(std::integral_constant<int, KT_U32>{}); break;
    case DFGPU_INT64: f(std::integral_constant<int, KT_I64>{}); break;
    case DFGPU_UINT8: f(std::integral_constant<int, KT_U8>{}); break;
    default: throw Error("key type has no direct-address instantiation");
  }
}
// f(KIND, KT): the hash table kind hashes any key set (KT_ANY); the direct-address kinds have one integer key
template <typename F>
static void with_kind_and_key(int kind, int probe_key_type, F&& f) {
  if (kind == KIND_HASH) {
    f(std::integral_constant<int, KIND_HASH>{}, std::integral_constant<int, KT_ANY>{});
    return;
  }
  if (kind == KIND_FLAT) {
    f(std::integral_constant<int, KIND_FLAT>{}, std::integral_constant<int, KT_ANY>{});
    return;
  }
  if (kind == KIND_FLAT16) {
    f(std::integral_constant<int, KIND_FLAT16>{}, std::integral_constant<int, KT_ANY>{});
    return;
  }
  if (kind == KIND_RETURNED) {
    f(std::integral_constant<int, KIND_RETURNED>{}, std::integral_constant<int, KT_ANY>{});
    return;
  }
  with_key_type(probe_key_type, [&](auto kt) {
    if (kind == KIND_ARRAY) f(std::integral_constant<int, KIND_ARRAY>{}, kt);
    else f(std::integral_constant<int, KIND_RANK>{}, kt);
  });
}

// The interleaved {bits, prefix} view of the rank map, made once by the first probe that asks every row for its rank (the fused probe
// kernel, the pairs paths, the grouped lookup): one 16-byte load per lookup there.  A selective probe — membership in the counts pass,
// ranks for the few listed rows — never builds it (SF300's Q3: 450 MB written and 0.18 ms per join table).
static void ensure_rank_tab(JoinTable& jt) {
  if (jt.kind != KIND_RANK) return;
  std::lock_guard<std::mutex> lk(jt.tab_mu);
  if (jt.rank_tab) return;
  Runtime& r = rt();
  const int64_t n_words = (int64_t)((jt.am_size - 1) >> 6) + 1;
  BufPtr tab = make_buf((size_t)n_words * 16);
  {
    ProfileScope ps("join_build_rank_interleave", n_words * 16);
    k_rank_interleave<<<grid_for(n_words, BLOCK), BLOCK, 0, r.stream>>>(jt.rank_bits->as<uint64_t>(), jt.rank_prefix->as<uint64_t>(), n_words, tab->as<ulonglong2>());
    DFGPU_HIP(hipGetLastError());
  }
  call_epilogue();  // several host threads: complete before another thread's stream reads it (one thread: its stream orders it, no wait)
  jt.rank_tab = tab;
  std::lock_guard<std::mutex> lk2(jt.mu);
  jt.info.table_bytes += n_words * 16;
}

// rank map over keys that are not in ascending row order: the rank -> row permutation, made once, by the first probe that needs
// build rows (several probe partitions may arrive together: CollectLeft)
static void ensure_rank_perm(JoinTable& jt) {
  if (jt.kind != KIND_RANK || !jt.rank_needs_perm) return;
  ensure_rank_tab(jt);
  std::lock_guard<std::mutex> lk(jt.mu);
  if (jt.rank_perm) return;
  Runtime& r = rt();
  const int64_t nb = jt.build.nrows;
  const KeySet ks = make_keyset(jt.build, jt.key_cols);
  BufPtr perm = make_buf((size_t)std::max<int64_t>(nb, 1) * 4);
  {
    ProfileScope ps("join_build_rank_perm", nb * (ks.c[0].width + 4));
    k_rank_perm_tab<<<grid_for(nb, BLOCK), BLOCK, 0, r.stream>>>(ks.c[0], nb, jt.am_offset, jt.rank_tab->as<ulonglong2>(), perm->as<uint32_t>());
    DFGPU_HIP(hipGetLastError());
  }
  DFGPU_HIP(hipStreamSynchronize(r.stream));  // complete before another thread's stream reads it
  jt.rank_perm = perm;
  jt.info.table_bytes += nb * 4;
}

static ProbeCtx make_ctx(JoinTable& jt, const Table& probe, const std::vector<int>& pk, bool need_build_rows = true, bool want_tab = true) {
  if (want_tab) ensure_rank_tab(jt);
  if (need_build_rows) ensure_rank_perm(jt);
  ProbeCtx c{};
  c.bkeys = make_keyset(jt.build, jt.key_cols);
  c.pkeys = make_keyset(probe, pk);
  for (int i = 0; i < c.bkeys.n; i++) {
    int bt = c.bkeys.c[i].type == DFGPU_DATE32 ? DFGPU_INT32 : c.bkeys.c[i].type;
    int pt = c.pkeys.c[i].type == DFGPU_DATE32 ? DFGPU_INT32 : c.pkeys.c[i].type;
    DFGPU_CHECK(bt == pt, "join key types differ between build and probe side (the planner inserts casts)");
  }
  c.heads = jt.heads ? jt.heads->as<uint32_t>() : nullptr;
  c.next = jt.next ? jt.next->as<uint32_t>() : nullptr;
  c.rank_tab = jt.rank_tab ? jt.rank_tab->as<ulonglong2>() : nullptr;
  c.rank_bits = jt.rank_bits ? jt.rank_bits->as<uint64_t>() : nullptr;
  c.rank_prefix = jt.rank_prefix ? jt.rank_prefix->as<uint64_t>() : nullptr;
  c.rank_perm = need_build_rows && jt.rank_perm ? jt.rank_perm->as<uint32_t>() : nullptr;
  c.am_offset = jt.am_offset;
  c.am_size = jt.am_size;
  c.hash_mask = jt.hash_mask;
  c.flat = jt.flat ? jt.flat->as<uint4>() : nullptr;
  c.flat_layout = jt.flat_layout;
  c.flat_shift = jt.flat_shift;
  c.flat_mask = jt.flat_mask;
  c.null_equals_null = jt.null_equality == DFGPU_NULL_EQUALS_NULL;
  c.force_collisions = jt.force_collisions;
  return c;
}

static Column mark_column(const uint8_t* bytes, int64_t n) {
  dfgpu_field f{};
  f.type = DFGPU_BOOL;
  Column c = alloc_column(f, "mark", n);
  pack_bytes_to_bitmap(bytes, n, c.data->as<uint64_t>());
  return c;
}

static int64_t null_count_of(const Column& c) {
  if (!c.validity) return 0;
  if (c.null_count >= 0) return c.null_count;
  Column tmp = c;
  count_nulls(tmp);
  return tmp.null_count;
}

// do the key columns of a row pack into 16 bytes (plus one bit per nullable column when NULL == NULL)?
static bool flat_layout_for(const KeySet& ks, bool null_equals_null, FlatLayout& L, bool& wide) {
  L = FlatLayout{};
  L.n = ks.n;
  int bytes = 0, null_bits = 0;
  for (int i = 0; i < ks.n; i++) {
    const int w = ks.c[i].width;
    if (w != 1 && w != 4 && w != 8 && w != 16) return false;
    if (bytes + w > 16) return false;
    L.off[i] = (uint8_t)bytes;
    L.width[i] = (uint8_t)w;
    L.null_bit[i] = -1;
    bytes += w;
    if (null_equals_null && ks.c[i].valid) null_bits++;
  }
  if (bytes * 8 + null_bits > 128) return false;
  int bit = bytes * 8;
  for (int i = 0; i < ks.n; i++)
    if (null_equals_null && ks.c[i].valid) L.null_bit[i] = (int8_t)bit++;
  wide = bit > 64;
  return true;
}
static std::unique_ptr<JoinTable> join_build_fixed_keys(const Table& build, const std::vector<int>& key_cols, int null_equality, const dfgpu_join_options& opts,
                                                        bool speculate = true);
// Utf8 key columns: interned on entry (ascending dictionary) into an extra column behind the caller's columns, which becomes the key —
// the string column itself stays where it is and travels as payload.  The probe side is interned the same way and its indices are
// rewritten into the build side's dictionary (with_build_dictionaries).
static std::unique_ptr<JoinTable> join_build(const Table& build, const std::vector<int>& key_cols, int null_equality, const dfgpu_join_options& opts) {
  Table coded;
  std::vector<int> kc = key_cols;
  bool any = false;
  for (size_t i = 0; i < kc.size(); i++) {
    DFGPU_CHECK(kc[i] >= 0 && kc[i] < (int)build.cols.size(), "join key column index out of range");
    const Column& c = build.cols[(size_t)kc[i]];
    if (c.field.type == DFGPU_BOOL) {   // a Boolean key: one byte per row behind the caller's columns (the Boolean column travels as payload)
      if (!any) coded = build;
      any = true;
      coded.cols.push_back(bool_as_u8(c, build.nrows));
      kc[i] = (int)coded.cols.size() - 1;
      continue;
    }
    if (c.field.type != DFGPU_UTF8 || c.dict) continue;
    if (!any) coded = build;
    any = true;
    coded.cols.push_back(dictionary_encode(c, true));
    kc[i] = (int)coded.cols.size() - 1;
  }
  return join_build_fixed_keys(any ? coded : build, kc, null_equality, opts);
}
static std::unique_ptr<JoinTable> join_build_fixed_keys(const Table& build, const std::vector<int>& key_cols, int null_equality, const dfgpu_join_options& opts,
                                                        bool speculate) {
  Runtime& r = rt();
  auto jt = std::make_unique<JoinTable>();
  jt->build = build;
  jt->key_cols = key_cols;
  jt->null_equality = null_equality;
  jt->force_collisions = opts.force_hash_collisions != 0;
  jt->probe_mode = opts.probe_mode;
  jt->null_aware = opts.null_aware != 0;
  if (jt->null_aware) {
    DFGPU_CHECK(key_cols.size() == 1,
                "null_aware anti join only supports single column join key, got " + std::to_string(key_cols.size()) + " columns");
    DFGPU_CHECK(key_cols[0] >= 0 && key_cols[0] < (int)build.cols.size(), "build key column out of range");
    jt->build_side_has_null = null_count_of(build.cols[key_cols[0]]) > 0;
  }
  const int64_t nb = build.nrows;
  DFGPU_CHECK(nb < 0xFFFFFFFFll, "build side has >= u32::MAX rows (the reference switches to JoinHashMapU64; not supported on GPU)");
  KeySet ks = make_keyset(build, key_cols);
  jt->info.build_rows = nb;

  if (opts.table_mode == 4) {
    // LDS-staged radix-partitioned table (radix_join.hip): any key set, duplicates, NULL == NULL
    jt->kind = KIND_RADIX;
    jt->radix = radix_join_build(build, key_cols, null_equality == DFGPU_NULL_EQUALS_NULL, jt->force_collisions);
    jt->keys_unique = false;  // not established: every probe takes the pairs path
    jt->info.table_bytes = radix_join_table_bytes(*jt->radix);
    jt->info.table_kind = KIND_RADIX;
    jt->info.build_keys_unique = 0;
    return jt;
  }
  // ---- key statistics: ArrayMap::try_new bounds (array_map.rs:175-203) + ascending-order check
  bool have_stats = false, ascending = false;
  uint64_t range = 0;
  long long kmin = 0;
  int64_t n_valid_keys = 0;
  // Large builds guess their statistics instead of measuring them first: keys that arrive in ascending order (a primary key in
  // table order: what dbgen, a sorted file or an ordered scan hands over) have min = the first key, max = the last, and a sample of
  // neighbours says whether that is plausible.  The rank map is then built in ONE pass over the keys that verifies the guess on the
  // way (k_rank_setbits<.., VERIFY>): 150 M keys, 0.31 ms of the step's 9.8.  A wrong guess costs the verifying pass and is
  // repaired by the measured path below (`speculate` = false).
  bool speculated = false;
  const bool spec_off = false;
  // Statistics this (immutable) key column already carries — measured by an earlier build over the same table, by dfgpu_column_minmax,
  // by a sort — are taken as they are: no sample, no verifying pass, no read-back (the reference's planner reads a table's column
  // statistics the same way, common/src/stats.rs; a build that has to measure or guess leaves what it learned on the column, below)
  std::shared_ptr<ColStats> known;
  if (ks.n == 1 && opts.table_mode != 1 && opts.table_mode != 5 && is_integer_like(ks.c[0].type) && ks.c[0].type != DFGPU_UINT64 && option_on("join.cached_key_stats", true)) {
    const Column& kc = build.cols[key_cols[0]];
    known = std::atomic_load(&kc.stats);
    const bool null_block = null_equality == DFGPU_NULL_EQUALS_NULL && kc.has_nulls();
    if (known && !null_block && nb > 0 && known->valid > 0) {
      range = (uint64_t)known->max - (uint64_t)known->min;
      have_stats = range != UINT64_MAX;
      kmin = known->min;
      ascending = known->ascending;
      n_valid_keys = known->valid;
    } else {
      known.reset();
    }
  }
  if (!known && speculate && !spec_off && (opts.table_mode == 0 || opts.table_mode == 3) && ks.n == 1 && is_integer_like(ks.c[0].type) && ks.c[0].type != DFGPU_UINT64 &&
      !ks.c[0].valid && nb >= (1 << 22)) {
    constexpr int S = 4096;
    BufPtr ends = make_zero_buf(24);
    with_key_type(ks.c[0].type, [&](auto kt) { k_key_ends<decltype(kt)::value><<<S / BLOCK, BLOCK, 0, r.stream>>>(ks.c[0], nb, nb / S, S, ends->as<long long>()); });
    long long e[3] = {0, 0, 1};
    d2h(e, ends->ptr, 24);
    const uint64_t grange = (uint64_t)e[1] - (uint64_t)e[0];
    if (e[2] == 0 && e[1] > e[0] && grange != UINT64_MAX && grange + 1 >= (uint64_t)nb && grange < (1ull << 40)) {
      const double gdense = (double)nb / ((double)grange + 1.0);
      if (grange < (uint64_t)opts.perfect_hash_join_small_build_threshold || gdense >= RANK_MAP_MIN_KEY_DENSITY || opts.table_mode == 3) {
        speculated = true;
        have_stats = true;
        ascending = true;
        range = grange;
        kmin = e[0];
        n_valid_keys = nb;
      }
    }
  }
  if (!known && !speculated && opts.table_mode != 1 && opts.table_mode != 5 && ks.n == 1 && is_integer_like(ks.c[0].type) && ks.c[0].type != DFGPU_UINT64) {
    const Column& kc = build.cols[key_cols[0]];
    bool null_block = null_equality == DFGPU_NULL_EQUALS_NULL && kc.has_nulls();
    if (!null_block && nb > 0) {
      BufPtr mm = make_buf(sizeof(MinMax));
      static const MinMax init{INT64_MAX, INT64_MIN, 0, 0, 0};
      h2d_async(mm->ptr, &init, sizeof init);
      {
        ProfileScope ps("join_build_key_stats", nb * ks.c[0].width);
        const int g = std::min(grid_for(nb, BLOCK * BUILD_UNROLL), 512);   // (see column_stats: the closing atomics)
        with_key_type(ks.c[0].type, [&](auto kt) {
          constexpr int T = decltype(kt)::value;
          if (ks.c[0].valid) k_key_minmax<T, true><<<g, BLOCK, 0, r.stream>>>(ks.c[0], nb, mm->as<MinMax>());
          else k_key_minmax<T, false><<<g, BLOCK, 0, r.stream>>>(ks.c[0], nb, mm->as<MinMax>());
        });
      }
      MinMax res;
      d2h(&res, mm->ptr, sizeof res);
      n_valid_keys = (int64_t)res.valid;
      if (res.valid > 0) {
        range = (uint64_t)res.smax - (uint64_t)res.smin;  // ArrayMap::calculate_range (wrapping)
        have_stats = range != UINT64_MAX;
        kmin = res.smin;
        ascending = res.unsorted == 0;
        // (what was measured stays with the column: the next build over this table reads it)
        std::atomic_store(&const_cast<Column&>(kc).stats, std::make_shared<ColStats>(ColStats{res.smin, res.smax, (int64_t)res.valid, res.unsorted == 0, res.descends == 0}));
      }
    }
  }
  const double dense = have_stats ? (double)nb / ((double)range + 1.0) : 0.0;
  // try_create_array_map gating (hash_join/exec.rs:111-191) with the caller's knob values
  bool am_ok = have_stats && !(range >= (uint64_t)opts.perfect_hash_join_small_build_threshold && dense <= opts.perfect_hash_join_min_key_density);
  if (opts.table_mode == 2) am_ok = have_stats;
  am_ok = am_ok && range < (1ull << 34);  // HBM guard: <= 64 GiB of u32 slots
  // rank map: 1/4 byte per value of the range, so it pays far below ArrayMap's density gate
  bool rank_ok = have_stats && (opts.table_mode == 0 || opts.table_mode == 3) && range < (1ull << 40) &&
                 (range < (uint64_t)opts.perfect_hash_join_small_build_threshold || dense >= RANK_MAP_MIN_KEY_DENSITY || opts.table_mode == 3);
  DFGPU_CHECK(!speculated || rank_ok, "internal: speculative build statistics outside the rank map's gate");
  DFGPU_CHECK(!(opts.table_mode == 2 && !am_ok), "direct-address join table requested but not applicable");
  DFGPU_CHECK(!(opts.table_mode == 3 && !rank_ok), "rank-map join table requested but not applicable");
  // Build keys that do NOT arrive in ascending order need the rank -> row permutation: one more dependent random access per
  // probe row, and once bitmap + directory + permutation outgrow the 256 MiB Infinity Cache every one of them is a 128-byte
  // line of HBM.  ArrayMap answers with ONE access then (measured, SF100 sizes, shuffled unique keys: 22.5 ms vs 37.5 ms,
  // profiles/r2_join_shapes_v2.md), so `auto` prefers it when the reference's own gating admits it.
  // (round 3: the permutation is built lazily, so the choice no longer has to be made blind — a key-only probe of the rank map
  // needs neither it nor ArrayMap's 4 bytes per VALUE of the range: 150 M shuffled keys x 600 M probes, key-only: 23.7 ms -> see
  // profiles/r3_join_shapes.md; probes that do gather build rows pay rank -> perm -> row, within 10 % of ArrayMap's two accesses.)
  constexpr int64_t MALL_BYTES = (int64_t)256 << 20;

  BufPtr flag = make_zero_buf(4);
  int dup = 0;
  bool flat_wide = false;
  // A build side that `auto` would hand to the LDS radix join once duplicates show up (below) need not build a rank map to see them
  // (1.5 + 0.9 ms of 23.5 for 150 M rows with every key three times): a sample of its rows says so in 30 microseconds when it is so.
  if (rank_ok && !ascending && !speculated && opts.table_mode == 0 && opts.probe_mode == 4 && ks.n == 1 && nb * 8 > MALL_BYTES && nb >= ((int64_t)1 << 22)) {
    const int64_t m = (int64_t)1 << 18;
    const uint64_t slots = (uint64_t)1 << 20;
    BufPtr table = make_zero_buf((size_t)slots * 8);
    {
      ProfileScope ps("join_build_sample_duplicates", m * 2 * ks.c[0].width);
      with_key_type(ks.c[0].type, [&](auto kt) {
        k_sample_duplicates<decltype(kt)::value><<<grid_for(2 * m, BLOCK), BLOCK, 0, r.stream>>>(ks.c[0], nb, nb / m, m, table->as<unsigned long long>(), slots - 1, flag->as<int>());
      });
      DFGPU_HIP(hipGetLastError());
    }
    int seen = 0;
    d2h(&seen, flag->ptr, 4);
    if (seen) {
      dup = 1;
      rank_ok = false;
      DFGPU_HIP(hipMemsetAsync(flag->ptr, 0, 4, r.stream));
    }
  }
  if (rank_ok) {
    const int64_t n_words = (int64_t)(range >> 6) + 1;
    jt->rank_bits = make_zero_buf((size_t)n_words * 8);
    {
      ProfileScope ps("join_build_rank_map", nb * ks.c[0].width);
      int g = std::min(grid_for(nb, BLOCK * BUILD_UNROLL), 2048);
      unsigned long long* bits = jt->rank_bits->as<unsigned long long>();
      // keys in no order over a bitmap beyond the caches: every atomicOr is a line of its own (150 M shuffled keys: 5.75 ms).  One
      // stable radix pass groups the keys by the top bits of their range first (sort.hip: 64 groups), the bits of a group
      // then land in ~1/64 of the bitmap
      KeyCol kc0 = ks.c[0];
      BufPtr grouped_keys;
      // (round 4) many keys in no order over a bitmap beyond the caches: grouped by key range, bits set in LDS
      constexpr int RB_BITS = 10;
      const bool lds_bits = !ascending && nb > (1 << 22) && n_words * 8 > ((int64_t)16 << 20) && (range >> RB_BITS) < (1ull << 20) - 64;
      if (lds_bits) {
        const GroupSpec gs = group_spec((uint64_t)kmin, range + 1, 1 << RB_BITS);
        GroupedRows gr = group_rows_by_key(kc0, nb, gs, RB_BITS, nullptr, true, false, {}, {}, "join_build_group_keys");
        const int P = 1 << RB_BITS;
        std::vector<uint64_t> gfirst((size_t)P + 1);
        uint64_t max_span = 0;
        for (int q = 0; q <= P; q++) {   // first value of group q: the smallest idx with floor(idx * mul / 2^64) >= q
          const unsigned __int128 v = ((((unsigned __int128)(unsigned)q) << 64) + gs.mul - 1) / gs.mul;
          gfirst[(size_t)q] = v > (unsigned __int128)(range + 1) || q == P ? range + 1 : (uint64_t)v;
          if (q) max_span = std::max(max_span, gfirst[(size_t)q] - gfirst[(size_t)q - 1]);
        }
        BufPtr d_first = make_buf(gfirst.size() * 8);
        h2d_async(d_first->ptr, gfirst.data(), gfirst.size() * 8);
        const size_t lds = (size_t)((max_span >> 6) + 2) * 8;
        DFGPU_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_rank_bits_grouped), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        k_rank_bits_grouped<<<P, RB_THREADS, lds, r.stream>>>(gr.keys->as<uint64_t>(), gr.bounds->as<uint64_t>(), d_first->as<uint64_t>(), (uint64_t)kmin, bits, flag->as<int>());
        DFGPU_HIP(hipGetLastError());
        DFGPU_HIP(hipStreamSynchronize(r.stream));   // gfirst is a local the copy reads
      } else {
      if (!ascending && !kc0.valid && kc0.width == 8 && nb > (1 << 22) && n_words * 8 > ((int64_t)16 << 20)) {
        const Column& kcol = build.cols[(size_t)key_cols[0]];
        grouped_keys = kcol.data;
        if (kcol.data_offset != 0) {
          grouped_keys = make_buf((size_t)nb * 8);
          DFGPU_HIP(hipMemcpyAsync(grouped_keys->ptr, kcol.ptr(), (size_t)nb * 8, hipMemcpyDeviceToDevice, r.stream));
        }
        int range_bits = 0;
        while (range_bits < 64 && ((range + 1) >> range_bits)) range_bits++;
        radix_group_keys(grouped_keys, nb, std::max(0, range_bits - 6), 6);
        kc0.data = grouped_keys->ptr;
      }
      // many keys in no order: byte map + pack instead of one atomic per key (the byte map is a temporary of one byte per VALUE of
      // the range: up to 4 GiB of it)
      const bool byte_map = !ascending && nb > (1 << 20) && range < (1ull << 32);
      if (byte_map) {
        BufPtr bytes = make_zero_buf((size_t)n_words * 64);
        with_key_type(kc0.type, [&](auto kt) {
          constexpr int T = decltype(kt)::value;
          if (kc0.valid) k_rank_setbytes<T, true><<<g, BLOCK, 0, r.stream>>>(kc0, nb, (uint64_t)kmin, bytes->as<uint8_t>());
          else k_rank_setbytes<T, false><<<g, BLOCK, 0, r.stream>>>(kc0, nb, (uint64_t)kmin, bytes->as<uint8_t>());
        });
        DFGPU_HIP(hipGetLastError());
        pack_bytes_to_bitmap(bytes->as<uint8_t>(), n_words * 64, jt->rank_bits->as<uint64_t>());
      } else {
      with_key_type(kc0.type, [&](auto kt) {
        constexpr int T = decltype(kt)::value;
        // ascending implies no NULL keys
        if (speculated) k_rank_setbits<T, false, true, true><<<g, BLOCK, 0, r.stream>>>(kc0, nb, (uint64_t)kmin, bits, flag->as<int>(), range);
        else if (ascending) k_rank_setbits<T, false, true><<<g, BLOCK, 0, r.stream>>>(kc0, nb, (uint64_t)kmin, bits, flag->as<int>());
        else if (kc0.valid) k_rank_setbits<T, true, false><<<g, BLOCK, 0, r.stream>>>(kc0, nb, (uint64_t)kmin, bits, flag->as<int>());
        else k_rank_setbits<T, false, false><<<g, BLOCK, 0, r.stream>>>(kc0, nb, (uint64_t)kmin, bits, flag->as<int>());
      });
      }
      }
    }
    jt->rank_prefix = make_buf((size_t)(n_words + 1) * 8);
    // A build side that fills its key range densely enough for the probes to look every row's rank up (below 0.15 the selective flavour
    // takes over: membership from the bitmap, ranks of the few listed rows from bitmap + prefix) gets the interleaved table from the
    // prefix scan's own down-sweep; the others leave it to the first probe that asks (ensure_rank_tab).
    BufPtr eager_tab;
    if (ascending && (double)nb >= 0.15 * ((double)range + 1.0) && option_on("join.eager_rank_tab", true)) {
      eager_tab = make_buf((size_t)n_words * 16);
      scan_bitmap_words_tab(jt->rank_bits->as<uint64_t>(), n_words, jt->rank_prefix->as<uint64_t>(), eager_tab->ptr);
    } else
    scan_mask_popcounts(jt->rank_bits->as<uint64_t>(), nullptr, n_words * 64, jt->rank_prefix->as<uint64_t>());
    // keys in no order set their bits without looking: two rows with one key set one bit
    if (!ascending) dup = (int64_t)read_u64(jt->rank_prefix->as<uint64_t>() + n_words) != n_valid_keys;
    if (speculated) {
      int wrong = 0;
      d2h(&wrong, flag->ptr, 4);
      if (wrong) {  // not ascending after all (or a key outside [first, last]): measure, then build
        ProfileScope ps("join_build_speculation_missed", 0);
        jt.reset();
        return join_build_fixed_keys(build, key_cols, null_equality, opts, false);
      }
      // the guess held for every key: first = min, last = max, strictly ascending, no NULLs — the column's statistics from now on
      std::atomic_store(&const_cast<Column&>(build.cols[key_cols[0]]).stats,
                        std::make_shared<ColStats>(ColStats{kmin, (long long)((uint64_t)kmin + range), nb, true, true}));
    }
    if (!dup) {
      jt->kind = KIND_RANK;
      jt->am_offset = (uint64_t)kmin;
      jt->am_size = range + 1;
      jt->rank_needs_perm = !ascending;  // the permutation itself waits for a probe that needs build rows (ensure_rank_perm)
      // (bitmap and prefix stay as they are: the membership-only passes read the bitmap, the listed emit both; the interleaved view
      // waits for a probe that wants it — ensure_rank_tab — unless the scan above already left it)
      jt->info.table_bytes = n_words * 16;
      if (eager_tab) {
        jt->rank_tab = eager_tab;
        jt->info.table_bytes += n_words * 16;
      }
    } else {
      DFGPU_CHECK(opts.table_mode != 3, "rank-map join table requested but the build keys are not unique");
      jt->rank_bits.reset();
      jt->rank_prefix.reset();
    }
  }
  // Duplicate build keys at a size where the chains live in HBM: when the plan does not observe the probe order
  // (probe_mode 4) the LDS radix join beats walking them (SF100 sizes, keys x3, M:N: 26.9 ms vs 34.1 ms ArrayMap, 59.5 ms
  // chained table)
  if (opts.table_mode == 0 && dup && opts.probe_mode == 4 && ks.n == 1 && nb * 8 > MALL_BYTES) {
    jt->rank_bits.reset();
    jt->kind = KIND_RADIX;
    jt->radix = radix_join_build(build, key_cols, null_equality == DFGPU_NULL_EQUALS_NULL, jt->force_collisions);
    jt->keys_unique = false;
    jt->info.table_bytes = radix_join_table_bytes(*jt->radix);
    jt->info.table_kind = KIND_RADIX;
    jt->info.build_keys_unique = 0;
    return jt;
  }
  if (jt->kind != KIND_RANK && am_ok) {
    jt->kind = KIND_ARRAY;
    jt->array_map = true;
    jt->am_offset = (uint64_t)kmin;
    jt->am_size = range + 1;
    jt->heads = make_zero_buf(jt->am_size * 4);
    auto am_build = [&](uint32_t* next) {
      const int g = std::min(grid_for(nb, BLOCK * BUILD_UNROLL), 2048);
      with_key_type(ks.c[0].type, [&](auto kt) {
        constexpr int T = decltype(kt)::value;
        if (ks.c[0].valid) k_am_build<T, true><<<g, BLOCK, 0, r.stream>>>(ks.c[0], nb, jt->am_offset, jt->heads->as<uint32_t>(), next, flag->as<int>());
        else k_am_build<T, false><<<g, BLOCK, 0, r.stream>>>(ks.c[0], nb, jt->am_offset, jt->heads->as<uint32_t>(), next, flag->as<int>());
      });
    };
    // duplicates already known (the rank map saw them): build the chains in the first pass
    if (dup) jt->next = make_zero_buf((size_t)nb * 4);
    {
      ProfileScope ps("join_build_array_map", nb * ks.c[0].width);
      am_build(dup ? jt->next->as<uint32_t>() : nullptr);
    }
    if (!dup) {
      d2h(&dup, flag->ptr, 4);
      if (dup && opts.table_mode == 0 && opts.probe_mode == 4 && ks.n == 1 && nb * 8 > MALL_BYTES) {
        // duplicates at a size where the chains would live in HBM, and the plan does not observe the probe order: LDS radix join
        jt->heads.reset();
        jt->array_map = false;
        jt->kind = KIND_RADIX;
        jt->radix = radix_join_build(build, key_cols, null_equality == DFGPU_NULL_EQUALS_NULL, jt->force_collisions);
        jt->keys_unique = false;
        jt->info.table_bytes = radix_join_table_bytes(*jt->radix);
        jt->info.table_kind = KIND_RADIX;
        jt->info.build_keys_unique = 0;
        return jt;
      }
      if (dup) {  // duplicates: rebuild with chains
        DFGPU_HIP(hipMemsetAsync(jt->heads->ptr, 0, jt->am_size * 4, r.stream));
        jt->next = make_zero_buf((size_t)nb * 4);
        ProfileScope ps("join_build_array_map", nb * ks.c[0].width);
        am_build(jt->next->as<uint32_t>());
      }
    }
    jt->info.table_bytes = (int64_t)jt->am_size * 4 + (dup ? nb * 4 : 0);
  } else if (jt->kind != KIND_RANK && opts.table_mode != 1 && flat_layout_for(ks, null_equality == DFGPU_NULL_EQUALS_NULL, jt->flat_layout, flat_wide)) {
    // the key columns pack into 16 bytes: hash table with the keys inline (one access per probe row, no re-check)
    jt->kind = flat_wide ? KIND_FLAT16 : KIND_FLAT;
    dup = 0;
    uint64_t cap = 64;
    int log2cap = 6;
    while (cap < (uint64_t)nb * 2) {
      cap <<= 1;
      log2cap++;
    }
    jt->flat_mask = cap - 1;
    jt->flat_shift = 64 - log2cap;
    jt->next = make_zero_buf((size_t)(nb ? nb : 1) * 4);
    jt->flat = make_buf((size_t)cap * (flat_wide ? 32 : 16));
    BufPtr owner = make_zero_buf((size_t)cap * 4);
    BufPtr more = make_zero_buf((size_t)(nb ? nb : 1) * 4);   // per owner row: rows with the same key behind it
    int64_t kb = 0;
    for (int i = 0; i < ks.n; i++) kb += nb * ks.c[i].width;
    const int nen = null_equality == DFGPU_NULL_EQUALS_NULL;
    {
      ProfileScope ps("join_build_flat_table", kb + nb * 4 + (int64_t)cap * (flat_wide ? 32 : 16));
      if (nb) {
        const int g = grid_for(nb, BLOCK);
        if (flat_wide) k_flat_claim<true><<<g, BLOCK, 0, r.stream>>>(ks, jt->flat_layout, nb, nen, jt->force_collisions, jt->flat_shift, jt->flat_mask, owner->as<uint32_t>(), jt->next->as<uint32_t>(), flag->as<int>(), more->as<uint32_t>());
        else k_flat_claim<false><<<g, BLOCK, 0, r.stream>>>(ks, jt->flat_layout, nb, nen, jt->force_collisions, jt->flat_shift, jt->flat_mask, owner->as<uint32_t>(), jt->next->as<uint32_t>(), flag->as<int>(), more->as<uint32_t>());
      }
      const int gf = grid_for((int64_t)cap, BLOCK);
      if (flat_wide) k_flat_fill<true><<<gf, BLOCK, 0, r.stream>>>(ks, jt->flat_layout, (int64_t)cap, nen, owner->as<uint32_t>(), more->as<uint32_t>(), jt->flat->as<uint4>());
      else k_flat_fill<false><<<gf, BLOCK, 0, r.stream>>>(ks, jt->flat_layout, (int64_t)cap, nen, owner->as<uint32_t>(), more->as<uint32_t>(), jt->flat->as<uint4>());
      DFGPU_HIP(hipGetLastError());
    }
    d2h(&dup, flag->ptr, 4);
    jt->info.table_bytes = (int64_t)cap * (flat_wide ? 32 : 16) + nb * 4;
  } else if (jt->kind != KIND_RANK) {
    DFGPU_CHECK(opts.table_mode != 5, "flat (inline-key) join table requested but the key columns do not pack into 16 bytes");
    jt->kind = KIND_HASH;
    dup = 0;
    uint64_t cap = 64;
    while (cap < (uint64_t)nb * 2) cap <<= 1;
    jt->hash_mask = cap - 1;
    jt->heads = make_zero_buf(cap * 4);
    jt->next = make_zero_buf((size_t)(nb ? nb : 1) * 4);
    if (nb) {
      int64_t kb = 0;
      for (int i = 0; i < ks.n; i++) kb += nb * ks.c[i].width;
      {
        ProfileScope ps("join_build_hash_map", kb + nb * 8);
        k_hm_build<<<grid_for(nb, BLOCK), BLOCK, 0, r.stream>>>(ks, nb, jt->hash_mask, null_equality == DFGPU_NULL_EQUALS_NULL, jt->force_collisions,
                                                                jt->heads->as<uint32_t>(), jt->next->as<uint32_t>());
      }
      ProfileScope ps("join_build_check_unique", kb + nb * 4);
      k_hm_check_unique<<<grid_for(nb, BLOCK), BLOCK, 0, r.stream>>>(ks, nb, null_equality == DFGPU_NULL_EQUALS_NULL, jt->next->as<uint32_t>(), flag->as<int>());
      d2h(&dup, flag->ptr, 4);
    }
    jt->info.table_bytes = (int64_t)cap * 4 + nb * 4;
  }
  DFGPU_HIP(hipGetLastError());
  jt->keys_unique = dup == 0;
  jt->info.used_array_map = jt->kind == KIND_ARRAY;
  jt->info.build_keys_unique = jt->keys_unique;
  jt->info.table_kind = jt->kind;
  jt->info.build_keys_ascending = ascending;
  return jt;
}

// Dictionary-encoded key columns join on their indices, which only means joining on the strings when both sides use the
// same dictionary.  A probe side encoded differently (another Parquet file, another table) gets its key indices rewritten
// into the build side's dictionary; probe strings the build dictionary does not hold get an index no build row carries.
// A Utf8 probe key (the build key was interned by join_build, or arrived dictionary-encoded) is interned into an extra column behind the
// caller's columns and `pk` is pointed at it: the string column keeps its place for the output.
static Table with_build_dictionaries(const JoinTable& jt, const Table& probe, std::vector<int>& pk) {
  DFGPU_CHECK(pk.size() == jt.key_cols.size(), "probe key count differs from build key count");
  Table fixed;
  bool changed = false;
  for (size_t i = 0; i < pk.size(); i++) {
    DFGPU_CHECK(pk[i] >= 0 && pk[i] < (int)probe.cols.size(), "join key column index out of range");
    const Column& bc = jt.build.cols[jt.key_cols[i]];
    if (probe.cols[pk[i]].field.type == DFGPU_BOOL) {   // the build side's Boolean key was widened to bytes by join_build: so is this one
      DFGPU_CHECK(bc.field.type == DFGPU_UINT8 && !bc.dict, "join key types differ between build and probe side (the planner inserts casts)");
      if (!changed) fixed = probe;
      changed = true;
      fixed.cols.push_back(bool_as_u8(probe.cols[pk[i]], probe.nrows));
      pk[i] = (int)fixed.cols.size() - 1;
      continue;
    }
    if (bc.dict && probe.cols[pk[i]].field.type == DFGPU_UTF8 && !probe.cols[pk[i]].dict) {
      Column enc = dictionary_encode(probe.cols[pk[i]], true);
      if (!same_dictionary(bc.dict, enc.dict)) {
        for (size_t k = 0; k < bc.dict->valid.size(); k++) DFGPU_CHECK(bc.dict->valid[k], "join keys with NULL dictionary values and different dictionaries are not supported on the GPU path");
        enc = remap_to_dictionary(enc, bc.dict);
      }
      if (!changed) fixed = probe;
      changed = true;
      fixed.cols.push_back(enc);
      pk[i] = (int)fixed.cols.size() - 1;
      continue;
    }
    const Column& pc = probe.cols[pk[i]];
    DFGPU_CHECK((bc.dict != nullptr) == (pc.dict != nullptr), "join key " + std::to_string(i) + ": one side is dictionary-encoded, the other is not (the planner inserts casts)");
    if (!bc.dict || same_dictionary(bc.dict, pc.dict)) continue;
    for (size_t k = 0; k < bc.dict->valid.size(); k++) DFGPU_CHECK(bc.dict->valid[k], "join keys with NULL dictionary values and different dictionaries are not supported on the GPU path");
    for (size_t k = 0; k < pc.dict->valid.size(); k++) DFGPU_CHECK(pc.dict->valid[k], "join keys with NULL dictionary values and different dictionaries are not supported on the GPU path");
    if (!changed) fixed = probe;  // shallow: columns share their buffers
    changed = true;
    fixed.cols[pk[i]] = remap_to_dictionary(pc, bc.dict);
  }
  return changed ? fixed : probe;
}

// the calling thread moves to the device the join table lives on (dfgpu.h: handles carry their device)
static JoinTable* unwrap_join(dfgpu_join_t ht) {
  DFGPU_CHECK(ht != nullptr, "null join handle");
  JoinTable* jt = reinterpret_cast<JoinTable*>(ht);
  if (jt->build.device >= 0) use_device(jt->build.device);
  return jt;
}

ColStats column_stats(Column& kc, int64_t nrows) {
  DFGPU_CHECK(is_integer_like(kc.field.type) && kc.field.type != DFGPU_UINT64, "dfgpu_column_minmax: integer columns only");
  // computed before for these rows (tables are immutable).  The cache slot is read and written atomically: two threads that read
  // the same table at once (the same statistics, computed twice at worst) must not tear it.
  if (auto cached = std::atomic_load(&kc.stats)) return *cached;
  Runtime& r = rt();
  MinMax res{INT64_MAX, INT64_MIN, 0, 0, 0};
  if (nrows > 0) {
    KeyCol k{kc.ptr(), kc.valid_words(), kc.field.type, type_width(kc.field.type)};
    BufPtr mm = make_buf(sizeof(MinMax));
    h2d_async(mm->ptr, &res, sizeof res);
    {
      ProfileScope ps("column_minmax", nrows * k.width);
      // (every workgroup ends with one set of five atomics on ONE line, ~12 ns each when contended: 2048 workgroups spent 0.1 ms on them
      // alone — the whole pass over a 9 M-row column; two workgroups per CU stream just as fast)
      const int g = std::min(grid_for(nrows, BLOCK * BUILD_UNROLL), 512);
      with_key_type(k.type, [&](auto kt) {
        constexpr int T = decltype(kt)::value;
        if (k.valid) k_key_minmax<T, true><<<g, BLOCK, 0, r.stream>>>(k, nrows, mm->as<MinMax>());
        else k_key_minmax<T, false><<<g, BLOCK, 0, r.stream>>>(k, nrows, mm->as<MinMax>());
      });
      DFGPU_HIP(hipGetLastError());
    }
    d2h(&res, mm->ptr, sizeof res);
    auto fresh = std::make_shared<ColStats>(ColStats{res.smin, res.smax, (int64_t)res.valid, res.valid > 0 && res.unsorted == 0, res.valid > 0 && res.descends == 0});
    std::atomic_store(&kc.stats, fresh);
    return *fresh;
  }
  return ColStats{res.smin, res.smax, 0, false, false};
}

static bool needs_visited(int join_type) {
  return join_type == DFGPU_JOIN_LEFT || join_type == DFGPU_JOIN_FULL || join_type == DFGPU_JOIN_LEFT_SEMI || join_type == DFGPU_JOIN_LEFT_ANTI ||
         join_type == DFGPU_JOIN_LEFT_MARK;
}

// Do neighbouring probe rows look up neighbouring keys?  A direct-address table bigger than the caches answers a CLUSTERED probe
// (foreign keys in the parent's order: lineitem -> orders) from lines it has just fetched, and a random one with one 64-byte unit
// of Infinity Cache / HBM traffic per row — which decides between probe flavours below.  From the column's cached statistics when
// they exist (a nondecreasing column is clustered), else from 4096 evenly spaced neighbour pairs: clustered = nine in ten of them
// lie within `window` key values of each other (the stretch of the table an L2 keeps without effort).  Round 3 asked whether
// the neighbours ASCEND, which rows sorted by another column with many ties do — runs of a few hundred ascending keys that jump
// all over the table (SF100 lineitem ordered by l_extendedprice: 99 % ascents, no locality).
template <int KT>
__global__ void k_sample_near(KeyCol k, int64_t n, int64_t every, int samples, uint64_t window, int* __restrict__ near) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= samples) return;
  const int64_t r = (int64_t)i * every;
  if (r + 1 >= n) return;
  const uint64_t a = load_key<KT>(k, r), b = load_key<KT>(k, r + 1);
  const uint64_t d = (int64_t)b >= (int64_t)a ? b - a : a - b;
  if (d <= window) atomicAdd(near, 1);
}
static bool probe_keys_clustered(const Column& kc, int64_t n, uint64_t window) {
  if (auto cached = std::atomic_load(&kc.stats))
    if (cached->nondecreasing) return true;
  // (a column with NULLs is sampled like any other: the values under its NULLs are few and say nothing either way)
  if (n < (1 << 16) || !is_integer_like(kc.field.type) || kc.field.type == DFGPU_UINT64) return true;
  // asked before of these rows with this window?  (the buffer remembers one answer: DevBuf::hint)
  const uint64_t tag = (fmix64((uint64_t)kc.data_offset * 0x9E3779B97F4A7C15ull ^ (uint64_t)n ^ (window << 20)) & ~3ull) | 2ull;
  if (kc.data) {
    const uint64_t h = kc.data->hint.load(std::memory_order_relaxed);
    if ((h & ~1ull) == tag) return (h & 1ull) != 0;
  }
  constexpr int S = 4096;
  BufPtr cnt = make_zero_buf(4);
  const KeyCol k{kc.ptr(), nullptr, kc.field.type, type_width(kc.field.type)};
  with_key_type(k.type, [&](auto kt) { k_sample_near<decltype(kt)::value><<<S / BLOCK, BLOCK, 0, rt().stream>>>(k, n, n / S, S, window, cnt->as<int>()); });
  int near = 0;
  d2h(&near, cnt->ptr, 4);
  const bool clustered = near * 10 >= S * 9;
  if (kc.data) kc.data->hint.store(tag | (clustered ? 1ull : 0ull), std::memory_order_relaxed);
  return clustered;
}

// do 64 K evenly spaced probe rows ALL find their key?  (the speculation of the placed probe is only worth trying then)
template <int KIND, int KT>
__global__ __launch_bounds__(BLOCK) void k_sample_all_hit(ProbeCtx c, int64_t np, int64_t every_words, int samples, int* __restrict__ misses) {
  const int64_t n_words = (np + 63) >> 6;
  const int i = (int)(((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6);
  if (i >= samples) return;
  const int64_t w = (int64_t)i * every_words;
  if (w >= n_words) return;
  uint64_t word[1];
  hit_words<KIND, KT, 1>(c, w, np, word);
  const int64_t rem = np - (w << 6);
  const uint64_t want = rem >= 64 ? ~0ull : ((1ull << rem) - 1ull);
  if (lane_id() == 0 && (word[0] & want) != want) atomicAdd(misses, 1);
}

// ---- grouped probe with positional return: host side
// groups of a rank map's key range: enough of them that a group's slice of the table plus of the build payload stays inside an
// XCD's 4 MiB L2 with room for the streams passing through (2.5 MB aimed at), at most 2^10 (a run of a tile is 8192 / groups rows)
static int gp_bits_for(const JoinTable& jt, int64_t payload_bytes_per_row) {
  if (const int64_t e = option_int("join.grouped_bits", 0)) return std::min(10, std::max(1, (int)e));  // (test hook)
  const double slice = (double)((jt.am_size >> 6) + 1) * 16.0 + (double)jt.build.nrows * (double)payload_bytes_per_row;
  int bits = 6;
  while (bits < 10 && slice / (double)(1 << bits) > 2.5 * 1048576.0) bits++;
  return bits;
}
static GroupSpec gp_spec_for(const JoinTable& jt, int nbits) { return group_spec(jt.am_offset, jt.am_size, 1 << nbits); }
// rank map over build keys in no order: the build columns `cols` copied into RANK order, once per join table and column (a probe
// in group order reads build payload at rank positions; through rank -> row -> column it would be a random line per row)
// the packed form: `cols` (<= 12 bytes together) as one 16-byte record per rank, field i at byte rec_off[i] (word 0 free)
static BufPtr ensure_rank_records(JoinTable& jt, const std::vector<int>& cols, const std::vector<int>& rec_off) {
  ensure_rank_tab(jt);   // (before jt.mu is taken: it takes its own lock, then jt.mu)
  std::lock_guard<std::mutex> lk(jt.mu);
  auto it = jt.rank_records.find(cols);
  if (it != jt.rank_records.end()) return it->second;
  Runtime& r = rt();
  const int64_t nb = jt.build.nrows;
  const KeySet ks = make_keyset(jt.build, jt.key_cols);
  std::vector<const void*> src;
  std::vector<int> width;
  int64_t bytes = 0;
  for (int c : cols) {
    const Column& col = jt.build.cols[(size_t)c];
    src.push_back(col.ptr());
    width.push_back(type_width(col.field.type));
    bytes += width.back();
  }
  const int nbits = gp_bits_for(jt, 16);
  GroupedRows gr = group_rows_by_key(ks.c[0], nb, gp_spec_for(jt, nbits), nbits, nullptr, true, false, src, width, "join_build_group_payload");
  BufPtr recs = make_buf((size_t)std::max<int64_t>(nb, 1) * 16);
  GroupCols gc{};
  PlaceRec pr{};
  gc.n = (int)src.size();
  for (int q = 0; q < gc.n; q++) {
    gc.src[q] = gr.cols[(size_t)q]->ptr;
    gc.dst[q] = nullptr;
    gc.width[q] = width[(size_t)q];
    pr.off[q] = rec_off[(size_t)q];
  }
  pr.out16 = recs->as<uint4>();
  {
    ProfileScope ps("join_build_rank_payload", gr.rows * (8 + bytes + 16));
    BufPtr tickets = make_zero_buf(8 * 32 * 4);
    k_gp_place<<<r.num_cus * 8, BLOCK, 0, r.stream>>>(jt.rank_tab->as<ulonglong2>(), jt.am_offset, gr.keys->as<uint64_t>(), gr.bounds->as<uint64_t>(), gr.P, gc,
                                                      tickets->as<unsigned>(), pr);
    DFGPU_HIP(hipGetLastError());
  }
  DFGPU_HIP(hipStreamSynchronize(r.stream));  // complete before another thread's stream reads the records
  jt.rank_records[cols] = recs;
  jt.info.table_bytes += nb * 16;
  return recs;
}
static void ensure_rank_payload(JoinTable& jt, const std::vector<int>& cols) {
  ensure_rank_tab(jt);   // (before jt.mu is taken: it takes its own lock, then jt.mu)
  std::lock_guard<std::mutex> lk(jt.mu);
  std::vector<int> missing;
  for (int c : cols)
    if (!jt.rank_payload.count(c) && std::find(missing.begin(), missing.end(), c) == missing.end()) missing.push_back(c);
  if (missing.empty()) return;
  Runtime& r = rt();
  const int64_t nb = jt.build.nrows;
  const KeySet ks = make_keyset(jt.build, jt.key_cols);
  for (size_t a = 0; a < missing.size(); a += GP_MAX_COLS) {
    std::vector<const void*> src;
    std::vector<int> width;
    int64_t bytes = 0;
    for (size_t q = a; q < std::min(missing.size(), a + GP_MAX_COLS); q++) {
      const Column& c = jt.build.cols[(size_t)missing[q]];
      src.push_back(c.ptr());
      width.push_back(type_width(c.field.type));
      bytes += width.back();
    }
    const int nbits = gp_bits_for(jt, bytes);
    GroupedRows gr = group_rows_by_key(ks.c[0], nb, gp_spec_for(jt, nbits), nbits, nullptr, true, false, src, width, "join_build_group_payload");
    GroupCols gc{};
    gc.n = (int)src.size();
    for (int q = 0; q < gc.n; q++) {
      BufPtr dst = make_buf((size_t)std::max<int64_t>(nb, 1) * width[(size_t)q]);
      gc.src[q] = gr.cols[(size_t)q]->ptr;
      gc.dst[q] = dst->ptr;
      gc.width[q] = width[(size_t)q];
      jt.rank_payload[missing[a + (size_t)q]] = dst;
      jt.info.table_bytes += nb * width[(size_t)q];
    }
    ProfileScope ps("join_build_rank_payload", gr.rows * (8 + 2 * bytes));
    BufPtr tickets = make_zero_buf(8 * 32 * 4);
    k_gp_place<<<r.num_cus * 8, BLOCK, 0, r.stream>>>(jt.rank_tab->as<ulonglong2>(), jt.am_offset, gr.keys->as<uint64_t>(), gr.bounds->as<uint64_t>(), gr.P, gc,
                                                      tickets->as<unsigned>(), PlaceRec{});
    DFGPU_HIP(hipGetLastError());
  }
  DFGPU_HIP(hipStreamSynchronize(r.stream));  // complete before another thread's stream reads the copies
}
struct ReturnedProbe {
  BufPtr dest, rec;
  int R = 0;
  int64_t hits = 0;
  std::vector<int> mul, add;  // per build output column: element index = dest * mul + add
};
// steps 1 and 2 of the grouped probe; false = not applicable (the records would be wider than 64 bytes)
static bool grouped_probe_lookup(JoinTable& jt, const Table& probe, int pk0, const std::vector<int>& bout, const uint64_t* row_mask, ReturnedProbe& rp) {
  ensure_rank_tab(jt);   // (before jt.mu is taken: it takes its own lock, then jt.mu)
  Runtime& r = rt();
  const int64_t np = probe.nrows;
  // record layout: the match id in the first word (KIND_RETURNED reads it there), then the 4-byte build columns, the 8-byte ones,
  // the 16-byte ones — every field aligned to its own width — and the single bytes at the end
  RetLayout L{};
  std::vector<int> field_off(bout.size(), 0);
  rp.mul.assign(bout.size(), 0);
  rp.add.assign(bout.size(), 0);
  int off = 4;
  int64_t payload = 0;
  for (int w : {4, 8, 16, 1})
    for (size_t i = 0; i < bout.size(); i++)
      if (type_width(jt.build.cols[(size_t)bout[i]].field.type) == w) {
        off = (off + w - 1) / w * w;
        field_off[i] = off;
        off += w;
        payload += w;
      }
  L.R = off <= 4 ? 4 : off <= 8 ? 8 : (off + 15) / 16 * 16;
  if (L.R > 64 || (int)bout.size() > MAX_JOIN_COLS) return false;
  L.n = (int)bout.size();
  // build keys in no order: the payload is read at rank positions from a rank-ordered copy — ONE 16-byte record per rank when the
  // returned record is 16 bytes (the build columns then cost the lookup one L2 access, not one per column), else column by column
  const bool packed = jt.rank_needs_perm && !bout.empty() && L.R == 16;
  BufPtr packed_recs;
  if (packed) packed_recs = ensure_rank_records(jt, bout, field_off);
  else if (jt.rank_needs_perm && !bout.empty()) ensure_rank_payload(jt, bout);
  L.packed16 = packed ? packed_recs->as<uint4>() : nullptr;
  std::unique_lock<std::mutex> lk(jt.mu);
  for (size_t i = 0; i < bout.size(); i++) {
    const Column& c = jt.build.cols[(size_t)bout[i]];
    const int w = type_width(c.field.type);
    L.src[i] = packed ? nullptr : jt.rank_needs_perm ? jt.rank_payload.at(bout[i])->ptr : c.ptr();
    L.width[i] = w;
    L.off[i] = field_off[i];
    rp.mul[i] = L.R / w;
    rp.add[i] = field_off[i] / w;
  }
  lk.unlock();
  rp.R = L.R;
  const Column& kc = probe.cols[(size_t)pk0];
  const KeyCol key{kc.ptr(), kc.valid_words(), kc.field.type, type_width(kc.field.type)};
  const int nbits = gp_bits_for(jt, payload);
  GroupedRows gr = group_rows_by_key(key, np, gp_spec_for(jt, nbits), nbits, row_mask, true, true, {}, {}, "join_probe_group_keys");
  rp.dest = gr.dest;
  rp.rec = make_buf((size_t)std::max<int64_t>(gr.rows, 1) * (size_t)L.R);
  BufPtr total = make_zero_buf(8);
  if (gr.rows) {
    ProfileScope ps("join_probe_grouped_lookup", gr.rows * (8 + L.R + payload));
    BufPtr tickets = make_zero_buf(8 * 32 * 4);
    k_gp_lookup<<<r.num_cus * 8, BLOCK, 0, r.stream>>>(jt.rank_tab->as<ulonglong2>(), jt.am_offset, jt.am_size, gr.keys->as<uint64_t>(), gr.bounds->as<uint64_t>(), gr.P, L,
                                                       rp.rec->as<uint8_t>(), total->as<unsigned long long>(), tickets->as<unsigned>());
    DFGPU_HIP(hipGetLastError());
  }
  rp.hits = (int64_t)read_u64(total->as<uint64_t>());
  return true;
}

static Table join_probe_with_filter(JoinTable& jt, const Table& probe, const std::vector<int>& pk, int join_type, const std::vector<int>& bout,
                                    const std::vector<int>& pout, const dfgpu_join_filter* jfp);
// `lazy` (optional, with row_mask == nullptr): a FilterExec below the probe side that has NOT been evaluated yet — `pred` is its
// predicate in the form the counts pass can evaluate per row, `mask()` evaluates it into a row mask for every other flavour.
struct LazyRowFilter {
  RowPred pred;
  int64_t pred_bytes_per_row;
  std::function<const uint64_t*()> mask;
};
static Table join_probe(JoinTable& jt, const Table& probe, const std::vector<int>& pk, int join_type, const std::vector<int>& bout_in,
                        const std::vector<int>& pout, const uint64_t* row_mask = nullptr, bool* mask_consumed = nullptr, const LazyRowFilter* lazy = nullptr) {
  Runtime& r = rt();
  if (row_mask) lazy = nullptr;
  auto need_mask = [&]() {   // a flavour that cannot evaluate the predicate itself: the mask after all
    if (lazy) {
      row_mask = lazy->mask();
      lazy = nullptr;
    }
  };
  // RightSemi / RightAnti emit probe columns only (joins/utils.rs:1628,1576): build_out_cols given by a C-ABI caller are ignored
  // on every path — the fused kernel would otherwise gather build rows of unmatched (anti) probe rows
  const std::vector<int> bout = (join_type == DFGPU_JOIN_RIGHT_SEMI || join_type == DFGPU_JOIN_RIGHT_ANTI) ? std::vector<int>{} : bout_in;
  DFGPU_CHECK(pk.size() == jt.key_cols.size(), "probe key count differs from build key count");
  DFGPU_CHECK(probe.device == jt.build.device, "the probe table lives on another device than the join table (move it with dfgpu_table_copy_to_device)");
  if (jt.kind == KIND_RADIX) {
    // the LDS radix join produces key-equal pairs; every join type is finished from them (row masks: the caller filters first)
    if (row_mask || lazy) {
      DFGPU_CHECK(mask_consumed != nullptr, "probe row mask without a fallback");
      *mask_consumed = false;
      return Table{};
    }
    if (join_type == DFGPU_JOIN_INNER && !jt.null_aware && pk.size() == 1) {
      // INNER, fixed-width payload without NULLs: the emit walk writes the output columns itself (radix_join.hip, round 6)
      for (int c : bout) DFGPU_CHECK(c >= 0 && c < (int)jt.build.cols.size(), "build output column out of range");
      for (int c : pout) DFGPU_CHECK(c >= 0 && c < (int)probe.cols.size(), "probe output column out of range");
      Table fused;
      if (radix_join_inner_columns(*jt.radix, jt.build, jt.key_cols, probe, pk, bout, pout, fused)) {
        std::lock_guard<std::mutex> lk(jt.mu);
        jt.info.probe_rows += probe.nrows;
        jt.info.output_rows += fused.nrows;
        return fused;
      }
    }
    return join_probe_with_filter(jt, probe, pk, join_type, bout, pout, nullptr);
  }
  const int64_t np = probe.nrows;
  const int64_t n_words = (np + 63) / 64;
  int kind = jt.kind;  // what the probe kernels look the keys up in: the table, or (KIND_RETURNED) what a grouped lookup left per probe row
  ProbeCtx ctx = make_ctx(jt, probe, pk, /*need_build_rows=*/false, /*want_tab=*/false);  // both decided below, once the probe flavour is known
  auto need_tab = [&]() {
    if (jt.kind == KIND_RANK) {
      ensure_rank_tab(jt);
      ctx.rank_tab = jt.rank_tab->as<ulonglong2>();
    }
  };
  for (int c : bout) DFGPU_CHECK(c >= 0 && c < (int)jt.build.cols.size(), "build output column out of range");
  for (int c : pout) DFGPU_CHECK(c >= 0 && c < (int)probe.cols.size(), "probe output column out of range");
  uint8_t* visited = nullptr;
  if (needs_visited(join_type)) {
    {
      std::lock_guard<std::mutex> lk(jt.mu);
      if (!jt.visited) {
        jt.visited = make_zero_buf((size_t)jt.build.nrows + 64);
        DFGPU_HIP(hipStreamSynchronize(rt().stream));  // zeroed before any other thread's stream marks rows in it
      }
    }
    visited = jt.visited->as<uint8_t>();
  }
  {
    std::lock_guard<std::mutex> lk(jt.mu);
    jt.info.probe_rows += np;
  }
  int64_t key_bytes = 0;
  for (int i = 0; i < ctx.pkeys.n; i++) key_bytes += np * ctx.pkeys.c[i].width;
  Table out;

  const bool probe_side_only = join_type == DFGPU_JOIN_RIGHT_SEMI || join_type == DFGPU_JOIN_RIGHT_ANTI;
  const bool build_side_only = join_type == DFGPU_JOIN_LEFT_SEMI || join_type == DFGPU_JOIN_LEFT_ANTI || join_type == DFGPU_JOIN_LEFT_MARK;
  bool payload_nullable = false;
  // (a Utf8 payload column takes the same route as a nullable one: pairs, then take — strings.hip gather_strings)
  for (int c : bout) payload_nullable |= jt.build.cols[c].has_nulls() || jt.build.cols[c].field.type == DFGPU_UTF8;
  for (int c : pout) payload_nullable |= probe.cols[c].has_nulls() || probe.cols[c].field.type == DFGPU_UTF8;
  const bool fast_inner = join_type == DFGPU_JOIN_INNER && jt.keys_unique && !payload_nullable &&
                          (int)(bout.size() + pout.size()) <= MAX_JOIN_COLS;

  // single-pass flavour: output columns are allocated for the upper bound (np rows) because the
  // row count is only known when the kernel ends; HBM is sized for that (288 GB)
  int64_t out_row_bytes = 0;
  if (!payload_nullable) {
    for (int c : bout) out_row_bytes += type_width(jt.build.cols[c].field.type);
    for (int c : pout) out_row_bytes += type_width(probe.cols[c].field.type);
  }
  const bool fused_ok = np < (1ll << 40) && !payload_nullable && (int)(bout.size() + pout.size()) <= MAX_JOIN_COLS &&
                        bout.size() + pout.size() > 0 && (fast_inner || probe_side_only);
  // probe_mode: 0 / 1 = output in probe order like the reference (exec.rs:3349), exact allocation: the PLACED flavour (tile
  // counts -> scan -> the fused kernel with known tile offsets) when it applies, else lookup -> scan -> materialise;
  // 2 = single pass ordered by decoupled look-back, 3 = single pass unordered (errors when not applicable);
  // 4 = a planner's hint "no ancestor needs the probe order": single pass unordered when applicable, the general path otherwise
  const bool want_single = jt.probe_mode == 2 || jt.probe_mode == 3 || jt.probe_mode == 4;
  const bool use_fused = fused_ok && np > 0;
  // the general M:N path in one pass (k_probe_pairs_single): flat tables under the planner's hint that nobody observes the order
  const bool single_pass_pairs = !use_fused && join_type == DFGPU_JOIN_INNER && jt.probe_mode == 4 && (jt.kind == KIND_FLAT || jt.kind == KIND_FLAT16) &&
                                 !row_mask && !lazy && np >= (1 << 16) && np < 0xFFFFFFFFll &&
                                 true;
  // A probe whose output holds no build column and that marks no build row only asks whether the key is THERE: over a rank map of
  // keys in no particular order it needs the bitmap alone, not the rank -> row permutation (which is built by the first probe
  // that does need rows)
  const bool rows_unused = bout.empty() && !needs_visited(join_type) && (use_fused || probe_side_only);
  // what counts as "beyond the caches": 4 x the L2 of an XCD (16 MiB on MI355X), Policy::beyond_cache_bytes
  const int64_t big_bytes = option_int("join.beyond_cache_bytes", (int64_t)policy().beyond_cache_bytes());
  const bool big_table = (jt.kind == KIND_RANK || jt.kind == KIND_ARRAY) && (int64_t)(jt.kind == KIND_RANK ? (jt.am_size >> 6) * 16 : jt.am_size * 4) > big_bytes;
  static thread_local bool in_grouped_probe = false;  // the probe over keys this function grouped itself: clustered by construction
  // from how many probe rows grouping them first is considered: the extra move has to repay itself (Policy::rows_worth_a_pass)
  const int64_t grouped_min_rows = option_int("join.grouped_min_rows", policy().rows_worth_a_pass());
  const bool unclustered = fused_ok && big_table && np > grouped_min_rows && pk.size() == 1 && !in_grouped_probe &&
                           !probe_keys_clustered(probe.cols[(size_t)pk[0]], np,
                                                 option_int("join.near_window", 0) ? (uint64_t)option_int("join.near_window", 0)   // (test hook)
                                                 : jt.kind == KIND_RANK ? ((uint64_t)512 << 10) / 16 * 64 : ((uint64_t)512 << 10) / 4);
  const bool group_env = option_on("join.grouped_probe", true);  // (A/B switch)
  // the key-only probe whose order nobody observes: its keys are grouped and probed in group order (below)
  const bool grouped_keys_only = unclustered && group_env && jt.probe_mode == 4 && rows_unused && !row_mask && !lazy && pout.size() == 1 && pout[0] == pk[0] &&
                                 ctx.pkeys.c[0].width == 8 && !probe.cols[(size_t)pk[0]].validity;
  // every other unclustered probe of a big rank map: the keys travel to the table group by group and what they found comes back
  // to the probe rows (k_gp_lookup above); the build rows are read at rank positions, so no rank -> row permutation is needed
  const bool returned_env = true;
  bool returned = use_fused && unclustered && group_env && returned_env && !grouped_keys_only && jt.kind == KIND_RANK && jt.probe_mode != 2 &&
                  is_integer_like(probe.cols[(size_t)pk[0]].field.type);
  if (trace_on("join"))   // one line per probe on stderr: what decided the probe flavour
    fprintf(stderr, "[dfgpu join_probe] np=%lld kind=%d probe_mode=%d fused_ok=%d big_table=%d unclustered=%d grouped_keys_only=%d returned=%d rows_unused=%d row_mask=%d\n",
            (long long)np, jt.kind, jt.probe_mode, (int)fused_ok, (int)big_table, (int)unclustered, (int)grouped_keys_only, (int)returned, (int)rows_unused, row_mask != nullptr);
  ReturnedProbe rp;
  if (returned) need_mask();
  if (returned) returned = grouped_probe_lookup(jt, probe, pk[0], bout, row_mask, rp);
  if (returned) {
    kind = KIND_RETURNED;
    ctx.ret_dest = rp.dest->as<uint32_t>();
    ctx.ret_rec = rp.rec->as<uint8_t>();
    ctx.ret_R = rp.R;
  }
  if (!returned && !rows_unused && jt.kind == KIND_RANK && jt.rank_needs_perm) {
    ensure_rank_perm(jt);
    ctx.rank_perm = jt.rank_perm->as<uint32_t>();
    ctx.rank_tab = jt.rank_tab->as<ulonglong2>();
  }
  DFGPU_CHECK(!((jt.probe_mode == 2 || jt.probe_mode == 3) && !fused_ok),
              "single-pass probe requested but not applicable (needs <=1 match per probe row and non-nullable payload)");
  int fused_mode = !want_single ? FUSED_PLACED : jt.probe_mode == 2 ? FUSED_LOOKBACK : FUSED_UNORDERED;
  // Few output rows expected — a FilterExec fused below the probe side, or a build side that covers little of its key range
  // (the hit rate of foreign keys drawn from that range): counts + hit words, then k_join_emit_listed.  The counts pass
  // knows the answer before the second kernel is chosen, so a wrong guess costs the counts pass, not the result; the
  // output is in probe order, which every probe_mode accepts.
  bool listed = fused_ok && fused_mode != FUSED_LOOKBACK && (kind == KIND_RANK || kind == KIND_ARRAY) &&
                (row_mask != nullptr || lazy != nullptr || (double)jt.build.nrows < 0.15 * (double)jt.am_size);
  // what the grouped lookup found decides the placement: every probe row matched — row i of the output is probe row i, no counts
  // pass; else the cursor when nobody observes the order, else counts + placed
  const bool returned_all_hit = returned && rp.hits == np && !row_mask && join_type != DFGPU_JOIN_RIGHT_ANTI;
  if (returned && !returned_all_hit && fused_mode != FUSED_LOOKBACK) fused_mode = want_single ? FUSED_UNORDERED : FUSED_PLACED;
  if (returned_all_hit) fused_mode = FUSED_PLACED;
  if (listed) fused_mode = FUSED_PLACED;
  // The unordered flavour claims every 2048-row tile's output range with one returning atomicAdd on ONE cursor, and a contended
  // agent-scope atomic retires one claim per ~12 ns (scripts/microbench/tile_atomics.hip): 3.5 ms for an SF100 probe's 293 K
  // tiles.  A tile of narrow rows streams in less than that — a key-only join moves 16 B per probe row: 5 ns per tile — so the
  // cursor, not HBM, would set the pace (3.64 ms); counts + placed have no cursor (0.92 + 1.62 ms) and give probe order besides.
  // Probe keys in NO order against a direct-address table beyond the caches: every lookup is a line of its own, and counts +
  // placed would make every lookup twice (150 M shuffled keys x 600 M random probes, key-only: 11.2 + 11.9 ms).  Two answers:
  //  * the probe reads nothing but its key and nobody observes its order (SELECT l.k ... JOIN, the reference's hj.rs shapes;
  //    semi / anti joins on the key alone): ONE stable radix pass groups the probe keys by the top bits of their range
  //    (sort.hip's pass, 64 groups: a group's slice of the table is ~1/64 of it and sits in L2), then the ordinary probe runs
  //    over the grouped keys — lookups hit lines their neighbours fetched;
  //  * otherwise the single-pass probe: one lookup per row.
  if (grouped_keys_only) {
    const Column& kc = probe.cols[(size_t)pk[0]];
    BufPtr keys = kc.data;
    if (kc.data_offset != 0) {  // a slice of a larger buffer: the pass wants its own
      keys = make_buf((size_t)np * 8);
      DFGPU_HIP(hipMemcpyAsync(keys->ptr, kc.ptr(), (size_t)np * 8, hipMemcpyDeviceToDevice, r.stream));
    }
    int64_t n_grouped = np;
    if (jt.kind == KIND_RANK && join_type != DFGPU_JOIN_RIGHT_ANTI) {
      // round 4: grouped by their position in the table's key range (grouped.hip: one pass, 2^9 groups or so); keys outside the
      // range — they match nothing, and an Inner / RightSemi probe emits nothing for them — drop out here
      const KeyCol key{kc.ptr(), nullptr, kc.field.type, 8};
      const int nbits = gp_bits_for(jt, 0);
      GroupedRows gr = group_rows_by_key(key, np, gp_spec_for(jt, nbits), nbits, nullptr, true, false, {}, {}, "join_probe_group_keys");
      keys = gr.keys;
      n_grouped = gr.rows;
    } else {
    int range_bits = 0;
    while (range_bits < 64 && (jt.am_size >> range_bits)) range_bits++;
    // groups = keys sharing bits [range_bits - 6, range_bits) of their value: at most two stretches of the key range each
    radix_group_keys(keys, np, std::max(0, range_bits - 6), 6);
    }
    Table grouped;
    grouped.nrows = n_grouped;
    grouped.device = probe.device;
    Column gk = kc;
    gk.data = keys;
    gk.data_offset = 0;
    gk.length = n_grouped;
    gk.stats.reset();
    grouped.cols.push_back(std::move(gk));
    in_grouped_probe = true;
    Table res;
    try {
      res = join_probe(jt, grouped, {0}, join_type, bout_in, {0}, nullptr, nullptr);
    } catch (...) {
      in_grouped_probe = false;
      throw;
    }
    in_grouped_probe = false;
    std::lock_guard<std::mutex> lk(jt.mu);
    jt.info.probe_rows -= n_grouped;  // counted by the inner call as well
    return res;
  }
  // (the probe over keys grouped above: its lookups hit L2 but are still one line request per row — one pass, not two)
  if (fused_ok && fused_mode == FUSED_UNORDERED && np > (1 << 22) && !unclustered && !in_grouped_probe && !returned) {
    int64_t row_bytes = key_bytes / std::max<int64_t>(np, 1) + out_row_bytes;
    for (int c : pout) {
      bool is_key = false;
      for (int k : pk) is_key |= k == c;
      if (!is_key) row_bytes += type_width(probe.cols[c].field.type);
    }
    // tile bytes / (6000 B per ns) below one claim's 12 ns: < 35 B per probe row.  (At two claims' worth the counts pass cost more
    // than the cursor did: a 40-byte semi-join probe of 72 M rows went from 0.35 to 0.8 ms.)
    if (row_bytes * (FUSED_W * BLOCK) < 12 * 6000) fused_mode = FUSED_PLACED;
  }
  // A probe-side row mask is applied in place by the at-most-one-match probes (fused, or lookup -> scan ->
  // materialise); the general pairs path needs the caller to filter first.
  const bool one_match_path = np > 0 && (probe_side_only || fast_inner);
  const bool use_fused_now = use_fused;
  // the counts pass of the listed flavour evaluates the predicate itself; everything else reads a mask
  const bool pred_in_counts = lazy != nullptr && use_fused && listed && fused_mode == FUSED_PLACED && !returned && (kind == KIND_RANK || kind == KIND_ARRAY);
  if (!pred_in_counts) need_mask();
  if (row_mask || pred_in_counts) {
    DFGPU_CHECK(mask_consumed != nullptr, "probe row mask without a fallback");
    *mask_consumed = use_fused || one_match_path;
    if (!*mask_consumed) return Table{};
    ctx.row_mask = row_mask;
  }
  // the interleaved rank table: every flavour but the listed one (membership from the bitmap in the counts pass, ranks of the few
  // listed rows from bitmap + prefix) looks every row's rank up through it
  const bool listed_flavour = use_fused_now && listed && fused_mode == FUSED_PLACED && kind == KIND_RANK;
  if (!listed_flavour) need_tab();
  if (use_fused_now && fused_mode != FUSED_PLACED) {
    // single-pass flavours: output columns are allocated for the upper bound (np rows) because the row count is only
    // known when the kernel ends; HBM is sized for that (288 GB)
    size_t free_b = 0, total_b = 0;
    DFGPU_HIP(hipMemGetInfo(&free_b, &total_b));
    DFGPU_CHECK(np * out_row_bytes <= (int64_t)free_b + r.cached, "single-pass probe: the np-row upper bound of the output does not fit in HBM");
  }

  if (use_fused_now) {
    const int64_t tile_words = (int64_t)FUSED_W * (BLOCK / WAVE);
    const int64_t n_tiles = (n_words + tile_words - 1) / tile_words;
    const int invert = join_type == DFGPU_JOIN_RIGHT_ANTI;
    ctx.row_mask = row_mask;
    BufPtr state = fused_mode == FUSED_LOOKBACK ? make_zero_buf((size_t)n_tiles * 8) : nullptr;
    BufPtr ctl = make_zero_buf(sizeof(FusedCtl));
    BufPtr out_words;
    int64_t n_alloc = np;
    // ---- speculation for the probe-order output: a foreign-key probe (every row finds its key) needs no counts pass — row i of
    // the output IS probe row i.  64 K sampled rows that all hit make it worth trying; the kernel verifies every tile and the
    // host starts over with the counts when one disagrees (the reference's order, exec.rs:3349, at the unordered flavour's cost)
    static thread_local bool no_speculation = false;
    bool speculate = false;
    if (returned_all_hit && !no_speculation) {
      speculate = true;   // (not a guess here: the grouped lookup counted its hits; the kernel's per-tile check stays as the safety net)
    } else if (fused_mode == FUSED_PLACED && !listed && !row_mask && !invert && !no_speculation && np >= (1 << 22) &&
               (kind == KIND_RANK || kind == KIND_ARRAY || kind == KIND_FLAT || kind == KIND_FLAT16) &&
        true) {
      constexpr int S = 1024;  // sampled words
      BufPtr miss = make_zero_buf(4);
      with_kind_and_key(kind, ctx.pkeys.c[0].type, [&](auto kd, auto kt) {
        k_sample_all_hit<decltype(kd)::value, decltype(kt)::value><<<S * WAVE / BLOCK, BLOCK, 0, r.stream>>>(ctx, np, std::max<int64_t>(1, n_words / S), S, miss->as<int>());
      });
      DFGPU_HIP(hipGetLastError());
      int m = 0;
      d2h(&m, miss->ptr, 4);
      speculate = m == 0;
    }
    if (fused_mode == FUSED_PLACED && !speculate) {
      // pass 1: output rows per tile (reads the probe keys only), then the tiles' exclusive prefix
      BufPtr counts = make_buf((size_t)n_tiles * 4);
      state = make_buf((size_t)(n_tiles + 1) * 8);
      if (listed) out_words = make_buf((size_t)n_words * 8);
      {
        ProfileScope ps("join_probe_tile_counts", key_bytes + (pred_in_counts ? np * lazy->pred_bytes_per_row : 0));
        with_kind_and_key(kind, ctx.pkeys.c[0].type, [&](auto kd, auto kt) {
          constexpr int K = decltype(kd)::value, T = decltype(kt)::value;
          if constexpr (kind_is_direct<K>()) {
            const bool nt = option_on("join.counts_nt", true);
            if (pred_in_counts) {
              if (nt) k_join_tile_counts<K, T, FUSED_W, true, true><<<(unsigned)n_tiles, BLOCK, 0, r.stream>>>(ctx, np, invert, nullptr, counts->as<uint32_t>(), out_words->as<uint64_t>(), lazy->pred);
              else k_join_tile_counts<K, T, FUSED_W, true><<<(unsigned)n_tiles, BLOCK, 0, r.stream>>>(ctx, np, invert, nullptr, counts->as<uint32_t>(), out_words->as<uint64_t>(), lazy->pred);
              return;
            }
            if (nt) {
              k_join_tile_counts<K, T, FUSED_W, false, true><<<(unsigned)n_tiles, BLOCK, 0, r.stream>>>(ctx, np, invert, row_mask, counts->as<uint32_t>(), out_words ? out_words->as<uint64_t>() : nullptr);
              return;
            }
          }
          k_join_tile_counts<K, T, FUSED_W><<<(unsigned)n_tiles, BLOCK, 0, r.stream>>>(
              ctx, np, invert, row_mask, counts->as<uint32_t>(), out_words ? out_words->as<uint64_t>() : nullptr);
        });
        DFGPU_HIP(hipGetLastError());
      }
      scan_u32(counts->as<uint32_t>(), n_tiles, state->as<uint64_t>());
      n_alloc = (int64_t)read_u64(state->as<uint64_t>() + n_tiles);
      listed = listed && n_alloc * 4 <= np;  // denser than guessed: the placed kernel streams better than it lists
      if (!listed) need_tab();
      if (pred_in_counts && !listed) {
        // the placed kernel wants a row mask: the counts pass's output words ARE one — (hit & predicate) read as "rows that exist"
        // selects exactly the rows that come out, for the inverted (anti) probe as well
        row_mask = out_words->as<uint64_t>();
        ctx.row_mask = row_mask;
      }
    }
    JoinCopyCols jc{};
    jc.key_col = -1;
    int64_t bytes_in = key_bytes, bytes_per_out = 0, bytes_build_once = 0;
    for (int c : bout) {
      const Column& sc = jt.build.cols[c];
      out.cols.push_back(alloc_like(sc, n_alloc));
      jc.src[jc.n] = sc.ptr();
      jc.dst[jc.n] = out.cols.back().data->ptr;
      jc.width[jc.n] = type_width(sc.field.type);
      if (returned) {  // the column's values came back inside the records
        jc.src[jc.n] = rp.rec->ptr;
        jc.ret_mul[jc.n] = rp.mul[(size_t)jc.n];
        jc.ret_add[jc.n] = rp.add[(size_t)jc.n];
      }
      bytes_per_out += jc.width[jc.n];
      bytes_build_once += jt.build.nrows * jc.width[jc.n];
      jc.n++;
    }
    jc.n_build = jc.n;
    for (int c : pout) {
      const Column& sc = probe.cols[c];
      out.cols.push_back(alloc_like(sc, n_alloc));
      jc.src[jc.n] = sc.ptr();
      jc.dst[jc.n] = out.cols.back().data->ptr;
      jc.width[jc.n] = type_width(sc.field.type);
      bool is_key = false;
      for (int k : pk) is_key |= k == c;
      if (!is_key) bytes_in += np * jc.width[jc.n];  // a key column that is also payload is read once
      if (is_key && (kind == KIND_ARRAY || kind == KIND_RANK) && pk.size() == 1 && jc.key_col < 0 && !sc.validity) jc.key_col = jc.n;
      bytes_per_out += jc.width[jc.n];
      jc.n++;
    }
    hipEvent_t ea = nullptr, eb = nullptr;
    if (r.profiling) {
      DFGPU_HIP(hipEventCreate(&ea));
      DFGPU_HIP(hipEventCreate(&eb));
      DFGPU_HIP(hipEventRecord(ea, r.stream));
    }
    int64_t n_out = n_alloc;
    if (fused_mode != FUSED_PLACED || n_alloc > 0) {
      // look-back: persistent workgroups pulling tickets; unordered / placed: one workgroup per tile
      const unsigned g = fused_mode == FUSED_LOOKBACK ? (unsigned)std::min<int64_t>(n_tiles, (int64_t)r.num_cus * 8) : (unsigned)n_tiles;
      uint64_t* st = state ? state->as<uint64_t>() : nullptr;
      auto launch = [&](auto kern) { kern<<<g, BLOCK, 0, r.stream>>>(ctx, jc, np, invert, st, ctl->as<FusedCtl>(), row_mask); };
      with_kind_and_key(kind, ctx.pkeys.c[0].type, [&](auto kd, auto kt) {
        constexpr int K = decltype(kd)::value, T = decltype(kt)::value;
        constexpr bool KR = kind_is_direct<K>();  // the direct-address kinds hold the one integer key in registers
        if constexpr (KR) {
          if (listed) {
            // (sparse: fewer than one row in `join.listed_sparse_den` comes out)
            const int64_t den = option_int("join.listed_sparse_den", 32);   // (0: never)
            if (den > 0 && n_alloc * den <= np) {
              const int64_t groups = (n_words + EL_WORDS_SPARSE - 1) / EL_WORDS_SPARSE;
              k_join_emit_listed<K, T, EL_WORDS_SPARSE><<<(unsigned)groups, BLOCK, 0, r.stream>>>(ctx, jc, np, out_words->as<uint64_t>(), st, (int)(EL_WORDS_SPARSE / tile_words));
            } else {
              const int64_t groups = (n_words + EL_WORDS - 1) / EL_WORDS;
              k_join_emit_listed<K, T, EL_WORDS><<<(unsigned)groups, BLOCK, 0, r.stream>>>(ctx, jc, np, out_words->as<uint64_t>(), st, (int)(EL_WORDS / tile_words));
            }
            return;
          }
        }
        if (fused_mode == FUSED_LOOKBACK) launch(k_join_probe_fused<K, T, FUSED_W, FUSED_LOOKBACK, false>);
        else if (fused_mode == FUSED_PLACED) {
          if (KR && jc.key_col >= 0) launch(k_join_probe_fused<K, T, FUSED_W, FUSED_PLACED, KR>);
          else launch(k_join_probe_fused<K, T, FUSED_W, FUSED_PLACED, false>);
        } else {
          if (KR && jc.key_col >= 0) launch(k_join_probe_fused<K, T, FUSED_W, FUSED_UNORDERED, KR>);
          else launch(k_join_probe_fused<K, T, FUSED_W, FUSED_UNORDERED, false>);
        }
      });
      DFGPU_HIP(hipGetLastError());
      if (r.profiling) DFGPU_HIP(hipEventRecord(eb, r.stream));
      if (fused_mode != FUSED_PLACED) n_out = (int64_t)read_u64(reinterpret_cast<const uint64_t*>(&ctl->as<FusedCtl>()->total));
      if (speculate) {
        unsigned missed = 0;
        d2h(&missed, &ctl->as<FusedCtl>()->ticket, 4);
        if (missed) {  // some probe row has no partner after all: the same probe again, counted
          if (r.profiling) {
            std::lock_guard<std::mutex> lk(r.mu);
            r.recs.push_back(Runtime::Rec{"join_probe_speculation_missed", ea, eb, 0, std::this_thread::get_id()});
          }
          {
            std::lock_guard<std::mutex> lk(jt.mu);
            jt.info.probe_rows -= np;
          }
          out = Table{};
          no_speculation = true;
          Table again;
          try {
            again = join_probe(jt, probe, pk, join_type, bout_in, pout, row_mask, mask_consumed);
          } catch (...) {
            no_speculation = false;
            throw;
          }
          no_speculation = false;
          return again;
        }
      }
    } else if (r.profiling) {
      DFGPU_HIP(hipEventRecord(eb, r.stream));
    }
    // algorithmic bytes of this launch (SURVEY 8d config 3 ii): every referenced probe column once, the build payload
    // columns once, the output written once (re-reads of build rows matched by several probe rows, table words, tile
    // state are overhead); known only now that n_out is
    if (r.profiling) {
      std::lock_guard<std::mutex> lk(r.mu);
      r.recs.push_back(Runtime::Rec{listed ? "join_probe_listed" : fused_mode == FUSED_PLACED ? "join_probe_placed" : "join_probe_fused", ea, eb,
                                    bytes_in + (n_out > 0 ? bytes_build_once : 0) + n_out * bytes_per_out, std::this_thread::get_id()});
    }
    out.nrows = n_out;
    for (Column& c : out.cols) c.length = n_out;
  } else if (np > 0 && (probe_side_only || (build_side_only && jt.keys_unique) || fast_inner)) {
  // LeftSemi/LeftAnti/LeftMark must mark EVERY matching build row: first-match suffices only for unique keys
    // ---- at most one match per probe row
    BufPtr mask = make_buf(bitmap_bytes(np));
    BufPtr first = fast_inner ? make_buf((size_t)np * 4) : nullptr;
    {
      ProfileScope ps("join_probe_lookup", key_bytes);  // algorithmic: the probe keys (table reads / match ids are overhead)
      int g = grid_for(n_words, (BLOCK / WAVE) * PROBE_UNROLL);
      int invert = join_type == DFGPU_JOIN_RIGHT_ANTI;
      with_kind_and_key(jt.kind, ctx.pkeys.c[0].type, [&](auto kd, auto kt) {
        k_probe_first<decltype(kd)::value, decltype(kt)::value><<<g, BLOCK, 0, r.stream>>>(ctx, np, invert, first ? first->as<uint32_t>() : nullptr,
                                                                                           mask->as<uint64_t>(), visited);
      });
      DFGPU_HIP(hipGetLastError());
    }
    if (probe_side_only) {
      out = compact_table(probe, pout, mask->as<uint64_t>(), nullptr);
    } else if (build_side_only) {
      out.nrows = 0;  // emitted by dfgpu_join_emit_unmatched
      for (int c : bout) out.cols.push_back(alloc_like(jt.build.cols[c], 0));
    } else {
      BufPtr prefix = make_buf((size_t)(n_words + 1) * 8);
      scan_mask_popcounts(mask->as<uint64_t>(), nullptr, np, prefix->as<uint64_t>());
      const int64_t n_out = (int64_t)read_u64(prefix->as<uint64_t>() + n_words);
      out.nrows = n_out;
      JoinCopyCols jc{};
      // algorithmic bytes: probe payload read once, build payload read per output row, output
      // written once (mask / match-id traffic is overhead and not counted)
      int64_t bytes = 0;
      for (int c : bout) {
        const Column& sc = jt.build.cols[c];
        out.cols.push_back(alloc_like(sc, n_out));
        jc.src[jc.n] = sc.ptr();
        jc.dst[jc.n] = out.cols.back().data->ptr;
        jc.width[jc.n] = type_width(sc.field.type);
        bytes += 2 * n_out * jc.width[jc.n];
        jc.n++;
      }
      jc.n_build = jc.n;
      for (int c : pout) {
        const Column& sc = probe.cols[c];
        out.cols.push_back(alloc_like(sc, n_out));
        jc.src[jc.n] = sc.ptr();
        jc.dst[jc.n] = out.cols.back().data->ptr;
        jc.width[jc.n] = type_width(sc.field.type);
        bytes += (np + n_out) * jc.width[jc.n];
        jc.n++;
      }
      if (n_out > 0 && jc.n > 0) {
        ProfileScope ps("join_probe_materialize", bytes);
        k_join_materialize<<<grid_for(n_words, (BLOCK / WAVE) * PROBE_UNROLL), BLOCK, 0, r.stream>>>(jc, mask->as<uint64_t>(), prefix->as<uint64_t>(),
                                                                                                      first->as<uint32_t>(), np);
        DFGPU_HIP(hipGetLastError());
      }
    }
  } else if (single_pass_pairs) {
    // ---- general M:N path in one pass (flat tables, INNER, order unobserved): sample -> pairs at a cursor -> gathers
    constexpr int64_t SAMPLE = 1 << 16;
    const int64_t every = std::max<int64_t>(1, np / SAMPLE), n_sample = (np + every - 1) / every;
    BufPtr ctl = make_zero_buf(16);
    with_kind(jt.kind, [&](auto kt) {
      constexpr int K = decltype(kt)::value;
      if constexpr (kind_is_flat<K>()) k_probe_pairs_sample<K><<<grid_for(n_sample, BLOCK), BLOCK, 0, r.stream>>>(ctx, np, every, n_sample, ctl->as<unsigned long long>());
    });
    unsigned long long seen = 0;
    d2h(&seen, ctl->ptr, 8);
    // (the sample's pairs scaled to the probe side, a quarter more and a constant: an underestimate costs one more pass at the exact size)
    unsigned long long capacity = (unsigned long long)((double)seen * (double)np / (double)n_sample * 1.25) + (1ull << 16);
    BufPtr ob, op;
    unsigned long long total = 0;
    for (int attempt = 0; attempt < 2; attempt++) {
      ob = make_buf((size_t)capacity * 8);
      op = make_buf((size_t)capacity * 8);
      DFGPU_HIP(hipMemsetAsync(ctl->ptr, 0, 16, r.stream));
      {
        ProfileScope ps("join_probe_pairs_single", key_bytes + (int64_t)capacity * 16);
        const int g = (int)std::min<int64_t>((n_words + PROBE_UNROLL * (BLOCK / WAVE) - 1) / (PROBE_UNROLL * (BLOCK / WAVE)), (int64_t)r.num_cus * 16);
        with_kind(jt.kind, [&](auto kt) {
          constexpr int K = decltype(kt)::value;
          if constexpr (kind_is_flat<K>()) k_probe_pairs_single<K><<<g, BLOCK, 0, r.stream>>>(ctx, np, ctl->as<unsigned long long>(), capacity, ob->as<int64_t>(), op->as<int64_t>());
        });
        DFGPU_HIP(hipGetLastError());
      }
      d2h(&total, ctl->ptr, 8);
      if (total <= capacity) break;
      DFGPU_CHECK(attempt == 0, "join: the single-pass probe overflowed a buffer of its exact size");
      capacity = total;   // (the cursor counted every pair: the second try fits)
    }
    const int64_t n_out = (int64_t)total;
    out.nrows = n_out;
    for (Column& c : gather_columns(jt.build, bout, ob->as<int64_t>(), n_out, false)) out.cols.push_back(std::move(c));
    for (Column& c : gather_columns(probe, pout, op->as<int64_t>(), n_out, false)) out.cols.push_back(std::move(c));
  } else {
    // ---- general M:N path: counts -> scan -> pairs -> gathers
    BufPtr row_counts = make_buf((size_t)(np ? np : 1) * 4);
    BufPtr row_first = make_buf((size_t)(np ? np : 1) * 4);
    BufPtr word_counts = make_buf((size_t)(n_words ? n_words : 1) * 4);
    BufPtr prefix = make_buf((size_t)(n_words + 1) * 8);
    int g = grid_for(n_words, BLOCK / WAVE);
    if (np) {
      ProfileScope ps("join_probe_count", key_bytes + np * 4);
      with_kind(jt.kind, [&](auto kt) {
        k_probe_count<decltype(kt)::value><<<g, BLOCK, 0, r.stream>>>(ctx, np, join_type, row_counts->as<uint32_t>(), word_counts->as<uint32_t>(), visited, row_first->as<uint32_t>());
      });
      DFGPU_HIP(hipGetLastError());
    }
    scan_u32(word_counts->as<uint32_t>(), n_words, prefix->as<uint64_t>());
    const int64_t n_out = (int64_t)read_u64(prefix->as<uint64_t>() + n_words);
    BufPtr ob = make_buf((size_t)(n_out ? n_out : 1) * 8), op = make_buf((size_t)(n_out ? n_out : 1) * 8);
    BufPtr om = join_type == DFGPU_JOIN_RIGHT_MARK ? make_buf((size_t)n_out + 64) : nullptr;
    if (n_out) {
      ProfileScope ps("join_probe_emit", key_bytes + np * 4 + n_out * 16);
      with_kind(jt.kind, [&](auto kt) {
        k_probe_emit<decltype(kt)::value><<<g, BLOCK, 0, r.stream>>>(ctx, np, join_type, row_counts->as<uint32_t>(), prefix->as<uint64_t>(), ob->as<int64_t>(), op->as<int64_t>(), om ? om->as<uint8_t>() : nullptr, row_first->as<uint32_t>());
      });
      DFGPU_HIP(hipGetLastError());
    }
    out.nrows = n_out;
    const bool build_null_possible = join_type == DFGPU_JOIN_RIGHT || join_type == DFGPU_JOIN_FULL;
    if (join_type != DFGPU_JOIN_RIGHT_MARK)
      for (Column& c : gather_columns(jt.build, bout, ob->as<int64_t>(), n_out, build_null_possible)) out.cols.push_back(std::move(c));
    for (Column& c : gather_columns(probe, pout, op->as<int64_t>(), n_out, false)) out.cols.push_back(std::move(c));
    if (join_type == DFGPU_JOIN_RIGHT_MARK) out.cols.push_back(mark_column(om->as<uint8_t>(), n_out));
  }
  jt.info.output_rows += out.nrows;
  return out;
}

// HashJoinExec with a JoinFilter: the general (pairs) path with the filter between pair generation and the
// per-JoinType adjustment.  Not a tuned path: pairs and the intermediate batch are materialised.
// null_aware anti joins (stream.rs:755-808): bookkeeping before the probe.  Returns what the probe has to do.
enum NullAwareAction { NA_PROCEED, NA_EMPTY, NA_MASK_NULL_PROBE_KEYS };
static NullAwareAction null_aware_before_probe(JoinTable& jt, const Table& probe, const std::vector<int>& pk, int join_type, bool has_filter) {
  if (!jt.null_aware) return NA_PROCEED;
  DFGPU_CHECK(join_type == DFGPU_JOIN_LEFT_ANTI || join_type == DFGPU_JOIN_RIGHT_ANTI,
              "null_aware can only be true for LeftAnti joins and RightAnti joins with `CollectLeft` `PartitionMode`");
  DFGPU_CHECK(!(join_type == DFGPU_JOIN_RIGHT_ANTI && has_filter), "null_aware RightAnti join does not support a join filter");
  DFGPU_CHECK(pk.size() == 1 && pk[0] >= 0 && pk[0] < (int)probe.cols.size(), "probe key column out of range");
  const Column& key = probe.cols[pk[0]];
  if (join_type == DFGPU_JOIN_RIGHT_ANTI) {
    if (jt.build_side_has_null) return NA_EMPTY;
    if (jt.build.nrows == 0 || probe.nrows == 0) return NA_PROCEED;  // NOT IN (empty set) is TRUE for every probe row, NULL keys included
    return key.has_nulls() ? NA_MASK_NULL_PROBE_KEYS : NA_PROCEED;
  }
  if (probe.nrows > 0) jt.probe_side_non_empty = true;
  if (null_count_of(key) > 0) jt.probe_side_has_null = true;
  return NA_PROCEED;  // LeftAnti emits nothing per probe table; the flags act in dfgpu_join_emit_unmatched
}

static Table empty_selection(const Table& t, const std::vector<int>& cols) {
  BufPtr zero = make_zero_buf(bitmap_bytes(t.nrows));
  return compact_table(t, cols, zero->as<uint64_t>(), nullptr);
}

static Table join_probe_null_aware(JoinTable& jt, const Table& probe, const std::vector<int>& pk, int join_type, const std::vector<int>& bout,
                                   const std::vector<int>& pout) {
  switch (null_aware_before_probe(jt, probe, pk, join_type, false)) {
    case NA_EMPTY:
      for (int c : pout) DFGPU_CHECK(c >= 0 && c < (int)probe.cols.size(), "probe output column out of range");
      {
        std::lock_guard<std::mutex> lk(jt.mu);
        jt.info.probe_rows += probe.nrows;
      }
      return empty_selection(probe, pout);
    case NA_MASK_NULL_PROBE_KEYS: {
      bool consumed = false;
      Table res = join_probe(jt, probe, pk, join_type, bout, pout, probe.cols[pk[0]].valid_words(), &consumed);
      if (!consumed) {  // table kinds without an in-place row mask (LDS radix partitions): drop the NULL-key rows first
        std::vector<int> all(probe.cols.size());
        for (size_t i = 0; i < all.size(); i++) all[i] = (int)i;
        const Table filtered = compact_table(probe, all, probe.cols[pk[0]].valid_words(), nullptr);
        res = join_probe(jt, filtered, pk, join_type, bout, pout);
      }
      return res;
    }
    default:
      return join_probe(jt, probe, pk, join_type, bout, pout);
  }
}

// key-equal (build row, probe row) pairs of one probe table = an Inner join on the keys; nothing is marked visited yet.
// Chained / direct-address tables: counts -> scan -> pairs in probe order; radix table: LDS join, partition order.
struct Pairs {
  BufPtr ob, op;
  int64_t m = 0;
};
static Pairs key_equal_pairs(JoinTable& jt, const Table& probe, const std::vector<int>& pk) {
  Runtime& r = rt();
  Pairs P;
  if (jt.kind == KIND_RADIX) {
    radix_join_pairs(*jt.radix, jt.build, jt.key_cols, probe, pk, jt.null_equality == DFGPU_NULL_EQUALS_NULL, jt.force_collisions, P.ob, P.op, P.m);
    return P;
  }
  const int64_t np = probe.nrows;
  const int64_t n_words = (np + 63) / 64;
  ProbeCtx ctx = make_ctx(jt, probe, pk);
  BufPtr row_counts = make_buf((size_t)(np ? np : 1) * 4), word_counts = make_buf((size_t)(n_words ? n_words : 1) * 4);
  BufPtr row_first = make_buf((size_t)(np ? np : 1) * 4);
  BufPtr prefix = make_buf((size_t)(n_words + 1) * 8);
  const int g = grid_for(n_words, BLOCK / WAVE);
  if (np) with_kind(jt.kind, [&](auto kt) {
    k_probe_count<decltype(kt)::value><<<g, BLOCK, 0, r.stream>>>(ctx, np, DFGPU_JOIN_INNER, row_counts->as<uint32_t>(), word_counts->as<uint32_t>(), nullptr, row_first->as<uint32_t>());
  });
  scan_u32(word_counts->as<uint32_t>(), n_words, prefix->as<uint64_t>());
  P.m = (int64_t)read_u64(prefix->as<uint64_t>() + n_words);
  P.ob = make_buf((size_t)(P.m ? P.m : 1) * 8);
  P.op = make_buf((size_t)(P.m ? P.m : 1) * 8);
  if (P.m) with_kind(jt.kind, [&](auto kt) {
    k_probe_emit<decltype(kt)::value><<<g, BLOCK, 0, r.stream>>>(ctx, np, DFGPU_JOIN_INNER, row_counts->as<uint32_t>(), prefix->as<uint64_t>(), P.ob->as<int64_t>(), P.op->as<int64_t>(), nullptr, row_first->as<uint32_t>());
  });
  DFGPU_HIP(hipGetLastError());
  return P;
}

// pairs -> [JoinFilter] -> the join type's output (apply_join_filter_to_indices + adjust_indices_by_join_type, joins/utils.rs:1248-1318,1432-1488)
static Table join_probe_with_filter(JoinTable& jt, const Table& probe, const std::vector<int>& pk, int join_type, const std::vector<int>& bout,
                                    const std::vector<int>& pout, const dfgpu_join_filter* jfp) {
  Runtime& r = rt();
  DFGPU_CHECK(pk.size() == jt.key_cols.size(), "probe key count differs from build key count");
  DFGPU_CHECK(probe.device == jt.build.device, "the probe table lives on another device than the join table (move it with dfgpu_table_copy_to_device)");
  for (int c : bout) DFGPU_CHECK(c >= 0 && c < (int)jt.build.cols.size(), "build output column out of range");
  for (int c : pout) DFGPU_CHECK(c >= 0 && c < (int)probe.cols.size(), "probe output column out of range");
  const int64_t np = probe.nrows;
  const int64_t n_words = (np + 63) / 64;
  const int g = grid_for(n_words, BLOCK / WAVE);
  {
    std::lock_guard<std::mutex> lk(jt.mu);
    jt.info.probe_rows += np;
  }
  Pairs P = key_equal_pairs(jt, probe, pk);
  const int64_t m = P.m;
  BufPtr ob = P.ob, op = P.op;
  if (!jfp && join_type == DFGPU_JOIN_INNER) {
    // Inner join without a residual filter: the pairs ARE the output rows (no visited bytes, no per-probe-row hit counts)
    Table out;
    out.nrows = m;
    for (Column& c : gather_columns(jt.build, bout, ob->as<int64_t>(), m, false)) out.cols.push_back(std::move(c));
    for (Column& c : gather_columns(probe, pout, op->as<int64_t>(), m, false)) out.cols.push_back(std::move(c));
    DFGPU_HIP(hipStreamSynchronize(r.stream));
    jt.info.output_rows += out.nrows;
    return out;
  }
  // ---- intermediate batch + filter expression -> pass mask over the pairs (no filter: every pair passes)
  const int64_t m_words = (m + 63) / 64;
  BufPtr pass = make_buf((size_t)(m_words ? m_words : 1) * 8);
  DFGPU_HIP(hipMemsetAsync(pass->ptr, jfp ? 0 : 0xFF, (size_t)(m_words ? m_words : 1) * 8, r.stream));
  if (m && jfp) {
    const dfgpu_join_filter& jf = *jfp;
    Table inter;
    inter.nrows = m;
    for (int i = 0; i < jf.n_columns; i++) {
      const bool left = jf.column_side[i] == 0;
      const Table& side = left ? jt.build : probe;
      DFGPU_CHECK(jf.column_index[i] >= 0 && jf.column_index[i] < (int)side.cols.size(), "join filter column index out of range");
      inter.cols.push_back(gather_column(side.cols[jf.column_index[i]], (left ? ob : op)->as<int64_t>(), m, false));
    }
    Datum d = evaluate(jf.expression, inter);
    DFGPU_CHECK(d.col.field.type == DFGPU_BOOL, "join filter expression must be Boolean");
    Column mc = datum_to_column(d, m, "");
    if (mc.validity) and_bitmaps(mc.data->as<uint64_t>(), mc.valid_words(), m_words, pass->as<uint64_t>());
    else pass = mc.data;
  }
  // ---- matches = passing pairs
  BufPtr hits = make_zero_buf((size_t)(np ? np : 1) * 4);
  uint8_t* visited = nullptr;
  if (needs_visited(join_type)) {
    {
      std::lock_guard<std::mutex> lk(jt.mu);
      if (!jt.visited) {
        jt.visited = make_zero_buf((size_t)jt.build.nrows + 64);
        DFGPU_HIP(hipStreamSynchronize(rt().stream));  // zeroed before any other thread's stream marks rows in it
      }
    }
    visited = jt.visited->as<uint8_t>();
  }
  if (m) k_pair_tally<<<grid_for(m, BLOCK), BLOCK, 0, r.stream>>>(ob->as<int64_t>(), op->as<int64_t>(), pass->as<uint64_t>(), m, hits->as<uint32_t>(), visited);
  Table out;
  const bool pairs_out = join_type == DFGPU_JOIN_INNER || join_type == DFGPU_JOIN_LEFT || join_type == DFGPU_JOIN_RIGHT || join_type == DFGPU_JOIN_FULL;
  if (pairs_out) {
    BufPtr pprefix = make_buf((size_t)(m_words + 1) * 8);
    scan_mask_popcounts(pass->as<uint64_t>(), nullptr, m, pprefix->as<uint64_t>());
    const int64_t n_pass = m ? (int64_t)read_u64(pprefix->as<uint64_t>() + m_words) : 0;
    int64_t n_un = 0;
    BufPtr umask, uprefix;
    const bool probe_outer = join_type == DFGPU_JOIN_RIGHT || join_type == DFGPU_JOIN_FULL;
    if (probe_outer && np) {
      umask = make_buf(bitmap_bytes(np));
      uprefix = make_buf((size_t)(n_words + 1) * 8);
      k_hits_mask<<<g, BLOCK, 0, r.stream>>>(hits->as<uint32_t>(), np, 0, umask->as<uint64_t>(), nullptr);
      scan_mask_popcounts(umask->as<uint64_t>(), nullptr, np, uprefix->as<uint64_t>());
      n_un = (int64_t)read_u64(uprefix->as<uint64_t>() + n_words);
    }
    const int64_t n_out = n_pass + n_un;
    BufPtr ob2 = make_buf((size_t)(n_out ? n_out : 1) * 8), op2 = make_buf((size_t)(n_out ? n_out : 1) * 8);
    if (n_pass) k_pairs_compact<<<grid_for(m_words, BLOCK / WAVE), BLOCK, 0, r.stream>>>(ob->as<int64_t>(), op->as<int64_t>(), pass->as<uint64_t>(), pprefix->as<uint64_t>(), m,
                                                                                        ob2->as<int64_t>(), op2->as<int64_t>());
    if (n_un) k_append_unmatched<<<g, BLOCK, 0, r.stream>>>(umask->as<uint64_t>(), uprefix->as<uint64_t>(), np, n_pass, ob2->as<int64_t>(), op2->as<int64_t>());
    DFGPU_HIP(hipGetLastError());
    out.nrows = n_out;
    for (Column& c : gather_columns(jt.build, bout, ob2->as<int64_t>(), n_out, probe_outer)) out.cols.push_back(std::move(c));
    for (Column& c : gather_columns(probe, pout, op2->as<int64_t>(), n_out, false)) out.cols.push_back(std::move(c));
  } else if (join_type == DFGPU_JOIN_RIGHT_SEMI || join_type == DFGPU_JOIN_RIGHT_ANTI) {
    BufPtr mask = make_zero_buf(bitmap_bytes(np ? np : 1));
    if (np) k_hits_mask<<<g, BLOCK, 0, r.stream>>>(hits->as<uint32_t>(), np, join_type == DFGPU_JOIN_RIGHT_SEMI, mask->as<uint64_t>(), nullptr);
    out = compact_table(probe, pout, mask->as<uint64_t>(), nullptr);
  } else if (join_type == DFGPU_JOIN_RIGHT_MARK) {
    BufPtr mask = make_zero_buf(bitmap_bytes(np ? np : 1));
    BufPtr bytes = make_zero_buf((size_t)np + 64);
    if (np) k_hits_mask<<<g, BLOCK, 0, r.stream>>>(hits->as<uint32_t>(), np, 1, mask->as<uint64_t>(), bytes->as<uint8_t>());
    out.nrows = np;
    for (int c : pout) out.cols.push_back(probe.cols[c]);
    out.cols.push_back(mark_column(bytes->as<uint8_t>(), np));
  } else {
    // LeftSemi / LeftAnti / LeftMark: emitted from the visited bits by dfgpu_join_emit_unmatched
    out.nrows = 0;
    for (int c : bout) out.cols.push_back(alloc_like(jt.build.cols[c], 0));
  }
  DFGPU_HIP(hipStreamSynchronize(r.stream));
  jt.info.output_rows += out.nrows;
  return out;
}

}  // namespace dfgpu

using namespace dfgpu;

extern "C" {

int dfgpu_join_build(dfgpu_table_t build, const int* key_cols, int nkeys, int null_equality, const dfgpu_join_options* opts, dfgpu_join_t* out) {
  return guarded([&] {
    require_init();
    DFGPU_CHECK(nkeys >= 1 && key_cols && out, "join needs at least one key column");
    dfgpu_join_options o{1024, DFGPU_DEFAULT_MIN_KEY_DENSITY, 0, 0, 0, 0};
    if (opts) o = *opts;
    auto jt = join_build(*unwrap(build), std::vector<int>(key_cols, key_cols + nkeys), null_equality, o);
    *out = reinterpret_cast<dfgpu_join_t>(jt.release());
  });
}

// ---- streaming build side: collect_left_input (hash_join/exec.rs:2569-2705) accumulates the build child's batches, charging
// a MemoryReservation per batch (exec.rs:2608 try_grow), then concat_batches + the table build.
struct JoinBuilder {
  std::vector<int> key_cols;
  int null_equality = 0;
  dfgpu_join_options opts{};
  std::vector<std::unique_ptr<Table>> batches;  // shallow copies: the pushed tables' buffers stay alive through them
  int64_t rows = 0, bytes = 0;
  dfgpu_reservation_t reservation = nullptr;
};
static int64_t table_bytes(const Table& t) {
  int64_t b = 0;
  for (const Column& c : t.cols) b += (int64_t)data_bytes(c.field.type, c.length) + (c.validity ? (int64_t)bitmap_bytes(c.length) : 0);
  return b;
}

int dfgpu_join_builder_create(const int* key_cols, int nkeys, int null_equality, const dfgpu_join_options* opts, dfgpu_join_builder_t* out) {
  return guarded([&] {
    require_init();
    DFGPU_CHECK(nkeys >= 1 && key_cols && out, "join needs at least one key column");
    auto b = std::make_unique<JoinBuilder>();
    b->key_cols.assign(key_cols, key_cols + nkeys);
    b->null_equality = null_equality;
    b->opts = dfgpu_join_options{1024, DFGPU_DEFAULT_MIN_KEY_DENSITY, 0, 0, 0, 0};
    if (opts) b->opts = *opts;
    *out = reinterpret_cast<dfgpu_join_builder_t>(b.release());
  });
}
int dfgpu_join_builder_push(dfgpu_join_builder_t h, dfgpu_table_t batch) {
  return guarded([&] {
    require_init();
    DFGPU_CHECK(h != nullptr, "null join builder");
    JoinBuilder* b = reinterpret_cast<JoinBuilder*>(h);
    const Table& t = *unwrap(batch);
    DFGPU_CHECK(b->batches.empty() || t.cols.size() == b->batches[0]->cols.size(), "join builder: batches differ in their column count");
    // the concatenated copy + the join table are still to come: grow the reservation by this batch's share of them
    // (2 x its bytes bounds copy + table for every table kind); refusal = the reference's ResourcesExhausted
    const int64_t tb = table_bytes(t);
    dfgpu_reservation_t more = nullptr;
    if (dfgpu_mem_try_reserve(2 * tb, &more) != 0) throw Error(dfgpu_last_error());
    if (b->reservation) {
      int64_t held = 0;
      (void)dfgpu_mem_reservation_size(b->reservation, &held);
      (void)dfgpu_mem_release(b->reservation);
      (void)dfgpu_mem_release(more);
      b->reservation = nullptr;
      if (dfgpu_mem_try_reserve(held + 2 * tb, &b->reservation) != 0) throw Error(dfgpu_last_error());
    } else {
      b->reservation = more;
    }
    b->batches.push_back(std::make_unique<Table>(t));
    b->rows += t.nrows;
    b->bytes += tb;
  });
}
int dfgpu_join_builder_finish(dfgpu_join_builder_t h, dfgpu_join_t* out) {
  return guarded([&] {
    require_init();
    DFGPU_CHECK(h != nullptr && out != nullptr, "null argument");
    std::unique_ptr<JoinBuilder> b(reinterpret_cast<JoinBuilder*>(h));
    struct Release {
      dfgpu_reservation_t r;
      ~Release() { if (r) (void)dfgpu_mem_release(r); }
    } rel{b->reservation};
    DFGPU_CHECK(!b->batches.empty(), "join builder: no batch was pushed (push an empty batch for an empty build side)");
    std::unique_ptr<Table> whole;
    if (b->batches.size() == 1) {
      whole = std::make_unique<Table>(*b->batches[0]);
    } else {
      std::vector<dfgpu_table_t> hs;
      for (auto& t : b->batches) hs.push_back(wrap(t.get()));
      dfgpu_table_t cat = nullptr;
      if (dfgpu_table_concat(hs.data(), (int)hs.size(), &cat) != 0) throw Error(dfgpu_last_error());  // concat_batches (exec.rs:2705)
      whole.reset(unwrap(cat));
    }
    b->batches.clear();
    auto jt = join_build(*whole, b->key_cols, b->null_equality, b->opts);
    *out = reinterpret_cast<dfgpu_join_t>(jt.release());
  });
}
int dfgpu_join_builder_free(dfgpu_join_builder_t h) {
  return guarded([&] {
    if (!h) return;
    std::unique_ptr<JoinBuilder> b(reinterpret_cast<JoinBuilder*>(h));
    if (b->reservation) (void)dfgpu_mem_release(b->reservation);
  });
}

// what a hash join of these sizes will hold on the device at its peak, for admission control (the optimizer rule declines —
// keeps the CPU operator, which can spill — when dfgpu_mem_try_reserve refuses this much)
int dfgpu_join_estimate_bytes(int64_t build_rows, int64_t build_row_bytes, int64_t probe_rows, int64_t output_rows, int64_t output_row_bytes, int64_t* out) {
  return guarded([&] {
    DFGPU_CHECK(out != nullptr && build_rows >= 0 && probe_rows >= 0, "bad argument");
    // join table: the widest kind (ArrayMap over a sparse range / chained table: ~2 x 4 B x max(rows, range)) is bounded by 16 B per
    // build row for the reference's own density gate; the LDS radix join holds 12 B records of both sides twice (ping-pong) + pairs
    const int64_t table = build_rows * 16;
    const int64_t radix = (build_rows + probe_rows) * 24 + (output_rows < 0 ? probe_rows : output_rows) * 16;
    const int64_t outb = (output_rows < 0 ? probe_rows : output_rows) * output_row_bytes;
    *out = build_rows * build_row_bytes /* concat_batches copy */ + std::max(table, radix) + outb + probe_rows * 5 /* match ids, masks */;
  });
}

int dfgpu_column_minmax(dfgpu_table_t table, int column, int64_t* out_min, int64_t* out_max, int64_t* out_valid, int* out_ascending) {
  return guarded([&] {
    require_init();
    Table& t = *unwrap(table);
    DFGPU_CHECK(column >= 0 && column < (int)t.cols.size(), "column index out of range");
    const ColStats st = column_stats(t.cols[column], t.nrows);
    if (out_min) *out_min = st.min;
    if (out_max) *out_max = st.max;
    if (out_valid) *out_valid = st.valid;
    if (out_ascending) *out_ascending = st.ascending;
  });
}

int dfgpu_column_inlist(dfgpu_table_t th, int column, int64_t max_size, int64_t max_distinct_values, int64_t* out_values, int64_t capacity, int64_t* out_n) {
  return guarded([&] {
    require_init();
    Table& t = *unwrap(th);
    DFGPU_CHECK(out_n && column >= 0 && column < (int)t.cols.size(), "bad argument");
    const Column& c = t.cols[(size_t)column];
    DFGPU_CHECK(is_integer_like(c.field.type) && c.field.type != DFGPU_UINT64, "dfgpu_column_inlist: integer columns only");
    const int w = type_width(c.field.type);
    *out_n = -1;
    if (t.nrows == 0) {
      *out_n = 0;
      return;
    }
    if (max_distinct_values <= 0 || t.nrows * w > max_size) return;  // PushdownStrategy::Map
    std::vector<uint8_t> raw((size_t)t.nrows * w);
    d2h(raw.data(), c.ptr(), raw.size());
    std::vector<uint64_t> valid;
    if (c.validity) {
      valid.resize(bitmap_bytes(t.nrows) / 8);
      d2h(valid.data(), c.validity->ptr, valid.size() * 8);
    }
    std::vector<int64_t> v;
    v.reserve((size_t)t.nrows);
    for (int64_t i = 0; i < t.nrows; i++) {
      if (c.validity && !((valid[(size_t)(i >> 6)] >> (i & 63)) & 1)) continue;
      switch (c.field.type) {
        case DFGPU_INT32: case DFGPU_DATE32: v.push_back(((const int32_t*)raw.data())[i]); break;
        case DFGPU_UINT32: v.push_back(((const uint32_t*)raw.data())[i]); break;
        case DFGPU_UINT8: v.push_back(raw[(size_t)i]); break;
        default: v.push_back(((const int64_t*)raw.data())[i]); break;
      }
    }
    std::sort(v.begin(), v.end());
    v.erase(std::unique(v.begin(), v.end()), v.end());
    if ((int64_t)v.size() > max_distinct_values) return;
    *out_n = (int64_t)v.size();
    for (int64_t k = 0; k < (int64_t)v.size() && k < capacity && out_values; k++) out_values[k] = v[(size_t)k];
  });
}

int dfgpu_join_probe(dfgpu_join_t ht, dfgpu_table_t probe, const int* probe_key_cols, int join_type, const int* build_out_cols, int n_build_out,
                     const int* probe_out_cols, int n_probe_out, dfgpu_table_t* out) {
  return guarded([&] {
    require_init();
    DFGPU_CHECK(ht && out, "null argument");
    JoinTable* jt = unwrap_join(ht);
    DFGPU_CHECK(join_type >= DFGPU_JOIN_INNER && join_type <= DFGPU_JOIN_RIGHT_MARK, "bad join type");
    std::vector<int> pk(probe_key_cols, probe_key_cols + jt->key_cols.size());
    std::vector<int> bo(build_out_cols, build_out_cols + n_build_out), po(probe_out_cols, probe_out_cols + n_probe_out);
    const Table pt = with_build_dictionaries(*jt, *unwrap(probe), pk);
    auto o = std::make_unique<Table>(join_probe_null_aware(*jt, pt, pk, join_type, bo, po));
    *out = wrap(o.release());
  });
}

// ---- bounded probe (round 6): HashJoinStream emits at most batch_size rows per poll and resumes where it stopped
// (get_matched_indices_with_limit_offset + MapOffset, joins/join_hash_map.rs:389-484; hash_join/stream.rs:396-437).  The whole-table
// probe above materialises everything a probe table produces; an M:N blow-up beyond free HBM has to be taken in pieces instead.
// rows [begin, end) of a table as a zero-copy view (begin a multiple of 64: validity words and bit-packed values start on a word);
// Utf8 columns are copied (their offsets start at 0 by contract)
static Table view_rows(const Table& t, int64_t begin, int64_t end) {
  Table out;
  out.device = t.device;
  out.nrows = end - begin;
  for (const Column& c : t.cols) {
    if (c.field.type == DFGPU_UTF8) {
      out.cols.push_back(slice_strings(c, begin, end - begin));
      continue;
    }
    Column v = c;
    v.length = end - begin;
    v.stats.reset();
    v.data_offset = c.data_offset + (c.field.type == DFGPU_BOOL ? (size_t)(begin / 8) : (size_t)begin * type_width(c.field.type));
    if (c.validity) {
      v.validity = std::make_shared<DevBuf>((char*)c.validity->ptr + begin / 8, bitmap_bytes(end - begin), std::static_pointer_cast<void>(c.validity));
      v.null_count = -1;
    }
    out.cols.push_back(std::move(v));
  }
  return out;
}
// the number of leading 64-row words of the probe whose output stays within `limit` rows (one thread: a binary search over the prefix)
__global__ void k_words_within(const uint64_t* __restrict__ prefix, int64_t n_words, uint64_t limit, long long* __restrict__ out) {
  int64_t lo = 0, hi = n_words;   // the largest w with prefix[w] <= limit (prefix[0] = 0)
  while (lo < hi) {
    const int64_t m = (lo + hi + 1) >> 1;
    if (prefix[m] <= limit) lo = m;
    else hi = m - 1;
  }
  out[0] = lo;
  out[1] = (long long)prefix[lo];
}

// AND of `column <op> literal` / `literal <op> column` comparisons over fixed-width integer-like columns of `t` -> RowPred.
// false: the predicate has another shape (k_cmp and a row mask serve it).
static bool simple_row_pred(const dfgpu_expr& e, const Table& t, RowPred& out, int64_t& bytes_per_row) {
  out = RowPred{};
  bytes_per_row = 0;
  if (!e.nodes || e.root < 0 || e.root >= e.n_nodes) return false;
  std::vector<int> todo{e.root};
  while (!todo.empty()) {
    const int i = todo.back();
    todo.pop_back();
    if (i < 0 || i >= e.n_nodes) return false;
    const dfgpu_expr_node& nd = e.nodes[i];
    if (nd.op == DFGPU_EXPR_AND) {
      todo.push_back(nd.right);
      todo.push_back(nd.left);
      continue;
    }
    if (nd.op < DFGPU_EXPR_EQ || nd.op > DFGPU_EXPR_GE || nd.left < 0 || nd.right < 0 || nd.left >= e.n_nodes || nd.right >= e.n_nodes) return false;
    const dfgpu_expr_node *a = &e.nodes[nd.left], *b = &e.nodes[nd.right];
    int op = nd.op;
    if (a->op == DFGPU_EXPR_LITERAL && b->op == DFGPU_EXPR_COLUMN) {   // literal <op> column == column <mirrored op> literal
      std::swap(a, b);
      op = op == DFGPU_EXPR_LT ? DFGPU_EXPR_GT : op == DFGPU_EXPR_LE ? DFGPU_EXPR_GE : op == DFGPU_EXPR_GT ? DFGPU_EXPR_LT : op == DFGPU_EXPR_GE ? DFGPU_EXPR_LE : op;
    }
    auto int_like = [](int ty) { return ty == DFGPU_INT32 || ty == DFGPU_DATE32 || ty == DFGPU_INT64 || ty == DFGPU_UINT8 || ty == DFGPU_UINT32; };
    int lit_type = b->field.type;
    if (b->op == DFGPU_EXPR_CAST && b->left >= 0 && b->left < e.n_nodes && e.nodes[b->left].op == DFGPU_EXPR_LITERAL) {
      // CAST(integer literal AS the column's type) where the value is representable: the literal itself (expressions/cast.rs; the
      // planner usually folds it, a hand-built plan may not)
      const dfgpu_expr_node* l = &e.nodes[b->left];
      const long long v = (long long)l->lit_lo;
      const int to = b->field.type;
      const bool fits = to == DFGPU_INT64 || ((to == DFGPU_INT32 || to == DFGPU_DATE32) && v >= INT32_MIN && v <= INT32_MAX) || (to == DFGPU_UINT8 && v >= 0 && v <= 255) ||
                        (to == DFGPU_UINT32 && v >= 0 && v <= (long long)UINT32_MAX);
      if (!int_like(l->field.type) || !int_like(to) || !fits) return false;
      lit_type = to;
      b = l;
    }
    if (a->op != DFGPU_EXPR_COLUMN || b->op != DFGPU_EXPR_LITERAL || b->is_null || out.n >= ROWPRED_MAX) return false;
    if (a->column < 0 || a->column >= (int)t.cols.size()) return false;
    const Column& c = t.cols[(size_t)a->column];
    const int ty = c.field.type;
    if (c.dict || lit_type != ty) return false;    // (dictionary codes compare through their strings: evaluate() binds those)
    if (!int_like(ty)) return false;
    const int k = out.n++;
    out.data[k] = c.ptr();
    out.valid[k] = c.has_nulls() ? c.valid_words() : nullptr;
    out.lit[k] = (long long)b->lit_lo;    // sign-extended (signed types) / zero-extended (unsigned) to 128 bits by the caller: the low word is the value
    out.width[k] = (uint8_t)type_width(ty);
    out.is_unsigned[k] = ty == DFGPU_UINT8 || ty == DFGPU_UINT32;
    out.op[k] = (uint8_t)rowpred_sel(op);
    bytes_per_row += type_width(ty);
  }
  return out.n > 0;
}

int dfgpu_join_probe_bounded(dfgpu_join_t ht, dfgpu_table_t probe, const int* probe_key_cols, int join_type, const int* build_out_cols, int n_build_out,
                             const int* probe_out_cols, int n_probe_out, int64_t probe_offset, int64_t max_output_rows, dfgpu_table_t* out, int64_t* next_offset) {
  return guarded([&] {
    require_init();
    DFGPU_CHECK(ht && out && next_offset, "null argument");
    JoinTable* jt = unwrap_join(ht);
    DFGPU_CHECK(join_type >= DFGPU_JOIN_INNER && join_type <= DFGPU_JOIN_RIGHT_MARK, "bad join type");
    DFGPU_CHECK(max_output_rows >= 1, "dfgpu_join_probe_bounded: max_output_rows must be positive");
    std::vector<int> pk(probe_key_cols, probe_key_cols + jt->key_cols.size());
    std::vector<int> bo(build_out_cols, build_out_cols + n_build_out), po(probe_out_cols, probe_out_cols + n_probe_out);
    const Table whole = with_build_dictionaries(*jt, *unwrap(probe), pk);
    const int64_t n = whole.nrows;
    DFGPU_CHECK(probe_offset >= 0 && probe_offset <= n && (probe_offset % 64 == 0 || probe_offset == n), "dfgpu_join_probe_bounded: probe_offset must be 0, a value it returned, or the row count");
    Runtime& r = rt();
    // candidate rows: four times the output bound (a probe whose rows mostly miss takes more of them per call), whole words
    const int64_t left = n - probe_offset;
    int64_t cand = std::min<int64_t>(left, (std::max<int64_t>(max_output_rows, 64) + 63) / 64 * 64 * 4);
    int64_t take = cand;
    if (cand > 0 && jt->kind != KIND_RADIX && !jt->null_aware) {
      // output rows per 64-row word of the candidate (the general path's counting kernel: every table kind, every join type's
      // per-probe-row multiplicity, nothing marked), their prefix, and how many leading words stay within the bound
      const Table candt = view_rows(whole, probe_offset, probe_offset + cand);
      const int64_t n_words = (cand + 63) / 64;
      ProbeCtx ctx = make_ctx(*jt, candt, pk);
      BufPtr row_counts = make_buf((size_t)cand * 4), word_counts = make_buf((size_t)n_words * 4), prefix = make_buf((size_t)(n_words + 1) * 8), res = make_buf(16);
      {
        ProfileScope ps("join_probe_bound_count", cand * 8);
        with_kind(jt->kind, [&](auto kt) {
          k_probe_count<decltype(kt)::value><<<grid_for(n_words, BLOCK / WAVE), BLOCK, 0, r.stream>>>(ctx, cand, join_type, row_counts->as<uint32_t>(), word_counts->as<uint32_t>(), nullptr, nullptr);
        });
        DFGPU_HIP(hipGetLastError());
      }
      scan_u32(word_counts->as<uint32_t>(), n_words, prefix->as<uint64_t>());
      k_words_within<<<1, 1, 0, r.stream>>>(prefix->as<uint64_t>(), n_words, (uint64_t)max_output_rows, res->as<long long>());
      long long h[2] = {0, 0};
      d2h(h, res->ptr, 16);
      int64_t words = std::max<int64_t>(h[0], 1);   // (progress: one word is taken even if it alone exceeds the bound — up to 64 probe rows' matches)
      take = std::min<int64_t>(cand, words * 64);
    } else if (cand > 0) {
      // the LDS radix table answers pairs, not per-row counts (and a null-aware anti join looks at the whole probe side): the bound
      // applies to the PROBE rows taken per call
      take = std::min<int64_t>(cand, (std::max<int64_t>(max_output_rows, 64) + 63) / 64 * 64);
    }
    const Table part = view_rows(whole, probe_offset, probe_offset + take);
    auto o = std::make_unique<Table>(join_probe_null_aware(*jt, part, pk, join_type, bo, po));
    *next_offset = probe_offset + take;
    *out = wrap(o.release());
  });
}

int dfgpu_join_probe_filtered(dfgpu_join_t ht, dfgpu_table_t probe, const dfgpu_expr* probe_predicate, const int* probe_key_cols, int join_type,
                              const int* build_out_cols, int n_build_out, const int* probe_out_cols, int n_probe_out, dfgpu_table_t* out) {
  return guarded([&] {
    require_init();
    DFGPU_CHECK(ht && out && probe_predicate, "null argument");
    JoinTable* jt = unwrap_join(ht);
    DFGPU_CHECK(join_type >= DFGPU_JOIN_INNER && join_type <= DFGPU_JOIN_RIGHT_MARK, "bad join type");
    std::vector<int> pk(probe_key_cols, probe_key_cols + jt->key_cols.size());
    const Table pt = with_build_dictionaries(*jt, *unwrap(probe), pk);
    std::vector<int> bo(build_out_cols, build_out_cols + n_build_out), po(probe_out_cols, probe_out_cols + n_probe_out);
    // FilterExec predicate -> row mask (NULL => dropped, filter.rs:1396-1419) — evaluated only when somebody asks for the mask: a
    // predicate the selective probe's counts pass can evaluate per row (simple_row_pred) never becomes one
    BufPtr mask;
    auto make_mask = [&]() -> const uint64_t* {
      if (!mask) {
        Datum m = evaluate(*probe_predicate, pt);
        DFGPU_CHECK(m.col.field.type == DFGPU_BOOL, "Cannot create filter with non-boolean predicate");
        Column mc = datum_to_column(m, pt.nrows, "");
        mask = mc.data;
        if (mc.validity) {
          mask = make_buf(bitmap_bytes(pt.nrows));
          and_bitmaps(mc.data->as<uint64_t>(), mc.valid_words(), (pt.nrows + 63) / 64, mask->as<uint64_t>());
        }
      }
      return mask->as<uint64_t>();
    };
    LazyRowFilter lazy;
    lazy.mask = make_mask;
    const bool simple = !jt->null_aware && pt.nrows > 0 && option_on("join.pred_in_counts", true) && simple_row_pred(*probe_predicate, pt, lazy.pred, lazy.pred_bytes_per_row);
    if (!simple) make_mask();
    bool consumed = false;
    Table res;
    if (!jt->null_aware) res = join_probe(*jt, pt, pk, join_type, bo, po, simple ? nullptr : mask->as<uint64_t>(), &consumed, simple ? &lazy : nullptr);
    if (!consumed) make_mask();
    if (jt->null_aware) {
      // NOT IN semantics look at the NULL keys of the rows that pass the filter: materialise it first
      std::vector<int> all(pt.cols.size());
      for (size_t i = 0; i < all.size(); i++) all[i] = (int)i;
      Table filtered = compact_table(pt, all, mask->as<uint64_t>(), nullptr);
      res = join_probe_null_aware(*jt, filtered, pk, join_type, bo, po);
    } else if (!consumed) {
      // general path: FilterExec materialised, then the probe
      std::vector<int> all(pt.cols.size());
      for (size_t i = 0; i < all.size(); i++) all[i] = (int)i;
      Table filtered = compact_table(pt, all, mask->as<uint64_t>(), nullptr);
      res = join_probe(*jt, filtered, pk, join_type, bo, po);
    }
    DFGPU_HIP(hipStreamSynchronize(rt().stream));
    auto o = std::make_unique<Table>(std::move(res));
    *out = wrap(o.release());
  });
}

int dfgpu_join_probe_with_filter(dfgpu_join_t ht, dfgpu_table_t probe, const int* probe_key_cols, int join_type, const dfgpu_join_filter* filter,
                                 const int* build_out_cols, int n_build_out, const int* probe_out_cols, int n_probe_out, dfgpu_table_t* out) {
  return guarded([&] {
    require_init();
    DFGPU_CHECK(ht && out && filter && filter->n_columns >= 0, "null argument");
    JoinTable* jt = unwrap_join(ht);
    DFGPU_CHECK(join_type >= DFGPU_JOIN_INNER && join_type <= DFGPU_JOIN_RIGHT_MARK, "bad join type");
    std::vector<int> pk(probe_key_cols, probe_key_cols + jt->key_cols.size());
    std::vector<int> bo(build_out_cols, build_out_cols + n_build_out), po(probe_out_cols, probe_out_cols + n_probe_out);
    const Table pt = with_build_dictionaries(*jt, *unwrap(probe), pk);
    null_aware_before_probe(*jt, pt, pk, join_type, true);  // LeftAnti: flags only; RightAnti: rejected
    auto o = std::make_unique<Table>(join_probe_with_filter(*jt, pt, pk, join_type, bo, po, filter));
    *out = wrap(o.release());
  });
}

int dfgpu_join_emit_unmatched(dfgpu_join_t ht, int join_type, const int* build_out_cols, int n_build_out, const dfgpu_field* probe_fields,
                              const char* const* probe_names, int n_probe_out, dfgpu_table_t* out) {
  return guarded([&] {
    require_init();
    JoinTable* jt = unwrap_join(ht);
    DFGPU_CHECK(needs_visited(join_type), "join type has no build-side emission");
    Runtime& r = rt();
    const int64_t nb = jt->build.nrows;
    if (!jt->visited) jt->visited = make_zero_buf((size_t)nb + 64);
    std::vector<int> bo(build_out_cols, build_out_cols + n_build_out);
    auto o = std::make_unique<Table>();
    if (join_type == DFGPU_JOIN_LEFT_MARK) {
      o->nrows = nb;
      for (int c : bo) o->cols.push_back(jt->build.cols[c]);
      o->cols.push_back(mark_column(jt->visited->as<uint8_t>(), nb));
    } else {
      BufPtr mask = make_buf(bitmap_bytes(nb));
      int want_visited = join_type == DFGPU_JOIN_LEFT_SEMI;
      if (nb) k_visited_mask<<<grid_for((nb + 63) / 64, BLOCK / WAVE), BLOCK, 0, r.stream>>>(jt->visited->as<uint8_t>(), nb, want_visited, mask->as<uint64_t>());
      const uint64_t* also = nullptr;
      if (jt->null_aware) {
        DFGPU_CHECK(join_type == DFGPU_JOIN_LEFT_ANTI, "null_aware can only be true for LeftAnti joins and RightAnti joins with `CollectLeft` `PartitionMode`");
        // stream.rs:1016-1076: a NULL on the probe side makes every NOT IN unknown; else NULL build keys are unknown
        // unless the probe side was empty (NULL NOT IN (empty set) is TRUE)
        if (jt->probe_side_has_null) DFGPU_HIP(hipMemsetAsync(mask->ptr, 0, bitmap_bytes(nb), r.stream));
        else if (jt->probe_side_non_empty) also = jt->build.cols[jt->key_cols[0]].valid_words();
      }
      *o = compact_table(jt->build, bo, mask->as<uint64_t>(), also);
      if (join_type == DFGPU_JOIN_LEFT || join_type == DFGPU_JOIN_FULL) {
        // unmatched build rows carry an all-NULL probe side
        for (int i = 0; i < n_probe_out; i++) {
          if (probe_fields[i].type == DFGPU_UTF8) {   // all-NULL strings: zero offsets, no bytes
            Column c;
            c.field = probe_fields[i];
            c.field.nullable = 1;
            c.name = probe_names && probe_names[i] ? probe_names[i] : "";
            c.length = o->nrows;
            c.offsets = make_zero_buf((size_t)(o->nrows + 1) * 8 + 16);
            c.data = make_buf(16);
            c.validity = make_zero_buf(bitmap_bytes(o->nrows));
            c.null_count = o->nrows;
            o->cols.push_back(std::move(c));
            continue;
          }
          Column c = alloc_column(probe_fields[i], probe_names && probe_names[i] ? probe_names[i] : "", o->nrows);
          if (o->nrows) DFGPU_HIP(hipMemsetAsync(c.data->ptr, 0, data_bytes(c.field.type, o->nrows), r.stream));
          c.validity = make_zero_buf(bitmap_bytes(o->nrows));
          c.null_count = o->nrows;
          o->cols.push_back(std::move(c));
        }
      }
    }
    jt->info.output_rows += o->nrows;
    *out = wrap(o.release());
  });
}

// HashTableLookupExpr (hash_join/partitioned_hash_eval.rs:278): the Map strategy of the join's dynamic filter — is the row's key in the
// build side's table?  One Boolean per probe row (never NULL; a NULL key is FALSE unless NULL == NULL and the build side holds one).
int dfgpu_join_contains(dfgpu_join_t ht, dfgpu_table_t probe, const int* probe_key_cols, dfgpu_table_t* out) {
  return guarded([&] {
    require_init();
    DFGPU_CHECK(ht && probe_key_cols && out, "null argument");
    JoinTable* jt = unwrap_join(ht);
    DFGPU_CHECK(jt->kind != KIND_RADIX, "dfgpu_join_contains: the LDS radix table answers pairs, not membership (build with table_mode 0-3)");
    std::vector<int> pk(probe_key_cols, probe_key_cols + jt->key_cols.size());
    const Table pt = with_build_dictionaries(*jt, *unwrap(probe), pk);
    DFGPU_CHECK(pt.device == jt->build.device, "the probe table lives on another device than the join table");
    const int64_t np = pt.nrows, n_words = (np + 63) / 64;
    // membership needs no build row: over a rank map of keys in no order the bitmap alone answers (no permutation is built)
    ProbeCtx ctx = make_ctx(*jt, pt, pk, /*need_build_rows=*/false);
    dfgpu_field f{};
    f.type = DFGPU_BOOL;
    auto o = std::make_unique<Table>();
    o->nrows = np;
    o->cols.push_back(alloc_column(f, "contains", np));
    if (np > 0) {
      int64_t key_bytes = 0;
      for (int i = 0; i < ctx.pkeys.n; i++) key_bytes += np * ctx.pkeys.c[i].width;
      ProfileScope ps("join_contains", key_bytes + np / 8);
      const int g = grid_for(n_words, (BLOCK / WAVE) * PROBE_UNROLL);
      with_kind_and_key(jt->kind, ctx.pkeys.c[0].type, [&](auto kd, auto kt) {
        k_probe_first<decltype(kd)::value, decltype(kt)::value><<<g, BLOCK, 0, rt().stream>>>(ctx, np, 0, nullptr, o->cols[0].data->as<uint64_t>(), nullptr);
      });
      DFGPU_HIP(hipGetLastError());
    }
    *out = wrap(o.release());
  });
}

}  // extern "C"

// ---- the visited marks of a REPLICATED build side across ranks (exchange.hip dfgpu_exchange_join_visited): CollectLeft with
// build-side emission, the probe partitions on different GPUs.  In the reference all probe partitions mark ONE bitmap
// (hash_join/exec.rs:1312-1330); here every rank marks its copy, the copies are OR-ed, and the last step — reporting build rows by
// their marks — sees the union everywhere.  One bit per build row, then the null-aware flags (probe_side_has_null, non_empty).
namespace dfgpu {
__global__ __launch_bounds__(BLOCK) void k_visited_or_bits(const uint64_t* __restrict__ bits, int64_t n, uint8_t* __restrict__ visited) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK)
    if ((bits[i >> 6] >> (i & 63)) & 1ull) visited[i] = 1;
}
std::vector<uint8_t> join_visited_export(dfgpu_join_t h) {
  JoinTable* jt = unwrap_join(h);
  const int64_t nb = jt->build.nrows;
  const size_t bb = bitmap_bytes(nb);
  std::vector<uint8_t> out(bb + 8, 0);
  {
    std::lock_guard<std::mutex> lk(jt->mu);
    if (!jt->visited) jt->visited = make_zero_buf((size_t)nb + 64);
  }
  if (nb) {
    BufPtr bits = make_buf(bb);
    pack_bytes_to_bitmap(jt->visited->as<uint8_t>(), nb, bits->as<uint64_t>());
    d2h(out.data(), bits->ptr, bb);
  }
  out[bb] = jt->probe_side_has_null ? 1 : 0;
  out[bb + 1] = jt->probe_side_non_empty ? 1 : 0;
  return out;
}
void join_visited_merge(dfgpu_join_t h, const uint8_t* merged, size_t nbytes) {
  JoinTable* jt = unwrap_join(h);
  const int64_t nb = jt->build.nrows;
  const size_t bb = bitmap_bytes(nb);
  DFGPU_CHECK(nbytes == bb + 8, "join visited merge: the ranks' build sides differ in size (the build side must be replicated)");
  if (nb) {
    BufPtr bits = make_buf(bb);
    h2d_async(bits->ptr, merged, bb);
    k_visited_or_bits<<<grid_for(nb, BLOCK), BLOCK, 0, rt().stream>>>(bits->as<uint64_t>(), nb, jt->visited->as<uint8_t>());
    DFGPU_HIP(hipGetLastError());
    DFGPU_HIP(hipStreamSynchronize(rt().stream));  // `merged` is the caller's
  }
  jt->probe_side_has_null = jt->probe_side_has_null || merged[bb] != 0;
  jt->probe_side_non_empty = jt->probe_side_non_empty || merged[bb + 1] != 0;
}
}  // namespace dfgpu

extern "C" {

int dfgpu_join_get_info(dfgpu_join_t ht, dfgpu_join_info* out) {
  return guarded([&] { *out = reinterpret_cast<JoinTable*>(ht)->info; });
}
int dfgpu_join_free(dfgpu_join_t ht) {
  return guarded([&] {
    if (ht) delete unwrap_join(ht);  // buffers go back to the pool of the table's device
  });
}

}  // extern "C"
