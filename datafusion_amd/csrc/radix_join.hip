// radix_join.hip — the general hash join as LDS-staged hash-table partitions (BASELINE north_star).
//
// Reference semantics: JoinHashMap::update_from_iter + get_matched_indices (joins/join_hash_map.rs:307-338,389-484),
// chains of equal hashes (joins/chain.rs:29-69), key re-check equal_rows_arr (joins/utils.rs:2191-2260).  The reference keeps
// ONE table over the whole build side and walks it with dependent random loads; on MI355X a random access costs a 128-byte
// line of HBM (profiles/r2_fetch_calib.md) once the table outgrows the 256 MiB Infinity Cache, so here both sides are radix
// partitioned on the top bits of a mixed key until a build partition fits the LDS of one workgroup, and every partition
// pair is joined with a chained table that lives in LDS — the table walk never leaves the CU:
//   1. records  : (mixed key u64, row id u32) per row whose key can match (NULL keys dropped under NullEqualsNothing).
//                 One integer key column <= 64 bits: mixed key = fmix64(value) — a bijection, so equal mixed keys ARE equal
//                 keys and no re-check is needed.  Anything else (several columns, Decimal128, NULL == NULL): mixed key =
//                 the 64-bit row hash (create_hashes semantics, device.hpp hash_row) and hash-equal pairs are re-checked on
//                 the real key columns (equal_rows_arr), which is also what the `force_hash_collisions` test mode exercises.
//   2. partition: stable LSD radix passes (sort.hip k_rs_scatter2: ballot ranking, LDS-staged contiguous runs) over the top
//                 B bits, B chosen so that a build partition averages 1024 rows; partition p = key >> (64 - B).
//   3. join     : a task = (partition, <= 8192 probe rows of it).  The workgroup loads the build partition into LDS in chunks
//                 of 2048 rows (keys, row ids, u32 heads + u16 next chains: 44 KB, 3 workgroups per CU), streams its probe rows
//                 against it and counts matches; task counts are prefix-summed and a second identical walk emits the
//                 (build row, probe row) pairs at exact offsets (no atomics, exact allocation for M:N outputs).  Partitions
//                 that exceed a chunk (skew, duplicates, forced collisions) take several chunks per task — correct at any skew.
// The pairs feed the same per-JoinType finishing as the JoinFilter path (visited bytes, probe hit counts, unmatched rows),
// so all ten join types, NullEqualsNull and residual filters work on top of it.  Output order is partition order: valid where
// the plan does not observe HashJoinExec's probe-side order (it is chosen by table_mode 4, or by `auto` under probe_mode 4).
#include <algorithm>
#include <cstdlib>
#include <functional>

#include "device.hpp"
#include "internal.hpp"
#include "onesweep.hpp"

namespace dfgpu {

static bool force_fused_off() { return !option_on("join.radix_fused_emit", true); }   // (A/B switch: the pairs + gathers of round 5)
constexpr int RJ_CAP = 3072;     // build rows per LDS chunk (round 6: partitions average ~2300 rows, so two 8-bit passes reach them up to 157 M build rows)
constexpr int RJ_HEADS = 4096;   // chain heads (power of two, 2 x RJ_CAP)
constexpr int RJ_TASK_ROWS = 8192;
constexpr int RJ_STAGE = 1024;   // matches of one 256-row probe tile listed in LDS before they are written ({build position u16, probe row u32}: 6 KB; keys 24 + heads 16 + chains 6:
                                 // 52 KB, 3 workgroups per CU).  The M:N benchmark shape averages 384 matches per tile: a stage of 384 overflowed every other tile

struct RadixTask {
  uint32_t part;
  uint32_t rows;
  uint64_t q0;
};

// ---------------------------------------------------------------------------------------------- records
// EXACT: ks.n == 1, integer key.  `mask` / `prefix` (optional): rows that take part and their dense positions.
template <bool EXACT>
__global__ __launch_bounds__(BLOCK) void k_rj_records(KeySet ks, int64_t n, int force_collisions, const uint64_t* __restrict__ mask, const uint64_t* __restrict__ prefix,
                                                     uint64_t* __restrict__ key, uint32_t* __restrict__ rid) {
  const int64_t n_words = (n + 63) >> 6;
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * BLOCK) >> 6;
  for (int64_t w = wave; w < n_words; w += n_waves) {
    const int64_t i = (w << 6) + lane_id();
    uint64_t m = ~0ull;
    const int64_t rem = n - (w << 6);
    if (rem < 64) m = (~0ull) >> (64 - rem);
    if (mask) m &= mask[w];
    if (!((m >> lane_id()) & 1ull)) continue;
    const int64_t d = mask ? (int64_t)(prefix[w] + mbcnt(m)) : i;
    uint64_t k;
    if (EXACT) {
      uint64_t lo, hi;
      load_words(ks.c[0], i, lo, hi);
      k = fmix64(lo);
    } else {
      bool any_null;
      k = hash_row(ks, i, SEED_JOIN, any_null);
    }
    if (force_collisions) k = 0;
    key[d] = k;
    rid[d] = (uint32_t)i;
  }
}
// AND of the key columns' validity words (rows with a NULL key never match under NullEqualsNothing)
__global__ __launch_bounds__(BLOCK) void k_rj_valid_mask(KeySet ks, int64_t n_words, uint64_t* __restrict__ mask) {
  for (int64_t w = (int64_t)blockIdx.x * BLOCK + threadIdx.x; w < n_words; w += (int64_t)gridDim.x * BLOCK) {
    uint64_t m = ~0ull;
    for (int c = 0; c < ks.n; c++)
      if (ks.c[c].valid) m &= ks.c[c].valid[w];
    mask[w] = m;
  }
}
// ---------------------------------------------------------------------------------------------- the partitioner (round 6)
// Records partitioned on the top B bits of the mixed key by TWO stable 8-bit passes (three beyond 2^16 partitions) — round 5 ran three
// 6-bit passes of three kernels each over a record array made by a kernel of its own (116 bytes moved per row; this: 52):
//   k_rp_hist   one read of the key column: per-tile counts of the first digit (-> a scan gives every tile's offsets: the first pass waits
//               for nobody) and the totals of the later digits (-> their passes' bin bases);
//   k_rp_pass   BUILD: reads the KEY COLUMN, mixes, ranks (wave-private ballot ranking), stages the tile in LDS in digit order and writes
//               (key, row id) runs — no record array exists before the first scatter; later passes read records and find their offsets by
//               the decoupled look-back of onesweep.hpp.
//   k_rp_starts the partitions' first positions by binary search over the partitioned keys (65 K threads x 28 probes; the pass over
//               all keys that marked boundaries cost a read of the whole array).
constexpr int RP_MAX_PASSES = 3;
constexpr int RP_HIST_GROUP = 4;   // tiles a workgroup counts between two barriers
struct RpDigits {
  int shift[RP_MAX_PASSES], bits[RP_MAX_PASSES];
  int n;
};
template <bool EXACT>
__device__ __forceinline__ uint64_t rp_mixed_key(const KeySet& ks, int64_t i) {
  if (EXACT) {
    uint64_t lo, hi;
    load_words(ks.c[0], i, lo, hi);
    return fmix64(lo);
  }
  bool any_null;
  return hash_row(ks, i, SEED_JOIN, any_null);
}
// the keys of a lane's ITEMS rows, all loads in flight together where the key is one 64- or 32-bit integer column
template <bool EXACT, int ITEMS>
__device__ __forceinline__ void rp_tile_keys(const KeySet& ks, int64_t base, const int (&off)[ITEMS], uint64_t (&key)[ITEMS]) {
  if (EXACT && (ks.c[0].type == 2 || ks.c[0].type == 7)) {
    const uint64_t* col = reinterpret_cast<const uint64_t*>(ks.c[0].data) + base;
#pragma unroll
    for (int c = 0; c < ITEMS; c++) key[c] = col[off[c]];
#pragma unroll
    for (int c = 0; c < ITEMS; c++) key[c] = fmix64(key[c]);
  } else if (EXACT && (ks.c[0].type == 1 || ks.c[0].type == 8)) {
    int32_t v[ITEMS];
    const int32_t* col = reinterpret_cast<const int32_t*>(ks.c[0].data) + base;
#pragma unroll
    for (int c = 0; c < ITEMS; c++) v[c] = col[off[c]];
#pragma unroll
    for (int c = 0; c < ITEMS; c++) key[c] = fmix64((uint64_t)(int64_t)v[c]);
  } else {
#pragma unroll
    for (int c = 0; c < ITEMS; c++) key[c] = rp_mixed_key<EXACT>(ks, base + off[c]);
  }
}
template <bool EXACT, int ITEMS>
__global__ __launch_bounds__(BLOCK) void k_rp_hist(KeySet ks, const uint64_t* __restrict__ mask, int64_t n, int64_t n_tiles, RpDigits dg, uint32_t* __restrict__ counts,
                                                  unsigned long long* __restrict__ totals) {
  constexpr int TILE = BLOCK * ITEMS;
  __shared__ unsigned int s_h0[RP_HIST_GROUP][256];
  __shared__ unsigned int s_h[RP_MAX_PASSES - 1][256];
#pragma unroll
  for (int p = 0; p < RP_MAX_PASSES - 1; p++) s_h[p][threadIdx.x] = 0;
  const int64_t n_groups = (n_tiles + RP_HIST_GROUP - 1) / RP_HIST_GROUP;
  const unsigned m0 = (1u << dg.bits[0]) - 1u, m1 = (1u << dg.bits[1]) - 1u, m2 = (1u << dg.bits[2]) - 1u;
  for (int64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
#pragma unroll
    for (int u = 0; u < RP_HIST_GROUP; u++) s_h0[u][threadIdx.x] = 0;
    __syncthreads();
    for (int u = 0; u < RP_HIST_GROUP; u++) {
      const int64_t t = g * RP_HIST_GROUP + u;
      if (t >= n_tiles) break;
      const int64_t lo = t * TILE;
      int off[ITEMS];
      uint64_t key[ITEMS];
#pragma unroll
      for (int c = 0; c < ITEMS; c++) {
        const int64_t i = lo + c * BLOCK + threadIdx.x;
        off[c] = (int)((i < n ? i : n - 1) - lo);
      }
      rp_tile_keys<EXACT, ITEMS>(ks, lo, off, key);
#pragma unroll
      for (int c = 0; c < ITEMS; c++) {
        const int64_t i = lo + c * BLOCK + threadIdx.x;
        if (i >= n || (mask && !bit_at(mask, i))) continue;
        atomicAdd(&s_h0[u][(unsigned)(key[c] >> dg.shift[0]) & m0], 1u);
        if (dg.n > 1) atomicAdd(&s_h[0][(unsigned)(key[c] >> dg.shift[1]) & m1], 1u);
        if (dg.n > 2) atomicAdd(&s_h[1][(unsigned)(key[c] >> dg.shift[2]) & m2], 1u);
      }
    }
    __syncthreads();
    for (int u = 0; u < RP_HIST_GROUP; u++) {
      const int64_t t = g * RP_HIST_GROUP + u;
      if (t < n_tiles) counts[(int64_t)threadIdx.x * n_tiles + t] = s_h0[u][threadIdx.x];
    }
    __syncthreads();
  }
  for (int p = 1; p < dg.n; p++)
    if (s_h[p - 1][threadIdx.x]) atomicAdd(&totals[p * 256 + threadIdx.x], (unsigned long long)s_h[p - 1][threadIdx.x]);
}
// BUILD: tile t = source rows [t * TILE, ...), `offsets` (digit-major, scanned) holds every run's start; else: records in, look-back.
// THREADS x ITEMS rows per tile: 256 x 8 (2048 rows), 256 x 16 (4096 rows: 16-row runs, but ~200 VGPRs = 8 waves per CU) or 512 x 8 (the same
// tile at half the registers per thread: 16 waves per CU).  The digits' bookkeeping (counts, scan, offsets, look-back) belongs to the first
// 256 threads = waves 0-3; every wave ranks and stages its own 64 x ITEMS-row segment.
template <bool BUILD, bool EXACT, int ITEMS, int THREADS>
__global__ __launch_bounds__(THREADS, (THREADS == 512 ? 1 : ITEMS > 8 ? 2 : 4)) void k_rp_pass(KeySet ks, const uint64_t* __restrict__ mask, const uint64_t* __restrict__ key_in,
                                                     const uint32_t* __restrict__ rid_in, int64_t n, int shift, int bits, int64_t n_tiles, const uint64_t* __restrict__ offsets,
                                                     const unsigned long long* __restrict__ totals, uint32_t* __restrict__ tile_state, unsigned* __restrict__ ticket,
                                                     uint64_t* __restrict__ key_out, uint32_t* __restrict__ rid_out, int xcd_static) {
  constexpr int NWAVE = THREADS / WAVE;
  constexpr int TILE = THREADS * ITEMS;
  __shared__ uint64_t s_key[TILE];
  __shared__ uint32_t s_rid[TILE];
  __shared__ uint8_t s_dig[TILE];
  __shared__ uint16_t s_cnt[NWAVE][256];
  __shared__ uint16_t s_start[256];
  __shared__ unsigned int s_goff[256];
  __shared__ unsigned int s_base[256];
  __shared__ unsigned int s_wtot[4];
  __shared__ unsigned int s_tile, s_nst;
  const unsigned mask_d = (1u << bits) - 1u;
  const int wave = threadIdx.x >> 6;
  const unsigned lane = lane_id();
  const bool dig_thread = threadIdx.x < 256;   // (wave-uniform: waves 0-3)
  if (!BUILD) {   // bin bases of this pass: exclusive scan of the digit totals
    unsigned v = 0, inc = 0;
    if (dig_thread) {
      v = (unsigned)totals[threadIdx.x];
      inc = wave_inclusive_sum<unsigned>(v);
      if (lane == 63) s_wtot[wave] = inc;
    }
    __syncthreads();
    if (dig_thread) {
      unsigned b = 0;
      for (int w = 0; w < wave; w++) b += s_wtot[w];
      s_base[threadIdx.x] = b + inc - v;
    }
    __syncthreads();
  }
  for (int64_t round = 0;; round++) {
    if (!(BUILD && xcd_static) && threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
    for (int i = threadIdx.x; i < NWAVE * 256; i += THREADS) (&s_cnt[0][0])[i] = 0;
    __syncthreads();
    int64_t t;
    if (BUILD && xcd_static) {   // every XCD walks a contiguous eighth of the tiles: the runs neighbouring tiles append to a digit meet in ONE L2
      const int64_t b = (int64_t)blockIdx.x + round * (int64_t)gridDim.x;
      if (b >= n_tiles) return;
      t = xcd_tile(b, n_tiles);
    } else {
      t = (int64_t)s_tile;
      if (t >= n_tiles) return;
    }
    const int64_t lo = t * TILE;
    const int tile_rows = (int)((n - lo) < TILE ? (n - lo) : TILE);
    uint64_t key[ITEMS];
    uint32_t rid[BUILD ? 1 : ITEMS];   // (BUILD: a row's id is its position)
    unsigned dr[ITEMS];                // digit | rank << 8
    unsigned take = 0;                 // bit c: the lane's c-th row takes part
    if (BUILD) {
      int off[ITEMS];
#pragma unroll
      for (int c = 0; c < ITEMS; c++) {
        const int j = (wave * ITEMS + c) * WAVE + (int)lane;
        off[c] = j < tile_rows ? j : 0;
        if (j < tile_rows && (!mask || bit_at(mask, lo + off[c]))) take |= 1u << c;
      }
      rp_tile_keys<EXACT, ITEMS>(ks, lo, off, key);
    } else {
#pragma unroll
      for (int c = 0; c < ITEMS; c++) {
        const int j = (wave * ITEMS + c) * WAVE + (int)lane;
        const int64_t src = lo + (j < tile_rows ? j : 0);
        key[c] = key_in[src];
        rid[BUILD ? 0 : c] = rid_in[src];
        if (j < tile_rows) take |= 1u << c;
      }
    }
#pragma unroll
    for (int c = 0; c < ITEMS; c++) {
      const bool in = (take >> c) & 1u;
      const unsigned dig = in ? ((unsigned)(key[c] >> shift) & mask_d) : 0u;
      uint64_t peers = ballot64(in);
      for (int b = 0; b < bits; b++) {
        const uint64_t bal = ballot64((dig >> b) & 1u);
        peers &= ((dig >> b) & 1u) ? bal : ~bal;
      }
      const unsigned r_in_wave = mbcnt(peers);
      const unsigned base = s_cnt[wave][dig];
      if (in && r_in_wave == 0) s_cnt[wave][dig] = (uint16_t)(base + (unsigned)__popcll(peers));
      dr[c] = dig | ((base + r_in_wave) << 8);
    }
    __syncthreads();
    unsigned run = 0, inc = 0;   // thread d: digit d's rows in this tile
    if (dig_thread) {
#pragma unroll
      for (int w = 0; w < NWAVE; w++) {
        const unsigned v = s_cnt[w][threadIdx.x];
        s_cnt[w][threadIdx.x] = (uint16_t)run;
        run += v;
      }
      if (!BUILD && t > 0 && threadIdx.x <= mask_d) os_store(&tile_state[t * 256 + threadIdx.x], OS_AGG | run);
      inc = wave_inclusive_sum<unsigned>(run);
      if (lane == 63) s_wtot[wave] = inc;
    }
    __syncthreads();
    if (dig_thread) {
      unsigned base = 0;
      for (int w = 0; w < wave; w++) base += s_wtot[w];
      s_start[threadIdx.x] = (uint16_t)(base + inc - run);
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < ITEMS; c++) {
      if ((take >> c) & 1u) {
        const unsigned dig = dr[c] & 255u;
        const unsigned q = (unsigned)s_start[dig] + (unsigned)s_cnt[wave][dig] + (dr[c] >> 8);
        s_key[q] = key[c];
        s_rid[q] = BUILD ? (uint32_t)(lo + (wave * ITEMS + c) * WAVE + (int)lane) : rid[BUILD ? 0 : c];
        s_dig[q] = (uint8_t)dig;
      }
    }
    if (threadIdx.x <= mask_d) {
      if (BUILD) {
        s_goff[threadIdx.x] = (unsigned)offsets[(int64_t)threadIdx.x * n_tiles + t] - (unsigned)s_start[threadIdx.x];
      } else {
        const unsigned excl = os_look_back(tile_state, t, threadIdx.x);
        os_store(&tile_state[t * 256 + threadIdx.x], OS_PFX | (excl + run));
        s_goff[threadIdx.x] = s_base[threadIdx.x] + excl - (unsigned)s_start[threadIdx.x];
      }
    }
    if (threadIdx.x == 255) s_nst = (unsigned)s_start[255] + run;   // rows of the tile that take part (digit 255's run is the last)
    __syncthreads();
    const int n_staged = (int)s_nst;
    for (int qq = threadIdx.x; qq < n_staged; qq += THREADS) {
      const unsigned dst = (unsigned)qq + s_goff[s_dig[qq]];
      key_out[dst] = s_key[qq];   // (plain stores: neighbouring tiles' runs of a digit meet in the L2 — non-temporal ones cost + 30 % at 4096-row
      rid_out[dst] = s_rid[qq];   //  tiles and + 90 % at 2048-row tiles)
    }
    __syncthreads();
  }
}
// starts[p] = first position whose partition id (key >> shift) is >= p, p in [0, P]
__global__ __launch_bounds__(BLOCK) void k_rp_starts(const uint64_t* __restrict__ key, int64_t n, int shift, int64_t P, unsigned long long* __restrict__ starts) {
  for (int64_t p = (int64_t)blockIdx.x * BLOCK + threadIdx.x; p <= P; p += (int64_t)gridDim.x * BLOCK) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if ((int64_t)(shift >= 64 ? 0ull : key[mid] >> shift) < p) lo = mid + 1;
      else hi = mid;
    }
    starts[p] = (unsigned long long)lo;
  }
}

struct RadixSide {
  BufPtr key, rid, starts;
  std::vector<uint64_t> h_starts;  // P + 1 entries
  int64_t n = 0;                   // records (rows that take part)
};
struct RadixTable {
  int bits = 0;
  bool exact = false;
  RadixSide build;
};

static KeySet rj_keyset(const Table& t, const std::vector<int>& cols) {
  KeySet ks{};
  DFGPU_CHECK((int)cols.size() <= MAX_KEYS && !cols.empty(), "bad number of join key columns");
  ks.n = (int)cols.size();
  for (int i = 0; i < ks.n; i++) {
    DFGPU_CHECK(cols[i] >= 0 && cols[i] < (int)t.cols.size(), "join key column index out of range");
    const Column& c = t.cols[cols[i]];
    DFGPU_CHECK(c.field.type != DFGPU_BOOL, "Boolean join keys are not supported on the GPU path");
    ks.c[i] = KeyCol{c.ptr(), c.valid_words(), c.field.type, type_width(c.field.type)};
  }
  return ks;
}

// records of one side, partitioned by the top `bits` bits of the mixed key
static RadixSide rj_partition(const Table& t, const std::vector<int>& key_cols, int bits, bool exact, bool null_equals_null, bool force_collisions, const char* what) {
  Runtime& r = rt();
  RadixSide s;
  const int64_t n = t.nrows;
  DFGPU_CHECK(n < 0xFFFFFFFFll, "the radix join addresses rows with 32 bits");
  const int64_t P = (int64_t)1 << bits;
  KeySet ks = rj_keyset(t, key_cols);
  bool nullable = false;
  for (int i = 0; i < ks.n; i++) nullable |= ks.c[i].valid != nullptr;
  const int64_t n_words = (n + 63) / 64;
  BufPtr mask, prefix;
  s.n = n;
  // NULL keys match nothing under NullEqualsNothing; an EXACT table is only built when NULL == NULL cannot match either
  // (no NULL build key), so NULL probe keys are dropped there too (the value under a NULL is arbitrary)
  if (nullable && (!null_equals_null || exact) && n) {
    mask = make_buf(bitmap_bytes(n));
    k_rj_valid_mask<<<grid_for(n_words, BLOCK), BLOCK, 0, r.stream>>>(ks, n_words, mask->as<uint64_t>());
  }
  int64_t key_bytes = 0;
  for (int i = 0; i < ks.n; i++) key_bytes += n * ks.c[i].width;
  const uint64_t* mk = mask ? mask->as<uint64_t>() : nullptr;
  const bool two_pass = bits > 0 && n > 1 && n < ((int64_t)1 << 30) && option_on("join.radix_onesweep", true);
  if (two_pass) {
    // (round 6) the partitioner above: digits of at most 8 bits over the top `bits` bits, least significant first
    RpDigits dg{};
    dg.n = (bits + 7) / 8;
    for (int p = 0, pos = 64 - bits; p < dg.n; p++) {
      const int b = (64 - pos + (dg.n - p) - 1) / (dg.n - p);
      dg.shift[p] = pos;
      dg.bits[p] = b;
      pos += b;
    }
    // tile shape: 512 threads x 8 rows (default), 256 x 16, or 256 x 8 — k_rp_pass
    const int threads = option_int("join.radix_tile_threads", 512) >= 512 ? 512 : 256;
    const int items = threads == 512 ? 8 : (option_int("join.radix_tile_items", 16) > 8 ? 16 : 8);
    const int64_t tile = (int64_t)threads * items;
    const int hist_items = (int)(tile / BLOCK);   // (the counting pass walks the same tiles with 256 threads)
    const int64_t n_tiles = (n + tile - 1) / tile;
    BufPtr counts = make_buf((size_t)256 * n_tiles * 4), offsets = make_buf(((size_t)256 * n_tiles + 1) * 8);
    BufPtr totals = make_zero_buf((size_t)RP_MAX_PASSES * 256 * 8 + 64);   // + the passes' tickets
    unsigned* tickets = reinterpret_cast<unsigned*>(totals->as<unsigned long long>() + RP_MAX_PASSES * 256);
    const int wg_per_cu = threads == 512 ? 2 : items > 8 ? 2 : 4;
    auto with_shape = [&](auto f) {   // f(exact, items, threads)
      if (threads == 512) {
        if (exact) f(std::true_type{}, std::integral_constant<int, 8>{}, std::integral_constant<int, 512>{});
        else f(std::false_type{}, std::integral_constant<int, 8>{}, std::integral_constant<int, 512>{});
      } else if (items > 8) {
        if (exact) f(std::true_type{}, std::integral_constant<int, 16>{}, std::integral_constant<int, 256>{});
        else f(std::false_type{}, std::integral_constant<int, 16>{}, std::integral_constant<int, 256>{});
      } else {
        if (exact) f(std::true_type{}, std::integral_constant<int, 8>{}, std::integral_constant<int, 256>{});
        else f(std::false_type{}, std::integral_constant<int, 8>{}, std::integral_constant<int, 256>{});
      }
    };
    {
      ProfileScope ps(what[0] == 'b' ? "radix_join_build_hist" : "radix_join_probe_hist", key_bytes);
      const int64_t n_groups = (n_tiles + RP_HIST_GROUP - 1) / RP_HIST_GROUP;
      const int hgrid = (int)std::min<int64_t>(n_groups, (int64_t)r.num_cus * 64);
      if (hist_items > 8) {
        if (exact) k_rp_hist<true, 16><<<hgrid, BLOCK, 0, r.stream>>>(ks, mk, n, n_tiles, dg, counts->as<uint32_t>(), totals->as<unsigned long long>());
        else k_rp_hist<false, 16><<<hgrid, BLOCK, 0, r.stream>>>(ks, mk, n, n_tiles, dg, counts->as<uint32_t>(), totals->as<unsigned long long>());
      } else {
        if (exact) k_rp_hist<true, 8><<<hgrid, BLOCK, 0, r.stream>>>(ks, mk, n, n_tiles, dg, counts->as<uint32_t>(), totals->as<unsigned long long>());
        else k_rp_hist<false, 8><<<hgrid, BLOCK, 0, r.stream>>>(ks, mk, n, n_tiles, dg, counts->as<uint32_t>(), totals->as<unsigned long long>());
      }
      DFGPU_HIP(hipGetLastError());
      scan_u32(counts->as<uint32_t>(), (int64_t)256 * n_tiles, offsets->as<uint64_t>());
    }
    if (mask) s.n = (int64_t)read_u64(offsets->as<uint64_t>() + (int64_t)256 * n_tiles);
    const int64_t cap = std::max<int64_t>(mask ? s.n : n, 1);
    BufPtr k0 = make_buf((size_t)cap * 8), r0 = make_buf((size_t)cap * 4), k1, r1, state;
    if (dg.n > 1 && s.n > 0) {
      k1 = make_buf((size_t)cap * 8);
      r1 = make_buf((size_t)cap * 4);
    }
    if (s.n > 0) {
      {
        ProfileScope ps("radix_join_partition_pass", key_bytes + s.n * 12);
        const int xcd_static = option_on("join.radix_xcd", true) ? 1 : 0;
        const int grid = (int)std::min<int64_t>((n_tiles + 7) / 8 * 8, (int64_t)r.num_cus * wg_per_cu);
        with_shape([&](auto ex, auto it, auto th) {
          k_rp_pass<true, decltype(ex)::value, decltype(it)::value, decltype(th)::value><<<grid, threads, 0, r.stream>>>(ks, mk, nullptr, nullptr, n, dg.shift[0], dg.bits[0], n_tiles, offsets->as<uint64_t>(),
                                                                                                   nullptr, nullptr, tickets, k0->as<uint64_t>(), r0->as<uint32_t>(), xcd_static);
        });
        DFGPU_HIP(hipGetLastError());
      }
      const int64_t rec_tiles = (s.n + tile - 1) / tile;
      const int rgrid = (int)std::min<int64_t>(rec_tiles, (int64_t)r.num_cus * wg_per_cu);
      for (int p = 1; p < dg.n; p++) {
        ProfileScope ps("radix_join_partition_pass", s.n * 24);
        if (!state) state = make_buf((size_t)rec_tiles * 256 * 4);
        DFGPU_HIP(hipMemsetAsync(state->ptr, 0, (size_t)rec_tiles * 256 * 4, r.stream));
        with_shape([&](auto ex, auto it, auto th) {
          (void)ex;
          k_rp_pass<false, true, decltype(it)::value, decltype(th)::value><<<rgrid, threads, 0, r.stream>>>(ks, nullptr, k0->as<uint64_t>(), r0->as<uint32_t>(), s.n, dg.shift[p], dg.bits[p], rec_tiles, nullptr,
                                                                                      totals->as<unsigned long long>() + p * 256, state->as<uint32_t>(), tickets + p, k1->as<uint64_t>(),
                                                                                      r1->as<uint32_t>(), 0);
        });
        DFGPU_HIP(hipGetLastError());
        std::swap(k0, k1);
        std::swap(r0, r1);
      }
    }
    s.key = k0;
    s.rid = r0;
  } else {
  s.key = make_buf((size_t)std::max<int64_t>(s.n, 1) * 8);
  s.rid = make_buf((size_t)std::max<int64_t>(s.n, 1) * 4);
  if (nullable && (!null_equals_null || exact) && n) {   // (the record kernel wants the rows' dense positions)
    prefix = make_buf((size_t)(n_words + 1) * 8);
    scan_mask_popcounts(mask->as<uint64_t>(), nullptr, n, prefix->as<uint64_t>());
    s.n = (int64_t)read_u64(prefix->as<uint64_t>() + n_words);
  }
  if (n) {
    ProfileScope ps(what[0] == 'b' ? "radix_join_build_records" : "radix_join_probe_records", key_bytes + s.n * 12);
    const int g = grid_for(n_words, BLOCK / WAVE);
    const uint64_t* pf = prefix ? prefix->as<uint64_t>() : nullptr;
    if (exact) k_rj_records<true><<<g, BLOCK, 0, r.stream>>>(ks, n, force_collisions, mk, pf, s.key->as<uint64_t>(), s.rid->as<uint32_t>());
    else k_rj_records<false><<<g, BLOCK, 0, r.stream>>>(ks, n, force_collisions, mk, pf, s.key->as<uint64_t>(), s.rid->as<uint32_t>());
    DFGPU_HIP(hipGetLastError());
  }
  if (bits > 0 && s.n > 1) radix_sort_pairs(s.key, s.rid, s.n, 64 - bits, bits);
  }
  // partition boundaries
  s.h_starts.assign((size_t)P + 1, 0);
  if (s.n) {
    BufPtr st = make_buf((size_t)(P + 1) * 8);
    k_rp_starts<<<grid_for(P + 1, BLOCK), BLOCK, 0, r.stream>>>(s.key->as<uint64_t>(), s.n, 64 - bits, P, st->as<unsigned long long>());
    DFGPU_HIP(hipGetLastError());
    d2h(s.h_starts.data(), st->ptr, (size_t)(P + 1) * 8);
    s.starts = st;
  } else {
    s.starts = make_zero_buf((size_t)(P + 1) * 8);
  }
  DFGPU_HIP(hipStreamSynchronize(r.stream));
  return s;
}

// ---------------------------------------------------------------------------------------------- the LDS join
struct RjVerify {
  KeySet bkeys, pkeys;
  int null_equals_null;
};
// The output columns of an INNER join written by the emit walk itself (round 6): a pair's build row and probe row are in registers the
// moment the pair is found — its columns are gathered and written right there, and the (build row, probe row) list — 16 bytes a pair
// written, then read back by one gather per column — never exists (M:N at SF100 sizes, 300 M pairs: 4.8 GB of pairs each way).  A column
// that IS the join key (EXACT records: mixed key = fmix64(value), a bijection) comes out of the record's key by unfmix64: no access at all.
constexpr int RJ_MAX_COLS = 12;
struct RjCols {
  const void* src[RJ_MAX_COLS];
  void* dst[RJ_MAX_COLS];
  int width[RJ_MAX_COLS];
  int n_build;   // the first n_build entries read the build row, the others the probe row
  int n;
  unsigned key_cols;   // bit c: entry c is the join key column (EXACT): the value is unfmix64(record key)
  int need_brow;       // some build column is not the key: the build row id is needed
};
__device__ __forceinline__ uint64_t unfmix64(uint64_t x) {   // fmix64's inverse (the multipliers' inverses mod 2^64; x ^= x >> 33 undoes itself)
  x ^= x >> 33; x *= 0x9cb4b2f8129337dbULL;
  x ^= x >> 33; x *= 0x4f74430c22a54005ULL;
  x ^= x >> 33;
  return x;
}
__device__ __forceinline__ void rj_emit_row(const RjCols& cols, uint64_t key, uint32_t brow, uint32_t prow, unsigned long long o) {
  for (int c = 0; c < cols.n; c++) {
    const int w = cols.width[c];
    if ((cols.key_cols >> c) & 1u) {
      const uint64_t v = unfmix64(key);
      if (w == 8) reinterpret_cast<uint64_t*>(cols.dst[c])[o] = v;
      else if (w == 4) reinterpret_cast<uint32_t*>(cols.dst[c])[o] = (uint32_t)v;
      else reinterpret_cast<uint8_t*>(cols.dst[c])[o] = (uint8_t)v;
      continue;
    }
    const int64_t s = c < cols.n_build ? (int64_t)brow : (int64_t)prow;
    switch (w) {
      case 16: reinterpret_cast<uint4*>(cols.dst[c])[o] = reinterpret_cast<const uint4*>(cols.src[c])[s]; break;
      case 8: reinterpret_cast<uint64_t*>(cols.dst[c])[o] = reinterpret_cast<const uint64_t*>(cols.src[c])[s]; break;
      case 4: reinterpret_cast<uint32_t*>(cols.dst[c])[o] = reinterpret_cast<const uint32_t*>(cols.src[c])[s]; break;
      default: reinterpret_cast<uint8_t*>(cols.dst[c])[o] = reinterpret_cast<const uint8_t*>(cols.src[c])[s]; break;
    }
  }
}
// EMIT: 0 = count the task's matches, 1 = write (build row, probe row) pairs, 2 = write the output columns (RjCols).
// ONE chain walk per probe row in either mode, and no workgroup barrier inside a task's probe loop (round 6; round 5's emit counted a
// tile's matches, scanned the counts across the workgroup and walked again): the walk is wave-uniform — a step's matches are ranked by
// ballot and listed in the WAVE's own stage in LDS ({build position, probe row}, RJ_WSTAGE entries) — and a wave that has finished its 64
// probe rows (or filled its stage) takes its output range from the task's cursor (one LDS atomic) and writes its list out, consecutive
// lanes to consecutive rows.  The order of a task's output rows is therefore arbitrary within the task (the radix join's output order is
// not the probe's anyway).
constexpr int RJ_WSTAGE = RJ_STAGE / (BLOCK / WAVE);
// BROW (EMIT == 2): some output column reads the build ROW (a payload column): its id is fetched in the write-out.  A compile-time switch: as a
// run-time one the compiler guarded the unused load's register with s_waitcnt vmcnt(0) — every write-out round then waited for the stores of
// the round before and for the prefetched probe tile (3.7 ms for the walk that costs the count pass 1.4).
template <bool EXACT, int EMIT, bool BROW = true>
__global__ __launch_bounds__(BLOCK) void k_rj_join(const uint64_t* __restrict__ bkey, const uint32_t* __restrict__ brid, const uint64_t* __restrict__ bstart,
                                                  const uint64_t* __restrict__ pkey, const uint32_t* __restrict__ prid, const RadixTask* __restrict__ tasks,
                                                  int64_t n_tasks, RjVerify v, unsigned long long* __restrict__ task_counts,
                                                  const uint64_t* __restrict__ task_off, int64_t* __restrict__ out_b, int64_t* __restrict__ out_p, RjCols cols = RjCols{}) {
  __shared__ uint64_t s_key[RJ_CAP];
  __shared__ uint32_t s_head[RJ_HEADS];
  __shared__ uint16_t s_next[RJ_CAP];
  __shared__ unsigned long long s_wtot[BLOCK / WAVE];
  __shared__ uint16_t s_ob[EMIT ? RJ_STAGE : 1];   // the emit walks' stages, one per wave: a match = (build row's position in the chunk, probe row)
  __shared__ uint32_t s_op[EMIT ? RJ_STAGE : 1];
  __shared__ unsigned long long s_run;             // EMIT: next free output position of the task
  const unsigned lane = lane_id();
  const int wave = threadIdx.x >> 6;
  uint16_t* w_ob = s_ob + (EMIT ? wave * RJ_WSTAGE : 0);
  uint32_t* w_op = s_op + (EMIT ? wave * RJ_WSTAGE : 0);
  for (int64_t t = blockIdx.x; t < n_tasks; t += gridDim.x) {
    const RadixTask task = tasks[t];
    const uint64_t b0 = bstart[task.part], b1 = bstart[task.part + 1];
    unsigned long long mine = 0;                                   // COUNT: this thread's matches over the whole task
    if (EMIT && threadIdx.x == 0) s_run = task_off[t];
    for (uint64_t c0 = b0; c0 < b1; c0 += RJ_CAP) {
      const int nbk = (int)((b1 - c0) < (uint64_t)RJ_CAP ? (b1 - c0) : (uint64_t)RJ_CAP);
      for (int i = threadIdx.x; i < RJ_HEADS; i += BLOCK) s_head[i] = 0;
      __syncthreads();
      for (int i = threadIdx.x; i < nbk; i += BLOCK) {
        const uint64_t k = bkey[c0 + i];
        s_key[i] = k;
        const uint32_t old = atomicExch(&s_head[(uint32_t)k & (RJ_HEADS - 1)], (uint32_t)i + 1u);
        s_next[i] = (uint16_t)old;
      }
      __syncthreads();
      // the NEXT tile's probe records are requested before the current tile is walked: one HBM latency per 256-row tile, exposed, was
      // 1.5 ms of each walk (781 K tiles over 768 resident workgroups, ~1.5 us each)
      uint64_t k_nx = 0;
      uint32_t pr_nx = 0;
      if (threadIdx.x < task.rows) {
        k_nx = pkey[task.q0 + threadIdx.x];
        pr_nx = prid[task.q0 + threadIdx.x];
      }
      for (uint32_t r0 = 0; r0 < task.rows; r0 += BLOCK) {
        const uint32_t j = r0 + threadIdx.x;
        const bool in = j < task.rows;
        const uint64_t k = k_nx;
        const uint32_t pr = pr_nx;
        if (j + BLOCK < task.rows) {
          k_nx = pkey[task.q0 + j + BLOCK];
          pr_nx = prid[task.q0 + j + BLOCK];
        }
        uint32_t cur = in ? s_head[(uint32_t)k & (RJ_HEADS - 1)] : 0u;
        if (!EMIT) {
          while (cur) {
            const uint32_t bi = cur - 1;
            if (s_key[bi] == k && (EXACT || keys_equal(v.bkeys, (int64_t)brid[c0 + bi], v.pkeys, (int64_t)pr, v.null_equals_null != 0, true))) mine++;
            cur = s_next[bi];
          }
          continue;
        }
        unsigned listed = 0;   // matches in the wave's stage (uniform)
        auto flush = [&]() {
          unsigned long long base = 0;
          if (lane == 0) base = atomicAdd(&s_run, (unsigned long long)listed);
          base = __shfl(base, 0, 64);
          for (unsigned q = lane; q < listed; q += WAVE) {
            const uint32_t bi = w_ob[q], prow = w_op[q];
            if (EMIT == 2) {   // (EXACT: the record key of a match is the build row's)
              const uint32_t brow = BROW ? brid[c0 + bi] : 0u;
              rj_emit_row(cols, s_key[bi], brow, prow, base + q);
            } else {
              out_b[base + q] = (int64_t)brid[c0 + bi];
              out_p[base + q] = (int64_t)prow;
            }
          }
          listed = 0;
        };
        while (ballot64(cur != 0)) {   // (wave-uniform)
          bool hit = false;
          uint32_t bi = 0;
          if (cur) {
            bi = cur - 1;
            hit = s_key[bi] == k && (EXACT || keys_equal(v.bkeys, (int64_t)brid[c0 + bi], v.pkeys, (int64_t)pr, v.null_equals_null != 0, true));
            cur = s_next[bi];
          }
          const uint64_t hm = ballot64(hit);
          if (hm == 0) continue;
          const unsigned nh = (unsigned)__popcll(hm);
          if (listed + nh > (unsigned)RJ_WSTAGE) flush();
          if (hit) {
            const unsigned slot = listed + mbcnt(hm);
            w_ob[slot] = (uint16_t)bi;
            w_op[slot] = pr;
          }
          listed += nh;
        }
        if (listed) flush();
      }
      __syncthreads();  // the next chunk overwrites the table
    }
    if (!EMIT) {
      mine = wave_sum<unsigned long long>(mine);
      if (lane == 0) s_wtot[wave] = mine;
      __syncthreads();
      if (threadIdx.x == 0) {
        unsigned long long tot = 0;
        for (int w = 0; w < BLOCK / WAVE; w++) tot += s_wtot[w];
        task_counts[t] = tot;
      }
      __syncthreads();
    }
  }
}

// ---------------------------------------------------------------------------------------------- host
std::shared_ptr<RadixTable> radix_join_build(const Table& build, const std::vector<int>& key_cols, bool null_equals_null, bool force_collisions) {
  auto rt_ = std::make_shared<RadixTable>();
  KeySet ks = rj_keyset(build, key_cols);
  bool nullable = false;
  for (int i = 0; i < ks.n; i++) nullable |= ks.c[i].valid != nullptr;
  rt_->exact = ks.n == 1 && is_integer_like(ks.c[0].type) && !(null_equals_null && nullable) && !force_collisions;
  // ~2300 build rows per partition on average at most (RJ_CAP less five standard deviations of a partition's size when every key comes
  // three times): a partition that needs several LDS chunks per task costs far more than another pass over HBM (profiles/r2_radix_sweep.md)
  const int max_bits = 24;
  int bits = 0;
  const int64_t part_rows = std::max<int64_t>(1, option_int("join.radix_partition_rows", 2400));   // (test hook: several passes over a small input)
  while (bits < max_bits && (part_rows << bits) < build.nrows) bits++;
  if (force_collisions) bits = 0;
  rt_->bits = bits;
  rt_->build = rj_partition(build, key_cols, bits, rt_->exact, null_equals_null, force_collisions, "build");
  return rt_;
}

int64_t radix_join_table_bytes(const RadixTable& t) { return t.build.n * 12 + (((int64_t)1 << t.bits) + 1) * 8; }
int radix_join_bits(const RadixTable& t) { return t.bits; }

// inner pairs (build row, probe row) of all key-equal rows, in partition order — or, with `columns` (an INNER join's output columns: a
// callback that allocates them for m rows and fills in the destinations), the joined rows themselves
static void radix_join_run(const RadixTable& t, const Table& build, const std::vector<int>& build_keys, const Table& probe, const std::vector<int>& probe_keys,
                           bool null_equals_null, bool force_collisions, BufPtr& out_b, BufPtr& out_p, int64_t& m, const std::function<RjCols(int64_t)>* columns) {
  Runtime& r = rt();
  m = 0;
  out_b = make_buf(8);
  out_p = make_buf(8);
  if (t.build.n == 0 || probe.nrows == 0) return;
  // the probe key types must allow the build side's choice of record key
  KeySet pk = rj_keyset(probe, probe_keys), bk = rj_keyset(build, build_keys);
  RadixSide ps = rj_partition(probe, probe_keys, t.bits, t.exact, null_equals_null, force_collisions, "probe");
  if (ps.n == 0) return;
  // tasks: every partition with rows on both sides, its probe rows cut into pieces
  std::vector<RadixTask> tasks;
  const int64_t P = (int64_t)1 << t.bits;
  for (int64_t p = 0; p < P; p++) {
    const uint64_t b0 = t.build.h_starts[(size_t)p], b1 = t.build.h_starts[(size_t)p + 1], q0 = ps.h_starts[(size_t)p], q1 = ps.h_starts[(size_t)p + 1];
    if (b0 == b1 || q0 == q1) continue;
    for (uint64_t q = q0; q < q1; q += RJ_TASK_ROWS) tasks.push_back(RadixTask{(uint32_t)p, (uint32_t)std::min<uint64_t>(RJ_TASK_ROWS, q1 - q), q});
  }
  if (tasks.empty()) return;
  const int64_t nt = (int64_t)tasks.size();
  BufPtr d_tasks = make_buf((size_t)nt * sizeof(RadixTask));
  DFGPU_HIP(hipMemcpyAsync(d_tasks->ptr, tasks.data(), (size_t)nt * sizeof(RadixTask), hipMemcpyHostToDevice, r.stream));
  BufPtr d_counts = make_buf((size_t)nt * 8), d_off = make_buf((size_t)nt * 8);
  RjVerify v{bk, pk, null_equals_null ? 1 : 0};
  const int grid = (int)std::min<int64_t>(nt, (int64_t)r.num_cus * 12);
  const int64_t rec_bytes = (t.build.n + ps.n) * 12;
  {
    ProfileScope psc("radix_join_count", rec_bytes);
    if (t.exact)
      k_rj_join<true, 0><<<grid, BLOCK, 0, r.stream>>>(t.build.key->as<uint64_t>(), t.build.rid->as<uint32_t>(), t.build.starts->as<uint64_t>(), ps.key->as<uint64_t>(),
                                                           ps.rid->as<uint32_t>(), d_tasks->as<RadixTask>(), nt, v, d_counts->as<unsigned long long>(), nullptr, nullptr, nullptr);
    else
      k_rj_join<false, 0><<<grid, BLOCK, 0, r.stream>>>(t.build.key->as<uint64_t>(), t.build.rid->as<uint32_t>(), t.build.starts->as<uint64_t>(), ps.key->as<uint64_t>(),
                                                            ps.rid->as<uint32_t>(), d_tasks->as<RadixTask>(), nt, v, d_counts->as<unsigned long long>(), nullptr, nullptr, nullptr);
    DFGPU_HIP(hipGetLastError());
  }
  std::vector<uint64_t> counts((size_t)nt), off((size_t)nt);
  d2h(counts.data(), d_counts->ptr, (size_t)nt * 8);
  uint64_t tot = 0;
  for (int64_t i = 0; i < nt; i++) {
    off[(size_t)i] = tot;
    tot += counts[(size_t)i];
  }
  m = (int64_t)tot;
  if (m == 0) return;
  DFGPU_HIP(hipMemcpyAsync(d_off->ptr, off.data(), (size_t)nt * 8, hipMemcpyHostToDevice, r.stream));
  if (columns) {
    const RjCols cols = (*columns)(m);
    int64_t row_bytes = 0;
    for (int c = 0; c < cols.n; c++) row_bytes += cols.width[c];
    ProfileScope pse("radix_join_emit_columns", rec_bytes + m * row_bytes * 2);
    DFGPU_CHECK(t.exact, "internal: the fused column emit needs exact record keys");
    if (cols.need_brow)
      k_rj_join<true, 2, true><<<grid, BLOCK, 0, r.stream>>>(t.build.key->as<uint64_t>(), t.build.rid->as<uint32_t>(), t.build.starts->as<uint64_t>(), ps.key->as<uint64_t>(),
                                                             ps.rid->as<uint32_t>(), d_tasks->as<RadixTask>(), nt, v, nullptr, d_off->as<uint64_t>(), nullptr, nullptr, cols);
    else
      k_rj_join<true, 2, false><<<grid, BLOCK, 0, r.stream>>>(t.build.key->as<uint64_t>(), t.build.rid->as<uint32_t>(), t.build.starts->as<uint64_t>(), ps.key->as<uint64_t>(),
                                                              ps.rid->as<uint32_t>(), d_tasks->as<RadixTask>(), nt, v, nullptr, d_off->as<uint64_t>(), nullptr, nullptr, cols);
    DFGPU_HIP(hipGetLastError());
    DFGPU_HIP(hipStreamSynchronize(r.stream));  // `off` and `tasks` are locals the copies read
    return;
  }
  out_b = make_buf((size_t)m * 8);
  out_p = make_buf((size_t)m * 8);
  {
    ProfileScope pse("radix_join_emit", rec_bytes + m * 16);
    if (t.exact)
      k_rj_join<true, 1><<<grid, BLOCK, 0, r.stream>>>(t.build.key->as<uint64_t>(), t.build.rid->as<uint32_t>(), t.build.starts->as<uint64_t>(), ps.key->as<uint64_t>(),
                                                          ps.rid->as<uint32_t>(), d_tasks->as<RadixTask>(), nt, v, nullptr, d_off->as<uint64_t>(), out_b->as<int64_t>(), out_p->as<int64_t>());
    else
      k_rj_join<false, 1><<<grid, BLOCK, 0, r.stream>>>(t.build.key->as<uint64_t>(), t.build.rid->as<uint32_t>(), t.build.starts->as<uint64_t>(), ps.key->as<uint64_t>(),
                                                           ps.rid->as<uint32_t>(), d_tasks->as<RadixTask>(), nt, v, nullptr, d_off->as<uint64_t>(), out_b->as<int64_t>(), out_p->as<int64_t>());
    DFGPU_HIP(hipGetLastError());
  }
  DFGPU_HIP(hipStreamSynchronize(r.stream));  // `off` and `tasks` are locals the copies read
}

void radix_join_pairs(const RadixTable& t, const Table& build, const std::vector<int>& build_keys, const Table& probe, const std::vector<int>& probe_keys,
                      bool null_equals_null, bool force_collisions, BufPtr& out_b, BufPtr& out_p, int64_t& m) {
  radix_join_run(t, build, build_keys, probe, probe_keys, null_equals_null, force_collisions, out_b, out_p, m, nullptr);
}

// INNER join, output columns written by the emit walk (no pair list, no gathers).  false: this table / these columns do not qualify
// (record keys that are hashes, nullable or variable-width payload, too many columns) — the caller takes the pairs.
bool radix_join_inner_columns(const RadixTable& t, const Table& build, const std::vector<int>& build_keys, const Table& probe, const std::vector<int>& probe_keys,
                              const std::vector<int>& bout, const std::vector<int>& pout, Table& out) {
  if (!t.exact || force_fused_off() || (int)(bout.size() + pout.size()) > RJ_MAX_COLS || bout.size() + pout.size() == 0) return false;
  auto fixed = [](const Column& c) { return !c.has_nulls() && c.field.type != DFGPU_UTF8 && c.field.type != DFGPU_BOOL; };
  for (int c : bout)
    if (!fixed(build.cols[(size_t)c])) return false;
  for (int c : pout)
    if (!fixed(probe.cols[(size_t)c])) return false;
  if (probe.cols[(size_t)probe_keys[0]].has_nulls() || build.cols[(size_t)build_keys[0]].has_nulls()) return false;   // (NULL keys are dropped from the records: fine, but keep it simple)
  out = Table{};
  out.device = probe.device;
  std::function<RjCols(int64_t)> columns = [&](int64_t m) {
    RjCols rc{};
    out.nrows = m;
    for (int c : bout) {
      const Column& sc = build.cols[(size_t)c];
      out.cols.push_back(alloc_like(sc, m));
      rc.src[rc.n] = sc.ptr();
      rc.dst[rc.n] = out.cols.back().data->ptr;
      rc.width[rc.n] = type_width(sc.field.type);
      if (c == build_keys[0]) rc.key_cols |= 1u << rc.n;
      rc.n++;
    }
    rc.n_build = rc.n;
    for (int c = 0; c < rc.n_build; c++)
      if (!((rc.key_cols >> c) & 1u)) rc.need_brow = 1;
    for (int c : pout) {
      const Column& sc = probe.cols[(size_t)c];
      out.cols.push_back(alloc_like(sc, m));
      rc.src[rc.n] = sc.ptr();
      rc.dst[rc.n] = out.cols.back().data->ptr;
      rc.width[rc.n] = type_width(sc.field.type);
      if (c == probe_keys[0]) rc.key_cols |= 1u << rc.n;
      rc.n++;
    }
    return rc;
  };
  BufPtr ob, op;
  int64_t m = 0;
  radix_join_run(t, build, build_keys, probe, probe_keys, false, false, ob, op, m, &columns);
  if (m == 0) {   // no pair: the callback never ran
    out.nrows = 0;
    out.cols.clear();
    for (int c : bout) out.cols.push_back(alloc_like(build.cols[(size_t)c], 0));
    for (int c : pout) out.cols.push_back(alloc_like(probe.cols[(size_t)c], 0));
  }
  return true;
}

}  // namespace dfgpu
