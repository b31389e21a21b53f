// grouped.hip — rows moved into groups of their key (round 4): the pass under every operator that has to turn random accesses
// into cache-resident ones.
//
// Where it is used: a hash-join probe whose keys arrive in no order against a table beyond the caches (join.hip: the grouped
// probe with positional return — the probe KEYS travel to the table, group by group, and what they found comes back to the
// probe rows through `dest`), the rank-ordered copy of a shuffled build side's payload, and the key-only grouped probe.
// The reference has no counterpart: its probe walks one table with dependent random loads (joins/join_hash_map.rs:389-484),
// which on MI355X costs a 64-byte unit of HBM per row once the table outgrows the 4 MiB L2 of an XCD (profiles/r2_fetch_calib.md,
// profiles/r3_random_access.md: ~50 G random lines/s against 270 G/s for L2 hits).
//
// Shape of the pass (two kernels, both one workgroup per CHUNK of consecutive tiles):
//   k_gp_hist     keys -> group (LDS atomics) -> counts[group][chunk]
//   scan          exclusive prefix over the group-major count matrix: every (group, chunk) gets its output range
//   k_gp_scatter  per tile: keys -> group, rank inside (tile, group) by a returning LDS atomic, exclusive scan of the tile's
//                 counts, the tile staged in LDS in group order, every group's run written contiguously (TILE / groups rows
//                 per run: 128 bytes of keys at 8192 rows and 512 groups); `dest[row]` = where the row went; carried columns
//                 take the same route one after the other.  The chunk keeps its running output cursors in LDS, so the count
//                 matrix has one column per chunk (thousands), not per tile.
// Rows whose key is NULL, masked out, or outside the table's key range take no part (dest = ~0): they can match nothing.
// Order inside a group is arbitrary (the ranks come from atomics): callers that need the probe order get it back through `dest`.
#include <algorithm>
#include <cstdlib>

#include "device.hpp"
#include "internal.hpp"
#include "grouped.hpp"

namespace dfgpu {

constexpr int GP_RANK_BITS = 13;   // rank inside (tile, group) < 8192

__device__ __forceinline__ bool gp_row_takes_part(const KeyCol& key, const uint64_t* __restrict__ row_mask, int64_t i) {
  bool ok = true;
  if (key.valid) ok = (key.valid[i >> 6] >> (i & 63)) & 1ull;
  if (row_mask) ok = ok && ((row_mask[i >> 6] >> (i & 63)) & 1ull);
  return ok;
}

template <int KT, int THREADS, int ITEMS>
__global__ __launch_bounds__(THREADS) void k_gp_hist(KeyCol key, int64_t n, GroupSpec gs, int P, const uint64_t* __restrict__ row_mask, int tiles_per_chunk,
                                                     int64_t n_chunks, uint32_t* __restrict__ counts) {
  __shared__ unsigned s_cnt[GP_MAX_GROUPS];
  constexpr int TILE = THREADS * ITEMS;
  for (int64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
    for (int i = threadIdx.x; i < P; i += THREADS) s_cnt[i] = 0;
    __syncthreads();
    const int64_t lo = chunk * (int64_t)tiles_per_chunk * TILE;
    const int64_t hi = (lo + (int64_t)tiles_per_chunk * TILE) < n ? (lo + (int64_t)tiles_per_chunk * TILE) : n;
    for (int64_t base = lo; base < hi; base += TILE) {
      uint64_t k[ITEMS];
#pragma unroll
      for (int c = 0; c < ITEMS; c++) {  // all loads of the tile in flight together
        const int64_t i = base + (int64_t)c * THREADS + threadIdx.x;
        k[c] = load_key<KT>(key, i < hi ? i : hi - 1);
      }
#pragma unroll
      for (int c = 0; c < ITEMS; c++) {
        const int64_t i = base + (int64_t)c * THREADS + threadIdx.x;
        const uint64_t idx = k[c] - gs.offset;
        if (i < hi && idx < gs.size && gp_row_takes_part(key, row_mask, i)) atomicAdd(&s_cnt[__umul64hi(idx, gs.mul)], 1u);
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < P; i += THREADS) counts[(int64_t)i * n_chunks + chunk] = s_cnt[i];
    __syncthreads();
  }
}

// dynamic LDS: stage [TILE * stage_width] | cnt, goff, delta u32 [P each] | wave totals | start u16 [P] | group of every staged row
// u16 [TILE] (only when columns are carried: a key says its own group).  WPS = workgroups the launch bounds make room for per
// CU-quarter (waves per SIMD); PREFETCH = the next tile's keys are loaded while this one is written out.
template <int KT, int THREADS, int ITEMS, int WPS, bool PREFETCH>
__global__ __launch_bounds__(THREADS, WPS) void k_gp_scatter(KeyCol key, int64_t n, GroupSpec gs, int P, const uint64_t* __restrict__ row_mask, int tiles_per_chunk,
                                                             int64_t n_chunks, const uint64_t* __restrict__ offsets, uint64_t* __restrict__ out_keys,
                                                             uint32_t* __restrict__ dest, GroupCols cols, int stage_width, int narrow_keys) {
  extern __shared__ __align__(16) unsigned char gp_smem[];
  constexpr int TILE = THREADS * ITEMS;
  constexpr int NWAVE = THREADS / WAVE;
  constexpr int GPT = (GP_MAX_GROUPS + THREADS - 1) / THREADS;   // groups per thread in the tile's scan
  static_assert(TILE <= (1 << GP_RANK_BITS), "rank field too narrow");
  const bool carry = cols.n > 0;
  unsigned char* s_stage = gp_smem;
  unsigned* s_cnt = reinterpret_cast<unsigned*>(gp_smem + (size_t)TILE * stage_width);
  unsigned* s_goff = s_cnt + P;
  unsigned* s_delta = s_goff + P;     // goff - start of the current tile: where a staged position goes
  unsigned* s_wtot = s_delta + P;
  uint16_t* s_start = reinterpret_cast<uint16_t*>(s_wtot + NWAVE);
  uint16_t* s_g = s_start + P;
  const unsigned lane = lane_id();
  const int wave = threadIdx.x >> 6;
  for (int64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
    const int64_t lo = chunk * (int64_t)tiles_per_chunk * TILE;
    const int64_t hi = (lo + (int64_t)tiles_per_chunk * TILE) < n ? (lo + (int64_t)tiles_per_chunk * TILE) : n;
    for (int i = threadIdx.x; i < P; i += THREADS) {
      s_goff[i] = (unsigned)offsets[(int64_t)i * n_chunks + chunk];
      s_cnt[i] = 0;
    }
    __syncthreads();
    uint64_t k[ITEMS], knext[PREFETCH ? ITEMS : 1];
    if (PREFETCH) {
#pragma unroll
      for (int c = 0; c < ITEMS; c++) {
        const int64_t i = lo + (int64_t)c * THREADS + threadIdx.x;
        knext[PREFETCH ? c : 0] = lo < hi ? load_key<KT>(key, i < hi ? i : hi - 1) : 0ull;
      }
    }
    for (int64_t base = lo; base < hi; base += TILE) {
      unsigned gr[ITEMS];
      if (PREFETCH) {
#pragma unroll
        for (int c = 0; c < ITEMS; c++) k[c] = knext[PREFETCH ? c : 0];
      } else {
#pragma unroll
        for (int c = 0; c < ITEMS; c++) {
          const int64_t i = base + (int64_t)c * THREADS + threadIdx.x;
          k[c] = load_key<KT>(key, i < hi ? i : hi - 1);
        }
      }
      // ---- group and rank of every row of the tile
#pragma unroll
      for (int c = 0; c < ITEMS; c++) {
        const int64_t i = base + (int64_t)c * THREADS + threadIdx.x;
        const uint64_t idx = k[c] - gs.offset;
        gr[c] = 0xFFFFFFFFu;
        if (i < hi && idx < gs.size && gp_row_takes_part(key, row_mask, i)) {
          const unsigned g = (unsigned)__umul64hi(idx, gs.mul);
          gr[c] = (g << GP_RANK_BITS) | atomicAdd(&s_cnt[g], 1u);
        }
      }
      __syncthreads();
      // ---- exclusive scan of the tile's group counts: thread t owns groups [t * GPT, (t + 1) * GPT)
      {
        unsigned c_[GPT], v = 0;
#pragma unroll
        for (int q = 0; q < GPT; q++) {
          const int g = (int)threadIdx.x * GPT + q;
          c_[q] = g < P ? s_cnt[g] : 0u;
          v += c_[q];
        }
        const unsigned inc = wave_inclusive_sum<unsigned>(v);
        if (lane == 63) s_wtot[wave] = inc;
        __syncthreads();
        unsigned run = inc - v;
#pragma unroll
        for (int w = 0; w < NWAVE; w++) run += w < wave ? s_wtot[w] : 0u;
#pragma unroll
        for (int q = 0; q < GPT; q++) {
          const int g = (int)threadIdx.x * GPT + q;
          if (g < P) {
            s_start[g] = (uint16_t)run;
            s_delta[g] = s_goff[g] - run;
          }
          run += c_[q];
        }
      }
      __syncthreads();
      unsigned total = 0;
#pragma unroll
      for (int w = 0; w < NWAVE; w++) total += s_wtot[w];
      // ---- the tile in group order: keys into LDS, every row told where it goes
      unsigned pos[ITEMS];
#pragma unroll
      for (int c = 0; c < ITEMS; c++) {
        const int64_t i = base + (int64_t)c * THREADS + threadIdx.x;
        pos[c] = 0xFFFFFFFFu;
        if (gr[c] != 0xFFFFFFFFu) {
          const unsigned g = gr[c] >> GP_RANK_BITS, r = gr[c] & ((1u << GP_RANK_BITS) - 1u);
          pos[c] = (unsigned)s_start[g] + r;
          reinterpret_cast<uint64_t*>(s_stage)[pos[c]] = k[c];
          if (carry && !out_keys) s_g[pos[c]] = (uint16_t)g;   // (a staged key says its own group)
          if (dest) dest[i] = pos[c] + s_delta[g];
        } else if (dest && i < hi) {
          dest[i] = 0xFFFFFFFFu;
        }
      }
      // the next tile's keys are on their way while this one is written out
      if (PREFETCH) {
        const int64_t nb = base + TILE;
#pragma unroll
        for (int c = 0; c < ITEMS; c++) {
          const int64_t i = nb + (int64_t)c * THREADS + threadIdx.x;
          knext[PREFETCH ? c : 0] = nb < hi ? load_key<KT>(key, i < hi ? i : hi - 1) : 0ull;
        }
      }
      __syncthreads();
      // where every staged position goes — worked out ONCE per tile and kept in registers for the carried columns (round 6: every column's
      // write-out used to look its rows' groups and the groups' deltas up again: two LDS reads per row and column of a pass that is bound
      // by its LDS work).  A staged key says its own group; without staged keys the group comes from s_g.
      unsigned dq[ITEMS];
#pragma unroll
      for (int j = 0; j < ITEMS; j++) {
        const unsigned q = threadIdx.x + (unsigned)j * THREADS;
        dq[j] = 0xFFFFFFFFu;
        if (q < total) {
          if (out_keys) {
            const uint64_t kq = reinterpret_cast<const uint64_t*>(s_stage)[q];
            const unsigned d = q + s_delta[(unsigned)__umul64hi(kq - gs.offset, gs.mul)];
            dq[j] = d;
            // (narrow_keys: key - offset as 32 bits — a range below 2^32 moves half the key bytes)
            if (narrow_keys) reinterpret_cast<uint32_t*>(out_keys)[d] = (uint32_t)(kq - gs.offset);
            else out_keys[d] = kq;
          } else if (carry) {
            dq[j] = q + s_delta[s_g[q]];
          }
        }
      }
      // ---- carried columns: the same route, one after the other through the same staging buffer (the width is asked once per column,
      // outside the unrolled loops: the tile's loads leave back to back)
      for (int cc = 0; cc < cols.n; cc++) {
        __syncthreads();
        const int w = cols.width[cc];
        const void* csrc = cols.src[cc];
        void* cdst = cols.dst[cc];
        if (w == 16) {
#pragma unroll
          for (int c = 0; c < ITEMS; c++) {
            const int64_t i = base + (int64_t)c * THREADS + threadIdx.x;
            if (pos[c] != 0xFFFFFFFFu) reinterpret_cast<uint4*>(s_stage)[pos[c]] = reinterpret_cast<const uint4*>(csrc)[i];
          }
        } else if (w == 8) {
#pragma unroll
          for (int c = 0; c < ITEMS; c++) {
            const int64_t i = base + (int64_t)c * THREADS + threadIdx.x;
            if (pos[c] != 0xFFFFFFFFu) reinterpret_cast<uint64_t*>(s_stage)[pos[c]] = reinterpret_cast<const uint64_t*>(csrc)[i];
          }
        } else if (w == 4) {
          uint32_t v[ITEMS];
#pragma unroll
          for (int c = 0; c < ITEMS; c++) {
            const int64_t i = base + (int64_t)c * THREADS + threadIdx.x;
            if (pos[c] != 0xFFFFFFFFu) v[c] = csrc ? reinterpret_cast<const uint32_t*>(csrc)[i] : (uint32_t)i;   // (no source: the row's number)
          }
#pragma unroll
          for (int c = 0; c < ITEMS; c++)
            if (pos[c] != 0xFFFFFFFFu) reinterpret_cast<uint32_t*>(s_stage)[pos[c]] = v[c];
        } else {
          uint8_t v[ITEMS];
#pragma unroll
          for (int c = 0; c < ITEMS; c++) {
            const int64_t i = base + (int64_t)c * THREADS + threadIdx.x;
            if (pos[c] != 0xFFFFFFFFu) v[c] = reinterpret_cast<const uint8_t*>(csrc)[i];
          }
#pragma unroll
          for (int c = 0; c < ITEMS; c++)
            if (pos[c] != 0xFFFFFFFFu) s_stage[pos[c]] = v[c];
        }
        __syncthreads();
        if (w == 16) {
#pragma unroll
          for (int j = 0; j < ITEMS; j++)
            if (dq[j] != 0xFFFFFFFFu) reinterpret_cast<uint4*>(cdst)[dq[j]] = reinterpret_cast<const uint4*>(s_stage)[threadIdx.x + (unsigned)j * THREADS];
        } else if (w == 8) {
#pragma unroll
          for (int j = 0; j < ITEMS; j++)
            if (dq[j] != 0xFFFFFFFFu) reinterpret_cast<uint64_t*>(cdst)[dq[j]] = reinterpret_cast<const uint64_t*>(s_stage)[threadIdx.x + (unsigned)j * THREADS];
        } else if (w == 4) {
#pragma unroll
          for (int j = 0; j < ITEMS; j++)
            if (dq[j] != 0xFFFFFFFFu) reinterpret_cast<uint32_t*>(cdst)[dq[j]] = reinterpret_cast<const uint32_t*>(s_stage)[threadIdx.x + (unsigned)j * THREADS];
        } else {
#pragma unroll
          for (int j = 0; j < ITEMS; j++)
            if (dq[j] != 0xFFFFFFFFu) reinterpret_cast<uint8_t*>(cdst)[dq[j]] = s_stage[threadIdx.x + (unsigned)j * THREADS];
        }
      }
      __syncthreads();
      for (int i = threadIdx.x; i < P; i += THREADS) {
        s_goff[i] += s_cnt[i];
        s_cnt[i] = 0;
      }
      __syncthreads();
    }
  }
}

// ---- the record form (round 6): 32-bit keys (key - offset) and NC carried 4-byte columns leave as ONE (1 + NC)-dword record per row.
// What bounds k_gp_scatter at 2048 groups is its stores — runs of 4 rows are 16-byte pieces of a line per column (the pass without its
// global stores: 0.84 of 2.25 ms for 150 M orders) — and a row's columns side by side make one piece of three: 48 bytes per run, a third
// of the store instructions.  All planes of a tile are staged at once (the carried columns' loads leave together, one barrier pair per
// tile instead of one per column).  dynamic LDS: planes u32 [(1 + NC) * TILE] | cnt, goff, delta u32 [P each] | wave totals | start u16 [P]
template <int KT, int THREADS, int ITEMS, int NC>
__global__ __launch_bounds__(THREADS, 4) void k_gp_scatter_rec(KeyCol key, int64_t n, GroupSpec gs, int P, const uint64_t* __restrict__ row_mask, int tiles_per_chunk,
                                                               int64_t n_chunks, const uint64_t* __restrict__ offsets, uint32_t* __restrict__ out_rec, GroupCols cols) {
  extern __shared__ __align__(16) unsigned char gp_smem[];
  constexpr int TILE = THREADS * ITEMS;
  constexpr int NWAVE = THREADS / WAVE;
  constexpr int GPT = (GP_MAX_GROUPS + THREADS - 1) / THREADS;
  constexpr int RW = 1 + NC;
  static_assert(TILE <= (1 << GP_RANK_BITS), "rank field too narrow");
  uint32_t* s_plane = reinterpret_cast<uint32_t*>(gp_smem);                       // [RW][TILE]
  unsigned* s_cnt = reinterpret_cast<unsigned*>(gp_smem + (size_t)RW * TILE * 4);
  unsigned* s_goff = s_cnt + P;
  unsigned* s_delta = s_goff + P;
  unsigned* s_wtot = s_delta + P;
  uint16_t* s_start = reinterpret_cast<uint16_t*>(s_wtot + NWAVE);
  const unsigned lane = lane_id();
  const int wave = threadIdx.x >> 6;
  for (int64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
    const int64_t lo = chunk * (int64_t)tiles_per_chunk * TILE;
    const int64_t hi = (lo + (int64_t)tiles_per_chunk * TILE) < n ? (lo + (int64_t)tiles_per_chunk * TILE) : n;
    for (int i = threadIdx.x; i < P; i += THREADS) {
      s_goff[i] = (unsigned)offsets[(int64_t)i * n_chunks + chunk];
      s_cnt[i] = 0;
    }
    __syncthreads();
    uint64_t k[ITEMS], knext[ITEMS];
#pragma unroll
    for (int c = 0; c < ITEMS; c++) {
      const int64_t i = lo + (int64_t)c * THREADS + threadIdx.x;
      knext[c] = lo < hi ? load_key<KT>(key, i < hi ? i : hi - 1) : 0ull;
    }
    for (int64_t base = lo; base < hi; base += TILE) {
      unsigned gr[ITEMS];
#pragma unroll
      for (int c = 0; c < ITEMS; c++) k[c] = knext[c];
      // ---- group and rank of every row of the tile
#pragma unroll
      for (int c = 0; c < ITEMS; c++) {
        const int64_t i = base + (int64_t)c * THREADS + threadIdx.x;
        const uint64_t idx = k[c] - gs.offset;
        gr[c] = 0xFFFFFFFFu;
        if (i < hi && idx < gs.size && gp_row_takes_part(key, row_mask, i)) {
          const unsigned g = (unsigned)__umul64hi(idx, gs.mul);
          gr[c] = (g << GP_RANK_BITS) | atomicAdd(&s_cnt[g], 1u);
        }
      }
      // the carried columns' values of the tile: all of them on their way before the scan's barriers
      uint32_t cv[NC > 0 ? NC : 1][ITEMS];
#pragma unroll
      for (int cc = 0; cc < NC; cc++) {
        const uint32_t* csrc = reinterpret_cast<const uint32_t*>(cols.src[cc]);
        const int64_t sdw = cols.sdw[cc], soff = cols.soff[cc];   // (an 8-byte column: two planes, words 0 and 1 of elements two words apart)
#pragma unroll
        for (int c = 0; c < ITEMS; c++) {
          const int64_t i = base + (int64_t)c * THREADS + threadIdx.x;
          cv[cc][c] = gr[c] == 0xFFFFFFFFu ? 0u : (csrc ? csrc[i * sdw + soff] : (uint32_t)i);   // (no source: the row's number)
        }
      }
      __syncthreads();
      // ---- exclusive scan of the tile's group counts: thread t owns groups [t * GPT, (t + 1) * GPT)
      {
        unsigned c_[GPT], v = 0;
#pragma unroll
        for (int q = 0; q < GPT; q++) {
          const int g = (int)threadIdx.x * GPT + q;
          c_[q] = g < P ? s_cnt[g] : 0u;
          v += c_[q];
        }
        const unsigned inc = wave_inclusive_sum<unsigned>(v);
        if (lane == 63) s_wtot[wave] = inc;
        __syncthreads();
        unsigned run = inc - v;
#pragma unroll
        for (int w = 0; w < NWAVE; w++) run += w < wave ? s_wtot[w] : 0u;
#pragma unroll
        for (int q = 0; q < GPT; q++) {
          const int g = (int)threadIdx.x * GPT + q;
          if (g < P) {
            s_start[g] = (uint16_t)run;
            s_delta[g] = s_goff[g] - run;
          }
          run += c_[q];
        }
      }
      __syncthreads();
      unsigned total = 0;
#pragma unroll
      for (int w = 0; w < NWAVE; w++) total += s_wtot[w];
      // ---- the tile in group order, plane by plane
#pragma unroll
      for (int c = 0; c < ITEMS; c++) {
        if (gr[c] == 0xFFFFFFFFu) continue;
        const unsigned g = gr[c] >> GP_RANK_BITS, r = gr[c] & ((1u << GP_RANK_BITS) - 1u);
        const unsigned ps = (unsigned)s_start[g] + r;
        s_plane[ps] = (uint32_t)(k[c] - gs.offset);
#pragma unroll
        for (int cc = 0; cc < NC; cc++) s_plane[(cc + 1) * TILE + ps] = cv[cc][c];
      }
      // the next tile's keys are on their way while this one is written out
      {
        const int64_t nb = base + TILE;
#pragma unroll
        for (int c = 0; c < ITEMS; c++) {
          const int64_t i = nb + (int64_t)c * THREADS + threadIdx.x;
          knext[c] = nb < hi ? load_key<KT>(key, i < hi ? i : hi - 1) : 0ull;
        }
      }
      __syncthreads();
      struct __attribute__((packed, aligned(4))) Rec { uint32_t w[RW]; };
#pragma unroll
      for (int j = 0; j < ITEMS; j++) {
        const unsigned q = threadIdx.x + (unsigned)j * THREADS;
        if (q < total) {
          Rec rec;
          rec.w[0] = s_plane[q];
#pragma unroll
          for (int cc = 0; cc < NC; cc++) rec.w[cc + 1] = s_plane[(cc + 1) * TILE + q];
          const unsigned d = q + s_delta[(unsigned)__umul64hi((uint64_t)rec.w[0], gs.mul)];
          reinterpret_cast<Rec*>(out_rec)[d] = rec;
        }
      }
      __syncthreads();
      for (int i = threadIdx.x; i < P; i += THREADS) {
        s_goff[i] += s_cnt[i];
        s_cnt[i] = 0;
      }
      __syncthreads();
    }
  }
}

__global__ void k_gp_bounds(const uint64_t* __restrict__ offsets, int P, int64_t n_chunks, uint64_t* __restrict__ bounds) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g <= P) bounds[g] = offsets[(int64_t)g * n_chunks];
}

template <typename F>
static void gp_with_key_type(int dfgpu_type, F&& f) {
  switch (dfgpu_type) {
    case DFGPU_INT32: case DFGPU_DATE32: f(std::integral_constant<int, KT_I32>{}); break;
    case DFGPU_UINT32: f(std::integral_constant<int, KT_U32>{}); break;
    case DFGPU_INT64: case DFGPU_UINT64: f(std::integral_constant<int, KT_I64>{}); break;
    case DFGPU_UINT8: f(std::integral_constant<int, KT_U8>{}); break;
    default: throw Error("group_rows_by_key: the key column is not an integer column");
  }
}

GroupedRows group_rows_by_key(const KeyCol& key, int64_t n, const GroupSpec& gs, int nbits, const uint64_t* row_mask, bool want_keys, bool want_dest,
                              const std::vector<const void*>& carry_src, const std::vector<int>& carry_width, const char* what, bool narrow_keys, bool records) {
  Runtime& r = rt();
  DFGPU_CHECK(n > 0 && n < 0xFFFFFFFFll, "group_rows_by_key: row count out of range");
  DFGPU_CHECK(nbits >= 1 && (1 << nbits) <= GP_MAX_GROUPS, "group_rows_by_key: bad number of groups");
  DFGPU_CHECK(carry_src.size() == carry_width.size() && (int)carry_src.size() <= GP_MAX_COLS, "group_rows_by_key: bad carried columns");
  DFGPU_CHECK(gs.size > 0 && (uint64_t)(((unsigned __int128)(gs.size - 1) * gs.mul) >> 64) < (1ull << nbits), "group_rows_by_key: the key range does not fit the groups");
  const int P = 1 << nbits;
  GroupedRows out;
  out.P = P;
  int stage_width = 8;
  for (int w : carry_width) {
    DFGPU_CHECK(w == 1 || w == 4 || w == 8 || w == 16, "group_rows_by_key: carried columns are 1, 4, 8 or 16 bytes wide");
    stage_width = std::max(stage_width, w);
  }
  // Tile shape: 1024 threads x 8 rows = 8192-row tiles (runs of 8192 / P rows), the next tile's keys prefetched, one workgroup per CU.
  // Measured against it (profiles/r4_group_rows.md) and dropped: the same tile without the prefetch at two workgroups per CU (+ 5 %),
  // 512 threads x 8 rows at three per CU (half the run length: + 13 %) — the pass is bound by its LDS work, not by latency.
  // Round 6, also dropped: 16384-row tiles with 32-bit key registers (runs of twice the length): 16 rows per thread spill (148-620 bytes of
  // scratch per lane at the 128 VGPRs four waves per SIMD leave) — 2.55 against 2.25 ms for 150 M orders into 1831 windows.
  constexpr int THREADS = 1024;
  constexpr int ITEMS = 8;
  const int TILE = THREADS * ITEMS;
  const int64_t n_tiles = (n + TILE - 1) / TILE;
  // chunks: enough of them to fill the chip a few times over, few enough to keep the count matrix small
  int tiles_per_chunk = (int)std::min<int64_t>(32, std::max<int64_t>(1, n_tiles / 2048));
  const int64_t n_chunks = (n_tiles + tiles_per_chunk - 1) / tiles_per_chunk;
  BufPtr counts = make_buf((size_t)P * n_chunks * 4);
  BufPtr offsets = make_buf(((size_t)P * n_chunks + 1) * 8);
  const int grid = (int)std::min<int64_t>(n_chunks, (int64_t)r.num_cus * 6);
  const int64_t key_bytes = n * key.width;
  {
    ProfileScope ps("group_rows_count", key_bytes);
    gp_with_key_type(key.type, [&](auto kt) {
      k_gp_hist<decltype(kt)::value, THREADS, ITEMS><<<grid, THREADS, 0, r.stream>>>(key, n, gs, P, row_mask, tiles_per_chunk, n_chunks, counts->as<uint32_t>());
    });
    DFGPU_HIP(hipGetLastError());
  }
  scan_u32(counts->as<uint32_t>(), (int64_t)P * n_chunks, offsets->as<uint64_t>());
  out.bounds = make_buf((size_t)(P + 1) * 8);
  k_gp_bounds<<<(P + 1 + 255) / 256, 256, 0, r.stream>>>(offsets->as<uint64_t>(), P, n_chunks, out.bounds->as<uint64_t>());
  out.rows = (int64_t)read_u64(offsets->as<uint64_t>() + (int64_t)P * n_chunks);
  const size_t out_rows = (size_t)std::max<int64_t>(out.rows, 1);
  DFGPU_CHECK(!narrow_keys || gs.size <= (1ull << 32), "group_rows_by_key: 32-bit keys need a key range below 2^32");
  out.key_width = narrow_keys ? 4 : 8;
  // the record form: 32-bit keys and one to three carried 4-byte columns side by side (k_gp_scatter_rec)
  int rec_words = 1;
  bool rec_ok = records && narrow_keys && want_keys && !want_dest && !carry_src.empty() && option_on("group.records", true);
  for (int w : carry_width) {
    rec_ok &= w == 4 || w == 8;
    rec_words += w / 4;
  }
  rec_ok &= rec_words <= 4;
  if (rec_ok) {
    const int NC = rec_words - 1;
    out.rec_dwords = rec_words;
    out.records = make_buf(out_rows * (size_t)out.rec_dwords * 4 + 16);
    GroupCols gc{};
    gc.n = NC;
    int64_t moved = out.rows * out.rec_dwords * 4;
    int plane = 0;
    for (size_t c = 0; c < carry_src.size(); c++) {
      DFGPU_CHECK(carry_src[c] || carry_width[c] == 4, "group_rows_by_key: a carried column without a source is the 32-bit row number");
      out.rec_off.push_back(1 + plane);
      for (int h = 0; h < carry_width[c] / 4; h++, plane++) {
        gc.src[plane] = carry_src[c];
        gc.width[plane] = 4;
        gc.sdw[plane] = carry_width[c] / 4;
        gc.soff[plane] = h;
      }
      moved += carry_src[c] ? n * carry_width[c] : 0;
    }
    const size_t lds = (size_t)TILE * 4 * (size_t)out.rec_dwords + (size_t)P * 12 + (size_t)(THREADS / WAVE) * 4 + (size_t)P * 2;
    ProfileScope ps(what ? what : "group_rows_scatter", key_bytes + moved);
    gp_with_key_type(key.type, [&](auto kt) {
      constexpr int T = decltype(kt)::value;
      auto launch = [&](auto kern) {
        DFGPU_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        kern<<<grid, THREADS, lds, r.stream>>>(key, n, gs, P, row_mask, tiles_per_chunk, n_chunks, offsets->as<uint64_t>(), out.records->as<uint32_t>(), gc);
      };
      if (NC == 1) launch(k_gp_scatter_rec<T, THREADS, ITEMS, 1>);
      else if (NC == 2) launch(k_gp_scatter_rec<T, THREADS, ITEMS, 2>);
      else launch(k_gp_scatter_rec<T, THREADS, ITEMS, 3>);   // (16-byte records: a whole line per 4-row run; 156 KB of the CU's 160 KB of LDS)
    });
    DFGPU_HIP(hipGetLastError());
    return out;
  }
  if (want_keys) out.keys = make_buf(out_rows * (size_t)out.key_width + 8);
  if (want_dest) out.dest = make_buf((size_t)n * 4);
  GroupCols gc{};
  gc.n = (int)carry_src.size();
  int64_t moved = (want_keys ? out.rows * out.key_width : 0) + (want_dest ? n * 4 : 0);
  for (int c = 0; c < gc.n; c++) {
    out.cols.push_back(make_buf(out_rows * (size_t)carry_width[c]));
    gc.src[c] = carry_src[c];
    gc.dst[c] = out.cols.back()->ptr;
    gc.width[c] = carry_width[c];
    DFGPU_CHECK(carry_src[c] || carry_width[c] == 4, "group_rows_by_key: a carried column without a source is the 32-bit row number");
    moved += (carry_src[c] ? n * carry_width[c] : 0) + out.rows * carry_width[c];
  }
  const size_t lds = (size_t)TILE * stage_width + (size_t)P * 12 + (size_t)(THREADS / WAVE) * 4 + (size_t)P * 2 + (gc.n > 0 ? (size_t)TILE * 2 : 0);
  {
    ProfileScope ps(what ? what : "group_rows_scatter", key_bytes + moved);
    gp_with_key_type(key.type, [&](auto kt) {
      constexpr int T = decltype(kt)::value;
      auto launch = [&](auto kern, int threads) {
        DFGPU_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        kern<<<grid, threads, lds, r.stream>>>(key, n, gs, P, row_mask, tiles_per_chunk, n_chunks, offsets->as<uint64_t>(), want_keys ? out.keys->as<uint64_t>() : nullptr,
                                               want_dest ? out.dest->as<uint32_t>() : nullptr, gc, stage_width, narrow_keys ? 1 : 0);
      };
      launch(k_gp_scatter<T, THREADS, ITEMS, 4, true>, THREADS);
    });
    DFGPU_HIP(hipGetLastError());
  }
  return out;
}

}  // namespace dfgpu
