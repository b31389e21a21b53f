// device.hpp — CDNA4 (gfx950, wave64) device-side helpers shared by all kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dfgpu {

typedef __int128 i128;
typedef unsigned __int128 u128;

constexpr int WAVE = 64;
constexpr int BLOCK = 256;  // 4 waves: one per SIMD

__device__ __forceinline__ unsigned lane_id() { return threadIdx.x & 63u; }

// Streamed-once data: non-temporal loads and stores (`global_load / global_store ... nt`) keep it out of the caches'
// replacement order.  What was measured (profiles/r2_streaming_nt.md): they pay where a wave writes whole lines that nobody
// else touches (the fused probe's all-match tiles: 9.93 -> 9.69 ms per SF100 step; the partition scatter's runs: 11.0 ->
// 8.7 ms) and cost where neighbouring waves share output lines (FilterExec's compaction +10-15 %).  DFGPU_STREAM_NT: bit 0
// loads, bit 1 stores (A/B builds).
#ifndef DFGPU_STREAM_NT
#define DFGPU_STREAM_NT 3
#endif
template <typename T>
__device__ __forceinline__ T stream_load(const T* p) {
#if DFGPU_STREAM_NT & 1
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}
template <typename T>
__device__ __forceinline__ void stream_store(T* p, T v) {
#if DFGPU_STREAM_NT & 2
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}
typedef unsigned stream_v4 __attribute__((ext_vector_type(4)));
template <>
__device__ __forceinline__ uint4 stream_load<uint4>(const uint4* p) {
#if DFGPU_STREAM_NT & 1
  const stream_v4 t = __builtin_nontemporal_load(reinterpret_cast<const stream_v4*>(p));
  return make_uint4(t.x, t.y, t.z, t.w);
#else
  return *p;
#endif
}
template <>
__device__ __forceinline__ void stream_store<uint4>(uint4* p, uint4 v) {
#if DFGPU_STREAM_NT & 2
  const stream_v4 t = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(t, reinterpret_cast<stream_v4*>(p));
#else
  *p = v;
#endif
}

// number of set bits of `mask` strictly below this lane (v_mbcnt_lo/hi)
__device__ __forceinline__ unsigned mbcnt(uint64_t mask) {
  return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}
__device__ __forceinline__ uint64_t ballot64(bool p) { return __ballot(p); }

// inclusive wave scan (64 lanes) via DPP-lowered shuffles
template <typename T>
__device__ __forceinline__ T wave_inclusive_sum(T v) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    T o = __shfl_up(v, d, 64);
    if ((int)lane_id() >= d) v += o;
  }
  return v;
}
// ---- DPP (data-parallel primitive) lane movement: VALU-only, no LDS crossbar traffic (ds_bpermute)
// ctrl: 0x110+n = row_shr:n (within a row of 16 lanes), 0x142 = row_bcast15, 0x143 = row_bcast31
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint64_t dpp_move_u64(uint64_t v) {  // lanes without a source (or in a masked-off row) read 0
  unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)v, CTRL, ROW_MASK, 0xF, true);
  unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), CTRL, ROW_MASK, 0xF, true);
  return ((uint64_t)hi << 32) | lo;
}
// inclusive wave prefix sum (mod 2^64) in 6 DPP steps: Hillis-Steele inside each 16-lane row, then
// row 0's total into row 1 and row 2's into row 3 (row_bcast15), then lanes 0-31's total into rows 2-3
__device__ __forceinline__ uint64_t wave_inclusive_sum_dpp(uint64_t v) {
  v += dpp_move_u64<0x111, 0xF>(v);
  v += dpp_move_u64<0x112, 0xF>(v);
  v += dpp_move_u64<0x114, 0xF>(v);
  v += dpp_move_u64<0x118, 0xF>(v);
  v += dpp_move_u64<0x142, 0xA>(v);
  v += dpp_move_u64<0x143, 0xC>(v);
  return v;
}

// inclusive segmented OR over adjacent lanes (`head` = first lane of a run): the run's last lane ends up with the
// OR of the whole run.  Lets a wave merge the bits it sets in one bitmap word into ONE atomicOr.  Every lane of
// the wave must be executing.
__device__ __forceinline__ uint64_t wave_seg_or(uint64_t v, bool head) {
  bool f = head;
  const unsigned lane = lane_id();
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint64_t o = __shfl_up(v, d, 64);
    const int fo = __shfl_up((int)f, d, 64);
    if (lane >= (unsigned)d && !f) {
      v |= o;
      f = fo != 0;
    }
  }
  return v;
}

// Workgroup b runs on XCD b % 8 (MI355X_MICROARCH.md): tile = xcd_tile(b + k * gridDim.x, n_tiles) (gridDim.x a multiple of 8) gives every
// XCD a CONTIGUOUS eighth of the tiles, walked in order — the lines neighbouring tiles read or append to meet in ONE L2 instead of
// being split across eight (a partially written line leaves each L2 as a masked write).  A bijection of [0, n_tiles); speed only.
__device__ __forceinline__ int64_t xcd_tile(int64_t b, int64_t n_tiles) {
  const int64_t q = n_tiles >> 3, r = n_tiles & 7, x = b & 7, j = b >> 3;
  return x * q + (x < r ? x : r) + j;
}

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

// ----------------------------------------------------------------------------- hashing
// Same mixer as oracle/dforacle.c (murmur3 fmix64).  The reference hashes with foldhash
// (common/src/hash_utils.rs:27,41) whose values are unpinned by any reference test.
__host__ __device__ __forceinline__ uint64_t fmix64(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
  x ^= x >> 33;
  return x;
}
__host__ __device__ __forceinline__ uint64_t hash_u64(uint64_t v, uint64_t seed) {
  return fmix64(v ^ seed ^ 0x9E3779B97F4A7C15ULL);
}
constexpr uint64_t SEED_JOIN = 0xA98409FE2C1A0E6CULL;         // HASH_JOIN_SEED analogue (hash_join/exec.rs:105)
constexpr uint64_t SEED_AGG = 0x51D7348D9B2F63A5ULL;          // AGGREGATION_HASH_SEED analogue (aggregates/mod.rs:236)
constexpr uint64_t SEED_REPARTITION = 0ULL;                   // REPARTITION_RANDOM_STATE (repartition/mod.rs:650)

// ------------------------------------------------------------------------- key columns
// A key column as seen by hashing / equality kernels.
struct KeyCol {
  const void* data;
  const uint64_t* valid;  // may be null
  int32_t type;           // dfgpu_type
  int32_t width;          // bytes
};
constexpr int MAX_KEYS = 8;
struct KeySet {
  KeyCol c[MAX_KEYS];
  int n;
};

__device__ __forceinline__ bool bit_at(const uint64_t* words, int64_t i) { return (words[i >> 6] >> (i & 63)) & 1ull; }

// value widened to (lo, hi): signed ints sign-extended, floats by bit pattern with
// -0.0 -> +0.0 (hash_utils.rs:258-276)
__device__ __forceinline__ void load_words(const KeyCol& k, int64_t i, uint64_t& lo, uint64_t& hi) {
  hi = 0;
  switch (k.type) {
    case 1: case 8: lo = (uint64_t)(int64_t)((const int32_t*)k.data)[i]; break;  // INT32 / DATE32
    case 6: lo = ((const uint32_t*)k.data)[i]; break;
    case 2: case 7: lo = ((const uint64_t*)k.data)[i]; break;
    case 4: { uint64_t b = ((const uint64_t*)k.data)[i]; lo = (b << 1) == 0 ? 0 : b; break; }
    case 5: lo = ((const uint8_t*)k.data)[i]; break;
    case 3: { const uint64_t* p = (const uint64_t*)k.data + 2 * i; lo = p[0]; hi = p[1]; break; }
    default: lo = 0;
  }
}
__device__ __forceinline__ uint64_t hash_value(const KeyCol& k, int64_t i, uint64_t seed) {
  uint64_t lo, hi;
  load_words(k, i, lo, hi);
  uint64_t h = hash_u64(lo, seed);
  if (k.type == 3) h = fmix64(hi ^ h);
  return h;
}
// ---- typed key access.  The generic load_words() switches on the column type per element; the
// compiler then cannot hoist loads out of the switch arms and waits on every single one
// (s_waitcnt vmcnt(0) after each global_load in the ISA).  Hot kernels are therefore instantiated
// per key type so that N independent loads are issued back-to-back.
enum KeyT : int { KT_I32 = 0, KT_U32 = 1, KT_I64 = 2, KT_U8 = 3, KT_ANY = 4 };
// NT: a non-temporal load (global_load ... nt) — the key column streams by once; read that way it does not push a table the same
// kernel looks up out of the L2 (scripts/microbench/stream_width.hip: +9 % on the read side alone, +17 % with a 5.6 MB bitmap beside it)
template <int KT, bool NT = false>
__device__ __forceinline__ uint64_t load_key(const KeyCol& k, int64_t i) {
  if (NT) {
    if (KT == KT_I32) return (uint64_t)(int64_t)__builtin_nontemporal_load((const int32_t*)k.data + i);
    if (KT == KT_U32) return __builtin_nontemporal_load((const uint32_t*)k.data + i);
    if (KT == KT_I64) return __builtin_nontemporal_load((const uint64_t*)k.data + i);
    if (KT == KT_U8) return __builtin_nontemporal_load((const uint8_t*)k.data + i);
  }
  if (KT == KT_I32) return (uint64_t)(int64_t)((const int32_t*)k.data)[i];
  if (KT == KT_U32) return ((const uint32_t*)k.data)[i];
  if (KT == KT_I64) return ((const uint64_t*)k.data)[i];
  if (KT == KT_U8) return ((const uint8_t*)k.data)[i];
  uint64_t lo, hi;
  load_words(k, i, lo, hi);
  return lo;
}
// create_hashes semantics (hash_utils.rs:1239-1252): col 0 seeds with `seed`, col i>=1
// re-seeds with the running hash; NULL leaves the running hash (initially 0) untouched.
__device__ __forceinline__ uint64_t hash_row(const KeySet& ks, int64_t i, uint64_t seed, bool& any_null) {
  uint64_t h = 0;
  any_null = false;
  for (int c = 0; c < ks.n; c++) {
    if (ks.c[c].valid && !bit_at(ks.c[c].valid, i)) { any_null = true; continue; }
    h = hash_value(ks.c[c], i, c == 0 ? seed : h);
  }
  return h;
}
// equal_rows_arr (joins/utils.rs:2191-2260).  `float_total_order`: the join compares Float64 keys with arrow-ord's eq — IEEE 754
// totalOrder equality, i.e. equality of the bit patterns: -0.0 and +0.0 are different keys (they still hash alike) —
// which is what the oracle restates (dforacle.c keys_equal)
__device__ __forceinline__ bool keys_equal(const KeySet& a, int64_t ia, const KeySet& b, int64_t ib, bool null_equals_null, bool float_total_order = false) {
  for (int c = 0; c < a.n; c++) {
    bool va = !a.c[c].valid || bit_at(a.c[c].valid, ia);
    bool vb = !b.c[c].valid || bit_at(b.c[c].valid, ib);
    if (!va || !vb) {
      if (null_equals_null && !va && !vb) continue;
      return false;
    }
    uint64_t alo, ahi, blo, bhi;
    load_words(a.c[c], ia, alo, ahi);
    load_words(b.c[c], ib, blo, bhi);
    if (float_total_order && a.c[c].type == 4) {
      alo = ((const uint64_t*)a.c[c].data)[ia];
      blo = ((const uint64_t*)b.c[c].data)[ib];
    }
    if (alo != blo || ahi != bhi) return false;
  }
  return true;
}

// date_part(YEAR | MONTH | DAY, Date32): days since 1970-01-01 -> proleptic Gregorian civil date (the algorithm chrono's
// NaiveDate::from_num_days_from_ce restates; era = 400-year cycle of 146097 days starting on 0000-03-01)
__host__ __device__ __forceinline__ int32_t date32_part(int32_t days, int part) {
  const int64_t z = (int64_t)days + 719468;
  const int64_t era = (z >= 0 ? z : z - 146096) / 146097;
  const int64_t doe = z - era * 146097;                                   // [0, 146096]
  const int64_t yoe = (doe - doe / 1460 + doe / 36524 - doe / 146096) / 365;  // [0, 399]
  const int64_t doy = doe - (365 * yoe + yoe / 4 - yoe / 100);            // [0, 365], year starts on March 1
  const int64_t mp = (5 * doy + 2) / 153;                                 // [0, 11]
  const int64_t d = doy - (153 * mp + 2) / 5 + 1;
  const int64_t m = mp < 10 ? mp + 3 : mp - 9;
  const int64_t y = yoe + era * 400 + (m <= 2 ? 1 : 0);
  return (int32_t)(part == 0 ? y : part == 1 ? m : d);
}

// h % n for a small runtime n (partition counts): the compiler expands a 64-bit remainder by a runtime value into a
// ~150-instruction loop, which made the partition passes compute-bound.  Exact: h = hi * 2^32 + lo, so
// h mod n = ((hi mod n) * (2^32 mod n) + (lo mod n)) mod n, each 32-bit remainder by Lemire's fastmod
// (M = floor((2^64 - 1) / n) + 1; x mod n = mulhi64(M * x, n) for every 32-bit x).
struct FastMod {
  uint64_t M;
  uint32_t n, c;  // c = 2^32 mod n
};
inline FastMod fastmod_for(uint32_t n) { return FastMod{~0ull / n + 1ull, n, (uint32_t)((1ull << 32) % n)}; }
__host__ __device__ __forceinline__ uint32_t fastmod_u32(uint32_t x, const FastMod& f) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (uint32_t)__umul64hi(f.M * x, (uint64_t)f.n);
#else
  return (uint32_t)(((unsigned __int128)(f.M * x) * f.n) >> 64);
#endif
}
__host__ __device__ __forceinline__ uint32_t fastmod_u64(uint64_t h, const FastMod& f) {
  if ((f.n & (f.n - 1)) == 0) return (uint32_t)h & (f.n - 1);
  const uint32_t a = fastmod_u32((uint32_t)(h >> 32), f), b = fastmod_u32((uint32_t)h, f);
  return fastmod_u32(a * f.c + b, f);
}

// grid sizing for HBM-bound grid-stride kernels: enough workgroups to fill 256 CUs x 8
inline int grid_for(int64_t work_items, int per_block) {
  int64_t b = (work_items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > 256 * 8) b = 256 * 8;
  return (int)b;
}

// ---- decoupled look-back over per-tile states (one 8-byte agent-scope atomic granule per tile: status in the top two bits — 0 not
// published, TS_AGG the tile's own aggregate, TS_PFX its inclusive prefix — the value below).  Tiles are handed out in order by an
// atomic ticket, so a tile only ever waits on tiles whose workgroup is already resident.
constexpr uint64_t TS_AGG = 1ull << 62, TS_PFX = 2ull << 62, TS_VAL = (1ull << 62) - 1;

__device__ __forceinline__ uint64_t ts_load(const uint64_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ts_store(uint64_t* p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// wave 0 of a tile in ordered mode: exclusive prefix of `agg` over all earlier tiles
__device__ __forceinline__ uint64_t lookback_exclusive(uint64_t* __restrict__ tile_state, int64_t tile, uint64_t agg) {
  const unsigned lane = lane_id();
  uint64_t excl = 0;
  if (tile > 0) {
    if (lane == 0) ts_store(&tile_state[tile], TS_AGG | agg);
    // each lane polls LB consecutive predecessors, nearest first => a 64*LB-tile window per round
    constexpr int LB = 4;
    int64_t base = tile - 1;
    for (;;) {
      uint64_t sum, pfx_lanes;
      for (;;) {
        uint64_t st[LB];
#pragma unroll
        for (int q = 0; q < LB; q++) {
          const int64_t idx = base - ((int64_t)lane * LB + q);
          st[q] = idx >= 0 ? ts_load(&tile_state[idx]) : TS_PFX;  // virtual inclusive prefix 0 before tile 0
        }
        sum = 0;
        bool lane_pfx = false, lane_block = false;  // block = an unpublished tile sits before this lane's first prefix
#pragma unroll
        for (int q = 0; q < LB; q++) {
          const unsigned status = (unsigned)(st[q] >> 62);
          if (!lane_pfx && !lane_block) {
            if (status == 0) lane_block = true;
            else {
              sum += st[q] & TS_VAL;
              lane_pfx = status == 2;
            }
          }
        }
        pfx_lanes = ballot64(lane_pfx);
        const uint64_t block_lanes = ballot64(lane_block);
        const int first_pfx = pfx_lanes ? __builtin_ctzll(pfx_lanes) : 64;
        const int first_block = block_lanes ? __builtin_ctzll(block_lanes) : 64;
        // lanes before the nearest prefix lane must be fully published (a blocked lane never
        // reports a prefix, so first_block != first_pfx)
        if (first_block > first_pfx || !block_lanes) break;
        __builtin_amdgcn_s_sleep(1);
      }
      const int first_pfx = pfx_lanes ? __builtin_ctzll(pfx_lanes) : 64;
      excl += wave_sum((int)lane <= first_pfx ? sum : 0ull);
      if (first_pfx < 64) break;
      base -= 64 * LB;
    }
  }
  if (lane == 0) ts_store(&tile_state[tile], TS_PFX | (excl + agg));
  return excl;
}

}  // namespace dfgpu
