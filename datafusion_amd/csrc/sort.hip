// sort.hip — K11/K12: SortExec and TopK on device.
//
// Reference: sort_batch = arrow-ord lexsort_to_indices + take_arrays (physical-plan/src/sorts/
// sort.rs:894-914); TopK::insert_batch row-encodes the sort keys with arrow-row's RowConverter and
// keeps the k smallest rows in a heap (physical-plan/src/topk/mod.rs:397-470).
//
// Device design: the idea of arrow-row — one order-preserving fixed-width key per row, compared as
// an unsigned integer — but RANGE-COMPRESSED, because every radix pass moves the whole key:
//   1. k_key_ranges : min / max of each key column's order-preserving transform (sign bit flipped,
//                     f64 total order) over the non-null rows.
//   2. k_pack_keys  : key = concatenation, most significant column first, of
//                       [null flag bit if the column is nullable][value - min (ASC) or max - value (DESC)]
//                     in ceil(log2(max - min + 1)) bits per column — TPC-H (o_orderdate, o_orderkey DESC)
//                     packs into 42 bits of ONE u64 instead of 12 normalised bytes.
//   3. full sort    : stable LSD radix sort of (key words, u32 row id) over the used bits only, up to
//                     8 bits per pass.  A pass = per-tile digit histogram -> exclusive scan -> scatter.
//                     The scatter ranks a 256-row chunk with wave64 ballot peer masks + a cross-wave
//                     prefix in LDS, stages the tile sorted by digit in LDS and writes every digit's
//                     run contiguously (coalesced), instead of one scattered 8-byte store per row.
//      TopK         : MSD radix select over the same digits narrows the candidates to the bucket
//                     holding the k-th row (Q3: k = 10), then the few survivors are sorted.
//   4. take         : output columns gathered by the sorted row ids.  Ties keep input order (stable).
#include <algorithm>
#include <cstdlib>

#include "device.hpp"
#include "internal.hpp"
#include "onesweep.hpp"
#include "records.hpp"

namespace dfgpu {

constexpr int MAX_SORT_KEYS = 8;
constexpr int MAX_KEY_WORDS = 3;  // 192 packed bits

struct PackCol {
  const void* data;
  const uint64_t* valid;
  int type;
  int desc;
  int nulls_first;
  int has_null_bit;
  uint64_t base_lo, base_hi;  // min (ASC) or max (DESC) of the transformed value
  int bits;                   // value field width
  int shift;                  // bit offset of the value field in the packed key; the null bit sits at shift + bits
  // one-word keys are packed in MIXED RADIX instead (k_pack_keys64): key = sum(digit_k * mult_k), digit_k = value - base (or
  // base - value) [+ range_k when the NULL flag is set], mult_k = product of the later columns' digit ranges.  No holes: a
  // date column spanning 2406 days costs log2(2406) bits, not 12, and keys spread evenly over [0, P) when the columns do.
  uint64_t mult, range;
};
struct PackCols {
  PackCol c[MAX_SORT_KEYS];
  int n;
};

// order-preserving transform of row i of a key column to an unsigned 128-bit integer
__device__ __forceinline__ u128 key_transform(int type, const void* data, int64_t i) {
  switch (type) {
    case DFGPU_INT32: case DFGPU_DATE32: return (u128)((uint32_t)((const int32_t*)data)[i] ^ 0x80000000u);
    case DFGPU_UINT32: return (u128)((const uint32_t*)data)[i];
    case DFGPU_INT64: return (u128)(((const uint64_t*)data)[i] ^ 0x8000000000000000ull);
    case DFGPU_UINT64: return (u128)((const uint64_t*)data)[i];
    case DFGPU_FLOAT64: {
      uint64_t b = ((const uint64_t*)data)[i];
      return (u128)((b >> 63) ? ~b : (b ^ 0x8000000000000000ull));  // f64::total_cmp order
    }
    case DFGPU_DECIMAL128: {
      const uint64_t* p = (const uint64_t*)data + 2 * i;
      return ((u128)(p[1] ^ 0x8000000000000000ull) << 64) | (u128)p[0];
    }
    default: return (u128)((const uint8_t*)data)[i];
  }
}

// per-block min / max of every key column's transform over valid rows: out[(block * n + col) * 2 + {0,1}]
__global__ __launch_bounds__(BLOCK) void k_key_ranges(PackCols pc, int64_t n, u128* __restrict__ out) {
  __shared__ u128 s_mn[BLOCK / WAVE], s_mx[BLOCK / WAVE];
  for (int c = 0; c < pc.n; c++) {
    const PackCol& k = pc.c[c];
    u128 mn = ~(u128)0, mx = 0;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
      if (k.valid && !bit_at(k.valid, i)) continue;
      u128 t = key_transform(k.type, k.data, i);
      mn = t < mn ? t : mn;
      mx = t > mx ? t : mx;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      uint64_t olo = __shfl_xor((uint64_t)mn, d, 64), ohi = __shfl_xor((uint64_t)(mn >> 64), d, 64);
      u128 o = ((u128)ohi << 64) | olo;
      mn = o < mn ? o : mn;
      olo = __shfl_xor((uint64_t)mx, d, 64);
      ohi = __shfl_xor((uint64_t)(mx >> 64), d, 64);
      o = ((u128)ohi << 64) | olo;
      mx = o > mx ? o : mx;
    }
    if (lane_id() == 0) {
      s_mn[threadIdx.x >> 6] = mn;
      s_mx[threadIdx.x >> 6] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int w = 1; w < BLOCK / WAVE; w++) {
        mn = s_mn[w] < mn ? s_mn[w] : mn;
        mx = s_mx[w] > mx ? s_mx[w] : mx;
      }
      out[((int64_t)blockIdx.x * pc.n + c) * 2] = mn;
      out[((int64_t)blockIdx.x * pc.n + c) * 2 + 1] = mx;
    }
    __syncthreads();
  }
}

// OR a value of up to 64 bits into a 192-bit key at bit offset `shift` (static word indices: stays in registers)
__device__ __forceinline__ void or_bits(uint64_t& w0, uint64_t& w1, uint64_t& w2, int shift, uint64_t v) {
  const int wi = shift >> 6, sb = shift & 63;
  const uint64_t lo = v << sb, hi = sb ? (v >> (64 - sb)) : 0ull;
  if (wi == 0) { w0 |= lo; w1 |= hi; }
  else if (wi == 1) { w1 |= lo; w2 |= hi; }
  else if (wi == 2) { w2 |= lo; }
}

__global__ __launch_bounds__(BLOCK) void k_pack_keys(PackCols pc, int64_t n, int nwords, uint64_t* __restrict__ o0, uint64_t* __restrict__ o1,
                                                     uint64_t* __restrict__ o2, uint32_t* __restrict__ idx) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    uint64_t w0 = 0, w1 = 0, w2 = 0;
    for (int c = 0; c < pc.n; c++) {
      const PackCol& k = pc.c[c];
      const bool ok = !k.valid || bit_at(k.valid, i);
      if (ok && k.bits > 0) {
        const u128 t = key_transform(k.type, k.data, i);
        const u128 base = ((u128)k.base_hi << 64) | k.base_lo;
        const u128 e = k.desc ? base - t : t - base;
        or_bits(w0, w1, w2, k.shift, (uint64_t)e);
        if (k.bits > 64) or_bits(w0, w1, w2, k.shift + 64, (uint64_t)(e >> 64));
      }
      // null flag: NULLS FIRST => nulls 0 / values 1; NULLS LAST => values 0 / nulls 1
      if (k.has_null_bit && (ok == (k.nulls_first != 0))) or_bits(w0, w1, w2, k.shift + k.bits, 1ull);
    }
    o0[i] = w0;
    if (nwords > 1) o1[i] = w1;
    if (nwords > 2) o2[i] = w2;
    idx[i] = (uint32_t)i;
  }
}

// The common case — every key column at most 64 bits wide and the packed key one word — without 128-bit arithmetic, two rows
// per thread and no row-id column (the first radix pass takes positions as ids): k_pack_keys ran at 1.7 TB/s on u128 shifts.
__device__ __forceinline__ uint64_t key_transform64(int type, const void* data, int64_t i) {
  switch (type) {
    case DFGPU_INT32: case DFGPU_DATE32: return (uint64_t)((uint32_t)((const int32_t*)data)[i] ^ 0x80000000u);
    case DFGPU_UINT32: return (uint64_t)((const uint32_t*)data)[i];
    case DFGPU_INT64: return ((const uint64_t*)data)[i] ^ 0x8000000000000000ull;
    case DFGPU_UINT64: return ((const uint64_t*)data)[i];
    case DFGPU_FLOAT64: {
      const uint64_t b = ((const uint64_t*)data)[i];
      return (b >> 63) ? ~b : (b ^ 0x8000000000000000ull);
    }
    default: return (uint64_t)((const uint8_t*)data)[i];
  }
}
// the mixed-radix key of row i
__device__ __forceinline__ uint64_t pack_key64(const PackCols& pc, int64_t i) {
  uint64_t w = 0;
  for (int c = 0; c < pc.n; c++) {
    const PackCol& k = pc.c[c];
    const bool ok = !k.valid || bit_at(k.valid, i);
    uint64_t digit = 0;
    if (ok && k.range > 1) {
      const uint64_t t = key_transform64(k.type, k.data, i);
      digit = k.desc ? k.base_lo - t : t - k.base_lo;
    }
    // NULL flag as the column's top digit: NULLS FIRST => nulls 0 / values 1; NULLS LAST => values 0 / nulls 1
    if (k.has_null_bit && (ok == (k.nulls_first != 0))) digit += k.range;
    w += digit * k.mult;
  }
  return w;
}
__global__ __launch_bounds__(BLOCK) void k_pack_keys64(PackCols pc, int64_t n, uint64_t* __restrict__ o0) {
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  for (int64_t i0 = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i0 < n; i0 += 2 * stride) {
    uint64_t w[2] = {0, 0};
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int64_t i = i0 + u * stride;
      if (i >= n) continue;
      w[u] = pack_key64(pc, i);
    }
    o0[i0] = w[0];
    if (i0 + stride < n) o0[i0 + stride] = w[1];
  }
}
// keys of listed rows (ids), or of every `every`-th row (ids == null): the TopK sample and the TopK survivors
__global__ __launch_bounds__(BLOCK) void k_pack_keys64_rows(PackCols pc, const int64_t* __restrict__ ids, int64_t every, int64_t m, uint64_t* __restrict__ out) {
  for (int64_t j = (int64_t)blockIdx.x * BLOCK + threadIdx.x; j < m; j += (int64_t)gridDim.x * BLOCK) out[j] = pack_key64(pc, ids ? ids[j] : j * every);
}
// mask of the rows whose key is <= limit, computed from the key columns (no packed key array)
__global__ __launch_bounds__(BLOCK) void k_key_limit_mask(PackCols pc, int64_t n, const uint64_t* __restrict__ limit_p, uint64_t* __restrict__ mask) {
  const uint64_t limit = *limit_p;   // (left on the device by k_select_kth: no host round trip between the sample and this pass)
  const int64_t n_words = (n + 63) >> 6;
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * BLOCK) >> 6;
  for (int64_t w = wave; w < n_words; w += n_waves) {
    const int64_t i = (w << 6) + lane_id();
    const uint64_t m = ballot64(i < n && pack_key64(pc, i) <= limit);
    if (lane_id() == 0) mask[w] = m;
  }
}

// floor(x / div) for a divisor fixed per sort: magic = floor(2^64 / div) underestimates the quotient by at most 2
struct DivBy {
  uint64_t div, magic;  // div == 0: no division (x itself)
};
static DivBy div_by(uint64_t d) { return DivBy{d, d > 1 ? (uint64_t)(((u128)1 << 64) / d) : 0ull}; }
__device__ __forceinline__ uint64_t div_apply(uint64_t x, const DivBy& d) {
  if (d.div <= 1) return x;
  uint64_t q = __umul64hi(x, d.magic);
  uint64_t r = x - q * d.div;
  while (r >= d.div) {
    q++;
    r -= d.div;
  }
  return q;
}

// ------------------------------------------------------------------------------ LSD radix pass
// A digit never straddles a key word: (word, shift, bits <= 8).
// rows per thread: a tile of BLOCK * items rows is staged in LDS (keys + ids), 2048 rows for 1-2 key words
constexpr int rs_items(int nwords) { return nwords >= 3 ? 4 : 8; }
struct KeyWords {
  const uint64_t* w[MAX_KEY_WORDS];
};
struct SortBufs {
  uint64_t* w[MAX_KEY_WORDS];
  uint32_t* idx;
};

// per-tile digit histogram: counts[digit * n_tiles + tile]
__global__ __launch_bounds__(BLOCK) void k_rs_hist(const uint64_t* __restrict__ word, int64_t n, DivBy dv, int shift, int bits, int items, int64_t n_tiles,
                                                   uint32_t* __restrict__ counts) {
  __shared__ unsigned int sh[256];
  const unsigned mask = (1u << bits) - 1u;
  for (int64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    sh[threadIdx.x] = 0;
    __syncthreads();
    const int64_t lo = t * (int64_t)(BLOCK * items);
    for (int c = 0; c < items; c++) {
      const int64_t i = lo + c * BLOCK + threadIdx.x;
      if (i < n) atomicAdd(&sh[(unsigned)(div_apply(word[i], dv) >> shift) & mask], 1u);
    }
    __syncthreads();
    if ((int)threadIdx.x <= (int)mask) counts[(int64_t)threadIdx.x * n_tiles + t] = sh[threadIdx.x];
    __syncthreads();
  }
}

// stable scatter of one tile by one digit, second generation.  A wave owns a CONTIGUOUS 64 x ITEMS-row segment of the tile and
// ranks its rows against wave-private digit counters in LDS (ballot peer masks; LDS operations of one wave are ordered, so
// no block barrier is needed between items) — 4 block barriers per tile instead of 3 per item.  The tile is then staged in LDS
// sorted by digit and every digit's run is written contiguously.
template <int NW, int ITEMS>
__global__ __launch_bounds__(BLOCK) void k_rs_scatter2(KeyWords k, const uint32_t* __restrict__ idx_in, int64_t n, DivBy dv, int dword, int shift, int bits, int64_t n_tiles,
                                                      const uint64_t* __restrict__ offsets, SortBufs out, int xcd_map) {
  constexpr int TILE = BLOCK * ITEMS;
  constexpr int NWAVE = BLOCK / WAVE;
  __shared__ uint64_t s_key[NW][TILE];
  __shared__ uint32_t s_idx[TILE];
  __shared__ uint8_t s_dig[TILE];
  __shared__ unsigned int s_cnt[NWAVE][256];   // ranking: rows of each digit seen so far by the wave; then the wave's exclusive prefix over earlier waves
  __shared__ unsigned int s_start[256];        // exclusive scan of the tile's digit counts
  __shared__ unsigned int s_wtot[NWAVE];
  __shared__ unsigned long long s_goff[256];   // global output offset of each digit's run
  const unsigned mask = (1u << bits) - 1u;
  const int wave = threadIdx.x >> 6;
  const unsigned lane = lane_id();
  for (int64_t tl = blockIdx.x; tl < n_tiles; tl += gridDim.x) {
    const int64_t t = xcd_map ? xcd_tile(tl, n_tiles) : tl;
    const int64_t lo = t * TILE;
    const int tile_rows = (int)((n - lo) < TILE ? (n - lo) : TILE);
#pragma unroll
    for (int w = 0; w < NWAVE; w++) s_cnt[w][threadIdx.x] = 0;
    if ((int)threadIdx.x <= (int)mask) s_goff[threadIdx.x] = offsets[(int64_t)threadIdx.x * n_tiles + t];
    __syncthreads();
    uint64_t key[NW][ITEMS];
    uint32_t id[ITEMS];
    unsigned dig[ITEMS], rank[ITEMS];
#pragma unroll
    for (int c = 0; c < ITEMS; c++) {  // all loads of the segment in flight together
      const int j = (wave * ITEMS + c) * WAVE + (int)lane;
      const int64_t src = lo + (j < tile_rows ? j : 0);
#pragma unroll
      for (int w = 0; w < NW; w++) key[w][c] = k.w[w][src];
      id[c] = idx_in ? idx_in[src] : (uint32_t)src;
    }
#pragma unroll
    for (int c = 0; c < ITEMS; c++) {
      const int j = (wave * ITEMS + c) * WAVE + (int)lane;
      const bool in = j < tile_rows;
      uint64_t kw = 0;
#pragma unroll
      for (int w = 0; w < NW; w++)
        if (w == dword) kw = key[w][c];
      dig[c] = in ? ((unsigned)(div_apply(kw, dv) >> shift) & mask) : 0u;
      uint64_t peers = ballot64(in);
      for (int b = 0; b < bits; b++) {
        const uint64_t bal = ballot64((dig[c] >> b) & 1u);
        peers &= ((dig[c] >> b) & 1u) ? bal : ~bal;
      }
      const unsigned r_in_wave = mbcnt(peers);
      const unsigned base = s_cnt[wave][dig[c]];
      if (in && r_in_wave == 0) s_cnt[wave][dig[c]] = base + (unsigned)__popcll(peers);
      rank[c] = base + r_in_wave;
    }
    __syncthreads();
    {  // thread d: digit d's total, the waves' exclusive prefixes, and the exclusive scan over digits
      unsigned run = 0;
#pragma unroll
      for (int w = 0; w < NWAVE; w++) {
        const unsigned v = s_cnt[w][threadIdx.x];
        s_cnt[w][threadIdx.x] = run;
        run += v;
      }
      const unsigned inc = wave_inclusive_sum<unsigned>(run);
      if (lane == 63) s_wtot[wave] = inc;
      __syncthreads();
      unsigned base = 0;
      for (int w = 0; w < wave; w++) base += s_wtot[w];
      s_start[threadIdx.x] = base + inc - run;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < ITEMS; c++) {
      const int j = (wave * ITEMS + c) * WAVE + (int)lane;
      if (j < tile_rows) {
        const unsigned q = s_start[dig[c]] + s_cnt[wave][dig[c]] + rank[c];
#pragma unroll
        for (int w = 0; w < NW; w++) s_key[w][q] = key[w][c];
        s_idx[q] = id[c];
        s_dig[q] = (uint8_t)dig[c];
      }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < ITEMS; c++) {
      const int q = c * BLOCK + threadIdx.x;
      if (q < tile_rows) {
        const unsigned d = s_dig[q];
        const unsigned long long dst = s_goff[d] + (unsigned)(q - (int)s_start[d]);
#pragma unroll
        for (int w = 0; w < NW; w++) out.w[w][dst] = s_key[w][q];
        if (out.idx) out.idx[dst] = s_idx[q];   // (null: the caller groups keys and wants no row ids)
      }
    }
    __syncthreads();
  }
}

// The carried sort's pass (round 4, sort_table): k_rs_scatter2 with a 16-BYTE RECORD per row in the place of the row id — the
// columns of the output that the packed key does not hold (orders sorted by (o_orderdate, o_orderkey): o_custkey and
// o_shippriority), so that no take by row id ends the sort.  BUILD: the first pass reads the record's fields from the source
// columns (row = position); later passes read the records the pass before wrote.  Same ranking, same stability.
template <int ITEMS, bool BUILD>
__global__ __launch_bounds__(BLOCK) void k_rs_scatter_kv(const uint64_t* __restrict__ key_in, const uint4* __restrict__ rec_in, PackLayout L, int64_t n, DivBy dv, int shift,
                                                        int bits, int64_t n_tiles, const uint64_t* __restrict__ offsets, uint64_t* __restrict__ key_out,
                                                        uint4* __restrict__ rec_out, int xcd_map) {
  constexpr int TILE = BLOCK * ITEMS;
  constexpr int NWAVE = BLOCK / WAVE;
  extern __shared__ __align__(16) unsigned char kv_smem[];   // records | keys | digits of the staged tile (beyond the 64 KB static LDS allows at 16 rows per thread)
  uint4* s_rec = reinterpret_cast<uint4*>(kv_smem);
  uint64_t* s_key = reinterpret_cast<uint64_t*>(kv_smem + (size_t)TILE * 16);
  uint8_t* s_dig = kv_smem + (size_t)TILE * 24;
  __shared__ unsigned int s_cnt[NWAVE][256];
  __shared__ unsigned int s_start[256];
  __shared__ unsigned int s_wtot[NWAVE];
  __shared__ unsigned long long s_goff[256];
  const unsigned mask = (1u << bits) - 1u;
  const int wave = threadIdx.x >> 6;
  const unsigned lane = lane_id();
  for (int64_t tl = blockIdx.x; tl < n_tiles; tl += gridDim.x) {
    const int64_t t = xcd_map ? xcd_tile(tl, n_tiles) : tl;
    const int64_t lo = t * TILE;
    const int tile_rows = (int)((n - lo) < TILE ? (n - lo) : TILE);
#pragma unroll
    for (int w = 0; w < NWAVE; w++) s_cnt[w][threadIdx.x] = 0;
    if ((int)threadIdx.x <= (int)mask) s_goff[threadIdx.x] = offsets[(int64_t)threadIdx.x * n_tiles + t];
    __syncthreads();
    uint64_t key[ITEMS];
    uint4 rec[ITEMS];
    unsigned dig[ITEMS], rank[ITEMS];
#pragma unroll
    for (int c = 0; c < ITEMS; c++) {  // all loads of the segment in flight together
      const int j = (wave * ITEMS + c) * WAVE + (int)lane;
      const int64_t src = lo + (j < tile_rows ? j : 0);
      key[c] = key_in[src];
      if (BUILD) {
        uint64_t sl[2];
        record_build<2>(L, src, sl);
        rec[c] = uint4{(unsigned)sl[0], (unsigned)(sl[0] >> 32), (unsigned)sl[1], (unsigned)(sl[1] >> 32)};
      } else {
        const uint4 v = rec_in[src];   // (component by component: the 16-byte struct copy kept the whole array in scratch memory)
        rec[c] = uint4{v.x, v.y, v.z, v.w};
      }
    }
#pragma unroll
    for (int c = 0; c < ITEMS; c++) {
      const int j = (wave * ITEMS + c) * WAVE + (int)lane;
      const bool in = j < tile_rows;
      dig[c] = in ? ((unsigned)(div_apply(key[c], dv) >> shift) & mask) : 0u;
      uint64_t peers = ballot64(in);
      for (int b = 0; b < bits; b++) {
        const uint64_t bal = ballot64((dig[c] >> b) & 1u);
        peers &= ((dig[c] >> b) & 1u) ? bal : ~bal;
      }
      const unsigned r_in_wave = mbcnt(peers);
      const unsigned base = s_cnt[wave][dig[c]];
      if (in && r_in_wave == 0) s_cnt[wave][dig[c]] = base + (unsigned)__popcll(peers);
      rank[c] = base + r_in_wave;
    }
    __syncthreads();
    {
      unsigned run = 0;
#pragma unroll
      for (int w = 0; w < NWAVE; w++) {
        const unsigned v = s_cnt[w][threadIdx.x];
        s_cnt[w][threadIdx.x] = run;
        run += v;
      }
      const unsigned inc = wave_inclusive_sum<unsigned>(run);
      if (lane == 63) s_wtot[wave] = inc;
      __syncthreads();
      unsigned base = 0;
      for (int w = 0; w < wave; w++) base += s_wtot[w];
      s_start[threadIdx.x] = base + inc - run;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < ITEMS; c++) {
      const int j = (wave * ITEMS + c) * WAVE + (int)lane;
      if (j < tile_rows) {
        const unsigned q = s_start[dig[c]] + s_cnt[wave][dig[c]] + rank[c];
        s_key[q] = key[c];
        s_rec[q] = rec[c];
        s_dig[q] = (uint8_t)dig[c];
      }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < ITEMS; c++) {
      const int q = c * BLOCK + threadIdx.x;
      if (q < tile_rows) {
        const unsigned d = s_dig[q];
        const unsigned long long dst = s_goff[d] + (unsigned)(q - (int)s_start[d]);
        key_out[dst] = s_key[q];
        rec_out[dst] = s_rec[q];
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------ onesweep pass (round 5)
// The carried sort's top passes as ONE kernel per pass: no per-tile histogram kernel (a second read of the keys), no scan of a
// (digit x tile) matrix, no strided load of a tile's 256 offsets.  The digit totals of ALL passes come from one read of the key
// columns up front (k_os_hist); a tile then finds where its run of every digit starts with a decoupled look-back over the tiles
// before it — per digit, by the thread that owns the digit: {status, count} words of 32 bits, published AGG(regate) as soon as the
// tile's counts are known and PFX (inclusive prefix) once its own look-back is done.  Tiles are handed out by an atomic ticket, so a tile
// only ever waits on tiles whose workgroup is already resident.  While its predecessors publish, the tile stages its keys in LDS in digit
// order (that needs local positions only); the look-back's answer is needed for the write-out alone.  Keys and records go through the
// SAME 32 KB staging buffer one after the other: 38 KB of LDS per workgroup, four workgroups per CU.
// Measured with the phases switched off one by one (150 M orders, two passes, profiles/r5_sort_phases.md): the tiles copied where they lie
// 3.3 ms (this launch geometry streams at 4.3 TB/s), + ranking / staging / scattered write-out 4.0, + the look-back 5.1 (a look-back per
// CHUNK of 8 tiles, counted in a sweep of its own, was slower: 6.1).
// Stable (wave-private ballot ranking, as k_rs_scatter_kv).  Row counts below 2^30 (the status words hold 30-bit prefixes).
constexpr int OS_ITEMS = 8;
constexpr int OS_TILE = BLOCK * OS_ITEMS;
constexpr int OS_MAX_PASSES = 4;
struct OsDigits {
  int shift[OS_MAX_PASSES], bits[OS_MAX_PASSES];
  int n;
};

// A tile's keys and records read off the SOURCE columns, column by column: the loads of all of a lane's rows from one column are in
// flight together (the row-by-row form — pack_key64 / record_build per row — walks the column list once per row, every load behind a
// uniform branch of its own: the first pass of the carried sorts spent 2.5 ms there for 3.6 GB).  Key columns without NULLs only (what
// the carried sorts take).
template <int ITEMS, typename K>   // (K = uint32_t where the whole key fits 32 bits: half the registers)
__device__ __forceinline__ void tile_keys(const PackCols& pc, const uint32_t (&src)[ITEMS], K (&key)[ITEMS]) {
#pragma unroll
  for (int c = 0; c < ITEMS; c++) key[c] = 0;
  for (int k = 0; k < pc.n; k++) {
    const PackCol& col = pc.c[k];
    uint64_t t[ITEMS];
    switch (col.type) {
      case DFGPU_INT32: case DFGPU_DATE32:
#pragma unroll
        for (int c = 0; c < ITEMS; c++) t[c] = (uint64_t)(reinterpret_cast<const uint32_t*>(col.data)[src[c]] ^ 0x80000000u);
        break;
      case DFGPU_UINT32:
#pragma unroll
        for (int c = 0; c < ITEMS; c++) t[c] = (uint64_t)reinterpret_cast<const uint32_t*>(col.data)[src[c]];
        break;
      case DFGPU_INT64:
#pragma unroll
        for (int c = 0; c < ITEMS; c++) t[c] = reinterpret_cast<const uint64_t*>(col.data)[src[c]] ^ 0x8000000000000000ull;
        break;
      case DFGPU_UINT64:
#pragma unroll
        for (int c = 0; c < ITEMS; c++) t[c] = reinterpret_cast<const uint64_t*>(col.data)[src[c]];
        break;
      default:
#pragma unroll
        for (int c = 0; c < ITEMS; c++) t[c] = (uint64_t)reinterpret_cast<const uint8_t*>(col.data)[src[c]];
        break;
    }
    const uint64_t base = col.base_lo, mult = col.mult;
    const bool desc = col.desc != 0;
#pragma unroll
    for (int c = 0; c < ITEMS; c++) key[c] += (K)((desc ? base - t[c] : t[c] - base) * mult);
  }
}
template <int ITEMS, int NS>
__device__ __forceinline__ void tile_records(const PackLayout& L, const uint32_t (&src)[ITEMS], uint64_t (&sl)[ITEMS][NS]) {
#pragma unroll
  for (int c = 0; c < ITEMS; c++)
#pragma unroll
    for (int q = 0; q < NS; q++) sl[c][q] = 0;
  for (int f = 0; f < L.n; f++) {
    const int o = L.offset[f];
    switch (L.width[f]) {
      case 16: {
        uint4 v[ITEMS];
#pragma unroll
        for (int c = 0; c < ITEMS; c++) {
          const uint4 u = reinterpret_cast<const uint4*>(L.src[f])[src[c]];
          v[c] = uint4{u.x, u.y, u.z, u.w};
        }
#pragma unroll
        for (int c = 0; c < ITEMS; c++) {
          slot_or<NS>(sl[c], o >> 3, ((uint64_t)v[c].y << 32) | v[c].x);
          slot_or<NS>(sl[c], (o >> 3) + 1, ((uint64_t)v[c].w << 32) | v[c].z);
        }
        break;
      }
      case 8: {
        uint64_t v[ITEMS];
#pragma unroll
        for (int c = 0; c < ITEMS; c++) v[c] = reinterpret_cast<const uint64_t*>(L.src[f])[src[c]];
#pragma unroll
        for (int c = 0; c < ITEMS; c++) slot_or<NS>(sl[c], o >> 3, v[c]);
        break;
      }
      case 4: {
        uint32_t v[ITEMS];
#pragma unroll
        for (int c = 0; c < ITEMS; c++) v[c] = reinterpret_cast<const uint32_t*>(L.src[f])[src[c]];
#pragma unroll
        for (int c = 0; c < ITEMS; c++) slot_or<NS>(sl[c], o >> 3, (uint64_t)v[c] << ((o & 4) * 8));
        break;
      }
      default: {
        uint8_t v[ITEMS];
#pragma unroll
        for (int c = 0; c < ITEMS; c++) v[c] = reinterpret_cast<const uint8_t*>(L.src[f])[src[c]];
#pragma unroll
        for (int c = 0; c < ITEMS; c++) slot_or<NS>(sl[c], o >> 3, (uint64_t)v[c] << ((o & 7) * 8));
        break;
      }
    }
  }
}

// digit totals of every pass: hist[pass * 256 + digit].  BUILD: the keys are computed from the key columns (no packed key array exists)
template <bool BUILD>
__global__ __launch_bounds__(BLOCK) void k_os_hist(const uint64_t* __restrict__ key_in, PackCols pc, int64_t n, DivBy dv, OsDigits dg, unsigned long long* __restrict__ hist) {
  __shared__ unsigned int s_h[OS_MAX_PASSES][256];
#pragma unroll
  for (int p = 0; p < OS_MAX_PASSES; p++) s_h[p][threadIdx.x] = 0;
  __syncthreads();
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  for (int64_t i0 = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i0 < n; i0 += 4 * stride) {
    uint64_t k[4];
    if (BUILD) {   // (column by column: tile_keys)
      uint32_t src[4];
#pragma unroll
      for (int u = 0; u < 4; u++) src[u] = (uint32_t)(i0 + u * stride < n ? i0 + u * stride : n - 1);
      tile_keys<4, uint64_t>(pc, src, k);
    } else {
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int64_t i = i0 + u * stride;
        k[u] = i < n ? key_in[i] : 0ull;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (i0 + u * stride >= n) continue;
      const uint64_t b = div_apply(k[u], dv);
      for (int p = 0; p < dg.n; p++) atomicAdd(&s_h[p][(unsigned)(b >> dg.shift[p]) & ((1u << dg.bits[p]) - 1u)], 1u);
    }
  }
  __syncthreads();
  for (int p = 0; p < dg.n; p++)
    if (s_h[p][threadIdx.x]) atomicAdd(&hist[p * 256 + threadIdx.x], (unsigned long long)s_h[p][threadIdx.x]);
}
// k_key_limit_mask for key columns without NULLs (integer / date types): four mask words per wave iteration, the keys of a lane's four rows
// loaded column by column before any is packed (tile_keys)
__global__ __launch_bounds__(BLOCK) void k_key_limit_mask_plain(PackCols pc, int64_t n, const uint64_t* __restrict__ limit_p, uint64_t* __restrict__ mask) {
  const uint64_t limit = *limit_p;
  const int64_t n_words = (n + 63) >> 6;
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * BLOCK) >> 6;
  for (int64_t w0 = wave * 4; w0 < n_words; w0 += n_waves * 4) {
    uint32_t src[4];
    uint64_t k[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int64_t i = ((w0 + u) << 6) + lane_id();
      src[u] = (uint32_t)(i < n ? i : n - 1);
    }
    tile_keys<4, uint64_t>(pc, src, k);
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int64_t i = ((w0 + u) << 6) + lane_id();
      const uint64_t m = ballot64(i < n && k[u] <= limit);
      if (lane_id() == 0 && w0 + u < n_words) mask[w0 + u] = m;
    }
  }
}
// exclusive scan of every pass's 256 totals (one workgroup of 256 threads per pass)
__global__ __launch_bounds__(BLOCK) void k_os_bases(const unsigned long long* __restrict__ hist, unsigned long long* __restrict__ base) {
  __shared__ unsigned long long s_w[BLOCK / WAVE];
  const unsigned long long v = hist[blockIdx.x * 256 + threadIdx.x];
  const unsigned long long inc = wave_inclusive_sum_dpp(v);
  if (lane_id() == 63) s_w[threadIdx.x >> 6] = inc;
  __syncthreads();
  unsigned long long b = 0;
  for (int w = 0; w < (int)(threadIdx.x >> 6); w++) b += s_w[w];
  base[blockIdx.x * 256 + threadIdx.x] = b + inc - v;
}

template <bool BUILD>
__global__ __launch_bounds__(BLOCK, (BUILD ? 3 : 4)) void k_os_pass(const uint64_t* __restrict__ key_in, const uint4* __restrict__ rec_in, PackCols pc, PackLayout L, int64_t n, DivBy dv, int shift,
                                                     int bits, int64_t n_tiles, const unsigned long long* __restrict__ bin_base, uint32_t* __restrict__ tile_state,
                                                     unsigned* __restrict__ ticket, uint64_t* __restrict__ key_out, uint4* __restrict__ rec_out) {
  constexpr int NWAVE = BLOCK / WAVE;
  __shared__ __align__(16) unsigned char s_stage[OS_TILE * 16];   // the tile in digit order: its keys, then its records
  __shared__ uint8_t s_dig[OS_TILE];
  __shared__ uint16_t s_cnt[NWAVE][256];   // ranking: rows of each digit seen so far by the wave; then the wave's exclusive prefix over earlier waves
  __shared__ uint16_t s_start[256];        // exclusive scan of the tile's digit counts
  __shared__ unsigned int s_goff[256];     // staged slot q of digit d goes to output position q + s_goff[d] (32-bit wrap-around arithmetic: n < 2^30)
  __shared__ unsigned int s_wtot[NWAVE];
  __shared__ unsigned int s_tile;
  uint64_t* s_key = reinterpret_cast<uint64_t*>(s_stage);
  uint4* s_rec = reinterpret_cast<uint4*>(s_stage);
  const unsigned mask = (1u << bits) - 1u;
  const int wave = threadIdx.x >> 6;
  const unsigned lane = lane_id();
  for (;;) {
    if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
#pragma unroll
    for (int w = 0; w < NWAVE; w++) s_cnt[w][threadIdx.x] = 0;
    __syncthreads();
    const int64_t t = (int64_t)s_tile;
    if (t >= n_tiles) return;
    const int64_t lo = t * OS_TILE;
    const int tile_rows = (int)((n - lo) < OS_TILE ? (n - lo) : OS_TILE);
    uint64_t key[OS_ITEMS];
    uint4 rec[OS_ITEMS];
    unsigned dig[OS_ITEMS], rank[OS_ITEMS];
    if (BUILD) {   // (column by column: tile_keys)
      uint32_t src[OS_ITEMS];   // (row numbers below 2^30)
#pragma unroll
      for (int c = 0; c < OS_ITEMS; c++) {
        const int j = (wave * OS_ITEMS + c) * WAVE + (int)lane;
        src[c] = (uint32_t)(lo + (j < tile_rows ? j : 0));
      }
      tile_keys<OS_ITEMS, uint64_t>(pc, src, key);
      uint64_t sl[OS_ITEMS][2];
      tile_records<OS_ITEMS, 2>(L, src, sl);
#pragma unroll
      for (int c = 0; c < OS_ITEMS; c++) rec[c] = uint4{(unsigned)sl[c][0], (unsigned)(sl[c][0] >> 32), (unsigned)sl[c][1], (unsigned)(sl[c][1] >> 32)};
    } else {
#pragma unroll
      for (int c = 0; c < OS_ITEMS; c++) {  // all loads of the wave's segment in flight together
        const int j = (wave * OS_ITEMS + c) * WAVE + (int)lane;
        const int64_t src = lo + (j < tile_rows ? j : 0);
        key[c] = key_in[src];
        const uint4 v = rec_in[src];   // (component by component: the 16-byte struct copy kept the whole array in scratch memory)
        rec[c] = uint4{v.x, v.y, v.z, v.w};
      }
    }
#pragma unroll
    for (int c = 0; c < OS_ITEMS; c++) {
      const int j = (wave * OS_ITEMS + c) * WAVE + (int)lane;
      const bool in = j < tile_rows;
      dig[c] = in ? ((unsigned)(div_apply(key[c], dv) >> shift) & mask) : 0u;
      uint64_t peers = ballot64(in);
      for (int b = 0; b < bits; b++) {
        const uint64_t bal = ballot64((dig[c] >> b) & 1u);
        peers &= ((dig[c] >> b) & 1u) ? bal : ~bal;
      }
      const unsigned r_in_wave = mbcnt(peers);
      const unsigned base = s_cnt[wave][dig[c]];
      if (in && r_in_wave == 0) s_cnt[wave][dig[c]] = (uint16_t)(base + (unsigned)__popcll(peers));
      rank[c] = base + r_in_wave;
    }
    __syncthreads();
    unsigned run = 0;   // thread d: digit d's rows in this tile
    {
#pragma unroll
      for (int w = 0; w < NWAVE; w++) {
        const unsigned v = s_cnt[w][threadIdx.x];
        s_cnt[w][threadIdx.x] = (uint16_t)run;
        run += v;
      }
      // the tile's count of this digit is public from here on: the tiles behind it can add it up while this one stages its rows
      if (t > 0 && threadIdx.x <= mask) os_store(&tile_state[t * 256 + threadIdx.x], OS_AGG | run);
      const unsigned inc = wave_inclusive_sum<unsigned>(run);
      if (lane == 63) s_wtot[wave] = inc;
      __syncthreads();
      unsigned base = 0;
      for (int w = 0; w < wave; w++) base += s_wtot[w];
      s_start[threadIdx.x] = (uint16_t)(base + inc - run);
    }
    __syncthreads();
    unsigned q[OS_ITEMS];
#pragma unroll
    for (int c = 0; c < OS_ITEMS; c++) {
      const int j = (wave * OS_ITEMS + c) * WAVE + (int)lane;
      q[c] = 0xFFFFFFFFu;
      if (j < tile_rows) {
        q[c] = (unsigned)s_start[dig[c]] + (unsigned)s_cnt[wave][dig[c]] + rank[c];
        s_key[q[c]] = key[c];
        s_dig[q[c]] = (uint8_t)dig[c];
      }
    }
    // ---- look-back: thread d adds up digit d's counts over the tiles before this one, nearest first, until it meets an inclusive prefix
    if (threadIdx.x <= mask) {
      const unsigned excl = os_look_back(tile_state, t, threadIdx.x);
      os_store(&tile_state[t * 256 + threadIdx.x], OS_PFX | (excl + run));
      s_goff[threadIdx.x] = (unsigned)bin_base[threadIdx.x] + excl - (unsigned)s_start[threadIdx.x];
    }
    __syncthreads();
    for (int qq = threadIdx.x; qq < tile_rows; qq += BLOCK) key_out[qq + s_goff[s_dig[qq]]] = s_key[qq];
    __syncthreads();
#pragma unroll
    for (int c = 0; c < OS_ITEMS; c++)
      if (q[c] != 0xFFFFFFFFu) s_rec[q[c]] = rec[c];
    __syncthreads();
    for (int qq = threadIdx.x; qq < tile_rows; qq += BLOCK) rec_out[qq + s_goff[s_dig[qq]]] = s_rec[qq];
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------ TopK narrowing
// histogram of one digit over candidate rows
__global__ __launch_bounds__(BLOCK) void k_select_hist(const uint64_t* __restrict__ word, const uint8_t* __restrict__ state, int64_t n, int shift, int bits,
                                                       unsigned long long* __restrict__ hist) {
  __shared__ unsigned int sh[256];
  const unsigned mask = (1u << bits) - 1u;
  sh[threadIdx.x] = 0;
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK)
    if (state[i] == 1) atomicAdd(&sh[(unsigned)(word[i] >> shift) & mask], 1u);
  __syncthreads();
  if (sh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)sh[threadIdx.x]);
}
// state: 0 out, 1 candidate, 2 selected.  digit < pivot -> selected, == pivot stays candidate, > out
__global__ __launch_bounds__(BLOCK) void k_select_apply(const uint64_t* __restrict__ word, uint8_t* __restrict__ state, int64_t n, int shift, int bits, unsigned pivot) {
  const unsigned mask = (1u << bits) - 1u;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    if (state[i] != 1) continue;
    unsigned d = (unsigned)(word[i] >> shift) & mask;
    state[i] = d < pivot ? 2 : (d == pivot ? 1 : 0);
  }
}
__global__ __launch_bounds__(BLOCK) void k_state_mask(const uint8_t* __restrict__ state, int64_t n, uint64_t* __restrict__ mask) {
  const int64_t n_words = (n + 63) >> 6;
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * BLOCK) >> 6;
  for (int64_t w = wave; w < n_words; w += n_waves) {
    int64_t i = (w << 6) + lane_id();
    uint64_t m = ballot64(i < n && state[i] != 0);
    if (lane_id() == 0) mask[w] = m;
  }
}
__global__ __launch_bounds__(BLOCK) void k_fill_bytes(uint8_t v, int64_t n, uint8_t* out) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) out[i] = v;
}
__global__ __launch_bounds__(BLOCK) void k_idx_to_i64(const uint32_t* __restrict__ idx, const int64_t* __restrict__ remap, int64_t n, int64_t* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    int64_t v = idx[i];
    out[i] = remap ? remap[v] : v;
  }
}
// row ids of the set bits of a mask, in order: ids[prefix + rank] = row
__global__ __launch_bounds__(BLOCK) void k_mask_to_ids(const uint64_t* __restrict__ mask, const uint64_t* __restrict__ prefix, int64_t n, int64_t* __restrict__ ids) {
  const int64_t n_words = (n + 63) >> 6;
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * BLOCK) >> 6;
  for (int64_t w = wave; w < n_words; w += n_waves) {
    uint64_t m = mask[w];
    if ((m >> lane_id()) & 1ull) ids[prefix[w] + mbcnt(m)] = (w << 6) + lane_id();
  }
}
// the same for a SPARSE mask (TopK's survivors: one row in hundreds): a thread per word — the word loads are coalesced and nearly every
// thread is done after one of them; a wave per word walked 286 words one after the other, two dependent loads each (0.11 ms for 150 M rows)
__global__ __launch_bounds__(BLOCK) void k_mask_to_ids_sparse(const uint64_t* __restrict__ mask, const uint64_t* __restrict__ prefix, int64_t n, int64_t* __restrict__ ids) {
  const int64_t n_words = (n + 63) >> 6;
  for (int64_t w = (int64_t)blockIdx.x * BLOCK + threadIdx.x; w < n_words; w += (int64_t)gridDim.x * BLOCK) {
    uint64_t m = mask[w];
    if (!m) continue;
    int64_t at = (int64_t)prefix[w];
    while (m) {
      ids[at++] = (w << 6) + __builtin_ctzll(m);
      m &= m - 1;
    }
  }
}
static void mask_to_ids(const uint64_t* mask, const uint64_t* prefix, int64_t n, int64_t set_bits, int64_t* ids) {
  const int64_t n_words = (n + 63) >> 6;
  if (set_bits * 16 < n) k_mask_to_ids_sparse<<<grid_for(n_words, BLOCK), BLOCK, 0, rt().stream>>>(mask, prefix, n, ids);
  else k_mask_to_ids<<<grid_for(n_words, BLOCK / WAVE), BLOCK, 0, rt().stream>>>(mask, prefix, n, ids);
}
__global__ void k_iota_u32(int64_t n, uint32_t* out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = (uint32_t)i;
}

// ------------------------------------------------------------------------------- host
struct Digit {
  int word, shift, bits;
  uint64_t div = 0;  // the digit is taken from key / div instead of the key (one-word keys; the top digits of the two-level sort)
};
struct SortedKeys {
  BufPtr w[MAX_KEY_WORDS];
  BufPtr idx;
  int nwords = 0;
};

// digits of a packed key of `total_bits` bits, least significant first; <= 8 bits each, none straddles a word
// the scatter passes walk their tiles in XCD-contiguous order (device.hpp xcd_tile; measured against tile t on XCD t % 8: - 4 to - 8 %,
// profiles/r5_sort_phases.md)
static int sort_xcd_map() { return 1; }
static int max_digit_bits() { return 8; }   // (digit widths and tile sizes were swept in round 2: profiles/r2_radix_sweep.md)
static std::vector<Digit> key_digits(int total_bits) {
  std::vector<Digit> ds;
  const int mb = max_digit_bits();
  for (int w = 0; w * 64 < total_bits; w++) {
    const int wbits = std::min(64, total_bits - w * 64);
    const int nd = (wbits + mb - 1) / mb;
    int pos = 0;
    for (int d = 0; d < nd; d++) {
      const int b = (wbits - pos + (nd - d) - 1) / (nd - d);  // spread the bits evenly over the passes
      ds.push_back({w, pos, b});
      pos += b;
    }
  }
  return ds;
}

// stable LSD radix sort of n (key, idx) elements over the given digits; returns the buffers holding the result
static SortedKeys radix_sort(SortedKeys in, int64_t n, const std::vector<Digit>& digits, bool want_ids) {
  const int xcd_map = sort_xcd_map();   // A/B knob of the XCD-contiguous tile order (device.hpp xcd_tile)
  Runtime& r = rt();
  if (n <= 1 || digits.empty()) return in;
  const int nwords = in.nwords;
  const int items = nwords == 1 ? 16 : rs_items(nwords);   // rows per thread (one key word: 16; measured profiles/r2_radix_sweep.md)
  const int64_t tile = (int64_t)BLOCK * items;
  const int64_t n_tiles = (n + tile - 1) / tile;
  SortedKeys cur = in, alt;
  alt.nwords = nwords;
  for (int wd = 0; wd < nwords; wd++) alt.w[wd] = make_buf((size_t)n * 8);
  if (want_ids) alt.idx = make_buf((size_t)n * 4);
  BufPtr counts = make_buf((size_t)256 * n_tiles * 4);
  BufPtr offsets = make_buf((size_t)(256 * n_tiles + 1) * 8);
  const int grid = (int)std::min<int64_t>(n_tiles, 256 * 8);
  for (const Digit& d : digits) {
    KeyWords ck{};
    SortBufs ob{};
    for (int wd = 0; wd < nwords; wd++) {
      ck.w[wd] = cur.w[wd]->as<uint64_t>();
      ob.w[wd] = alt.w[wd]->as<uint64_t>();
    }
    if (!alt.idx && want_ids) alt.idx = make_buf((size_t)n * 4);  // the input had implicit row ids (radix_sort_pairs)
    ob.idx = alt.idx ? alt.idx->as<uint32_t>() : nullptr;
    const int nb = 1 << d.bits;
    ProfileScope ps("radix_sort_pass", n * 8 + n * (nwords * 8 + 4) * 2);
    const DivBy dv = div_by(d.div);
    k_rs_hist<<<grid, BLOCK, 0, r.stream>>>(ck.w[d.word], n, dv, d.shift, d.bits, items, n_tiles, counts->as<uint32_t>());
    scan_u32(counts->as<uint32_t>(), (int64_t)nb * n_tiles, offsets->as<uint64_t>());
    const uint32_t* idx_in = cur.idx ? cur.idx->as<uint32_t>() : nullptr;
    switch (nwords) {
      case 1: k_rs_scatter2<1, 16><<<grid, BLOCK, 0, r.stream>>>(ck, idx_in, n, dv, d.word, d.shift, d.bits, n_tiles, offsets->as<uint64_t>(), ob, xcd_map); break;
      case 2: k_rs_scatter2<2, rs_items(2)><<<grid, BLOCK, 0, r.stream>>>(ck, idx_in, n, dv, d.word, d.shift, d.bits, n_tiles, offsets->as<uint64_t>(), ob, xcd_map); break;
      default: k_rs_scatter2<3, rs_items(3)><<<grid, BLOCK, 0, r.stream>>>(ck, idx_in, n, dv, d.word, d.shift, d.bits, n_tiles, offsets->as<uint64_t>(), ob, xcd_map); break;
    }
    DFGPU_HIP(hipGetLastError());
    std::swap(cur, alt);
  }
  return cur;
}

// ------------------------------------------------------------------------------ top digits in HBM, the rest in LDS
// A one-word key of T bits over n rows: the stable passes above run over the TOP bits only — as many 8-bit digits as it takes
// for a bucket (= rows sharing those bits) to hold ~2300 rows on average: 2 passes for 150 M rows instead of 6 — and every
// bucket, now contiguous and in stable order, is finished by ONE workgroup inside LDS (stable LSD passes over the remaining
// low bits with the same wave-private ranking; elements live in registers between passes, LDS holds one copy).  Whether every
// bucket fits is known exactly before the buckets are sorted (bucket bounds are read off the top-sorted keys); keys whose top
// bits are skewed beyond that take the all-HBM passes instead.
constexpr int LS_ITEMS = 16;
constexpr int LS_CAP = BLOCK * LS_ITEMS;  // 4096 elements per bucket in LDS

// bounds of the buckets of keys sorted by (key >> shift): starts / ends (both zero for an empty bucket) and the largest size
__global__ __launch_bounds__(BLOCK) void k_bucket_bounds(const uint64_t* __restrict__ key, int64_t n, DivBy width, uint32_t* __restrict__ starts,
                                                         uint32_t* __restrict__ ends) {
  // one load and one division per row: the neighbours' bucket numbers come from the adjacent lanes (the wave's first / last
  // lane divides the element before / after the wave's 64 rows itself)
  const unsigned lane = lane_id();
  const int64_t n_round = (n + BLOCK - 1) / BLOCK * BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n_round; i += (int64_t)gridDim.x * BLOCK) {
    const bool in = i < n;
    const uint64_t b = in ? div_apply(key[i], width) : ~0ull;
    uint64_t prev = __shfl_up(b, 1, 64), next = __shfl_down(b, 1, 64);
    if (lane == 0) prev = (in && i > 0) ? div_apply(key[i - 1], width) : ~0ull;
    if (lane == 63) next = (i + 1 < n) ? div_apply(key[i + 1], width) : ~0ull;
    if (!in) continue;
    if (i == 0 || prev != b) starts[b] = (uint32_t)i;
    if (i == n - 1 || next != b) ends[b] = (uint32_t)(i + 1);
  }
}
__global__ __launch_bounds__(BLOCK) void k_bucket_max(const uint32_t* __restrict__ starts, const uint32_t* __restrict__ ends, int64_t n_buckets, unsigned* __restrict__ max_size) {
  unsigned mx = 0;
  for (int64_t b = (int64_t)blockIdx.x * BLOCK + threadIdx.x; b < n_buckets; b += (int64_t)gridDim.x * BLOCK) {
    const unsigned sz = ends[b] - starts[b];
    mx = sz > mx ? sz : mx;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const unsigned o = __shfl_xor(mx, d, 64);
    mx = o > mx ? o : mx;
  }
  if (lane_id() == 0 && mx) atomicMax(max_size, mx);
}

// What the carried sort's last step writes instead of row ids: the key columns DECODED from the sorted packed key (mixed radix:
// digit_c = key / mult_c mod range_c, value = base +- digit — pack_key64 read backwards; integer / date columns without NULLs) and
// the record's fields.
struct SortEmit {
  int n_keys;
  void* key_dst[MAX_SORT_KEYS];
  int key_type[MAX_SORT_KEYS];
  int key_desc[MAX_SORT_KEYS];
  uint64_t key_base[MAX_SORT_KEYS], key_mult[MAX_SORT_KEYS];
  DivBy key_div[MAX_SORT_KEYS];
  const uint4* rec;     // records in the order the top passes left (id = position there)
  PackLayout fields;    // the record's fields and their output columns (dst)
};
__device__ __forceinline__ void sort_emit_row(const SortEmit& e, uint64_t full_key, const uint4& v, int64_t pos) {
  uint64_t rem = full_key;
  for (int c = 0; c < e.n_keys; c++) {   // most significant column first
    const uint64_t digit = div_apply(rem, e.key_div[c]);
    rem -= digit * e.key_mult[c];
    const uint64_t t = e.key_desc[c] ? e.key_base[c] - digit : e.key_base[c] + digit;
    switch (e.key_type[c]) {
      case DFGPU_INT32: case DFGPU_DATE32: reinterpret_cast<int32_t*>(e.key_dst[c])[pos] = (int32_t)((uint32_t)t ^ 0x80000000u); break;
      case DFGPU_UINT32: reinterpret_cast<uint32_t*>(e.key_dst[c])[pos] = (uint32_t)t; break;
      case DFGPU_INT64: reinterpret_cast<uint64_t*>(e.key_dst[c])[pos] = t ^ 0x8000000000000000ull; break;
      case DFGPU_UINT64: reinterpret_cast<uint64_t*>(e.key_dst[c])[pos] = t; break;
      default: reinterpret_cast<uint8_t*>(e.key_dst[c])[pos] = (uint8_t)t; break;
    }
  }
  if (e.fields.n > 0) {
    const uint64_t sl[2] = {((uint64_t)v.y << 32) | v.x, ((uint64_t)v.w << 32) | v.z};
    record_split<2>(e.fields, pos, sl);
  }
}
// ------------------------------------------------------------------------------ LSD carried sort: narrow keys (round 5)
// A sort whose packed key is NARROW (<= 32 bits: a date, a few small codes, dictionary indices ...) needs no bucket sort at all: a few
// stable onesweep passes over the ROWS finish it, the first reading the SOURCE columns and the last writing the OUTPUT columns.  A row
// travels as one record of 16, 24 or 32 bytes: every column the key does not cover, plus the 32-bit key itself in the record's last
// word (so no key array exists either; the key columns are decoded from that word at the end).  The record moves as a 16-byte slice and,
// beyond 16 bytes, a second slice of 8 or 16 bytes in an array of its own, staged through LDS one slice after the other.
// `reverse`: the first pass reads the input back to front — how a trailing key column that is strictly ascending IN INPUT ORDER and
// wanted DESC is honoured without being part of the key (sort_lsd_carried).  Same ranking, look-back and ticket as k_os_pass.
struct LsdIo {
  const uint4* in0;
  const void* in1;
  uint4* out0;
  void* out1;
};
__device__ __forceinline__ void lsd_emit_fields(const PackLayout& L, int slice, const uint4& v, int64_t pos) {
  for (int c = 0; c < L.n; c++) {
    const int o = L.offset[c];
    if ((o >> 4) != slice) continue;
    const int oo = o & 15;
    const uint32_t w32 = (oo >> 2) == 0 ? v.x : (oo >> 2) == 1 ? v.y : (oo >> 2) == 2 ? v.z : v.w;
    switch (L.width[c]) {
      case 16: reinterpret_cast<uint4*>(L.dst[c])[pos] = v; break;
      case 8: reinterpret_cast<uint2*>(L.dst[c])[pos] = oo ? make_uint2(v.z, v.w) : make_uint2(v.x, v.y); break;
      case 4: reinterpret_cast<uint32_t*>(L.dst[c])[pos] = w32; break;
      default: reinterpret_cast<uint8_t*>(L.dst[c])[pos] = (uint8_t)(w32 >> ((oo & 3) * 8)); break;
    }
  }
}
__device__ __forceinline__ void lsd_emit_keys(const SortEmit& emit, uint32_t key, int64_t pos) {
  uint64_t rem = key;
  for (int c = 0; c < emit.n_keys; c++) {   // most significant column first (sort_emit_row)
    const uint64_t digit = div_apply(rem, emit.key_div[c]);
    rem -= digit * emit.key_mult[c];
    const uint64_t v = emit.key_desc[c] ? emit.key_base[c] - digit : emit.key_base[c] + digit;
    switch (emit.key_type[c]) {
      case DFGPU_INT32: case DFGPU_DATE32: reinterpret_cast<int32_t*>(emit.key_dst[c])[pos] = (int32_t)((uint32_t)v ^ 0x80000000u); break;
      case DFGPU_UINT32: reinterpret_cast<uint32_t*>(emit.key_dst[c])[pos] = (uint32_t)v; break;
      case DFGPU_INT64: reinterpret_cast<uint64_t*>(emit.key_dst[c])[pos] = v ^ 0x8000000000000000ull; break;
      case DFGPU_UINT64: reinterpret_cast<uint64_t*>(emit.key_dst[c])[pos] = v; break;
      default: reinterpret_cast<uint8_t*>(emit.key_dst[c])[pos] = (uint8_t)v; break;
    }
  }
}
__device__ __forceinline__ uint32_t word_of(const uint4& v, int k) { return k == 0 ? v.x : k == 1 ? v.y : k == 2 ? v.z : v.w; }
// W2: bytes of the record's second slice (0, 8 or 16); key_off: byte offset of the key word inside the record
// AHEAD: the offsets of every unit of work are known before the pass starts (sort_lsd_carried: two passes over keys of at most 13
// bits — the digit-totals pass counts per input segment, jointly over both digits): a workgroup takes a UNIT — a stretch of consecutive
// rows [unit_lo, unit_hi) whose run of every digit starts at ahead_off[digit * ahead_stride + unit] — walks its tiles in order with the
// cursors in LDS, and waits for nobody: no tile states, no look-back (0.6-0.9 ms of a pass, profiles/r5_sort_phases.md).
struct LsdAhead {
  const int64_t* unit_lo;
  const int64_t* unit_hi;
  const unsigned long long* off;
  int64_t stride;
  int64_t n_units;
};
template <bool BUILD, bool EMIT, int W2, bool AHEAD = false>
__global__ __launch_bounds__(BLOCK, (BUILD ? (W2 == 0 ? 3 : 2) : (W2 == 0 ? 4 : 3))) void k_lsd_pass(LsdIo io, PackCols pc, PackLayout L, int key_off, int64_t n, int reverse, int shift, int bits,
                                                                       int64_t n_tiles, const unsigned long long* __restrict__ bin_base, uint32_t* __restrict__ tile_state,
                                                                       unsigned* __restrict__ ticket, SortEmit emit, LsdAhead ahead) {
  constexpr int NWAVE = BLOCK / WAVE;
  constexpr int NS = W2 == 0 ? 2 : 4;
  __shared__ __align__(16) uint4 s_rec[OS_TILE];   // the tile in digit order, one slice of its records at a time
  __shared__ uint8_t s_dig[OS_TILE];
  __shared__ uint16_t s_cnt[NWAVE][256];
  __shared__ uint16_t s_start[256];
  __shared__ unsigned int s_goff[256];
  __shared__ unsigned int s_wtot[NWAVE];
  __shared__ unsigned int s_tile;
  const unsigned mask = (1u << bits) - 1u;
  const int wave = threadIdx.x >> 6;
  const unsigned lane = lane_id();
  const int key_slice = key_off >> 4, key_word = (key_off & 15) >> 2;
  __shared__ unsigned int s_cur[256];   // AHEAD: where the unit's next row of every digit goes
  int64_t u_lo = 0, u_hi = 0;           // AHEAD: what is left of the current unit
  for (;;) {
    int64_t t = 0, lo;
    int tile_rows;
    if (AHEAD) {
      if (u_lo >= u_hi) {   // the next unit
        if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
        __syncthreads();
        const int64_t u = (int64_t)s_tile;
        if (u >= ahead.n_units) return;
        u_lo = ahead.unit_lo[u];
        u_hi = ahead.unit_hi[u];
        if (threadIdx.x <= mask) s_cur[threadIdx.x] = (unsigned)ahead.off[(int64_t)threadIdx.x * ahead.stride + u];
        __syncthreads();   // (s_tile is read before the next unit's ticket overwrites it)
        if (u_lo >= u_hi) continue;
      }
      lo = u_lo;
      tile_rows = (int)((u_hi - u_lo) < OS_TILE ? (u_hi - u_lo) : OS_TILE);
      u_lo += OS_TILE;
#pragma unroll
      for (int w = 0; w < NWAVE; w++) s_cnt[w][threadIdx.x] = 0;
      __syncthreads();
    } else {
      if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
#pragma unroll
      for (int w = 0; w < NWAVE; w++) s_cnt[w][threadIdx.x] = 0;
      __syncthreads();
      t = (int64_t)s_tile;
      if (t >= n_tiles) return;
      lo = t * OS_TILE;
      tile_rows = (int)((n - lo) < OS_TILE ? (n - lo) : OS_TILE);
    }
    uint4 rec0[OS_ITEMS], rec1[OS_ITEMS];
    unsigned dig[OS_ITEMS], rank[OS_ITEMS];
    if (BUILD) {   // (column by column: tile_keys)
      uint32_t src[OS_ITEMS];   // (row numbers below 2^30)
#pragma unroll
      for (int c = 0; c < OS_ITEMS; c++) {
        const int j = (wave * OS_ITEMS + c) * WAVE + (int)lane;
        const int64_t row = lo + (j < tile_rows ? j : 0);
        src[c] = (uint32_t)(reverse ? n - 1 - row : row);   // (position `row` of the order the stable passes see)
      }
      uint32_t key[OS_ITEMS];
      tile_keys<OS_ITEMS, uint32_t>(pc, src, key);
      uint64_t sl[OS_ITEMS][NS];
      tile_records<OS_ITEMS, NS>(L, src, sl);
#pragma unroll
      for (int c = 0; c < OS_ITEMS; c++) {
        slot_or<NS>(sl[c], key_off >> 3, (uint64_t)key[c] << ((key_off & 4) * 8));
        rec0[c] = uint4{(unsigned)sl[c][0], (unsigned)(sl[c][0] >> 32), (unsigned)sl[c][1], (unsigned)(sl[c][1] >> 32)};
        rec1[c] = uint4{0u, 0u, 0u, 0u};
        if (W2) rec1[c] = uint4{(unsigned)sl[c][NS - 2], (unsigned)(sl[c][NS - 2] >> 32), (unsigned)sl[c][NS - 1], (unsigned)(sl[c][NS - 1] >> 32)};
      }
    }
#pragma unroll
    for (int c = 0; c < OS_ITEMS; c++) {   // all loads of the wave's segment in flight together
      const int j = (wave * OS_ITEMS + c) * WAVE + (int)lane;
      const int64_t row = lo + (j < tile_rows ? j : 0);
      if (!BUILD) {
        rec1[c] = uint4{0u, 0u, 0u, 0u};
        const uint4 v = io.in0[row];   // (component by component: a 16-byte struct copy kept the whole array in scratch memory)
        rec0[c] = uint4{v.x, v.y, v.z, v.w};
        if (W2 == 16) {
          const uint4 u = reinterpret_cast<const uint4*>(io.in1)[row];
          rec1[c] = uint4{u.x, u.y, u.z, u.w};
        } else if (W2 == 8) {
          const uint2 u = reinterpret_cast<const uint2*>(io.in1)[row];
          rec1[c] = uint4{u.x, u.y, 0u, 0u};
        }
      }
    }
#pragma unroll
    for (int c = 0; c < OS_ITEMS; c++) {
      const int j = (wave * OS_ITEMS + c) * WAVE + (int)lane;
      const bool in = j < tile_rows;
      const uint32_t key0 = word_of(rec0[c], key_word), key1 = word_of(rec1[c], key_word);
      const uint32_t key = key_slice ? key1 : key0;
      dig[c] = in ? ((key >> shift) & mask) : 0u;
      uint64_t peers = ballot64(in);
      for (int b = 0; b < bits; b++) {
        const uint64_t bal = ballot64((dig[c] >> b) & 1u);
        peers &= ((dig[c] >> b) & 1u) ? bal : ~bal;
      }
      const unsigned r_in_wave = mbcnt(peers);
      const unsigned base = s_cnt[wave][dig[c]];
      if (in && r_in_wave == 0) s_cnt[wave][dig[c]] = (uint16_t)(base + (unsigned)__popcll(peers));
      rank[c] = base + r_in_wave;
    }
    __syncthreads();
    unsigned run = 0;   // thread d: digit d's rows in this tile
    {
#pragma unroll
      for (int w = 0; w < NWAVE; w++) {
        const unsigned v = s_cnt[w][threadIdx.x];
        s_cnt[w][threadIdx.x] = (uint16_t)run;
        run += v;
      }
      if (!AHEAD && t > 0 && threadIdx.x <= mask) os_store(&tile_state[t * 256 + threadIdx.x], OS_AGG | run);
      const unsigned inc = wave_inclusive_sum<unsigned>(run);
      if (lane == 63) s_wtot[wave] = inc;
      __syncthreads();
      unsigned base = 0;
      for (int w = 0; w < wave; w++) base += s_wtot[w];
      s_start[threadIdx.x] = (uint16_t)(base + inc - run);
    }
    __syncthreads();
    unsigned q[OS_ITEMS];
#pragma unroll
    for (int c = 0; c < OS_ITEMS; c++) {
      const int j = (wave * OS_ITEMS + c) * WAVE + (int)lane;
      q[c] = 0xFFFFFFFFu;
      if (j < tile_rows) {
        q[c] = (unsigned)s_start[dig[c]] + (unsigned)s_cnt[wave][dig[c]] + rank[c];
        s_rec[q[c]] = rec0[c];
        s_dig[q[c]] = (uint8_t)dig[c];
      }
    }
    // ---- look-back (k_os_pass): thread d adds up digit d's counts over the tiles before this one until it meets an inclusive prefix
    if (threadIdx.x <= mask) {
      if (AHEAD) {
        s_goff[threadIdx.x] = s_cur[threadIdx.x] - (unsigned)s_start[threadIdx.x];
        s_cur[threadIdx.x] += run;
      } else {
        const unsigned excl = os_look_back(tile_state, t, threadIdx.x);
        os_store(&tile_state[t * 256 + threadIdx.x], OS_PFX | (excl + run));
        s_goff[threadIdx.x] = (unsigned)bin_base[threadIdx.x] + excl - (unsigned)s_start[threadIdx.x];
      }
    }
    __syncthreads();
    // ---- write-out, one slice after the other (EMIT: the slice's fields to their columns; the key columns decoded from the key word)
    for (int qq = threadIdx.x; qq < tile_rows; qq += BLOCK) {
      const unsigned pos = qq + s_goff[s_dig[qq]];
      const uint4 v = s_rec[qq];
      if (!EMIT) {
        io.out0[pos] = v;
      } else {
        lsd_emit_fields(emit.fields, 0, v, (int64_t)pos);
        if (key_slice == 0) lsd_emit_keys(emit, word_of(v, key_word), (int64_t)pos);
      }
    }
    if (W2) {
      __syncthreads();
#pragma unroll
      for (int c = 0; c < OS_ITEMS; c++)
        if (q[c] != 0xFFFFFFFFu) s_rec[q[c]] = rec1[c];
      __syncthreads();
      for (int qq = threadIdx.x; qq < tile_rows; qq += BLOCK) {
        const unsigned pos = qq + s_goff[s_dig[qq]];
        const uint4 v = s_rec[qq];
        if (!EMIT) {
          if (W2 == 16) reinterpret_cast<uint4*>(io.out1)[pos] = v;
          else reinterpret_cast<uint2*>(io.out1)[pos] = make_uint2(v.x, v.y);
        } else {
          lsd_emit_fields(emit.fields, 1, v, (int64_t)pos);
          if (key_slice == 1) lsd_emit_keys(emit, word_of(v, key_word), (int64_t)pos);
        }
      }
    }
    __syncthreads();
  }
}

// one workgroup per bucket: stable LSD sort of the bucket's (key, id) elements by the key's low `low_bits` bits, in LDS
template <typename LK, bool EMIT = false>  // the key inside its bucket: 32 bits when `width` allows (less LDS and registers: more buckets in flight per CU)
__global__ __launch_bounds__(BLOCK, (sizeof(LK) == 4 ? 4 : 3)) void k_local_sort(const uint64_t* __restrict__ key, const uint32_t* __restrict__ idx_in, const uint32_t* __restrict__ starts,
                                                      const uint32_t* __restrict__ ends, int64_t n_buckets, uint64_t width, int low_bits, uint32_t* __restrict__ idx_out,
                                                      SortEmit emit = SortEmit{}) {
  constexpr int NWAVE = BLOCK / WAVE;
  __shared__ LK s_key[LS_CAP];
  __shared__ uint32_t s_idx[LS_CAP];
  __shared__ unsigned int s_cnt[NWAVE][256];
  __shared__ unsigned int s_start[256];
  __shared__ unsigned int s_wtot[NWAVE];
  const int wave = threadIdx.x >> 6;
  const unsigned lane = lane_id();
  const int n_pass = (low_bits + 7) / 8;
  for (int64_t b = blockIdx.x; b < n_buckets; b += gridDim.x) {
    const uint32_t lo = starts[b];
    const int m = (int)(ends[b] - lo);
    if (m == 0) continue;
    LK k[LS_ITEMS];
    uint32_t id[LS_ITEMS];
    // position of item c of this lane: a wave owns a contiguous segment, so position order = (wave, item, lane) order; the
    // segments are sized for THIS bucket (items = ceil(m / 256) rows per thread), so the four waves share its rows evenly
    const int items = (m + BLOCK - 1) / BLOCK;
#pragma unroll
    for (int c = 0; c < LS_ITEMS; c++) {
      if (c >= items) continue;
      const int j = (wave * items + c) * WAVE + (int)lane;
      const int64_t src = (int64_t)lo + (j < m ? j : 0);
      k[c] = (LK)(key[src] - (uint64_t)b * width);  // the key inside its bucket: < width <= 2^low_bits
      id[c] = idx_in ? idx_in[src] : (uint32_t)src;
    }
    // (EMIT: asking for the records' lines here, ahead of the LDS passes — one word of each kept in a register until the end — was
    // measured and lost: sort_local_emit 5.4 -> 6.8 ms; the early loads are waited for at the first barrier)
    if (m > 1) {
      int pos = 0;
      for (int p = 0; p < n_pass; p++) {
        const int bits = (low_bits - pos + (n_pass - p) - 1) / (n_pass - p);  // spread the bits evenly over the passes
        const unsigned dmask = (1u << bits) - 1u;
#pragma unroll
        for (int w = 0; w < NWAVE; w++) s_cnt[w][threadIdx.x] = 0;
        __syncthreads();
        unsigned dig[LS_ITEMS], rank[LS_ITEMS];
#pragma unroll
        for (int c = 0; c < LS_ITEMS; c++) {
          if (c >= items) continue;
          const int j = (wave * items + c) * WAVE + (int)lane;
          const bool in = j < m;
          dig[c] = in ? ((unsigned)(k[c] >> pos) & dmask) : 0u;
          uint64_t peers = ballot64(in);
          for (int q = 0; q < bits; q++) {
            const uint64_t bal = ballot64((dig[c] >> q) & 1u);
            peers &= ((dig[c] >> q) & 1u) ? bal : ~bal;
          }
          const unsigned r_in_wave = mbcnt(peers);
          const unsigned base = s_cnt[wave][dig[c]];
          if (in && r_in_wave == 0) s_cnt[wave][dig[c]] = base + (unsigned)__popcll(peers);
          rank[c] = base + r_in_wave;
        }
        __syncthreads();
        {  // thread d: digit d's total, the waves' exclusive prefixes, and the exclusive scan over digits
          unsigned run = 0;
#pragma unroll
          for (int w = 0; w < NWAVE; w++) {
            const unsigned v = s_cnt[w][threadIdx.x];
            s_cnt[w][threadIdx.x] = run;
            run += v;
          }
          const unsigned inc = wave_inclusive_sum<unsigned>(run);
          if (lane == 63) s_wtot[wave] = inc;
          __syncthreads();
          unsigned base = 0;
          for (int w = 0; w < wave; w++) base += s_wtot[w];
          s_start[threadIdx.x] = base + inc - run;
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < LS_ITEMS; c++) {
          if (c >= items) continue;
          const int j = (wave * items + c) * WAVE + (int)lane;
          if (j < m) {
            const unsigned q = s_start[dig[c]] + s_cnt[wave][dig[c]] + rank[c];
            s_key[q] = k[c];
            s_idx[q] = id[c];
          }
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < LS_ITEMS; c++) {  // back into registers in position order
          if (c >= items) continue;
          const int j = (wave * items + c) * WAVE + (int)lane;
          if (j < m) {
            k[c] = s_key[j];
            id[c] = s_idx[j];
          }
        }
        __syncthreads();
        pos += bits;
      }
    }
    if (EMIT) {
      // (the records are fetched row by row between the stores.  Fetching a thread's sixteen first took the kernel to 256 VGPRs and one
      // wave per SIMD: 10.4 ms; four at a time: 5.6 ms against 5.4 — what bounds this step is a random line of HBM per row)
#pragma unroll
      for (int c = 0; c < LS_ITEMS; c++) {
        if (c >= items) continue;
        const int j = (wave * items + c) * WAVE + (int)lane;
        if (j >= m) continue;
        uint4 v = uint4{0u, 0u, 0u, 0u};
        if (emit.fields.n > 0) v = emit.rec[id[c]];
        sort_emit_row(emit, (uint64_t)b * width + (uint64_t)k[c], v, (int64_t)lo + j);
      }
      continue;
    }
#pragma unroll
    for (int c = 0; c < LS_ITEMS; c++) {
      if (c >= items) continue;
      const int j = (wave * items + c) * WAVE + (int)lane;
      if (j < m) idx_out[(int64_t)lo + j] = id[c];
    }
  }
}

static SortedKeys radix_sort(SortedKeys in, int64_t n, const std::vector<Digit>& digits, bool want_ids = true);

// ------------------------------------------------------------------------------ the k-th smallest of m one-word keys, on the device
// MSD radix select by ONE workgroup: eight 8-bit digits from the top, a 256-bin histogram in LDS per digit over the keys that still
// share the chosen prefix.  The TopK uses it twice: the c-th smallest of its 16 K samples is the limit of the marking pass (round 5
// copied the samples to the host and ran std::nth_element between two kernels), and the n_out-th smallest of the rows that pass
// narrows them again — exactly — to what one workgroup sorts in LDS.  m up to a few hundred thousand: the keys stay in L2.
constexpr int SELECT_THREADS = 1024;
__global__ __launch_bounds__(SELECT_THREADS) void k_select_kth(const uint64_t* __restrict__ keys, int64_t m, int64_t k /* 1-based */, int top_shift, uint64_t* __restrict__ out) {
  __shared__ unsigned s_hist[256];
  __shared__ unsigned s_wtot[4];
  __shared__ uint64_t s_prefix, s_mask;
  __shared__ long long s_k;
  if (threadIdx.x == 0) {
    s_prefix = 0;
    s_mask = 0;
    s_k = k < 1 ? 1 : (k > m ? m : k);
  }
  // (digits above `top_shift` are zero in every key: the packed key has total_bits bits)
  const int64_t m_round = (m + SELECT_THREADS - 1) / SELECT_THREADS * SELECT_THREADS;
  for (int shift = top_shift; shift >= 0; shift -= 8) {
    if (threadIdx.x < 256) s_hist[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t prefix = s_prefix, mask = s_mask;
    for (int64_t i = threadIdx.x; i < m_round; i += SELECT_THREADS) {
      const uint64_t v = i < m ? keys[i] : 0;
      bool act = i < m && (v & mask) == prefix;
      const unsigned bin = (unsigned)(v >> shift) & 255u;
      // the smallest keys of a table share their top digits: 150 K atomics on ONE LDS word took 0.3 ms a digit.  Two rounds of "the first
      // active lane's bin, counted once for the wave", then whoever is left adds its own
#pragma unroll
      for (int round = 0; round < 2; round++) {
        const uint64_t am = ballot64(act);
        if (am == 0) break;
        const int leader = __builtin_ctzll(am);
        const unsigned lb = __shfl(bin, leader, 64);
        const uint64_t same = ballot64(act && bin == lb);
        if ((int)lane_id() == leader) atomicAdd(&s_hist[lb], (unsigned)__popcll(same));
        act = act && bin != lb;
      }
      if (act) atomicAdd(&s_hist[bin], 1u);
    }
    __syncthreads();
    // the bin that holds the k-th key: an inclusive scan over the 256 bins by 256 threads (one thread walking the bins cost 8 us a digit)
    const long long need = s_k;
    __syncthreads();   // (everybody has read s_k / s_prefix / s_mask before the one thread below rewrites them)
    if (threadIdx.x < 256) {
      const unsigned h = s_hist[threadIdx.x];
      const unsigned inc_w = wave_inclusive_sum(h);
      if (lane_id() == 63) s_wtot[threadIdx.x >> 6] = inc_w;
    }
    __syncthreads();
    if (threadIdx.x < 256) {
      const unsigned h = s_hist[threadIdx.x];
      unsigned before = 0;
      for (int w = 0; w < (int)(threadIdx.x >> 6); w++) before += s_wtot[w];
      const unsigned inc_w = wave_inclusive_sum(h);
      const long long inc = (long long)before + inc_w, excl = inc - h;
      const bool last = threadIdx.x == 255;
      if ((excl < need && inc >= need) || (last && inc < need)) {   // (the second form cannot happen for 1 <= k <= m: a guard, not a case)
        s_k = need - excl;
        s_prefix = prefix | ((uint64_t)threadIdx.x << shift);
        s_mask = mask | (255ull << shift);
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = s_prefix;
}
__global__ __launch_bounds__(BLOCK) void k_strided_u64(const uint64_t* __restrict__ in, int64_t every, int64_t m, uint64_t* __restrict__ out) {
  const int64_t j = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (j < m) out[j] = in[j * every];
}
// (key, id) of the rows whose key is <= *limit, appended at a cursor (in no order: the LDS sort orders ties by id); the cursor counts
// every such row, also those beyond `cap`
__global__ __launch_bounds__(BLOCK) void k_take_le(const uint64_t* __restrict__ keys, int64_t m, const uint64_t* __restrict__ limit_p, unsigned* __restrict__ cursor, int cap,
                                                   uint64_t* __restrict__ out_keys, uint32_t* __restrict__ out_ids) {
  const uint64_t limit = *limit_p;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < m; i += (int64_t)gridDim.x * BLOCK) {
    const uint64_t v = keys[i];
    if (v <= limit) {
      const unsigned at = atomicAdd(cursor, 1u);
      if (at < (unsigned)cap) {
        out_keys[at] = v;
        out_ids[at] = (uint32_t)i;
      }
    }
  }
}

// ------------------------------------------------------------------------------ a few thousand rows: one workgroup, in LDS
// The rows a TopK narrows down to, Q1's four groups, any ORDER BY over a small result: the radix passes above are 5 launches per
// 8-bit digit (histogram, three scan kernels, scatter) whatever n is — 25 launches and 0.25 ms for the ~3000 survivors of Q3's TopK.
// One workgroup sorts up to SMALL_SORT_CAP (key words, row id) elements in LDS with a bitonic network; the id breaks ties,
// which makes the order total and the sort stable (sorts/sort.rs:894-914: lexsort_to_indices over the batch; ties by input order
// is what the LSD passes gave).
constexpr int SMALL_SORT_CAP = 4096;
constexpr int SMALL_SORT_THREADS = 1024;
template <int NW>
__global__ __launch_bounds__(SMALL_SORT_THREADS) void k_small_sort(KeyWords in, const uint32_t* __restrict__ idx_in, int n, int padded, uint32_t* __restrict__ idx_out) {
  extern __shared__ uint64_t s_mem[];
  uint64_t* s_key = s_mem;                                            // [NW][padded]
  uint32_t* s_pos = reinterpret_cast<uint32_t*>(s_mem + (size_t)NW * padded);   // [padded]
  for (int i = threadIdx.x; i < padded; i += SMALL_SORT_THREADS) {
#pragma unroll
    for (int w = 0; w < NW; w++) s_key[w * padded + i] = i < n ? in.w[w][i] : ~0ull;
    // the element's id — its row's position in the sort's input, unique — orders equal keys, whatever order the elements arrive in
    s_pos[i] = i < n ? (idx_in ? idx_in[i] : (uint32_t)i) : 0xFFFFFFFFu;
  }
  __syncthreads();
  for (int k = 2; k <= padded; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = threadIdx.x; t < (padded >> 1); t += SMALL_SORT_THREADS) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));   // the t-th index with bit j clear
        const int q = i | j;
        const bool up = (i & k) == 0;
        bool gt = false, decided = false;                       // element i > element q ?
#pragma unroll
        for (int w = NW - 1; w >= 0; w--) {
          const uint64_t a = s_key[w * padded + i], b = s_key[w * padded + q];
          if (!decided && a != b) {
            gt = a > b;
            decided = true;
          }
        }
        if (!decided) gt = s_pos[i] > s_pos[q];
        if (gt == up) {
#pragma unroll
          for (int w = 0; w < NW; w++) {
            const uint64_t a = s_key[w * padded + i];
            s_key[w * padded + i] = s_key[w * padded + q];
            s_key[w * padded + q] = a;
          }
          const uint32_t pa = s_pos[i];
          s_pos[i] = s_pos[q];
          s_pos[q] = pa;
        }
      }
      __syncthreads();
    }
  }
  for (int i = threadIdx.x; i < n; i += SMALL_SORT_THREADS) idx_out[i] = s_pos[i];
}
// row ids of up to SMALL_SORT_CAP packed keys in sorted order (stable)
static BufPtr small_sort_ids(const SortedKeys& sk, int64_t n) {
  Runtime& r = rt();
  int padded = 2;
  while (padded < n) padded <<= 1;
  BufPtr out = make_buf((size_t)std::max<int64_t>(n, 1) * 4);
  KeyWords kw{};
  for (int w = 0; w < sk.nwords; w++) kw.w[w] = sk.w[w]->as<uint64_t>();
  const uint32_t* idp = sk.idx ? sk.idx->as<uint32_t>() : nullptr;
  const size_t lds = (size_t)padded * (sk.nwords * 8 + 4);
  ProfileScope ps("sort_small", n * (sk.nwords * 8 + 8));
  auto launch = [&](auto kern) {
    DFGPU_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    kern<<<1, SMALL_SORT_THREADS, lds, r.stream>>>(kw, idp, (int)n, padded, out->as<uint32_t>());
  };
  switch (sk.nwords) {
    case 1: launch(k_small_sort<1>); break;
    case 2: launch(k_small_sort<2>); break;
    default: launch(k_small_sort<3>); break;
  }
  DFGPU_HIP(hipGetLastError());
  return out;
}

// row ids of a one-word key in sorted order by "top digits in HBM + buckets in LDS"; null when that does not apply (then
// `clobbered` tells whether the key buffer was used as scratch by the top passes and has to be packed again)
static BufPtr sorted_ids_local(const SortedKeys& sk, int64_t n, uint64_t key_space, bool& clobbered) {
  Runtime& r = rt();
  clobbered = false;
  // key_space = number of values the mixed-radix key can take (0: the key is not of that kind).  Buckets are key / width with
  // width = ceil(key_space / 2^top_bits): equal slices of the key space whatever its size, ~2300 rows each when keys spread evenly
  int top_bits = 0;
  while (top_bits < 32 && (n >> top_bits) > 2304) top_bits += 8;
  if (sk.nwords != 1 || n < 2 || n >= 0xFFFFFFFFll || key_space < 2 || (key_space >> top_bits) < 2) return nullptr;
  const int64_t n_buckets = (int64_t)1 << top_bits;
  const uint64_t width = (key_space + (uint64_t)n_buckets - 1) / (uint64_t)n_buckets;
  int low_bits = 0;
  while (low_bits < 64 && ((width - 1) >> low_bits)) low_bits++;
  if (low_bits == 0) return nullptr;
  SortedKeys cur = sk;
  if (top_bits) {
    std::vector<Digit> top;
    for (int pos = 0; pos < top_bits; pos += 8) top.push_back({0, pos, std::min(8, top_bits - pos), width});
    cur = radix_sort(sk, n, top);
    clobbered = true;
  }
  BufPtr starts = make_zero_buf((size_t)n_buckets * 4), ends = make_zero_buf((size_t)n_buckets * 4), mx = make_zero_buf(4);
  {
    ProfileScope ps("sort_bucket_bounds", n * 8);
    k_bucket_bounds<<<grid_for(n, BLOCK), BLOCK, 0, r.stream>>>(cur.w[0]->as<uint64_t>(), n, div_by(width), starts->as<uint32_t>(), ends->as<uint32_t>());
    k_bucket_max<<<grid_for(n_buckets, BLOCK), BLOCK, 0, r.stream>>>(starts->as<uint32_t>(), ends->as<uint32_t>(), n_buckets, mx->as<unsigned>());
  }
  unsigned largest = 0;
  d2h(&largest, mx->ptr, 4);
  if (largest > (unsigned)LS_CAP) return nullptr;  // skewed keys: the caller finishes with the all-HBM passes
  BufPtr out = make_buf((size_t)n * 4);
  ProfileScope ps("sort_local_buckets", n * 16);
  const unsigned lg = (unsigned)std::min<int64_t>(n_buckets, (int64_t)r.num_cus * 16);
  const uint32_t* idp = cur.idx ? cur.idx->as<uint32_t>() : nullptr;
  if (low_bits <= 32) k_local_sort<uint32_t><<<lg, BLOCK, 0, r.stream>>>(cur.w[0]->as<uint64_t>(), idp, starts->as<uint32_t>(), ends->as<uint32_t>(), n_buckets, width, low_bits, out->as<uint32_t>());
  else k_local_sort<uint64_t><<<lg, BLOCK, 0, r.stream>>>(cur.w[0]->as<uint64_t>(), idp, starts->as<uint32_t>(), ends->as<uint32_t>(), n_buckets, width, low_bits, out->as<uint32_t>());
  DFGPU_HIP(hipGetLastError());
  return out;
}

// (key, row id) pairs sorted by bits [lo_bit, lo_bit + nbits) of the key — the radix partitioning of the LDS hash join
// (radix_join.hip): nbits / 8 stable passes.  `idx` may be null on entry: the row id of pair i is then i.
void radix_sort_pairs(BufPtr& key, BufPtr& idx, int64_t n, int lo_bit, int nbits) {
  if (n <= 1 || nbits <= 0) {
    if (!idx) {
      idx = make_buf((size_t)std::max<int64_t>(n, 1) * 4);
      if (n) k_iota_u32<<<grid_for(n, BLOCK), BLOCK, 0, rt().stream>>>(n, idx->as<uint32_t>());
    }
    return;
  }
  std::vector<Digit> digits;
  // 6-bit digits: a 64-way scatter of a 4096-row tile writes 64-row runs; 256-way passes cost 2x per pass (profiles/r2_radix_sweep.md)
  const int mb = 6;
  const int nd = (nbits + mb - 1) / mb;
  int pos = lo_bit;
  for (int d = 0; d < nd; d++) {
    const int b = (lo_bit + nbits - pos + (nd - d) - 1) / (nd - d);
    digits.push_back({0, pos, b});
    pos += b;
  }
  SortedKeys in;
  in.nwords = 1;
  in.w[0] = key;
  in.idx = idx;
  SortedKeys out = radix_sort(in, n, digits);
  key = out.w[0];
  idx = out.idx;
}

// keys alone grouped (stably) by bits [lo_bit, lo_bit + nbits) of their value, ONE pass per <= 6 bits, no row ids: what the hash join
// does to keys that arrive in no order before it looks them up / sets their bits (join.hip)
void radix_group_keys(BufPtr& key, int64_t n, int lo_bit, int nbits) {
  if (n <= 1 || nbits <= 0) return;
  std::vector<Digit> digits;
  const int mb = 6, nd = (nbits + mb - 1) / mb;
  int pos = lo_bit;
  for (int d = 0; d < nd; d++) {
    const int b = (lo_bit + nbits - pos + (nd - d) - 1) / (nd - d);
    digits.push_back({0, pos, b});
    pos += b;
  }
  SortedKeys in;
  in.nwords = 1;
  in.w[0] = key;
  key = radix_sort(in, n, digits, /*want_ids=*/false).w[0];
}

static int bits_for(u128 range) {
  int b = 0;
  while (range) {
    b++;
    range >>= 1;
  }
  return b;
}

static Table sort_table(const Table& in, const std::vector<int>& key_cols, const uint8_t* desc, const uint8_t* nulls_first, int64_t fetch);
// the rows of `in` in ascending order of the given key columns (stable); for the library's own use (strings.hip orders the distinct
// strings of a dictionary by their prefix words)
Table sort_table_ascending(const Table& in, const std::vector<int>& key_cols) {
  const std::vector<uint8_t> zeros(key_cols.size(), 0);
  return sort_table(in, key_cols, zeros.data(), zeros.data(), -1);
}
__global__ __launch_bounds__(BLOCK) void k_build_records16(PackLayout L, int64_t n, uint4* __restrict__ rec) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    uint64_t sl[2];
    record_build<2>(L, i, sl);
    rec[i] = uint4{(unsigned)sl[0], (unsigned)(sl[0] >> 32), (unsigned)sl[1], (unsigned)(sl[1] >> 32)};
  }
}
// the carried sort's last step: bucket bounds off the top-sorted keys, then one workgroup per bucket sorts it in LDS and writes the OUTPUT
// (key columns decoded from the sorted key, the record's fields from the records).  false = a bucket is too large for LDS (skewed keys).
static bool carried_emit(const Table& in, const std::vector<int>& key_cols, const PackCols& pc, const std::vector<int>& payload, const std::vector<int>& order,
                         const PackLayout& L, const BufPtr& cur_key, const BufPtr& cur_rec, const BufPtr& cur_idx, int64_t n, int64_t n_buckets, uint64_t width,
                         int low_bits, Table& out) {
  Runtime& r = rt();
  BufPtr starts = make_zero_buf((size_t)n_buckets * 4), ends = make_zero_buf((size_t)n_buckets * 4), mx = make_zero_buf(4);
  {
    ProfileScope ps("sort_bucket_bounds", n * 8);
    k_bucket_bounds<<<grid_for(n, BLOCK), BLOCK, 0, r.stream>>>(cur_key->as<uint64_t>(), n, div_by(width), starts->as<uint32_t>(), ends->as<uint32_t>());
    k_bucket_max<<<grid_for(n_buckets, BLOCK), BLOCK, 0, r.stream>>>(starts->as<uint32_t>(), ends->as<uint32_t>(), n_buckets, mx->as<unsigned>());
  }
  unsigned largest = 0;
  d2h(&largest, mx->ptr, 4);
  if (largest > (unsigned)LS_CAP) return false;  // skewed keys: the caller's paths (the packed keys are intact)
  // the output columns and who writes them
  out.cols.assign(in.cols.size(), Column{});
  SortEmit e{};
  e.n_keys = pc.n;
  int out_bytes = 0;
  for (int k = 0; k < pc.n; k++) {
    const int c = key_cols[(size_t)k];
    if (!out.cols[(size_t)c].data) out.cols[(size_t)c] = alloc_like(in.cols[(size_t)c], n);
    e.key_dst[k] = out.cols[(size_t)c].data->ptr;
    e.key_type[k] = pc.c[k].type;
    e.key_desc[k] = pc.c[k].desc;
    e.key_base[k] = pc.c[k].base_lo;
    e.key_mult[k] = pc.c[k].mult;
    e.key_div[k] = div_by(pc.c[k].mult);
    out_bytes += type_width(in.cols[(size_t)c].field.type);
  }
  e.rec = cur_rec->as<uint4>();
  e.fields = L;
  for (int q = 0; q < L.n; q++) {
    const int c = payload[(size_t)order[(size_t)q]];
    out.cols[(size_t)c] = alloc_like(in.cols[(size_t)c], n);
    e.fields.dst[q] = out.cols[(size_t)c].data->ptr;
    out_bytes += L.width[q];
  }
  {
    ProfileScope ps("sort_local_emit", n * (int64_t)(8 + 16 + out_bytes));
    const unsigned lg = (unsigned)std::min<int64_t>(n_buckets, (int64_t)r.num_cus * 16);
    const uint32_t* idp = cur_idx ? cur_idx->as<uint32_t>() : nullptr;   // (null: a row's id is its position — in the top passes' order, or of a one-bucket input)
    if (low_bits <= 32) k_local_sort<uint32_t, true><<<lg, BLOCK, 0, r.stream>>>(cur_key->as<uint64_t>(), idp, starts->as<uint32_t>(), ends->as<uint32_t>(), n_buckets, width, low_bits, nullptr, e);
    else k_local_sort<uint64_t, true><<<lg, BLOCK, 0, r.stream>>>(cur_key->as<uint64_t>(), idp, starts->as<uint32_t>(), ends->as<uint32_t>(), n_buckets, width, low_bits, nullptr, e);
    DFGPU_HIP(hipGetLastError());
  }
  DFGPU_HIP(hipStreamSynchronize(r.stream));
  return true;
}

// the carried sort (see sort_table); false = does not apply (nothing was touched: `keys` are intact)
static bool sort_carried(const Table& in, const std::vector<int>& key_cols, const PackCols& pc, const BufPtr& keys, int64_t n, uint64_t key_space, Table& out,
                         bool& clobbered) {
  Runtime& r = rt();
  clobbered = false;
  // DFGPU_SORT_CARRIED: 0 = off; passes = the records travel through the top passes (measured: the passes lose more than the take
  // they save, 11.1 ms against 9.9 for 150 M orders); default = row ids through the passes as before, records taken by row id INSIDE the bucket sort
  const std::string mode_s = option_str("sort.carried", "");
  const char* mode_env = mode_s.empty() ? nullptr : mode_s.c_str();
  const bool off = mode_env && mode_env[0] == '0';
  const bool through_passes = mode_env && mode_env[0] == 'p';
  const int64_t min_rows = option_int("sort.carried_min_rows", policy().rows_worth_a_pass());   // (below that the take's lines are cache hits)
  if (off || n < min_rows || n < 2 || n >= 0xFFFFFFFFll || key_space < 2) return false;
  // every key column can be read back from the packed key
  for (int k = 0; k < pc.n; k++) {
    const PackCol& c = pc.c[k];
    const bool int_like = c.type == DFGPU_INT32 || c.type == DFGPU_DATE32 || c.type == DFGPU_UINT32 || c.type == DFGPU_INT64 || c.type == DFGPU_UINT64 || c.type == DFGPU_UINT8;
    if (c.valid || c.has_null_bit || !int_like || c.range == 0 || c.mult == 0) return false;
  }
  // the other columns: one 16-byte record
  std::vector<int> payload;
  for (int c = 0; c < (int)in.cols.size(); c++)
    if (std::find(key_cols.begin(), key_cols.end(), c) == key_cols.end()) payload.push_back(c);
  PackLayout L{};
  int R = 0;
  std::vector<int> order;
  if (payload.empty() || !plan_record_layout(in, payload, L, R, order) || R != 16) return false;
  // buckets as sorted_ids_local cuts them
  int top_bits = 0;
  while (top_bits < 32 && (n >> top_bits) > 2304) top_bits += 8;
  if ((key_space >> top_bits) < 2) return false;
  const int64_t n_buckets = (int64_t)1 << top_bits;
  const uint64_t width = (key_space + (uint64_t)n_buckets - 1) / (uint64_t)n_buckets;
  int low_bits = 0;
  while (low_bits < 64 && ((width - 1) >> low_bits)) low_bits++;
  if (low_bits == 0) return false;
  constexpr int ITEMS = 8;   // rows per thread of the record-carrying pass (4: 7.06 ms, 8: 6.75 ms, 16: 9.03 ms for the two passes over 150 M orders)
  const int64_t tile = (int64_t)BLOCK * ITEMS, n_tiles = (n + tile - 1) / tile;
  BufPtr cur_key = keys, cur_rec, cur_idx;
  if (!through_passes) {
    // records in the input's row order (one streaming pass), row ids through the top passes
    cur_rec = make_buf((size_t)n * 16 + 64);
    {
      int payload_bytes = 0;
      for (int q = 0; q < L.n; q++) payload_bytes += L.width[q];
      ProfileScope ps("sort_build_records", n * (int64_t)(payload_bytes + 16));
      k_build_records16<<<grid_for(n, BLOCK), BLOCK, 0, r.stream>>>(L, n, cur_rec->as<uint4>());
      DFGPU_HIP(hipGetLastError());
    }
    if (top_bits) {
      SortedKeys sk0;
      sk0.nwords = 1;
      sk0.w[0] = keys;
      std::vector<Digit> top;
      for (int pos = 0; pos < top_bits; pos += 8) top.push_back({0, pos, std::min(8, top_bits - pos), width});
      SortedKeys cur = radix_sort(sk0, n, top);
      clobbered = true;   // (the key buffer was scratch of the passes)
      cur_key = cur.w[0];
      cur_idx = cur.idx;
    }
  } else if (top_bits) {
    BufPtr counts = make_buf((size_t)256 * n_tiles * 4), offsets = make_buf((size_t)(256 * n_tiles + 1) * 8);
    BufPtr key_a = make_buf((size_t)n * 8), key_b = top_bits > 8 ? make_buf((size_t)n * 8) : nullptr;
    BufPtr rec_a = make_buf((size_t)n * 16), rec_b = top_bits > 8 ? make_buf((size_t)n * 16) : nullptr;
    const int grid = (int)std::min<int64_t>(n_tiles, 256 * 8);
    const DivBy dv = div_by(width);
    int payload_bytes = 0;
    for (int q = 0; q < L.n; q++) payload_bytes += L.width[q];
    bool first = true;
    for (int pos = 0; pos < top_bits; pos += 8) {
      const int bits = std::min(8, top_bits - pos);
      BufPtr& dst_key = first || cur_key == key_b ? key_a : key_b;
      BufPtr& dst_rec = first || cur_rec == rec_b ? rec_a : rec_b;
      ProfileScope ps("sort_carried_pass", n * 8 + n * (int64_t)(8 + (first ? payload_bytes : 16) + 8 + 16));
      k_rs_hist<<<grid, BLOCK, 0, r.stream>>>(cur_key->as<uint64_t>(), n, dv, pos, bits, ITEMS, n_tiles, counts->as<uint32_t>());
      scan_u32(counts->as<uint32_t>(), (int64_t)(1 << bits) * n_tiles, offsets->as<uint64_t>());
      auto launch = [&](auto kern) {
        const size_t lds = (size_t)tile * (8 + 16 + 1);
        DFGPU_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        kern<<<grid, BLOCK, lds, r.stream>>>(cur_key->as<uint64_t>(), first ? nullptr : cur_rec->as<uint4>(), L, n, dv, pos, bits, n_tiles, offsets->as<uint64_t>(),
                                           dst_key->as<uint64_t>(), dst_rec->as<uint4>(), sort_xcd_map());
      };
      if (first) launch(k_rs_scatter_kv<ITEMS, true>);
      else launch(k_rs_scatter_kv<ITEMS, false>);
      DFGPU_HIP(hipGetLastError());
      cur_key = dst_key;
      cur_rec = dst_rec;
      first = false;
    }
  } else {
    cur_rec = make_buf((size_t)n * 16 + 64);   // (a table of one bucket: no top pass to build the records on the way)
    k_build_records16<<<grid_for(n, BLOCK), BLOCK, 0, r.stream>>>(L, n, cur_rec->as<uint4>());
    DFGPU_HIP(hipGetLastError());
  }
  return carried_emit(in, key_cols, pc, payload, order, L, cur_key, cur_rec, cur_idx, n, n_buckets, width, low_bits, out);
}

// The carried sort with onesweep top passes (round 5; the default form).  The first pass reads the SOURCE columns — the key is packed and
// the 16-byte record built on the fly — so no packed key array and no record array in input order are ever written: per row the sort
// moves 12 B (histogram of the key columns) + 24 + 24 B per top pass + 24 + 24 B for the bucket sort that writes the output.
// false = does not apply, or the keys are skewed beyond what a bucket holds (nothing was touched; the caller's older paths take over).
static bool sort_carried_onesweep(const Table& in, const std::vector<int>& key_cols, const PackCols& pc, int64_t n, uint64_t key_space, Table& out) {
  Runtime& r = rt();
  const std::string mode_s = option_str("sort.carried", "");   // A/B switch: unset / "o" = this form; "0", "p", "i" = the round-4 forms
  const char* mode_env = mode_s.empty() ? nullptr : mode_s.c_str();
  if (mode_env && mode_env[0] != 'o') return false;
  const int64_t min_rows = option_int("sort.carried_min_rows", policy().rows_worth_a_pass());
  if (n < min_rows || n < 2 || n >= ((int64_t)1 << 30) || key_space < 2) return false;
  for (int k = 0; k < pc.n; k++) {   // every key column can be read back from the packed key
    const PackCol& c = pc.c[k];
    const bool int_like = c.type == DFGPU_INT32 || c.type == DFGPU_DATE32 || c.type == DFGPU_UINT32 || c.type == DFGPU_INT64 || c.type == DFGPU_UINT64 || c.type == DFGPU_UINT8;
    if (c.valid || c.has_null_bit || !int_like || c.range == 0 || c.mult == 0) return false;
  }
  std::vector<int> payload;
  for (int c = 0; c < (int)in.cols.size(); c++)
    if (std::find(key_cols.begin(), key_cols.end(), c) == key_cols.end()) payload.push_back(c);
  PackLayout L{};
  int R = 0;
  std::vector<int> order;
  if (payload.empty() || !plan_record_layout(in, payload, L, R, order) || R != 16) return false;
  int top_bits = 0;
  while (top_bits < 32 && (n >> top_bits) > 2304) top_bits += 8;
  if (top_bits == 0 || top_bits > 8 * OS_MAX_PASSES || (key_space >> top_bits) < 2) return false;
  const int64_t n_buckets = (int64_t)1 << top_bits;
  const uint64_t width = (key_space + (uint64_t)n_buckets - 1) / (uint64_t)n_buckets;
  int low_bits = 0;
  while (low_bits < 64 && ((width - 1) >> low_bits)) low_bits++;
  if (low_bits == 0) return false;
  const DivBy dv = div_by(width);
  OsDigits dg{};
  for (int pos = 0; pos < top_bits; pos += 8) {
    dg.shift[dg.n] = pos;
    dg.bits[dg.n] = std::min(8, top_bits - pos);
    dg.n++;
  }
  const int64_t n_tiles = (n + OS_TILE - 1) / OS_TILE;
  BufPtr hist = make_zero_buf((size_t)OS_MAX_PASSES * 256 * 8), bases = make_buf((size_t)OS_MAX_PASSES * 256 * 8);
  BufPtr tickets = make_zero_buf((size_t)OS_MAX_PASSES * 4);
  BufPtr state = make_buf((size_t)n_tiles * 256 * 4);
  int64_t key_col_bytes = 0, payload_bytes = 0;
  for (int k = 0; k < pc.n; k++) key_col_bytes += n * (pc.c[k].type == DFGPU_UINT8 ? 1 : type_width(pc.c[k].type));
  for (int q = 0; q < L.n; q++) payload_bytes += L.width[q];
  {
    ProfileScope ps("sort_digit_totals", key_col_bytes);
    k_os_hist<true><<<r.num_cus * 8, BLOCK, 0, r.stream>>>(nullptr, pc, n, dv, dg, hist->as<unsigned long long>());
    k_os_bases<<<dg.n, BLOCK, 0, r.stream>>>(hist->as<unsigned long long>(), bases->as<unsigned long long>());
    DFGPU_HIP(hipGetLastError());
  }
  BufPtr key_a = make_buf((size_t)n * 8), rec_a = make_buf((size_t)n * 16 + 64);
  BufPtr key_b = dg.n > 1 ? make_buf((size_t)n * 8) : nullptr, rec_b = dg.n > 1 ? make_buf((size_t)n * 16 + 64) : nullptr;
  BufPtr cur_key, cur_rec;
  const int grid = (int)std::min<int64_t>(n_tiles, (int64_t)r.num_cus * 4);
  for (int p = 0; p < dg.n; p++) {
    const bool first = p == 0;
    BufPtr& dst_key = (first || cur_key == key_b) ? key_a : key_b;
    BufPtr& dst_rec = (first || cur_rec == rec_b) ? rec_a : rec_b;
    ProfileScope ps("sort_onesweep_pass", first ? key_col_bytes + n * (payload_bytes + 24) : n * 48);
    DFGPU_HIP(hipMemsetAsync(state->ptr, 0, (size_t)n_tiles * 256 * 4, r.stream));
    if (first)
      k_os_pass<true><<<grid, BLOCK, 0, r.stream>>>(nullptr, nullptr, pc, L, n, dv, dg.shift[p], dg.bits[p], n_tiles, bases->as<unsigned long long>() + p * 256,
                                                    state->as<uint32_t>(), tickets->as<unsigned>() + p, dst_key->as<uint64_t>(), dst_rec->as<uint4>());
    else
      k_os_pass<false><<<grid, BLOCK, 0, r.stream>>>(cur_key->as<uint64_t>(), cur_rec->as<uint4>(), pc, L, n, dv, dg.shift[p], dg.bits[p], n_tiles,
                                                     bases->as<unsigned long long>() + p * 256, state->as<uint32_t>(), tickets->as<unsigned>() + p, dst_key->as<uint64_t>(),
                                                     dst_rec->as<uint4>());
    DFGPU_HIP(hipGetLastError());
    cur_key = dst_key;
    cur_rec = dst_rec;
  }
  return carried_emit(in, key_cols, pc, payload, order, L, cur_key, cur_rec, nullptr, n, n_buckets, width, low_bits, out);
}

// The digit-totals pass of the AHEAD form: the input in COARSE segments of `cf` FINE segments of `fine` rows; per fine segment the counts
// of digit 1 (c1f[d1 * n_fine + f]: pass 1's units), per coarse segment the JOINT counts of both digits (joint[(d2 * B1 + d1) * n_coarse + g]:
// the rows of coarse segment g with digit 1 = d1 lie together in pass 1's output — stable passes — and are pass 2's units).  Exclusive
// scans of the two arrays as they lie ARE the output offsets of the two passes.  One workgroup of 1024 threads per coarse segment.
constexpr int LSD_JOINT_MAX = 8192;
__global__ __launch_bounds__(1024) void k_lsd_joint(PackCols pc, int64_t n, int reverse, int bits1, int bits2, int64_t fine, int cf, int64_t n_fine, int64_t n_coarse,
                                                     uint32_t* __restrict__ c1f, uint32_t* __restrict__ joint) {
  __shared__ unsigned s_joint[LSD_JOINT_MAX];
  __shared__ unsigned s_fine[256];
  const int B1 = 1 << bits1, B2 = 1 << bits2;
  const int64_t g = blockIdx.x;
  for (int i = threadIdx.x; i < B1 * B2; i += 1024) s_joint[i] = 0;
  for (int k = 0; k < cf; k++) {
    const int64_t f = g * cf + k;
    if (f >= n_fine) break;
    if ((int)threadIdx.x < B1) s_fine[threadIdx.x] = 0;
    __syncthreads();
    const int64_t lo = f * fine, hi = (lo + fine) < n ? (lo + fine) : n;
    for (int64_t base = lo; base < hi; base += 1024 * 4) {
      uint32_t src[4], key[4];
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const int64_t row = base + (int64_t)c * 1024 + threadIdx.x;
        const int64_t r = row < hi ? row : hi - 1;
        src[c] = (uint32_t)(reverse ? n - 1 - r : r);
      }
      tile_keys<4, uint32_t>(pc, src, key);
#pragma unroll
      for (int c = 0; c < 4; c++) {
        if (base + (int64_t)c * 1024 + threadIdx.x < hi) {
          const unsigned d1 = key[c] & (unsigned)(B1 - 1), d2 = (key[c] >> bits1) & (unsigned)(B2 - 1);
          atomicAdd(&s_fine[d1], 1u);
          atomicAdd(&s_joint[d2 * B1 + d1], 1u);
        }
      }
    }
    __syncthreads();
    if ((int)threadIdx.x < B1) c1f[(int64_t)threadIdx.x * n_fine + f] = s_fine[threadIdx.x];
    __syncthreads();
  }
  __syncthreads();
  for (int i = threadIdx.x; i < B1 * B2; i += 1024) joint[(int64_t)i * n_coarse + g] = s_joint[i];
}
// units of the two passes: pass 1 = the fine segments; pass 2 = (digit 1, coarse segment) cells, where pass 1 put them
__global__ void k_lsd_units(const unsigned long long* __restrict__ off1, int64_t n, int64_t fine, int cf, int64_t n_fine, int64_t n_coarse, int B1,
                            int64_t* __restrict__ lo1, int64_t* __restrict__ hi1, int64_t* __restrict__ lo2, int64_t* __restrict__ hi2) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_fine) {
    lo1[i] = i * fine;
    hi1[i] = (i + 1) * fine < n ? (i + 1) * fine : n;
  }
  if (i < (int64_t)B1 * n_coarse) {
    const int64_t d1 = i / n_coarse, g = i % n_coarse;
    lo2[i] = (int64_t)off1[d1 * n_fine + g * cf];
    hi2[i] = (g + 1) * cf < n_fine ? (int64_t)off1[d1 * n_fine + (g + 1) * cf] : (int64_t)off1[(d1 + 1) * n_fine];
  }
}

// The LSD carried sort (round 5): the packed key — after dropping what the input's own order already settles — fits 32 bits, so two to
// four stable passes over 16-32-byte records sort the table, the last one writing the output columns; no bucket sort, no key array.
// What the input's order settles: a key column that is STRICTLY ASCENDING in row order (a primary key the table was loaded by, a row
// number) decides every tie among the columns before it exactly as the input order does (ASC) or as the reversed input order does (DESC),
// and no column after it matters — so it and everything behind it leave the key, and the first pass reads the rows back to front for DESC.
// `ORDER BY o_orderdate, o_orderkey DESC` over orders is then a 12-bit sort.  false = does not apply (nothing was touched).
static bool sort_lsd_carried(const Table& in, const std::vector<int>& key_cols, const PackCols& pc, int64_t n, Table& out) {
  Runtime& r = rt();
  if (!option_on("sort.lsd", true)) return false;
  const int64_t min_rows = option_int("sort.carried_min_rows", policy().rows_worth_a_pass());
  if (n < min_rows || n < 2 || n >= ((int64_t)1 << 30)) return false;
  // the key columns that stay
  int kept = pc.n, reverse = 0;
  for (int k = 0; k < pc.n; k++) {
    const Column& c = in.cols[(size_t)key_cols[(size_t)k]];
    const bool stats_cheap = !c.validity && (c.field.type == DFGPU_INT32 || c.field.type == DFGPU_DATE32 || c.field.type == DFGPU_INT64);
    if (stats_cheap && column_stats(const_cast<Column&>(c), n).ascending) {
      kept = k;
      reverse = pc.c[k].desc ? 1 : 0;
      break;
    }
  }
  if (kept == 0) return false;   // (already in order, or its reverse: not this path's business)
  PackCols kc = pc;
  kc.n = kept;
  u128 product = 1;
  for (int k = kept - 1; k >= 0; k--) {
    PackCol& c = kc.c[k];
    const bool int_like = c.type == DFGPU_INT32 || c.type == DFGPU_DATE32 || c.type == DFGPU_UINT32 || c.type == DFGPU_INT64 || c.type == DFGPU_UINT64 || c.type == DFGPU_UINT8;
    if (c.valid || c.has_null_bit || !int_like || c.range == 0) return false;   // (the key columns are read back from the key)
    c.mult = (uint64_t)product;
    product *= (u128)c.range;
    if (product > ((u128)1 << 32)) return false;
  }
  if (product < 2) return false;
  const int total_bits = bits_for(product - 1);
  const int n_pass = (total_bits + 7) / 8;
  if (n_pass > OS_MAX_PASSES) return false;
  OsDigits dg{};
  for (int p = 0, pos = 0; p < n_pass; p++) {   // digits of (nearly) equal width: 12 bits = 6 + 6 (8 + 4, 4 + 8, 7 + 5 measured within 5 % of it)
    const int b = (total_bits - pos + (n_pass - p) - 1) / (n_pass - p);
    dg.shift[p] = pos;
    dg.bits[p] = b;
    pos += b;
    dg.n++;
  }
  // the record: every other column (the dropped key columns among them) + the key word
  std::vector<int> kept_cols(key_cols.begin(), key_cols.begin() + kept), payload;
  for (int c = 0; c < (int)in.cols.size(); c++)
    if (std::find(kept_cols.begin(), kept_cols.end(), c) == kept_cols.end()) payload.push_back(c);
  PackLayout L{};
  int R = 0, key_off = 0;
  std::vector<int> order;
  if (!payload.empty()) {
    if (!plan_record_layout(in, payload, L, R, order)) return false;
    int bytes = 0;
    for (int q = 0; q < L.n; q++) bytes = std::max(bytes, L.offset[q] + L.width[q]);
    key_off = (bytes + 3) / 4 * 4;
  }
  const int rec_bytes = (key_off + 4 + 7) / 8 * 8;
  if (rec_bytes > 32) return false;
  const int w2 = rec_bytes <= 16 ? 0 : rec_bytes - 16;
  for (int k = 0; k < kept; k++)   // a key column listed twice, or kept and dropped: leave it to the general paths
    for (int j = k + 1; j < pc.n; j++)
      if (key_cols[(size_t)k] == key_cols[(size_t)j]) return false;
  // output columns and who writes them
  out.cols.assign(in.cols.size(), Column{});
  SortEmit e{};
  e.n_keys = kept;
  int64_t key_col_bytes = 0, row_bytes = 0;
  for (int k = 0; k < kept; k++) {
    const int c = key_cols[(size_t)k];
    out.cols[(size_t)c] = alloc_like(in.cols[(size_t)c], n);
    e.key_dst[k] = out.cols[(size_t)c].data->ptr;
    e.key_type[k] = kc.c[k].type;
    e.key_desc[k] = kc.c[k].desc;
    e.key_base[k] = kc.c[k].base_lo;
    e.key_mult[k] = kc.c[k].mult;
    e.key_div[k] = div_by(kc.c[k].mult);
    const int w = kc.c[k].type == DFGPU_UINT8 ? 1 : type_width(kc.c[k].type);
    key_col_bytes += n * w;
    row_bytes += w;
  }
  e.fields = L;
  for (int q = 0; q < L.n; q++) {
    const int c = payload[(size_t)order[(size_t)q]];
    out.cols[(size_t)c] = alloc_like(in.cols[(size_t)c], n);
    e.fields.dst[q] = out.cols[(size_t)c].data->ptr;
    row_bytes += L.width[q];
  }
  const int64_t n_tiles = (n + OS_TILE - 1) / OS_TILE;
  BufPtr tickets = make_zero_buf((size_t)OS_MAX_PASSES * 4);
  // two passes over at most 13 bits: every unit's offsets ahead of time (k_lsd_joint), no look-back
  const bool ahead = dg.n == 2 && (1 << total_bits) <= LSD_JOINT_MAX && option_on("sort.lsd_ahead", true);
  BufPtr hist, bases, state, c1f, joint, off1, off2, units;
  LsdAhead ah[2] = {};
  if (ahead) {
    // fine segments of 4 tiles (pass 1's units: small enough that the workgroups' reads and writes stay near each other, 16 and 64 tiles
    // measured 0.1 / 0.5 ms slower), coarse segments of 30 of them (pass 2's units hold about cf * fine / B1 rows — 3840 at 64 digits —
    // and the joint histogram stays a few MB); a unit that would end right at a whole number of tiles is made a little smaller (8192 +- 90
    // rows is a fifth, nearly empty tile more than every third time)
    const int64_t fine = (int64_t)OS_TILE * 4;
    int cf = 30;
    {
      const int64_t mean = (int64_t)cf * fine / ((int64_t)1 << dg.bits[0]);
      if (mean % OS_TILE == 0 || mean % OS_TILE > OS_TILE - 256) cf -= 1;
    }
    const int64_t n_fine = (n + fine - 1) / fine, n_coarse = (n_fine + cf - 1) / cf;
    const int B1 = 1 << dg.bits[0], B2 = 1 << dg.bits[1];
    c1f = make_buf((size_t)B1 * n_fine * 4);
    joint = make_buf((size_t)B1 * B2 * n_coarse * 4);
    off1 = make_buf(((size_t)B1 * n_fine + 1) * 8);
    off2 = make_buf(((size_t)B1 * B2 * n_coarse + 1) * 8);
    units = make_buf((size_t)(2 * n_fine + 2 * B1 * n_coarse) * 8);
    {
      ProfileScope ps("sort_digit_totals", key_col_bytes);
      k_lsd_joint<<<(unsigned)n_coarse, 1024, 0, r.stream>>>(kc, n, reverse, dg.bits[0], dg.bits[1], fine, cf, n_fine, n_coarse, c1f->as<uint32_t>(), joint->as<uint32_t>());
      DFGPU_HIP(hipGetLastError());
    }
    scan_u32(c1f->as<uint32_t>(), (int64_t)B1 * n_fine, off1->as<uint64_t>());
    scan_u32(joint->as<uint32_t>(), (int64_t)B1 * B2 * n_coarse, off2->as<uint64_t>());
    int64_t* up = units->as<int64_t>();
    const int64_t nu = std::max<int64_t>(n_fine, (int64_t)B1 * n_coarse);
    k_lsd_units<<<(unsigned)((nu + 255) / 256), 256, 0, r.stream>>>(off1->as<unsigned long long>(), n, fine, cf, n_fine, n_coarse, B1, up, up + n_fine, up + 2 * n_fine,
                                                                    up + 2 * n_fine + (int64_t)B1 * n_coarse);
    DFGPU_HIP(hipGetLastError());
    ah[0] = LsdAhead{up, up + n_fine, off1->as<unsigned long long>(), n_fine, n_fine};
    ah[1] = LsdAhead{up + 2 * n_fine, up + 2 * n_fine + (int64_t)B1 * n_coarse, off2->as<unsigned long long>(), (int64_t)B1 * n_coarse, (int64_t)B1 * n_coarse};
  } else {
    hist = make_zero_buf((size_t)OS_MAX_PASSES * 256 * 8);
    bases = make_buf((size_t)OS_MAX_PASSES * 256 * 8);
    state = make_buf((size_t)n_tiles * 256 * 4);
    ProfileScope ps("sort_digit_totals", key_col_bytes);
    k_os_hist<true><<<r.num_cus * 8, BLOCK, 0, r.stream>>>(nullptr, kc, n, div_by(1), dg, hist->as<unsigned long long>());
    k_os_bases<<<dg.n, BLOCK, 0, r.stream>>>(hist->as<unsigned long long>(), bases->as<unsigned long long>());
    DFGPU_HIP(hipGetLastError());
  }
  BufPtr a0, a1, b0, b1;
  if (dg.n > 1) {
    a0 = make_buf((size_t)n * 16 + 64);
    if (w2) a1 = make_buf((size_t)n * w2 + 64);
  }
  if (dg.n > 2) {
    b0 = make_buf((size_t)n * 16 + 64);
    if (w2) b1 = make_buf((size_t)n * w2 + 64);
  }
  const int grid = (int)std::min<int64_t>(n_tiles, (int64_t)r.num_cus * 4);
  LsdIo io{};
  for (int p = 0; p < dg.n; p++) {
    const bool first = p == 0, last = p == dg.n - 1;
    if (!last) {
      const bool to_a = first || io.in0 == b0->as<uint4>();
      io.out0 = (to_a ? a0 : b0)->as<uint4>();
      io.out1 = w2 ? (to_a ? a1 : b1)->ptr : nullptr;
    }
    ProfileScope ps(last ? "sort_lsd_pass_out" : "sort_lsd_pass", n * (int64_t)((first ? row_bytes : rec_bytes) + (last ? row_bytes : rec_bytes)));
    if (!ahead) DFGPU_HIP(hipMemsetAsync(state->ptr, 0, (size_t)n_tiles * 256 * 4, r.stream));
    auto launch = [&](auto kern) {
      kern<<<grid, BLOCK, 0, r.stream>>>(io, kc, L, key_off, n, reverse, dg.shift[p], dg.bits[p], n_tiles, ahead ? nullptr : bases->as<unsigned long long>() + p * 256,
                                         ahead ? nullptr : state->as<uint32_t>(), tickets->as<unsigned>() + p, e, ah[ahead ? p : 0]);
    };
    auto pick = [&](auto w2c) {
      constexpr int W2 = decltype(w2c)::value;
      if (ahead) {   // (two passes: the first builds, the last emits)
        if (first) launch(k_lsd_pass<true, false, W2, true>);
        else launch(k_lsd_pass<false, true, W2, true>);
      } else if (first && last) launch(k_lsd_pass<true, true, W2>);
      else if (first) launch(k_lsd_pass<true, false, W2>);
      else if (last) launch(k_lsd_pass<false, true, W2>);
      else launch(k_lsd_pass<false, false, W2>);
    };
    if (w2 == 0) pick(std::integral_constant<int, 0>{});
    else if (w2 == 8) pick(std::integral_constant<int, 8>{});
    else pick(std::integral_constant<int, 16>{});
    DFGPU_HIP(hipGetLastError());
    io.in0 = io.out0;
    io.in1 = io.out1;
  }
  DFGPU_HIP(hipStreamSynchronize(r.stream));
  return true;
}

static Table sort_table(const Table& in, const std::vector<int>& key_cols, const uint8_t* desc, const uint8_t* nulls_first, int64_t fetch) {
  Runtime& r = rt();
  const int64_t n = in.nrows;
  DFGPU_CHECK(n < 0xFFFFFFFFll, "sort input exceeds u32 row ids");
  DFGPU_CHECK(!key_cols.empty() && (int)key_cols.size() <= MAX_SORT_KEYS, "bad number of sort keys");
  PackCols pc{};
  pc.n = (int)key_cols.size();
  int64_t key_col_bytes = 0;
  for (int k = 0; k < pc.n; k++) {
    DFGPU_CHECK(key_cols[k] >= 0 && key_cols[k] < (int)in.cols.size(), "sort key column out of range");
    const Column& c = in.cols[key_cols[k]];
    DFGPU_CHECK(c.field.type != DFGPU_BOOL, "Boolean sort keys are not supported on the GPU path");
    DFGPU_CHECK(!c.dict || c.dict->sorted, "ORDER BY on a dictionary-encoded string column needs a dictionary in ascending order (index order must be string order)");
    pc.c[k] = PackCol{c.ptr(), c.valid_words(), c.field.type, desc[k] != 0, nulls_first[k] != 0, c.validity != nullptr, 0, 0, 0, 0};
    key_col_bytes += n * (c.field.type == DFGPU_UINT8 ? 1 : type_width(c.field.type));
  }
  int64_t n_out = fetch >= 0 ? std::min(fetch, n) : n;

  Table out;
  out.nrows = n_out;
  if (n_out == 0) {
    for (auto& c : in.cols) out.cols.push_back(alloc_like(c, 0));
    return out;
  }
  // ---- value ranges -> field widths and positions (last key column = least significant)
  {
    // value ranges: signed integer key columns without NULLs take them from the column's cached statistics (column_stats, shared
    // with the join's map gating and the dynamic filter: computed once per immutable table); anything else is reduced here
    bool from_stats = true;
    for (int k = 0; k < pc.n; k++) {
      const Column& c = in.cols[key_cols[k]];
      from_stats &= !c.validity && (c.field.type == DFGPU_INT32 || c.field.type == DFGPU_DATE32 || c.field.type == DFGPU_INT64);
    }
    std::vector<u128> h;
    int grid = 1;
    if (from_stats) {
      h.resize((size_t)pc.n * 2);
      for (int k = 0; k < pc.n; k++) {
        const Column& c = in.cols[key_cols[k]];
        const ColStats st = column_stats(const_cast<Column&>(c), n);  // fills the column's shared cache; the rows do not change
        const bool w32 = c.field.type != DFGPU_INT64;
        h[(size_t)k * 2] = w32 ? (u128)((uint32_t)(int32_t)st.min ^ 0x80000000u) : (u128)((uint64_t)st.min ^ 0x8000000000000000ull);
        h[(size_t)k * 2 + 1] = w32 ? (u128)((uint32_t)(int32_t)st.max ^ 0x80000000u) : (u128)((uint64_t)st.max ^ 0x8000000000000000ull);
      }
    } else {
      grid = grid_for(n, BLOCK * 4);
      BufPtr rb = make_buf((size_t)grid * pc.n * 2 * sizeof(u128));
      {
        ProfileScope ps("sort_key_ranges", key_col_bytes);
        k_key_ranges<<<grid, BLOCK, 0, r.stream>>>(pc, n, rb->as<u128>());
        DFGPU_HIP(hipGetLastError());
      }
      h.resize((size_t)grid * pc.n * 2);
      d2h(h.data(), rb->ptr, h.size() * sizeof(u128));
    }
    int pos = 0;
    u128 product = 1;
    for (int k = pc.n - 1; k >= 0; k--) {
      u128 mn = ~(u128)0, mx = 0;
      for (int b = 0; b < grid; b++) {
        mn = std::min(mn, h[((size_t)b * pc.n + k) * 2]);
        mx = std::max(mx, h[((size_t)b * pc.n + k) * 2 + 1]);
      }
      PackCol& c = pc.c[k];
      if (mn > mx) mn = mx = 0;  // no valid row
      const u128 base = c.desc ? mx : mn;
      c.base_lo = (uint64_t)base;
      c.base_hi = (uint64_t)(base >> 64);
      c.bits = bits_for(mx - mn);
      c.shift = pos;
      pos += c.bits + (c.has_null_bit ? 1 : 0);
      // mixed radix (one-word keys): this column's digit range and the product of the ranges behind it
      const u128 span = mx - mn + 1;
      c.range = span > (u128)0x7FFFFFFFFFFFFFFFull ? 0 : (uint64_t)span;
      c.mult = product > (u128)0x7FFFFFFFFFFFFFFFull ? 0 : (uint64_t)product;
      if (c.range == 0 || c.type == DFGPU_DECIMAL128) product = ~(u128)0;
      else if (product <= ((u128)1 << 63)) product *= (u128)c.range * (c.has_null_bit ? 2 : 1);
    }
    DFGPU_CHECK(pos <= 64 * MAX_KEY_WORDS, "packed sort key longer than 192 bits is not supported on the GPU path");
    // `product` = number of distinct key values the mixed-radix packing can produce; it is used when that fits 63 bits
    const bool narrow = product <= ((u128)1 << 63);
    const uint64_t key_space = narrow ? (uint64_t)product : 0;
    const int total_bits = narrow ? bits_for(product - 1) : pos;
    const int nwords = std::max(1, (total_bits + 63) / 64);
    SortedKeys sk;
    sk.nwords = nwords;
    for (int wd = 0; wd < nwords; wd++) sk.w[wd] = make_buf((size_t)n * 8);
    const bool topk = fetch >= 0 && n > 4096 && n_out < n / 4 && total_bits > 0;
    auto pack_keys = [&]() {
    sk.idx.reset();
    if (narrow) {
      // row ids stay implicit until the first radix pass (radix_sort: idx_in == null means id = position); the TopK narrowing
      // reads the key words only and numbers its survivors afresh
      ProfileScope ps("sort_pack_keys", key_col_bytes + n * 8);
      k_pack_keys64<<<grid_for(n, BLOCK * 2), BLOCK, 0, r.stream>>>(pc, n, sk.w[0]->as<uint64_t>());
      DFGPU_HIP(hipGetLastError());
    } else {
      sk.idx = make_buf((size_t)n * 4);
      ProfileScope ps("sort_pack_keys", key_col_bytes + n * (nwords * 8 + 4));
      k_pack_keys<<<grid_for(n, BLOCK), BLOCK, 0, r.stream>>>(pc, n, nwords, sk.w[0]->as<uint64_t>(), nwords > 1 ? sk.w[1]->as<uint64_t>() : nullptr,
                                                             nwords > 2 ? sk.w[2]->as<uint64_t>() : nullptr, sk.idx->as<uint32_t>());
      DFGPU_HIP(hipGetLastError());
    }
    };
    const std::vector<Digit> digits = key_digits(total_bits);  // least significant first
    BufPtr remap;                                               // survivor position -> original row id (TopK path)
    int64_t m = n;
    // ---- TopK by a sampled limit (one-word keys): the keys of 64 K evenly spaced rows give a limit that a few thousand rows
    // stay under; ONE pass over the key columns marks them (no packed key array is ever written), and the rest of the sort
    // sees only those.  The k-th smallest key lies under the c-th smallest of S samples unless fewer than k of the n keys do —
    // c is chosen so that ~c n / S >> k rows are expected there; if the marked rows are fewer than k (never seen) or too many
    // (a few distinct keys: ties), the radix select below takes over.
    bool limited = false;
    if (topk && narrow && n >= (1 << 20)) {
      // (16 K samples: the host picks the c-th smallest of them — std::nth_element over 64 K took 0.3 ms of a 1.3 ms TopK; with 16 K the
      //  limit lets ~10 K rows per million through instead of ~2.5 K, which the survivors' sort does not notice: 1.30 -> 0.96 ms; 8 K: 0.94, 4 K: 1.06)
      const int64_t S = 1 << 14, every = n / S;
      const int64_t c = std::min<int64_t>(S, (n_out * S + n - 1) / n * 2 + 16);
      if (c < S / 4) {
        BufPtr d_sample = make_buf((size_t)S * 8);
        const int64_t n_words = (n + 63) / 64;
        BufPtr mask = make_buf(bitmap_bytes(n));
        BufPtr prefix = make_buf((size_t)(n_words + 1) * 8);
        const int top_shift = total_bits > 0 ? ((std::min(total_bits, 64) - 1) / 8) * 8 : 0;
        BufPtr d_limit = make_buf(16);   // [0] the c-th smallest sample, [1] the n_out-th smallest survivor
        uint64_t* limit = d_limit->as<uint64_t>();
        {
          // the c-th smallest of the samples, selected on the device (k_select_kth): nothing crosses PCIe before the marking pass
          ProfileScope ps("topk_limit_sample", S * 16);
          k_pack_keys64_rows<<<grid_for(S, BLOCK), BLOCK, 0, r.stream>>>(pc, nullptr, every, S, d_sample->as<uint64_t>());
          k_select_kth<<<1, SELECT_THREADS, 0, r.stream>>>(d_sample->as<uint64_t>(), S, c, top_shift, limit);
          DFGPU_HIP(hipGetLastError());
        }
        {
          ProfileScope ps("topk_limit_pass", key_col_bytes + n / 8);
          bool plain_keys = n < ((int64_t)1 << 32);
          for (int k = 0; k < pc.n; k++) {
            const PackCol& c = pc.c[k];
            plain_keys = plain_keys && !c.valid && !c.has_null_bit && c.range > 0 &&
                         (c.type == DFGPU_INT32 || c.type == DFGPU_DATE32 || c.type == DFGPU_UINT32 || c.type == DFGPU_INT64 || c.type == DFGPU_UINT64 || c.type == DFGPU_UINT8);
          }
          if (plain_keys) k_key_limit_mask_plain<<<grid_for(n_words, (BLOCK / WAVE) * 4), BLOCK, 0, r.stream>>>(pc, n, limit, mask->as<uint64_t>());
          else k_key_limit_mask<<<grid_for(n_words, BLOCK / WAVE), BLOCK, 0, r.stream>>>(pc, n, limit, mask->as<uint64_t>());
          DFGPU_HIP(hipGetLastError());
        }
        scan_mask_popcounts(mask->as<uint64_t>(), nullptr, n, prefix->as<uint64_t>());
        const int64_t under = (int64_t)read_u64(prefix->as<uint64_t>() + n_words);
        if (under >= n_out && under <= std::max<int64_t>(1 << 22, 64 * n_out)) {
          m = under;
          remap = make_buf((size_t)m * 8);
          mask_to_ids(mask->as<uint64_t>(), prefix->as<uint64_t>(), n, m, remap->as<int64_t>());
          SortedKeys sv;
          sv.nwords = 1;
          sv.w[0] = make_buf((size_t)m * 8);
          k_pack_keys64_rows<<<grid_for(m, BLOCK), BLOCK, 0, r.stream>>>(pc, remap->as<int64_t>(), 0, m, sv.w[0]->as<uint64_t>());
          sv.idx = make_buf((size_t)m * 4);
          k_iota_u32<<<grid_for(m, BLOCK), BLOCK, 0, r.stream>>>(m, sv.idx->as<uint32_t>());
          DFGPU_HIP(hipGetLastError());
          sk = sv;
          limited = true;
          // ---- second narrowing: the rows that passed are narrowed once more, by a limit sampled among THEM, down to what one workgroup
          // sorts in LDS (a few hundred rows); too few (never seen) or too many (ties) and the radix passes sort all who passed the first
          if (m > SMALL_SORT_CAP && m <= ((int64_t)1 << 22) && option_on("sort.topk_second_narrowing", true)) {
            BufPtr cursor = make_zero_buf(4);
            SortedKeys s2;
            s2.nwords = 1;
            s2.w[0] = make_buf((size_t)SMALL_SORT_CAP * 8);
            s2.idx = make_buf((size_t)SMALL_SORT_CAP * 4);
            {
              // (the limit: the c2-th smallest of up to 16 K evenly spaced survivors, as for the first pass — the exact n_out-th smallest of
              // all 150 K survivors would be six passes of one workgroup over them, 0.4 ms)
              ProfileScope ps("topk_second_narrowing", m * 8 * 2);
              const int64_t S2 = std::min<int64_t>(S, m), every2 = m / S2;
              const int64_t c2 = std::min<int64_t>(S2, (n_out * S2 + m - 1) / m * 2 + 16);
              k_strided_u64<<<grid_for(S2, BLOCK), BLOCK, 0, r.stream>>>(sv.w[0]->as<uint64_t>(), every2, S2, d_sample->as<uint64_t>());
              k_select_kth<<<1, SELECT_THREADS, 0, r.stream>>>(d_sample->as<uint64_t>(), S2, c2, top_shift, limit + 1);
              k_take_le<<<grid_for(m, BLOCK * 4), BLOCK, 0, r.stream>>>(sv.w[0]->as<uint64_t>(), m, limit + 1, cursor->as<unsigned>(), SMALL_SORT_CAP, s2.w[0]->as<uint64_t>(),
                                                                         s2.idx->as<uint32_t>());
              DFGPU_HIP(hipGetLastError());
            }
            unsigned taken = 0;
            d2h(&taken, cursor->ptr, 4);
            if ((int64_t)taken >= n_out && taken <= (unsigned)SMALL_SORT_CAP) {   // (else: ties beyond the LDS sort's capacity — the radix passes sort all survivors)
              sk = s2;
              m = taken;
            }
          }
        }
      }
    }
    // a full sort by one mixed-radix word whose other columns fit a 16-byte record: the onesweep carried sort reads the source columns itself
    // one workgroup sorts it in LDS (below) — unless the carried sorts are forced onto small tables (sort.carried_min_rows=0: their tests)
    const bool small_input = n <= SMALL_SORT_CAP && option_on("sort.small", true) && option_int("sort.carried_min_rows", 1) != 0;
    if (!small_input && !topk && !limited && n_out == n && sort_lsd_carried(in, key_cols, pc, n, out)) return out;
    if (!small_input && !topk && !limited && narrow && nwords == 1 && n_out == n && sort_carried_onesweep(in, key_cols, pc, n, key_space, out)) return out;
    if (!limited) pack_keys();
    if (topk && !limited) {
      // ---- TopK: MSD radix select narrows to the rows that can still be among the first k
      BufPtr state = make_buf((size_t)n + 64);
      k_fill_bytes<<<grid_for(n, BLOCK), BLOCK, 0, r.stream>>>(1, n, state->as<uint8_t>());
      BufPtr hist = make_buf(256 * 8);
      int64_t selected = 0, candidates = n;
      for (int di = (int)digits.size() - 1; di >= 0 && selected + candidates > std::max<int64_t>(4096, 2 * n_out); di--) {
        const Digit& d = digits[(size_t)di];
        DFGPU_HIP(hipMemsetAsync(hist->ptr, 0, 256 * 8, r.stream));
        ProfileScope ps("topk_select_pass", n * 9);
        k_select_hist<<<grid_for(n, BLOCK * 4), BLOCK, 0, r.stream>>>(sk.w[d.word]->as<uint64_t>(), state->as<uint8_t>(), n, d.shift, d.bits, hist->as<unsigned long long>());
        unsigned long long hh[256];
        d2h(hh, hist->ptr, sizeof hh);
        int64_t need = n_out - selected, acc = 0;
        unsigned pivot = (1u << d.bits) - 1u;
        for (unsigned v = 0; v < (1u << d.bits); v++) {
          if (acc + (int64_t)hh[v] >= need) { pivot = v; break; }
          acc += (int64_t)hh[v];
        }
        k_select_apply<<<grid_for(n, BLOCK), BLOCK, 0, r.stream>>>(sk.w[d.word]->as<uint64_t>(), state->as<uint8_t>(), n, d.shift, d.bits, pivot);
        selected += acc;
        candidates = (int64_t)hh[pivot];
      }
      // compact survivors (selected + remaining candidates), preserving input order
      const int64_t n_words = (n + 63) / 64;
      BufPtr mask = make_buf(bitmap_bytes(n));
      BufPtr prefix = make_buf((size_t)(n_words + 1) * 8);
      k_state_mask<<<grid_for(n_words, BLOCK / WAVE), BLOCK, 0, r.stream>>>(state->as<uint8_t>(), n, mask->as<uint64_t>());
      scan_mask_popcounts(mask->as<uint64_t>(), nullptr, n, prefix->as<uint64_t>());
      m = (int64_t)read_u64(prefix->as<uint64_t>() + n_words);
      remap = make_buf((size_t)(m ? m : 1) * 8);
      mask_to_ids(mask->as<uint64_t>(), prefix->as<uint64_t>(), n, m, remap->as<int64_t>());
      // survivors' packed keys, re-indexed 0..m-1
      SortedKeys sv;
      sv.nwords = nwords;
      dfgpu_field f64w{};
      f64w.type = DFGPU_UINT64;
      for (int wd = 0; wd < nwords; wd++) {
        Column kc;
        kc.field = f64w;
        kc.length = n;
        kc.data = sk.w[wd];
        sv.w[wd] = gather_column(kc, remap->as<int64_t>(), m, false).data;
      }
      // survivor ids 0..m-1 are positions into `remap`
      sv.idx = make_buf((size_t)(m ? m : 1) * 4);
      if (m) k_iota_u32<<<grid_for(m, BLOCK), BLOCK, 0, r.stream>>>(m, sv.idx->as<uint32_t>());
      sk = sv;
    }
    std::vector<int> allc(in.cols.size());
    for (size_t i = 0; i < allc.size(); i++) allc[i] = (int)i;
    // ---- carried sort (round 4): a full sort whose key is one mixed-radix word over integer / date columns without NULLs and whose
    // OTHER columns fit a 16-byte record.  The record travels with the key through the top passes (the first pass reads it from the
    // source columns — or, the default, row ids travel and the records are fetched by row id), the bucket sort in LDS writes the OUTPUT: key
    // columns decoded from the sorted key, the record's fields from the records.  No separate take: orders by (o_orderdate, o_orderkey DESC)
    // 9.9 ms against 11.9 (the take alone was 5.7: a random line per row, profiles/r3_sort_clustered.md).
    if (!small_input && !remap && narrow && nwords == 1 && !sk.idx && n_out == n && m == n) {
      bool keys_clobbered = false;
      if (sort_carried(in, key_cols, pc, sk.w[0], n, key_space, out, keys_clobbered)) return out;
      if (keys_clobbered) pack_keys();   // (skewed keys: the paths below start from the packed keys again)
    }
    // (round 3's 'clustered take' — one more stable pass that moved keys AND a 32-byte record per row into the order of the bucket
    // number's top digit, so that the final take read records from 18 MB groups — measured 17.9 ms against 12.4-13.2 and stayed
    // opt-in for a round (profiles/r3_sort_clustered.md); round 4 removed it: the carried sort above is what became of the idea)
    bool clobbered = false;
    BufPtr sorted_idx;
    if (m <= SMALL_SORT_CAP && total_bits > 0 && (small_input || m < n) && option_on("sort.small", true)) sorted_idx = small_sort_ids(sk, m);   // the whole input, or a TopK's survivors
    else if (!remap) sorted_idx = sorted_ids_local(sk, m, key_space, clobbered);
    if (!sorted_idx) {
      if (clobbered) pack_keys();
      SortedKeys sorted = radix_sort(sk, m, digits);
      if (!sorted.idx) {  // nothing to sort by (one row, or every key column constant): ids are the positions
        sorted.idx = make_buf((size_t)std::max<int64_t>(m, 1) * 4);
        if (m) k_iota_u32<<<grid_for(m, BLOCK), BLOCK, 0, r.stream>>>(m, sorted.idx->as<uint32_t>());
      }
      sorted_idx = sorted.idx;
    }
    if (remap) {  // TopK survivors: positions among the survivors -> row ids of the input
      BufPtr take_idx = make_buf((size_t)n_out * 8);
      k_idx_to_i64<<<grid_for(n_out, BLOCK), BLOCK, 0, r.stream>>>(sorted_idx->as<uint32_t>(), remap->as<int64_t>(), n_out, take_idx->as<int64_t>());
      DFGPU_HIP(hipGetLastError());
      out.cols = gather_columns(in, allc, take_idx->as<int64_t>(), n_out, false);
    } else {
      out.cols = gather_columns(in, allc, nullptr, n_out, false, sorted_idx->as<uint32_t>());
    }
  }
  DFGPU_HIP(hipStreamSynchronize(r.stream));
  return out;
}

}  // namespace dfgpu

using namespace dfgpu;

extern "C" int dfgpu_sort(dfgpu_table_t input, const int* key_cols, const uint8_t* descending, const uint8_t* nulls_first, int nkeys, int64_t fetch,
                          dfgpu_table_t* out) {
  return guarded([&] {
    require_init();
    const Table& in = *unwrap(input);
    std::vector<int> keys(key_cols, key_cols + nkeys);
    // Utf8 sort keys: interned with an ascending dictionary (index order = byte order of the strings = arrow's order for Utf8), the
    // indices are the key; the strings themselves travel as payload (the take of strings, strings.hip)
    Table work;
    bool interned = false;
    for (int k = 0; k < nkeys; k++) {
      DFGPU_CHECK(keys[(size_t)k] >= 0 && keys[(size_t)k] < (int)in.cols.size(), "sort key column out of range");
      const Column& c = in.cols[(size_t)keys[(size_t)k]];
      if (c.field.type == DFGPU_BOOL) {   // a Boolean sort key orders false < true: one byte per row, the Boolean column travels as payload
        if (!interned) work = in;
        interned = true;
        work.cols.push_back(bool_as_u8(c, in.nrows));
        keys[(size_t)k] = (int)work.cols.size() - 1;
        continue;
      }
      if (c.field.type != DFGPU_UTF8 || c.dict) continue;
      if (!interned) work = in;
      interned = true;
      work.cols.push_back(dictionary_encode(c, true));
      keys[(size_t)k] = (int)work.cols.size() - 1;
    }
    auto t = std::make_unique<Table>(sort_table(interned ? work : in, keys, descending, nulls_first, fetch));
    if (interned) t->cols.resize(in.cols.size());
    *out = wrap(t.release());
  });
}
