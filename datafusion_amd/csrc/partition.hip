// partition.hip — K1 + K10: row hashing and RepartitionExec(Hash).
//
// BatchPartitioner::partition_iter, Hash arm (physical-plan/src/repartition/mod.rs:1111-1150):
// partition = create_hashes(keys; REPARTITION seed 0) % n; each partition keeps input order.
// On device: one workgroup owns a 2048-row tile; per partition p a wave's membership is one ballot, so
//   pass 1  part id per row + per-(partition, tile) counts  (popcounts of the ballots),
//   scan    one exclusive scan over the partition-major count matrix gives every
//           (partition, tile) its output offset in a single partition-major buffer,
//   pass 2  scatter: the tile is staged in LDS sorted by partition (stable), every partition's
//           run is written contiguously.
// The n output tables are zero-copy slices of that one buffer per column (what the RCCL
// all-to-all sends from: contiguous per destination).
#include <cstdlib>

#include "device.hpp"
#include "internal.hpp"

namespace dfgpu {

Table compact_table(const Table& in, const std::vector<int>& cols, const uint64_t* mask, const uint64_t* mask_valid);

__global__ __launch_bounds__(BLOCK) void k_hash_rows(KeySet ks, int64_t n, uint64_t seed, int force_collisions, uint64_t* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    bool any_null;
    uint64_t h = hash_row(ks, i, seed, any_null);
    out[i] = force_collisions ? 0 : h;
  }
}

static KeySet keyset_of(const std::vector<const Column*>& keys) {
  KeySet ks{};
  DFGPU_CHECK((int)keys.size() <= MAX_KEYS && !keys.empty(), "bad number of key columns");
  ks.n = (int)keys.size();
  for (int i = 0; i < ks.n; i++) {
    DFGPU_CHECK(keys[i]->field.type != DFGPU_BOOL, "Boolean hash keys are not supported on the GPU path");
    ks.c[i] = KeyCol{keys[i]->ptr(), keys[i]->valid_words(), keys[i]->field.type, type_width(keys[i]->field.type)};
  }
  return ks;
}

void hash_columns(const std::vector<const Column*>& keys, int64_t n, uint64_t seed, uint64_t* out, bool force_collisions) {
  if (n == 0) return;
  KeySet ks = keyset_of(keys);
  int64_t bytes = n * 8;
  for (int i = 0; i < ks.n; i++) bytes += n * ks.c[i].width;
  ProfileScope ps("hash_rows", bytes);
  k_hash_rows<<<grid_for(n, BLOCK), BLOCK, 0, rt().stream>>>(ks, n, seed, force_collisions, out);
  DFGPU_HIP(hipGetLastError());
}

constexpr int MAX_PARTS = 64;
constexpr int PT_ITEMS = 4;                 // rows per thread (a multiple of 4).  Measured in one run, 600 M rows into 8 partitions: 4 -> 11.0 ms,
                                            // 8 -> 12.0 ms, 16 -> 17.5 ms: the 16 KB of LDS a 1024-row tile stages leave room for 8 workgroups per CU
constexpr int PT_TILE = BLOCK * PT_ITEMS;   // 1024 rows per workgroup tile

// pass 1: part[i] = partition of row i; counts[p * n_tiles + t] = rows of tile t routed to partition p.
// Per 64 rows the membership of a partition is one ballot; lane q keeps partition q's running count.
__global__ __launch_bounds__(BLOCK) void k_part_count(KeySet ks, int64_t n, int nparts, FastMod fm, int64_t n_tiles, uint8_t* __restrict__ part,
                                                      uint32_t* __restrict__ counts) {
  __shared__ unsigned int sh[MAX_PARTS];
  for (int64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    if (threadIdx.x < MAX_PARTS) sh[threadIdx.x] = 0;
    __syncthreads();
    const int64_t lo = t * PT_TILE;
    uint32_t mine = 0;
    // a thread takes 4 consecutive rows and stores their partition ids as ONE 32-bit word: per-row byte stores
    // made this kernel run at 1.6 TB/s (profiles/r1_ops_v4_traffic.md)
#pragma unroll
    for (int c = 0; c < PT_ITEMS / 4; c++) {
      const int64_t i0 = lo + ((int64_t)c * BLOCK + threadIdx.x) * 4;
      int p[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        p[k] = -1;
        if (i0 + k < n) {
          bool any_null;
          uint64_t h = hash_row(ks, i0 + k, SEED_REPARTITION, any_null);
          p[k] = (int)fastmod_u64(h, fm);
        }
      }
      if (i0 + 3 < n) {
        *reinterpret_cast<uint32_t*>(part + i0) = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
      } else {
#pragma unroll
        for (int k = 0; k < 4; k++)
          if (i0 + k < n) part[i0 + k] = (uint8_t)p[k];
      }
      for (int q = 0; q < nparts; q++) {
        uint32_t cnt = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) cnt += (uint32_t)__popcll(ballot64(p[k] == q));
        if ((int)lane_id() == q) mine += cnt;
      }
    }
    if ((int)lane_id() < nparts && mine) atomicAdd(&sh[lane_id()], mine);
    __syncthreads();
    if ((int)threadIdx.x < nparts) counts[(int64_t)threadIdx.x * n_tiles + t] = sh[threadIdx.x];
    __syncthreads();
  }
}

constexpr int PART_MAX_COLS = 12;
struct PartCols {
  const void* src[PART_MAX_COLS];
  void* dst[PART_MAX_COLS];
  int width[PART_MAX_COLS];
  int n;
};
template <typename T>
__device__ __forceinline__ void stage_column(const void* __restrict__ src, void* __restrict__ dst, void* s_val, const uint8_t* s_dig, const unsigned* s_start,
                                             const unsigned long long* s_goff, int64_t lo, int tile_rows, const unsigned (&q)[PT_ITEMS]) {
  T* sv = reinterpret_cast<T*>(s_val);
#pragma unroll
  for (int c = 0; c < PT_ITEMS; c++) {
    const int j = c * BLOCK + threadIdx.x;
    if (q[c] != 0xFFFFFFFFu) sv[q[c]] = reinterpret_cast<const T*>(src)[lo + j];
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < PT_ITEMS; c++) {
    const int qq = c * BLOCK + threadIdx.x;
    if (qq < tile_rows) {
      const unsigned p = s_dig[qq];
      // a partition's run of a tile is >= 1 KB of consecutive rows written once: non-temporal stores (the scatter 11.0-12.0 ->
      // 8.7-10.9 ms by box; non-temporal LOADS of the input made it slower)
      stream_store(reinterpret_cast<T*>(dst) + s_goff[p] + (unsigned)(qq - (int)s_start[p]), sv[qq]);
    }
  }
  __syncthreads();
}
// pass 2: stable scatter of one tile.  Every row's position inside the tile sorted by partition is computed once
// (wave64 ballot peer masks + a cross-wave prefix in LDS, as sort.hip's radix pass); each column is then staged
// through LDS in that order and every partition's run is written contiguously — 2048 / nparts rows per run
// instead of the ~8-row runs a 64-row wave scatters on its own (measured 33 % of HBM peak).
// MASKED: rows whose part id is 0xFF take no part (a predicate dropped them: partition_by_key_range under a row mask)
template <bool MASKED = false>
__global__ __launch_bounds__(BLOCK) void k_part_scatter(PartCols cols, const uint8_t* __restrict__ part, const uint64_t* __restrict__ prefix, int64_t n,
                                                        int nparts, int nbits, int64_t n_tiles) {
  __shared__ uint4 s_val[PT_TILE];
  __shared__ uint8_t s_dig[PT_TILE];
  __shared__ unsigned int s_wave[BLOCK / WAVE][MAX_PARTS];
  __shared__ unsigned int s_run[MAX_PARTS], s_start[MAX_PARTS];
  __shared__ unsigned long long s_goff[MAX_PARTS];
  const int wave = threadIdx.x >> 6;
  for (int64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const int64_t lo = t * PT_TILE;
    const int tile_rows = (int)((n - lo) < PT_TILE ? (n - lo) : PT_TILE);
    if (threadIdx.x < MAX_PARTS) {
      s_run[threadIdx.x] = 0;
      for (int w = 0; w < BLOCK / WAVE; w++) s_wave[w][threadIdx.x] = 0;
      if ((int)threadIdx.x < nparts) s_goff[threadIdx.x] = prefix[(int64_t)threadIdx.x * n_tiles + t];
    }
    __syncthreads();
    unsigned dig[PT_ITEMS], rank[PT_ITEMS], q[PT_ITEMS];
    bool inr[PT_ITEMS];
#pragma unroll
    for (int c = 0; c < PT_ITEMS; c++) {
      const int j = c * BLOCK + threadIdx.x;
      bool in = j < tile_rows;
      dig[c] = in ? part[lo + j] : 0u;
      if (MASKED && dig[c] == 0xFFu) {
        in = false;
        dig[c] = 0u;
      }
      inr[c] = in;
      uint64_t peers = ballot64(in);
      for (int b = 0; b < nbits; b++) {
        const uint64_t bal = ballot64((dig[c] >> b) & 1u);
        peers &= ((dig[c] >> b) & 1u) ? bal : ~bal;
      }
      const unsigned r_in_wave = mbcnt(peers);
      if (in && r_in_wave == 0) s_wave[wave][dig[c]] = (unsigned)__popcll(peers);
      __syncthreads();
      if (in) {
        unsigned r = s_run[dig[c]] + r_in_wave;
        for (int w = 0; w < wave; w++) r += s_wave[w][dig[c]];
        rank[c] = r;
      }
      __syncthreads();
      if (threadIdx.x < MAX_PARTS) {
        unsigned tot = 0;
        for (int w = 0; w < BLOCK / WAVE; w++) {
          tot += s_wave[w][threadIdx.x];
          s_wave[w][threadIdx.x] = 0;
        }
        s_run[threadIdx.x] += tot;
      }
      __syncthreads();
    }
    if (threadIdx.x < WAVE) {  // exclusive scan of the <= 64 partition counts: one wave
      const unsigned cnt = s_run[threadIdx.x];
      const unsigned inc = wave_inclusive_sum<unsigned>(cnt);
      s_start[threadIdx.x] = inc - cnt;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < PT_ITEMS; c++) {
      q[c] = 0xFFFFFFFFu;   // (a row that takes no part)
      if (inr[c]) {
        q[c] = s_start[dig[c]] + rank[c];
        s_dig[q[c]] = (uint8_t)dig[c];
      }
    }
    __syncthreads();
    // rows that leave the tile: all of them, or — MASKED — those that take part
    const int out_rows = MASKED ? (int)(s_start[nparts - 1] + s_run[nparts - 1]) : tile_rows;
    for (int c = 0; c < cols.n; c++) {
      switch (cols.width[c]) {
        case 16: stage_column<uint4>(cols.src[c], cols.dst[c], s_val, s_dig, s_start, s_goff, lo, out_rows, q); break;
        case 8: stage_column<uint64_t>(cols.src[c], cols.dst[c], s_val, s_dig, s_start, s_goff, lo, out_rows, q); break;
        case 4: stage_column<uint32_t>(cols.src[c], cols.dst[c], s_val, s_dig, s_start, s_goff, lo, out_rows, q); break;
        case 1: stage_column<uint8_t>(cols.src[c], cols.dst[c], s_val, s_dig, s_start, s_goff, lo, out_rows, q); break;
      }
    }
  }
}
// pass 1, second generation (round 2): a thread keeps the partition numbers of its 8 rows as 16-bit counter fields packed in
// registers — one shuffle reduction per wave and tile instead of 4 ballots + popcounts per partition and 4 rows — and the
// remainder h % nparts is Lemire's fastmod (device.hpp; the compiler's 64-bit remainder by a runtime value is a ~150-instruction
// loop): 2.47 -> 1.25 ms for 600 M rows.  (A second-generation scatter with wave-private ranking that recomputes the partition
// from the key instead of reading `part` was measured too: 10.6 ms against 10.2 ms — the scatter sits at 84 % of the copy
// ceiling and is not bound by its barriers; with its 8 loads per thread batched it dropped to 15.3 ms.  A SINGLE-PASS placement
// — no count pass, the tile hashes its keys and claims its range in every partition's slack-sized region with one atomic — was
// measured as well: 11.4-12.0 ms against 11.9-13.0 ms for the two passes in the same runs: the scatter with the hash folded in
// costs what both passes cost, so the deterministic two-pass placement stays.)
// KT: one key column of that type without NULLs — the 4 x (PT_ITEMS / 4) keys of a thread are loaded before any is hashed (hash_row
// walks the column list per row and waits on every load: device.hpp; 1.44 -> ms for 600 M Int64 keys); KT_ANY = any key set
template <int NPK, int KT = KT_ANY>  // nparts <= 4 * NPK
__global__ __launch_bounds__(BLOCK) void k_part_count2(KeySet ks, int64_t n, int nparts, FastMod fm, int64_t n_tiles, uint8_t* __restrict__ part,
                                                       uint32_t* __restrict__ counts) {
  __shared__ unsigned int s_cnt[BLOCK / WAVE][4 * NPK];
  const int wave = threadIdx.x >> 6;
  const unsigned lane = lane_id();
  for (int64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const int64_t lo = t * PT_TILE;
    uint64_t acc[NPK];
#pragma unroll
    for (int q = 0; q < NPK; q++) acc[q] = 0;
    uint64_t kv[PT_ITEMS];
    if (KT != KT_ANY) {
#pragma unroll
      for (int c = 0; c < PT_ITEMS / 4; c++) {
        const int64_t i0 = lo + ((int64_t)c * BLOCK + threadIdx.x) * 4;
#pragma unroll
        for (int k = 0; k < 4; k++) kv[c * 4 + k] = load_key<KT>(ks.c[0], i0 + k < n ? i0 + k : n - 1);
      }
    }
#pragma unroll
    for (int c = 0; c < PT_ITEMS / 4; c++) {
      const int64_t i0 = lo + ((int64_t)c * BLOCK + threadIdx.x) * 4;
      uint32_t packed = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        if (i0 + k < n) {
          bool any_null;
          const unsigned p = fastmod_u64(KT != KT_ANY ? hash_u64(kv[c * 4 + k], SEED_REPARTITION) : hash_row(ks, i0 + k, SEED_REPARTITION, any_null), fm);
          packed |= p << (8 * k);
          const uint64_t inc = 1ull << ((p & 3u) * 16);
#pragma unroll
          for (int q = 0; q < NPK; q++) acc[q] += (p >> 2) == (unsigned)q ? inc : 0ull;
        }
      }
      if (i0 + 3 < n) {
        *reinterpret_cast<uint32_t*>(part + i0) = packed;
      } else {
#pragma unroll
        for (int k = 0; k < 4; k++)
          if (i0 + k < n) part[i0 + k] = (uint8_t)(packed >> (8 * k));
      }
    }
#pragma unroll
    for (int q = 0; q < NPK; q++) acc[q] = wave_sum(acc[q]);  // <= 64 * 8 = 512 per field: no carry between fields
    if ((int)lane < nparts) {
      uint64_t v = 0;
#pragma unroll
      for (int q = 0; q < NPK; q++) v = (lane >> 2) == (unsigned)q ? acc[q] : v;
      s_cnt[wave][lane] = (unsigned)(v >> ((lane & 3u) * 16)) & 0xFFFFu;
    }
    __syncthreads();
    if ((int)threadIdx.x < nparts) {
      unsigned tot = 0;
#pragma unroll
      for (int w = 0; w < BLOCK / WAVE; w++) tot += s_cnt[w][threadIdx.x];
      counts[(int64_t)threadIdx.x * n_tiles + t] = tot;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------- rows grouped by the RANGE their key falls in
// part[i] = ((key[i] - kmin) >> shift) & mask (the caller guarantees < nparts <= 64) + the per-(partition, tile) counts k_part_scatter
// places rows by.  What the aggregate does to rows whose groups are too many for one workgroup's LDS: afterwards every
// partition's groups fit (aggregate.hip dense_accumulate_partitioned).
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_part_count_range(const T* __restrict__ key, int64_t n, long long kmin, int shift, unsigned mask, int nparts, int64_t n_tiles,
                                                            const uint64_t* __restrict__ row_mask, const uint64_t* __restrict__ row_mask_valid,
                                                            uint8_t* __restrict__ part, uint32_t* __restrict__ counts) {
  __shared__ unsigned int sh[MAX_PARTS];
  for (int64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    if (threadIdx.x < MAX_PARTS) sh[threadIdx.x] = 0;
    __syncthreads();
    const int64_t lo = t * PT_TILE;
#pragma unroll
    for (int c = 0; c < PT_ITEMS / 4; c++) {
      const int64_t i0 = lo + ((int64_t)c * BLOCK + threadIdx.x) * 4;
      uint32_t packed = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        if (i0 + k < n) {
          const int64_t i = i0 + k;
          // a row the predicate dropped (false or NULL) takes no part: id 0xFF
          if (row_mask && !(((row_mask[i >> 6] & (row_mask_valid ? row_mask_valid[i >> 6] : ~0ull)) >> (i & 63)) & 1ull)) {
            packed |= 0xFFu << (8 * k);
            continue;
          }
          const unsigned p = (unsigned)(((unsigned long long)((long long)key[i] - kmin)) >> shift) & mask;
          packed |= (p & 0xFFu) << (8 * k);
          atomicAdd(&sh[p & (MAX_PARTS - 1)], 1u);
        }
      }
      if (i0 + 3 < n) {
        *reinterpret_cast<uint32_t*>(part + i0) = packed;
      } else {
#pragma unroll
        for (int k = 0; k < 4; k++)
          if (i0 + k < n) part[i0 + k] = (uint8_t)(packed >> (8 * k));
      }
    }
    __syncthreads();
    if ((int)threadIdx.x < nparts) counts[(int64_t)threadIdx.x * n_tiles + t] = sh[threadIdx.x];
    __syncthreads();
  }
}

RangePartition partition_by_key_range(const void* key, int key_type, int64_t n, long long kmin, int shift, unsigned mask, int nparts,
                                      const std::vector<const void*>& src, const std::vector<int>& widths, bool want_bounds, const uint64_t* row_mask,
                                      const uint64_t* row_mask_valid) {
  Runtime& r = rt();
  DFGPU_CHECK(nparts >= 1 && nparts <= MAX_PARTS && src.size() == widths.size() && n > 0, "partition_by_key_range: bad arguments");
  const int64_t n_tiles = (n + PT_TILE - 1) / PT_TILE;
  const int tile_grid = (int)std::min<int64_t>(n_tiles, 256 * 8);
  int nbits = 0;
  while ((1 << nbits) < nparts) nbits++;
  BufPtr part = make_buf((size_t)n + 64);
  BufPtr counts = make_buf((size_t)nparts * n_tiles * 4);
  BufPtr prefix = make_buf((size_t)(nparts * n_tiles + 1) * 8);
  {
    ProfileScope ps("partition_count_range", n * type_width(key_type) + n);
    switch (key_type) {
      case DFGPU_INT64: k_part_count_range<int64_t><<<tile_grid, BLOCK, 0, r.stream>>>((const int64_t*)key, n, kmin, shift, mask, nparts, n_tiles, row_mask, row_mask_valid, part->as<uint8_t>(), counts->as<uint32_t>()); break;
      case DFGPU_UINT32: k_part_count_range<uint32_t><<<tile_grid, BLOCK, 0, r.stream>>>((const uint32_t*)key, n, kmin, shift, mask, nparts, n_tiles, row_mask, row_mask_valid, part->as<uint8_t>(), counts->as<uint32_t>()); break;
      case DFGPU_UINT8: k_part_count_range<uint8_t><<<tile_grid, BLOCK, 0, r.stream>>>((const uint8_t*)key, n, kmin, shift, mask, nparts, n_tiles, row_mask, row_mask_valid, part->as<uint8_t>(), counts->as<uint32_t>()); break;
      default: k_part_count_range<int32_t><<<tile_grid, BLOCK, 0, r.stream>>>((const int32_t*)key, n, kmin, shift, mask, nparts, n_tiles, row_mask, row_mask_valid, part->as<uint8_t>(), counts->as<uint32_t>()); break;
    }
    DFGPU_HIP(hipGetLastError());
  }
  scan_u32(counts->as<uint32_t>(), (int64_t)nparts * n_tiles, prefix->as<uint64_t>());
  RangePartition out;
  out.bounds.resize((size_t)nparts + 1);
  if (want_bounds) {
    // the partitions' first offsets = every n_tiles-th entry of the prefix: ONE strided copy (64 separate 8-byte copies cost 0.6 ms)
    DFGPU_HIP(hipMemcpy2DAsync(out.bounds.data(), 8, prefix->ptr, (size_t)n_tiles * 8, 8, (size_t)nparts, hipMemcpyDeviceToHost, r.stream));
    DFGPU_HIP(hipStreamSynchronize(r.stream));
    out.bounds[(size_t)nparts] = (uint64_t)n;
  } else {
    out.bounds.clear();
  }
  out.rows = n;
  if (row_mask) out.rows = (int64_t)read_u64(prefix->as<uint64_t>() + (int64_t)nparts * n_tiles);   // rows that take part
  for (size_t c = 0; c < src.size(); c++) out.cols.push_back(make_buf((size_t)n * widths[c] + 64));
  for (size_t c0 = 0; c0 < src.size(); c0 += PART_MAX_COLS) {
    PartCols pc{};
    int64_t bytes = n;
    pc.n = (int)std::min<size_t>(PART_MAX_COLS, src.size() - c0);
    for (int k = 0; k < pc.n; k++) {
      pc.src[k] = src[c0 + k];
      pc.dst[k] = out.cols[c0 + k]->ptr;
      pc.width[k] = widths[c0 + k];
      DFGPU_CHECK(pc.width[k] == 16 || pc.width[k] == 8 || pc.width[k] == 4 || pc.width[k] == 1, "partition_by_key_range: column width");
      bytes += 2 * n * pc.width[k];
    }
    ProfileScope ps("partition_scatter", bytes);
    if (row_mask) k_part_scatter<true><<<tile_grid, BLOCK, 0, r.stream>>>(pc, part->as<uint8_t>(), prefix->as<uint64_t>(), n, nparts, nbits, n_tiles);
    else k_part_scatter<false><<<tile_grid, BLOCK, 0, r.stream>>>(pc, part->as<uint8_t>(), prefix->as<uint64_t>(), n, nparts, nbits, n_tiles);
    DFGPU_HIP(hipGetLastError());
  }
  return out;
}

// mask of rows routed to partition q (fallback path for nullable / Boolean payload columns)
__global__ __launch_bounds__(BLOCK) void k_part_mask(const uint8_t* __restrict__ part, int64_t n, int q, uint64_t* __restrict__ mask) {
  const int64_t n_words = (n + 63) >> 6;
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * BLOCK) >> 6;
  for (int64_t w = wave; w < n_words; w += n_waves) {
    int64_t i = (w << 6) + lane_id();
    uint64_t m = ballot64(i < n && part[i] == q);
    if (lane_id() == 0) mask[w] = m;
  }
}

static std::vector<Table> partition_table_fixed_keys(const Table& in, const std::vector<int>& key_cols, int nparts);
__global__ void k_part_bounds(const uint64_t* __restrict__ prefix, int64_t n_tiles, int nparts, uint64_t* __restrict__ bounds) {
  if ((int)threadIdx.x < nparts) bounds[threadIdx.x] = prefix[(int64_t)threadIdx.x * n_tiles];
}
// String keys (Utf8 bytes or dictionary-encoded) are routed on a hash of their BYTES (strings.hip string_hash_column), carried as an
// extra column behind the caller's columns through the partitioning and dropped from the partitions: indices of a dictionary mean
// nothing across tables, calls or ranks.
std::vector<Table> partition_table(const Table& in, const std::vector<int>& key_cols, int nparts) {
  Table work;
  std::vector<int> kc = key_cols;
  bool any = false;
  for (size_t i = 0; i < kc.size(); i++) {
    DFGPU_CHECK(kc[i] >= 0 && kc[i] < (int)in.cols.size(), "partition key column out of range");
    const Column& c = in.cols[(size_t)kc[i]];
    if (c.field.type == DFGPU_BOOL) {   // Boolean keys are hashed as one byte per row (hash_utils.rs:306-345: value by value)
      if (!any) work = in;
      any = true;
      work.cols.push_back(bool_as_u8(c, in.nrows));
      kc[i] = (int)work.cols.size() - 1;
      continue;
    }
    if (c.field.type != DFGPU_UTF8 && !c.dict) continue;
    if (!any) work = in;
    any = true;
    work.cols.push_back(string_hash_column(c));
    kc[i] = (int)work.cols.size() - 1;
  }
  if (!any) return partition_table_fixed_keys(in, key_cols, nparts);
  std::vector<Table> parts = partition_table_fixed_keys(work, kc, nparts);
  for (Table& p : parts) p.cols.resize(in.cols.size());
  return parts;
}
static std::vector<Table> partition_table_fixed_keys(const Table& in, const std::vector<int>& key_cols, int nparts) {
  Runtime& r = rt();
  DFGPU_CHECK(nparts >= 1 && nparts <= MAX_PARTS, "dfgpu_partition supports 1..64 partitions");
  const int64_t n = in.nrows;
  const int64_t n_words = (n + 63) / 64;
  std::vector<const Column*> keys;
  for (int c : key_cols) {
    DFGPU_CHECK(c >= 0 && c < (int)in.cols.size(), "partition key column out of range");
    keys.push_back(&in.cols[c]);
  }
  KeySet ks = keyset_of(keys);
  std::vector<Table> outs(nparts);
  if (n == 0) {
    for (auto& o : outs) {
      o.nrows = 0;
      for (auto& c : in.cols) o.cols.push_back(alloc_like(c, 0));
    }
    return outs;
  }
  const int64_t n_tiles = (n + PT_TILE - 1) / PT_TILE;
  const int tile_grid = (int)std::min<int64_t>(n_tiles, 256 * 8);
  int nbits = 0;
  while ((1 << nbits) < nparts) nbits++;
  bool simple = true;
  for (auto& c : in.cols) simple &= !c.validity && c.field.type != DFGPU_BOOL && c.field.type != DFGPU_UTF8;
  const bool gen2 = nparts <= 16;   // (packed 16-bit counters in registers; more partitions take the ballot-counting pass)
  const FastMod fm = fastmod_for((uint32_t)nparts);
  BufPtr part = make_buf((size_t)n + 64);
  BufPtr counts = make_buf((size_t)nparts * n_tiles * 4);
  BufPtr prefix = make_buf((size_t)(nparts * n_tiles + 1) * 8);
  int64_t key_bytes = 0;
  for (int i = 0; i < ks.n; i++) key_bytes += n * ks.c[i].width;
  {
    ProfileScope ps("partition_count", key_bytes + n);
    if (!gen2) k_part_count<<<tile_grid, BLOCK, 0, r.stream>>>(ks, n, nparts, fm, n_tiles, part->as<uint8_t>(), counts->as<uint32_t>());
    else {
      // one integer key column without NULLs: the typed kernel
      int kt = KT_ANY;
      if (ks.n == 1 && !ks.c[0].valid) kt = ks.c[0].type == DFGPU_INT64 || ks.c[0].type == DFGPU_UINT64 ? KT_I64 : ks.c[0].type == DFGPU_INT32 || ks.c[0].type == DFGPU_DATE32 ? KT_I32 : KT_ANY;
      auto launch = [&](auto kern) { kern<<<tile_grid, BLOCK, 0, r.stream>>>(ks, n, nparts, fm, n_tiles, part->as<uint8_t>(), counts->as<uint32_t>()); };
      if (nparts <= 8) {
        if (kt == KT_I64) launch(k_part_count2<2, KT_I64>);
        else if (kt == KT_I32) launch(k_part_count2<2, KT_I32>);
        else launch(k_part_count2<2, KT_ANY>);
      } else {
        if (kt == KT_I64) launch(k_part_count2<4, KT_I64>);
        else if (kt == KT_I32) launch(k_part_count2<4, KT_I32>);
        else launch(k_part_count2<4, KT_ANY>);
      }
    }
    DFGPU_HIP(hipGetLastError());
  }
  scan_u32(counts->as<uint32_t>(), (int64_t)nparts * n_tiles, prefix->as<uint64_t>());
  // partition boundaries = prefix at the start of each partition's row of the matrix.  The scatter does not need them on the host:
  // it is launched first, they are read while it runs (the device idled for the read-back's round trip before)
  std::vector<uint64_t> bounds(nparts + 1);
  auto read_bounds = [&]() {
    BufPtr d_bounds = make_buf((size_t)nparts * 8);
    k_part_bounds<<<1, MAX_PARTS, 0, r.stream>>>(prefix->as<uint64_t>(), n_tiles, nparts, d_bounds->as<uint64_t>());
    DFGPU_HIP(hipGetLastError());
    d2h(bounds.data(), d_bounds->ptr, (size_t)nparts * 8);
    bounds[nparts] = (uint64_t)n;
  };

  if (simple) {
    std::vector<Column> whole;
    for (auto& c : in.cols) whole.push_back(alloc_like(c, n));
    for (size_t c0 = 0; c0 < in.cols.size(); c0 += PART_MAX_COLS) {
      PartCols pc{};
      int64_t bytes = n;
      pc.n = (int)std::min<size_t>(PART_MAX_COLS, in.cols.size() - c0);
      for (int k = 0; k < pc.n; k++) {
        pc.src[k] = in.cols[c0 + k].ptr();
        pc.dst[k] = whole[c0 + k].data->ptr;
        pc.width[k] = type_width(in.cols[c0 + k].field.type);
        bytes += 2 * n * pc.width[k];
      }
      ProfileScope ps("partition_scatter", bytes);
      k_part_scatter<false><<<tile_grid, BLOCK, 0, r.stream>>>(pc, part->as<uint8_t>(), prefix->as<uint64_t>(), n, nparts, nbits, n_tiles);
      DFGPU_HIP(hipGetLastError());
    }
    read_bounds();
    for (int p = 0; p < nparts; p++) {
      outs[p].nrows = (int64_t)(bounds[p + 1] - bounds[p]);
      for (size_t ci = 0; ci < in.cols.size(); ci++) {
        Column c = whole[ci];
        c.length = outs[p].nrows;
        c.data_offset = (size_t)bounds[p] * type_width(c.field.type);
        outs[p].cols.push_back(std::move(c));
      }
    }
  } else {
    std::vector<int> all;
    for (int i = 0; i < (int)in.cols.size(); i++) all.push_back(i);
    BufPtr mask = make_buf(bitmap_bytes(n));
    read_bounds();
    for (int p = 0; p < nparts; p++) {
      k_part_mask<<<grid_for(n_words, BLOCK / WAVE), BLOCK, 0, r.stream>>>(part->as<uint8_t>(), n, p, mask->as<uint64_t>());
      outs[p] = compact_table(in, all, mask->as<uint64_t>(), nullptr);
    }
  }
  return outs;
}

}  // namespace dfgpu

using namespace dfgpu;

extern "C" {

int dfgpu_hash_columns(dfgpu_table_t input, const int* key_cols, int nkeys, uint64_t seed, uint64_t* out_device) {
  return guarded([&] {
    require_init();
    Table* t = unwrap(input);
    std::vector<const Column*> keys;
    for (int i = 0; i < nkeys; i++) {
      DFGPU_CHECK(key_cols[i] >= 0 && key_cols[i] < (int)t->cols.size(), "key column out of range");
      keys.push_back(&t->cols[key_cols[i]]);
    }
    hash_columns(keys, t->nrows, seed, out_device, false);
  });
}

int dfgpu_partition(dfgpu_table_t input, const int* key_cols, int nkeys, int nparts, dfgpu_table_t* outs) {
  return guarded([&] {
    require_init();
    auto parts = partition_table(*unwrap(input), std::vector<int>(key_cols, key_cols + nkeys), nparts);
    for (int p = 0; p < nparts; p++) outs[p] = wrap(new Table(std::move(parts[p])));
  });
}

}  // extern "C"
