// scan.hip — device-wide exclusive prefix sums (u32 counts -> u64 offsets).  Three launches:
// per-chunk reduce, single-workgroup scan of the chunk sums, per-chunk downsweep.  The input
// is a functor so the popcount of selection-mask words is computed on the fly (the mask is
// read twice = 2 x N/8 bytes, negligible against the columns it selects).
#include "device.hpp"
#include "internal.hpp"

namespace dfgpu {

constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_CHUNK = BLOCK * SCAN_ITEMS;  // 2048 elements per workgroup

struct InU32 {
  const uint32_t* p;
  __device__ __forceinline__ uint32_t operator()(int64_t i) const { return p[i]; }
};
struct InMaskPopc {
  const uint64_t* mask;
  const uint64_t* valid;
  int64_t nrows;
  __device__ __forceinline__ uint32_t operator()(int64_t w) const {
    uint64_t m = mask[w];
    if (valid) m &= valid[w];
    int64_t rem = nrows - w * 64;
    if (rem < 64) m &= (rem <= 0) ? 0ull : ((~0ull) >> (64 - rem));
    return (uint32_t)__popcll(m);
  }
};

__device__ __forceinline__ uint64_t block_exclusive_scan(uint64_t v, uint64_t* total) {
  __shared__ uint64_t wsum[BLOCK / WAVE];
  uint64_t inc = wave_inclusive_sum(v);
  int w = threadIdx.x >> 6;
  if (lane_id() == 63) wsum[w] = inc;
  __syncthreads();
  uint64_t base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < BLOCK / WAVE; i++) {
    if (i < w) base += wsum[i];
    tot += wsum[i];
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

template <typename In>
__global__ __launch_bounds__(BLOCK) void k_scan_reduce(In in, int64_t n, uint64_t* chunk_sums) {
  int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK + (int64_t)threadIdx.x * SCAN_ITEMS;
  uint64_t s = 0;
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; j++)
    if (base + j < n) s += in(base + j);
  uint64_t tot;
  block_exclusive_scan(s, &tot);
  if (threadIdx.x == 0) chunk_sums[blockIdx.x] = tot;
}

// one workgroup: in-place exclusive scan of the chunk sums; writes the grand total to *total
__global__ __launch_bounds__(BLOCK) void k_scan_sums(uint64_t* sums, int64_t n_chunks, uint64_t* total) {
  uint64_t carry = 0;
  for (int64_t base = 0; base < n_chunks; base += BLOCK) {
    int64_t i = base + threadIdx.x;
    uint64_t v = i < n_chunks ? sums[i] : 0;
    uint64_t tot;
    uint64_t ex = block_exclusive_scan(v, &tot);
    if (i < n_chunks) sums[i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) *total = carry;
}

template <typename In>
__global__ __launch_bounds__(BLOCK) void k_scan_down(In in, int64_t n, const uint64_t* chunk_base, uint64_t* out) {
  int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK + (int64_t)threadIdx.x * SCAN_ITEMS;
  uint32_t v[SCAN_ITEMS];
  uint64_t s = 0;
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; j++) {
    v[j] = base + j < n ? in(base + j) : 0;
    s += v[j];
  }
  uint64_t tot;
  uint64_t ex = block_exclusive_scan(s, &tot) + chunk_base[blockIdx.x];
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; j++) {
    if (base + j < n) out[base + j] = ex;
    ex += v[j];
  }
}

// ---- one launch (round 6): chained scan with decoupled look-back.  Every workgroup takes the next chunk by ticket, reduces it,
// publishes the aggregate, wave 0 sums its predecessors' states back to the nearest published prefix (device.hpp lookback_exclusive)
// and the chunk is written — the input is read ONCE and there is one launch instead of three.
constexpr int CH_ITEMS = 16;
constexpr int CH_CHUNK = BLOCK * CH_ITEMS;  // 4096 elements per workgroup
template <typename In>
__global__ __launch_bounds__(BLOCK) void k_scan_chained(In in, int64_t n, uint64_t* __restrict__ state, unsigned* __restrict__ ticket, uint64_t* __restrict__ out) {
  __shared__ unsigned s_tile;
  __shared__ uint64_t s_excl;
  if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
  __syncthreads();
  const int64_t tile = s_tile;
  const int64_t n_tiles = (n + CH_CHUNK - 1) / CH_CHUNK;
  const int64_t base = tile * CH_CHUNK + (int64_t)threadIdx.x * CH_ITEMS;
  uint32_t v[CH_ITEMS];
  uint64_t s = 0;
#pragma unroll
  for (int j = 0; j < CH_ITEMS; j++) {
    v[j] = base + j < n ? in(base + j) : 0;
    s += v[j];
  }
  uint64_t tot;
  uint64_t ex = block_exclusive_scan(s, &tot);
  if (threadIdx.x < WAVE) {
    const uint64_t excl = lookback_exclusive(state, tile, tot);
    if (threadIdx.x == 0) {
      s_excl = excl;
      if (tile == n_tiles - 1) out[n] = excl + tot;
    }
  }
  __syncthreads();
  ex += s_excl;
#pragma unroll
  for (int j = 0; j < CH_ITEMS; j++) {
    if (base + j < n) out[base + j] = ex;
    ex += v[j];
  }
}

// the down-sweep over bitmap words that also leaves the rank map's interleaved {word, prefix} pairs (join.hip rank_tab): the words
// and their prefixes are in registers here — writing the pairs now saves the separate interleave pass its read of both arrays
__global__ __launch_bounds__(BLOCK) void k_scan_down_tab(InMaskPopc in, int64_t n, const uint64_t* chunk_base, uint64_t* out, ulonglong2* __restrict__ tab) {
  int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK + (int64_t)threadIdx.x * SCAN_ITEMS;
  uint64_t w[SCAN_ITEMS];
  uint64_t s = 0;
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; j++) {
    w[j] = base + j < n ? in.mask[base + j] : 0ull;
    s += (uint32_t)__popcll(w[j]);
  }
  uint64_t tot;
  uint64_t ex = block_exclusive_scan(s, &tot) + chunk_base[blockIdx.x];
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; j++) {
    if (base + j < n) {
      out[base + j] = ex;
      tab[base + j] = make_ulonglong2(w[j], ex);
    }
    ex += (uint32_t)__popcll(w[j]);
  }
}

template <typename In>
static void run_scan(In in, int64_t n, uint64_t* out, const char* name) {
  Runtime& r = rt();
  if (n == 0) {
    DFGPU_HIP(hipMemsetAsync(out, 0, 8, r.stream));
    return;
  }
  // (measured, profiles/r6_ops.md: over SF300's 28 M bitmap words the chained form takes 0.46 ms against 0.28 — every workgroup parks on
  // agent-scope loads of its predecessors' states behind its own streaming loads, as the look-back probe did in round 2 — so it serves the
  // SHORT scans, where one launch instead of three is the whole difference; scan.chained_max_tiles=0 turns it off, a large value forces it)
  const int64_t n_tiles_ch = (n + CH_CHUNK - 1) / CH_CHUNK;
  if (n_tiles_ch <= option_int("scan.chained_max_tiles", 8)) {
    const int64_t n_tiles = n_tiles_ch;
    BufPtr state = make_zero_buf((size_t)(n_tiles + 2) * 8);   // [n_tiles] states, then the ticket
    ProfileScope ps(name, 0);
    k_scan_chained<<<(unsigned)n_tiles, BLOCK, 0, r.stream>>>(in, n, state->as<uint64_t>(), reinterpret_cast<unsigned*>(state->as<uint64_t>() + n_tiles), out);
    DFGPU_HIP(hipGetLastError());
    return;
  }
  int64_t n_chunks = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
  BufPtr sums = make_buf((size_t)n_chunks * 8);
  ProfileScope ps(name, 0);
  k_scan_reduce<<<(unsigned)n_chunks, BLOCK, 0, r.stream>>>(in, n, sums->as<uint64_t>());
  k_scan_sums<<<1, BLOCK, 0, r.stream>>>(sums->as<uint64_t>(), n_chunks, out + n);
  k_scan_down<<<(unsigned)n_chunks, BLOCK, 0, r.stream>>>(in, n, sums->as<uint64_t>(), out);
  DFGPU_HIP(hipGetLastError());
}

void scan_mask_popcounts(const uint64_t* mask, const uint64_t* valid, int64_t nrows, uint64_t* out_prefix) {
  int64_t n_words = (nrows + 63) / 64;
  run_scan(InMaskPopc{mask, valid, nrows}, n_words, out_prefix, "scan_mask_popcounts");
}
void scan_u32(const uint32_t* in, int64_t n, uint64_t* out_prefix) { run_scan(InU32{in}, n, out_prefix, "scan_u32"); }
// prefix popcounts of `n_words` whole bitmap words (no ragged tail, no validity) + the interleaved {word, prefix} table
void scan_bitmap_words_tab(const uint64_t* words, int64_t n_words, uint64_t* out_prefix, void* out_tab) {
  Runtime& r = rt();
  DFGPU_CHECK(n_words > 0, "scan_bitmap_words_tab: empty bitmap");
  const InMaskPopc in{words, nullptr, n_words * 64};
  const int64_t n_chunks = (n_words + SCAN_CHUNK - 1) / SCAN_CHUNK;
  BufPtr sums = make_buf((size_t)n_chunks * 8);
  ProfileScope ps("scan_mask_popcounts", 0);
  k_scan_reduce<<<(unsigned)n_chunks, BLOCK, 0, r.stream>>>(in, n_words, sums->as<uint64_t>());
  k_scan_sums<<<1, BLOCK, 0, r.stream>>>(sums->as<uint64_t>(), n_chunks, out_prefix + n_words);
  k_scan_down_tab<<<(unsigned)n_chunks, BLOCK, 0, r.stream>>>(in, n_words, sums->as<uint64_t>(), out_prefix, reinterpret_cast<ulonglong2*>(out_tab));
  DFGPU_HIP(hipGetLastError());
}

}  // namespace dfgpu
