// What does the shape of a streaming READ kernel cost on MI355X?  The selective probe's counts pass reads a 4-byte predicate column and an
// 8-byte key column for 1.8 G rows (21.6 GB) and reached 4.47 TB/s as "one lane = one row, 8 rows per lane, one 2048-row tile per
// workgroup" (profiles/r6_q3_sf300.md) where 16-byte-per-lane copies reach 6.3 TB/s.  Variants over the same two columns:
//   A  lane = row:      8 x (4-byte load + 8-byte load) per lane, rows 64 apart, one tile per workgroup           (the kernel's shape)
//   B  lane = 8 rows:   2 x dwordx4 of dates + 4 x dwordx4 of keys per lane, one 2048-row tile per workgroup
//   C  B with 4 tiles per workgroup, the next tile's loads issued before the current tile is consumed (software pipeline)
//   D  B with non-temporal loads
// and the random side: 219 M lookups of a 5.6 MB bitmap (45 M bits) while 5.4 GB stream by, plain vs non-temporal streaming loads.
// build: hipcc -O3 --offload-arch=gfx950 stream_width.hip -o build/stream_width
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned v4u __attribute__((ext_vector_type(4)));

template <bool NT, typename T>
__device__ __forceinline__ T ld(const T* p) { return NT ? __builtin_nontemporal_load(p) : *p; }

// A: lane = row
template <bool NT>
__global__ __launch_bounds__(256) void k_row(const int* __restrict__ d, const long long* __restrict__ k, int64_t n, int lit, unsigned* __restrict__ counts) {
  __shared__ unsigned s[4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t w0 = (int64_t)blockIdx.x * 32 + wv * 8;
  int dv[8];
  long long kv[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int64_t p = ((w0 + j) << 6) + lane;
    dv[j] = ld<NT>(d + (p < n ? p : n - 1));
  }
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int64_t p = ((w0 + j) << 6) + lane;
    kv[j] = ld<NT>(k + (p < n ? p : n - 1));
  }
  unsigned c = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) c += __popcll(__ballot(dv[j] > lit && (kv[j] & 3) == 1));
  if (lane == 0) s[wv] = c;
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}

// E / F / G: A + the dependent lookup of a bitmap by clustered keys (lineitem against orders): SKIP = rows the predicate drops read word 0
// instead of their own word; WRITE = the 1-bit-per-row output words and the tile count leave as in the real kernel
template <bool NT, bool SKIP, bool WRITE>
__global__ __launch_bounds__(256) void k_row_lookup(const int* __restrict__ d, const long long* __restrict__ k, int64_t n, int lit, const unsigned long long* __restrict__ bits,
                                                    unsigned long long nbits, unsigned* __restrict__ counts, unsigned long long* __restrict__ out_words) {
  __shared__ unsigned s[4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t w0 = (int64_t)blockIdx.x * 32 + wv * 8;
  int dv[8];
  unsigned long long kv[8], bw[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int64_t p = ((w0 + j) << 6) + lane;
    dv[j] = ld<NT>(d + (p < n ? p : n - 1));
  }
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int64_t p = ((w0 + j) << 6) + lane;
    kv[j] = (unsigned long long)ld<NT>(k + (p < n ? p : n - 1));
  }
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const bool ok = kv[j] < nbits && (!SKIP || dv[j] > lit);
    bw[j] = bits[ok ? (kv[j] >> 6) : 0];
  }
  unsigned c = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const unsigned long long word = __ballot(dv[j] > lit && kv[j] < nbits && ((bw[j] >> (kv[j] & 63)) & 1));
    if (WRITE && lane == 0) out_words[w0 + j] = word;
    c += __popcll(word);
  }
  if (lane == 0) s[wv] = c;
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}

// B / C / D: lane = 8 consecutive rows; TILES tiles of 2048 rows per workgroup, pipelined
template <bool NT, int TILES>
__global__ __launch_bounds__(256) void k_vec(const int* __restrict__ d, const long long* __restrict__ k, int64_t n, int lit, unsigned* __restrict__ counts) {
  __shared__ unsigned s[4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  unsigned c = 0;
  v4u dv[2], kv[4], dn[2], kn[4];
  auto issue = [&](int t, v4u (&dd)[2], v4u (&kk)[4]) {
    const int64_t row = ((int64_t)blockIdx.x * TILES + t) * 2048 + (int64_t)threadIdx.x * 8;
    const int64_t r = row + 8 <= n ? row : 0;   // (sizes here are multiples of the tile)
    const v4u* dp = reinterpret_cast<const v4u*>(d + r);
    const v4u* kp = reinterpret_cast<const v4u*>(k + r);
    dd[0] = ld<NT>(dp); dd[1] = ld<NT>(dp + 1);
    kk[0] = ld<NT>(kp); kk[1] = ld<NT>(kp + 1); kk[2] = ld<NT>(kp + 2); kk[3] = ld<NT>(kp + 3);
  };
  issue(0, dv, kv);
#pragma unroll
  for (int t = 0; t < TILES; t++) {
    if (t + 1 < TILES) issue(t + 1, dn, kn);
    unsigned bits = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const int dd = (int)dv[q >> 2][q & 3];
      const unsigned klo = kv[q >> 1][(q & 1) * 2];
      bits |= (unsigned)(dd > lit && (klo & 3) == 1) << q;
    }
    c += __popc(bits);
    if (t + 1 < TILES) {
#pragma unroll
      for (int q = 0; q < 2; q++) dv[q] = dn[q];
#pragma unroll
      for (int q = 0; q < 4; q++) kv[q] = kn[q];
    }
  }
  // wave sum
  for (int o = 32; o; o >>= 1) c += __shfl_xor(c, o);
  if (lane == 0) s[wv] = c;
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}

// random lookups of a bitmap while two columns stream by (the orders side of Q3's semi-join)
template <bool NT>
__global__ __launch_bounds__(256) void k_lookup(const int* __restrict__ d, const long long* __restrict__ k, int64_t n, int lit, const unsigned long long* __restrict__ bits,
                                                unsigned long long nbits, unsigned* __restrict__ counts) {
  __shared__ unsigned s[4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t w0 = (int64_t)blockIdx.x * 32 + wv * 8;
  int dv[8];
  unsigned long long kv[8], bw[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int64_t p = ((w0 + j) << 6) + lane;
    dv[j] = ld<NT>(d + (p < n ? p : n - 1));
  }
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int64_t p = ((w0 + j) << 6) + lane;
    kv[j] = (unsigned long long)ld<NT>(k + (p < n ? p : n - 1));
  }
#pragma unroll
  for (int j = 0; j < 8; j++) bw[j] = bits[(dv[j] > lit && kv[j] < nbits) ? (kv[j] >> 6) : 0];
  unsigned c = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) c += __popcll(__ballot(dv[j] > lit && ((bw[j] >> (kv[j] & 63)) & 1)));
  if (lane == 0) s[wv] = c;
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}

__global__ void k_fill(int* d, long long* k, int64_t n, unsigned long long range, int clustered) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    unsigned long long x = (unsigned long long)i * 0x9E3779B97F4A7C15ull;
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    d[i] = (int)(x % 2406);
    k[i] = clustered ? (long long)(((i / 4) / 8) * 32 + (i / 4) % 8 + 1) : (long long)((x >> 11) % range);   // TPC-H order keys: 8 used of every 32, ~4 lines each
  }
}

int main() {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto timeit = [&](const char* what, double bytes, auto launch) {
    float best = 1e9;
    for (int it = 0; it < 6; it++) {
      CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    CK(hipGetLastError());
    printf("%-78s %8.3f ms  %7.1f GB/s\n", what, best, bytes / (best * 1e-3) / 1e9);
  };
  {
    const int64_t n = (int64_t)1800 * 1000 * 1000 / 8192 * 8192;
    int* d; long long* k; unsigned* counts;
    CK(hipMalloc(&d, n * 4)); CK(hipMalloc(&k, n * 8)); CK(hipMalloc(&counts, (n / 2048 + 1) * 4));
    k_fill<<<4096, 256>>>(d, k, n, 1, 1); CK(hipDeviceSynchronize());
    const double bytes = (double)n * 12;
    const unsigned tiles = (unsigned)(n / 2048);
    timeit("A  lane = row, 8 x (4 B + 8 B) loads per lane, 1 tile / workgroup", bytes, [&] { k_row<false><<<tiles, 256>>>(d, k, n, 1200, counts); });
    timeit("A' the same, non-temporal loads", bytes, [&] { k_row<true><<<tiles, 256>>>(d, k, n, 1200, counts); });
    {
      const unsigned long long nbits = (unsigned long long)n + 64;   // keys reach n / 4 * 4: the bitmap of the whole key range, 225 MB
      unsigned long long *bits, *ow;
      CK(hipMalloc(&bits, nbits / 8 + 64)); CK(hipMemset(bits, 0x10, nbits / 8 + 64)); CK(hipMalloc(&ow, n / 8 + 64));
      timeit("E  A' + bitmap lookup by clustered keys, dropped rows read word 0", bytes, [&] { k_row_lookup<true, true, false><<<tiles, 256>>>(d, k, n, 1200, bits, nbits, counts, ow); });
      timeit("F  A' + bitmap lookup by clustered keys, every row reads its own word", bytes, [&] { k_row_lookup<true, false, false><<<tiles, 256>>>(d, k, n, 1200, bits, nbits, counts, ow); });
      timeit("G  E + output words written", bytes, [&] { k_row_lookup<true, true, true><<<tiles, 256>>>(d, k, n, 1200, bits, nbits, counts, ow); });
      timeit("H  F + output words written", bytes, [&] { k_row_lookup<true, false, true><<<tiles, 256>>>(d, k, n, 1200, bits, nbits, counts, ow); });
      timeit("H0 H with plain loads", bytes, [&] { k_row_lookup<false, false, true><<<tiles, 256>>>(d, k, n, 1200, bits, nbits, counts, ow); });
      CK(hipFree(bits)); CK(hipFree(ow));
    }
    timeit("B  lane = 8 rows (2 + 4 dwordx4 loads), 1 tile / workgroup", bytes, [&] { k_vec<false, 1><<<tiles, 256>>>(d, k, n, 1200, counts); });
    timeit("D  B with non-temporal loads", bytes, [&] { k_vec<true, 1><<<tiles, 256>>>(d, k, n, 1200, counts); });
    timeit("C  lane = 8 rows, 4 tiles / workgroup, software-pipelined", bytes, [&] { k_vec<false, 4><<<tiles / 4, 256>>>(d, k, n, 1200, counts); });
    timeit("C' the same, non-temporal loads", bytes, [&] { k_vec<true, 4><<<tiles / 4, 256>>>(d, k, n, 1200, counts); });
    timeit("C8 lane = 8 rows, 8 tiles / workgroup, software-pipelined", bytes, [&] { k_vec<false, 8><<<tiles / 8, 256>>>(d, k, n, 1200, counts); });
    CK(hipFree(d)); CK(hipFree(k)); CK(hipFree(counts));
  }
  {
    const int64_t n = (int64_t)450 * 1000 * 1000 / 8192 * 8192;
    for (unsigned long long nbits : {45000000ull, 30000000ull, 20000000ull, 150000000ull}) {
      int* d; long long* k; unsigned* counts; unsigned long long* bits;
      CK(hipMalloc(&d, n * 4)); CK(hipMalloc(&k, n * 8)); CK(hipMalloc(&counts, (n / 2048 + 1) * 4)); CK(hipMalloc(&bits, nbits / 8 + 64)); CK(hipMemset(bits, 0x11, nbits / 8 + 64));
      k_fill<<<4096, 256>>>(d, k, n, nbits, 0); CK(hipDeviceSynchronize());
      const unsigned tiles = (unsigned)(n / 2048);
      char what[160];
      snprintf(what, sizeof what, "lookup: 450 M rows, ~50 %% pass, bitmap %.1f MB, plain streaming loads", nbits / 8e6);
      timeit(what, (double)n * 12, [&] { k_lookup<false><<<tiles, 256>>>(d, k, n, 1200, bits, nbits, counts); });
      snprintf(what, sizeof what, "lookup: 450 M rows, ~50 %% pass, bitmap %.1f MB, NON-TEMPORAL streaming loads", nbits / 8e6);
      timeit(what, (double)n * 12, [&] { k_lookup<true><<<tiles, 256>>>(d, k, n, 1200, bits, nbits, counts); });
      CK(hipFree(d)); CK(hipFree(k)); CK(hipFree(counts)); CK(hipFree(bits));
    }
  }
  return 0;
}
