// How fast can one workgroup per tile claim an output range with a returning atomicAdd on ONE cursor?
// (the UNORDERED flavour of k_join_probe_fused: 293 K tiles at SF100).  Variants: no atomic, one cursor, 8 cursors
// (by XCD = blockIdx % 8), and the atomic placed between two barriers as in the kernel.
// build: hipcc -O3 --offload-arch=gfx950 tile_atomics.hip -o build/tile_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
struct alignas(128) Cur { unsigned long long v; char pad[120]; };
template <int MODE>
__global__ __launch_bounds__(256) void k(const uint64_t* __restrict__ keys, Cur* cur, uint64_t* sink, int loads) {
  __shared__ uint64_t s_pre;
  uint64_t acc = 0;
  const int64_t base = (int64_t)blockIdx.x * 2048 + threadIdx.x;
  for (int j = 0; j < loads; j++) acc += keys[base + j * 256];
  __syncthreads();
  if (threadIdx.x == 0) {
    uint64_t pre = 0;
    if (MODE == 1) pre = atomicAdd(&cur[0].v, 100ull);
    if (MODE == 2) pre = atomicAdd(&cur[blockIdx.x & 7].v, 100ull);
    if (MODE == 3) pre = atomicAdd(&cur[blockIdx.x & 63].v, 100ull);
    s_pre = pre;
  }
  __syncthreads();
  if (acc + s_pre == 0x1234567ull) sink[0] = acc;
}
int main() {
  const int64_t tiles = 293000;
  uint64_t* keys; Cur* cur; uint64_t* sink;
  CK(hipMalloc(&keys, tiles * 2048 * 8)); CK(hipMemset(keys, 1, tiles * 2048 * 8));
  CK(hipMalloc(&cur, 64 * sizeof(Cur))); CK(hipMemset(cur, 0, 64 * sizeof(Cur)));
  CK(hipMalloc(&sink, 8));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int loads : {0, 8}) {
    for (int mode = 0; mode < 4; mode++) {
      float best = 1e9;
      for (int it = 0; it < 5; it++) {
        CK(hipEventRecord(a));
        if (mode == 0) k<0><<<tiles, 256>>>(keys, cur, sink, loads);
        if (mode == 1) k<1><<<tiles, 256>>>(keys, cur, sink, loads);
        if (mode == 2) k<2><<<tiles, 256>>>(keys, cur, sink, loads);
        if (mode == 3) k<3><<<tiles, 256>>>(keys, cur, sink, loads);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
      }
      printf("loads/lane=%d cursors=%s: %.3f ms for %lld tiles = %.1f ns/tile\n", loads, mode == 0 ? "none" : mode == 1 ? "1" : mode == 2 ? "8" : "64", best, (long long)tiles, best * 1e6 / tiles);
    }
  }
  return 0;
}
