// Microbenchmark: what ONE random access per row costs on gfx950, per wave and CU — the quantity that bounds the keyed interning pass
// and the in-place LDS accumulation (profiles/r3_agg_multikey_sq.md: both wait on a random L2 hit per row).  Every lane of every wave
// reads (or atomically updates) a pseudo-random element of a table; tables from L1-sized to L2-sized in HBM, and in LDS.
// Build: hipcc -O3 --offload-arch=gfx950 random_access.hip -o random_access      Run: ./random_access
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); return 1; } } while (0)
constexpr int BLOCK = 256;
constexpr int ROWS_PER_THREAD = 256;
__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
// MODE 0: global 4-byte plain loads      1: global 16-byte plain loads      2: global 4-byte relaxed atomic (agent-scope) loads
//      3: global 64-bit atomicAdd (no return)
template <int MODE, int U>
__global__ __launch_bounds__(BLOCK) void k_global(const uint4* __restrict__ table, unsigned long long* cells, uint32_t mask, unsigned long long* out) {
  const uint32_t tid = blockIdx.x * BLOCK + threadIdx.x;
  unsigned long long acc = 0;
  for (int it = 0; it < ROWS_PER_THREAD; it += U) {
    uint32_t idx[U];
#pragma unroll
    for (int u = 0; u < U; u++) idx[u] = mix(tid * 2654435761u + (uint32_t)(it + u)) & mask;
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (MODE == 0) acc += reinterpret_cast<const uint32_t*>(table)[idx[u]];
      else if (MODE == 1) { const uint4 e = table[idx[u]]; acc += e.x + e.z; }
      else if (MODE == 2) acc += __hip_atomic_load(reinterpret_cast<const uint32_t*>(table) + idx[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else atomicAdd(&cells[idx[u]], 1ull);
    }
  }
  if (acc == 0x1234567ull) out[0] = acc;
}
// MODE 0: LDS 4-byte reads   1: LDS 8-byte reads   2: LDS 16-byte reads   3: LDS 32-bit atomicAdd   4: LDS 64-bit atomicAdd   5: LDS 4-byte stores
template <int MODE, int U>
__global__ __launch_bounds__(BLOCK) void k_lds(uint32_t mask, unsigned long long* out) {
  extern __shared__ uint4 s_tab[];
  for (uint32_t x = threadIdx.x; x <= mask; x += BLOCK) s_tab[x] = make_uint4(x, 1, 2, 3);
  __syncthreads();
  const uint32_t tid = blockIdx.x * BLOCK + threadIdx.x;
  unsigned long long acc = 0;
  for (int it = 0; it < ROWS_PER_THREAD; it += U) {
    uint32_t idx[U];
#pragma unroll
    for (int u = 0; u < U; u++) idx[u] = mix(tid * 2654435761u + (uint32_t)(it + u)) & mask;
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (MODE == 0) acc += reinterpret_cast<volatile uint32_t*>(s_tab)[idx[u]];
      else if (MODE == 1) acc += reinterpret_cast<volatile unsigned long long*>(s_tab)[idx[u]];
      else if (MODE == 2) { const uint4 e = s_tab[idx[u]]; acc += e.x + e.z; }
      else if (MODE == 3) atomicAdd(reinterpret_cast<uint32_t*>(s_tab) + idx[u], 1u);
      else if (MODE == 4) atomicAdd(reinterpret_cast<unsigned long long*>(s_tab) + idx[u], 1ull);
      else reinterpret_cast<volatile uint32_t*>(s_tab)[idx[u]] = 0u;
    }
  }
  __syncthreads();
  if (acc == 0x1234567ull || threadIdx.x == 0) out[blockIdx.x & 1023] = acc + s_tab[1].x;
}
template <typename F>
int timed(const char* name, int blocks, F launch) {
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  launch();
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a));
  launch();
  CHECK(hipEventRecord(b));
  CHECK(hipDeviceSynchronize());
  float ms; CHECK(hipEventElapsedTime(&ms, a, b));
  const double rows = (double)blocks * BLOCK * ROWS_PER_THREAD;
  printf("%-64s %8.3f ms  %7.1f G rows/s  %6.1f clk per wave-access per CU (2.1 GHz)  => %5.2f ms per 600 M rows\n", name, ms, rows / ms / 1e6,
         (ms * 1e-3) * 2.1e9 * 256 / (rows / 64), 600e6 / (rows / ms));
  return 0;
}
int main() {
  unsigned long long* out; CHECK(hipMalloc(&out, 8192));
  uint4* table; CHECK(hipMalloc(&table, (size_t)64 << 20)); CHECK(hipMemset(table, 1, (size_t)64 << 20));
  unsigned long long* cells; CHECK(hipMalloc(&cells, (size_t)64 << 20)); CHECK(hipMemset(cells, 0, (size_t)64 << 20));
  const int blocks = 256 * 32;   // 8 workgroups of 4 waves resident per CU, four rounds
  char name[128];
  for (int kb : {16, 64, 256, 1024, 4096, 32768}) {
    const uint32_t m4 = (uint32_t)(kb * 1024 / 4 - 1), m16 = (uint32_t)(kb * 1024 / 16 - 1), m8 = (uint32_t)(kb * 1024 / 8 - 1);
    snprintf(name, sizeof name, "global  4-byte plain loads, %5d KB table, 1 row in flight", kb);
    if (timed(name, blocks, [&] { k_global<0, 1><<<blocks, BLOCK>>>(table, cells, m4, out); })) return 1;
    snprintf(name, sizeof name, "global  4-byte plain loads, %5d KB table, 4 rows in flight", kb);
    if (timed(name, blocks, [&] { k_global<0, 4><<<blocks, BLOCK>>>(table, cells, m4, out); })) return 1;
    snprintf(name, sizeof name, "global 16-byte plain loads, %5d KB table, 4 rows in flight", kb);
    if (timed(name, blocks, [&] { k_global<1, 4><<<blocks, BLOCK>>>(table, cells, m16, out); })) return 1;
    snprintf(name, sizeof name, "global  4-byte atomic loads, %5d KB table, 4 rows in flight", kb);
    if (timed(name, blocks, [&] { k_global<2, 4><<<blocks, BLOCK>>>(table, cells, m4, out); })) return 1;
    snprintf(name, sizeof name, "global 64-bit atomicAdd,    %5d KB table, 4 rows in flight", kb);
    if (timed(name, blocks, [&] { k_global<3, 4><<<blocks, BLOCK>>>(table, cells, m8, out); })) return 1;
  }
  for (int kb : {16, 64}) {
    const uint32_t m16 = (uint32_t)(kb * 1024 / 16 - 1);
    const size_t lds = (size_t)kb * 1024;
    CHECK(hipFuncSetAttribute((const void*)k_lds<0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CHECK(hipFuncSetAttribute((const void*)k_lds<1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CHECK(hipFuncSetAttribute((const void*)k_lds<2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CHECK(hipFuncSetAttribute((const void*)k_lds<3, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CHECK(hipFuncSetAttribute((const void*)k_lds<4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CHECK(hipFuncSetAttribute((const void*)k_lds<5, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    snprintf(name, sizeof name, "LDS  4-byte reads,  %3d KB table per workgroup", kb);
    if (timed(name, blocks, [&] { k_lds<0, 4><<<blocks, BLOCK, lds>>>(m16 * 4 + 3, out); })) return 1;
    snprintf(name, sizeof name, "LDS  8-byte reads,  %3d KB table per workgroup", kb);
    if (timed(name, blocks, [&] { k_lds<1, 4><<<blocks, BLOCK, lds>>>(m16 * 2 + 1, out); })) return 1;
    snprintf(name, sizeof name, "LDS 16-byte reads,  %3d KB table per workgroup", kb);
    if (timed(name, blocks, [&] { k_lds<2, 4><<<blocks, BLOCK, lds>>>(m16, out); })) return 1;
    snprintf(name, sizeof name, "LDS 32-bit atomicAdd, %3d KB table per workgroup", kb);
    if (timed(name, blocks, [&] { k_lds<3, 4><<<blocks, BLOCK, lds>>>(m16 * 4 + 3, out); })) return 1;
    snprintf(name, sizeof name, "LDS 64-bit atomicAdd, %3d KB table per workgroup", kb);
    if (timed(name, blocks, [&] { k_lds<4, 4><<<blocks, BLOCK, lds>>>(m16 * 2 + 1, out); })) return 1;
    snprintf(name, sizeof name, "LDS  4-byte stores, %3d KB table per workgroup", kb);
    if (timed(name, blocks, [&] { k_lds<5, 4><<<blocks, BLOCK, lds>>>(m16 * 4 + 3, out); })) return 1;
  }
  return 0;
}
