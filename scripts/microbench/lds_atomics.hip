// Microbenchmark: LDS atomic throughput on gfx950 (per-CU lane-ops per clock) for the access patterns the
// low-cardinality aggregate kernel can choose from.  Build: hipcc -O3 --offload-arch=gfx950 lds_atomics.hip -o lds_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); return 1; } } while (0)
constexpr int ITERS = 4096;
// MODE 0: u64 add, each lane its own address (stride 8 B)          1: u64 add, 2 lanes per address
//      2: u64 add, 4 lanes per address                             3: u64 add, all lanes one address
//      4: u32 add, own address                                     5: u64 add returning, own address
//      6: plain ds_write_b64 own address                           7: u64 add, own address, 16 lanes active (exec mask)
//      8: read-modify-write non-atomic b64 own address             9: f64 add own address
template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned long long* out, int iters) {
  __shared__ unsigned long long s[256 * 4];
  for (int x = threadIdx.x; x < 1024; x += 256) s[x] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long acc = 0;
  int idx;
  switch (MODE) {
    case 1: idx = wave * 64 + (lane >> 1); break;
    case 2: idx = wave * 64 + (lane >> 2); break;
    case 3: idx = wave * 64; break;
    default: idx = wave * 64 + lane; break;
  }
  for (int it = 0; it < iters; it++) {
    unsigned long long v = (unsigned long long)(it + lane);
#pragma unroll
    for (int u = 0; u < 8; u++) {
      if (MODE == 4) atomicAdd(reinterpret_cast<unsigned int*>(s) + idx + u * 256, (unsigned int)v);
      else if (MODE == 5) acc += atomicAdd(&s[idx + (u & 3) * 256], v);
      else if (MODE == 6) { s[idx + (u & 3) * 256] = v; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); }
      else if (MODE == 7) { if (lane < 16) atomicAdd(&s[idx + (u & 3) * 256], v); }
      else if (MODE == 8) { volatile unsigned long long* p = &s[idx + (u & 3) * 256]; *p = *p + v; }
      else if (MODE == 9) atomicAdd(reinterpret_cast<double*>(&s[idx + (u & 3) * 256]), (double)v);
      else atomicAdd(&s[idx + (u & 3) * 256], v);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = s[0] + s[257] + acc;
}
template <int MODE>
int run(const char* name, unsigned long long* d) {
  const int blocks = 256 * 4;  // 4 workgroups (16 waves) per CU
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  k<MODE><<<blocks, 256>>>(d, 64);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a));
  k<MODE><<<blocks, 256>>>(d, ITERS);
  CHECK(hipEventRecord(b));
  CHECK(hipDeviceSynchronize());
  float ms; CHECK(hipEventElapsedTime(&ms, a, b));
  double lane_ops = (double)blocks * 256 * ITERS * 8 * (MODE == 7 ? 0.25 : 1.0);
  double wave_ops = (double)blocks * 4 * ITERS * 8;
  printf("%-44s %8.3f ms  %7.2f G lane-ops/s  %6.2f lane-ops/clk/CU  %6.1f clk per wave-instr per CU (2.4 GHz)\n", name, ms, lane_ops / ms / 1e6,
         lane_ops / (ms * 1e-3) / 256 / 2.4e9, (ms * 1e-3) * 2.4e9 * 256 / wave_ops);
  return 0;
}
int main() {
  unsigned long long* d;
  CHECK(hipMalloc(&d, 1 << 20));
  run<0>("ds_add_u64 own address", d);
  run<1>("ds_add_u64 2 lanes/address", d);
  run<2>("ds_add_u64 4 lanes/address", d);
  run<3>("ds_add_u64 all lanes one address", d);
  run<4>("ds_add_u32 own address", d);
  run<5>("ds_add_rtn_u64 own address", d);
  run<6>("ds_write_b64 own address", d);
  run<7>("ds_add_u64 own address, 16 of 64 lanes", d);
  run<8>("ds_read+ds_write b64 (non-atomic RMW)", d);
  run<9>("ds_add_f64 own address", d);
  return 0;
}
