// onesweep.hpp — the decoupled look-back shared by the onesweep passes of sort.hip and the radix join's partitioner (radix_join.hip):
// per (tile, digit) one 32-bit {status, count} word, published AGG(regate) as soon as the tile's counts are known and PFX (inclusive
// prefix) once its own look-back is done.  Row counts below 2^30 (the status words hold 30-bit prefixes).
#pragma once
#include <cstdint>

#include "device.hpp"

namespace dfgpu {

constexpr uint32_t OS_AGG = 1u << 30, OS_PFX = 2u << 30, OS_VAL = (1u << 30) - 1u;
__device__ __forceinline__ uint32_t os_load(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void os_store(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// The look-back of digit `d` for tile t: the sum of the digit's counts over the tiles before t — nearest first, AGG words added up until
// an inclusive prefix (PFX) is met.  OS_LB predecessors are read at once: an agent-scope load is a round trip to the memory side of the
// fabric (no XCD's L2 may answer it), a microsecond, and tiles retire every few tens of nanoseconds — one word per round trip cannot keep
// up with that and the tiles queue behind their look-backs (measured: 0.6-0.9 ms of a 2-2.6 ms pass; profiles/r5_sort_phases.md).
constexpr int OS_LB = 8;
__device__ __forceinline__ unsigned os_look_back(const uint32_t* tile_state, int64_t t, unsigned d) {
  unsigned excl = 0;
  int64_t p = t - 1;
  bool done = t == 0;
  while (!done) {
    uint32_t st[OS_LB];
#pragma unroll
    for (int u = 0; u < OS_LB; u++) st[u] = p - u >= 0 ? os_load(&tile_state[(p - u) * 256 + d]) : OS_PFX;
    int adv = 0;
    bool stop = false;
#pragma unroll
    for (int u = 0; u < OS_LB; u++) {
      const uint32_t status = st[u] >> 30;
      if (!stop) {
        if (status == 0) {
          stop = true;   // not published yet: read again from here
        } else {
          excl += st[u] & OS_VAL;
          adv++;
          if (status == 2) done = stop = true;
        }
      }
    }
    p -= adv;
    if (!done && adv == 0) __builtin_amdgcn_s_sleep(1);
  }
  return excl;
}

}  // namespace dfgpu
