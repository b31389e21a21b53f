// records.hpp — row-major records of several fixed-width columns (shared by filter.hip's packed take and sort.hip's clustered take).
// A random access costs a whole 128-byte line of HBM whatever the element width (profiles/r2_fetch_calib.md), so rows that are
// going to be taken in random order travel as ONE record of 16 / 32 / 48 / 64 bytes instead of one access per column.
#pragma once
#include "device.hpp"

namespace dfgpu {

constexpr int PACK_MAX_COLS = 8;
struct PackLayout {
  const void* src[PACK_MAX_COLS];
  void* dst[PACK_MAX_COLS];
  int width[PACK_MAX_COLS];
  int offset[PACK_MAX_COLS];
  int n;
};
// a record lives in R / 8 registers; fields are placed / extracted with constant-index selects (a runtime-indexed array would
// live in scratch memory), the record itself moves as whole 16-byte loads / stores
template <int NS>
__device__ __forceinline__ uint64_t slot_get(const uint64_t (&s)[NS], int k) {
  uint64_t v = 0;
#pragma unroll
  for (int q = 0; q < NS; q++) v = k == q ? s[q] : v;
  return v;
}
template <int NS>
__device__ __forceinline__ void slot_or(uint64_t (&s)[NS], int k, uint64_t v) {
#pragma unroll
  for (int q = 0; q < NS; q++) s[q] |= k == q ? v : 0ull;
}
// row `i` of the layout's source columns as a record
template <int NS>
__device__ __forceinline__ void record_build(const PackLayout& L, int64_t i, uint64_t (&s)[NS]) {
#pragma unroll
  for (int q = 0; q < NS; q++) s[q] = 0;
  for (int c = 0; c < L.n; c++) {
    const int o = L.offset[c];
    switch (L.width[c]) {
      case 16: {
        const uint4 v = reinterpret_cast<const uint4*>(L.src[c])[i];
        slot_or<NS>(s, o >> 3, ((uint64_t)v.y << 32) | v.x);
        slot_or<NS>(s, (o >> 3) + 1, ((uint64_t)v.w << 32) | v.z);
        break;
      }
      case 8: slot_or<NS>(s, o >> 3, reinterpret_cast<const uint64_t*>(L.src[c])[i]); break;
      case 4: slot_or<NS>(s, o >> 3, (uint64_t)reinterpret_cast<const uint32_t*>(L.src[c])[i] << ((o & 4) * 8)); break;
      default: slot_or<NS>(s, o >> 3, (uint64_t)reinterpret_cast<const uint8_t*>(L.src[c])[i] << ((o & 7) * 8)); break;
    }
  }
}
template <int NS>
__device__ __forceinline__ void record_store(uint8_t* __restrict__ rec, int64_t i, const uint64_t (&s)[NS]) {
#pragma unroll
  for (int q = 0; q < NS / 2; q++)
    reinterpret_cast<uint4*>(rec + i * (NS * 8))[q] = uint4{(unsigned)s[2 * q], (unsigned)(s[2 * q] >> 32), (unsigned)s[2 * q + 1], (unsigned)(s[2 * q + 1] >> 32)};
}
template <int NS>
__device__ __forceinline__ void record_load(const uint8_t* __restrict__ rec, int64_t i, uint64_t (&s)[NS]) {
  const uint4* src = reinterpret_cast<const uint4*>(rec + i * (NS * 8));
#pragma unroll
  for (int q = 0; q < NS / 2; q++) {
    const uint4 v = src[q];
    s[2 * q] = ((uint64_t)v.y << 32) | v.x;
    s[2 * q + 1] = ((uint64_t)v.w << 32) | v.z;
  }
}
// the record's fields to row `i` of the layout's destination columns
template <int NS>
__device__ __forceinline__ void record_split(const PackLayout& L, int64_t i, const uint64_t (&s)[NS]) {
  for (int c = 0; c < L.n; c++) {
    const int o = L.offset[c];
    switch (L.width[c]) {
      case 16: {
        const uint64_t lo = slot_get<NS>(s, o >> 3), hi = slot_get<NS>(s, (o >> 3) + 1);
        reinterpret_cast<uint4*>(L.dst[c])[i] = uint4{(unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32)};
        break;
      }
      case 8: reinterpret_cast<uint64_t*>(L.dst[c])[i] = slot_get<NS>(s, o >> 3); break;
      case 4: reinterpret_cast<uint32_t*>(L.dst[c])[i] = (uint32_t)(slot_get<NS>(s, o >> 3) >> ((o & 4) * 8)); break;
      default: reinterpret_cast<uint8_t*>(L.dst[c])[i] = (uint8_t)(slot_get<NS>(s, o >> 3) >> ((o & 7) * 8)); break;
    }
  }
}

}  // namespace dfgpu
