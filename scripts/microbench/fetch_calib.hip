// Microbenchmark: what rocprofv3's FETCH_SIZE / WRITE_SIZE report on gfx950 for the access patterns of the join probe.
// MI355X_MICROARCH.md §HBM: FETCH_SIZE reads 1/2 of the bytes of a WIDE coalesced streaming read (16 B / lane); other
// widths and WRITE_SIZE are "uncalibrated: calibrate on a known byte count in your own access pattern".  Every kernel
// here moves a known number of bytes over buffers far larger than the 256 MiB Infinity Cache:
//   stream4 / stream8 / stream16 : coalesced streaming reads of 4 / 8 / 16 B per lane (probe keys, payload columns)
//   gather4_clustered            : 4-byte reads at src[i / 4] — the build-payload gather of an N:1 join over key-sorted inputs
//   gather4_random               : 4-byte reads at hashed positions — one 4-byte value per touched line
//   gather16_random              : 16-byte reads at hashed positions (Decimal128 payload gathered after a sort)
//   write4 / write16             : coalesced streaming writes
// Build: hipcc -O3 --offload-arch=gfx950 fetch_calib.hip -o fetch_calib ; run under `rocprofv3 --pmc FETCH_SIZE` and
// `--pmc WRITE_SIZE` (separate passes) and divide each kernel's known bytes by counter * 1024 (scripts/fetch_calib.sh).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); return 1; } } while (0)

template <typename T>
__global__ __launch_bounds__(256) void stream_read(const T* __restrict__ src, int64_t n, unsigned long long* sink) {
  unsigned long long acc = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    T v = src[i];
    const unsigned* w = reinterpret_cast<const unsigned*>(&v);
    for (unsigned k = 0; k < sizeof(T) / 4; k++) acc += w[k];
  }
  if (acc == 0x123456789abcdefull) *sink = acc;
}
__global__ __launch_bounds__(256) void gather4_clustered(const unsigned* __restrict__ src, int64_t n, unsigned long long* sink) {
  unsigned long long acc = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) acc += src[i >> 2];
  if (acc == 0x123456789abcdefull) *sink = acc;
}
__device__ __forceinline__ uint64_t mix(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return x;
}
template <typename T>
__global__ __launch_bounds__(256) void gather_random(const T* __restrict__ src, int64_t n_src, int64_t n, unsigned long long* sink) {
  unsigned long long acc = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    T v = src[mix((uint64_t)i) % (uint64_t)n_src];
    acc += *reinterpret_cast<const unsigned*>(&v);
  }
  if (acc == 0x123456789abcdefull) *sink = acc;
}
template <typename T>
__global__ __launch_bounds__(256) void stream_write(T* __restrict__ dst, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    T v;
    unsigned* w = reinterpret_cast<unsigned*>(&v);
    for (unsigned k = 0; k < sizeof(T) / 4; k++) w[k] = (unsigned)i + k;
    dst[i] = v;
  }
}

int main() {
  const int64_t BYTES = int64_t(4) << 30;  // 4 GiB per buffer: 16x the Infinity Cache
  void *a = nullptr, *b = nullptr;
  unsigned long long* sink = nullptr;
  CHECK(hipMalloc(&a, BYTES));
  CHECK(hipMalloc(&b, BYTES));
  CHECK(hipMalloc(&sink, 8));
  CHECK(hipMemset(a, 1, BYTES));
  CHECK(hipMemset(b, 1, BYTES));
  const int G = 256 * 8;
  const int64_t n_gather = int64_t(256) << 20;  // 256 Mi gathered elements
  for (int rep = 0; rep < 3; rep++) {
    stream_read<unsigned><<<G, 256>>>((const unsigned*)a, BYTES / 4, sink);
    stream_read<uint2><<<G, 256>>>((const uint2*)a, BYTES / 8, sink);
    stream_read<uint4><<<G, 256>>>((const uint4*)a, BYTES / 16, sink);
    gather4_clustered<<<G, 256>>>((const unsigned*)a, BYTES, sink);  // BYTES lanes read src[0 .. BYTES/4): BYTES distinct bytes
    gather_random<unsigned><<<G, 256>>>((const unsigned*)a, BYTES / 4, n_gather, sink);
    gather_random<uint4><<<G, 256>>>((const uint4*)a, BYTES / 16, n_gather, sink);
    stream_write<unsigned><<<G, 256>>>((unsigned*)b, BYTES / 4);
    stream_write<uint4><<<G, 256>>>((uint4*)b, BYTES / 16);
  }
  CHECK(hipDeviceSynchronize());
  // known bytes per launch, for scripts/fetch_calib.sh
  const int64_t n_clustered = BYTES;  // elements issued; distinct bytes read = n (one 4-byte source element per 4 lanes)
  printf("KNOWN stream_read<unsigned int> %lld\n", (long long)BYTES);
  printf("KNOWN stream_read<HIP_vector_type<unsigned int, 2u>> %lld\n", (long long)BYTES);
  printf("KNOWN stream_read<HIP_vector_type<unsigned int, 4u>> %lld\n", (long long)BYTES);
  printf("KNOWN gather4_clustered %lld\n", (long long)n_clustered);
  printf("KNOWN_LINES64 gather_random<unsigned int> %lld\n", (long long)n_gather * 64);
  printf("KNOWN_LINES64 gather_random<HIP_vector_type<unsigned int, 4u>> %lld\n", (long long)n_gather * 64);
  printf("KNOWN stream_write<unsigned int> %lld\n", (long long)BYTES);
  printf("KNOWN stream_write<HIP_vector_type<unsigned int, 4u>> %lld\n", (long long)BYTES);
  return 0;
}
