// runtime.hip — process-wide runtime of libdfgpu.so: device binding, the library stream, the
// HBM pool allocator, the thread-local error channel and per-kernel HIP-event profiling.
#include "internal.hpp"

#include <algorithm>

namespace dfgpu {

static thread_local std::string g_last_error;
void set_last_error(const std::string& msg) { g_last_error = msg; }

Runtime& rt() {
  static Runtime r;
  return r;
}
void require_init() { DFGPU_CHECK(rt().initialised, "dfgpu_init() has not been called"); }

// ------------------------------------------------------------------------------- pool
// Blocks are rounded to 512 B (small) or 2 MiB (large) so that the large, repeated
// allocations of a query pipeline (column buffers of equal row counts) hit the cache.
static size_t round_size(size_t n) {
  const size_t small = 512, big = size_t(2) << 20;
  if (n <= (size_t(1) << 20)) return (n + small - 1) / small * small;
  return (n + big - 1) / big * big;
}

void* Runtime::alloc(size_t bytes) {
  size_t cap = round_size(bytes ? bytes : 1);
  {
    std::lock_guard<std::mutex> lk(mu);
    auto it = free_blocks.lower_bound(cap);
    // accept a cached block up to 25 % larger than requested
    if (it != free_blocks.end() && it->first <= cap + cap / 4) {
      void* p = it->second;
      size_t c = it->first;
      free_blocks.erase(it);
      cached -= (int64_t)c;
      live[p] = c;
      in_use += (int64_t)c;
      peak = std::max(peak, in_use);
      return p;
    }
  }
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, cap);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    trim();  // give cached blocks back to the driver and retry once
    e = hipMalloc(&p, cap);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      // ResourcesExhausted, as MemoryReservation::try_grow would report (hash_join/exec.rs:2608)
      throw Error("Resources exhausted: failed to allocate " + std::to_string(cap) + " bytes of HBM (" +
                  std::to_string(in_use) + " in use)");
    }
  }
  std::lock_guard<std::mutex> lk(mu);
  live[p] = cap;
  in_use += (int64_t)cap;
  peak = std::max(peak, in_use);
  return p;
}

void Runtime::free(void* p) {
  if (!p) return;
  std::lock_guard<std::mutex> lk(mu);
  auto it = live.find(p);
  if (it == live.end()) return;
  size_t cap = it->second;
  live.erase(it);
  in_use -= (int64_t)cap;
  free_blocks.emplace(cap, p);
  cached += (int64_t)cap;
}

void Runtime::trim() {
  std::multimap<size_t, void*> blocks;
  {
    std::lock_guard<std::mutex> lk(mu);
    blocks.swap(free_blocks);
    cached = 0;
  }
  if (blocks.empty()) return;
  (void)hipStreamSynchronize(stream);
  for (auto& kv : blocks) (void)hipFree(kv.second);
}

DevBuf::DevBuf(size_t n) : ptr(rt().alloc(n)), bytes(n) {}
DevBuf::~DevBuf() { rt().free(ptr); }

BufPtr make_zero_buf(size_t bytes) {
  BufPtr b = make_buf(bytes);
  DFGPU_HIP(hipMemsetAsync(b->ptr, 0, bytes ? bytes : 1, rt().stream));
  return b;
}

void d2h(void* dst, const void* src, size_t n) {
  DFGPU_HIP(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, rt().stream));
  DFGPU_HIP(hipStreamSynchronize(rt().stream));
}
void h2d_async(void* dst, const void* src, size_t n) {
  DFGPU_HIP(hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, rt().stream));
}
uint64_t read_u64(const uint64_t* dev) {
  uint64_t v = 0;
  d2h(&v, dev, 8);
  return v;
}

// -------------------------------------------------------------------------- profiling
ProfileScope::ProfileScope(const char* n, int64_t algorithmic_bytes) : name(n), bytes(algorithmic_bytes) {
  Runtime& r = rt();
  if (!r.profiling) return;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  (void)hipEventRecord(a, r.stream);
}
ProfileScope::~ProfileScope() {
  if (!a) return;
  Runtime& r = rt();
  (void)hipEventRecord(b, r.stream);
  std::lock_guard<std::mutex> lk(r.mu);   // scan threads decode column chunks concurrently (parquet.hip)
  r.recs.push_back({name, a, b, bytes});
}

void Runtime::collect() {
  if (recs.empty()) return;
  (void)hipStreamSynchronize(stream);
  for (auto& rec : recs) {
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, rec.a, rec.b);
    (void)hipEventDestroy(rec.a);
    (void)hipEventDestroy(rec.b);
    auto it = std::find_if(stats.begin(), stats.end(), [&](const dfgpu_kernel_stat& s) { return rec.name == s.name; });
    if (it == stats.end()) {
      dfgpu_kernel_stat s{};
      std::strncpy(s.name, rec.name.c_str(), sizeof(s.name) - 1);
      stats.push_back(s);
      it = stats.end() - 1;
    }
    it->calls += 1;
    it->total_ms += ms;
    it->algorithmic_bytes += rec.bytes;
  }
  recs.clear();
}

std::string type_name(const dfgpu_field& f) {
  switch (f.type) {
    case DFGPU_INT32: return "Int32";
    case DFGPU_INT64: return "Int64";
    case DFGPU_DECIMAL128: return "Decimal128(" + std::to_string(f.precision) + "," + std::to_string(f.scale) + ")";
    case DFGPU_FLOAT64: return "Float64";
    case DFGPU_UINT8: return "UInt8";
    case DFGPU_UINT32: return "UInt32";
    case DFGPU_UINT64: return "UInt64";
    case DFGPU_DATE32: return "Date32";
    case DFGPU_BOOL: return "Boolean";
  }
  return "?";
}

Column alloc_column(const dfgpu_field& f, const std::string& name, int64_t n, bool with_validity) {
  Column c;
  c.field = f;
  c.name = name;
  c.length = n;
  c.data = make_buf(data_bytes(f.type, n) + 16);
  if (with_validity) {
    c.validity = make_buf(bitmap_bytes(n));
    c.null_count = -1;
  }
  return c;
}

}  // namespace dfgpu

using namespace dfgpu;

extern "C" {

int dfgpu_abi_version(void) { return DFGPU_ABI_VERSION; }

const char* dfgpu_last_error(void) { return g_last_error.c_str(); }

int dfgpu_device_count(int* out) {
  return guarded([&] {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      n = 0;
    }
    *out = n;
  });
}

int dfgpu_init(int device) {
  return guarded([&] {
    Runtime& r = rt();
    if (r.initialised) {
      DFGPU_CHECK(r.device == device, "dfgpu_init: already bound to device " + std::to_string(r.device));
      return;
    }
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) {
      (void)hipGetLastError();
      throw Error("dfgpu_init: no HIP device visible (this library has no CPU fallback)");
    }
    DFGPU_CHECK(device >= 0 && device < n, "dfgpu_init: device index out of range");
    DFGPU_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    DFGPU_HIP(hipGetDeviceProperties(&prop, device));
    r.num_cus = prop.multiProcessorCount;
    DFGPU_HIP(hipStreamCreateWithFlags(&r.stream, hipStreamNonBlocking));
    r.device = device;
    r.initialised = true;
  });
}

int dfgpu_shutdown(void) {
  return guarded([&] {
    Runtime& r = rt();
    if (!r.initialised) return;
    r.collect();
    r.trim();
    (void)hipStreamDestroy(r.stream);
    r.stream = nullptr;
    r.initialised = false;
  });
}

int dfgpu_sync(void) {
  return guarded([&] {
    require_init();
    DFGPU_HIP(hipStreamSynchronize(rt().stream));
  });
}

void* dfgpu_stream(void) { return (void*)rt().stream; }

int dfgpu_mem_stats(int64_t* in_use, int64_t* cached, int64_t* peak) {
  return guarded([&] {
    Runtime& r = rt();
    std::lock_guard<std::mutex> lk(r.mu);
    if (in_use) *in_use = r.in_use;
    if (cached) *cached = r.cached;
    if (peak) *peak = r.peak;
  });
}
int dfgpu_mem_trim(void) {
  return guarded([&] { rt().trim(); });
}

int dfgpu_profile_enable(int on) {
  return guarded([&] {
    require_init();
    rt().collect();
    rt().profiling = on != 0;
  });
}
int dfgpu_profile_reset(void) {
  return guarded([&] {
    rt().collect();
    rt().stats.clear();
  });
}
int dfgpu_profile_count(int* out) {
  return guarded([&] {
    rt().collect();
    *out = (int)rt().stats.size();
  });
}
int dfgpu_profile_get(int i, dfgpu_kernel_stat* out) {
  return guarded([&] {
    rt().collect();
    DFGPU_CHECK(i >= 0 && i < (int)rt().stats.size(), "profile index out of range");
    *out = rt().stats[i];
  });
}

}  // extern "C"
