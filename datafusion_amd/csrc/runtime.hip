// runtime.hip — process-wide runtime of libdfgpu.so: device binding, the library stream, the
// HBM pool allocator, the thread-local error channel and per-kernel HIP-event profiling.
#include "internal.hpp"

#include <chrono>
#include <algorithm>
#include <cstring>
#include <atomic>

namespace dfgpu {

static thread_local std::string g_last_error;
void set_last_error(const std::string& msg) { g_last_error = msg; }

// ------------------------------------------------------------------------------- devices
// g_runtimes[d] exists for every visible device d once dfgpu_init ran; `initialised` says which ones the caller bound.
static std::mutex g_devices_mu;
static std::vector<std::unique_ptr<Runtime>> g_runtimes;
static std::vector<int> g_initialised;         // in dfgpu_init order; [0] is the default device of new threads
static Runtime g_uninitialised;                // what rt() answers before dfgpu_init (initialised == false)
static thread_local int t_device = -1;         // the calling thread's current device (-1: default)
static thread_local int t_hip_device = -1;     // what this thread last passed to hipSetDevice

const std::vector<int>& initialised_devices() { return g_initialised; }
int current_device() {
  if (t_device >= 0) return t_device;
  return g_initialised.empty() ? -1 : g_initialised[0];
}
Runtime& rt_of(int device) {
  DFGPU_CHECK(device >= 0 && device < (int)g_runtimes.size() && g_runtimes[device] && g_runtimes[device]->initialised,
              "device " + std::to_string(device) + " has not been initialised (dfgpu_init)");
  return *g_runtimes[device];
}
void use_device(int device) {
  if (device == t_device && device == t_hip_device) return;
  (void)rt_of(device);
  if (t_hip_device != device) {
    DFGPU_HIP(hipSetDevice(device));
    t_hip_device = device;
  }
  t_device = device;
}
Runtime& rt() {
  const int d = current_device();
  if (d < 0) return g_uninitialised;
  if (t_hip_device != d) {  // HIP's current device is per thread: a fresh host thread would otherwise allocate and launch on device 0
    if (hipSetDevice(d) == hipSuccess) t_hip_device = d;
    else (void)hipGetLastError();
  }
  return *g_runtimes[d];
}
void require_init() { DFGPU_CHECK(rt().initialised, "dfgpu_init() has not been called"); }
// ------------------------------------------------------------------------------- options
static std::mutex g_opt_mu;
static std::map<std::string, std::string> g_options;     // dfgpu_set_option
const Policy& policy() { return rt().policy; }
// "name=value,name=value" of DFGPU_OPTIONS: looked at on every call (a handful of characters), so that a test's monkeypatched
// environment and an operator's shell both work without a restart
static bool env_option(const char* name, std::string& out) {
  const char* e = std::getenv("DFGPU_OPTIONS");
  if (!e || !*e) return false;
  const size_t len = std::strlen(name);
  for (const char* p = e; *p;) {
    const char* end = std::strchr(p, ',');
    const size_t n = end ? (size_t)(end - p) : std::strlen(p);
    if (n > len && std::strncmp(p, name, len) == 0 && p[len] == '=') {
      out.assign(p + len + 1, n - len - 1);
      return true;
    }
    p += n;
    if (*p == ',') p++;
  }
  return false;
}
static bool find_option(const char* name, std::string& out) {
  {
    std::lock_guard<std::mutex> lk(g_opt_mu);
    auto it = g_options.find(name);
    if (it != g_options.end()) {
      out = it->second;
      return true;
    }
  }
  return env_option(name, out);
}
int64_t option_int(const char* name, int64_t dflt) {
  std::string v;
  return find_option(name, v) && !v.empty() ? (int64_t)std::atoll(v.c_str()) : dflt;
}
bool option_on(const char* name, bool dflt) {
  std::string v;
  if (!find_option(name, v) || v.empty()) return dflt;
  return !(v == "0" || v == "off" || v == "false");
}
std::string option_str(const char* name, const char* dflt) {
  std::string v;
  return find_option(name, v) ? v : std::string(dflt ? dflt : "");
}
bool trace_on(const char* what) {
  const char* e = std::getenv("DFGPU_TRACE");
  if (!e || !*e) return false;
  return std::strstr(e, what) != nullptr || std::strstr(e, "all") != nullptr;
}


// ------------------------------------------------------------------------------- threads and streams
// per (host thread, device): the thread's stream and the blocks it freed during the current call
struct ThreadDevice {
  hipStream_t stream = nullptr;
  uint64_t generation = 0;                            // of the runtime the stream was resolved against
  bool borrowed = false;                              // taken from the device's spare list (goes back when the thread ends)
  std::vector<std::pair<size_t, void*>> pending;      // freed by this thread, reusable by it, by everyone after the call's drain
};
struct SpareStreams {                                  // process lifetime (never destroyed: threads may end after static destructors ran)
  std::mutex mu;
  std::map<int, std::vector<hipStream_t>> by_device;
};
static SpareStreams& spare_streams() {
  static SpareStreams* s = new SpareStreams();
  return *s;
}
static std::atomic<bool> g_multi_thread{false};        // a second host thread has driven the library
static std::atomic<std::thread::id> g_first_thread{};
struct ThreadState {
  std::vector<ThreadDevice> dev;                       // indexed by device id
  ~ThreadState() {
    for (size_t d = 0; d < dev.size(); d++) {
      ThreadDevice& td = dev[d];
      if (!td.pending.empty() && d < g_runtimes.size() && g_runtimes[d]) {   // (a thread that ends outside a call: its blocks go home)
        if (td.stream) (void)hipStreamSynchronize(td.stream);
        std::lock_guard<std::mutex> lk(g_runtimes[d]->mu);
        for (auto& b : td.pending) {
          g_runtimes[d]->free_blocks.emplace(b.first, b.second);
          g_runtimes[d]->cached += (int64_t)b.first;
        }
      }
      if (td.stream && td.borrowed) {
        std::lock_guard<std::mutex> lk(spare_streams().mu);
        spare_streams().by_device[(int)d].push_back(td.stream);
      }
    }
  }
};
static thread_local ThreadState t_state;
static ThreadDevice& thread_device(int device) {
  if ((int)t_state.dev.size() <= device) t_state.dev.resize((size_t)device + 1);
  return t_state.dev[(size_t)device];
}

StreamRef::operator hipStream_t() const { return r ? r->thread_stream() : nullptr; }

hipStream_t Runtime::thread_stream() {
  if (!initialised) return nullptr;
  ThreadDevice& td = thread_device(device);
  if (td.stream && td.generation == generation) return td.stream;
  if (td.stream && !td.borrowed) td.stream = nullptr;   // the first stream of an earlier dfgpu_init: gone with its dfgpu_shutdown
  td.generation = generation;
  if (td.stream) return td.stream;                      // (a borrowed stream outlives shutdown / init)
  const std::thread::id me = std::this_thread::get_id();
  std::thread::id none{};
  if (g_first_thread.compare_exchange_strong(none, me) || g_first_thread.load() == me) {
    td.stream = first_stream;   // the thread that came first keeps the device's first stream
    return td.stream;
  }
  // a further thread: from now on calls drain their streams before they return.  Whatever the other threads had in flight when
  // this one arrived — including work on blocks that already went back to the pool — is waited for once, here.
  if (!g_multi_thread.exchange(true)) (void)hipDeviceSynchronize();
  {
    std::lock_guard<std::mutex> lk(spare_streams().mu);
    auto& spare = spare_streams().by_device[device];
    if (!spare.empty()) {
      td.stream = spare.back();
      spare.pop_back();
    }
  }
  if (!td.stream) DFGPU_HIP(hipStreamCreateWithFlags(&td.stream, hipStreamNonBlocking));
  td.borrowed = true;
  return td.stream;
}

void call_epilogue() noexcept {
  if (!g_multi_thread.load(std::memory_order_relaxed)) return;
  for (size_t d = 0; d < t_state.dev.size(); d++) {
    ThreadDevice& td = t_state.dev[d];
    if (!td.stream) continue;
    if (d >= g_runtimes.size() || !g_runtimes[d]) continue;
    if (hipStreamSynchronize(td.stream) != hipSuccess) (void)hipGetLastError();
    if (td.pending.empty()) continue;
    Runtime& r = *g_runtimes[d];
    std::lock_guard<std::mutex> lk(r.mu);
    for (auto& b : td.pending) {
      r.free_blocks.emplace(b.first, b.second);
      r.cached += (int64_t)b.first;
    }
    td.pending.clear();
  }
}

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
  if (g_multi_thread.load(std::memory_order_relaxed)) {
    // a block this thread freed earlier in the call: reusable in its own stream's order
    ThreadDevice& td = thread_device(device);
    size_t best = td.pending.size();
    for (size_t i = 0; i < td.pending.size(); i++)
      if (td.pending[i].first >= cap && td.pending[i].first <= cap + cap / 4 && (best == td.pending.size() || td.pending[i].first < td.pending[best].first)) best = i;
    if (best < td.pending.size()) {
      const size_t c = td.pending[best].first;
      void* p = td.pending[best].second;
      td.pending.erase(td.pending.begin() + (long)best);
      std::lock_guard<std::mutex> lk(mu);
      live[p] = c;
      in_use += (int64_t)c;
      peak = std::max(peak, in_use);
      return p;
    }
  }
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
  const auto t_driver = std::chrono::steady_clock::now();
  hipError_t e = hipMalloc(&p, cap);
  driver_allocs.fetch_add(1, std::memory_order_relaxed);
  driver_alloc_ns.fetch_add(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_driver).count(), std::memory_order_relaxed);
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
  size_t cap;
  {
    std::lock_guard<std::mutex> lk(mu);
    auto it = live.find(p);
    if (it == live.end()) return;
    cap = it->second;
    live.erase(it);
    in_use -= (int64_t)cap;
    if (!g_multi_thread.load(std::memory_order_seq_cst)) {
      free_blocks.emplace(cap, p);
      cached += (int64_t)cap;
      return;
    }
  }
  thread_device(device).pending.emplace_back(cap, p);   // everybody's after this thread's call has drained its stream
}

void Runtime::trim() {
  scan_upload_caches_drop(device);   // the Parquet scan workers' upload blocks are pool blocks on loan: back first, then the pool itself
  call_epilogue();                   // (several threads: what this thread just freed waits on its pending list until its stream has drained)
  std::multimap<size_t, void*> blocks;
  {
    std::lock_guard<std::mutex> lk(mu);
    blocks.swap(free_blocks);
    cached = 0;
  }
  if (blocks.empty()) return;
  (void)hipDeviceSynchronize();   // every thread's stream: a cached block may have been freed on any of them
  for (auto& kv : blocks) (void)hipFree(kv.second);
}

DevBuf::DevBuf(size_t n) : bytes(n), owner(&rt()) { ptr = owner->alloc(n); }
DevBuf::~DevBuf() {
  if (owner) owner->free(ptr);  // foreign memory (owner == nullptr) goes with `foreign`
}

BufPtr make_zero_buf(size_t bytes) {
  BufPtr b = make_buf(bytes);
  DFGPU_HIP(hipMemsetAsync(b->ptr, 0, bytes ? bytes : 1, rt().stream));
  return b;
}

// Small read-backs (a row count, a flag, 256 histogram bins) land in a pinned block of the calling thread first: a copy into PAGEABLE
// memory makes the runtime stage it through a buffer of its own (a blit kernel + a wait: ~40 us per read-back on this platform, and a
// plan step has a dozen of them — Q3 at SF300 spent 1.3 ms of 10.7 between kernels); into pinned memory it is one DMA and the wait.
void d2h(void* dst, const void* src, size_t n) {
  constexpr size_t PIN = 64 << 10;
  static thread_local void* t_pin = nullptr;   // (never freed: a thread's scratch, 64 KB)
  if (n <= PIN && option_on("runtime.pinned_readback", true)) {
    if (!t_pin && hipHostMalloc(&t_pin, PIN, hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError();
      t_pin = nullptr;
    }
    if (t_pin) {
      DFGPU_HIP(hipMemcpyAsync(t_pin, src, n, hipMemcpyDeviceToHost, rt().stream));
      DFGPU_HIP(hipStreamSynchronize(rt().stream));
      std::memcpy(dst, t_pin, n);
      return;
    }
  }
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
dfgpu_metrics& thread_metrics() {
  static thread_local dfgpu_metrics m{};
  return m;
}

ProfileScope::ProfileScope(const char* n, int64_t algorithmic_bytes) : name(n), bytes(algorithmic_bytes) {
  Runtime& r = rt();
  thread_metrics().hbm_bytes_algorithmic += algorithmic_bytes;
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
  r.recs.push_back({name, a, b, bytes, std::this_thread::get_id()});
}

void Runtime::collect() {
  if (recs.empty()) return;
  (void)hipDeviceSynchronize();   // the events were recorded on their threads' streams
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
    if (launches.size() >= 65536) launches.erase(launches.begin(), launches.begin() + 32768);
    launches.push_back(Launch{(int)(it - stats.begin()), ms, rec.bytes});
    kernel_ns_by_thread[rec.thread] += (int64_t)((double)ms * 1e6);
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
  if (f.type == DFGPU_UTF8) {  // n empty strings (row-selecting operators build string columns themselves, strings.hip)
    c.data = make_buf(16);
    c.offsets = make_zero_buf((size_t)(n + 1) * 8);
  } else {
    c.data = make_buf(data_bytes(f.type, n) + 16);
  }
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
int dfgpu_set_option(const char* name, const char* value) {
  return guarded([&] {
    std::lock_guard<std::mutex> lk(g_opt_mu);
    if (!name) {   // every option back to its default
      g_options.clear();
      return;
    }
    if (!value) g_options.erase(name);
    else g_options[name] = value;
  });
}


int dfgpu_metrics_reset(void) {
  thread_metrics() = dfgpu_metrics{};
  return 0;
}
int dfgpu_metrics_get(dfgpu_metrics* out) {
  if (!out) return 1;
  if (current_device() >= 0) {  // device time of this thread's profiled launches (drains the stream: events must have completed)
    Runtime& r = rt();
    std::lock_guard<std::mutex> lk(r.mu);
    if (r.profiling) r.collect();
    auto it = r.kernel_ns_by_thread.find(std::this_thread::get_id());
    if (it != r.kernel_ns_by_thread.end()) {
      thread_metrics().kernel_ns += it->second;
      r.kernel_ns_by_thread.erase(it);
    }
  }
  *out = thread_metrics();
  return 0;
}

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

int dfgpu_init(const int* device_ids, int n_devices) {
  return guarded([&] {
    DFGPU_CHECK(device_ids != nullptr && n_devices >= 1, "dfgpu_init: give at least one device id");
    std::lock_guard<std::mutex> lk(g_devices_mu);
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) {
      (void)hipGetLastError();
      throw Error("dfgpu_init: no HIP device visible (this library has no CPU fallback)");
    }
    if ((int)g_runtimes.size() < n) g_runtimes.resize(n);
    for (int i = 0; i < n_devices; i++) {
      const int device = device_ids[i];
      DFGPU_CHECK(device >= 0 && device < n, "dfgpu_init: device index out of range");
      if (g_runtimes[device] && g_runtimes[device]->initialised) continue;  // idempotent per device
      DFGPU_HIP(hipSetDevice(device));
      t_hip_device = device;
      auto r = std::make_unique<Runtime>();
      hipDeviceProp_t prop;
      DFGPU_HIP(hipGetDeviceProperties(&prop, device));
      r->num_cus = prop.multiProcessorCount;
      r->policy.num_cus = prop.multiProcessorCount;
      if (prop.l2CacheSize > 0) r->policy.l2_bytes = (size_t)prop.l2CacheSize;
      if (prop.maxSharedMemoryPerMultiProcessor > 0) r->policy.lds_per_cu = (size_t)prop.maxSharedMemoryPerMultiProcessor;
      r->policy.xcds = std::max(1, prop.multiProcessorCount / 32);   // CDNA3 / CDNA4: 32 CUs per XCD
      DFGPU_HIP(hipStreamCreateWithFlags(&r->first_stream, hipStreamNonBlocking));
      r->stream.r = r.get();
      static std::atomic<uint64_t> generations{0};
      r->generation = ++generations;
      r->device = device;
      r->initialised = true;
      g_runtimes[device] = std::move(r);
      g_initialised.push_back(device);
    }
    // devices of one process reach each other's HBM directly (xGMI): dfgpu_table_copy_to_device, single-process exchanges
    for (int a : g_initialised)
      for (int b : g_initialised) {
        if (a == b) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, a, b) == hipSuccess && can) {
          (void)hipSetDevice(a);
          hipError_t pe = hipDeviceEnablePeerAccess(b, 0);
          if (pe != hipSuccess) (void)hipGetLastError();  // already enabled
        }
      }
    if (t_device < 0) t_device = g_initialised[0];
    DFGPU_HIP(hipSetDevice(t_device));
    t_hip_device = t_device;
  });
}

int dfgpu_set_device(int device) {
  return guarded([&] { use_device(device); });
}
int dfgpu_get_device(int* out) {
  return guarded([&] {
    require_init();
    *out = current_device();
  });
}

int dfgpu_shutdown(void) {
  return guarded([&] {
    std::lock_guard<std::mutex> lk(g_devices_mu);
    for (int d : g_initialised) {
      Runtime& r = *g_runtimes[d];
      (void)hipSetDevice(d);
      r.collect();
      r.trim();
      r.initialised = false;   // (thread_stream() answers null from here on)
      (void)hipStreamDestroy(r.first_stream);
      r.first_stream = nullptr;
      {
        std::lock_guard<std::mutex> sl(spare_streams().mu);
        for (hipStream_t s : spare_streams().by_device[d]) (void)hipStreamDestroy(s);
        spare_streams().by_device[d].clear();
      }
    }
    g_initialised.clear();
    t_device = t_hip_device = -1;
  });
}

int dfgpu_sync(void) {
  return guarded([&] {
    require_init();
    if (g_multi_thread.load()) DFGPU_HIP(hipDeviceSynchronize());   // every thread's stream
    else DFGPU_HIP(hipStreamSynchronize(rt().stream));
  });
}

void* dfgpu_stream(void) { return (void*)(hipStream_t)rt().stream; }

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

// ---- admission control: MemoryPool / MemoryReservation (execution/src/memory_pool/mod.rs:188 try_grow) for the device pool
namespace {
struct Reservation {
  int device;
  int64_t bytes;
};
std::mutex g_res_mu;
std::map<int, int64_t> g_reserved;  // device -> outstanding reservations
std::map<int, int64_t> g_limit;     // device -> limit (0 / absent: 92 % of the device's memory)
int64_t limit_of(int device) {
  auto it = g_limit.find(device);
  if (it != g_limit.end() && it->second > 0) return it->second;
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
    (void)hipGetLastError();
    return INT64_MAX;
  }
  return (int64_t)((double)total_b * 0.92);
}
}  // namespace

int dfgpu_mem_set_limit(int64_t bytes) {
  return guarded([&] {
    require_init();
    std::lock_guard<std::mutex> lk(g_res_mu);
    g_limit[current_device()] = bytes;
  });
}
int dfgpu_mem_limit(int64_t* limit, int64_t* reserved) {
  return guarded([&] {
    require_init();
    std::lock_guard<std::mutex> lk(g_res_mu);
    if (limit) *limit = limit_of(current_device());
    if (reserved) *reserved = g_reserved[current_device()];
  });
}
int dfgpu_mem_try_reserve(int64_t bytes, dfgpu_reservation_t* out) {
  return guarded([&] {
    require_init();
    DFGPU_CHECK(out != nullptr && bytes >= 0, "bad argument");
    Runtime& r = rt();
    int64_t in_use = 0;
    {
      std::lock_guard<std::mutex> lk(r.mu);
      in_use = r.in_use;
    }
    std::lock_guard<std::mutex> lk(g_res_mu);
    const int64_t lim = limit_of(r.device), held = g_reserved[r.device];
    if (in_use + held + bytes > lim)
      throw Error("Resources exhausted: Failed to allocate additional " + std::to_string(bytes) + " bytes of HBM with " + std::to_string(in_use) +
                  " bytes in tables and " + std::to_string(held) + " bytes already reserved - maximum available is " + std::to_string(lim));
    g_reserved[r.device] = held + bytes;
    *out = reinterpret_cast<dfgpu_reservation_t>(new Reservation{r.device, bytes});
  });
}
int dfgpu_mem_reservation_size(dfgpu_reservation_t h, int64_t* out) {
  return guarded([&] {
    DFGPU_CHECK(h && out, "null argument");
    *out = reinterpret_cast<Reservation*>(h)->bytes;
  });
}
int dfgpu_mem_release(dfgpu_reservation_t h) {
  return guarded([&] {
    if (!h) return;
    std::unique_ptr<Reservation> res(reinterpret_cast<Reservation*>(h));
    std::lock_guard<std::mutex> lk(g_res_mu);
    g_reserved[res->device] -= res->bytes;
  });
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
    rt().launches.clear();
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

int dfgpu_profile_launches(const char* name, int64_t capacity, double* ms, int64_t* bytes, int64_t* out_n) {
  return guarded([&] {
    DFGPU_CHECK(name && out_n && capacity >= 0 && (capacity == 0 || (ms && bytes)), "null argument");
    Runtime& r = rt();
    r.collect();
    int64_t n = 0;
    for (const Runtime::Launch& l : r.launches) {
      if (std::strcmp(r.stats[(size_t)l.stat].name, name) != 0) continue;
      if (n < capacity) {
        ms[n] = l.ms;
        bytes[n] = l.bytes;
      }
      n++;
    }
    *out_n = n;
  });
}

}  // extern "C"
