// table.hip — device tables (Arrow-layout columns in HBM) and the Arrow C Data Interface
// boundary: import (host RecordBatch -> HBM, pinned with hipHostRegister + async copies on a
// side stream) and export (HBM -> host struct array with release callbacks).
#include <chrono>
#include <algorithm>

#include "device.hpp"
#include "internal.hpp"

#include <cstdio>
#include <cstdlib>
#include <map>

namespace dfgpu {

// dst bits [off, off + n) |= src bits [0, n) (src null = all ones); dst starts zeroed; one thread per destination word
__global__ __launch_bounds__(BLOCK) void k_bitmap_place(const uint64_t* __restrict__ src, int64_t off, int64_t n, unsigned long long* __restrict__ dst) {
  const int64_t w0 = off >> 6, w1 = (off + n + 63) >> 6;
  for (int64_t w = w0 + (int64_t)blockIdx.x * BLOCK + threadIdx.x; w < w1; w += (int64_t)gridDim.x * BLOCK) {
    const int64_t first = max((w << 6), off), last = min((w << 6) + 64, off + n);   // destination bit range of this word
    if (first >= last) continue;
    const int64_t s = first - off;                        // source bit of `first`
    const int len = (int)(last - first);
    uint64_t v;
    if (!src) {
      v = ~0ull;
    } else {
      const int64_t sw = s >> 6;
      const int sb = (int)(s & 63);
      v = src[sw] >> sb;
      if (sb && sb + len > 64) v |= src[sw + 1] << (64 - sb);
    }
    if (len < 64) v &= (1ull << len) - 1ull;
    atomicOr(&dst[w], (unsigned long long)(v << (first & 63)));
  }
}

// dst bits [0, n) = src bits [off, off + n); one thread per destination word, padding bits of the last word cleared
__global__ __launch_bounds__(BLOCK) void k_bitmap_extract(const uint64_t* __restrict__ src, int64_t off, int64_t n, uint64_t* __restrict__ dst) {
  const int64_t nw = (n + 63) >> 6;
  for (int64_t w = (int64_t)blockIdx.x * BLOCK + threadIdx.x; w < nw; w += (int64_t)gridDim.x * BLOCK) {
    const int64_t s = off + (w << 6);
    const int sb = (int)(s & 63);
    const int64_t last_src_word = (off + n - 1) >> 6;
    uint64_t v = src[s >> 6] >> sb;
    if (sb && (s >> 6) + 1 <= last_src_word) v |= src[(s >> 6) + 1] << (64 - sb);
    const int64_t rem = n - (w << 6);
    if (rem < 64) v &= (1ull << rem) - 1ull;
    dst[w] = v;
  }
}

// dictionary indices of one concat part rewritten into the merged dictionary: out[i] = remap[in[i]]; the slot under a NULL row
// may hold anything (Arrow leaves it undefined): an index outside the part's dictionary becomes 0
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_remap_indices(const T* __restrict__ in, const int32_t* __restrict__ remap, int64_t n_remap, int64_t n, T* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    const uint64_t k = (uint64_t)in[i];
    out[i] = k < (uint64_t)n_remap ? (T)remap[k] : (T)0;
  }
}

// ---------------------------------------------------------------- format strings
dfgpu_field parse_format(const char* fmt, bool nullable) {
  dfgpu_field f{};
  f.nullable = nullable ? 1 : 0;
  std::string s(fmt);
  if (s == "i") f.type = DFGPU_INT32;
  else if (s == "l") f.type = DFGPU_INT64;
  else if (s == "g") f.type = DFGPU_FLOAT64;
  else if (s == "C") f.type = DFGPU_UINT8;
  else if (s == "I") f.type = DFGPU_UINT32;
  else if (s == "L") f.type = DFGPU_UINT64;
  else if (s == "tdD") f.type = DFGPU_DATE32;
  else if (s == "b") f.type = DFGPU_BOOL;
  else if (s == "u" || s == "U" || s == "vu") f.type = DFGPU_UTF8;  // Utf8 / LargeUtf8 / Utf8View: offsets + bytes in HBM (strings.hip)
  else if (s.rfind("d:", 0) == 0) {
    int p = 0, sc = 0, bits = 128;
    int n = std::sscanf(s.c_str(), "d:%d,%d,%d", &p, &sc, &bits);
    DFGPU_CHECK(n >= 2 && bits == 128, "unsupported decimal format '" + s + "' (only Decimal128)");
    f.type = DFGPU_DECIMAL128;
    f.precision = p;
    f.scale = sc;
  } else {
    // nested / binary / temporal types other than Date32: the GPU rule leaves such operators on the CPU
    throw Error("unsupported Arrow type format '" + s + "' for GPU execution");
  }
  return f;
}
std::string format_of(const dfgpu_field& f) {
  switch (f.type) {
    case DFGPU_INT32: return "i";
    case DFGPU_INT64: return "l";
    case DFGPU_FLOAT64: return "g";
    case DFGPU_UINT8: return "C";
    case DFGPU_UINT32: return "I";
    case DFGPU_UINT64: return "L";
    case DFGPU_DATE32: return "tdD";
    case DFGPU_BOOL: return "b";
    case DFGPU_DECIMAL128: return "d:" + std::to_string(f.precision) + "," + std::to_string(f.scale);
    case DFGPU_UTF8: return "u";
  }
  throw Error("format_of: bad type");
}

// ---------------------------------------------------------------- import
struct Uploader {
  hipStream_t copy_stream = nullptr;
  std::vector<void*> registered;
  std::vector<void*> staged;  // host temporaries to free after the copies
  Uploader() {
    // the destination blocks come from the pool in the calling thread's stream order: whatever that stream still has in flight on a
    // recycled block must be done before the side stream writes into it
    DFGPU_HIP(hipStreamSynchronize(rt().stream));
    DFGPU_HIP(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
  }
  void upload(void* dst, const void* src, size_t n) {
    if (n == 0) return;
    thread_metrics().h2d_bytes += (int64_t)n;
    // pin large host buffers so the copy engine streams at PCIe rate without a bounce buffer
    if (n >= (size_t(1) << 20)) {   // (from 1 MiB: a pageable copy of that size makes the driver pin the pages itself, and it keeps track of them afterwards)
      if (hipHostRegister(const_cast<void*>(src), n, hipHostRegisterDefault) == hipSuccess) registered.push_back(const_cast<void*>(src));
      else (void)hipGetLastError();
    }
    DFGPU_HIP(hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, copy_stream));
  }
  void finish() {
    DFGPU_HIP(hipStreamSynchronize(copy_stream));
    // (two columns may share pages of one host allocation: the second unregister of such a page fails, harmlessly — but HIP keeps
    // the error as the thread's LAST error, and the next launch check of an unrelated operator would report it)
    for (void* p : registered)
      if (hipHostUnregister(p) != hipSuccess) (void)hipGetLastError();
    for (void* p : staged) std::free(p);
    registered.clear();
    staged.clear();
  }
  ~Uploader() {
    if (copy_stream) {
      (void)hipStreamSynchronize(copy_stream);
      for (void* p : registered)
        if (hipHostUnregister(p) != hipSuccess) (void)hipGetLastError();
      for (void* p : staged) std::free(p);
      (void)hipStreamDestroy(copy_stream);
    }
  }
};

// copy `n` bits starting at bit `offset` of src into a fresh word-padded host bitmap
static uint64_t* shift_bitmap(const uint8_t* src, int64_t offset, int64_t n) {
  size_t bytes = bitmap_bytes(n);
  uint64_t* out = (uint64_t*)std::calloc(bytes ? bytes : 8, 1);
  uint8_t* o = (uint8_t*)out;
  if ((offset & 7) == 0) {
    std::memcpy(o, src + (offset >> 3), (size_t)((n + 7) / 8));
  } else {
    for (int64_t i = 0; i < n; i++) {
      int64_t s = offset + i;
      if ((src[s >> 3] >> (s & 7)) & 1) o[i >> 3] |= (uint8_t)(1u << (i & 7));
    }
  }
  // clear padding bits beyond n
  if (n & 63) out[n >> 6] &= (~0ull) >> (64 - (n & 63));
  return out;
}

// strings of a Utf8 / LargeUtf8 dictionary (host side)
static std::shared_ptr<const DictValues> import_dictionary(const ArrowArray* d, const ArrowSchema* ds, const char* index_format, const char* col_name) {
  const std::string vf(ds->format);
  DFGPU_CHECK(vf == "u" || vf == "U", std::string("dictionary column '") + col_name + "': only Utf8 / LargeUtf8 values are supported, got '" + vf + "'");
  const std::string xf(index_format);
  DFGPU_CHECK(xf == "C" || xf == "i" || xf == "I" || xf == "l" || xf == "L",
              std::string("dictionary column '") + col_name + "': index type '" + xf + "' has no device type (cast the indices to UInt8 / Int32)");
  DFGPU_CHECK(d && d->n_buffers == 3, "malformed dictionary array");
  auto dv = std::make_shared<DictValues>();
  dv->index_format = xf;
  dv->value_format = vf;
  const uint8_t* validity = (const uint8_t*)d->buffers[0];
  const char* data = (const char*)d->buffers[2];
  bool sorted = true;
  std::map<std::string, int> seen;
  for (int64_t i = 0; i < d->length; i++) {
    const int64_t k = d->offset + i;
    int64_t b, e;
    if (vf == "u") { b = ((const int32_t*)d->buffers[1])[k]; e = ((const int32_t*)d->buffers[1])[k + 1]; }
    else { b = ((const int64_t*)d->buffers[1])[k]; e = ((const int64_t*)d->buffers[1])[k + 1]; }
    const bool ok = !validity || ((validity[k >> 3] >> (k & 7)) & 1);
    dv->values.emplace_back(ok && data ? std::string(data + b, (size_t)(e - b)) : std::string());
    dv->valid.push_back(ok ? 1 : 0);
    if (ok) {
      // grouping / joining on the indices is grouping on the strings only if every string has ONE index
      DFGPU_CHECK(seen.emplace(dv->values.back(), 1).second, std::string("dictionary column '") + col_name + "': duplicate dictionary value '" + dv->values.back() + "'");
    }
    if (i > 0 && !(dv->values[(size_t)i - 1] < dv->values[(size_t)i])) sorted = false;
    if (!ok) sorted = false;
  }
  dv->sorted = sorted;
  return dv;
}

static void import_validity(Column& c, const uint8_t* validity, int64_t null_count, int64_t off, int64_t nrows, Uploader& up) {
  if (!(validity && null_count != 0 && nrows)) return;
  uint64_t* h = shift_bitmap(validity, off, nrows);
  int64_t valid = 0;
  for (size_t wd = 0; wd < bitmap_bytes(nrows) / 8; wd++) valid += __builtin_popcountll(h[wd]);
  if (valid != nrows) {
    c.validity = make_buf(bitmap_bytes(nrows));
    c.null_count = nrows - valid;
    up.staged.push_back(h);
    up.upload(c.validity->ptr, h, bitmap_bytes(nrows));
  } else {
    std::free(h);
  }
}
// Utf8 ("u": 32-bit offsets), LargeUtf8 ("U": 64-bit offsets), Utf8View ("vu": 16-byte views over variadic buffers) ->
// 64-bit offsets starting at 0 + contiguous bytes.  Contiguous layouts upload their byte range as it is; views are first
// laid out contiguously on the host (the CPU-side half of a scan: arrow-rs would `gc()` them the same way).
static Column import_string_column(const ArrowArray* a, const ArrowSchema* s, Column c, int64_t parent_offset, int64_t nrows, Uploader& up) {
  const std::string fmt(s->format);
  const int64_t off = a->offset + parent_offset;
  int64_t* ho = (int64_t*)std::malloc((size_t)(nrows + 1) * 8);
  up.staged.push_back(ho);
  const uint8_t* validity = (const uint8_t*)a->buffers[0];
  const char* bytes = nullptr;
  int64_t total = 0;
  if (fmt == "vu") {
    DFGPU_CHECK(a->n_buffers >= 3, "Utf8View: expected views + variadic buffers + sizes");
    struct View { int32_t len; char inl[12]; };
    const View* v = (const View*)a->buffers[1] + off;
    ho[0] = 0;
    for (int64_t i = 0; i < nrows; i++) {
      const bool ok = !validity || ((validity[(off + i) >> 3] >> ((off + i) & 7)) & 1);
      ho[i + 1] = ho[i] + (ok ? v[i].len : 0);
    }
    total = ho[nrows];
    char* hb = (char*)std::malloc((size_t)total + 8);
    up.staged.push_back(hb);
    for (int64_t i = 0; i < nrows; i++) {
      const int64_t len = ho[i + 1] - ho[i];
      if (len == 0) continue;
      if (len <= 12) std::memcpy(hb + ho[i], v[i].inl, (size_t)len);
      else {
        int32_t buf, boff;
        std::memcpy(&buf, v[i].inl + 4, 4);
        std::memcpy(&boff, v[i].inl + 8, 4);
        std::memcpy(hb + ho[i], (const char*)a->buffers[2 + buf] + boff, (size_t)len);
      }
    }
    bytes = hb;
  } else {
    DFGPU_CHECK(a->n_buffers == 3, "Utf8: expected validity + offsets + data buffers");
    int64_t first = 0;
    ho[0] = 0;
    if (nrows && fmt == "u") {
      const int32_t* o = (const int32_t*)a->buffers[1] + off;
      first = o[0];
      for (int64_t i = 0; i <= nrows; i++) ho[i] = (int64_t)o[i] - first;
    } else if (nrows) {
      const int64_t* o = (const int64_t*)a->buffers[1] + off;
      first = o[0];
      for (int64_t i = 0; i <= nrows; i++) ho[i] = o[i] - first;
    }
    total = ho[nrows];
    bytes = (const char*)a->buffers[2] + first;
  }
  c.offsets = make_buf((size_t)(nrows + 1) * 8 + 16);
  up.upload(c.offsets->ptr, ho, (size_t)(nrows + 1) * 8);
  c.data = make_buf((size_t)total + 16);
  if (total) up.upload(c.data->ptr, bytes, (size_t)total);
  import_validity(c, validity, a->null_count, off, nrows, up);
  return c;
}

static Column import_column(const ArrowArray* a, const ArrowSchema* s, int64_t parent_offset, int64_t nrows, Uploader& up) {
  Column c;
  c.field = parse_format(s->format, (s->flags & 2) != 0);
  if (s->dictionary) c.dict = import_dictionary(a->dictionary, s->dictionary, s->format, s->name ? s->name : "");
  c.name = s->name ? s->name : "";
  c.length = nrows;
  if (c.field.type == DFGPU_UTF8 && !s->dictionary) return import_string_column(a, s, c, parent_offset, nrows, up);
  DFGPU_CHECK(a->n_buffers == 2, "expected 2 buffers for a fixed-width column");
  int64_t off = a->offset + parent_offset;
  DFGPU_CHECK(a->length >= nrows + parent_offset || a->length == nrows, "child array shorter than struct");
  const uint8_t* validity = (const uint8_t*)a->buffers[0];
  const uint8_t* data = (const uint8_t*)a->buffers[1];
  if (c.field.type == DFGPU_BOOL) {
    c.data = make_buf(bitmap_bytes(nrows) + 16);
    if (nrows) {
      uint64_t* h = shift_bitmap(data, off, nrows);
      up.staged.push_back(h);
      up.upload(c.data->ptr, h, bitmap_bytes(nrows));
    }
  } else {
    int w = type_width(c.field.type);
    c.data = make_buf((size_t)nrows * w + 16);
    if (nrows) up.upload(c.data->ptr, data + (size_t)off * w, (size_t)nrows * w);
  }
  if (validity && a->null_count != 0 && nrows) {
    uint64_t* h = shift_bitmap(validity, off, nrows);
    int64_t valid = 0;
    for (size_t wd = 0; wd < bitmap_bytes(nrows) / 8; wd++) valid += __builtin_popcountll(h[wd]);
    if (valid != nrows) {
      c.validity = make_buf(bitmap_bytes(nrows));
      c.null_count = nrows - valid;
      up.staged.push_back(h);
      up.upload(c.validity->ptr, h, bitmap_bytes(nrows));
    } else {
      std::free(h);
    }
  }
  return c;
}

// device -> host copy of an export on the library stream, counted in the calling thread's metrics
static void export_copy(void* dst, const void* src, size_t n) {
  thread_metrics().d2h_bytes += (int64_t)n;
  DFGPU_HIP(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, rt().stream));
}

// ---------------------------------------------------------------- export
// Exported buffers are PINNED host memory from a process-wide pool: a device-to-host copy into pageable memory goes through
// the driver's bounce buffers (measured 6.7 GB/s, profiles/r1_ops_v5.md); into pinned memory it is one DMA at PCIe rate.
// hipHostMalloc is slow (it pins pages), so blocks are cached by size and come back when the consumer releases the batch —
// a stream of equally sized output batches (LimitedBatchCoalescer's fixed target, coalesce/mod.rs:27-120) reuses them.
std::atomic<int64_t> g_pinned_driver_allocs{0}, g_pinned_driver_ns{0};   // pool misses: hipHostMalloc calls and their host time
struct PinnedPool {
  std::mutex mu;
  std::multimap<size_t, void*> free_blocks;
  std::map<void*, size_t> live;
  size_t cached = 0;
  static size_t cap() {
    static const size_t v = (size_t)8 << 30;
    return v;
  }
  void* alloc(size_t n) {
    size_t c = 4096;
    while (c < n) c <<= 1;
    if (c > ((size_t)64 << 20)) c = (n + ((size_t)16 << 20) - 1) / ((size_t)16 << 20) * ((size_t)16 << 20);
    {
      std::lock_guard<std::mutex> lk(mu);
      auto it = free_blocks.lower_bound(c);
      if (it != free_blocks.end() && it->first <= c + c / 4) {
        void* p = it->second;
        cached -= it->first;
        live[p] = it->first;
        free_blocks.erase(it);
        return p;
      }
    }
    void* p = nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    DFGPU_HIP(hipHostMalloc(&p, c, hipHostMallocDefault));
    g_pinned_driver_allocs.fetch_add(1, std::memory_order_relaxed);
    g_pinned_driver_ns.fetch_add(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(), std::memory_order_relaxed);
    std::lock_guard<std::mutex> lk(mu);
    live[p] = c;
    return p;
  }
  // for staging vectors (internal.hpp StageVec): plain memory where nothing can be pinned — the host halves of the scan run
  // without a GPU (dfgpu_parquet_inspect_chunk, dfgpu_ipc_open) — or with DFGPU_PINNED_STAGING=0
  void* alloc_or_plain(size_t n) {
    static const bool off = false;
    if (!off && current_device() >= 0) {
      try {
        return alloc(n);
      } catch (const Error&) {
        (void)hipGetLastError();
      }
    }
    // plain blocks are kept too: a fresh block of this size is page-faulted in 4 KB at a time (38 MB: 9 ms against 2.7 ms for
    // the copy into a block that was touched before)
    size_t c = 4096;
    while (c < n) c <<= 1;
    {
      std::lock_guard<std::mutex> lk(mu);
      auto it = free_plain.lower_bound(c);
      if (it != free_plain.end() && it->first <= c + c / 4) {
        void* p = it->second;
        cached_plain -= it->first;
        plain[p] = it->first;
        free_plain.erase(it);
        return p;
      }
    }
    void* p = std::malloc(c);
    if (!p) throw std::bad_alloc();
    std::lock_guard<std::mutex> lk(mu);
    plain[p] = c;
    return p;
  }
  std::multimap<size_t, void*> free_plain;
  std::map<void*, size_t> plain;
  size_t cached_plain = 0;
  void release(void* p) {
    if (!p) return;
    size_t c = 0;
    {
      std::lock_guard<std::mutex> lk(mu);
      auto pl = plain.find(p);
      if (pl != plain.end()) {
        const size_t pc = pl->second;
        plain.erase(pl);
        if (cached_plain + pc <= ((size_t)1 << 30)) {
          free_plain.emplace(pc, p);
          cached_plain += pc;
        } else {
          std::free(p);
        }
        return;
      }
      auto it = live.find(p);
      if (it == live.end()) return;
      c = it->second;
      live.erase(it);
      if (cached + c <= cap()) {
        free_blocks.emplace(c, p);
        cached += c;
        return;
      }
    }
    (void)hipHostFree(p);
  }
};
static PinnedPool& pinned() {
  static PinnedPool* p = new PinnedPool();  // never destroyed: release callbacks may run at interpreter exit
  return *p;
}

void* pinned_alloc(size_t n) { return pinned().alloc(n); }
void* stage_alloc(size_t n) { return pinned().alloc_or_plain(n); }
void pinned_release(void* p) { pinned().release(p); }

struct ExportPrivate {
  std::vector<void*> host_buffers;
  std::vector<void*> pinned_buffers;
  std::vector<const void*> buffer_ptrs;
  std::vector<ArrowArray*> children;
  ArrowArray* dictionary = nullptr;
};
static void release_array(ArrowArray* a) {
  if (!a || !a->release) return;
  auto* p = (ExportPrivate*)a->private_data;
  for (ArrowArray* c : p->children) {
    if (c->release) c->release(c);
    delete c;
  }
  if (p->dictionary) {
    if (p->dictionary->release) p->dictionary->release(p->dictionary);
    delete p->dictionary;
  }
  for (void* b : p->host_buffers) std::free(b);
  for (void* b : p->pinned_buffers) pinned().release(b);
  delete p;
  a->release = nullptr;
}
struct SchemaPrivate {
  std::string format, name;
  std::vector<ArrowSchema*> children;
  ArrowSchema* dictionary = nullptr;
};
static void release_schema(ArrowSchema* s) {
  if (!s || !s->release) return;
  auto* p = (SchemaPrivate*)s->private_data;
  for (ArrowSchema* c : p->children) {
    if (c->release) c->release(c);
    delete c;
  }
  if (p->dictionary) {
    if (p->dictionary->release) p->dictionary->release(p->dictionary);
    delete p->dictionary;
  }
  delete p;
  s->release = nullptr;
}
static void fill_schema(ArrowSchema* s, const std::string& fmt, const std::string& name, bool nullable) {
  auto* p = new SchemaPrivate{fmt, name, {}, nullptr};
  std::memset(s, 0, sizeof(*s));
  s->format = p->format.c_str();
  s->name = p->name.c_str();
  s->flags = nullable ? 2 : 0;
  s->release = release_schema;
  s->private_data = p;
}

// the dictionary values as a Utf8 / LargeUtf8 array the consumer owns (host copies of the strings)
static ArrowArray* export_dictionary(const DictValues& dv) {
  auto* a = new ArrowArray();
  std::memset(a, 0, sizeof(*a));
  auto* p = new ExportPrivate();
  const size_t n = dv.values.size();
  const bool large = dv.value_format == "U";
  size_t total = 0;
  for (const std::string& v : dv.values) total += v.size();
  void* offs = std::malloc((n + 1) * (large ? 8 : 4));
  char* data = (char*)std::malloc(total ? total : 8);
  uint8_t* vbits = nullptr;
  int64_t nulls = 0;
  for (size_t i = 0; i < n; i++) nulls += dv.valid[i] ? 0 : 1;
  if (nulls) vbits = (uint8_t*)std::calloc((n + 7) / 8 + 8, 1);
  size_t pos = 0;
  for (size_t i = 0; i < n; i++) {
    if (large) ((int64_t*)offs)[i] = (int64_t)pos; else ((int32_t*)offs)[i] = (int32_t)pos;
    std::memcpy(data + pos, dv.values[i].data(), dv.values[i].size());
    pos += dv.values[i].size();
    if (vbits && dv.valid[i]) vbits[i >> 3] |= (uint8_t)(1u << (i & 7));
  }
  if (large) ((int64_t*)offs)[n] = (int64_t)pos; else ((int32_t*)offs)[n] = (int32_t)pos;
  if (vbits) p->host_buffers.push_back(vbits);
  p->host_buffers.push_back(offs);
  p->host_buffers.push_back(data);
  p->buffer_ptrs = {vbits, offs, data};
  a->length = (int64_t)n;
  a->null_count = nulls;
  a->n_buffers = 3;
  a->buffers = p->buffer_ptrs.data();
  a->release = release_array;
  a->private_data = p;
  return a;
}

void bitmap_place(const uint64_t* src, int64_t off, int64_t n, uint64_t* dst) {
  if (n > 0) k_bitmap_place<<<grid_for((n + 127) / 64, BLOCK), BLOCK, 0, rt().stream>>>(src, off, n, (unsigned long long*)dst);
}

Column remap_to_dictionary(const Column& c, const std::shared_ptr<const DictValues>& target) {
  DFGPU_CHECK(c.dict && target, "remap_to_dictionary: both columns must be dictionary-encoded");
  if (same_dictionary(c.dict, target)) return c;
  std::map<std::string, int32_t> index;
  int32_t null_index = -1;
  for (size_t k = 0; k < target->values.size(); k++) {
    if (target->valid[k]) index.emplace(target->values[k], (int32_t)k);
    else null_index = (int32_t)k;
  }
  const int32_t missing = (int32_t)target->values.size();
  const int w = type_width(c.field.type);
  DFGPU_CHECK(w > 1 || missing <= 255, "the dictionary index type cannot hold the marker for values missing from the other dictionary");
  std::vector<int32_t> remap(c.dict->values.size() + 1, missing);
  for (size_t k = 0; k < c.dict->values.size(); k++) {
    if (!c.dict->valid[k]) remap[k] = null_index >= 0 ? null_index : missing;
    else {
      auto it = index.find(c.dict->values[k]);
      if (it != index.end()) remap[k] = it->second;
    }
  }
  Column n = alloc_column(c.field, c.name, c.length);
  n.validity = c.validity;
  n.null_count = c.null_count;
  n.dict = target;
  if (c.length == 0) return n;
  BufPtr d_remap = make_buf(remap.size() * 4 + 16);
  DFGPU_HIP(hipMemcpyAsync(d_remap->ptr, remap.data(), remap.size() * 4, hipMemcpyHostToDevice, rt().stream));
  DFGPU_HIP(hipStreamSynchronize(rt().stream));   // `remap` is a local
  const int g = grid_for(c.length, BLOCK);
  const int64_t nd = (int64_t)c.dict->values.size();
  switch (w) {
    case 1: k_remap_indices<uint8_t><<<g, BLOCK, 0, rt().stream>>>((const uint8_t*)c.ptr(), d_remap->as<int32_t>(), nd, c.length, (uint8_t*)n.data->ptr); break;
    case 4: k_remap_indices<uint32_t><<<g, BLOCK, 0, rt().stream>>>((const uint32_t*)c.ptr(), d_remap->as<int32_t>(), nd, c.length, (uint32_t*)n.data->ptr); break;
    default: k_remap_indices<uint64_t><<<g, BLOCK, 0, rt().stream>>>((const uint64_t*)c.ptr(), d_remap->as<int32_t>(), nd, c.length, (uint64_t*)n.data->ptr); break;
  }
  DFGPU_HIP(hipGetLastError());
  DFGPU_HIP(hipStreamSynchronize(rt().stream));   // d_remap is released on return
  return n;
}

}  // namespace dfgpu

using namespace dfgpu;

extern "C" {

int dfgpu_table_import(struct ArrowArray* array, struct ArrowSchema* schema, dfgpu_table_t* out) {
  int rc = guarded([&] {
    require_init();
    DFGPU_CHECK(array && schema && out, "null argument");
    DFGPU_CHECK(std::string(schema->format) == "+s", "dfgpu_table_import expects a struct array (RecordBatch)");
    DFGPU_CHECK(array->n_children == schema->n_children, "array/schema children mismatch");
    auto t = std::make_unique<Table>();
    t->nrows = array->length;
    Uploader up;
    for (int64_t i = 0; i < array->n_children; i++)
      t->cols.push_back(import_column(array->children[i], schema->children[i], array->offset, array->length, up));
    up.finish();
    *out = wrap(t.release());
  });
  // the call consumes both structures whether or not it succeeded
  if (array && array->release) array->release(array);
  if (schema && schema->release) schema->release(schema);
  return rc;
}

// rows [offset, offset + length) of `t` as a struct array in pinned host memory.  Bit-packed buffers (validity, Boolean values)
// start at a word boundary: the exported child carries the Arrow `offset` (< 64) of its first row.
static void export_rows(Table* t, int64_t offset, int64_t length, struct ArrowArray* out_array, struct ArrowSchema* out_schema) {
  Runtime& r = rt();
  DFGPU_CHECK(out_array && out_schema, "null argument");
  DFGPU_CHECK(offset >= 0 && length >= 0 && offset + length <= t->nrows, "export: row range out of bounds");
  auto* ap = new ExportPrivate();
  fill_schema(out_schema, "+s", "", false);
  auto* sp = (SchemaPrivate*)out_schema->private_data;
  const int64_t lead = offset & 63;          // rows ahead of `offset` inside its 64-row word
  const int64_t row0 = offset - lead;
  for (Column& c : t->cols) {
    if (c.validity && c.null_count < 0) count_nulls(c);
    auto* ca = new ArrowArray();
    std::memset(ca, 0, sizeof(*ca));
    auto* cp = new ExportPrivate();
    const bool bits = c.field.type == DFGPU_BOOL;
    const bool with_valid = c.validity && c.null_count != 0;
    const bool shifted = (bits || with_valid) && lead != 0;   // every buffer of the child starts `lead` rows early
    const int64_t first = shifted ? row0 : offset, rows = shifted ? length + lead : length;
    if (c.field.type == DFGPU_UTF8) {
      // strings: offsets first (their ends give the byte range), rebased to the slice; Utf8 while the bytes fit 32-bit
      // offsets, LargeUtf8 beyond
      PinnedBuf ho_buf((size_t)(rows + 1) * 8);   // (pinned: internal.hpp PinnedBuf)
      int64_t* const ho = ho_buf.as<int64_t>();
      ho[0] = 0;
      if (rows) d2h(ho, (const char*)c.offsets->ptr + (size_t)first * 8, (size_t)(rows + 1) * 8);
      const int64_t b0 = ho[0], nbytes = ho[(size_t)rows] - b0;
      const bool large = nbytes > 0x7FFFFFFFll;
      void* hoff = pinned().alloc((size_t)(rows + 1) * (large ? 8 : 4) + 8);
      cp->pinned_buffers.push_back(hoff);
      for (int64_t i = 0; i <= rows; i++) {
        if (large) ((int64_t*)hoff)[i] = ho[(size_t)i] - b0;
        else ((int32_t*)hoff)[i] = (int32_t)(ho[(size_t)i] - b0);
      }
      void* hb = pinned().alloc((size_t)nbytes + 8);
      cp->pinned_buffers.push_back(hb);
      if (nbytes) export_copy(hb, (const char*)c.ptr() + b0, (size_t)nbytes);
      void* hv = nullptr;
      int64_t nulls = 0;
      if (with_valid) {
        const size_t vb = bitmap_bytes(rows);
        hv = pinned().alloc(vb ? vb : 8);
        cp->pinned_buffers.push_back(hv);
        if (vb) export_copy(hv, (const char*)c.validity->ptr + (size_t)(first >> 6) * 8, vb);
        nulls = (offset == 0 && length == c.length) ? c.null_count : -1;
      }
      cp->buffer_ptrs = {hv, hoff, hb};
      ca->length = length;
      ca->offset = shifted ? lead : 0;
      ca->null_count = nulls;
      ca->n_buffers = 3;
      ca->buffers = cp->buffer_ptrs.data();
      ca->release = release_array;
      ca->private_data = cp;
      ap->children.push_back(ca);
      auto* cs = new ArrowSchema();
      fill_schema(cs, large ? "U" : "u", c.name, true);
      sp->children.push_back(cs);
      continue;
    }
    const size_t db = bits ? bitmap_bytes(rows) : (size_t)rows * type_width(c.field.type);
    void* hd = pinned().alloc(db ? db : 8);
    cp->pinned_buffers.push_back(hd);
    if (db) {
      const char* src = (const char*)c.ptr() + (bits ? (size_t)(first >> 6) * 8 : (size_t)first * type_width(c.field.type));
      export_copy(hd, src, db);
    }
    void* hv = nullptr;
    int64_t nulls = 0;
    if (with_valid) {
      const size_t vb = bitmap_bytes(rows);
      hv = pinned().alloc(vb ? vb : 8);
      cp->pinned_buffers.push_back(hv);
      if (vb) export_copy(hv, (const char*)c.validity->ptr + (size_t)(first >> 6) * 8, vb);
      nulls = (offset == 0 && length == c.length) ? c.null_count : -1;  // a slice's count is left to the consumer
    }
    cp->buffer_ptrs = {hv, hd};
    ca->length = length;
    ca->offset = shifted ? lead : 0;
    ca->null_count = nulls;
    ca->n_buffers = 2;
    ca->buffers = cp->buffer_ptrs.data();
    ca->release = release_array;
    ca->private_data = cp;
    ap->children.push_back(ca);
    auto* cs = new ArrowSchema();
    fill_schema(cs, c.dict ? c.dict->index_format : format_of(c.field), c.name, true);
    if (c.dict) {  // dictionary-encoded strings: indices from the device, values from the host
      cp->dictionary = export_dictionary(*c.dict);
      ca->dictionary = cp->dictionary;
      auto* ds = new ArrowSchema();
      fill_schema(ds, c.dict->value_format, "", true);
      ((SchemaPrivate*)cs->private_data)->dictionary = ds;
      cs->dictionary = ds;
    }
    sp->children.push_back(cs);
  }
  DFGPU_HIP(hipStreamSynchronize(r.stream));
  std::memset(out_array, 0, sizeof(*out_array));
  ap->buffer_ptrs = {nullptr};
  out_array->length = length;
  out_array->n_buffers = 1;
  out_array->buffers = ap->buffer_ptrs.data();
  out_array->n_children = (int64_t)ap->children.size();
  out_array->children = ap->children.data();
  out_array->release = release_array;
  out_array->private_data = ap;
  out_schema->n_children = (int64_t)sp->children.size();
  out_schema->children = sp->children.data();
}

int dfgpu_table_export(dfgpu_table_t th, struct ArrowArray* out_array, struct ArrowSchema* out_schema) {
  return guarded([&] {
    require_init();
    Table* t = unwrap(th);
    export_rows(t, 0, t->nrows, out_array, out_schema);
  });
}

int dfgpu_table_export_batch(dfgpu_table_t th, int64_t offset, int64_t length, struct ArrowArray* out_array, struct ArrowSchema* out_schema) {
  return guarded([&] {
    require_init();
    export_rows(unwrap(th), offset, length, out_array, out_schema);
  });
}

int dfgpu_host_register(void* ptr, size_t bytes) {
  return guarded([&] {
    require_init();
    DFGPU_CHECK(ptr != nullptr && bytes > 0, "dfgpu_host_register: empty range");
    DFGPU_HIP(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
  });
}
int dfgpu_host_unregister(void* ptr) {
  return guarded([&] {
    require_init();
    DFGPU_HIP(hipHostUnregister(ptr));
  });
}

int dfgpu_table_export_into(dfgpu_table_t th, int64_t offset, int64_t length, void* const* data_buffers, void* const* validity_buffers) {
  return guarded([&] {
    require_init();
    Table* t = unwrap(th);
    Runtime& r = rt();
    DFGPU_CHECK(data_buffers != nullptr, "null argument");
    DFGPU_CHECK(offset >= 0 && length >= 0 && offset + length <= t->nrows, "export: row range out of bounds");
    DFGPU_CHECK((offset & 63) == 0 || validity_buffers == nullptr, "dfgpu_table_export_into: bitmaps are copied as whole words, the row offset must be a multiple of 64");
    for (size_t i = 0; i < t->cols.size(); i++) {
      Column& c = t->cols[i];
      const bool bits = c.field.type == DFGPU_BOOL;
      DFGPU_CHECK(c.field.type != DFGPU_UTF8, "dfgpu_table_export_into: Utf8 columns have no fixed size per row (use dfgpu_table_export_batch)");
      DFGPU_CHECK(!bits || (offset & 63) == 0, "dfgpu_table_export_into: Boolean columns need a row offset that is a multiple of 64");
      const size_t db = bits ? bitmap_bytes(length) : (size_t)length * type_width(c.field.type);
      const char* src = (const char*)c.ptr() + (bits ? (size_t)(offset >> 6) * 8 : (size_t)offset * type_width(c.field.type));
      if (db && data_buffers[i]) export_copy(data_buffers[i], src, db);
      if (validity_buffers && validity_buffers[i]) {
        const size_t vb = bitmap_bytes(length);
        if (c.validity) export_copy(validity_buffers[i], (const char*)c.validity->ptr + (size_t)(offset >> 6) * 8, vb);
        else std::memset(validity_buffers[i], 0xFF, vb);
      }
    }
    DFGPU_HIP(hipStreamSynchronize(r.stream));
  });
}

int dfgpu_table_dictionary_lookup(dfgpu_table_t th, int column, const char* utf8, int64_t len, int64_t* out_code) {
  return guarded([&] {
    Table* t = unwrap_quiet(th);
    DFGPU_CHECK(column >= 0 && column < (int)t->cols.size() && utf8 && out_code, "bad argument");
    const Column& c = t->cols[column];
    DFGPU_CHECK(c.dict != nullptr, "column '" + c.name + "' is not dictionary-encoded");
    const std::string want(utf8, (size_t)len);
    *out_code = -1;
    for (size_t i = 0; i < c.dict->values.size(); i++)
      if (c.dict->valid[i] && c.dict->values[i] == want) {
        *out_code = (int64_t)i;
        break;
      }
  });
}

// SQL LIKE over one string (arrow-string like.rs: `%` = any run of characters incl. none, `_` = exactly one character — a
// Unicode scalar, not a byte —, `\\` makes the next pattern character literal; ILIKE folds ASCII case).  Iterative with
// backtracking to the last `%`.
static size_t utf8_len(unsigned char c) { return c < 0x80 ? 1 : (c >> 5) == 0x6 ? 2 : (c >> 4) == 0xE ? 3 : (c >> 3) == 0x1E ? 4 : 1; }
static bool like_match(const std::string& s, const std::string& pat, bool fold_case) {
  auto lower = [&](unsigned char c) { return fold_case && c >= 'A' && c <= 'Z' ? (unsigned char)(c + 32) : c; };
  size_t si = 0, pi = 0, star_p = std::string::npos, star_s = 0;
  while (si < s.size()) {
    bool step = false;
    if (pi < pat.size()) {
      const unsigned char pc = (unsigned char)pat[pi];
      if (pc == '%') {
        star_p = ++pi;
        star_s = si;
        continue;
      }
      if (pc == '_') {
        si += std::min(utf8_len((unsigned char)s[si]), s.size() - si);
        pi++;
        step = true;
      } else {
        const size_t lit = (pc == '\\' && pi + 1 < pat.size()) ? pi + 1 : pi;
        if (lower((unsigned char)pat[lit]) == lower((unsigned char)s[si])) {
          si++;
          pi = lit + 1;
          step = true;
        }
      }
    }
    if (step) continue;
    if (star_p == std::string::npos) return false;
    star_s += std::min(utf8_len((unsigned char)s[star_s]), s.size() - star_s);  // let the last % swallow one more character
    si = star_s;
    pi = star_p;
  }
  while (pi < pat.size() && pat[pi] == '%') pi++;
  return pi == pat.size();
}

}  // extern "C"
namespace dfgpu {
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_dict_match(const T* __restrict__ codes, const uint8_t* __restrict__ match, int64_t n_match, int64_t n, uint64_t* __restrict__ out) {
  const int64_t n_words = (n + 63) >> 6;
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * BLOCK) >> 6;
  for (int64_t w = wave; w < n_words; w += n_waves) {
    const int64_t i = (w << 6) + lane_id();
    bool hit = false;
    if (i < n) {
      const uint64_t k = (uint64_t)codes[i];
      hit = k < (uint64_t)n_match && match[k] != 0;
    }
    const uint64_t word = ballot64(hit);
    if (lane_id() == 0) out[w] = word;
  }
}
// `dictionary-encoded column LIKE pattern` when the matching indices do not form a few runs (a dictionary with one value per
// row, TPC-H's p_name LIKE '%green%'): the pattern is matched against the dictionary's values on the host, the rows look their
// index up in the resulting table.  NULL rows stay NULL (the column's validity).
Column dictionary_like_column(const Column& c, const std::string& pattern, bool case_insensitive) {
  DFGPU_CHECK(c.dict != nullptr, "LIKE: the column is not dictionary-encoded");
  dfgpu_field bf{};
  bf.type = DFGPU_BOOL;
  bf.nullable = 1;
  Column o = alloc_column(bf, "", c.length);
  o.validity = c.validity;
  o.null_count = c.null_count;
  if (c.length == 0) return o;
  std::vector<uint8_t> match(c.dict->values.size() + 1, 0);
  for (size_t i = 0; i < c.dict->values.size(); i++) match[i] = c.dict->valid[i] && like_match(c.dict->values[i], pattern, case_insensitive) ? 1 : 0;
  BufPtr d_match = make_buf(match.size() + 16);
  DFGPU_HIP(hipMemcpyAsync(d_match->ptr, match.data(), match.size(), hipMemcpyHostToDevice, rt().stream));
  DFGPU_HIP(hipStreamSynchronize(rt().stream));  // `match` is a local
  const int64_t nd = (int64_t)c.dict->values.size();
  const int g = grid_for((c.length + 63) / 64, BLOCK / WAVE);
  ProfileScope ps("dictionary_like", c.length * type_width(c.field.type));
  switch (type_width(c.field.type)) {
    case 1: k_dict_match<uint8_t><<<g, BLOCK, 0, rt().stream>>>((const uint8_t*)c.ptr(), d_match->as<uint8_t>(), nd, c.length, o.data->as<uint64_t>()); break;
    case 4: k_dict_match<uint32_t><<<g, BLOCK, 0, rt().stream>>>((const uint32_t*)c.ptr(), d_match->as<uint8_t>(), nd, c.length, o.data->as<uint64_t>()); break;
    default: k_dict_match<uint64_t><<<g, BLOCK, 0, rt().stream>>>((const uint64_t*)c.ptr(), d_match->as<uint8_t>(), nd, c.length, o.data->as<uint64_t>()); break;
  }
  DFGPU_HIP(hipGetLastError());
  DFGPU_HIP(hipStreamSynchronize(rt().stream));  // d_match is released on return
  return o;
}
}  // namespace dfgpu
extern "C" {

int dfgpu_table_dictionary_like(dfgpu_table_t th, int column, const char* pattern, int64_t len, int case_insensitive, int64_t* out_codes, int64_t capacity,
                                int64_t* out_n) {
  return guarded([&] {
    Table* t = unwrap_quiet(th);
    DFGPU_CHECK(column >= 0 && column < (int)t->cols.size() && pattern && out_n, "bad argument");
    const Column& c = t->cols[column];
    DFGPU_CHECK(c.dict != nullptr, "column '" + c.name + "' is not dictionary-encoded");
    const std::string pat(pattern, (size_t)len);
    int64_t n = 0;
    for (size_t i = 0; i < c.dict->values.size(); i++)
      if (c.dict->valid[i] && like_match(c.dict->values[i], pat, case_insensitive != 0)) {
        if (out_codes && n < capacity) out_codes[n] = (int64_t)i;
        n++;
      }
    *out_n = n;
  });
}

int dfgpu_table_alloc(int ncols, const dfgpu_field* fields, const char* const* names, int64_t nrows, dfgpu_table_t* out) {
  return guarded([&] {
    require_init();
    auto t = std::make_unique<Table>();
    t->nrows = nrows;
    for (int i = 0; i < ncols; i++) t->cols.push_back(alloc_column(fields[i], names && names[i] ? names[i] : "", nrows));
    *out = wrap(t.release());
  });
}

int dfgpu_table_free(dfgpu_table_t t) {
  return guarded([&] { delete reinterpret_cast<Table*>(t); });
}
int dfgpu_table_num_rows(dfgpu_table_t t, int64_t* out) {
  return guarded([&] { *out = unwrap_quiet(t)->nrows; });
}
int dfgpu_table_num_columns(dfgpu_table_t t, int* out) {
  return guarded([&] { *out = (int)unwrap_quiet(t)->cols.size(); });
}
int dfgpu_table_column(dfgpu_table_t th, int i, dfgpu_column_view* out) {
  return guarded([&] {
    Table* t = unwrap_quiet(th);
    DFGPU_CHECK(i >= 0 && i < (int)t->cols.size(), "column index out of range");
    Column& c = t->cols[i];
    if (c.validity && c.null_count < 0) count_nulls(c);
    out->field = c.field;
    out->length = c.length;
    out->null_count = c.null_count;
    out->data = c.ptr();
    out->validity = c.validity ? (const uint8_t*)c.validity->ptr : nullptr;
    out->name = c.name.c_str();
    out->offsets = c.offsets ? c.offsets->as<int64_t>() : nullptr;
  });
}
int dfgpu_table_select(dfgpu_table_t th, const int* cols, int ncols, dfgpu_table_t* out) {
  return guarded([&] {
    Table* t = unwrap_quiet(th);
    auto o = std::make_unique<Table>();
    o->nrows = t->nrows;
    for (int i = 0; i < ncols; i++) {
      DFGPU_CHECK(cols[i] >= 0 && cols[i] < (int)t->cols.size(), "column index out of range");
      o->cols.push_back(t->cols[cols[i]]);
    }
    *out = wrap_quiet(o.release());
  });
}
int dfgpu_table_hstack(dfgpu_table_t a, dfgpu_table_t b, dfgpu_table_t* out) {
  return guarded([&] {
    Table *ta = unwrap_quiet(a), *tb = unwrap_quiet(b);
    DFGPU_CHECK(ta->nrows == tb->nrows, "hstack: row counts differ");
    auto o = std::make_unique<Table>(*ta);
    for (auto& c : tb->cols) o->cols.push_back(c);
    *out = wrap_quiet(o.release());
  });
}

int dfgpu_table_slice(dfgpu_table_t th, int64_t offset, int64_t length, dfgpu_table_t* out) {
  return guarded([&] {
    require_init();
    Table* t = unwrap(th);
    DFGPU_CHECK(offset >= 0 && length >= 0 && offset + length <= t->nrows, "slice out of range");
    auto o = std::make_unique<Table>();
    o->nrows = length;
    const int64_t nw = (length + 63) / 64;
    for (auto& c : t->cols) {
      Column n;
      if (c.field.type == DFGPU_UTF8) {
        n = slice_strings(c, offset, length);
      } else {
        n = alloc_like(c, length);
        if (length && c.field.type == DFGPU_BOOL) {
          k_bitmap_extract<<<grid_for(nw, BLOCK), BLOCK, 0, rt().stream>>>((const uint64_t*)c.ptr(), offset, length, n.data->as<uint64_t>());
        } else if (length) {
          const int w = type_width(c.field.type);
          DFGPU_HIP(hipMemcpyAsync(n.data->ptr, (const char*)c.ptr() + (size_t)offset * w, (size_t)length * w, hipMemcpyDeviceToDevice, rt().stream));
        }
      }
      if (c.validity && length) {  // the slice's validity bits start at bit 0 of a fresh word-padded bitmap
        n.validity = make_buf(bitmap_bytes(length));
        k_bitmap_extract<<<grid_for(nw, BLOCK), BLOCK, 0, rt().stream>>>(c.valid_words(), offset, length, n.validity->as<uint64_t>());
        n.null_count = -1;
      }
      DFGPU_HIP(hipGetLastError());
      o->cols.push_back(std::move(n));
    }
    *out = wrap(o.release());
  });
}

int dfgpu_table_concat(const dfgpu_table_t* parts, int nparts, dfgpu_table_t* out) {
  return guarded([&] {
    require_init();
    DFGPU_CHECK(nparts >= 1, "concat of zero tables");
    Table* first = unwrap(parts[0]);
    int64_t total = 0;
    for (int p = 0; p < nparts; p++) {
      Table* t = unwrap(parts[p]);
      DFGPU_CHECK(t->cols.size() == first->cols.size(), "concat: schema mismatch");
      total += t->nrows;
    }
    auto o = std::make_unique<Table>();
    o->nrows = total;
    for (size_t ci = 0; ci < first->cols.size(); ci++) {
      const Column& fc = first->cols[ci];
      if (fc.field.type == DFGPU_UTF8) {
        std::vector<const Column*> cs;
        for (int p = 0; p < nparts; p++) cs.push_back(&unwrap(parts[p])->cols[ci]);
        o->cols.push_back(concat_strings(cs, total));
        continue;
      }
      Column n = alloc_like(fc, total);
      if (fc.field.type == DFGPU_BOOL) {
        // bit-packed values: every part's bits are placed at its bit offset, exactly as the validity bitmaps are
        DFGPU_HIP(hipMemsetAsync(n.data->ptr, 0, bitmap_bytes(total) ? bitmap_bytes(total) : 1, rt().stream));
        bool nulls = false;
        for (int p = 0; p < nparts; p++) nulls |= unwrap(parts[p])->cols[ci].validity != nullptr;
        if (nulls) {
          n.validity = make_zero_buf(bitmap_bytes(total));
          n.null_count = -1;
        }
        int64_t at = 0;
        for (int p = 0; p < nparts; p++) {
          Table* t = unwrap(parts[p]);
          const Column& c = t->cols[ci];
          DFGPU_CHECK(c.field.type == DFGPU_BOOL, "concat: column type mismatch");
          if (t->nrows) {
            const int g = grid_for((t->nrows + 127) / 64, BLOCK);
            k_bitmap_place<<<g, BLOCK, 0, rt().stream>>>((const uint64_t*)c.ptr(), at, t->nrows, (unsigned long long*)n.data->ptr);
            if (nulls) k_bitmap_place<<<g, BLOCK, 0, rt().stream>>>(c.valid_words(), at, t->nrows, (unsigned long long*)n.validity->ptr);
          }
          at += t->nrows;
        }
        DFGPU_HIP(hipGetLastError());
        o->cols.push_back(std::move(n));
        continue;
      }
      int w = type_width(fc.field.type);
      int64_t off = 0;
      bool any_nulls = false;
      for (int p = 0; p < nparts; p++) any_nulls |= unwrap(parts[p])->cols[ci].validity != nullptr;
      if (any_nulls) {
        n.validity = make_zero_buf(bitmap_bytes(total));
        n.null_count = -1;
      }
      // dictionary-encoded parts with different dictionaries (row groups of a Parquet file, partitions of an exchange):
      // one merged ascending dictionary, every part's indices rewritten into it
      std::shared_ptr<DictValues> merged;
      std::map<std::string, int32_t> merged_index;
      std::vector<BufPtr> keep;
      if (fc.dict) {
        bool same = true;
        for (int p = 0; p < nparts; p++) {
          const auto& d = unwrap(parts[p])->cols[ci].dict;
          same &= d == fc.dict || (d && d->values == fc.dict->values && d->valid == fc.dict->valid);
        }
        if (!same) {
          for (int p = 0; p < nparts; p++) {
            const auto& d = unwrap(parts[p])->cols[ci].dict;
            DFGPU_CHECK(d != nullptr, "concat: dictionary-encoded and plain columns mixed");
            for (size_t k = 0; k < d->values.size(); k++) merged_index.emplace(d->valid[k] ? d->values[k] : std::string("\0null", 5), 0);
          }
          merged = std::make_shared<DictValues>();
          merged->index_format = fc.dict->index_format;
          merged->value_format = fc.dict->value_format;
          int32_t next = 0;
          bool has_null_value = false;
          for (auto& kv : merged_index) {   // std::map iterates in ascending key order; the NULL marker sorts first
            kv.second = next++;
            const bool is_null = kv.first.size() == 5 && kv.first[0] == '\0';
            has_null_value |= is_null;
            merged->values.push_back(is_null ? std::string() : kv.first);
            merged->valid.push_back(is_null ? 0 : 1);
          }
          merged->sorted = !has_null_value;
          DFGPU_CHECK(w > 1 || next <= 256, "concat: the merged dictionary does not fit the UInt8 index type");
          n.dict = merged;
        }
      }
      for (int p = 0; p < nparts; p++) {
        Table* t = unwrap(parts[p]);
        const Column& c = t->cols[ci];
        DFGPU_CHECK(c.field.type == fc.field.type, "concat: column type mismatch");
        DFGPU_CHECK((c.dict != nullptr) == (fc.dict != nullptr), "concat: dictionary-encoded and plain columns mixed");
        if (merged && t->nrows) {
          // this part's indices -> indices of the merged dictionary
          std::vector<int32_t> remap(c.dict->values.size(), 0);
          for (size_t k = 0; k < remap.size(); k++) remap[k] = merged_index.at(c.dict->valid[k] ? c.dict->values[k] : std::string("\0null", 5));
          BufPtr d_remap = make_buf(remap.size() * 4 + 16);
          DFGPU_HIP(hipMemcpyAsync(d_remap->ptr, remap.data(), remap.size() * 4, hipMemcpyHostToDevice, rt().stream));
          DFGPU_HIP(hipStreamSynchronize(rt().stream));   // `remap` is a local
          char* dst = (char*)n.data->ptr + (size_t)off * w;
          const int g = grid_for(t->nrows, BLOCK);
          switch (w) {
            case 1: k_remap_indices<uint8_t><<<g, BLOCK, 0, rt().stream>>>((const uint8_t*)c.ptr(), d_remap->as<int32_t>(), (int64_t)remap.size(), t->nrows, (uint8_t*)dst); break;
            case 4: k_remap_indices<uint32_t><<<g, BLOCK, 0, rt().stream>>>((const uint32_t*)c.ptr(), d_remap->as<int32_t>(), (int64_t)remap.size(), t->nrows, (uint32_t*)dst); break;
            default: k_remap_indices<uint64_t><<<g, BLOCK, 0, rt().stream>>>((const uint64_t*)c.ptr(), d_remap->as<int32_t>(), (int64_t)remap.size(), t->nrows, (uint64_t*)dst); break;
          }
          keep.push_back(d_remap);
        }
        if (any_nulls && t->nrows)
          k_bitmap_place<<<grid_for((t->nrows + 127) / 64, BLOCK), BLOCK, 0, rt().stream>>>(c.valid_words(), off, t->nrows,
                                                                                            (unsigned long long*)n.validity->ptr);
        if (t->nrows && !merged)
          DFGPU_HIP(hipMemcpyAsync((char*)n.data->ptr + (size_t)off * w, c.ptr(), (size_t)t->nrows * w, hipMemcpyDeviceToDevice, rt().stream));
        off += t->nrows;
      }
      o->cols.push_back(std::move(n));
    }
    *out = wrap(o.release());
  });
}

}  // extern "C"
