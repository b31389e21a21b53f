// strings.hip — variable-length strings in HBM (DFGPU_UTF8: 64-bit offsets + bytes).
//
// What the reference does with string keys and predicates on the CPU, restated for HBM:
//   * interning (ArrowBytesMap::insert_if_new, physical-expr-common/src/binary_map.rs:215-420; GroupValuesByes,
//     aggregates/group_values/single_group_by/bytes.rs; ByteGroupValueBuilder equal_to / append, multi_group_by/bytes.rs):
//     hash the bytes (hash_utils.rs:401-640 hashes the same bytes), look the hash up, compare bytes with the stored
//     value, hand out group numbers in first-seen order.  Here: ONE pass in which every row claims or joins a slot of an
//     open-addressing table in HBM (the representative of a slot converges to the string's first row by atomicMin), then
//     representatives -> row bitmask -> popcount prefix -> dense numbers, exactly the aggregate's interning (aggregate.hip).
//     The result is a dictionary-encoded column; joins / GROUP BY / ORDER BY / repartition then run on 4-byte indices,
//     which is what makes string keys an HBM-friendly workload (a 25-byte key becomes a 4-byte one).
//   * comparisons and LIKE against a literal (arrow-ord cmp, arrow-string like.rs): one thread per string, the literal in
//     the kernel's argument block, a wave's 64 results leave as one ballot word.
//   * take / filter of a string column (arrow-select take_bytes / filter_bytes): lengths -> exclusive scan (= the new
//     offsets) -> byte copy.
// Strings are short (TPC-H: 10-40 bytes), so the unit of work is a thread per string moving 8 bytes at a time (unaligned
// 64-bit accesses through a packed struct: global memory takes them), not a wave per string.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string_view>
#include <thread>

#include <chrono>
#include <cstdio>

#include "device.hpp"
#include "internal.hpp"

namespace dfgpu {

void pack_bytes_to_bitmap(const uint8_t* bytes, int64_t n, uint64_t* words);

struct __attribute__((packed)) Unaligned64 {
  uint64_t v;
};
__device__ __forceinline__ uint64_t load_u64(const uint8_t* p) { return reinterpret_cast<const Unaligned64*>(p)->v; }
__device__ __forceinline__ void store_u64(uint8_t* p, uint64_t v) { reinterpret_cast<Unaligned64*>(p)->v = v; }

// bytes [0, len) of a string as a little-endian prefix: the last (partial) word zero-padded
__device__ __forceinline__ uint64_t tail_u64(const uint8_t* p, int64_t len) {
  uint64_t v = 0;
  for (int64_t k = 0; k < len; k++) v |= (uint64_t)p[k] << (8 * k);
  return v;
}
__device__ __forceinline__ uint64_t hash_bytes(const uint8_t* p, int64_t len) {
  uint64_t h = fmix64((uint64_t)len ^ SEED_AGG);
  int64_t k = 0;
  for (; k + 8 <= len; k += 8) h = fmix64(h ^ load_u64(p + k));
  if (k < len) h = fmix64(h ^ tail_u64(p + k, len - k) ^ 0x9E3779B97F4A7C15ULL);
  return h;
}
__device__ __forceinline__ bool bytes_equal(const uint8_t* a, const uint8_t* b, int64_t len) {
  int64_t k = 0;
  for (; k + 8 <= len; k += 8)
    if (load_u64(a + k) != load_u64(b + k)) return false;
  for (; k < len; k++)
    if (a[k] != b[k]) return false;
  return true;
}
// memcmp order = code point order of UTF-8 (Rust's str Ord, arrow-ord's byte-wise comparison)
__device__ __forceinline__ int bytes_compare(const uint8_t* a, int64_t la, const uint8_t* b, int64_t lb) {
  const int64_t m = la < lb ? la : lb;
  for (int64_t k = 0; k < m; k++)
    if (a[k] != b[k]) return a[k] < b[k] ? -1 : 1;
  return la < lb ? -1 : (la > lb ? 1 : 0);
}

// ------------------------------------------------------------------------------ take / filter
__global__ __launch_bounds__(BLOCK) void k_str_lengths(const int64_t* __restrict__ off, const uint64_t* __restrict__ valid, const int64_t* __restrict__ idx, int64_t n,
                                                       uint32_t* __restrict__ len, uint8_t* __restrict__ valid_bytes) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    const int64_t s = idx ? idx[i] : i;
    const bool ok = s >= 0 && (!valid || bit_at(valid, s));
    len[i] = ok ? (uint32_t)(off[s + 1] - off[s]) : 0u;
    if (valid_bytes) valid_bytes[i] = ok ? 1 : 0;
  }
}
__global__ __launch_bounds__(BLOCK) void k_str_copy(const int64_t* __restrict__ off, const uint8_t* __restrict__ bytes, const int64_t* __restrict__ idx, int64_t n,
                                                    const uint64_t* __restrict__ new_off, uint8_t* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    const int64_t len = (int64_t)(new_off[i + 1] - new_off[i]);
    if (len == 0) continue;
    const int64_t s = idx ? idx[i] : i;
    const uint8_t* src = bytes + off[s];
    uint8_t* dst = out + new_off[i];
    int64_t k = 0;
    for (; k + 8 <= len; k += 8) store_u64(dst + k, load_u64(src + k));
    for (; k < len; k++) dst[k] = src[k];
  }
}
// row ids of the set bits of a mask, in order
__global__ __launch_bounds__(BLOCK) void k_str_mask_ids(const uint64_t* __restrict__ mask, const uint64_t* __restrict__ mask_valid, const uint64_t* __restrict__ prefix,
                                                        int64_t n, int64_t* __restrict__ ids) {
  const int64_t n_words = (n + 63) >> 6;
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * BLOCK) >> 6;
  for (int64_t w = wave; w < n_words; w += n_waves) {
    uint64_t m = mask[w];
    if (mask_valid) m &= mask_valid[w];
    const int64_t rem = n - (w << 6);
    if (rem < 64) m &= (~0ull) >> (64 - rem);
    if ((m >> lane_id()) & 1ull) ids[prefix[w] + mbcnt(m)] = (w << 6) + lane_id();
  }
}

Column alloc_string_column(const Column& like, int64_t n) {
  Column c;
  c.field = like.field;
  c.field.type = DFGPU_UTF8;
  c.name = like.name;
  c.length = n;
  c.offsets = make_buf((size_t)(n + 1) * 8 + 16);
  return c;
}

Column gather_strings(const Column& in, const int64_t* idx, int64_t n, bool idx_may_be_null) {
  Runtime& r = rt();
  DFGPU_CHECK(in.field.type == DFGPU_UTF8 && (in.offsets || in.length == 0), "gather_strings: not a string column");
  Column out = alloc_string_column(in, n);
  if (n == 0) {
    DFGPU_HIP(hipMemsetAsync(out.offsets->ptr, 0, 8, r.stream));
    out.data = make_buf(16);
    return out;
  }
  const bool need_valid = idx_may_be_null || in.validity;
  BufPtr len = make_buf((size_t)n * 4 + 16);
  BufPtr vb = need_valid ? make_buf((size_t)n + 64) : nullptr;
  const int g = grid_for(n, BLOCK);
  {
    ProfileScope ps("take_string_lengths", n * 24);
    k_str_lengths<<<g, BLOCK, 0, r.stream>>>(str_offsets(in), in.valid_words(), idx, n, len->as<uint32_t>(), vb ? vb->as<uint8_t>() : nullptr);
  }
  scan_u32(len->as<uint32_t>(), n, out.offsets->as<uint64_t>());
  const int64_t total = (int64_t)read_u64(out.offsets->as<uint64_t>() + n);
  out.data = make_buf((size_t)total + 16);
  {
    ProfileScope ps("take_string_bytes", 2 * total + n * 24);
    k_str_copy<<<g, BLOCK, 0, r.stream>>>(str_offsets(in), (const uint8_t*)in.ptr(), idx, n, out.offsets->as<uint64_t>(), (uint8_t*)out.data->ptr);
    DFGPU_HIP(hipGetLastError());
  }
  if (need_valid) {
    out.validity = make_buf(bitmap_bytes(n));
    pack_bytes_to_bitmap(vb->as<uint8_t>(), n, out.validity->as<uint64_t>());
    out.null_count = -1;
    count_nulls(out);
  }
  return out;
}

Column compact_strings(const Column& in, const uint64_t* mask, const uint64_t* mask_valid, const uint64_t* prefix, int64_t nrows, int64_t n_out) {
  BufPtr ids = make_buf((size_t)std::max<int64_t>(n_out, 1) * 8);
  if (n_out) {
    const int64_t n_words = (nrows + 63) / 64;
    k_str_mask_ids<<<grid_for(n_words, BLOCK / WAVE), BLOCK, 0, rt().stream>>>(mask, mask_valid, prefix, nrows, ids->as<int64_t>());
    DFGPU_HIP(hipGetLastError());
  }
  return gather_strings(in, ids->as<int64_t>(), n_out, false);
}

__global__ __launch_bounds__(BLOCK) void k_str_rebase(const int64_t* __restrict__ off, int64_t n, int64_t base, int64_t* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) out[i] = off[i] + base;
}
Column concat_strings(const std::vector<const Column*>& parts, int64_t total_rows) {
  Runtime& r = rt();
  DFGPU_CHECK(!parts.empty(), "concat of zero columns");
  Column out = alloc_string_column(*parts[0], total_rows);
  // byte totals of the parts (their last offsets)
  std::vector<int64_t> bytes(parts.size(), 0);
  for (size_t p = 0; p < parts.size(); p++)
    if (parts[p]->length) bytes[p] = (int64_t)read_u64((const uint64_t*)str_offsets(*parts[p]) + parts[p]->length);
  const int64_t total_bytes = std::accumulate(bytes.begin(), bytes.end(), (int64_t)0);
  out.data = make_buf((size_t)total_bytes + 16);
  bool any_nulls = false;
  for (const Column* c : parts) any_nulls |= c->validity != nullptr;
  if (any_nulls) {
    out.validity = make_zero_buf(bitmap_bytes(total_rows));
    out.null_count = -1;
  }
  int64_t row = 0, base = 0;
  for (size_t p = 0; p < parts.size(); p++) {
    const Column& c = *parts[p];
    DFGPU_CHECK(c.field.type == DFGPU_UTF8, "concat: column type mismatch");
    if (c.length) {
      k_str_rebase<<<grid_for(c.length, BLOCK), BLOCK, 0, r.stream>>>(str_offsets(c), c.length, base, out.offsets->as<int64_t>() + row);
      if (bytes[p]) DFGPU_HIP(hipMemcpyAsync((char*)out.data->ptr + base, c.ptr(), (size_t)bytes[p], hipMemcpyDeviceToDevice, r.stream));
      if (any_nulls) bitmap_place(c.valid_words(), row, c.length, out.validity->as<uint64_t>());
    }
    row += c.length;
    base += bytes[p];
  }
  DFGPU_HIP(hipMemcpyAsync(out.offsets->as<int64_t>() + total_rows, &base, 8, hipMemcpyHostToDevice, r.stream));
  DFGPU_HIP(hipStreamSynchronize(r.stream));  // `base` is a local
  return out;
}

Column slice_strings(const Column& in, int64_t offset, int64_t length) {
  Runtime& r = rt();
  Column out = alloc_string_column(in, length);
  if (length == 0) {
    DFGPU_HIP(hipMemsetAsync(out.offsets->ptr, 0, 8, r.stream));
    out.data = make_buf(16);
    return out;
  }
  int64_t ends[2];
  d2h(&ends[0], str_offsets(in) + offset, 8);
  d2h(&ends[1], str_offsets(in) + offset + length, 8);
  k_str_rebase<<<grid_for(length + 1, BLOCK), BLOCK, 0, r.stream>>>(str_offsets(in) + offset, length + 1, -ends[0], out.offsets->as<int64_t>());
  out.data = make_buf((size_t)(ends[1] - ends[0]) + 16);
  if (ends[1] > ends[0]) DFGPU_HIP(hipMemcpyAsync(out.data->ptr, (const char*)in.ptr() + ends[0], (size_t)(ends[1] - ends[0]), hipMemcpyDeviceToDevice, r.stream));
  return out;
}

// ------------------------------------------------------------------------------ interning
// slots[s] = (first row holding the slot's string) + 1, 0 = empty.  A row either claims an empty slot or finds a slot whose
// representative has the same bytes; in that case it lowers the representative to itself when it comes earlier in the
// table (atomicMin): after the pass every slot holds the FIRST row of its string, whatever order the waves ran in.
__global__ __launch_bounds__(BLOCK) void k_str_intern(const int64_t* __restrict__ off, const uint8_t* __restrict__ bytes, const uint64_t* __restrict__ valid, int64_t n,
                                                      unsigned* __restrict__ slots, uint64_t mask, uint32_t* __restrict__ row_slot) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    if (valid && !bit_at(valid, i)) {
      row_slot[i] = 0xFFFFFFFFu;
      continue;
    }
    const uint8_t* p = bytes + off[i];
    const int64_t len = off[i + 1] - off[i];
    uint64_t s = hash_bytes(p, len) & mask;
    const unsigned me = (unsigned)i + 1u;
    for (;;) {
      unsigned cur = __hip_atomic_load(&slots[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (cur == 0u) {
        cur = atomicCAS(&slots[s], 0u, me);
        if (cur == 0u) break;  // claimed
      }
      const int64_t rep = (int64_t)cur - 1;
      if (off[rep + 1] - off[rep] == len && bytes_equal(bytes + off[rep], p, len)) {
        if (me < cur) atomicMin(&slots[s], me);
        break;
      }
      s = (s + 1) & mask;
    }
    row_slot[i] = (unsigned)s;
  }
}
// representatives -> bitmask over row numbers (scan.hip turns it into first-seen numbers)
__global__ __launch_bounds__(BLOCK) void k_str_mark_reps(const unsigned* __restrict__ slots, uint64_t capacity, unsigned long long* __restrict__ rep_mask) {
  for (uint64_t s = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; s < capacity; s += (uint64_t)gridDim.x * BLOCK) {
    const unsigned v = slots[s];
    if (v) atomicOr(&rep_mask[(v - 1) >> 6], 1ull << ((v - 1) & 63));
  }
}
__global__ __launch_bounds__(BLOCK) void k_str_codes(const unsigned* __restrict__ slots, const uint32_t* __restrict__ row_slot, const uint64_t* __restrict__ rep_mask,
                                                     const uint64_t* __restrict__ prefix, const int32_t* __restrict__ renumber, int64_t n, int32_t* __restrict__ codes) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    const uint32_t s = row_slot[i];
    int32_t c = 0;
    if (s != 0xFFFFFFFFu) {
      const unsigned rep = slots[s] - 1u;
      c = (int32_t)(prefix[rep >> 6] + __popcll(rep_mask[rep >> 6] & ((1ull << (rep & 63)) - 1ull)));
      if (renumber) c = renumber[c];
    }
    codes[i] = c;
  }
}


// the first 24 bytes of every string as three big-endian words (zero-padded): integer order of (w0, w1, w2) = byte order of the
// prefixes; what the distinct strings of a dictionary are sorted by on the device
__global__ __launch_bounds__(BLOCK) void k_str_prefix_words(const int64_t* __restrict__ off, const uint8_t* __restrict__ bytes, int64_t n, uint64_t* __restrict__ w0,
                                                            uint64_t* __restrict__ w1, uint64_t* __restrict__ w2, uint32_t* __restrict__ id) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    const uint8_t* p = bytes + off[i];
    const int64_t len = off[i + 1] - off[i];
    uint64_t w[3] = {0, 0, 0};
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
      for (int b = 0; b < 8; b++) {
        const int at = q * 8 + b;
        w[q] = (w[q] << 8) | (at < len ? (uint64_t)p[at] : 0ull);
      }
    w0[i] = w[0];
    w1[i] = w[1];
    w2[i] = w[2];
    id[i] = (uint32_t)i;
  }
}
__global__ __launch_bounds__(BLOCK) void k_count_equal_neighbours(const uint64_t* __restrict__ w0, const uint64_t* __restrict__ w1, const uint64_t* __restrict__ w2, int64_t n,
                                                                  unsigned* __restrict__ count) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x + 1; i < n; i += (int64_t)gridDim.x * BLOCK)
    if (w0[i] == w0[i - 1] && w1[i] == w1[i - 1] && w2[i] == w2[i - 1]) atomicAdd(count, 1u);
}
// row numbers of `strings` (no NULLs) in ascending order of their first 24 bytes, ties in first-to-last row order; `ties` = pairs of
// neighbours in that order whose first 24 bytes agree (their order is the caller's to settle)
static BufPtr string_prefix_order(const Column& strings, int64_t n, int64_t& ties) {
  Table keys;
  keys.nrows = n;
  dfgpu_field f{};
  f.type = DFGPU_UINT64;
  for (int q = 0; q < 3; q++) keys.cols.push_back(alloc_column(f, "w" + std::to_string(q), n));
  f.type = DFGPU_UINT32;
  keys.cols.push_back(alloc_column(f, "id", n));
  k_str_prefix_words<<<grid_for(n, BLOCK), BLOCK, 0, rt().stream>>>(str_offsets(strings), (const uint8_t*)strings.ptr(), n, keys.cols[0].data->as<uint64_t>(),
                                                                     keys.cols[1].data->as<uint64_t>(), keys.cols[2].data->as<uint64_t>(), keys.cols[3].data->as<uint32_t>());
  DFGPU_HIP(hipGetLastError());
  Table sorted = sort_table_ascending(keys, {0, 1, 2});
  BufPtr cnt = make_zero_buf(4);
  k_count_equal_neighbours<<<grid_for(n, BLOCK), BLOCK, 0, rt().stream>>>((const uint64_t*)sorted.cols[0].ptr(), (const uint64_t*)sorted.cols[1].ptr(),
                                                                           (const uint64_t*)sorted.cols[2].ptr(), n, cnt->as<unsigned>());
  unsigned c = 0;
  d2h(&c, cnt->ptr, 4);
  ties = c;
  const Column& ids = sorted.cols[3];
  if (ids.data_offset == 0) return ids.data;
  BufPtr out = make_buf((size_t)n * 4);
  DFGPU_HIP(hipMemcpyAsync(out->ptr, ids.ptr(), (size_t)n * 4, hipMemcpyDeviceToDevice, rt().stream));
  return out;
}

Column dictionary_encode(const Column& in, bool sorted) {
  Runtime& r = rt();
  DFGPU_CHECK(in.field.type == DFGPU_UTF8, "dictionary_encode: column '" + in.name + "' is not a Utf8 column");
  const int64_t n = in.length;
  DFGPU_CHECK(n < 0x7FFFFFFFll, "dictionary_encode: more than 2^31 rows");
  dfgpu_field f{};
  f.type = DFGPU_INT32;
  f.nullable = 1;
  Column out = alloc_column(f, in.name, n);
  out.validity = in.validity;
  out.null_count = in.null_count;
  auto dv = std::make_shared<DictValues>();
  dv->index_format = "i";
  dv->value_format = "u";
  if (n == 0) {
    dv->sorted = true;
    out.dict = dv;
    return out;
  }
  // DFGPU_TRACE_DICT=1: the phases of this call on stderr (where its host time goes: profiles/r3_strings.md)
  const bool trace = trace_on("dict");
  auto t_last = std::chrono::steady_clock::now();
  auto phase = [&](const char* what) {
    if (!trace) return;
    (void)hipStreamSynchronize(r.stream);
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[dict] %-28s %8.3f ms   (pool misses so far: %lld hipMalloc calls, %.3f ms)\n", what, std::chrono::duration<double, std::milli>(now - t_last).count(),
            (long long)r.driver_allocs.load(), (double)r.driver_alloc_ns.load() * 1e-6);
    t_last = now;
  };
  // 2 slots per row: every row its own string is the worst case.  (A cache-sized table tried first — 1 Mi slots, claims counted,
  // the full-size table on overflow — was measured and dropped: at 150 K distinct strings the intern kernel went 1.81 -> 2.39 ms,
  // every collision being a byte compare with somebody else's string; profiles/r3_strings.md.)
  uint64_t capacity = 1024;
  while (capacity < (uint64_t)n * 2) capacity <<= 1;
  BufPtr slots = make_zero_buf((size_t)capacity * 4);
  BufPtr row_slot = make_buf((size_t)n * 4 + 16);
  const int64_t row_words = (n + 63) / 64;
  BufPtr rep_mask = make_zero_buf((size_t)row_words * 8);
  BufPtr prefix = make_buf((size_t)(row_words + 1) * 8);
  phase("allocations");
  {
    ProfileScope ps("string_intern", n * 12);
    k_str_intern<<<grid_for(n, BLOCK), BLOCK, 0, r.stream>>>(str_offsets(in), (const uint8_t*)in.ptr(), in.valid_words(), n, slots->as<unsigned>(), capacity - 1,
                                                             row_slot->as<uint32_t>());
    DFGPU_HIP(hipGetLastError());
  }
  phase("intern kernel");
  k_str_mark_reps<<<grid_for((int64_t)capacity, BLOCK), BLOCK, 0, r.stream>>>(slots->as<unsigned>(), capacity, rep_mask->as<unsigned long long>());
  scan_mask_popcounts(rep_mask->as<uint64_t>(), nullptr, n, prefix->as<uint64_t>());
  const int64_t G = (int64_t)read_u64(prefix->as<uint64_t>() + row_words);
  // the distinct strings in first-seen order -> host
  BufPtr ids = make_buf((size_t)std::max<int64_t>(G, 1) * 8);
  if (G) k_str_mask_ids<<<grid_for(row_words, BLOCK / WAVE), BLOCK, 0, r.stream>>>(rep_mask->as<uint64_t>(), nullptr, prefix->as<uint64_t>(), n, ids->as<int64_t>());
  Column plain = in;
  plain.validity.reset();  // representatives are valid rows
  phase("mark reps + scan + count");
  Column values = gather_strings(plain, ids->as<int64_t>(), G, false);
  phase("gather distinct strings");
  PinnedBuf hoff_buf((size_t)(G + 1) * 8);   // pinned: see internal.hpp PinnedBuf
  int64_t* const hoff = hoff_buf.as<int64_t>();
  d2h(hoff, values.offsets->ptr, (size_t)(G + 1) * 8);
  PinnedBuf hbytes_buf((size_t)hoff[(size_t)G] + 1);
  char* const hbytes = hbytes_buf.as<char>();
  if (hoff[(size_t)G]) d2h(hbytes, values.data->ptr, (size_t)hoff[(size_t)G]);
  phase("download offsets + bytes");
  dv->values.resize((size_t)G);
  dv->valid.assign((size_t)G, 1);
  auto value_at = [&](int32_t k) { return std::string_view(hbytes + hoff[(size_t)k], (size_t)(hoff[(size_t)k + 1] - hoff[(size_t)k])); };
  BufPtr renumber;
  if (sorted && G > 1) {
    // ascending order of the distinct strings (byte order = code point order): views into the one downloaded buffer, sorted in
    // chunks on host threads and merged pairwise — the 150 K distinct names of 30 M rows took 25 ms of std::string compares on
    // one core, which was ten times the device side of the call
    // every string's first 24 bytes as three big-endian words (zero-padded): integer compares decide almost every pair; equal
    // words mean equal prefixes, then the shorter string is the smaller one unless both run past 24 bytes (full compare)
    struct SortKey {
      uint64_t w[3];
      uint32_t len;
      int32_t idx;
    };
    std::vector<SortKey> order((size_t)G);
    const int TP = (int)std::max<int64_t>(1, std::min<int64_t>({16, (int64_t)std::thread::hardware_concurrency(), G / 8192}));
    auto parallel_for = [&](auto&& body) {  // body(k) for k in [0, G), in TP contiguous chunks
      if (TP <= 1) {
        for (int64_t k = 0; k < G; k++) body(k);
        return;
      }
      std::vector<std::thread> th;
      for (int t = 0; t < TP; t++) th.emplace_back([&, t] { for (int64_t k = G * t / TP; k < G * (t + 1) / TP; k++) body(k); });
      for (auto& x : th) x.join();
    };
    auto prefix_words = [&](int32_t idx, SortKey& o) {
      const std::string_view v = value_at(idx);
      o.len = (uint32_t)v.size();
      o.idx = idx;
      for (int q = 0; q < 3; q++) {
        uint64_t x = 0;
        for (int b = 0; b < 8; b++) {
          const size_t at = (size_t)q * 8 + (size_t)b;
          x = (x << 8) | (at < v.size() ? (uint8_t)v[at] : 0u);
        }
        o.w[q] = x;
      }
    };
    auto less = [&](const SortKey& a, const SortKey& b) {
      if (a.w[0] != b.w[0]) return a.w[0] < b.w[0];
      if (a.w[1] != b.w[1]) return a.w[1] < b.w[1];
      if (a.w[2] != b.w[2]) return a.w[2] < b.w[2];
      if (a.len <= 24 || b.len <= 24) return a.len < b.len;
      return value_at(a.idx) < value_at(b.idx);
    };
    if (G >= 16384) {
      // many distinct strings: the device orders them by their first 24 bytes (three prefix words through the sort operator's
      // radix passes: 150 K strings in a few launches, where 16 host threads sorting and merging took 3.3-4 ms); the host only
      // settles runs of equal prefixes with the full comparison — none for keys and names, a few for long common prefixes
      int64_t ties = 0;
      BufPtr d_order = string_prefix_order(values, G, ties);
      PinnedBuf ids_buf((size_t)G * 4);
      d2h(ids_buf.ptr, d_order->ptr, (size_t)G * 4);
      const uint32_t* ids = ids_buf.as<uint32_t>();
      phase("device sort of the prefix words");
      if (ties == 0) {
        for (int64_t k = 0; k < G; k++) order[(size_t)k].idx = (int32_t)ids[k];
      } else {
        parallel_for([&](int64_t k) { prefix_words((int32_t)ids[k], order[(size_t)k]); });
        for (int64_t k = 0; k < G;) {
          int64_t e = k + 1;
          while (e < G && order[(size_t)e].w[0] == order[(size_t)k].w[0] && order[(size_t)e].w[1] == order[(size_t)k].w[1] && order[(size_t)e].w[2] == order[(size_t)k].w[2]) e++;
          if (e - k > 1) std::stable_sort(order.begin() + k, order.begin() + e, less);
          k = e;
        }
      }
    } else {
    parallel_for([&](int64_t k) { prefix_words((int32_t)k, order[(size_t)k]); });
    phase("sort keys (prefix words)");
    const int T = (int)std::max<int64_t>(1, std::min<int64_t>({16, (int64_t)std::thread::hardware_concurrency(), G / 4096}));
    if (T <= 1) {
      std::sort(order.begin(), order.end(), less);
    } else {
      std::vector<int64_t> cut((size_t)T + 1);
      for (int t = 0; t <= T; t++) cut[(size_t)t] = G * t / T;
      {
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++) th.emplace_back([&, t] { std::sort(order.begin() + cut[(size_t)t], order.begin() + cut[(size_t)t + 1], less); });
        for (auto& x : th) x.join();
      }
      for (int width = 1; width < T; width *= 2) {  // merge runs [t, t + width) and [t + width, t + 2 width)
        std::vector<std::thread> th;
        for (int t = 0; t + width < T; t += 2 * width)
          th.emplace_back([&, t, width] {
            std::inplace_merge(order.begin() + cut[(size_t)t], order.begin() + cut[(size_t)(t + width)], order.begin() + cut[(size_t)std::min(T, t + 2 * width)], less);
          });
        for (auto& x : th) x.join();
      }
    }
    }
    phase("host sort (threads)");
    PinnedBuf rank_buf((size_t)G * 4);
    int32_t* const rank = rank_buf.as<int32_t>();
    parallel_for([&](int64_t k) {
      rank[(size_t)order[(size_t)k].idx] = (int32_t)k;
      dv->values[(size_t)k] = std::string(value_at(order[(size_t)k].idx));
    });
    renumber = make_buf((size_t)G * 4);
    DFGPU_HIP(hipMemcpyAsync(renumber->ptr, rank, (size_t)G * 4, hipMemcpyHostToDevice, r.stream));
    DFGPU_HIP(hipStreamSynchronize(r.stream));  // `rank_buf` is a local
  } else {
    for (int64_t k = 0; k < G; k++) dv->values[(size_t)k] = std::string(value_at((int32_t)k));  // first-seen order
  }
  dv->sorted = sorted || G <= 1;
  phase("dictionary strings + ranks");
  {
    ProfileScope ps("string_codes", n * 16);
    k_str_codes<<<grid_for(n, BLOCK), BLOCK, 0, r.stream>>>(slots->as<unsigned>(), row_slot->as<uint32_t>(), rep_mask->as<uint64_t>(), prefix->as<uint64_t>(),
                                                            renumber ? renumber->as<int32_t>() : nullptr, n, out.data->as<int32_t>());
    DFGPU_HIP(hipGetLastError());
  }
  DFGPU_HIP(hipStreamSynchronize(r.stream));
  phase("codes kernel");
  out.dict = dv;
  return out;
}

// ------------------------------------------------------------------------------ substr
// Characters [first, last) of a string, 0-based, in UTF-8 code points (SQL SUBSTRING / functions/src/unicode/substr.rs): start
// is 1-based and may lie below 1, in which case the positions before the string eat into the count.
struct SubstrSpec {
  int64_t first;  // first character kept (0-based, >= 0)
  int64_t last;   // one past the last character kept; INT64_MAX = to the end
};
static SubstrSpec substr_spec(int64_t start, bool has_count, int64_t count) {
  DFGPU_CHECK(!has_count || count >= 0, "negative substring length not allowed: substr(<str>, " + std::to_string(start) + ", " + std::to_string(count) + ")");
  SubstrSpec sp;
  sp.first = start > 1 ? start - 1 : 0;
  sp.last = INT64_MAX;
  if (has_count) {
    const __int128 end = (__int128)start - 1 + count;  // exclusive, 0-based
    sp.last = end < 0 ? 0 : end > (__int128)INT64_MAX ? INT64_MAX : (int64_t)end;
  }
  if (sp.last < sp.first) sp.last = sp.first;
  return sp;
}
// byte range of the characters [first, last) of the `len` bytes at p
__host__ __device__ inline void substr_bytes(const uint8_t* p, int64_t len, SubstrSpec sp, int64_t& b0, int64_t& b1) {
  int64_t ch = 0, k = 0;
  b0 = len;
  b1 = len;
  bool have0 = false;
  for (; k < len; k++) {
    if ((p[k] & 0xC0) != 0x80) {  // a character starts here
      if (!have0 && ch == sp.first) { b0 = k; have0 = true; }
      if (ch == sp.last) { b1 = k; break; }
      ch++;
    }
  }
  if (!have0) b0 = b1 = len;  // the string is shorter than `first` characters
  if (b1 < b0) b1 = b0;
}
__global__ __launch_bounds__(BLOCK) void k_substr_lengths(const int64_t* __restrict__ off, const uint8_t* __restrict__ bytes, const uint64_t* __restrict__ valid, int64_t n,
                                                          SubstrSpec sp, uint32_t* __restrict__ len, uint32_t* __restrict__ begin) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    int64_t b0 = 0, b1 = 0;
    if (!valid || bit_at(valid, i)) substr_bytes(bytes + off[i], off[i + 1] - off[i], sp, b0, b1);
    len[i] = (uint32_t)(b1 - b0);
    begin[i] = (uint32_t)b0;
  }
}
__global__ __launch_bounds__(BLOCK) void k_substr_copy(const int64_t* __restrict__ off, const uint8_t* __restrict__ bytes, const uint32_t* __restrict__ begin, int64_t n,
                                                       const uint64_t* __restrict__ new_off, uint8_t* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    const int64_t len = (int64_t)(new_off[i + 1] - new_off[i]);
    const uint8_t* src = bytes + off[i] + begin[i];
    uint8_t* dst = out + new_off[i];
    for (int64_t k = 0; k < len; k++) dst[k] = src[k];
  }
}

Column substr_column(const Column& in, int64_t start, bool has_count, int64_t count) {
  Runtime& r = rt();
  const SubstrSpec sp = substr_spec(start, has_count, count);
  if (in.dict) {
    // dictionary-encoded: the function runs over the dictionary's values on the host; the indices are re-pointed at the
    // ascending dictionary of the distinct results
    auto mapped = std::make_shared<DictValues>(*in.dict);
    std::vector<std::string> distinct;
    bool any_null = false;
    for (size_t k = 0; k < mapped->values.size(); k++) {
      if (!mapped->valid[k]) {
        any_null = true;
        continue;
      }
      const std::string& v = in.dict->values[k];
      int64_t b0, b1;
      substr_bytes(reinterpret_cast<const uint8_t*>(v.data()), (int64_t)v.size(), sp, b0, b1);
      mapped->values[k] = v.substr((size_t)b0, (size_t)(b1 - b0));
      distinct.push_back(mapped->values[k]);
    }
    mapped->sorted = false;
    std::sort(distinct.begin(), distinct.end());
    distinct.erase(std::unique(distinct.begin(), distinct.end()), distinct.end());
    auto target = std::make_shared<DictValues>();
    target->index_format = in.dict->index_format;
    target->value_format = in.dict->value_format;
    target->values = std::move(distinct);
    target->valid.assign(target->values.size(), 1);
    if (any_null) {
      target->values.push_back(std::string());
      target->valid.push_back(0);
    }
    target->sorted = !any_null;
    Column tmp = in;
    tmp.dict = mapped;
    if (same_dictionary(mapped, target)) {
      tmp.dict = target;
      return tmp;
    }
    return remap_to_dictionary(tmp, target);
  }
  DFGPU_CHECK(in.field.type == DFGPU_UTF8 && (in.offsets || in.length == 0), "substr: the argument is neither a Utf8 column nor a dictionary-encoded string column");
  const int64_t n = in.length;
  Column out = alloc_string_column(in, n);
  out.validity = in.validity;
  out.null_count = in.null_count;
  if (n == 0) {
    DFGPU_HIP(hipMemsetAsync(out.offsets->ptr, 0, 8, r.stream));
    out.data = make_buf(16);
    return out;
  }
  BufPtr len = make_buf((size_t)n * 4 + 16), begin = make_buf((size_t)n * 4 + 16);
  const int g = grid_for(n, BLOCK);
  {
    ProfileScope ps("substr_lengths", n * 24);
    k_substr_lengths<<<g, BLOCK, 0, r.stream>>>(str_offsets(in), (const uint8_t*)in.ptr(), in.valid_words(), n, sp, len->as<uint32_t>(), begin->as<uint32_t>());
    DFGPU_HIP(hipGetLastError());
  }
  scan_u32(len->as<uint32_t>(), n, out.offsets->as<uint64_t>());
  const int64_t total = (int64_t)read_u64(out.offsets->as<uint64_t>() + n);
  out.data = make_buf((size_t)total + 16);
  {
    ProfileScope ps("substr_bytes", 2 * total + n * 20);
    k_substr_copy<<<g, BLOCK, 0, r.stream>>>(str_offsets(in), (const uint8_t*)in.ptr(), begin->as<uint32_t>(), n, out.offsets->as<uint64_t>(), (uint8_t*)out.data->ptr);
    DFGPU_HIP(hipGetLastError());
  }
  DFGPU_HIP(hipStreamSynchronize(r.stream));  // `len` / `begin` are released on return
  return out;
}

// ------------------------------------------------------------------------------ comparisons / LIKE
constexpr int STR_LIT_MAX = 256;
struct StrLit {
  union {
    uint8_t p[STR_LIT_MAX];
    uint64_t w[STR_LIT_MAX / 8];  // the same bytes as words (a uniform index into a kernel argument: scalar loads)
  };
  int len;
};
// len bytes at q equal the literal's first len bytes
__device__ __forceinline__ bool lit_equal(const uint8_t* q, const StrLit& lit) {
  int k = 0;
  for (; k + 8 <= lit.len; k += 8)
    if (load_u64(q + k) != lit.w[k >> 3]) return false;
  for (; k < lit.len; k++)
    if (q[k] != lit.p[k]) return false;
  return true;
}
__device__ __forceinline__ bool cmp_result(int op, int c) {
  switch (op) {
    case DFGPU_EXPR_EQ: return c == 0;
    case DFGPU_EXPR_NE: return c != 0;
    case DFGPU_EXPR_LT: return c < 0;
    case DFGPU_EXPR_LE: return c <= 0;
    case DFGPU_EXPR_GT: return c > 0;
    default: return c >= 0;
  }
}
__global__ __launch_bounds__(BLOCK) void k_str_cmp_lit(int op, const int64_t* __restrict__ off, const uint8_t* __restrict__ bytes, int64_t n, StrLit lit,
                                                       uint64_t* __restrict__ out) {
  const int64_t n_words = (n + 63) >> 6;
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * BLOCK) >> 6;
  for (int64_t w = wave; w < n_words; w += n_waves) {
    const int64_t i = (w << 6) + lane_id();
    bool rr = false;
    if (i < n) {
      const int64_t len = off[i + 1] - off[i];
      if (op == DFGPU_EXPR_EQ || op == DFGPU_EXPR_NE) {
        const bool eq = len == lit.len && lit_equal(bytes + off[i], lit);
        rr = (op == DFGPU_EXPR_EQ) == eq;
      } else {
        rr = cmp_result(op, bytes_compare(bytes + off[i], len, lit.p, lit.len));
      }
    }
    const uint64_t word = ballot64(rr);
    if (lane_id() == 0) out[w] = word;
  }
}
__global__ __launch_bounds__(BLOCK) void k_str_cmp_cols(int op, const int64_t* __restrict__ aoff, const uint8_t* __restrict__ ab, const int64_t* __restrict__ boff,
                                                        const uint8_t* __restrict__ bb, int64_t n, uint64_t* __restrict__ out) {
  const int64_t n_words = (n + 63) >> 6;
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * BLOCK) >> 6;
  for (int64_t w = wave; w < n_words; w += n_waves) {
    const int64_t i = (w << 6) + lane_id();
    bool rr = false;
    if (i < n) rr = cmp_result(op, bytes_compare(ab + aoff[i], aoff[i + 1] - aoff[i], bb + boff[i], boff[i + 1] - boff[i]));
    const uint64_t word = ballot64(rr);
    if (lane_id() == 0) out[w] = word;
  }
}
__device__ __forceinline__ int utf8_char_len(uint8_t c) { return c < 0x80 ? 1 : (c >> 5) == 0x6 ? 2 : (c >> 4) == 0xE ? 3 : (c >> 3) == 0x1E ? 4 : 1; }
__device__ __forceinline__ uint8_t fold(uint8_t c, bool ci) { return ci && c >= 'A' && c <= 'Z' ? (uint8_t)(c + 32) : c; }
// the matcher of table.hip like_match (arrow-string like.rs semantics), one thread per string
__device__ __forceinline__ bool like_device(const uint8_t* s, int64_t slen, const StrLit& pat, bool ci) {
  int64_t si = 0, star_s = 0;
  int pi = 0, star_p = -1;
  while (si < slen) {
    bool step = false;
    if (pi < pat.len) {
      const uint8_t pc = pat.p[pi];
      if (pc == '%') {
        star_p = ++pi;
        star_s = si;
        continue;
      }
      if (pc == '_') {
        const int64_t cl = utf8_char_len(s[si]);
        si += cl < slen - si ? cl : slen - si;
        pi++;
        step = true;
      } else {
        const int lit = (pc == '\\' && pi + 1 < pat.len) ? pi + 1 : pi;
        if (fold(pat.p[lit], ci) == fold(s[si], ci)) {
          si++;
          pi = lit + 1;
          step = true;
        }
      }
    }
    if (step) continue;
    if (star_p < 0) return false;
    const int64_t cl = utf8_char_len(s[star_s]);
    star_s += cl < slen - star_s ? cl : slen - star_s;
    si = star_s;
    pi = star_p;
  }
  while (pi < pat.len && pat.p[pi] == '%') pi++;
  return pi == pat.len;
}
__global__ __launch_bounds__(BLOCK) void k_str_like(const int64_t* __restrict__ off, const uint8_t* __restrict__ bytes, int64_t n, StrLit pat, int ci,
                                                    uint64_t* __restrict__ out) {
  const int64_t n_words = (n + 63) >> 6;
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * BLOCK) >> 6;
  for (int64_t w = wave; w < n_words; w += n_waves) {
    const int64_t i = (w << 6) + lane_id();
    const bool rr = i < n && like_device(bytes + off[i], off[i + 1] - off[i], pat, ci != 0);
    const uint64_t word = ballot64(rr);
    if (lane_id() == 0) out[w] = word;
  }
}

// LIKE 'lit%' / '%lit' / '%lit%' without `_` or escapes (TPC-H's 'PROMO%', '%BRASS', '%green%'): a prefix / suffix compare, or a
// search for the literal, instead of the general matcher's backtracking loop (30 M 18-byte names, 'Customer#00001%': 1.13 -> 0.35 ms)
enum { AFFIX_PREFIX = 0, AFFIX_SUFFIX = 1, AFFIX_CONTAINS = 2 };
__global__ __launch_bounds__(BLOCK) void k_str_affix(int mode, const int64_t* __restrict__ off, const uint8_t* __restrict__ bytes, int64_t n, StrLit lit,
                                                     uint64_t* __restrict__ out) {
  const int64_t n_words = (n + 63) >> 6;
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * BLOCK) >> 6;
  for (int64_t w = wave; w < n_words; w += n_waves) {
    const int64_t i = (w << 6) + lane_id();
    bool rr = false;
    if (i < n) {
      const int64_t len = off[i + 1] - off[i];
      const uint8_t* p = bytes + off[i];
      if (len >= lit.len) {
        if (mode == AFFIX_PREFIX) {
          rr = lit_equal(p, lit);
        } else if (mode == AFFIX_SUFFIX) {
          rr = lit_equal(p + (len - lit.len), lit);
        } else {
          for (int64_t at = 0; at + lit.len <= len && !rr; at++) rr = p[at] == lit.p[0] && lit_equal(p + at, lit);
        }
      }
    }
    const uint64_t word = ballot64(rr);
    if (lane_id() == 0) out[w] = word;
  }
}
// the pattern as (mode, literal) when it is one of those three shapes with a non-empty literal; -1 otherwise
static int affix_pattern(const std::string& pat, std::string& lit) {
  if (pat.find('_') != std::string::npos || pat.find('\\') != std::string::npos) return -1;
  const size_t first = pat.find('%');
  if (first == std::string::npos) return -1;   // no wildcard at all: the general matcher (an equality)
  const bool lead = pat.front() == '%', trail = pat.back() == '%';
  const size_t b = lead ? 1 : 0, e = pat.size() - (trail && pat.size() > b ? 1 : 0);
  if (e <= b) return -1;
  lit = pat.substr(b, e - b);
  if (lit.find('%') != std::string::npos) return -1;
  if (lead && trail) return AFFIX_CONTAINS;
  return lead ? AFFIX_SUFFIX : AFFIX_PREFIX;
}

static StrLit make_lit(const std::string& s, const char* what) {
  DFGPU_CHECK(s.size() <= (size_t)STR_LIT_MAX, std::string(what) + " longer than 256 bytes is not supported on the GPU path");
  StrLit l{};
  std::memcpy(l.p, s.data(), s.size());
  l.len = (int)s.size();
  return l;
}
static int mirrored(int op) {
  switch (op) {
    case DFGPU_EXPR_LT: return DFGPU_EXPR_GT;
    case DFGPU_EXPR_LE: return DFGPU_EXPR_GE;
    case DFGPU_EXPR_GT: return DFGPU_EXPR_LT;
    case DFGPU_EXPR_GE: return DFGPU_EXPR_LE;
    default: return op;
  }
}

Datum string_binary(int op, const Datum& a, const Datum& b, int64_t nrows) {
  Runtime& r = rt();
  dfgpu_field bf{};
  bf.type = DFGPU_BOOL;
  bf.nullable = 1;
  const bool like = op == DFGPU_EXPR_LIKE || op == DFGPU_EXPR_ILIKE;
  DFGPU_CHECK(like || (op >= DFGPU_EXPR_EQ && op <= DFGPU_EXPR_GE), "operator not supported on string operands");
  DFGPU_CHECK(a.col.field.type == DFGPU_UTF8 && b.col.field.type == DFGPU_UTF8,
              "string comparison: both operands must be Utf8 (a dictionary-encoded column is compared through its indices: bind the literal with dfgpu_table_dictionary_lookup)");
  if (a.scalar && !b.scalar) {
    DFGPU_CHECK(!like, "LIKE takes the column on the left and the pattern on the right");
    return string_binary(mirrored(op), b, a, nrows);
  }
  Datum o;
  if (a.scalar && b.scalar) {  // two literals: fold on the host
    o.scalar = true;
    o.col.field = bf;
    o.scalar_null = a.scalar_null || b.scalar_null;
    if (!o.scalar_null) {
      DFGPU_CHECK(!like, "LIKE between two literals is folded by the planner");
      const int c = a.str.compare(b.str);
      const bool rr = op == DFGPU_EXPR_EQ ? c == 0 : op == DFGPU_EXPR_NE ? c != 0 : op == DFGPU_EXPR_LT ? c < 0 : op == DFGPU_EXPR_LE ? c <= 0 : op == DFGPU_EXPR_GT ? c > 0 : c >= 0;
      o.lit_lo = rr ? 1 : 0;
    }
    return o;
  }
  o.col = alloc_column(bf, "", nrows);
  const int64_t nw = (nrows + 63) / 64;
  if (b.scalar) {
    if (b.scalar_null) {  // comparison with NULL: NULL everywhere
      o.col.validity = make_zero_buf(bitmap_bytes(nrows));
      o.col.null_count = nrows;
      if (nw) DFGPU_HIP(hipMemsetAsync(o.col.data->ptr, 0, (size_t)nw * 8, r.stream));
      return o;
    }
    o.col.validity = a.col.validity;
    o.col.null_count = a.col.null_count;
    if (nrows == 0) return o;
    const StrLit lit = make_lit(b.str, like ? "a LIKE pattern" : "a string literal");
    const int g = grid_for(nw, BLOCK / WAVE);
    ProfileScope ps(like ? "string_like" : "string_cmp", nrows * 9);
    std::string affix;
    const int affix_mode = (like && op == DFGPU_EXPR_LIKE) ? affix_pattern(b.str, affix) : -1;
    if (affix_mode >= 0)
      k_str_affix<<<g, BLOCK, 0, r.stream>>>(affix_mode, str_offsets(a.col), (const uint8_t*)a.col.ptr(), nrows, make_lit(affix, "a LIKE pattern"), o.col.data->as<uint64_t>());
    else if (like) k_str_like<<<g, BLOCK, 0, r.stream>>>(str_offsets(a.col), (const uint8_t*)a.col.ptr(), nrows, lit, op == DFGPU_EXPR_ILIKE, o.col.data->as<uint64_t>());
    else k_str_cmp_lit<<<g, BLOCK, 0, r.stream>>>(op, str_offsets(a.col), (const uint8_t*)a.col.ptr(), nrows, lit, o.col.data->as<uint64_t>());
    DFGPU_HIP(hipGetLastError());
    return o;
  }
  DFGPU_CHECK(!like, "LIKE with a column as the pattern is not supported on the GPU path");
  if (a.col.validity && b.col.validity) {
    o.col.validity = make_buf(bitmap_bytes(nrows));
    and_bitmaps(a.col.valid_words(), b.col.valid_words(), nw, o.col.validity->as<uint64_t>());
    o.col.null_count = -1;
  } else {
    o.col.validity = a.col.validity ? a.col.validity : b.col.validity;
    o.col.null_count = o.col.validity ? -1 : 0;
  }
  if (nrows == 0) return o;
  ProfileScope ps("string_cmp", nrows * 18);
  k_str_cmp_cols<<<grid_for(nw, BLOCK / WAVE), BLOCK, 0, r.stream>>>(op, str_offsets(a.col), (const uint8_t*)a.col.ptr(), str_offsets(b.col), (const uint8_t*)b.col.ptr(),
                                                                     nrows, o.col.data->as<uint64_t>());
  DFGPU_HIP(hipGetLastError());
  return o;
}

}  // namespace dfgpu

using namespace dfgpu;

namespace dfgpu {
// indices (any width) -> 64-bit row ids into the dictionary; a NULL row, a NULL dictionary value or an index beyond it -> -1
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_dict_ids(const T* __restrict__ codes, const uint64_t* __restrict__ valid, const uint8_t* __restrict__ value_valid, int64_t n_values, int64_t n,
                                                    int64_t* __restrict__ ids) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    const uint64_t k = (uint64_t)codes[i];
    const bool ok = (!valid || bit_at(valid, i)) && k < (uint64_t)n_values && value_valid[k] != 0;
    ids[i] = ok ? (int64_t)k : -1;
  }
}
// ------------------------------------------------------------------------------ content hash (routing on a string key)
__global__ __launch_bounds__(BLOCK) void k_str_hash(const int64_t* __restrict__ off, const uint8_t* __restrict__ bytes, int64_t n, uint64_t* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) out[i] = hash_bytes(bytes + off[i], off[i + 1] - off[i]);
}
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_dict_hash(const T* __restrict__ codes, const uint64_t* __restrict__ value_hash, int64_t n_values, int64_t n, uint64_t* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    const uint64_t k = (uint64_t)codes[i];
    out[i] = k < (uint64_t)n_values ? value_hash[k] : 0;   // (NULL rows may hold any index: their validity bit says so)
  }
}
static Column dictionary_values_column(const DictValues& dv, BufPtr* value_valid);
// A UInt64 column holding a hash of every row's string BYTES (NULL rows stay NULL): equal strings give equal values whatever
// table, dictionary or rank they come from.  Hash repartitioning routes string keys on it — the two sides of a Partitioned join
// arrive in separate calls with dictionaries of their own, and equal strings must still meet on one rank.
Column string_hash_column(const Column& in) {
  Runtime& r = rt();
  dfgpu_field f{};
  f.type = DFGPU_UINT64;
  f.nullable = in.field.nullable;
  Column out = alloc_column(f, in.name, in.length);
  out.validity = in.validity;
  out.null_count = in.null_count;
  const int64_t n = in.length;
  if (in.dict) {
    Column values = dictionary_values_column(*in.dict, nullptr);
    const int64_t nv = values.length;
    BufPtr vh = make_buf((size_t)std::max<int64_t>(nv, 1) * 8);
    if (nv) k_str_hash<<<grid_for(nv, BLOCK), BLOCK, 0, r.stream>>>(str_offsets(values), (const uint8_t*)values.ptr(), nv, vh->as<uint64_t>());
    if (n) {
      const int g = grid_for(n, BLOCK);
      switch (type_width(in.field.type)) {
        case 1: k_dict_hash<uint8_t><<<g, BLOCK, 0, r.stream>>>((const uint8_t*)in.ptr(), vh->as<uint64_t>(), nv, n, out.data->as<uint64_t>()); break;
        case 4: k_dict_hash<uint32_t><<<g, BLOCK, 0, r.stream>>>((const uint32_t*)in.ptr(), vh->as<uint64_t>(), nv, n, out.data->as<uint64_t>()); break;
        default: k_dict_hash<uint64_t><<<g, BLOCK, 0, r.stream>>>((const uint64_t*)in.ptr(), vh->as<uint64_t>(), nv, n, out.data->as<uint64_t>()); break;
      }
    }
    DFGPU_HIP(hipGetLastError());
    DFGPU_HIP(hipStreamSynchronize(r.stream));  // `values` / `vh` are released on return
    return out;
  }
  DFGPU_CHECK(in.field.type == DFGPU_UTF8, "string_hash_column: not a string column");
  if (n) k_str_hash<<<grid_for(n, BLOCK), BLOCK, 0, r.stream>>>(str_offsets(in), (const uint8_t*)in.ptr(), n, out.data->as<uint64_t>());
  DFGPU_HIP(hipGetLastError());
  return out;
}

// the dictionary's values as a Utf8 column in HBM (value_valid: one byte per value, optional)
static Column dictionary_values_column(const DictValues& dv, BufPtr* value_valid) {
  Runtime& r = rt();
  const int64_t nv = (int64_t)dv.values.size();
  std::vector<int64_t> off((size_t)nv + 1, 0);
  for (int64_t k = 0; k < nv; k++) off[(size_t)k + 1] = off[(size_t)k] + (int64_t)dv.values[(size_t)k].size();
  std::vector<char> bytes((size_t)off[(size_t)nv] + 1);
  for (int64_t k = 0; k < nv; k++) std::memcpy(bytes.data() + off[(size_t)k], dv.values[(size_t)k].data(), dv.values[(size_t)k].size());
  Column values;
  values.field.type = DFGPU_UTF8;
  values.field.nullable = 1;
  values.length = nv;
  values.offsets = make_buf((size_t)(nv + 1) * 8 + 16);
  values.data = make_buf((size_t)off[(size_t)nv] + 16);
  h2d_async(values.offsets->ptr, off.data(), (size_t)(nv + 1) * 8);
  if (off[(size_t)nv]) h2d_async(values.data->ptr, bytes.data(), (size_t)off[(size_t)nv]);
  if (value_valid) {
    *value_valid = make_buf((size_t)nv + 16);
    if (nv) h2d_async((*value_valid)->ptr, dv.valid.data(), (size_t)nv);
  }
  DFGPU_HIP(hipStreamSynchronize(r.stream));  // the host vectors are the copies' sources
  return values;
}

// dictionary-encoded -> Utf8: the dictionary's values go to HBM as a string column once, every row takes its value (the same
// take that filters and joins move strings with)
Column dictionary_decode(const Column& in) {
  Runtime& r = rt();
  DFGPU_CHECK(in.dict != nullptr, "dictionary_decode: column '" + in.name + "' is not dictionary-encoded");
  const DictValues& dv = *in.dict;
  const int64_t nv = (int64_t)dv.values.size();
  std::vector<int64_t> off((size_t)nv + 1, 0);
  for (int64_t k = 0; k < nv; k++) off[(size_t)k + 1] = off[(size_t)k] + (int64_t)dv.values[(size_t)k].size();
  std::vector<char> bytes((size_t)off[(size_t)nv] + 1);
  for (int64_t k = 0; k < nv; k++) std::memcpy(bytes.data() + off[(size_t)k], dv.values[(size_t)k].data(), dv.values[(size_t)k].size());
  Column values;
  values.field.type = DFGPU_UTF8;
  values.field.nullable = 1;
  values.length = nv;
  values.offsets = make_buf((size_t)(nv + 1) * 8 + 16);
  values.data = make_buf((size_t)off[(size_t)nv] + 16);
  BufPtr vvalid = make_buf((size_t)nv + 16);
  h2d_async(values.offsets->ptr, off.data(), (size_t)(nv + 1) * 8);
  if (off[(size_t)nv]) h2d_async(values.data->ptr, bytes.data(), (size_t)off[(size_t)nv]);
  if (nv) h2d_async(vvalid->ptr, dv.valid.data(), (size_t)nv);
  const int64_t n = in.length;
  BufPtr ids = make_buf((size_t)std::max<int64_t>(n, 1) * 8);
  if (n) {
    const int g = grid_for(n, BLOCK);
    switch (type_width(in.field.type)) {
      case 1: k_dict_ids<uint8_t><<<g, BLOCK, 0, r.stream>>>((const uint8_t*)in.ptr(), in.valid_words(), vvalid->as<uint8_t>(), nv, n, ids->as<int64_t>()); break;
      case 4: k_dict_ids<uint32_t><<<g, BLOCK, 0, r.stream>>>((const uint32_t*)in.ptr(), in.valid_words(), vvalid->as<uint8_t>(), nv, n, ids->as<int64_t>()); break;
      default: k_dict_ids<uint64_t><<<g, BLOCK, 0, r.stream>>>((const uint64_t*)in.ptr(), in.valid_words(), vvalid->as<uint8_t>(), nv, n, ids->as<int64_t>()); break;
    }
    DFGPU_HIP(hipGetLastError());
  }
  Column out = gather_strings(values, ids->as<int64_t>(), n, true);
  out.name = in.name;
  out.field.nullable = in.field.nullable;
  DFGPU_HIP(hipStreamSynchronize(r.stream));  // the host vectors are the copies' sources
  return out;
}
}  // namespace dfgpu

extern "C" int dfgpu_table_dictionary_size(dfgpu_table_t th, int column, int64_t* out_n) {
  return guarded([&] {
    Table* t = unwrap_quiet(th);
    DFGPU_CHECK(out_n && column >= 0 && column < (int)t->cols.size(), "bad argument");
    const Column& c = t->cols[(size_t)column];
    *out_n = c.dict ? (int64_t)c.dict->values.size() : -1;
  });
}

extern "C" int dfgpu_table_dictionary_decode(dfgpu_table_t th, int column, dfgpu_table_t* out) {
  return guarded([&] {
    require_init();
    Table* t = unwrap(th);
    DFGPU_CHECK(out && column >= 0 && column < (int)t->cols.size(), "bad argument");
    auto o = std::make_unique<Table>(*t);
    o->cols[(size_t)column] = dictionary_decode(t->cols[(size_t)column]);
    *out = wrap(o.release());
  });
}

extern "C" int dfgpu_table_dictionary_encode(dfgpu_table_t th, int column, int sorted, dfgpu_table_t* out) {
  return guarded([&] {
    require_init();
    Table* t = unwrap(th);
    DFGPU_CHECK(out && column >= 0 && column < (int)t->cols.size(), "bad argument");
    auto o = std::make_unique<Table>(*t);
    o->cols[(size_t)column] = dictionary_encode(t->cols[(size_t)column], sorted != 0);
    *out = wrap(o.release());
  });
}
