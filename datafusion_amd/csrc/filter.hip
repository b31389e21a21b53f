// filter.hip — K8: FilterExec.  predicate -> bit mask (expr.hip) -> per-word popcount prefix
// (scan.hip) -> order-preserving compaction of every projected column.
//
// Compaction is wave-granular: one wave64 owns one 64-row mask word; the destination of lane l
// is prefix[word] + popcount(mask & lanes_below(l)) (v_mbcnt), so selected lanes store to
// consecutive addresses and no LDS or workgroup barrier is needed.  HBM traffic per row:
// N/8 B mask (x3: written once, read by scan and by compaction) + the columns themselves.
// Measured (profiles/r2_ops_v1.md): k_compact moves 5.7-5.8 TB/s of algorithmic bytes = 91-92 % of the 6.29 TB/s copy
// ceiling.  A variant that packs a wave's 256 / 512 rows through wave-private LDS so that every store is a full 64-lane
// store was tried in round 2 and ran 1.7-2.3x SLOWER (LDS round trip + the occupancy lost to 16-32 KB of LDS per
// workgroup buy nothing: L2 already merges the per-word pieces) — what is left of FilterExec's gap is the predicate
// pass, the scan and one host round trip for the output size, a fixed ~0.17 ms that weighs on a 1 ms SF10 operator.
//
// Reference: physical-plan/src/filter.rs:1339-1362 (filter_and_project), :1396-1419;
// arrow-select `filter_record_batch` (NULL predicate => row dropped).
#include <algorithm>
#include <cstdlib>

#include "device.hpp"
#include "internal.hpp"
#include "records.hpp"

namespace dfgpu {

constexpr int MAX_COLS = 12;
struct CopyCols {
  const void* src[MAX_COLS];
  void* dst[MAX_COLS];
  const uint64_t* src_valid[MAX_COLS];  // optional
  uint8_t* dst_valid_bytes[MAX_COLS];   // one byte per output row when src_valid is set
  int width[MAX_COLS];
  int n;
};

template <typename T>
__device__ __forceinline__ void copy_elem(const void* src, void* dst, int64_t s, int64_t d) {
  // (plain forms on purpose: neighbouring waves write pieces of the same output lines and L2 merges them; with non-temporal
  // stores every piece went to HBM on its own and the compaction ran 10-15 % slower, non-temporal loads changed nothing)
  reinterpret_cast<T*>(dst)[d] = reinterpret_cast<const T*>(src)[s];
}
__device__ __forceinline__ void copy_by_width(int width, const void* src, void* dst, int64_t s, int64_t d) {
  switch (width) {
    case 16: copy_elem<uint4>(src, dst, s, d); break;
    case 8: copy_elem<uint64_t>(src, dst, s, d); break;
    case 4: copy_elem<uint32_t>(src, dst, s, d); break;
    case 1: copy_elem<uint8_t>(src, dst, s, d); break;
  }
}

constexpr int COMPACT_UNROLL = 4;  // mask words per wave iteration (memory-level parallelism)

__global__ __launch_bounds__(BLOCK) void k_compact(CopyCols cols, const uint64_t* __restrict__ mask,
                                                   const uint64_t* __restrict__ mask_valid,
                                                   const uint64_t* __restrict__ prefix, int64_t nrows) {
  const int64_t n_words = (nrows + 63) >> 6;
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * BLOCK) >> 6;
  const unsigned lane = lane_id();
  for (int64_t w0 = wave * COMPACT_UNROLL; w0 < n_words; w0 += n_waves * COMPACT_UNROLL) {
    bool sel[COMPACT_UNROLL];
    int64_t dst[COMPACT_UNROLL];
#pragma unroll
    for (int j = 0; j < COMPACT_UNROLL; j++) {
      int64_t w = w0 + j;
      uint64_t m = 0;
      if (w < n_words) {
        m = mask[w];
        if (mask_valid) m &= mask_valid[w];
        int64_t rem = nrows - (w << 6);
        if (rem < 64) m &= (~0ull) >> (64 - rem);
      }
      sel[j] = (m >> lane) & 1ull;
      dst[j] = sel[j] ? (int64_t)(prefix[w] + mbcnt(m)) : 0;
    }
    for (int c = 0; c < cols.n; c++) {
      const int width = cols.width[c];
#pragma unroll
      for (int j = 0; j < COMPACT_UNROLL; j++) {
        if (sel[j]) {
          int64_t row = ((w0 + j) << 6) + lane;
          copy_by_width(width, cols.src[c], cols.dst[c], row, dst[j]);
          if (cols.src_valid[c]) cols.dst_valid_bytes[c][dst[j]] = bit_at(cols.src_valid[c], row) ? 1 : 0;
        }
      }
    }
  }
}

// compaction of a bit-packed (Boolean) column: the selected rows' bits (and validity bits) as one byte per output row,
// packed to words afterwards (k_pack_bytes) — same offsets as k_compact
__global__ __launch_bounds__(BLOCK) void k_compact_bits(const uint64_t* __restrict__ src_bits, const uint64_t* __restrict__ src_valid, const uint64_t* __restrict__ mask,
                                                       const uint64_t* __restrict__ mask_valid, const uint64_t* __restrict__ prefix, int64_t n,
                                                       uint8_t* __restrict__ dst_bytes, uint8_t* __restrict__ dst_valid_bytes) {
  const int64_t n_words = (n + 63) >> 6;
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * BLOCK) >> 6;
  const unsigned lane = lane_id();
  for (int64_t w = wave; w < n_words; w += n_waves) {
    uint64_t m = mask[w];
    if (mask_valid) m &= mask_valid[w];
    const int64_t rem = n - (w << 6);
    if (rem < 64) m &= (~0ull) >> (64 - rem);
    if (!((m >> lane) & 1ull)) continue;
    const int64_t d = (int64_t)(prefix[w] + mbcnt(m));
    dst_bytes[d] = (uint8_t)((src_bits[w] >> lane) & 1ull);
    if (dst_valid_bytes) dst_valid_bytes[d] = (uint8_t)((src_valid[w] >> lane) & 1ull);
  }
}

// one byte per row -> Arrow bitmap (wave ballot = one 64-bit word per wave)
__global__ __launch_bounds__(BLOCK) void k_pack_bytes(const uint8_t* __restrict__ bytes, int64_t n, uint64_t* __restrict__ words) {
  const int64_t n_words = (n + 63) >> 6;
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * BLOCK) >> 6;
  for (int64_t w = wave; w < n_words; w += n_waves) {
    int64_t i = (w << 6) + lane_id();
    uint64_t b = ballot64(i < n && bytes[i] != 0);
    if (lane_id() == 0) words[w] = b;
  }
}

__global__ __launch_bounds__(BLOCK) void k_count_bits(const uint64_t* __restrict__ words, int64_t nrows, unsigned long long* out) {
  const int64_t n_words = (nrows + 63) >> 6;
  uint64_t s = 0;
  for (int64_t w = (int64_t)blockIdx.x * BLOCK + threadIdx.x; w < n_words; w += (int64_t)gridDim.x * BLOCK) {
    uint64_t m = words[w];
    int64_t rem = nrows - (w << 6);
    if (rem < 64) m &= (~0ull) >> (64 - rem);
    s += __popcll(m);
  }
  s = wave_sum(s);
  // one atomic per workgroup (8192 waves' atomics on one address at the kernel's end were 0.08 of its 0.10 ms over 150 M bits)
  __shared__ unsigned long long s_part[BLOCK / WAVE];
  if (lane_id() == 0) s_part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
#pragma unroll
    for (int w = 0; w < BLOCK / WAVE; w++) t += s_part[w];
    if (t) atomicAdd(out, t);
  }
}

void pack_bytes_to_bitmap(const uint8_t* bytes, int64_t n, uint64_t* words) {
  if (n == 0) return;
  ProfileScope ps("pack_bytes", n + n / 8);
  k_pack_bytes<<<grid_for((n + 63) / 64, BLOCK / WAVE), BLOCK, 0, rt().stream>>>(bytes, n, words);
}

void count_nulls(Column& c) {
  if (!c.validity) {
    c.null_count = 0;
    return;
  }
  if (c.length == 0) {
    c.null_count = 0;
    c.validity.reset();
    return;
  }
  BufPtr cnt = make_zero_buf(8);
  k_count_bits<<<grid_for((c.length + 63) / 64, BLOCK), BLOCK, 0, rt().stream>>>(c.valid_words(), c.length, cnt->as<unsigned long long>());
  uint64_t valid = read_u64(cnt->as<uint64_t>());
  c.null_count = c.length - (int64_t)valid;
  if (c.null_count == 0) c.validity.reset();
}

Table compact_table(const Table& in, const std::vector<int>& cols, const uint64_t* mask, const uint64_t* mask_valid) {
  Runtime& r = rt();
  const int64_t n = in.nrows;
  const int64_t n_words = (n + 63) / 64;
  Table out;
  BufPtr prefix = make_buf((size_t)(n_words + 1) * 8);
  scan_mask_popcounts(mask, mask_valid, n, prefix->as<uint64_t>());
  const int64_t n_out = (int64_t)read_u64(prefix->as<uint64_t>() + n_words);
  out.nrows = n_out;
  std::vector<BufPtr> valid_bytes(cols.size());
  for (size_t i = 0; i < cols.size(); i++) {
    DFGPU_CHECK(cols[i] >= 0 && cols[i] < (int)in.cols.size(), "projection index out of range");
    const Column& c = in.cols[cols[i]];
    if (c.field.type == DFGPU_UTF8) {  // strings: lengths -> scan -> byte copy (strings.hip); done here, skipped below
      out.cols.push_back(compact_strings(c, mask, mask_valid, prefix->as<uint64_t>(), n, n_out));
      continue;
    }
    out.cols.push_back(alloc_like(c, n_out));
    if (c.validity) valid_bytes[i] = make_buf((size_t)n_out + 64);
  }
  if (n_out > 0) {
    // byte-addressable columns, MAX_COLS per launch; Boolean (bit-packed) columns one by one afterwards
    std::vector<int> wide, bits;
    for (size_t i = 0; i < cols.size(); i++) {
      if (in.cols[cols[i]].field.type == DFGPU_UTF8) continue;
      (in.cols[cols[i]].field.type == DFGPU_BOOL ? bits : wide).push_back((int)i);
    }
    for (int i : bits) {
      const Column& c = in.cols[cols[i]];
      BufPtr vals = make_buf((size_t)n_out + 64);
      k_compact_bits<<<grid_for(n_words, BLOCK / WAVE), BLOCK, 0, r.stream>>>((const uint64_t*)c.ptr(), c.valid_words(), mask, mask_valid, prefix->as<uint64_t>(), n,
                                                                               vals->as<uint8_t>(), valid_bytes[i] ? valid_bytes[i]->as<uint8_t>() : nullptr);
      DFGPU_HIP(hipGetLastError());
      pack_bytes_to_bitmap(vals->as<uint8_t>(), n_out, out.cols[i].data->as<uint64_t>());
    }
    for (size_t c0 = 0; c0 < wide.size(); c0 += MAX_COLS) {
      CopyCols cc{};
      int64_t bytes = 0;
      cc.n = (int)std::min<size_t>(MAX_COLS, wide.size() - c0);
      for (int k = 0; k < cc.n; k++) {
        const Column& c = in.cols[cols[wide[c0 + k]]];
        cc.src[k] = c.ptr();
        cc.dst[k] = out.cols[wide[c0 + k]].data->ptr;
        cc.width[k] = type_width(c.field.type);
        cc.src_valid[k] = c.valid_words();
        cc.dst_valid_bytes[k] = valid_bytes[wide[c0 + k]] ? valid_bytes[wide[c0 + k]]->as<uint8_t>() : nullptr;
        bytes += (n + n_out) * cc.width[k];
      }
      ProfileScope ps("compact", bytes + n / 8);
      k_compact<<<grid_for(n_words, (BLOCK / WAVE) * COMPACT_UNROLL), BLOCK, 0, r.stream>>>(cc, mask, mask_valid, prefix->as<uint64_t>(), n);
      DFGPU_HIP(hipGetLastError());
    }
    for (size_t i = 0; i < cols.size(); i++) {
      if (!valid_bytes[i]) continue;
      Column& oc = out.cols[i];
      oc.validity = make_buf(bitmap_bytes(n_out));
      pack_bytes_to_bitmap(valid_bytes[i]->as<uint8_t>(), n_out, oc.validity->as<uint64_t>());
      oc.null_count = -1;
    }
  }
  return out;
}

// ------------------------------------------------------------------------------ take
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_gather(const T* __restrict__ src, const uint64_t* __restrict__ src_valid,
                                                  const int64_t* __restrict__ idx, int64_t n, T* __restrict__ dst,
                                                  uint8_t* __restrict__ dst_valid_bytes) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    int64_t s = idx[i];
    T v{};
    bool ok = s >= 0;
    if (ok) {
      v = src[s];
      if (src_valid) ok = bit_at(src_valid, s);
    }
    dst[i] = v;
    if (dst_valid_bytes) dst_valid_bytes[i] = ok ? 1 : 0;
  }
}

// take of a bit-packed (Boolean) column: the taken bits as one byte per output row, packed afterwards
__global__ __launch_bounds__(BLOCK) void k_gather_bits(const uint64_t* __restrict__ src, const uint64_t* __restrict__ src_valid, const int64_t* __restrict__ idx, int64_t n,
                                                      uint8_t* __restrict__ dst_bytes, uint8_t* __restrict__ dst_valid_bytes) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    const int64_t s = idx[i];
    bool ok = s >= 0, v = false;
    if (ok) {
      v = (src[s >> 6] >> (s & 63)) & 1ull;
      if (src_valid) ok = bit_at(src_valid, s);
    }
    dst_bytes[i] = v ? 1 : 0;
    if (dst_valid_bytes) dst_valid_bytes[i] = ok ? 1 : 0;
  }
}

// arrow `take` (joins/utils.rs:1371,1379 build_batch_from_indices): idx < 0 => NULL
Column gather_column(const Column& in, const int64_t* idx, int64_t n, bool idx_may_be_null) {
  Runtime& r = rt();
  if (in.field.type == DFGPU_UTF8) return gather_strings(in, idx, n, idx_may_be_null);
  if (in.field.type == DFGPU_BOOL) {
    Column out = alloc_like(in, n);
    if (n == 0) return out;
    const bool need_valid = idx_may_be_null || in.validity;
    BufPtr vals = make_buf((size_t)n + 64), vb = need_valid ? make_buf((size_t)n + 64) : nullptr;
    {
      ProfileScope ps("gather", n * 10);
      k_gather_bits<<<grid_for(n, BLOCK), BLOCK, 0, r.stream>>>((const uint64_t*)in.ptr(), in.valid_words(), idx, n, vals->as<uint8_t>(), vb ? vb->as<uint8_t>() : nullptr);
      DFGPU_HIP(hipGetLastError());
    }
    pack_bytes_to_bitmap(vals->as<uint8_t>(), n, out.data->as<uint64_t>());
    if (need_valid) {
      out.validity = make_buf(bitmap_bytes(n));
      pack_bytes_to_bitmap(vb->as<uint8_t>(), n, out.validity->as<uint64_t>());
      out.null_count = -1;
      count_nulls(out);
    }
    return out;
  }
  Column out = alloc_like(in, n);
  if (n == 0) return out;
  bool need_valid = idx_may_be_null || in.validity;
  BufPtr vb = need_valid ? make_buf((size_t)n + 64) : nullptr;
  uint8_t* vbp = vb ? vb->as<uint8_t>() : nullptr;
  int w = type_width(in.field.type);
  int g = grid_for(n, BLOCK);
  {
    ProfileScope ps("gather", n * (8 + 2 * w));
    switch (w) {
      case 16: k_gather<uint4><<<g, BLOCK, 0, r.stream>>>((const uint4*)in.ptr(), in.valid_words(), idx, n, (uint4*)out.data->ptr, vbp); break;
      case 8: k_gather<uint64_t><<<g, BLOCK, 0, r.stream>>>((const uint64_t*)in.ptr(), in.valid_words(), idx, n, (uint64_t*)out.data->ptr, vbp); break;
      case 4: k_gather<uint32_t><<<g, BLOCK, 0, r.stream>>>((const uint32_t*)in.ptr(), in.valid_words(), idx, n, (uint32_t*)out.data->ptr, vbp); break;
      case 1: k_gather<uint8_t><<<g, BLOCK, 0, r.stream>>>((const uint8_t*)in.ptr(), in.valid_words(), idx, n, (uint8_t*)out.data->ptr, vbp); break;
    }
    DFGPU_HIP(hipGetLastError());
  }
  if (need_valid) {
    out.validity = make_buf(bitmap_bytes(n));
    pack_bytes_to_bitmap(vbp, n, out.validity->as<uint64_t>());
    out.null_count = -1;
    count_nulls(out);
  }
  return out;
}

// ------------------------------------------------------------------------------ take of several columns
// A random access costs a whole 128-byte line of HBM whatever the element width (profiles/r2_fetch_calib.md), so taking k
// columns by the same row ids costs k lines per row column-by-column.  Above the Infinity Cache's size the columns are first
// packed into row-major records with one streaming pass (16 / 32 / 48 / 64 bytes per row), then ONE line per row is touched
// and the record is split into the output columns — arrow `take` over a whole batch (joins/utils.rs:1332-1386, sorts/sort.rs:
// 894-914 take_arrays), laid out for HBM.
template <int R>
__global__ __launch_bounds__(BLOCK) void k_pack_rows(PackLayout L, int64_t n, uint8_t* __restrict__ rec) {
  constexpr int NS = R / 8;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    uint64_t s[NS];
    record_build<NS>(L, i, s);
    record_store<NS>(rec, i, s);
  }
}
template <int R, typename IT>  // IT: row ids as int64 (take) or uint32 (the sort's ids, taken as they are)
__global__ __launch_bounds__(BLOCK) void k_gather_rows(PackLayout L, const uint8_t* __restrict__ rec, const IT* __restrict__ idx, int64_t n) {
  constexpr int NS = R / 8;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    uint64_t s[NS];
    record_load<NS>(rec, (int64_t)idx[i], s);
    record_split<NS>(L, i, s);
  }
}

// ---- records laid out by somebody else (sort.hip's carried sort): can every column of `cols` travel in ONE record?
bool plan_record_layout(const Table& in, const std::vector<int>& cols, PackLayout& L, int& R, std::vector<int>& order) {
  L = PackLayout{};
  order.clear();
  if (cols.empty() || (int)cols.size() > PACK_MAX_COLS) return false;
  for (int c : cols) {
    const Column& col = in.cols[(size_t)c];
    if (col.validity || col.field.type == DFGPU_BOOL || col.field.type == DFGPU_UTF8) return false;
  }
  for (size_t k = 0; k < cols.size(); k++) order.push_back((int)k);
  // widest columns first keeps every field naturally aligned inside the record
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return type_width(in.cols[(size_t)cols[(size_t)a]].field.type) > type_width(in.cols[(size_t)cols[(size_t)b]].field.type); });
  int bytes = 0;
  for (int k : order) {
    const Column& col = in.cols[(size_t)cols[(size_t)k]];
    const int w = type_width(col.field.type);
    if (bytes + w > 64) return false;
    L.width[L.n] = w;
    L.offset[L.n] = bytes;
    L.src[L.n] = col.ptr();
    bytes += w;
    L.n++;
  }
  R = (bytes + 15) / 16 * 16;
  return true;
}
__global__ __launch_bounds__(BLOCK) void k_widen_ids(const uint32_t* __restrict__ in, int64_t n, int64_t* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) out[i] = (int64_t)in[i];
}
// A SMALL take (the few rows a TopK or an aggregate's final sort hands on): every plain column in ONE launch — blockIdx.y is the column.
// Ten launches of a four-row gather were 0.13 ms of a Q1 step whose kernels take 7.4.
constexpr int GM_MAX = 16;
struct GatherMany {
  const void* src[GM_MAX];
  void* dst[GM_MAX];
  int width[GM_MAX];
};
__global__ __launch_bounds__(BLOCK) void k_gather_many(GatherMany g, const int64_t* __restrict__ idx, const uint32_t* __restrict__ idx32, int64_t n) {
  const int c = blockIdx.y;
  const void* src = g.src[c];
  void* dst = g.dst[c];
  const int w = g.width[c];
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    const int64_t s = idx ? idx[i] : (int64_t)idx32[i];
    switch (w) {
      case 16: reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[s]; break;
      case 8: reinterpret_cast<uint64_t*>(dst)[i] = reinterpret_cast<const uint64_t*>(src)[s]; break;
      case 4: reinterpret_cast<uint32_t*>(dst)[i] = reinterpret_cast<const uint32_t*>(src)[s]; break;
      default: reinterpret_cast<uint8_t*>(dst)[i] = reinterpret_cast<const uint8_t*>(src)[s]; break;
    }
  }
}
std::vector<Column> gather_columns(const Table& in, const std::vector<int>& cols, const int64_t* idx, int64_t n, bool idx_may_be_null, const uint32_t* idx32) {
  Runtime& r = rt();
  if (n > 0 && n <= 65536 && !idx_may_be_null && (idx || idx32) && cols.size() >= 2) {
    std::vector<Column> out(cols.size());
    std::vector<int> rest;
    GatherMany gm{};
    int m = 0;
    int64_t bytes = 0;
    for (size_t k = 0; k < cols.size(); k++) {
      const Column& c = in.cols[cols[k]];
      if (m < GM_MAX && !c.validity && c.field.type != DFGPU_BOOL && c.field.type != DFGPU_UTF8) {
        out[k] = alloc_like(c, n);
        gm.src[m] = c.ptr();
        gm.dst[m] = out[k].data->ptr;
        gm.width[m] = type_width(c.field.type);
        bytes += n * (8 + 2 * gm.width[m]);
        m++;
      } else {
        rest.push_back((int)k);
      }
    }
    if (m) {
      ProfileScope ps("gather", bytes);
      k_gather_many<<<dim3((unsigned)grid_for(n, BLOCK), (unsigned)m), BLOCK, 0, r.stream>>>(gm, idx, idx32, n);
      DFGPU_HIP(hipGetLastError());
    }
    if (!rest.empty()) {
      BufPtr widened;
      const int64_t* ids = idx;
      if (!ids) {
        widened = make_buf((size_t)n * 8);
        k_widen_ids<<<grid_for(n, BLOCK), BLOCK, 0, r.stream>>>(idx32, n, widened->as<int64_t>());
        ids = widened->as<int64_t>();
      }
      for (int k : rest) out[(size_t)k] = gather_column(in.cols[cols[(size_t)k]], ids, n, idx_may_be_null);
    }
    return out;
  }
  BufPtr widened;
  auto ids64 = [&]() -> const int64_t* {  // columns that go one by one take 64-bit ids: widened on first use
    if (idx || !idx32) return idx;
    if (!widened) {
      widened = make_buf((size_t)std::max<int64_t>(n, 1) * 8);
      if (n) k_widen_ids<<<grid_for(n, BLOCK), BLOCK, 0, r.stream>>>(idx32, n, widened->as<int64_t>());
    }
    return widened->as<int64_t>();
  };
  std::vector<Column> out(cols.size());
  // what can travel as records: byte-addressable, non-nullable columns taken by non-negative ids, from a table whose columns
  // do not sit in the Infinity Cache anyway, for enough rows to repay the packing pass
  std::vector<int> packable;
  int64_t in_bytes = 0;
  for (size_t k = 0; k < cols.size(); k++) {
    const Column& c = in.cols[cols[k]];
    if (!idx_may_be_null && !c.validity && c.field.type != DFGPU_BOOL && c.field.type != DFGPU_UTF8) {
      packable.push_back((int)k);
      in_bytes += in.nrows * type_width(c.field.type);
    }
  }
  const bool pack = packable.size() >= 2 && in_bytes > ((int64_t)256 << 20) && n * 4 >= in.nrows;
  std::vector<bool> done(cols.size(), false);
  if (pack && n > 0) {
    // widest columns first keeps every field naturally aligned inside the record
    std::sort(packable.begin(), packable.end(), [&](int a, int b) { return type_width(in.cols[cols[a]].field.type) > type_width(in.cols[cols[b]].field.type); });
    size_t at = 0;
    while (at < packable.size()) {
      PackLayout L{};
      int bytes = 0;
      std::vector<int> members;
      while (at < packable.size() && L.n < PACK_MAX_COLS) {
        const int w = type_width(in.cols[cols[packable[at]]].field.type);
        if (bytes + w > 64) break;
        L.width[L.n] = w;
        L.offset[L.n] = bytes;
        L.src[L.n] = in.cols[cols[packable[at]]].ptr();
        bytes += w;
        members.push_back(packable[at]);
        L.n++;
        at++;
      }
      if (members.size() < 2) {  // a lone column gains nothing from a record
        continue;
      }
      const int R = (bytes + 15) / 16 * 16;
      BufPtr rec = make_buf((size_t)in.nrows * R + 64);
      for (int q = 0; q < L.n; q++) {
        out[members[q]] = alloc_like(in.cols[cols[members[q]]], n);
        L.dst[q] = out[members[q]].data->ptr;
        done[members[q]] = true;
      }
      {
        ProfileScope ps("take_pack_rows", in.nrows * (int64_t)(bytes + R));
        const int g = grid_for(in.nrows, BLOCK);
        switch (R) {
          case 16: k_pack_rows<16><<<g, BLOCK, 0, r.stream>>>(L, in.nrows, rec->as<uint8_t>()); break;
          case 32: k_pack_rows<32><<<g, BLOCK, 0, r.stream>>>(L, in.nrows, rec->as<uint8_t>()); break;
          case 48: k_pack_rows<48><<<g, BLOCK, 0, r.stream>>>(L, in.nrows, rec->as<uint8_t>()); break;
          default: k_pack_rows<64><<<g, BLOCK, 0, r.stream>>>(L, in.nrows, rec->as<uint8_t>()); break;
        }
      }
      {
        ProfileScope ps("take_gather_rows", n * (int64_t)(8 + R + bytes));
        const int g = grid_for(n, BLOCK);
        switch (R) {
          case 16: if (idx32) k_gather_rows<16, uint32_t><<<g, BLOCK, 0, r.stream>>>(L, rec->as<uint8_t>(), idx32, n); else k_gather_rows<16, int64_t><<<g, BLOCK, 0, r.stream>>>(L, rec->as<uint8_t>(), idx, n); break;
          case 32: if (idx32) k_gather_rows<32, uint32_t><<<g, BLOCK, 0, r.stream>>>(L, rec->as<uint8_t>(), idx32, n); else k_gather_rows<32, int64_t><<<g, BLOCK, 0, r.stream>>>(L, rec->as<uint8_t>(), idx, n); break;
          case 48: if (idx32) k_gather_rows<48, uint32_t><<<g, BLOCK, 0, r.stream>>>(L, rec->as<uint8_t>(), idx32, n); else k_gather_rows<48, int64_t><<<g, BLOCK, 0, r.stream>>>(L, rec->as<uint8_t>(), idx, n); break;
          default: if (idx32) k_gather_rows<64, uint32_t><<<g, BLOCK, 0, r.stream>>>(L, rec->as<uint8_t>(), idx32, n); else k_gather_rows<64, int64_t><<<g, BLOCK, 0, r.stream>>>(L, rec->as<uint8_t>(), idx, n); break;
        }
        DFGPU_HIP(hipGetLastError());
      }
    }
  }
  for (size_t k = 0; k < cols.size(); k++)
    if (!done[k]) out[k] = gather_column(in.cols[cols[k]], ids64(), n, idx_may_be_null);
  return out;
}

}  // namespace dfgpu

using namespace dfgpu;

// (round 4: a compound predicate — l_shipdate >= a AND l_shipdate <= b — compiled into ONE interpreted row program instead of a launch per
// operator was measured at SF10: 0.553 ms against 0.136 ms for the two comparisons and the AND column at a time — the interpreter's
// per-instruction overhead is four times what the extra passes over a 4-byte column cost.  Only a hiprtc-specialised mask kernel could
// win here; FilterExec stays column at a time.)
extern "C" int dfgpu_filter(dfgpu_table_t input, const dfgpu_expr* predicate, const int* projection, int nproj, dfgpu_table_t* out) {
  return guarded([&] {
    require_init();
    Table* t = unwrap(input);
    DFGPU_CHECK(predicate && out, "null argument");
    std::vector<int> cols;
    if (projection) cols.assign(projection, projection + nproj);
    else for (int i = 0; i < (int)t->cols.size(); i++) cols.push_back(i);
    auto o = std::make_unique<Table>();
    Datum d = evaluate(*predicate, *t);
    DFGPU_CHECK(d.col.field.type == DFGPU_BOOL || d.scalar, "filter predicate must be Boolean");
    if (d.scalar) {
      // literal predicate: all rows or none (FilterExec with a constant predicate)
      bool keep = !d.scalar_null && d.lit_lo != 0;
      if (keep) {
        o->nrows = t->nrows;
        for (int c : cols) o->cols.push_back(t->cols[c]);
      } else {
        o->nrows = 0;
        for (int c : cols) o->cols.push_back(alloc_like(t->cols[c], 0));
      }
    } else {
      *o = compact_table(*t, cols, d.col.data->as<uint64_t>(), d.col.valid_words());
    }
    *out = wrap(o.release());
  });
}
