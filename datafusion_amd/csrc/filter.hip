// filter.hip — K8: FilterExec.  predicate -> bit mask (expr.hip) -> per-word popcount prefix
// (scan.hip) -> order-preserving compaction of every projected column.
//
// Compaction is wave-granular: one wave64 owns one 64-row mask word; the destination of lane l
// is prefix[word] + popcount(mask & lanes_below(l)) (v_mbcnt), so selected lanes store to
// consecutive addresses and no LDS or workgroup barrier is needed.  HBM traffic per row:
// N/8 B mask (x3: written once, read by scan and by compaction) + the columns themselves.
//
// Reference: physical-plan/src/filter.rs:1339-1362 (filter_and_project), :1396-1419;
// arrow-select `filter_record_batch` (NULL predicate => row dropped).
#include "device.hpp"
#include "internal.hpp"

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
  if (lane_id() == 0 && s) atomicAdd(out, (unsigned long long)s);
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
    out.cols.push_back(alloc_like(c, n_out));
    if (c.validity) valid_bytes[i] = make_buf((size_t)n_out + 64);
  }
  if (n_out > 0) {
    // byte-addressable columns, MAX_COLS per launch; Boolean (bit-packed) columns one by one afterwards
    std::vector<int> wide, bits;
    for (size_t i = 0; i < cols.size(); i++) (in.cols[cols[i]].field.type == DFGPU_BOOL ? bits : wide).push_back((int)i);
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

// arrow `take` (joins/utils.rs:1371,1379 build_batch_from_indices): idx < 0 => NULL
Column gather_column(const Column& in, const int64_t* idx, int64_t n, bool idx_may_be_null) {
  Runtime& r = rt();
  DFGPU_CHECK(in.field.type != DFGPU_BOOL, "take: Boolean columns are not supported on the GPU path yet");
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

}  // namespace dfgpu

using namespace dfgpu;

extern "C" int dfgpu_filter(dfgpu_table_t input, const dfgpu_expr* predicate, const int* projection, int nproj, dfgpu_table_t* out) {
  return guarded([&] {
    require_init();
    Table* t = unwrap(input);
    DFGPU_CHECK(predicate && out, "null argument");
    std::vector<int> cols;
    if (projection) cols.assign(projection, projection + nproj);
    else for (int i = 0; i < (int)t->cols.size(); i++) cols.push_back(i);
    Datum d = evaluate(*predicate, *t);
    DFGPU_CHECK(d.col.field.type == DFGPU_BOOL || d.scalar, "filter predicate must be Boolean");
    auto o = std::make_unique<Table>();
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
