// sort.hip — K11/K12: SortExec and TopK on device.
//
// Reference: sort_batch = arrow-ord lexsort_to_indices + take_arrays (physical-plan/src/sorts/
// sort.rs:894-914); TopK::insert_batch row-encodes the sort keys with arrow-row's RowConverter and
// keeps the k smallest rows in a heap (physical-plan/src/topk/mod.rs:397-470).
//
// Device design: the same idea as arrow-row — every row's sort key is normalised into an
// order-preserving byte string (per column: optional null byte placed by `nulls_first`, value
// big-endian with the sign bit flipped / f64 total-order transform, all value bytes inverted for
// DESC) — packed into up to three u64 words.  Then
//   full sort : stable LSD radix sort (8-bit digits) of (key words, row id); digits on which all
//               rows agree are skipped; one wave64 owns one tile and ranks its 64-row chunks with
//               ballot-built peer masks, so the scatter is stable and deterministic;
//   TopK      : MSD radix *select* — one histogram pass per key byte narrows the candidate set to
//               the bucket holding the k-th row — then the few survivors are compacted and fully
//               sorted.  (Q3: k = 10 over ~10^6 groups = 1-2 passes.)
// Finally the output columns are gathered by the sorted row ids (take).  Ties keep input order.
#include "device.hpp"
#include "internal.hpp"

namespace dfgpu {

constexpr int MAX_SORT_KEYS = 8;
constexpr int MAX_KEY_BYTES = 24;

struct SortCol {
  const void* data;
  const uint64_t* valid;
  int type;
  int desc;
  int nulls_first;
  int with_null_byte;
};
struct SortCols {
  SortCol c[MAX_SORT_KEYS];
  int n;
  int key_bytes;
};

__device__ __forceinline__ int value_bytes(int type) {
  switch (type) {
    case DFGPU_INT32: case DFGPU_UINT32: case DFGPU_DATE32: return 4;
    case DFGPU_INT64: case DFGPU_UINT64: case DFGPU_FLOAT64: return 8;
    case DFGPU_DECIMAL128: return 16;
    default: return 1;
  }
}

// normalised key of row i, MSB-first into kb[0..key_bytes)
__device__ __forceinline__ void encode_row(const SortCols& sc, int64_t i, uint8_t* kb) {
  int pos = 0;
  for (int k = 0; k < sc.n; k++) {
    const SortCol& c = sc.c[k];
    bool ok = !c.valid || bit_at(c.valid, i);
    if (c.with_null_byte) kb[pos++] = ok ? 1 : (c.nulls_first ? 0 : 2);
    int nb = value_bytes(c.type);
    uint64_t lo = 0, hi = 0;
    if (ok) {
      switch (c.type) {
        case DFGPU_INT32: case DFGPU_DATE32: lo = (uint32_t)((const int32_t*)c.data)[i] ^ 0x80000000u; break;
        case DFGPU_UINT32: lo = ((const uint32_t*)c.data)[i]; break;
        case DFGPU_INT64: lo = ((const uint64_t*)c.data)[i] ^ 0x8000000000000000ull; break;
        case DFGPU_UINT64: lo = ((const uint64_t*)c.data)[i]; break;
        case DFGPU_FLOAT64: {
          uint64_t b = ((const uint64_t*)c.data)[i];
          lo = (b >> 63) ? ~b : (b ^ 0x8000000000000000ull);  // f64::total_cmp order
          break;
        }
        case DFGPU_DECIMAL128: {
          const uint64_t* p = (const uint64_t*)c.data + 2 * i;
          lo = p[0];
          hi = p[1] ^ 0x8000000000000000ull;
          break;
        }
        default: lo = ((const uint8_t*)c.data)[i]; break;
      }
      if (c.desc) { lo = ~lo; hi = ~hi; }
    }
    for (int b = nb - 1; b >= 0; b--) {
      uint8_t byte = b >= 8 ? (uint8_t)(hi >> ((b - 8) * 8)) : (uint8_t)(lo >> (b * 8));
      kb[pos++] = byte;
    }
  }
}

__global__ __launch_bounds__(BLOCK) void k_norm_keys(SortCols sc, int64_t n, int nwords, uint64_t* __restrict__ w0, uint64_t* __restrict__ w1,
                                                     uint64_t* __restrict__ w2, uint32_t* __restrict__ idx) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    uint8_t kb[MAX_KEY_BYTES];
#pragma unroll
    for (int b = 0; b < MAX_KEY_BYTES; b++) kb[b] = 0;
    encode_row(sc, i, kb);
    uint64_t w[3] = {0, 0, 0};
#pragma unroll
    for (int b = 0; b < MAX_KEY_BYTES; b++) w[b >> 3] |= (uint64_t)kb[b] << ((7 - (b & 7)) * 8);
    w0[i] = w[0];
    if (nwords > 1) w1[i] = w[1];
    if (nwords > 2) w2[i] = w[2];
    idx[i] = (uint32_t)i;
  }
}

// byte `b` (0 = most significant) of the normalised key of element i
struct KeyWords {
  const uint64_t* w[3];
};
__device__ __forceinline__ unsigned key_byte(const KeyWords& k, int64_t i, int b) { return (unsigned)(k.w[b >> 3][i] >> ((7 - (b & 7)) * 8)) & 0xFFu; }

// global histogram of every key byte (decides which LSD passes can be skipped)
__global__ __launch_bounds__(BLOCK) void k_all_digit_hist(KeyWords k, int64_t n, int key_bytes, unsigned long long* __restrict__ hist) {
  __shared__ unsigned int sh[MAX_KEY_BYTES * 256];
  for (int x = threadIdx.x; x < key_bytes * 256; x += BLOCK) sh[x] = 0;
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK)
    for (int b = 0; b < key_bytes; b++) atomicAdd(&sh[b * 256 + key_byte(k, i, b)], 1u);
  __syncthreads();
  for (int x = threadIdx.x; x < key_bytes * 256; x += BLOCK)
    if (sh[x]) atomicAdd(&hist[x], (unsigned long long)sh[x]);
}

// per-tile digit histogram; one wave per tile.  counts[digit * n_tiles + tile]
__global__ __launch_bounds__(WAVE) void k_tile_hist(KeyWords k, int64_t n, int byte, int64_t tile, int64_t n_tiles, uint32_t* __restrict__ counts) {
  __shared__ unsigned int sh[256];
  for (int64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    for (int x = threadIdx.x; x < 256; x += WAVE) sh[x] = 0;
    __syncthreads();
    int64_t lo = t * tile, hi = lo + tile < n ? lo + tile : n;
    for (int64_t i = lo + threadIdx.x; i < hi; i += WAVE) atomicAdd(&sh[key_byte(k, i, byte)], 1u);
    __syncthreads();
    for (int x = threadIdx.x; x < 256; x += WAVE) counts[(int64_t)x * n_tiles + t] = sh[x];
    __syncthreads();
  }
}

struct SortBufs {
  uint64_t* w[3];
  uint32_t* idx;
};
// stable scatter of one tile by one digit
__global__ __launch_bounds__(WAVE) void k_tile_scatter(KeyWords k, const uint32_t* __restrict__ idx_in, int64_t n, int byte, int nwords, int64_t tile,
                                                       int64_t n_tiles, const uint64_t* __restrict__ offsets, SortBufs out) {
  __shared__ unsigned long long off[256];
  for (int64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    for (int x = threadIdx.x; x < 256; x += WAVE) off[x] = offsets[(int64_t)x * n_tiles + t];
    __syncthreads();
    int64_t lo = t * tile, hi = lo + tile < n ? lo + tile : n;
    for (int64_t base = lo; base < hi; base += WAVE) {
      int64_t i = base + threadIdx.x;
      bool in = i < hi;
      unsigned d = in ? key_byte(k, i, byte) : 0u;
      // peers = lanes of this chunk with the same digit
      uint64_t peers = ballot64(in);
#pragma unroll
      for (int b = 0; b < 8; b++) {
        uint64_t bal = ballot64((d >> b) & 1u);
        peers &= ((d >> b) & 1u) ? bal : ~bal;
      }
      unsigned rank = mbcnt(peers);
      unsigned long long dst = in ? off[d] + rank : 0ull;
      __syncthreads();
      if (in && rank == 0) off[d] += (unsigned long long)__popcll(peers);
      __syncthreads();
      if (in) {
        out.w[0][dst] = k.w[0][i];
        if (nwords > 1) out.w[1][dst] = k.w[1][i];
        if (nwords > 2) out.w[2][dst] = k.w[2][i];
        out.idx[dst] = idx_in[i];
      }
    }
    __syncthreads();
  }
}

// TopK narrowing: histogram of byte `b` over candidate rows
__global__ __launch_bounds__(BLOCK) void k_select_hist(KeyWords k, const uint8_t* __restrict__ state, int64_t n, int byte, unsigned long long* __restrict__ hist) {
  __shared__ unsigned int sh[256];
  for (int x = threadIdx.x; x < 256; x += BLOCK) sh[x] = 0;
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK)
    if (state[i] == 1) atomicAdd(&sh[key_byte(k, i, byte)], 1u);
  __syncthreads();
  for (int x = threadIdx.x; x < 256; x += BLOCK)
    if (sh[x]) atomicAdd(&hist[x], (unsigned long long)sh[x]);
}
// state: 0 out, 1 candidate, 2 selected.  digit < pivot -> selected, == pivot stays candidate, > out
__global__ __launch_bounds__(BLOCK) void k_select_apply(KeyWords k, uint8_t* __restrict__ state, int64_t n, int byte, unsigned pivot) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    if (state[i] != 1) continue;
    unsigned d = key_byte(k, i, byte);
    state[i] = d < pivot ? 2 : (d == pivot ? 1 : 0);
  }
}
__global__ __launch_bounds__(BLOCK) void k_state_mask(const uint8_t* __restrict__ state, int64_t n, uint64_t* __restrict__ mask) {
  const int64_t n_words = (n + 63) >> 6;
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * BLOCK) >> 6;
  for (int64_t w = wave; w < n_words; w += n_waves) {
    int64_t i = (w << 6) + lane_id();
    uint64_t m = ballot64(i < n && state[i] != 0);
    if (lane_id() == 0) mask[w] = m;
  }
}
__global__ __launch_bounds__(BLOCK) void k_fill_bytes(uint8_t v, int64_t n, uint8_t* out) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) out[i] = v;
}
__global__ __launch_bounds__(BLOCK) void k_idx_to_i64(const uint32_t* __restrict__ idx, const int64_t* __restrict__ remap, int64_t n, int64_t* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    int64_t v = idx[i];
    out[i] = remap ? remap[v] : v;
  }
}
// row ids of the set bits of a mask, in order: ids[prefix + rank] = row
__global__ __launch_bounds__(BLOCK) void k_mask_to_ids(const uint64_t* __restrict__ mask, const uint64_t* __restrict__ prefix, int64_t n, int64_t* __restrict__ ids) {
  const int64_t n_words = (n + 63) >> 6;
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * BLOCK) >> 6;
  for (int64_t w = wave; w < n_words; w += n_waves) {
    uint64_t m = mask[w];
    if ((m >> lane_id()) & 1ull) ids[prefix[w] + mbcnt(m)] = (w << 6) + lane_id();
  }
}

__global__ void k_iota_u32(int64_t n, uint32_t* out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = (uint32_t)i;
}

// ------------------------------------------------------------------------------- host
struct SortedKeys {
  BufPtr w[3];
  BufPtr idx;
  int nwords = 0;
};

// stable LSD radix sort of n (key, idx) elements; returns the buffers holding the result
static SortedKeys radix_sort(SortedKeys in, int64_t n, int key_bytes) {
  Runtime& r = rt();
  if (n <= 1) return in;
  const int nwords = in.nwords;
  KeyWords kw{{in.w[0]->as<uint64_t>(), nwords > 1 ? in.w[1]->as<uint64_t>() : nullptr, nwords > 2 ? in.w[2]->as<uint64_t>() : nullptr}};
  // which digits vary?
  BufPtr gh = make_zero_buf((size_t)key_bytes * 256 * 8);
  k_all_digit_hist<<<grid_for(n, BLOCK * 4), BLOCK, 0, r.stream>>>(kw, n, key_bytes, gh->as<unsigned long long>());
  std::vector<unsigned long long> h((size_t)key_bytes * 256);
  d2h(h.data(), gh->ptr, h.size() * 8);
  std::vector<int> active;
  for (int b = key_bytes - 1; b >= 0; b--) {  // least significant byte first
    bool constant = false;
    for (int d = 0; d < 256; d++)
      if (h[(size_t)b * 256 + d] == (unsigned long long)n) constant = true;
    if (!constant) active.push_back(b);
  }
  if (active.empty()) return in;
  int64_t tile = (n + 4095) / 4096;
  tile = std::max<int64_t>(512, std::min<int64_t>(8192, (tile + 63) / 64 * 64));
  const int64_t n_tiles = (n + tile - 1) / tile;
  SortedKeys cur = in, alt;
  alt.nwords = nwords;
  for (int wd = 0; wd < nwords; wd++) alt.w[wd] = make_buf((size_t)n * 8);
  alt.idx = make_buf((size_t)n * 4);
  BufPtr counts = make_buf((size_t)256 * n_tiles * 4);
  BufPtr offsets = make_buf((size_t)(256 * n_tiles + 1) * 8);
  int grid = (int)std::min<int64_t>(n_tiles, 256 * 16);
  for (int b : active) {
    KeyWords ck{{cur.w[0]->as<uint64_t>(), nwords > 1 ? cur.w[1]->as<uint64_t>() : nullptr, nwords > 2 ? cur.w[2]->as<uint64_t>() : nullptr}};
    SortBufs ob{{alt.w[0]->as<uint64_t>(), nwords > 1 ? alt.w[1]->as<uint64_t>() : nullptr, nwords > 2 ? alt.w[2]->as<uint64_t>() : nullptr}, alt.idx->as<uint32_t>()};
    ProfileScope ps("radix_sort_pass", n * (nwords * 8 + 4) * 2);
    k_tile_hist<<<grid, WAVE, 0, r.stream>>>(ck, n, b, tile, n_tiles, counts->as<uint32_t>());
    scan_u32(counts->as<uint32_t>(), 256 * n_tiles, offsets->as<uint64_t>());
    k_tile_scatter<<<grid, WAVE, 0, r.stream>>>(ck, cur.idx->as<uint32_t>(), n, b, nwords, tile, n_tiles, offsets->as<uint64_t>(), ob);
    DFGPU_HIP(hipGetLastError());
    std::swap(cur, alt);
  }
  return cur;
}

static Table sort_table(const Table& in, const std::vector<int>& key_cols, const uint8_t* desc, const uint8_t* nulls_first, int64_t fetch) {
  Runtime& r = rt();
  const int64_t n = in.nrows;
  DFGPU_CHECK(n < 0xFFFFFFFFll, "sort input exceeds u32 row ids");
  DFGPU_CHECK(!key_cols.empty() && (int)key_cols.size() <= MAX_SORT_KEYS, "bad number of sort keys");
  SortCols sc{};
  sc.n = (int)key_cols.size();
  int kbytes = 0;
  for (int k = 0; k < sc.n; k++) {
    DFGPU_CHECK(key_cols[k] >= 0 && key_cols[k] < (int)in.cols.size(), "sort key column out of range");
    const Column& c = in.cols[key_cols[k]];
    DFGPU_CHECK(c.field.type != DFGPU_BOOL, "Boolean sort keys are not supported on the GPU path");
    sc.c[k] = SortCol{c.ptr(), c.valid_words(), c.field.type, desc[k] != 0, nulls_first[k] != 0, c.validity != nullptr};
    kbytes += (c.validity ? 1 : 0) + (c.field.type == DFGPU_UINT8 ? 1 : type_width(c.field.type));
  }
  DFGPU_CHECK(kbytes <= MAX_KEY_BYTES, "normalised sort key longer than 24 bytes is not supported on the GPU path");
  sc.key_bytes = kbytes;
  const int nwords = (kbytes + 7) / 8;
  int64_t n_out = fetch >= 0 ? std::min(fetch, n) : n;

  Table out;
  out.nrows = n_out;
  if (n_out == 0) {
    for (auto& c : in.cols) out.cols.push_back(alloc_column(c.field, c.name, 0));
    return out;
  }
  SortedKeys sk;
  sk.nwords = nwords;
  for (int wd = 0; wd < nwords; wd++) sk.w[wd] = make_buf((size_t)n * 8);
  sk.idx = make_buf((size_t)n * 4);
  {
    int64_t kb = 0;
    for (int k = 0; k < sc.n; k++) kb += n * (sc.c[k].type == DFGPU_UINT8 ? 1 : type_width(sc.c[k].type));
    ProfileScope ps("sort_normalize_keys", kb + n * (nwords * 8 + 4));
    k_norm_keys<<<grid_for(n, BLOCK), BLOCK, 0, r.stream>>>(sc, n, nwords, sk.w[0]->as<uint64_t>(), nwords > 1 ? sk.w[1]->as<uint64_t>() : nullptr,
                                                           nwords > 2 ? sk.w[2]->as<uint64_t>() : nullptr, sk.idx->as<uint32_t>());
    DFGPU_HIP(hipGetLastError());
  }
  BufPtr remap;  // survivor position -> original row id (TopK path)
  int64_t m = n;
  if (fetch >= 0 && n > 4096 && n_out < n / 4) {
    // ---- TopK: MSD radix select narrows to the rows that can still be among the first k
    KeyWords kw{{sk.w[0]->as<uint64_t>(), nwords > 1 ? sk.w[1]->as<uint64_t>() : nullptr, nwords > 2 ? sk.w[2]->as<uint64_t>() : nullptr}};
    BufPtr state = make_buf((size_t)n + 64);
    k_fill_bytes<<<grid_for(n, BLOCK), BLOCK, 0, r.stream>>>(1, n, state->as<uint8_t>());
    BufPtr hist = make_buf(256 * 8);
    int64_t selected = 0, candidates = n;
    for (int b = 0; b < kbytes && selected + candidates > std::max<int64_t>(4096, 2 * n_out); b++) {
      DFGPU_HIP(hipMemsetAsync(hist->ptr, 0, 256 * 8, r.stream));
      ProfileScope ps("topk_select_pass", n * 9);
      k_select_hist<<<grid_for(n, BLOCK * 4), BLOCK, 0, r.stream>>>(kw, state->as<uint8_t>(), n, b, hist->as<unsigned long long>());
      unsigned long long h[256];
      d2h(h, hist->ptr, sizeof h);
      int64_t need = n_out - selected, acc = 0;
      unsigned pivot = 255;
      for (unsigned d = 0; d < 256; d++) {
        if (acc + (int64_t)h[d] >= need) { pivot = d; break; }
        acc += (int64_t)h[d];
      }
      k_select_apply<<<grid_for(n, BLOCK), BLOCK, 0, r.stream>>>(kw, state->as<uint8_t>(), n, b, pivot);
      selected += acc;
      candidates = (int64_t)h[pivot];
    }
    // compact survivors (selected + remaining candidates), preserving input order
    const int64_t n_words = (n + 63) / 64;
    BufPtr mask = make_buf(bitmap_bytes(n));
    BufPtr prefix = make_buf((size_t)(n_words + 1) * 8);
    k_state_mask<<<grid_for(n_words, BLOCK / WAVE), BLOCK, 0, r.stream>>>(state->as<uint8_t>(), n, mask->as<uint64_t>());
    scan_mask_popcounts(mask->as<uint64_t>(), nullptr, n, prefix->as<uint64_t>());
    m = (int64_t)read_u64(prefix->as<uint64_t>() + n_words);
    remap = make_buf((size_t)m * 8);
    k_mask_to_ids<<<grid_for(n_words, BLOCK / WAVE), BLOCK, 0, r.stream>>>(mask->as<uint64_t>(), prefix->as<uint64_t>(), n, remap->as<int64_t>());
    // survivors' normalised keys, re-indexed 0..m-1
    SortedKeys sv;
    sv.nwords = nwords;
    dfgpu_field f64w{};
    f64w.type = DFGPU_UINT64;
    for (int wd = 0; wd < nwords; wd++) {
      Column kc;
      kc.field = f64w;
      kc.length = n;
      kc.data = sk.w[wd];
      sv.w[wd] = gather_column(kc, remap->as<int64_t>(), m, false).data;
    }
    // survivor ids 0..m-1 are positions into `remap`
    sv.idx = make_buf((size_t)(m ? m : 1) * 4);
    if (m) k_iota_u32<<<grid_for(m, BLOCK), BLOCK, 0, r.stream>>>(m, sv.idx->as<uint32_t>());
    sk = sv;
  }
  SortedKeys sorted = radix_sort(sk, m, kbytes);
  BufPtr take_idx = make_buf((size_t)n_out * 8);
  k_idx_to_i64<<<grid_for(n_out, BLOCK), BLOCK, 0, r.stream>>>(sorted.idx->as<uint32_t>(), remap ? remap->as<int64_t>() : nullptr, n_out, take_idx->as<int64_t>());
  DFGPU_HIP(hipGetLastError());
  for (auto& c : in.cols) out.cols.push_back(gather_column(c, take_idx->as<int64_t>(), n_out, false));
  DFGPU_HIP(hipStreamSynchronize(r.stream));
  return out;
}

}  // namespace dfgpu

using namespace dfgpu;

extern "C" int dfgpu_sort(dfgpu_table_t input, const int* key_cols, const uint8_t* descending, const uint8_t* nulls_first, int nkeys, int64_t fetch,
                          dfgpu_table_t* out) {
  return guarded([&] {
    require_init();
    auto t = std::make_unique<Table>(sort_table(*unwrap(input), std::vector<int>(key_cols, key_cols + nkeys), descending, nulls_first, fetch));
    *out = wrap(t.release());
  });
}
