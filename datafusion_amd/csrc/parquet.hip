// parquet.hip — scan -> device: one Parquet column chunk, as the bytes the reference's reader fetches
// from the object store (datasource-parquet; `ColumnChunkMetaData::byte_range` of the `parquet` crate), decoded straight
// into a device column.
//
// Split of the work:
//   host   page headers (Thrift compact protocol), page decompression (Snappy here, ZSTD through libzstd.so.1 —
//          `tpchgen-cli --parquet-compression 'ZSTD(1)'` is what the reference's benchmarks read, benchmarks/bench.sh:692),
//          definition levels -> validity bitmap (their bit-packing IS the Arrow bitmap layout), and the *run headers*
//          of the RLE / bit-packed hybrid index streams (one varint per run) -> a run table;
//   device one H2D copy of the value bytes per chunk, then ONE kernel decodes every page of the chunk: PLAIN values are
//          widened / byte-swapped to the Arrow layout, dictionary indices are unpacked from their runs and gathered
//          through the chunk's dictionary (kept in HBM, L2-resident: dictionaries are <= 1 MiB by the writers' limits).
// Supported: flat columns (max_repetition_level 0, max_definition_level <= 1), data pages v1 and v2, encodings PLAIN /
// PLAIN_DICTIONARY / RLE_DICTIONARY, physical types INT32 / INT64 / DOUBLE / FIXED_LEN_BYTE_ARRAY (decimals) and
// BYTE_ARRAY — read as field.type DFGPU_UTF8 it becomes a Utf8 column whatever its pages' encodings (decode_string_chunk); read
// as Int32 it must have every data page dictionary-encoded (it becomes a dictionary-encoded string column: indices on the
// device, strings on the host, dictionary sorted ascending).  Anything else is an error: the caller keeps the CPU scan.
#include "device.hpp"
#include "internal.hpp"

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <cstdio>
#include <functional>
#include <map>
#include <mutex>
#include <numeric>
#include <thread>

namespace dfgpu {
namespace {

// ------------------------------------------------------------------------------------------- Thrift compact protocol
struct Thrift {
  const uint8_t* p;
  const uint8_t* end;
  uint8_t byte() {
    DFGPU_CHECK(p < end, "parquet: truncated page header");
    return *p++;
  }
  uint64_t varint() {
    uint64_t v = 0;
    for (int shift = 0; shift < 64; shift += 7) {
      uint8_t b = byte();
      v |= (uint64_t)(b & 0x7F) << shift;
      if (!(b & 0x80)) return v;
    }
    throw Error("parquet: varint too long");
  }
  int64_t zigzag() {
    uint64_t v = varint();
    return (int64_t)(v >> 1) ^ -(int64_t)(v & 1);
  }
  // field header: returns false at STOP; sets type and advances `id`
  bool field(int& type, int& id) {
    uint8_t b = byte();
    if (b == 0) return false;
    type = b & 0x0F;
    int delta = b >> 4;
    id = delta ? id + delta : (int)zigzag();
    return true;
  }
  void skip(int type) {
    switch (type) {
      case 1: case 2: return;                       // BOOL true / false (in the field header)
      case 3: (void)byte(); return;                 // BYTE
      case 4: case 5: case 6: (void)zigzag(); return;  // I16 / I32 / I64
      case 7: DFGPU_CHECK(end - p >= 8, "parquet: truncated double"); p += 8; return;
      case 8: { uint64_t n = varint(); DFGPU_CHECK((uint64_t)(end - p) >= n, "parquet: truncated binary"); p += n; return; }
      case 9: case 10: {                            // LIST / SET
        uint8_t h = byte();
        uint64_t n = h >> 4;
        int et = h & 0x0F;
        if (n == 15) n = varint();
        for (uint64_t i = 0; i < n; i++) {
          if (et == 1 || et == 2) (void)byte();     // bools inside a list are one byte each
          else skip(et);
        }
        return;
      }
      case 11: {                                    // MAP
        uint64_t n = varint();
        if (n == 0) return;
        uint8_t kv = byte();
        for (uint64_t i = 0; i < n; i++) { skip(kv >> 4); skip(kv & 0x0F); }
        return;
      }
      case 12: {                                    // STRUCT
        int t, id = 0;
        while (field(t, id)) skip(t);
        return;
      }
    }
    throw Error("parquet: unknown thrift type " + std::to_string(type));
  }
};

enum { PAGE_DATA = 0, PAGE_INDEX = 1, PAGE_DICTIONARY = 2, PAGE_DATA_V2 = 3 };
enum { ENC_PLAIN = 0, ENC_PLAIN_DICTIONARY = 2, ENC_RLE = 3, ENC_BIT_PACKED = 4, ENC_DELTA_BINARY_PACKED = 5, ENC_RLE_DICTIONARY = 8, ENC_BYTE_STREAM_SPLIT = 9 };

struct PageHeader {
  int type = -1;
  int32_t uncompressed = 0, compressed = 0;
  int32_t num_values = 0, num_nulls = -1, num_rows = 0;
  int encoding = -1, def_encoding = ENC_RLE;
  int32_t def_bytes = 0, rep_bytes = 0;  // v2
  bool v2_compressed = true;
};

// parquet.thrift: PageHeader {1 type, 2 uncompressed_page_size, 3 compressed_page_size, 4 crc, 5 data_page_header,
// 6 index_page_header, 7 dictionary_page_header, 8 data_page_header_v2}
PageHeader parse_page_header(Thrift& t) {
  PageHeader h;
  int ft, id = 0;
  while (t.field(ft, id)) {
    if (id == 1 && ft == 5) h.type = (int)t.zigzag();
    else if (id == 2 && ft == 5) h.uncompressed = (int32_t)t.zigzag();
    else if (id == 3 && ft == 5) h.compressed = (int32_t)t.zigzag();
    else if ((id == 5 || id == 7 || id == 8) && ft == 12) {
      int st, sid = 0;
      while (t.field(st, sid)) {
        if (id == 5) {         // DataPageHeader {1 num_values, 2 encoding, 3 definition_level_encoding, 4 repetition_level_encoding, 5 statistics}
          if (sid == 1 && st == 5) h.num_values = (int32_t)t.zigzag();
          else if (sid == 2 && st == 5) h.encoding = (int)t.zigzag();
          else if (sid == 3 && st == 5) h.def_encoding = (int)t.zigzag();
          else t.skip(st);
        } else if (id == 7) {  // DictionaryPageHeader {1 num_values, 2 encoding, 3 is_sorted}
          if (sid == 1 && st == 5) h.num_values = (int32_t)t.zigzag();
          else if (sid == 2 && st == 5) h.encoding = (int)t.zigzag();
          else t.skip(st);
        } else {               // DataPageHeaderV2 {1 num_values, 2 num_nulls, 3 num_rows, 4 encoding, 5 definition_levels_byte_length,
                               //                   6 repetition_levels_byte_length, 7 is_compressed, 8 statistics}
          if (sid == 1 && st == 5) h.num_values = (int32_t)t.zigzag();
          else if (sid == 2 && st == 5) h.num_nulls = (int32_t)t.zigzag();
          else if (sid == 3 && st == 5) h.num_rows = (int32_t)t.zigzag();
          else if (sid == 4 && st == 5) h.encoding = (int)t.zigzag();
          else if (sid == 5 && st == 5) h.def_bytes = (int32_t)t.zigzag();
          else if (sid == 6 && st == 5) h.rep_bytes = (int32_t)t.zigzag();
          else if (sid == 7 && (st == 1 || st == 2)) h.v2_compressed = st == 1;
          else t.skip(st);
        }
      }
    } else {
      t.skip(ft);
    }
  }
  DFGPU_CHECK(h.type >= 0 && h.compressed >= 0 && h.uncompressed >= 0, "parquet: malformed page header");
  return h;
}

// ------------------------------------------------------------------------------------------------- decompression
// Snappy raw format (format_description.txt of google/snappy): varint length, then literal / copy elements
void snappy_decompress(const uint8_t* src, size_t n, uint8_t* dst, size_t dst_len) {
  Thrift v{src, src + n};
  uint64_t len = v.varint();
  DFGPU_CHECK(len == dst_len, "parquet: snappy length does not match the page header");
  const uint8_t* p = v.p;
  const uint8_t* end = src + n;
  size_t o = 0;
  while (p < end) {
    const uint8_t tag = *p++;
    if ((tag & 3) == 0) {
      size_t l = (tag >> 2) + 1;
      if (l > 60) {
        const int nb = (int)l - 60;
        DFGPU_CHECK(end - p >= nb, "parquet: truncated snappy literal");
        l = 0;
        for (int i = 0; i < nb; i++) l |= (size_t)p[i] << (8 * i);
        l += 1;
        p += nb;
      }
      DFGPU_CHECK((size_t)(end - p) >= l && o + l <= dst_len, "parquet: snappy literal overruns");
      std::memcpy(dst + o, p, l);
      p += l;
      o += l;
    } else {
      size_t l, off;
      if ((tag & 3) == 1) {
        DFGPU_CHECK(end - p >= 1, "parquet: truncated snappy copy");
        l = 4 + ((tag >> 2) & 7);
        off = ((size_t)(tag >> 5) << 8) | *p++;
      } else if ((tag & 3) == 2) {
        DFGPU_CHECK(end - p >= 2, "parquet: truncated snappy copy");
        l = (tag >> 2) + 1;
        off = (size_t)p[0] | ((size_t)p[1] << 8);
        p += 2;
      } else {
        DFGPU_CHECK(end - p >= 4, "parquet: truncated snappy copy");
        l = (tag >> 2) + 1;
        off = (size_t)p[0] | ((size_t)p[1] << 8) | ((size_t)p[2] << 16) | ((size_t)p[3] << 24);
        p += 4;
      }
      DFGPU_CHECK(off != 0 && off <= o && o + l <= dst_len, "parquet: snappy copy out of range");
      for (size_t i = 0; i < l; i++) dst[o + i] = dst[o + i - off];  // may overlap: byte by byte
      o += l;
    }
  }
  DFGPU_CHECK(o == dst_len, "parquet: snappy output shorter than the page header says");
}

using zstd_decompress_fn = size_t (*)(void*, size_t, const void*, size_t);
using zstd_iserror_fn = unsigned (*)(size_t);
void zstd_decompress(const uint8_t* src, size_t n, uint8_t* dst, size_t dst_len) {
  static zstd_decompress_fn dec = nullptr;
  static zstd_iserror_fn iserr = nullptr;
  static std::once_flag once;
  std::call_once(once, [] {
    void* h = dlopen("libzstd.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("libzstd.so", RTLD_NOW | RTLD_GLOBAL);
    if (h) {
      dec = (zstd_decompress_fn)dlsym(h, "ZSTD_decompress");
      iserr = (zstd_iserror_fn)dlsym(h, "ZSTD_isError");
    }
  });
  DFGPU_CHECK(dec && iserr, "parquet: ZSTD pages need libzstd.so.1 on this host");
  size_t r = dec(dst, dst_len, src, n);
  DFGPU_CHECK(!iserr(r) && r == dst_len, "parquet: ZSTD page did not decompress to the size in its header");
}

void decompress(int codec, const uint8_t* src, size_t n, uint8_t* dst, size_t dst_len) {
  switch (codec) {
    case DFGPU_PARQUET_UNCOMPRESSED:
      DFGPU_CHECK(n == dst_len, "parquet: uncompressed page with different sizes");
      std::memcpy(dst, src, n);
      return;
    case DFGPU_PARQUET_SNAPPY: snappy_decompress(src, n, dst, dst_len); return;
    case DFGPU_PARQUET_ZSTD: zstd_decompress(src, n, dst, dst_len); return;
  }
  throw Error("parquet: compression codec " + std::to_string(codec) + " is not supported on the GPU scan path");
}

// --------------------------------------------------------------------------------------- RLE / bit-packed hybrid runs
// Encodings.md "Run Length Encoding / Bit-Packing Hybrid": header varint h; h & 1: (h >> 1) groups of 8 bit-packed values,
// else an RLE run of (h >> 1) copies of one value stored in ceil(bit_width / 8) bytes.
struct Run {            // mirrored on the device
  int64_t start;        // first value position (chunk-wide, in the space of non-null values)
  int64_t payload;      // RLE: the repeated value; bit-packed: byte offset of the packed bits in the staging buffer
  int32_t count;
  int32_t bit_width;    // bit 31 set = bit-packed
};
constexpr int32_t RUN_PACKED = (int32_t)0x80000000;

// walks the runs of one hybrid stream holding `n` values; fn(is_packed, count, rle_value, packed_bytes_ptr)
template <typename F>
void walk_runs(const uint8_t* p, const uint8_t* end, int bit_width, int64_t n, F&& fn) {
  DFGPU_CHECK(bit_width >= 0 && bit_width <= 32, "parquet: bad bit width");
  const int vbytes = (bit_width + 7) / 8;
  int64_t done = 0;
  while (done < n) {
    Thrift v{p, end};
    uint64_t h = v.varint();
    p = v.p;
    if (h & 1) {
      // the varint is untrusted: bound the group count by what the page can hold BEFORE multiplying (groups * bit_width
      // would wrap for groups >= 2^58 and pass the overrun check with bytes the page does not have)
      const uint64_t ugroups = h >> 1;
      DFGPU_CHECK(ugroups > 0 && ugroups <= (uint64_t)INT32_MAX && ugroups <= (uint64_t)(end - p) / (uint64_t)std::max(bit_width, 1),
                  "parquet: bit-packed run overruns its page");
      const int64_t groups = (int64_t)ugroups;
      const int64_t bytes = groups * bit_width;
      DFGPU_CHECK(end - p >= bytes, "parquet: bit-packed run overruns its page");
      const int64_t cnt = std::min<int64_t>(groups * 8, n - done);   // the last group may be padding
      fn(true, cnt, 0ull, p);
      p += bytes;
      done += cnt;
    } else {
      DFGPU_CHECK((h >> 1) > 0 && (h >> 1) <= (uint64_t)INT64_MAX && end - p >= vbytes, "parquet: RLE run overruns its page");
      const int64_t cnt = (int64_t)(h >> 1);
      uint64_t val = 0;
      for (int i = 0; i < vbytes; i++) val |= (uint64_t)p[i] << (8 * i);
      p += vbytes;
      fn(false, std::min<int64_t>(cnt, n - done), val, nullptr);
      done += std::min<int64_t>(cnt, n - done);
    }
  }
}

// DELTA_BINARY_PACKED (Encodings.md "Delta Encoding"): <block size> <miniblocks per block> <total count> <first value (zigzag)>, then
// per block <min delta (zigzag)> <one bit width per miniblock> <miniblocks: deltas - min delta, bit-packed LSB first>.  A value is
// the running sum of the deltas, wrapping in the PHYSICAL type's width.  Decoded on the host into PLAIN little-endian values of
// `pw` bytes (the value of a page depends on all before it: a scan per page; files that carry it are rare — pyarrow writes it only
// on request) and staged like a PLAIN page.
void decode_delta_binary_packed(const uint8_t* p, const uint8_t* end, int pw, int64_t n, uint8_t* out) {
  Thrift t{p, end};
  const uint64_t block = t.varint(), minis = t.varint(), total = t.varint();
  DFGPU_CHECK(block > 0 && block % 128 == 0 && minis > 0 && block % minis == 0 && (block / minis) % 32 == 0, "parquet: malformed DELTA_BINARY_PACKED header");
  DFGPU_CHECK((int64_t)total >= n, "parquet: DELTA_BINARY_PACKED page holds fewer values than its header says");
  const uint64_t per_mini = block / minis;
  uint64_t cur = (uint64_t)t.zigzag();
  auto put = [&](int64_t i, uint64_t v) { std::memcpy(out + (size_t)i * pw, &v, (size_t)pw); };  // little endian: the low pw bytes (wraps)
  int64_t done = 0;
  if (n > 0) put(done++, cur);
  while (done < n) {
    const uint64_t min_delta = (uint64_t)t.zigzag();
    DFGPU_CHECK((uint64_t)(t.end - t.p) >= minis, "parquet: DELTA_BINARY_PACKED block overruns its page");
    const uint8_t* widths = t.p;
    t.p += minis;
    for (uint64_t m = 0; m < minis && done < n; m++) {
      const int bw = widths[m];
      DFGPU_CHECK(bw <= 64, "parquet: DELTA_BINARY_PACKED bit width");
      const uint64_t bytes = per_mini * (uint64_t)bw / 8;
      DFGPU_CHECK((uint64_t)(t.end - t.p) >= bytes, "parquet: DELTA_BINARY_PACKED miniblock overruns its page");
      for (uint64_t i = 0; i < per_mini && done < n; i++) {
        uint64_t d = 0;
        const uint64_t bit = i * (uint64_t)bw;
        for (int b = 0; b < bw; b++) d |= (uint64_t)((t.p[(bit + b) >> 3] >> ((bit + b) & 7)) & 1) << b;
        cur += min_delta + d;
        put(done++, cur);
      }
      t.p += bytes;
    }
  }
}
// BYTE_STREAM_SPLIT: byte k of value i at stream k * n + i
void decode_byte_stream_split(const uint8_t* p, const uint8_t* end, int pw, int64_t n, uint8_t* out) {
  DFGPU_CHECK(end - p >= n * pw, "parquet: BYTE_STREAM_SPLIT values overrun the page");
  for (int k = 0; k < pw; k++)
    for (int64_t i = 0; i < n; i++) out[(size_t)i * pw + k] = p[(size_t)k * n + i];
}

// a page of the chunk as the device sees it
struct PageDesc {
  int64_t value_start;   // first non-null value of the page, chunk-wide
  int64_t byte_offset;   // PLAIN: offset of the values in the staging buffer
  int32_t first_run;     // dictionary-encoded: [first_run, next page's first_run) in the run table
  int32_t kind;          // 0 PLAIN, 1 dictionary indices
};

struct ChunkPlan {       // everything the host learns from one chunk
  StageVec<PageDesc> pages;            // (everything uploaded lives in pinned staging: internal.hpp PinnedBuf)
  StageVec<Run> runs;
  StageVec<uint8_t> staging;           // value bytes of every page (PLAIN values, packed index bits)
  StageVec<uint64_t> validity;         // one bit per row; empty = no nulls
  std::vector<uint8_t> dict_page;      // PLAIN-encoded dictionary values (uncompressed)
  int32_t dict_count = 0;
  int64_t rows = 0, values = 0;        // rows incl. NULLs; non-null values
  dfgpu_parquet_chunk_info info{};
  // BYTE_ARRAY read as Utf8 (field.type DFGPU_UTF8): every non-null value as (offset, length) into str_bytes — a PLAIN page's
  // body goes there as it is (the offsets skip its 4-byte length prefixes), the dictionary's strings once
  StageVec<int64_t> str_src;           // (pinned staging: uploaded once and dropped, see internal.hpp PinnedBuf)
  StageVec<uint32_t> str_len;
  StageVec<uint8_t> str_bytes;
  std::vector<int64_t> dict_src;       // the dictionary's strings in str_bytes
  std::vector<uint32_t> dict_len;
  std::vector<uint8_t> bool_values;    // BOOLEAN: one byte per non-null value
};

void set_bits(StageVec<uint64_t>& bm, int64_t pos, int64_t n) {
  for (int64_t i = pos; i < pos + n;) {
    const int64_t w = i >> 6, b = i & 63;
    const int64_t take = std::min<int64_t>(64 - b, pos + n - i);
    const uint64_t mask = (take == 64 ? ~0ull : ((1ull << take) - 1)) << b;
    bm[(size_t)w] |= mask;
    i += take;
  }
}
// copy n bits from an LSB-first packed byte stream to bit position pos
void copy_bits(StageVec<uint64_t>& bm, int64_t pos, const uint8_t* src, int64_t n) {
  for (int64_t i = 0; i < n; i++)
    if ((src[i >> 3] >> (i & 7)) & 1) bm[(size_t)((pos + i) >> 6)] |= 1ull << ((pos + i) & 63);
}

int phys_width(const dfgpu_parquet_column& c) {
  switch (c.physical_type) {
    case DFGPU_PARQUET_INT32: return 4;
    case DFGPU_PARQUET_INT64: case DFGPU_PARQUET_DOUBLE: return 8;
    case DFGPU_PARQUET_FIXED_LEN_BYTE_ARRAY:
      DFGPU_CHECK(c.type_length >= 1 && c.type_length <= 16, "parquet: FIXED_LEN_BYTE_ARRAY longer than 16 bytes");
      return c.type_length;
    case DFGPU_PARQUET_BYTE_ARRAY: return 0;
    case DFGPU_PARQUET_BOOLEAN: return 0;   // bit-packed
  }
  throw Error("parquet: physical type " + std::to_string(c.physical_type) + " is not supported on the GPU scan path");
}

void check_target(const dfgpu_parquet_column& c) {
  const int t = c.field.type;
  bool ok = false;
  switch (c.physical_type) {
    case DFGPU_PARQUET_INT32: ok = t == DFGPU_INT32 || t == DFGPU_DATE32 || t == DFGPU_UINT8 || t == DFGPU_UINT32 || t == DFGPU_DECIMAL128; break;
    case DFGPU_PARQUET_INT64: ok = t == DFGPU_INT64 || t == DFGPU_UINT64 || t == DFGPU_DECIMAL128; break;
    case DFGPU_PARQUET_DOUBLE: ok = t == DFGPU_FLOAT64; break;
    case DFGPU_PARQUET_BOOLEAN: ok = t == DFGPU_BOOL; break;
    case DFGPU_PARQUET_FIXED_LEN_BYTE_ARRAY: ok = t == DFGPU_DECIMAL128; break;
    case DFGPU_PARQUET_BYTE_ARRAY: ok = t == DFGPU_INT32 || t == DFGPU_UTF8; break;   // dictionary indices of a string column, or the strings
  }
  DFGPU_CHECK(ok, "parquet: physical type " + std::to_string(c.physical_type) + " cannot be read as " + type_name(c.field));
  DFGPU_CHECK(c.max_repetition_level == 0, "parquet: repeated (nested) columns are not supported on the GPU scan path");
  DFGPU_CHECK(c.max_definition_level == 0 || c.max_definition_level == 1, "parquet: nested optional columns are not supported on the GPU scan path");
}

// parse + decompress every page of the chunk; no device work
ChunkPlan plan_chunk(const uint8_t* chunk, int64_t nbytes, const dfgpu_parquet_column& col) {
  check_target(col);
  const int pw = phys_width(col);
  ChunkPlan P;
  const bool nullable = col.max_definition_level == 1;
  if (nullable) P.validity.assign((size_t)((col.num_values + 63) / 64) + 1, 0ull);
  const uint8_t* p = chunk;
  const uint8_t* end = chunk + nbytes;
  std::vector<uint8_t> body;
  {
    // the staging buffer is sized once, from a walk over the page headers alone: growing it page by page re-copies what is
    // already there (a 38 MB chunk: three times the memory traffic of the copy itself)
    size_t need = 0;
    int64_t rows = 0;
    const uint8_t* q = chunk;
    while (rows < col.num_values && q < end) {
      Thrift t{q, end};
      PageHeader h = parse_page_header(t);
      if (h.compressed < 0 || h.uncompressed < 0 || end - t.p < h.compressed) break;   // (reported by the walk below)
      q = t.p + h.compressed;
      if (h.type == PAGE_DATA || h.type == PAGE_DATA_V2) {
        need += (size_t)h.uncompressed + 32;
        rows += h.num_values > 0 ? h.num_values : 0;
      }
    }
    if (col.field.type != DFGPU_UTF8 && col.physical_type != DFGPU_PARQUET_BOOLEAN) P.staging.reserve(need);
  }
  while (P.rows < col.num_values) {
    DFGPU_CHECK(p < end, "parquet: the chunk ends before its num_values rows");
    Thrift t{p, end};
    PageHeader h = parse_page_header(t);
    p = t.p;
    DFGPU_CHECK(end - p >= h.compressed, "parquet: page body overruns the chunk");
    const uint8_t* raw = p;
    p += h.compressed;
    P.info.n_pages++;
    P.info.compressed_bytes += h.compressed;
    P.info.uncompressed_bytes += h.uncompressed;
    if (h.type == PAGE_INDEX) continue;
    if (h.type == PAGE_DICTIONARY) {
      DFGPU_CHECK(P.dict_page.empty() && P.pages.empty(), "parquet: more than one dictionary page, or a dictionary page after data pages");
      DFGPU_CHECK(h.encoding == ENC_PLAIN || h.encoding == ENC_PLAIN_DICTIONARY, "parquet: dictionary page encoding " + std::to_string(h.encoding));
      P.dict_page.resize((size_t)h.uncompressed + 16);
      decompress(col.codec, raw, (size_t)h.compressed, P.dict_page.data(), (size_t)h.uncompressed);
      P.dict_page.resize((size_t)h.uncompressed);
      P.dict_count = h.num_values;
      P.info.n_dictionary_pages++;
      continue;
    }
    DFGPU_CHECK(h.type == PAGE_DATA || h.type == PAGE_DATA_V2, "parquet: unknown page type " + std::to_string(h.type));
    // a corrupt header must not walk the validity bitmap or the staging buffer out of bounds
    DFGPU_CHECK(h.num_values >= 0 && P.rows + h.num_values <= col.num_values, "parquet: pages hold more rows than the chunk's num_values");
    DFGPU_CHECK(h.def_bytes >= 0 && h.rep_bytes >= 0, "parquet: negative level byte length");
    // ---- uncompressed page body: v1 = [def levels][values] compressed together; v2 = levels uncompressed + values
    // Fixed-width values and dictionary indices are decoded on the device from the page body as it is, so the body is decompressed
    // straight into the staging buffer that is uploaded (pinned; the few level bytes in front of the values ride along); strings,
    // BOOLEAN and the host-decoded encodings go through a scratch body
    const bool direct = col.field.type != DFGPU_UTF8 && col.physical_type != DFGPU_PARQUET_BOOLEAN &&
                        (h.encoding == ENC_PLAIN || h.encoding == ENC_RLE_DICTIONARY || h.encoding == ENC_PLAIN_DICTIONARY);
    uint8_t* body_at;
    if (direct) {
      const size_t body_base = (P.staging.size() + 15) & ~size_t(15);
      P.staging.resize(body_base + (size_t)h.uncompressed + 16);   // +16: the unpacker reads whole 64-bit windows
      body_at = P.staging.data() + body_base;
    } else {
      if (body.size() < (size_t)h.uncompressed + 16) body.resize((size_t)h.uncompressed + 16);
      body_at = body.data();
    }
    const uint8_t* levels = nullptr;
    int64_t level_bytes = 0;
    const uint8_t* values;
    const uint8_t* values_end;
    if (h.type == PAGE_DATA) {
      decompress(col.codec, raw, (size_t)h.compressed, body_at, (size_t)h.uncompressed);
      const uint8_t* b = body_at;
      if (nullable) {
        DFGPU_CHECK(h.def_encoding == ENC_RLE, "parquet: definition levels with the deprecated BIT_PACKED encoding");
        DFGPU_CHECK(h.uncompressed >= 4, "parquet: data page too short for its level length");
        uint32_t lb;
        std::memcpy(&lb, b, 4);
        DFGPU_CHECK((int64_t)lb + 4 <= h.uncompressed, "parquet: definition levels overrun the page");
        levels = b + 4;
        level_bytes = lb;
        b += 4 + lb;
      }
      values = b;
      values_end = body_at + h.uncompressed;
      P.info.n_data_pages_v1++;
    } else {
      DFGPU_CHECK(h.rep_bytes == 0, "parquet: repetition levels in a flat column");
      DFGPU_CHECK((int64_t)h.def_bytes <= h.compressed && (int64_t)h.def_bytes <= h.uncompressed, "parquet: v2 level bytes overrun the page");
      levels = raw;
      level_bytes = h.def_bytes;
      const size_t vu = (size_t)(h.uncompressed - h.def_bytes), vc = (size_t)(h.compressed - h.def_bytes);
      decompress(h.v2_compressed ? col.codec : DFGPU_PARQUET_UNCOMPRESSED, raw + h.def_bytes, vc, body_at, vu);
      values = body_at;
      values_end = body_at + vu;
      P.info.n_data_pages_v2++;
    }
    // ---- definition levels -> validity bits, non-null count of the page
    int64_t nonnull = h.num_values;
    if (nullable) {
      nonnull = 0;
      int64_t row = P.rows;
      walk_runs(levels, levels + level_bytes, 1, h.num_values, [&](bool packed, int64_t cnt, uint64_t val, const uint8_t* bits) {
        if (packed) {
          copy_bits(P.validity, row, bits, cnt);
          for (int64_t i = 0; i < cnt; i++) nonnull += (bits[i >> 3] >> (i & 7)) & 1;
        } else if (val) {
          set_bits(P.validity, row, cnt);
          nonnull += cnt;
        }
        row += cnt;
      });
    } else if (level_bytes) {
      // a required column written with (empty) level data: nothing to read
    }
    // ---- values
    PageDesc d{};
    d.value_start = P.values;
    d.first_run = (int32_t)P.runs.size();
    const size_t base = (P.staging.size() + 15) & ~size_t(15);
    if (col.field.type == DFGPU_UTF8) {
      // ---- strings: every value's (offset, length); the bytes move once, on the device
      if (h.encoding == ENC_PLAIN) {
        const int64_t at = (int64_t)P.str_bytes.size();
        const uint8_t* q = values;
        for (int64_t j = 0; j < nonnull; j++) {
          DFGPU_CHECK(values_end - q >= 4, "parquet: PLAIN string values overrun the page");
          uint32_t len;
          std::memcpy(&len, q, 4);
          q += 4;
          DFGPU_CHECK((uint64_t)(values_end - q) >= len, "parquet: PLAIN string values overrun the page");
          P.str_src.push_back(at + (q - values));
          P.str_len.push_back(len);
          q += len;
        }
        P.str_bytes.insert(P.str_bytes.end(), values, q);
        P.info.n_plain_pages++;
      } else if (h.encoding == ENC_RLE_DICTIONARY || h.encoding == ENC_PLAIN_DICTIONARY) {
        DFGPU_CHECK(!P.dict_page.empty() || P.dict_count == 0, "parquet: dictionary-encoded page without a dictionary page");
        if (P.dict_src.empty() && P.dict_count) {  // the dictionary's strings, once
          const uint8_t* q = P.dict_page.data();
          const uint8_t* qe = q + P.dict_page.size();
          const int64_t at = (int64_t)P.str_bytes.size();
          for (int32_t i = 0; i < P.dict_count; i++) {
            DFGPU_CHECK(qe - q >= 4, "parquet: truncated string dictionary");
            uint32_t len;
            std::memcpy(&len, q, 4);
            q += 4;
            DFGPU_CHECK((uint64_t)(qe - q) >= len, "parquet: truncated string dictionary");
            P.dict_src.push_back(at + (q - P.dict_page.data()));
            P.dict_len.push_back(len);
            q += len;
          }
          P.str_bytes.insert(P.str_bytes.end(), P.dict_page.begin(), P.dict_page.end());
        }
        DFGPU_CHECK(values_end > values || nonnull == 0, "parquet: dictionary-encoded page without a bit width");
        if (nonnull) {
          const int bw = values[0];
          DFGPU_CHECK(bw <= 32, "parquet: dictionary index bit width " + std::to_string(bw));
          auto put = [&](uint64_t idx) {
            DFGPU_CHECK(idx < (uint64_t)P.dict_count, "parquet: dictionary index out of range");
            P.str_src.push_back(P.dict_src[(size_t)idx]);
            P.str_len.push_back(P.dict_len[(size_t)idx]);
          };
          walk_runs(values + 1, values_end, bw, nonnull, [&](bool packed, int64_t cnt, uint64_t val, const uint8_t* bits) {
            if (!packed) {
              for (int64_t i = 0; i < cnt; i++) put(val);
              (void)P.info.n_runs_rle++;
              return;
            }
            P.info.n_runs_bitpacked++;
            for (int64_t i = 0; i < cnt; i++) {  // bit-packed, LSB first
              const int64_t bit = i * bw;
              uint64_t v = 0;
              for (int b = 0; b < bw; b++) v |= (uint64_t)((bits[(bit + b) >> 3] >> ((bit + b) & 7)) & 1) << b;
              put(v);
            }
          });
        }
        P.info.n_dictionary_encoded_pages++;
      } else {
        throw Error("parquet: value encoding " + std::to_string(h.encoding) + " is not supported on the GPU scan path");
      }
      d.kind = 2;
      P.pages.push_back(d);
      P.rows += h.num_values;
      P.values += nonnull;
      continue;
    }
    if (col.physical_type == DFGPU_PARQUET_BOOLEAN) {
      // BOOLEAN: PLAIN = one bit per value, LSB first; RLE (data page v2, and v1 of newer writers) = <4-byte length> hybrid runs of
      // bit width 1.  Kept as one byte per non-null value here; the column's bitmap is assembled once all pages are in.
      const size_t at = P.bool_values.size();
      P.bool_values.resize(at + (size_t)nonnull);
      if (h.encoding == ENC_PLAIN) {
        DFGPU_CHECK((values_end - values) * 8 >= nonnull, "parquet: PLAIN BOOLEAN values overrun the page");
        for (int64_t i = 0; i < nonnull; i++) P.bool_values[at + (size_t)i] = (values[i >> 3] >> (i & 7)) & 1;
        P.info.n_plain_pages++;
      } else if (h.encoding == ENC_RLE) {
        DFGPU_CHECK(values_end - values >= 4, "parquet: RLE BOOLEAN page without its length");
        uint32_t lb;
        std::memcpy(&lb, values, 4);
        DFGPU_CHECK((int64_t)lb <= values_end - values - 4, "parquet: RLE BOOLEAN values overrun the page");
        int64_t i = 0;
        walk_runs(values + 4, values + 4 + lb, 1, nonnull, [&](bool packed, int64_t cnt, uint64_t val, const uint8_t* bits) {
          for (int64_t k = 0; k < cnt; k++) P.bool_values[at + (size_t)(i + k)] = packed ? ((bits[k >> 3] >> (k & 7)) & 1) : (uint8_t)(val & 1);
          i += cnt;
          (packed ? P.info.n_runs_bitpacked : P.info.n_runs_rle)++;
        });
      } else {
        throw Error("parquet: BOOLEAN value encoding " + std::to_string(h.encoding) + " is not supported on the GPU scan path");
      }
      d.kind = 3;
      P.pages.push_back(d);
      P.rows += h.num_values;
      P.values += nonnull;
      continue;
    }
    if (h.encoding == ENC_DELTA_BINARY_PACKED || h.encoding == ENC_BYTE_STREAM_SPLIT) {
      DFGPU_CHECK(pw == 4 || pw == 8 || (h.encoding == ENC_BYTE_STREAM_SPLIT && pw > 0), "parquet: value encoding " + std::to_string(h.encoding) + " on this physical type");
      DFGPU_CHECK(h.encoding == ENC_BYTE_STREAM_SPLIT || col.physical_type == DFGPU_PARQUET_INT32 || col.physical_type == DFGPU_PARQUET_INT64,
                  "parquet: DELTA_BINARY_PACKED on a non-integer column");
      P.staging.resize(base + (size_t)(nonnull * pw));
      if (h.encoding == ENC_DELTA_BINARY_PACKED) decode_delta_binary_packed(values, values_end, pw, nonnull, P.staging.data() + base);
      else decode_byte_stream_split(values, values_end, pw, nonnull, P.staging.data() + base);
      d.kind = 0;   // staged as PLAIN
      d.byte_offset = (int64_t)base;
      P.info.n_plain_pages++;
      P.pages.push_back(d);
      P.rows += h.num_values;
      P.values += nonnull;
      continue;
    }
    if (h.encoding == ENC_PLAIN) {
      DFGPU_CHECK(pw > 0, "parquet: PLAIN-encoded BYTE_ARRAY pages (strings outside a dictionary): read the column as Utf8 (field.type DFGPU_UTF8) instead of dictionary indices");
      DFGPU_CHECK(values_end - values >= nonnull * pw, "parquet: PLAIN values overrun the page");
      d.kind = 0;
      d.byte_offset = (int64_t)(values - P.staging.data());   // (direct: the values lie in the staging buffer already)
      P.info.n_plain_pages++;
    } else if (h.encoding == ENC_RLE_DICTIONARY || h.encoding == ENC_PLAIN_DICTIONARY) {
      DFGPU_CHECK(!P.dict_page.empty() || P.dict_count == 0, "parquet: dictionary-encoded page without a dictionary page");
      DFGPU_CHECK(values_end > values || nonnull == 0, "parquet: dictionary-encoded page without a bit width");
      d.kind = 1;
      if (nonnull) {
        const int bw = values[0];
        DFGPU_CHECK(bw <= 32, "parquet: dictionary index bit width " + std::to_string(bw));
        const uint8_t* rp = values + 1;
        const uint8_t* staged = P.staging.data();   // (direct: the runs lie in the staging buffer already)
        int64_t at = P.values;
        walk_runs(rp, values_end, bw, nonnull, [&](bool packed, int64_t cnt, uint64_t val, const uint8_t* bits) {
          Run r{};
          r.start = at;
          r.count = (int32_t)cnt;
          r.bit_width = packed ? (bw | RUN_PACKED) : bw;
          r.payload = packed ? (int64_t)(bits - staged) : (int64_t)val;
          if (!packed) DFGPU_CHECK(val < (uint64_t)std::max(P.dict_count, 1), "parquet: dictionary index out of range");
          P.runs.push_back(r);
          (packed ? P.info.n_runs_bitpacked : P.info.n_runs_rle)++;
          at += cnt;
        });
      }
      P.info.n_dictionary_encoded_pages++;
    } else {
      throw Error("parquet: value encoding " + std::to_string(h.encoding) + " is not supported on the GPU scan path");
    }
    P.pages.push_back(d);
    P.rows += h.num_values;
    P.values += nonnull;
  }
  DFGPU_CHECK(P.rows == col.num_values, "parquet: pages hold more rows than the chunk's num_values");
  P.info.values = P.rows;
  P.info.nulls = P.rows - P.values;
  if (!nullable || P.values == P.rows) P.validity.clear();
  return P;
}

}  // namespace

// ----------------------------------------------------------------------------------------------------- device side
struct PqArgs {
  const PageDesc* pages;
  const Run* runs;
  const uint8_t* staging;
  const void* dict;     // dictionary values already in the TARGET representation (width out_w)
  int32_t n_pages, n_runs, dict_count;
  int32_t phys_w;       // bytes per PLAIN value
  int32_t big_endian;   // PLAIN FIXED_LEN_BYTE_ARRAY decimals: big-endian two's complement
  int32_t sign_extend;  // PLAIN INT32 / INT64 widened to a signed wider target
};

__device__ __forceinline__ uint32_t pq_unpack(const uint8_t* base, int64_t bit, int bw) {
  // unaligned 64-bit window starting at the value's byte; bw <= 32 and bit & 7 <= 7 => 39 bits suffice
  const uint8_t* q = base + (bit >> 3);
  uint64_t w = 0;
#pragma unroll
  for (int i = 0; i < 5; i++) w |= (uint64_t)q[i] << (8 * i);
  return (uint32_t)((w >> (bit & 7)) & ((bw == 32) ? 0xFFFFFFFFull : ((1ull << bw) - 1)));
}

// one thread per non-null value: page by binary search (pages are few), run by binary search inside the page
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_pq_decode(PqArgs a, int64_t n_values, T* __restrict__ out) {
  for (int64_t v = (int64_t)blockIdx.x * BLOCK + threadIdx.x; v < n_values; v += (int64_t)gridDim.x * BLOCK) {
    int lo = 0, hi = a.n_pages - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (a.pages[mid].value_start <= v) lo = mid; else hi = mid - 1;
    }
    const PageDesc pg = a.pages[lo];
    T val;
    if (pg.kind == 0) {
      const uint8_t* src = a.staging + pg.byte_offset + (v - pg.value_start) * a.phys_w;
      if (a.big_endian) {
        // n-byte big-endian two's complement -> i128
        i128 x = (int8_t)src[0];
        for (int i = 1; i < a.phys_w; i++) x = (x << 8) | src[i];
        val = (T)x;
      } else if (a.phys_w == 4) {
        uint32_t u;
        memcpy(&u, src, 4);
        val = a.sign_extend ? (T)(int32_t)u : (T)u;
      } else {
        uint64_t u;
        memcpy(&u, src, 8);
        val = a.sign_extend ? (T)(int64_t)u : (T)u;
      }
    } else {
      int rl = pg.first_run, rh = (lo + 1 < a.n_pages ? a.pages[lo + 1].first_run : a.n_runs) - 1;
      while (rl < rh) {
        const int mid = (rl + rh + 1) >> 1;
        if (a.runs[mid].start <= v) rl = mid; else rh = mid - 1;
      }
      const Run r = a.runs[rl];
      uint32_t idx;
      if (r.bit_width < 0) {
        const int bw = r.bit_width & 0xFF;
        idx = pq_unpack(a.staging + r.payload, (v - r.start) * bw, bw);
      } else {
        idx = (uint32_t)r.payload;
      }
      if (idx >= (uint32_t)a.dict_count) idx = 0;   // corrupt index: stay inside the dictionary (the host checked RLE runs)
      val = reinterpret_cast<const T*>(a.dict)[idx];
    }
    out[v] = val;
  }
}

// NULL expansion: out[row] = valid(row) ? dense[rank(row)] : 0, rank from the per-word popcount prefix
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_pq_expand(const T* __restrict__ dense, const uint64_t* __restrict__ valid, const uint64_t* __restrict__ prefix, int64_t n_rows,
                                                     T* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n_rows; i += (int64_t)gridDim.x * BLOCK) {
    const uint64_t w = valid[i >> 6];
    const int b = (int)(i & 63);
    T v{};
    if ((w >> b) & 1ull) v = dense[prefix[i >> 6] + __popcll(w & ((1ull << b) - 1ull))];
    out[i] = v;
  }
}

namespace {

// dictionary page (PLAIN) -> host array in the target representation
StageVec<uint8_t> convert_dictionary(const ChunkPlan& P, const dfgpu_parquet_column& col, int out_w) {
  const int pw = phys_width(col);
  StageVec<uint8_t> out((size_t)std::max(P.dict_count, 1) * out_w, 0);
  DFGPU_CHECK((int64_t)P.dict_page.size() >= (int64_t)P.dict_count * pw, "parquet: dictionary page shorter than its value count");
  for (int32_t i = 0; i < P.dict_count; i++) {
    const uint8_t* s = P.dict_page.data() + (size_t)i * pw;
    i128 x;
    if (col.physical_type == DFGPU_PARQUET_FIXED_LEN_BYTE_ARRAY) {
      x = (int8_t)s[0];
      for (int k = 1; k < pw; k++) x = (x << 8) | s[k];
    } else if (pw == 4) {
      int32_t v;
      std::memcpy(&v, s, 4);
      x = is_signed_type(col.field.type) ? (i128)v : (i128)(uint32_t)v;
    } else {
      int64_t v;
      std::memcpy(&v, s, 8);
      x = (col.field.type == DFGPU_UINT64 || col.field.type == DFGPU_FLOAT64) ? (i128)(uint64_t)v : (i128)v;
    }
    std::memcpy(out.data() + (size_t)i * out_w, &x, (size_t)out_w);   // little endian: the low out_w bytes
  }
  return out;
}

// BYTE_ARRAY dictionary page -> strings; returns rank[old index] in the ascending order of the strings
std::shared_ptr<DictValues> string_dictionary(const ChunkPlan& P, const std::string& name, std::vector<int32_t>& rank) {
  std::vector<std::string> vals;
  const uint8_t* p = P.dict_page.data();
  const uint8_t* end = p + P.dict_page.size();
  for (int32_t i = 0; i < P.dict_count; i++) {
    DFGPU_CHECK(end - p >= 4, "parquet: truncated string dictionary");
    uint32_t len;
    std::memcpy(&len, p, 4);
    p += 4;
    DFGPU_CHECK((uint64_t)(end - p) >= len, "parquet: truncated string dictionary");
    vals.emplace_back((const char*)p, (size_t)len);
    p += len;
  }
  std::vector<int32_t> order((size_t)P.dict_count);
  std::iota(order.begin(), order.end(), 0);
  std::sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return vals[(size_t)x] < vals[(size_t)y]; });
  auto dv = std::make_shared<DictValues>();
  dv->index_format = "i";
  dv->value_format = "u";
  rank.assign((size_t)P.dict_count, 0);
  for (int32_t k = 0; k < P.dict_count; k++) {
    if (k > 0) DFGPU_CHECK(vals[(size_t)order[(size_t)k - 1]] != vals[(size_t)order[(size_t)k]], "parquet: column '" + name + "': duplicate dictionary value");
    rank[(size_t)order[(size_t)k]] = k;
    dv->values.push_back(vals[(size_t)order[(size_t)k]]);
    dv->valid.push_back(1);
  }
  dv->sorted = true;
  return dv;
}

template <typename T>
void launch_decode(const PqArgs& a, int64_t n_values, void* out) {
  if (n_values) k_pq_decode<T><<<grid_for(n_values, BLOCK), BLOCK, 0, rt().stream>>>(a, n_values, (T*)out);
}
template <typename T>
void launch_expand(const void* dense, const uint64_t* valid, const uint64_t* prefix, int64_t n_rows, void* out) {
  if (n_rows) k_pq_expand<T><<<grid_for(n_rows, BLOCK), BLOCK, 0, rt().stream>>>((const T*)dense, valid, prefix, n_rows, (T*)out);
}

// out[new_off[i] ...) = src_bytes[src[i] ...) for every row (a NULL row has length 0)
__global__ __launch_bounds__(BLOCK) void k_pq_string_copy(const int64_t* __restrict__ src, const uint64_t* __restrict__ new_off, const uint8_t* __restrict__ src_bytes, int64_t n,
                                                          uint8_t* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    const int64_t len = (int64_t)(new_off[i + 1] - new_off[i]);
    const uint8_t* s = src_bytes + src[i];
    uint8_t* d = out + new_off[i];
    for (int64_t k = 0; k < len; k++) d[k] = s[k];
  }
}

// BYTE_ARRAY -> a Utf8 column: the host found every value's (offset, length); the device spreads them over the NULL rows, scans
// the lengths into Arrow offsets and copies the bytes
Column decode_string_chunk(ChunkPlan& P, const dfgpu_parquet_column& col) {
  hipStream_t st = rt().stream;
  Column like;
  like.field = col.field;
  like.field.nullable = col.max_definition_level ? 1 : 0;
  like.name = col.name ? col.name : "";
  Column c = alloc_string_column(like, P.rows);
  if (P.rows == 0) {
    DFGPU_HIP(hipMemsetAsync(c.offsets->ptr, 0, 8, st));
    c.data = make_buf(16);
    return c;
  }
  DFGPU_CHECK((int64_t)P.str_src.size() == P.values && (int64_t)P.str_len.size() == P.values, "parquet: string value count mismatch");
  BufPtr d_bytes = make_buf(P.str_bytes.size() + 16), d_src = make_buf((size_t)std::max<int64_t>(P.values, 1) * 8 + 16),
         d_len = make_buf((size_t)std::max<int64_t>(P.values, 1) * 4 + 16);
  if (!P.str_bytes.empty()) DFGPU_HIP(hipMemcpyAsync(d_bytes->ptr, P.str_bytes.data(), P.str_bytes.size(), hipMemcpyHostToDevice, st));
  if (P.values) {
    DFGPU_HIP(hipMemcpyAsync(d_src->ptr, P.str_src.data(), (size_t)P.values * 8, hipMemcpyHostToDevice, st));
    DFGPU_HIP(hipMemcpyAsync(d_len->ptr, P.str_len.data(), (size_t)P.values * 4, hipMemcpyHostToDevice, st));
  }
  BufPtr src_rows = d_src, len_rows = d_len;
  if (!P.validity.empty()) {
    const size_t bb = bitmap_bytes(P.rows);
    c.validity = make_buf(bb);
    c.null_count = P.rows - P.values;
    DFGPU_HIP(hipMemcpyAsync(c.validity->ptr, P.validity.data(), bb, hipMemcpyHostToDevice, st));
    BufPtr prefix = make_buf((size_t)((P.rows + 63) / 64 + 1) * 8);
    scan_mask_popcounts(c.validity->as<uint64_t>(), nullptr, P.rows, prefix->as<uint64_t>());
    src_rows = make_buf((size_t)P.rows * 8 + 16);
    len_rows = make_buf((size_t)P.rows * 4 + 16);
    launch_expand<uint64_t>(d_src->ptr, c.valid_words(), prefix->as<uint64_t>(), P.rows, src_rows->ptr);
    launch_expand<uint32_t>(d_len->ptr, c.valid_words(), prefix->as<uint64_t>(), P.rows, len_rows->ptr);
  }
  scan_u32(len_rows->as<uint32_t>(), P.rows, c.offsets->as<uint64_t>());
  const int64_t total = (int64_t)read_u64(c.offsets->as<uint64_t>() + P.rows);
  c.data = make_buf((size_t)total + 16);
  {
    ProfileScope ps("parquet_decode_strings", (int64_t)P.str_bytes.size() + total + P.rows * 20);
    k_pq_string_copy<<<grid_for(P.rows, BLOCK), BLOCK, 0, st>>>(src_rows->as<int64_t>(), c.offsets->as<uint64_t>(), d_bytes->as<uint8_t>(), P.rows, (uint8_t*)c.data->ptr);
    DFGPU_HIP(hipGetLastError());
  }
  DFGPU_HIP(hipStreamSynchronize(st));   // the host vectors are the copies' sources
  return c;
}

// BOOLEAN chunk -> bit-packed column: the dense values are spread over the NULL rows on the host (a bit per row: the whole column
// is rows / 8 bytes)
Column decode_bool_chunk(const ChunkPlan& P, const dfgpu_parquet_column& col) {
  hipStream_t st = rt().stream;
  dfgpu_field f = col.field;
  f.nullable = col.max_definition_level ? 1 : 0;
  Column c = alloc_column(f, col.name ? col.name : "", P.rows);
  DFGPU_CHECK((int64_t)P.bool_values.size() == P.values, "parquet: BOOLEAN value count mismatch");
  std::vector<uint64_t> bits((size_t)((P.rows + 63) / 64) + 1, 0ull);
  int64_t v = 0;
  for (int64_t r = 0; r < P.rows; r++) {
    if (!P.validity.empty() && !((P.validity[(size_t)r >> 6] >> (r & 63)) & 1)) continue;
    if (P.bool_values[(size_t)v++]) bits[(size_t)r >> 6] |= 1ull << (r & 63);
  }
  const size_t bb = bitmap_bytes(P.rows);
  if (bb) DFGPU_HIP(hipMemcpyAsync(c.data->ptr, bits.data(), bb, hipMemcpyHostToDevice, st));
  if (!P.validity.empty()) {
    c.validity = make_buf(bb);
    c.null_count = P.rows - P.values;
    DFGPU_HIP(hipMemcpyAsync(c.validity->ptr, P.validity.data(), bb, hipMemcpyHostToDevice, st));
  }
  DFGPU_HIP(hipStreamSynchronize(st));
  return c;
}

}  // namespace

// ====================================================================================== pages decoded ON THE DEVICE (round 5)
// The scan -> device work SURVEY §8f N2 names, below the same entry points: for fixed-width targets whose pages are UNCOMPRESSED or SNAPPY
// the host reads the page HEADERS only (a few dozen Thrift structs per chunk) and the chunk's bytes cross PCIe as they lie in the file —
// compressed.  Then, per chunk:
//   k_pq_decompress    one wave per page: Snappy (raw format) decoded through a 32 KB ring of the output in LDS — the element stream is
//                      parsed from a 512-byte register window of the compressed bytes (readlane with a uniform index: the parse runs
//                      on the scalar side), every literal / copy is executed by the whole wave (one byte per lane; an overlapping copy
//                      reads `k mod offset`), 8 KB stretches of the ring are flushed to HBM with 16-byte stores; copies reaching back
//                      beyond the ring re-read flushed output (agent-scope loads after a release fence).  Uncompressed pages are copied.
//   k_pq_levels        one wave per data page of a nullable column: the definition levels' RLE / bit-packed hybrid runs (bit width 1)
//                      are walked on the device — bit-packed groups ARE Arrow validity bits and are OR-ed into the chunk's bitmap at
//                      the page's row offset, RLE runs set whole words — and the page's non-null count is the popcount on the way.
//   k_pq_decode_pages  one workgroup per data page: PLAIN values widened / byte-swapped; dictionary pages walk their run headers in
//                      batches (one lane parses varint headers into LDS, all lanes unpack / look up / store the batch's values — no
//                      run table crosses PCIe); DELTA_BINARY_PACKED pages parse block headers in batches, unpack the miniblocks' deltas
//                      in parallel and prefix-sum them across the workgroup; BYTE_STREAM_SPLIT gathers a value's bytes from its streams.
//                      The dictionary is the PLAIN dictionary page where it was decompressed (L2-resident), converted per lookup.
// What stays on the host: ZSTD pages (the FSE / Huffman entropy stages), BOOLEAN and Utf8 targets — those chunks take plan_chunk above.
// Every loop below consumes input or produces output in every iteration and is bounded by the page's byte counts: a corrupt page
// raises its error flag and ends.
enum { PQE_NONE = 0, PQE_SNAPPY = 1, PQE_LEVELS = 2, PQE_VALUES = 3, PQE_DICT_INDEX = 4, PQE_DELTA = 5, PQE_RUNS = 6 };
struct DevPage {
  int64_t src_off;       // the page body in the uploaded chunk
  int64_t dst_off;       // its uncompressed form in the body buffer (16-byte aligned)
  int64_t row_start;     // data pages: first row of the page, chunk-wide
  int32_t comp_bytes, uncomp_bytes;   // as the page header states them (v2: levels included)
  int32_t num_values;    // rows of the page (NULLs included); dictionary page: entries
  int32_t encoding;
  int32_t type;          // PAGE_DATA / PAGE_DICTIONARY / PAGE_DATA_V2
  int32_t def_bytes;     // v2: definition level bytes in front of the values (never compressed)
  int32_t codec;         // of the compressed part: DFGPU_PARQUET_UNCOMPRESSED / _SNAPPY
  int32_t in_chunk;      // the body is not compressed: it is read where it lies in the uploaded chunk (no copy)
};
__device__ __forceinline__ const uint8_t* pq_page_body(const DevPage& pg, const uint8_t* chunk, const uint8_t* body) {
  return pg.in_chunk ? chunk + pg.src_off : body + pg.dst_off;
}
struct DevPageState {
  int64_t values_off;    // the page's values inside its uncompressed body
  int64_t values_end;
  int32_t nonnull;
  int32_t error;
};
__host__ __device__ inline int64_t pq_align16(int64_t x) { return (x + 15) & ~(int64_t)15; }

// ---------------------------------------------------------------------------------------------------------------- Snappy
constexpr int SN_WIN = 32768, SN_FLUSH = 8192, SN_PIECE = 4096;
// one wave; returns 0 or PQE_SNAPPY.  `src_readable`: bytes that may be read from the 4-byte aligned address at or below src (the
// uploaded chunk is padded), dst 16-byte aligned.  Pages are below 2 GiB (Parquet's page sizes are i32): all positions are 32-bit,
// and everything but the byte moves themselves is uniform — the wave's one instruction stream is what bounds this kernel (one
// instruction per four cycles), so the loop is kept short: a tag and its operand bytes are two readlanes off the register window.
__device__ int snappy_wave(const uint8_t* __restrict__ src, int n_in, int64_t src_readable, uint8_t* dst, int dst_len, uint8_t* s_ring) {
  const unsigned lane = lane_id();
  const int a0 = (int)((uintptr_t)src & 3);
  const uint8_t* s4 = src - a0;          // 4-byte aligned; stream position p lives at s4[p + a0] (q = p + a0 below)
  const int readable = (int)(src_readable < 0x7FFFFFF0 ? src_readable : 0x7FFFFFF0);
  int wq = 0;                            // the register window covers q in [wq, wq + 512): wa the first 256 bytes, wb the rest
  uint32_t wa, wb;
  auto load_word = [&](int q) -> uint32_t { return q + 4 <= readable ? *reinterpret_cast<const uint32_t*>(s4 + q) : 0u; };
  wa = load_word(4 * (int)lane);
  wb = load_word(256 + 4 * (int)lane);
  // 8 stream bytes from q on (q uniform, wq <= q < wq + 256): tag + operands never span more than two dwords
  auto win = [&](int q) -> uint64_t {
    const int rel = q - wq, w = rel >> 2;
    const uint32_t lo_src = w < 64 ? wa : wb, hi_src = w + 1 < 64 ? wa : wb;
    const uint32_t d0 = __builtin_amdgcn_readlane(lo_src, w & 63), d1 = __builtin_amdgcn_readlane(hi_src, (w + 1) & 63);
    return (((uint64_t)d1 << 32) | d0) >> ((rel & 3) * 8);
  };
  int ip = 0, op = 0, flushed = 0;
  bool fence_due = false;   // flushed bytes a far copy may want to read back
  {   // preamble: the uncompressed length
    uint32_t len = 0;
    int shift = 0;
    for (;;) {
      if (ip >= n_in || shift > 28) return PQE_SNAPPY;
      const uint32_t b = (uint32_t)(win(ip + a0) & 0xFF);
      ip++;
      len |= (b & 0x7F) << shift;
      if (!(b & 0x80)) break;
      shift += 7;
    }
    if ((int)len != dst_len) return PQE_SNAPPY;
  }
  auto flush = [&]() {
    while (op - flushed >= SN_FLUSH) {
      const int r0 = flushed & (SN_WIN - 1);
      for (int j = (int)lane * 16; j < SN_FLUSH; j += 64 * 16) {
        const uint4 v = *reinterpret_cast<const uint4*>(s_ring + r0 + j);
        *reinterpret_cast<uint4*>(dst + flushed + j) = v;
      }
      flushed += SN_FLUSH;
      fence_due = true;
    }
  };
  while (ip < n_in) {
    int q = ip + a0;
    if (q - wq >= 256) {   // slide (or, behind a long literal, reload) the window
      if (q - wq < 512) {
        wa = wb;
        wq += 256;
        wb = load_word(wq + 256 + 4 * (int)lane);
      } else {
        wq = q & ~255;
        wa = load_word(wq + 4 * (int)lane);
        wb = load_word(wq + 256 + 4 * (int)lane);
      }
    }
    const uint64_t w8 = win(q);
    const uint32_t tag = (uint32_t)w8 & 0xFF;
    if ((tag & 3) == 0) {
      // ---- literal
      int l = (int)(tag >> 2) + 1, hdr = 1;
      if (l > 60) {
        const int nb = l - 60;
        l = (int)(uint32_t)((w8 >> 8) & (nb == 4 ? 0xFFFFFFFFull : ((1ull << (8 * nb)) - 1))) + 1;
        hdr = 1 + nb;
        if (l <= 0) return PQE_SNAPPY;
      }
      if (ip + hdr > n_in || l > n_in - ip - hdr || l > dst_len - op) return PQE_SNAPPY;
      if (l <= 64) {   // out of the register window: lane k takes byte k
        const int rel = q + hdr - wq + (int)lane, wi = rel >> 2;
        const uint32_t va = __shfl(wa, wi & 63), vb = __shfl(wb, wi & 63);
        const uint32_t v = wi < 64 ? va : vb;
        if ((int)lane < l) s_ring[(op + (int)lane) & (SN_WIN - 1)] = (uint8_t)(v >> ((rel & 3) * 8));
        op += l;
      } else {
        const uint8_t* lit = src + ip + hdr;
        for (int done = 0; done < l;) {
          const int piece = l - done < SN_PIECE ? l - done : SN_PIECE;
          // (4 KB per round trip: 16 aligned dwords per lane in flight, their bytes to the ring one by one — the ring position is
          //  not aligned to the stream's)
          const uint8_t* from = lit + done;
          const int mis = (int)((uintptr_t)from & 3);
          const uint32_t* words = reinterpret_cast<const uint32_t*>(from - mis);
          uint32_t d[16];
#pragma unroll
          for (int u = 0; u < 16; u++) {
            const int wi = (int)lane + 64 * u;
            d[u] = wi * 4 < piece + mis ? words[wi] : 0u;
          }
#pragma unroll
          for (int u = 0; u < 16; u++) {
            const int b0 = ((int)lane + 64 * u) * 4 - mis;   // piece-relative position of the dword's first byte
#pragma unroll
            for (int qq = 0; qq < 4; qq++)
              if (b0 + qq >= 0 && b0 + qq < piece) s_ring[(op + b0 + qq) & (SN_WIN - 1)] = (uint8_t)(d[u] >> (8 * qq));
          }
          if (piece + mis > SN_PIECE) {   // (the misaligned head pushed the tail beyond the 1024 dwords)
            for (int j = SN_PIECE - mis + (int)lane; j < piece; j += 64) s_ring[(op + j) & (SN_WIN - 1)] = from[j];
          }
          op += piece;
          done += piece;
          flush();
        }
      }
      ip += hdr + l;
    } else {
      // ---- copy
      int l, off, hdr;
      if ((tag & 3) == 1) {
        l = 4 + (int)((tag >> 2) & 7);
        off = (int)(((tag >> 5) << 8) | ((uint32_t)(w8 >> 8) & 0xFF));
        hdr = 2;
      } else if ((tag & 3) == 2) {
        l = (int)(tag >> 2) + 1;
        off = (int)((uint32_t)(w8 >> 8) & 0xFFFF);
        hdr = 3;
      } else {
        l = (int)(tag >> 2) + 1;
        const uint32_t o32 = (uint32_t)(w8 >> 8);
        if (o32 > 0x7FFFFFFFu) return PQE_SNAPPY;
        off = (int)o32;
        hdr = 5;
      }
      if (ip + hdr > n_in || off == 0 || off > op || l > dst_len - op) return PQE_SNAPPY;
      int k = (int)lane;
      if (off < l) {   // the pattern repeats: byte k comes from k mod off (exact for k, off <= 64: (k + 0.5) / off is never close to an integer)
        const int qd = (int)(((float)k + 0.5f) * __frcp_rn((float)off));
        k -= qd * off;
      }
      const int sp = op - off + k;
      uint8_t b = 0;
      if (off <= SN_WIN - 64) {
        if ((int)lane < l) b = s_ring[sp & (SN_WIN - 1)];
      } else {   // behind the ring: flushed long ago
        if (fence_due) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
          fence_due = false;
        }
        if ((int)lane < l) b = __hip_atomic_load(dst + sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __builtin_amdgcn_wave_barrier();
      if ((int)lane < l) s_ring[(op + (int)lane) & (SN_WIN - 1)] = b;
      op += l;
      ip += hdr;
    }
    __builtin_amdgcn_wave_barrier();
    if (op - flushed >= SN_FLUSH) flush();
  }
  if (op != dst_len) return PQE_SNAPPY;
  for (int j = flushed + (int)lane; j < op; j += 64) dst[j] = s_ring[j & (SN_WIN - 1)];
  return PQE_NONE;
}

// one wave per page: the body, uncompressed, to body + dst_off (v2: levels, then the values from the next 16-byte boundary on); a page
// that is not compressed stays where it is
__global__ __launch_bounds__(64) void k_pq_decompress(const DevPage* __restrict__ pages, const uint8_t* __restrict__ chunk, int64_t chunk_readable, uint8_t* __restrict__ body,
                                                      DevPageState* __restrict__ states) {
  __shared__ __align__(16) uint8_t s_ring[SN_WIN];
  const DevPage pg = pages[blockIdx.x];
  const unsigned lane = lane_id();
  const uint8_t* src = chunk + pg.src_off;
  uint8_t* dst = body + pg.dst_off;
  int64_t n_in = pg.comp_bytes, n_out = pg.uncomp_bytes, values_off = 0;
  int err = PQE_NONE;
  if (pg.in_chunk) {
    if (pg.type == PAGE_DATA_V2) {
      values_off = pg.def_bytes;
      n_out -= pg.def_bytes;
    }
    if (pg.comp_bytes != pg.uncomp_bytes) err = PQE_SNAPPY;
  } else {
    if (pg.type == PAGE_DATA_V2) {
      for (int64_t j = lane; j < pg.def_bytes; j += 64) dst[j] = src[j];
      values_off = pq_align16(pg.def_bytes);
      src += pg.def_bytes;
      dst += values_off;
      n_in -= pg.def_bytes;
      n_out -= pg.def_bytes;
    }
    if (n_out > 0) err = snappy_wave(src, (int)n_in, chunk_readable - (src - chunk) + (int64_t)((uintptr_t)src & 3), dst, (int)n_out, s_ring);
    else if (n_in > 1) err = PQE_SNAPPY;
  }
  if (lane == 0) {
    DevPageState st;
    st.values_off = values_off;
    st.values_end = values_off + n_out;
    st.nonnull = pg.num_values;
    st.error = err;
    states[blockIdx.x] = st;
  }
}

// ------------------------------------------------------------------------------------------------- definition levels
__device__ __forceinline__ bool dev_varint(const uint8_t* p, int64_t& pos, int64_t end, uint64_t& out) {
  uint64_t v = 0;
  for (int shift = 0; shift < 64; shift += 7) {
    if (pos >= end) return false;
    const uint8_t b = p[pos++];
    v |= (uint64_t)(b & 0x7F) << shift;
    if (!(b & 0x80)) {
      out = v;
      return true;
    }
  }
  return false;
}
// one wave per data page of a nullable column
__global__ __launch_bounds__(64) void k_pq_levels(const DevPage* __restrict__ pages, const uint8_t* __restrict__ chunk, const uint8_t* __restrict__ body,
                                                  DevPageState* __restrict__ states, unsigned long long* __restrict__ valid) {
  const DevPage pg = pages[blockIdx.x];
  if (pg.type == PAGE_DICTIONARY) return;
  DevPageState st = states[blockIdx.x];
  if (st.error) return;
  const unsigned lane = lane_id();
  const uint8_t* b = pq_page_body(pg, chunk, body);
  const uint8_t* lv;
  int64_t lvn;
  int err = PQE_NONE;
  if (pg.type == PAGE_DATA) {
    if (pg.uncomp_bytes < 4) err = PQE_LEVELS;
    const uint32_t lb = err ? 0u : ((uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24));
    if (!err && (int64_t)lb + 4 > pg.uncomp_bytes) err = PQE_LEVELS;
    lv = b + 4;
    lvn = lb;
    st.values_off = 4 + (int64_t)lb;
  } else {
    lv = b;
    lvn = pg.def_bytes;
  }
  int64_t pos = 0, row = 0;
  unsigned long long mine = 0;   // this lane's share of the non-null count
  while (!err && row < pg.num_values) {
    uint64_t h;
    if (!dev_varint(lv, pos, lvn, h)) { err = PQE_LEVELS; break; }
    const int64_t left = pg.num_values - row;
    if (h & 1) {
      const uint64_t groups = h >> 1;
      if (groups == 0 || groups > (uint64_t)(lvn - pos)) { err = PQE_LEVELS; break; }
      const int64_t cnt = (int64_t)groups * 8 < left ? (int64_t)groups * 8 : left;
      for (int64_t c0 = (int64_t)lane * 64; c0 < cnt; c0 += 64 * 64) {
        const int take = (int)(cnt - c0 < 64 ? cnt - c0 : 64);
        unsigned long long bits = 0;
        for (int q = 0; q < (take + 7) / 8; q++) bits |= (unsigned long long)lv[pos + c0 / 8 + q] << (8 * q);
        if (take < 64) bits &= (1ull << take) - 1;
        mine += (unsigned long long)__popcll(bits);
        const int64_t dest = pg.row_start + row + c0;
        const int sh = (int)(dest & 63);
        if (bits) {
          atomicOr(&valid[dest >> 6], bits << sh);
          if (sh && (bits >> (64 - sh))) atomicOr(&valid[(dest >> 6) + 1], bits >> (64 - sh));
        }
      }
      pos += (int64_t)groups;
      row += cnt;
    } else {
      const uint64_t c = h >> 1;
      if (c == 0 || pos >= lvn) { err = PQE_LEVELS; break; }
      const uint8_t val = lv[pos++];
      const int64_t cnt = c < (uint64_t)left ? (int64_t)c : left;
      if (val & 1) {
        const int64_t d0 = pg.row_start + row, d1 = d0 + cnt;   // bits [d0, d1)
        for (int64_t w = (d0 >> 6) + lane; w <= ((d1 - 1) >> 6); w += 64) {
          const int64_t lo = w * 64 > d0 ? w * 64 : d0, hi = (w + 1) * 64 < d1 ? (w + 1) * 64 : d1;
          const int nb = (int)(hi - lo);
          const unsigned long long m = (nb == 64 ? ~0ull : ((1ull << nb) - 1)) << (lo & 63);
          atomicOr(&valid[w], m);
        }
        if (lane == 0) mine += (unsigned long long)cnt;
      }
      row += cnt;
    }
  }
  const unsigned long long total = wave_sum<unsigned long long>(mine);
  if (lane == 0) {
    st.nonnull = (int32_t)total;
    st.error = err;
    states[blockIdx.x] = st;
  }
}

// ------------------------------------------------------------------------------------------------------------ values
struct PqDevArgs {
  const DevPage* pages;
  const DevPageState* states;
  const uint8_t* chunk;
  const uint8_t* body;
  const void* dict_target;   // dictionary in the target representation (string ranks), or null:
  const uint8_t* dict_raw;   // the PLAIN dictionary page in the body buffer
  int32_t n_pages, dict_count;
  int32_t phys_w, big_endian, sign_extend;
  int32_t* error;
  const int64_t* value_starts;   // per page: first non-null value of the page, chunk-wide (k_pq_value_starts)
};
// the exclusive prefix of the data pages' non-null counts, once per chunk (one workgroup; round 5 had every decode workgroup re-sum the
// earlier pages' counts — O(pages^2) loads, billions for a chunk written with tiny pages)
__global__ __launch_bounds__(BLOCK) void k_pq_value_starts(const DevPage* __restrict__ pages, const DevPageState* __restrict__ states, int n_pages, int64_t* __restrict__ starts) {
  __shared__ unsigned long long s_w[BLOCK / WAVE];
  __shared__ unsigned long long s_carry;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  for (int base = 0; base < n_pages; base += BLOCK) {
    const int q = base + (int)threadIdx.x;
    const unsigned long long v = q < n_pages && pages[q].type != PAGE_DICTIONARY ? (unsigned long long)states[q].nonnull : 0ull;
    const unsigned long long inc = wave_inclusive_sum<unsigned long long>(v);
    if (lane_id() == 63) s_w[threadIdx.x >> 6] = inc;
    __syncthreads();
    unsigned long long before = s_carry;
    for (int w = 0; w < (int)(threadIdx.x >> 6); w++) before += s_w[w];
    if (q < n_pages) starts[q] = (int64_t)(before + inc - v);
    __syncthreads();
    if (threadIdx.x == BLOCK - 1) s_carry = before + inc;
    __syncthreads();
  }
}
template <typename T>
__device__ __forceinline__ T pq_plain_value(const uint8_t* src, int phys_w, int big_endian, int sign_extend) {
  if (big_endian) {   // n-byte big-endian two's complement -> i128
    i128 x = (int8_t)src[0];
    for (int i = 1; i < phys_w; i++) x = (x << 8) | src[i];
    return (T)x;
  }
  if (phys_w == 4) {
    uint32_t u;
    memcpy(&u, src, 4);
    return sign_extend ? (T)(int32_t)u : (T)u;
  }
  uint64_t u;
  memcpy(&u, src, 8);
  return sign_extend ? (T)(int64_t)u : (T)u;
}
// the low phys_w bytes of a 64-bit value (DELTA_BINARY_PACKED wraps in the physical type's width) as the target type
template <typename T>
__device__ __forceinline__ T pq_from_u64(uint64_t v, int phys_w, int sign_extend) {
  if (phys_w == 4) return sign_extend ? (T)(int32_t)(uint32_t)v : (T)(uint32_t)v;
  return sign_extend ? (T)(int64_t)v : (T)v;
}
__device__ __forceinline__ uint64_t pq_unpack64(const uint8_t* base, uint64_t bit, int bw) {
  if (bw == 0) return 0;
  const uint8_t* q = base + (bit >> 3);
  const int s = (int)(bit & 7);
  uint64_t lo = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) lo |= (uint64_t)q[i] << (8 * i);
  uint64_t v = lo >> s;
  if (s + bw > 64) v |= (uint64_t)q[8] << (64 - s);
  return bw == 64 ? v : (v & ((1ull << bw) - 1));
}

constexpr int PQ_RB = 128;     // run headers parsed per batch
constexpr int PQ_DM = 64;      // DELTA miniblocks per batch
constexpr int PQ_WIN = 8192;   // bytes of a run stream staged in LDS per batch
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_pq_decode_pages(PqDevArgs a, T* __restrict__ out) {
  __shared__ int64_t s_start[PQ_RB > PQ_DM ? PQ_RB : PQ_DM];   // run: first value (page-relative) / miniblock: first value of the batch
  __shared__ int64_t s_pay[PQ_RB > PQ_DM ? PQ_RB : PQ_DM];     // RLE value or byte offset of the packed bits / miniblock: byte offset
  __shared__ uint64_t s_min[PQ_DM];                             // miniblock: min delta of its block
  __shared__ int32_t s_bw[PQ_RB > PQ_DM ? PQ_RB : PQ_DM];      // bit width (runs: | RUN_PACKED)
  __shared__ int64_t s_pos, s_batch_end;
  __shared__ int32_t s_n, s_err;
  __shared__ uint64_t s_wsum[BLOCK / WAVE], s_carry;
  const int p = blockIdx.x;
  const DevPage pg = a.pages[p];
  if (pg.type == PAGE_DICTIONARY) return;
  const DevPageState st = a.states[p];
  const int64_t value_start = a.value_starts[p];   // first non-null value of the page, chunk-wide
  const uint8_t* vals = pq_page_body(pg, a.chunk, a.body) + st.values_off;
  const int64_t vbytes = st.values_end - st.values_off;
  const int64_t n = st.nonnull;
  T* o = out + value_start;
  const int tid = threadIdx.x;
  // gridDim.y workgroups per page: PLAIN and BYTE_STREAM_SPLIT values are independent and shared out; a run / block stream is walked by one
  const int64_t part_first = (int64_t)blockIdx.y * BLOCK + tid, part_step = (int64_t)gridDim.y * BLOCK;
  const bool parallel_enc = pg.encoding == ENC_PLAIN || pg.encoding == ENC_BYTE_STREAM_SPLIT;
  if (!parallel_enc && blockIdx.y != 0) return;
  auto dict_value = [&](uint32_t idx) -> T {
    if (idx >= (uint32_t)a.dict_count) idx = 0;   // (a corrupt packed index stays inside the dictionary, as on the host path)
    return a.dict_target ? reinterpret_cast<const T*>(a.dict_target)[idx] : pq_plain_value<T>(a.dict_raw + (int64_t)idx * a.phys_w, a.phys_w, a.big_endian, a.sign_extend);
  };
  if (n == 0) return;
  if (pg.encoding == ENC_PLAIN) {
    if (n * a.phys_w > vbytes) {
      if (tid == 0) atomicMax(a.error, PQE_VALUES);
      return;
    }
    for (int64_t i = part_first; i < n; i += part_step) o[i] = pq_plain_value<T>(vals + i * a.phys_w, a.phys_w, a.big_endian, a.sign_extend);
    return;
  }
  if (pg.encoding == ENC_BYTE_STREAM_SPLIT) {
    if (n * a.phys_w > vbytes) {
      if (tid == 0) atomicMax(a.error, PQE_VALUES);
      return;
    }
    for (int64_t i = part_first; i < n; i += part_step) {
      uint8_t tmp[16];
      for (int k = 0; k < a.phys_w; k++) tmp[k] = vals[(int64_t)k * n + i];
      o[i] = pq_plain_value<T>(tmp, a.phys_w, a.big_endian, a.sign_extend);
    }
    return;
  }
  if (pg.encoding == ENC_RLE_DICTIONARY || pg.encoding == ENC_PLAIN_DICTIONARY) {
    // ---- run headers in batches: a window of the stream is staged in LDS by everybody (a header read off HBM by one lane is a
    // microsecond; a bit-packed run of 504 one-bit values is 64 bytes, so a window holds a batch's headers), thread 0 walks up to
    // PQ_RB headers inside it, everybody decodes the batch's values from HBM
    __shared__ uint8_t s_win[PQ_WIN];
    if (tid == 0) {
      s_pos = 1;
      s_err = vbytes < 1 || vals[0] > 32 ? PQE_RUNS : PQE_NONE;
    }
    __syncthreads();
    const int bw = vals[0];
    const int rle_bytes = (bw + 7) / 8;
    int64_t done = 0;
    while (done < n && !s_err) {
      const int64_t wbase = s_pos;
      const int64_t wlen = vbytes - wbase < PQ_WIN ? vbytes - wbase : PQ_WIN;
      for (int64_t j = tid; j < wlen; j += BLOCK) s_win[j] = vals[wbase + j];
      __syncthreads();
      if (tid == 0) {
        int64_t pos = 0, d = done;   // pos: window-relative
        int nr = 0, err = PQE_NONE;
        while (nr < PQ_RB && d < n) {
          if (pos + 10 > wlen && wbase + wlen < vbytes) break;   // the next header may leave the window: the next batch starts there
          uint64_t h;
          if (!dev_varint(s_win, pos, wlen, h)) { err = PQE_RUNS; break; }
          const int64_t left = n - d;
          if (h & 1) {
            const uint64_t groups = h >> 1;
            if (groups == 0 || groups > (uint64_t)INT32_MAX || groups * (uint64_t)bw > (uint64_t)(vbytes - wbase - pos)) { err = PQE_RUNS; break; }   // (no overflow: 2^31 x 32)
            s_start[nr] = d;
            s_pay[nr] = wbase + pos;
            s_bw[nr] = bw | RUN_PACKED;
            pos += (int64_t)groups * bw;
            d += (int64_t)groups * 8 < left ? (int64_t)groups * 8 : left;
          } else {
            const uint64_t c = h >> 1;
            if (c == 0 || pos + rle_bytes > wlen) { err = PQE_RUNS; break; }
            uint64_t v = 0;
            for (int i = 0; i < rle_bytes; i++) v |= (uint64_t)s_win[pos + i] << (8 * i);
            pos += rle_bytes;
            if (v >= (uint64_t)a.dict_count) { err = PQE_DICT_INDEX; break; }
            s_start[nr] = d;
            s_pay[nr] = (int64_t)v;
            s_bw[nr] = bw;
            d += c < (uint64_t)left ? (int64_t)c : left;
          }
          nr++;
          if (pos >= wlen) break;   // (a bit-packed run reaching beyond the window: its successor's header is not staged)
        }
        if (!err && nr == 0) err = PQE_RUNS;   // (no progress: cannot happen with a window of PQ_WIN >= 10 bytes unless the stream is cut)
        s_pos = wbase + pos;
        s_n = nr;
        s_batch_end = d;
        s_err = err;
      }
      __syncthreads();
      const int nr = s_n;
      const int64_t batch_end = s_batch_end;
      if (!s_err) {
        // eight consecutive values per thread: one search for the run, one 64-bit window of packed bits (a group of eight indices of up to
        // 7 bits), eight look-ups, 8 x sizeof(T) contiguous bytes stored
        const uint64_t mask = bw == 32 ? 0xFFFFFFFFull : ((1ull << bw) - 1);
        for (int64_t v0 = done + (int64_t)tid * 8; v0 < batch_end; v0 += (int64_t)BLOCK * 8) {
          int lo = 0, hi = nr - 1;
          while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (s_start[mid] <= v0) lo = mid; else hi = mid - 1;
          }
          const int64_t vend = v0 + 8 < batch_end ? v0 + 8 : batch_end;
          for (int64_t v = v0; v < vend;) {
            while (lo + 1 < nr && s_start[lo + 1] <= v) lo++;
            const int64_t rend = lo + 1 < nr ? s_start[lo + 1] : batch_end;
            const int take = (int)((vend < rend ? vend : rend) - v);
            if (s_bw[lo] < 0) {
              const uint64_t bit = (uint64_t)(v - s_start[lo]) * (uint64_t)bw;
              if (take * bw + (int)(bit & 7) <= 64) {
                const uint8_t* q = vals + s_pay[lo] + (bit >> 3);
                uint64_t w = 0;
#pragma unroll
                for (int i = 0; i < 8; i++) w |= (uint64_t)q[i] << (8 * i);
                w >>= (bit & 7);
                for (int j = 0; j < take; j++) {
                  o[v + j] = dict_value((uint32_t)(w & mask));
                  w >>= bw;
                }
              } else {
                for (int j = 0; j < take; j++) o[v + j] = dict_value((uint32_t)pq_unpack64(vals + s_pay[lo], bit + (uint64_t)j * (uint64_t)bw, bw));
              }
            } else {
              const T val = dict_value((uint32_t)s_pay[lo]);
              for (int j = 0; j < take; j++) o[v + j] = val;
            }
            v += take;
          }
        }
      }
      done = batch_end;
      __syncthreads();
    }
    if (tid == 0 && s_err) atomicMax(a.error, s_err);
    return;
  }
  if (pg.encoding == ENC_DELTA_BINARY_PACKED) {
    // ---- <block size> <miniblocks per block> <total count> <first value>, then blocks of <min delta> <bit widths> <miniblocks>
    __shared__ uint64_t s_hdr[4];
    if (tid == 0) {
      int64_t pos = 0;
      uint64_t block = 0, minis = 0, total = 0, first = 0;
      bool ok = dev_varint(vals, pos, vbytes, block) && dev_varint(vals, pos, vbytes, minis) && dev_varint(vals, pos, vbytes, total) && dev_varint(vals, pos, vbytes, first);
      ok = ok && block > 0 && block % 128 == 0 && block <= (1u << 20) && minis > 0 && minis <= PQ_DM && block % minis == 0 && (block / minis) % 32 == 0 && (int64_t)total >= n;
      s_hdr[0] = block;
      s_hdr[1] = minis;
      s_hdr[2] = (first >> 1) ^ (0 - (first & 1));   // zigzag
      s_pos = pos;
      s_err = ok ? PQE_NONE : PQE_DELTA;
      s_carry = s_hdr[2];
    }
    __syncthreads();
    if (s_err) {
      if (tid == 0) atomicMax(a.error, s_err);
      return;
    }
    const int64_t minis = (int64_t)s_hdr[1], per_mini = (int64_t)s_hdr[0] / minis;
    if (tid == 0) o[0] = pq_from_u64<T>(s_hdr[2], a.phys_w, a.sign_extend);
    int64_t done = 1;
    while (done < n && !s_err) {
      if (tid == 0) {
        int64_t pos = s_pos, bv = 0;
        int nm = 0, err = PQE_NONE;
        const int64_t left = n - done;
        while (nm + minis <= PQ_DM && bv < left) {
          uint64_t md;
          if (!dev_varint(vals, pos, vbytes, md) || pos + minis > vbytes) { err = PQE_DELTA; break; }
          md = (md >> 1) ^ (0 - (md & 1));
          const uint8_t* widths = vals + pos;
          pos += minis;
          for (int64_t m = 0; m < minis && bv < left; m++) {
            const int w = widths[m];
            if (w > 64 || pos + per_mini * w / 8 > vbytes) { err = PQE_DELTA; break; }
            s_start[nm] = bv;
            s_pay[nm] = pos;
            s_bw[nm] = w;
            s_min[nm] = md;
            pos += per_mini * w / 8;
            bv += per_mini;
            nm++;
          }
          if (err) break;
        }
        s_pos = pos;
        s_n = nm;
        s_batch_end = bv < left ? bv : left;   // values of this batch
        s_err = err;
      }
      __syncthreads();
      const int64_t bvals = s_batch_end;
      if (!s_err) {
        // deltas -> running sum, 4 consecutive values per thread and 1024 per sweep
        for (int64_t c0 = 0; c0 < bvals; c0 += BLOCK * 4) {
          uint64_t d[4], run = 0;
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int64_t i = c0 + (int64_t)tid * 4 + u;
            d[u] = 0;
            if (i < bvals) {
              const int m = (int)(i / per_mini);
              d[u] = pq_unpack64(vals + s_pay[m], (uint64_t)(i - s_start[m]) * (uint64_t)s_bw[m], s_bw[m]) + s_min[m];
            }
            run += d[u];
            d[u] = run;   // inclusive within the thread
          }
          const uint64_t inc = wave_inclusive_sum<uint64_t>(run);
          if (lane_id() == 63) s_wsum[tid >> 6] = inc;
          __syncthreads();
          uint64_t base = s_carry + inc - run;
          for (int w = 0; w < (tid >> 6); w++) base += s_wsum[w];
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int64_t i = c0 + (int64_t)tid * 4 + u;
            if (i < bvals) o[done + i] = pq_from_u64<T>(base + d[u], a.phys_w, a.sign_extend);
          }
          __syncthreads();
          if (tid == BLOCK - 1) s_carry = base + run;
          __syncthreads();
        }
      }
      done += bvals;
      __syncthreads();
    }
    if (tid == 0 && s_err) atomicMax(a.error, s_err);
    return;
  }
  if (tid == 0) atomicMax(a.error, PQE_VALUES);
}

namespace {

// what the host learns from the page headers alone
struct DevPlan {
  StageVec<DevPage> pages;
  bool host_decompress = false;   // the pages' bodies are decompressed by the host into the upload buffer (ZSTD; Snappy unless parquet.snappy=device)
  int64_t upload_bytes = 0;       // host_decompress: the uncompressed bodies, one behind the other
  int64_t body_bytes = 0, rows = 0;
  int dict_page = -1;
  int32_t dict_count = 0;
};
const char* pq_device_error(int e) {
  switch (e) {
    case PQE_SNAPPY: return "parquet: a Snappy page does not decompress to the size in its header";
    case PQE_LEVELS: return "parquet: definition levels overrun the page";
    case PQE_VALUES: return "parquet: PLAIN values overrun the page";
    case PQE_DICT_INDEX: return "parquet: dictionary index out of range";
    case PQE_DELTA: return "parquet: malformed DELTA_BINARY_PACKED page";
    case PQE_RUNS: return "parquet: bit-packed run overruns its page";
  }
  return "parquet: page decode error";
}
// false = this chunk is not for the device path (codec, target or an encoding it does not take): plan_chunk's path decodes it.
// Malformed headers raise the same errors as plan_chunk.
bool plan_chunk_device(const uint8_t* chunk, int64_t nbytes, const dfgpu_parquet_column& col, DevPlan& D) {
  if (col.field.type == DFGPU_UTF8 || col.physical_type == DFGPU_PARQUET_BOOLEAN) return false;
  if (col.codec != DFGPU_PARQUET_UNCOMPRESSED && col.codec != DFGPU_PARQUET_SNAPPY && col.codec != DFGPU_PARQUET_ZSTD) return false;
  // Who undoes the compression: a wave per page decodes Snappy at ~12 MB/s (k_pq_decompress: one instruction stream, a few hundred
  // cycles per element) where a host core does ~1.5 GB/s — so by default the host decompresses (as it must for ZSTD) straight into the
  // upload buffer and the device does everything after that; parquet.snappy=device sends the compressed bytes and decodes them there.
  D.host_decompress = col.codec == DFGPU_PARQUET_ZSTD || (col.codec == DFGPU_PARQUET_SNAPPY && option_str("parquet.snappy", "host") != "device");
  check_target(col);
  const int pw = phys_width(col);
  const uint8_t* p = chunk;
  const uint8_t* end = chunk + nbytes;
  bool data_seen = false;
  while (D.rows < col.num_values) {
    DFGPU_CHECK(p < end, "parquet: the chunk ends before its num_values rows");
    Thrift t{p, end};
    PageHeader h = parse_page_header(t);
    p = t.p;
    DFGPU_CHECK(end - p >= h.compressed, "parquet: page body overruns the chunk");
    const uint8_t* raw = p;
    p += h.compressed;
    if (h.type == PAGE_INDEX) continue;
    DevPage g{};
    g.src_off = raw - chunk;
    g.comp_bytes = h.compressed;
    g.uncomp_bytes = h.uncompressed;
    g.num_values = h.num_values;
    g.encoding = h.encoding;
    g.type = h.type;
    g.codec = col.codec;
    g.dst_off = D.body_bytes;
    if (h.type == PAGE_DICTIONARY) {
      DFGPU_CHECK(D.dict_page < 0 && !data_seen, "parquet: more than one dictionary page, or a dictionary page after data pages");
      DFGPU_CHECK(h.encoding == ENC_PLAIN || h.encoding == ENC_PLAIN_DICTIONARY, "parquet: dictionary page encoding " + std::to_string(h.encoding));
      DFGPU_CHECK(h.num_values >= 0, "parquet: negative dictionary size");
      if (pw > 0) DFGPU_CHECK((int64_t)h.uncompressed >= (int64_t)h.num_values * pw, "parquet: dictionary page shorter than its value count");
      D.dict_page = (int)D.pages.size();
      D.dict_count = h.num_values;
      g.in_chunk = g.codec == DFGPU_PARQUET_UNCOMPRESSED || D.host_decompress;
      if (!g.in_chunk) D.body_bytes += pq_align16((int64_t)h.uncompressed) + 16;
      D.upload_bytes += pq_align16((int64_t)h.uncompressed) + 16;
      D.pages.push_back(g);
      continue;
    }
    DFGPU_CHECK(h.type == PAGE_DATA || h.type == PAGE_DATA_V2, "parquet: unknown page type " + std::to_string(h.type));
    DFGPU_CHECK(h.num_values >= 0 && D.rows + h.num_values <= col.num_values, "parquet: pages hold more rows than the chunk's num_values");
    DFGPU_CHECK(h.def_bytes >= 0 && h.rep_bytes >= 0, "parquet: negative level byte length");
    if (h.type == PAGE_DATA) {
      if (col.max_definition_level == 1) DFGPU_CHECK(h.def_encoding == ENC_RLE, "parquet: definition levels with the deprecated BIT_PACKED encoding");
    } else {
      DFGPU_CHECK(h.rep_bytes == 0, "parquet: repetition levels in a flat column");
      DFGPU_CHECK((int64_t)h.def_bytes <= h.compressed && (int64_t)h.def_bytes <= h.uncompressed, "parquet: v2 level bytes overrun the page");
      g.def_bytes = h.def_bytes;
      if (!h.v2_compressed) g.codec = DFGPU_PARQUET_UNCOMPRESSED;
    }
    switch (h.encoding) {
      case ENC_PLAIN:
        DFGPU_CHECK(pw > 0, "parquet: PLAIN-encoded BYTE_ARRAY pages (strings outside a dictionary): read the column as Utf8 (field.type DFGPU_UTF8) instead of dictionary indices");
        break;
      case ENC_RLE_DICTIONARY: case ENC_PLAIN_DICTIONARY:
        DFGPU_CHECK(D.dict_page >= 0 || h.num_values == 0, "parquet: dictionary-encoded page without a dictionary page");
        break;
      case ENC_DELTA_BINARY_PACKED:
        DFGPU_CHECK(col.physical_type == DFGPU_PARQUET_INT32 || col.physical_type == DFGPU_PARQUET_INT64, "parquet: DELTA_BINARY_PACKED on a non-integer column");
        break;
      case ENC_BYTE_STREAM_SPLIT:
        DFGPU_CHECK(pw > 0, "parquet: value encoding " + std::to_string(h.encoding) + " on this physical type");
        break;
      default: throw Error("parquet: value encoding " + std::to_string(h.encoding) + " is not supported on the GPU scan path");
    }
    g.row_start = D.rows;
    D.rows += h.num_values;
    g.in_chunk = g.codec == DFGPU_PARQUET_UNCOMPRESSED || D.host_decompress;
    if (!g.in_chunk) D.body_bytes += pq_align16((int64_t)h.uncompressed) + 32;   // (v2: the values start at the next 16-byte boundary behind the levels)
    D.upload_bytes += pq_align16((int64_t)h.uncompressed) + 16;
    D.pages.push_back(g);
    data_seen = true;
  }
  DFGPU_CHECK(D.rows == col.num_values, "parquet: pages hold more rows than the chunk's num_values");
  return !D.pages.empty() && D.pages.size() <= 65535;
}

template <typename T>
void launch_decode_pages(const PqDevArgs& a, void* out) {
  k_pq_decode_pages<T><<<dim3((unsigned)a.n_pages, 8), BLOCK, 0, rt().stream>>>(a, (T*)out);
}

// Everything a chunk needs is enqueued in one go — upload, decompression, levels, values, NULL expansion, the pages' states and the
// decode error word on their way back to pinned memory — and nothing waits: `fin` (run by the caller once the stream has passed this
// point: a scan worker has several chunks in flight) reads what came back, raises a corrupt page's error and settles the NULL count.
extern thread_local double t_upload_ms[6];   // (defined below: where a worker's host time went, DFGPU_TRACE=scan)
// Device buffers the uploads land in, kept per host thread and handed out again only after the chunk that used one has been settled (its
// event has passed: no kernel reads it any more).  That is what lets the upload stream run WITHOUT waiting for the kernel stream — a block
// from the general pool may still be read by kernels this thread enqueued earlier, and an upload that has to wait for them is not handed
// to the copy engine by hipMemcpyAsync: the call itself waits (measured: 1-4 ms per chunk, 2-9 ms with an event wait in front of it).
struct UploadCache {
  std::vector<BufPtr> free;
  size_t cached_bytes = 0;
  std::mutex mu;   // the owner thread takes and gives; dfgpu_mem_trim / dfgpu_shutdown drop the blocks from another thread
  // sizes in classes (powers of two from 256 KB): the chunks of a scan's columns come in a handful of sizes, and a miss costs a drain of
  // the kernel stream
  static size_t size_class(size_t n) {
    size_t c = (size_t)256 << 10;
    while (c < n) c <<= 1;
    return c;
  }
  BufPtr take(size_t n, hipStream_t kernel_stream) {
    const size_t c = size_class(n);
    {
      std::lock_guard<std::mutex> lk(mu);
      for (size_t i = 0; i < free.size(); i++)
        if (free[i]->bytes == c) {
          BufPtr b = std::move(free[i]);
          free.erase(free.begin() + (long)i);
          cached_bytes -= c;
          return b;
        }
    }
    BufPtr b = make_buf(c);
    DFGPU_HIP(hipStreamSynchronize(kernel_stream));   // a pool block: whatever this thread's kernels still read of it is done after this
    return b;
  }
  // A few blocks per size class (the chunks in flight plus one), and no more than `parquet.upload_cache_bytes` (256 MiB) per thread and
  // device in all: with 16 scan workers that bounds what the caches hold at 4 GiB whatever the chunk sizes; what does not fit goes
  // back to the pool.
  void give(BufPtr b) {
    const size_t cap = (size_t)option_int("parquet.upload_cache_bytes", (int64_t)256 << 20);
    std::lock_guard<std::mutex> lk(mu);
    size_t same = 0;
    for (const BufPtr& f : free) same += f->bytes == b->bytes;
    if (same < 6 && free.size() < 48 && cached_bytes + b->bytes <= cap) {
      cached_bytes += b->bytes;
      free.push_back(std::move(b));
    }
  }
  void drop_all() {
    std::vector<BufPtr> gone;
    {
      std::lock_guard<std::mutex> lk(mu);
      gone.swap(free);
      cached_bytes = 0;
    }
  }   // (the blocks go back to the pool here, outside the lock)
};
// What a host thread keeps per DEVICE for the scan (a library thread may serve scans on different GPUs of the process): the upload stream
// and its event, the stream the pages' states come back on, the upload buffers.  The object itself is never destroyed (a thread's cache
// would otherwise give its blocks back to a pool that static destruction may already have taken down), but its BLOCKS belong to the
// runtime: every instance is registered, and dfgpu_mem_trim / dfgpu_shutdown (Runtime::trim -> scan_upload_caches_drop) hand all cached
// upload blocks of the device back before the pool itself is trimmed — so they count as reclaimable, and nothing survives a shutdown.
struct ScanThreadDevice {
  hipStream_t up = nullptr, readback = nullptr;
  hipEvent_t ev_up = nullptr;
  UploadCache cache;
  int device = -1;
};
static std::mutex g_scan_devs_mu;
static std::vector<ScanThreadDevice*>& scan_devs() {
  static std::vector<ScanThreadDevice*>* v = new std::vector<ScanThreadDevice*>();
  return *v;
}
static ScanThreadDevice& scan_thread_device(int device) {
  static thread_local std::map<int, ScanThreadDevice*>* per_device = new std::map<int, ScanThreadDevice*>();
  ScanThreadDevice*& p = (*per_device)[device];
  if (!p) {
    p = new ScanThreadDevice();
    p->device = device;
    std::lock_guard<std::mutex> lk(g_scan_devs_mu);
    scan_devs().push_back(p);
  }
  return *p;
}
struct DeviceChunkKeep {
  int device = -1;
  ~DeviceChunkKeep() {
    if (d_chunk && device >= 0) scan_thread_device(device).cache.give(std::move(d_chunk));
  }
  StageVec<uint8_t> staged, dict_host;
  StageVec<DevPage> pages;
  StageVec<DevPageState> states;
  StageVec<int32_t> err;
  BufPtr d_chunk, d_body, d_states, d_err, d_dict, dense, prefix, d_value_starts;
  std::vector<int32_t> page_type, page_rows;
  int64_t rows = 0;
};
Column decode_chunk_device(const DevPlan& D, const uint8_t* chunk, int64_t nbytes, const dfgpu_parquet_column& col, std::shared_ptr<void>* keep,
                           std::function<void(Column&)>* fin) {
  hipStream_t st = rt().stream;
  dfgpu_field f = col.field;
  f.nullable = col.max_definition_level ? 1 : 0;
  const int out_w = type_width(f.type);
  Column c = alloc_column(f, col.name ? col.name : "", D.rows);
  const int n_pages = (int)D.pages.size();
  auto K = std::make_shared<DeviceChunkKeep>();
  K->rows = D.rows;
  for (const DevPage& g : D.pages) {
    K->page_type.push_back(g.type);
    K->page_rows.push_back(g.num_values);
  }
  // what crosses PCIe (through pinned staging: internal.hpp PinnedBuf): the chunk as it lies in the file — or, when the host undoes the
  // compression, the pages' bodies decompressed straight into the staging buffer, each page then reading as an uncompressed one
  K->pages = D.pages;
  int64_t padded;
  const auto t_fill0 = std::chrono::steady_clock::now();
  if (D.host_decompress) {
    padded = (D.upload_bytes + 64 + 3) & ~(int64_t)3;
    K->staged.resize((size_t)padded);
    int64_t at = 0;
    // (the pages of a big chunk as jobs for whichever worker is free were measured: SF1 medians 15 ms against 10.6 — removed)
    for (DevPage& g : K->pages) {
      uint8_t* to = K->staged.data() + at;
      const uint8_t* raw = chunk + g.src_off;
      const int32_t lv = g.type == PAGE_DATA_V2 ? g.def_bytes : 0;   // (v2: the levels are never compressed)
      if (lv) std::memcpy(to, raw, (size_t)lv);
      decompress(g.codec, raw + lv, (size_t)(g.comp_bytes - lv), to + lv, (size_t)(g.uncomp_bytes - lv));
      g.src_off = at;
      g.comp_bytes = g.uncomp_bytes;
      g.codec = DFGPU_PARQUET_UNCOMPRESSED;
      at += pq_align16((int64_t)g.uncomp_bytes) + 16;
    }
  } else {
    padded = (nbytes + 64 + 3) & ~(int64_t)3;
    K->staged.resize((size_t)padded);   // (uninitialised: StageAllocator)
    std::memcpy(K->staged.data(), chunk, (size_t)nbytes);
  }
  auto tk = std::chrono::steady_clock::now();
  auto lap = [&](int i) {
    const auto now = std::chrono::steady_clock::now();
    t_upload_ms[i] += std::chrono::duration<double, std::milli>(now - tk).count();
    tk = now;
  };
  t_upload_ms[0] += std::chrono::duration<double, std::milli>(tk - t_fill0).count();   // the upload buffer filled (copy or decompression)
  // The uploads go on a stream of their own (per thread), into a buffer no kernel reads any more (UploadCache): the page table rides
  // behind the chunk's bytes in the same buffer.  The kernel stream waits for the upload's event on the device.
  ScanThreadDevice& tdv = scan_thread_device(current_device());
  K->device = current_device();
  if (!tdv.up) {
    DFGPU_HIP(hipStreamCreateWithFlags(&tdv.up, hipStreamNonBlocking));
    DFGPU_HIP(hipEventCreateWithFlags(&tdv.ev_up, hipEventDisableTiming));
  }
  hipStream_t up = tdv.up;
  hipEvent_t ev_up = tdv.ev_up;
  const size_t pages_at = ((size_t)padded + 63) & ~(size_t)63, pages_bytes = (size_t)n_pages * sizeof(DevPage);
  K->d_chunk = tdv.cache.take(pages_at + pages_bytes + 64, st);
  K->d_body = make_buf((size_t)D.body_bytes + 64);
  K->d_states = make_buf((size_t)n_pages * sizeof(DevPageState));
  lap(2);
  // From here on copies out of K's pinned staging and kernels on K's buffers are in flight: should anything below throw (a malformed string
  // dictionary, a decompression error), both streams are drained BEFORE K unwinds — its staging block must not return to the pinned pool, nor
  // its chunk buffer to the upload cache, while the copy engine still reads them
  struct InFlightGuard {
    hipStream_t a, b;
    bool armed = true;
    ~InFlightGuard() {
      if (!armed) return;
      (void)hipStreamSynchronize(a);
      (void)hipStreamSynchronize(b);
    }
  } in_flight{up, st};
  DFGPU_HIP(hipMemcpyAsync(K->d_chunk->ptr, K->staged.data(), (size_t)padded, hipMemcpyHostToDevice, up));
  DFGPU_HIP(hipMemcpyAsync((uint8_t*)K->d_chunk->ptr + pages_at, K->pages.data(), pages_bytes, hipMemcpyHostToDevice, up));
  DFGPU_HIP(hipEventRecord(ev_up, up));
  DFGPU_HIP(hipStreamWaitEvent(st, ev_up, 0));
  lap(3);   // the uploads enqueued
  const DevPage* d_pages = reinterpret_cast<const DevPage*>((const uint8_t*)K->d_chunk->ptr + pages_at);
  K->d_err = make_zero_buf(4);
  thread_metrics().h2d_bytes += padded;
  lap(1);   // buffers + the upload enqueued
  {
    ProfileScope ps(D.body_bytes ? "parquet_decompress_pages" : "parquet_page_states", nbytes + D.body_bytes);
    k_pq_decompress<<<n_pages, 64, 0, st>>>(d_pages, K->d_chunk->as<uint8_t>(), padded, K->d_body->as<uint8_t>(), K->d_states->as<DevPageState>());
    DFGPU_HIP(hipGetLastError());
  }
  const bool nullable = col.max_definition_level == 1;
  if (nullable) {
    c.validity = make_zero_buf(bitmap_bytes(D.rows) + 8);
    ProfileScope ps("parquet_levels", D.rows / 8);
    k_pq_levels<<<n_pages, 64, 0, st>>>(d_pages, K->d_chunk->as<uint8_t>(), K->d_body->as<uint8_t>(), K->d_states->as<DevPageState>(),
                                        reinterpret_cast<unsigned long long*>(c.validity->ptr));
    DFGPU_HIP(hipGetLastError());
  }
  // dictionary: numeric ones are read where the dictionary page lies (uncompressed in the chunk, or where it was decompressed); a string
  // dictionary is sorted on the host (the column's DictValues live there) and the ranks go up
  PqDevArgs a{};
  if (D.dict_page >= 0) {
    const DevPage& dp = D.pages[(size_t)D.dict_page];
    if (col.physical_type == DFGPU_PARQUET_BYTE_ARRAY) {
      ChunkPlan P;
      P.dict_page.resize((size_t)dp.uncomp_bytes + 16);
      if (D.host_decompress) std::memcpy(P.dict_page.data(), K->staged.data() + K->pages[(size_t)D.dict_page].src_off, (size_t)dp.uncomp_bytes);
      else decompress(col.codec, chunk + dp.src_off, (size_t)dp.comp_bytes, P.dict_page.data(), (size_t)dp.uncomp_bytes);
      P.dict_page.resize((size_t)dp.uncomp_bytes);
      P.dict_count = D.dict_count;
      std::vector<int32_t> rank;
      c.dict = string_dictionary(P, c.name, rank);
      K->dict_host.resize(std::max<size_t>(rank.size(), 1) * 4, 0);
      if (!rank.empty()) std::memcpy(K->dict_host.data(), rank.data(), rank.size() * 4);
      K->d_dict = make_buf(K->dict_host.size() + 16);
      DFGPU_HIP(hipMemcpyAsync(K->d_dict->ptr, K->dict_host.data(), K->dict_host.size(), hipMemcpyHostToDevice, st));
      a.dict_target = K->d_dict->ptr;
    } else {
      a.dict_raw = dp.in_chunk ? K->d_chunk->as<uint8_t>() + K->pages[(size_t)D.dict_page].src_off : K->d_body->as<uint8_t>() + dp.dst_off;
    }
  } else if (col.physical_type == DFGPU_PARQUET_BYTE_ARRAY) {
    ChunkPlan P;   // (a chunk of NULLs only: no dictionary page)
    std::vector<int32_t> rank;
    c.dict = string_dictionary(P, c.name, rank);
  }
  a.pages = d_pages;
  a.states = K->d_states->as<DevPageState>();
  a.chunk = K->d_chunk->as<uint8_t>();
  a.body = K->d_body->as<uint8_t>();
  a.n_pages = n_pages;
  a.dict_count = std::max(D.dict_count, 1);
  a.phys_w = std::max(phys_width(col), 1);
  a.big_endian = col.physical_type == DFGPU_PARQUET_FIXED_LEN_BYTE_ARRAY;
  a.sign_extend = is_signed_type(f.type) && f.type != DFGPU_FLOAT64;
  a.error = K->d_err->as<int32_t>();
  // a nullable column's values are decoded densely and spread over the rows by the validity bitmap (how many NULLs there are is
  // known on the device only at this point; a column without any pays one pass over its values for not waiting)
  void* target = c.data->ptr;
  if (nullable) {
    K->dense = make_buf((size_t)std::max<int64_t>(D.rows, 1) * out_w + 16);
    target = K->dense->ptr;
  }
  K->d_value_starts = make_buf((size_t)n_pages * 8);
  k_pq_value_starts<<<1, BLOCK, 0, st>>>(d_pages, K->d_states->as<DevPageState>(), n_pages, K->d_value_starts->as<int64_t>());
  a.value_starts = K->d_value_starts->as<int64_t>();
  {
    ProfileScope ps("parquet_decode_pages", D.body_bytes + D.rows * out_w);
    switch (out_w) {
      case 16: launch_decode_pages<i128>(a, target); break;
      case 8: launch_decode_pages<uint64_t>(a, target); break;
      case 4: launch_decode_pages<uint32_t>(a, target); break;
      default: launch_decode_pages<uint8_t>(a, target); break;
    }
    DFGPU_HIP(hipGetLastError());
  }
  if (nullable) {
    K->prefix = make_buf((size_t)((D.rows + 63) / 64 + 1) * 8);
    scan_mask_popcounts(c.validity->as<uint64_t>(), nullptr, D.rows, K->prefix->as<uint64_t>());
    switch (out_w) {
      case 16: launch_expand<i128>(K->dense->ptr, c.valid_words(), K->prefix->as<uint64_t>(), D.rows, c.data->ptr); break;
      case 8: launch_expand<uint64_t>(K->dense->ptr, c.valid_words(), K->prefix->as<uint64_t>(), D.rows, c.data->ptr); break;
      case 4: launch_expand<uint32_t>(K->dense->ptr, c.valid_words(), K->prefix->as<uint64_t>(), D.rows, c.data->ptr); break;
      default: launch_expand<uint8_t>(K->dense->ptr, c.valid_words(), K->prefix->as<uint64_t>(), D.rows, c.data->ptr); break;
    }
    c.null_count = -1;   // (settled by `fin`)
  }
  lap(4);   // kernels enqueued
  K->states.assign((size_t)n_pages, DevPageState{});
  K->err.assign(1, 0);
  thread_metrics().d2h_bytes += (int64_t)n_pages * (int64_t)sizeof(DevPageState) + 4;
  // `finish` runs once the stream is past this chunk: the pages' states and the error word come back on a stream of their own (on the
  // worker's stream the read-back would queue behind the chunks enqueued since; a kernel writing them into pinned host memory and a
  // hipMemcpyAsync issued here were both measured: the first makes later uploads on the stream wait on the host, the second returns only
  // when the stream has reached it)
  std::function<void(Column&)> finish = [K, nullable, n_pages](Column& col_out) {
    hipStream_t& readback = scan_thread_device(K->device).readback;   // (`finish` runs on a thread whose current device is the chunk's)
    if (!readback) DFGPU_HIP(hipStreamCreateWithFlags(&readback, hipStreamNonBlocking));
    DFGPU_HIP(hipMemcpyAsync(K->states.data(), K->d_states->ptr, (size_t)n_pages * sizeof(DevPageState), hipMemcpyDeviceToHost, readback));
    DFGPU_HIP(hipMemcpyAsync(K->err.data(), K->d_err->ptr, 4, hipMemcpyDeviceToHost, readback));
    DFGPU_HIP(hipStreamSynchronize(readback));
    int64_t values = 0;
    for (size_t q = 0; q < K->states.size(); q++) {
      const DevPageState& s = K->states[q];
      if (s.error) throw Error(pq_device_error(s.error));
      if (K->page_type[q] != PAGE_DICTIONARY) {
        DFGPU_CHECK(s.nonnull >= 0 && s.nonnull <= K->page_rows[q], "parquet: definition levels overrun the page");
        values += s.nonnull;
      }
    }
    if (K->err[0]) throw Error(pq_device_error(K->err[0]));
    if (nullable) {
      col_out.null_count = K->rows - values;
      if (values == K->rows) col_out.validity.reset();
    }
  };
  if (keep && fin) {
    *keep = K;
    *fin = std::move(finish);
    in_flight.armed = false;   // (the caller holds K until the stream has passed this chunk)
  } else {
    DFGPU_HIP(hipStreamSynchronize(st));
    in_flight.armed = false;
    finish(c);
  }
  return c;
}

}  // namespace

// (declared in internal.hpp: Runtime::trim calls them)
void scan_upload_caches_drop(int device) {
  std::vector<ScanThreadDevice*> mine;
  {
    std::lock_guard<std::mutex> lk(g_scan_devs_mu);
    for (ScanThreadDevice* s : scan_devs())
      if (device < 0 || s->device == device) mine.push_back(s);
  }
  for (ScanThreadDevice* s : mine) s->cache.drop_all();
}
int64_t scan_upload_caches_bytes(int device) {
  std::lock_guard<std::mutex> lk(g_scan_devs_mu);
  int64_t n = 0;
  for (ScanThreadDevice* s : scan_devs())
    if (device < 0 || s->device == device) {
      std::lock_guard<std::mutex> cl(s->cache.mu);
      n += (int64_t)s->cache.cached_bytes;
    }
  return n;
}

namespace {

thread_local double t_plan_ms = 0;   // host halves of this thread's chunks (DFGPU_TRACE_SCAN)
thread_local double t_upload_ms[6] = {0, 0, 0, 0, 0, 0};   // (device path: [0] upload buffer filled, [1] buffers + upload enqueued, [4] kernels enqueued)
// `keep` (optional): instead of waiting for the uploads, the host buffers they read from are handed to the caller, who lets go of
// them once the stream has passed this point — a scan worker plans its next chunk while this one crosses PCIe
struct ChunkSources {
  ChunkPlan P;
  StageVec<uint8_t> dict_host;
};
Column decode_chunk(const uint8_t* chunk, int64_t nbytes, const dfgpu_parquet_column& col, std::shared_ptr<void>* keep = nullptr,
                    std::function<void(Column&)>* fin = nullptr) {
  const auto t_plan0 = std::chrono::steady_clock::now();
  auto sources = std::make_shared<ChunkSources>();
  ChunkPlan& P = sources->P;
  if (option_on("parquet.device_decode", true)) {   // pages decoded on the device where the codec / target allow (see plan_chunk_device)
    DevPlan D;
    if (plan_chunk_device(chunk, nbytes, col, D)) {
      t_plan_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_plan0).count();
      return decode_chunk_device(D, chunk, nbytes, col, keep, fin);
    }
  }
  P = plan_chunk(chunk, nbytes, col);
  t_plan_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_plan0).count();
  if (col.field.type == DFGPU_UTF8) return decode_string_chunk(P, col);
  if (col.physical_type == DFGPU_PARQUET_BOOLEAN) return decode_bool_chunk(P, col);
  hipStream_t st = rt().stream;
  dfgpu_field f = col.field;
  f.nullable = col.max_definition_level ? 1 : 0;
  const int out_w = type_width(f.type);
  Column c = alloc_column(f, col.name ? col.name : "", P.rows);
  // dictionary in the target representation
  StageVec<uint8_t>& dict_host = sources->dict_host;
  if (col.physical_type == DFGPU_PARQUET_BYTE_ARRAY) {
    std::vector<int32_t> rank;
    c.dict = string_dictionary(P, c.name, rank);
    dict_host.resize((size_t)std::max<size_t>(rank.size(), 1) * 4, 0);
    if (!rank.empty()) std::memcpy(dict_host.data(), rank.data(), rank.size() * 4);
  } else if (P.dict_count) {
    dict_host = convert_dictionary(P, col, out_w);
  }
  BufPtr d_dict = make_buf(dict_host.size() + 16), d_stage = make_buf(P.staging.size() + 16);
  BufPtr d_pages = make_buf(P.pages.size() * sizeof(PageDesc) + 16), d_runs = make_buf(P.runs.size() * sizeof(Run) + 16);
  auto tk = std::chrono::steady_clock::now();
  auto lap = [&](int i) {
    const auto now = std::chrono::steady_clock::now();
    t_upload_ms[i] += std::chrono::duration<double, std::milli>(now - tk).count();
    tk = now;
  };
  if (!dict_host.empty()) DFGPU_HIP(hipMemcpyAsync(d_dict->ptr, dict_host.data(), dict_host.size(), hipMemcpyHostToDevice, st));
  lap(0);
  if (!P.staging.empty()) DFGPU_HIP(hipMemcpyAsync(d_stage->ptr, P.staging.data(), P.staging.size(), hipMemcpyHostToDevice, st));
  lap(1);
  if (!P.pages.empty()) DFGPU_HIP(hipMemcpyAsync(d_pages->ptr, P.pages.data(), P.pages.size() * sizeof(PageDesc), hipMemcpyHostToDevice, st));
  lap(2);
  if (!P.runs.empty()) DFGPU_HIP(hipMemcpyAsync(d_runs->ptr, P.runs.data(), P.runs.size() * sizeof(Run), hipMemcpyHostToDevice, st));
  lap(3);
  PqArgs a{};
  a.pages = d_pages->as<PageDesc>();
  a.runs = d_runs->as<Run>();
  a.staging = d_stage->as<uint8_t>();
  a.dict = d_dict->ptr;
  a.n_pages = (int32_t)P.pages.size();
  a.n_runs = (int32_t)P.runs.size();
  a.dict_count = std::max(P.dict_count, 1);
  a.phys_w = std::max(phys_width(col), 1);
  a.big_endian = col.physical_type == DFGPU_PARQUET_FIXED_LEN_BYTE_ARRAY;
  a.sign_extend = is_signed_type(f.type) && f.type != DFGPU_FLOAT64;
  const bool has_nulls = !P.validity.empty();
  BufPtr dense;
  void* target = c.data->ptr;
  if (has_nulls) {
    dense = make_buf((size_t)std::max<int64_t>(P.values, 1) * out_w + 16);
    target = dense->ptr;
  }
  {
    ProfileScope ps("parquet_decode", (int64_t)P.staging.size() + P.values * out_w);
    switch (out_w) {
      case 16: launch_decode<i128>(a, P.values, target); break;
      case 8: launch_decode<uint64_t>(a, P.values, target); break;
      case 4: launch_decode<uint32_t>(a, P.values, target); break;
      default: launch_decode<uint8_t>(a, P.values, target); break;
    }
  }
  if (has_nulls) {
    const size_t bb = bitmap_bytes(P.rows);
    c.validity = make_buf(bb);
    c.null_count = P.rows - P.values;
    DFGPU_HIP(hipMemcpyAsync(c.validity->ptr, P.validity.data(), bb, hipMemcpyHostToDevice, st));
    BufPtr prefix = make_buf((size_t)((P.rows + 63) / 64 + 1) * 8);
    scan_mask_popcounts(c.validity->as<uint64_t>(), nullptr, P.rows, prefix->as<uint64_t>());
    switch (out_w) {
      case 16: launch_expand<i128>(dense->ptr, c.valid_words(), prefix->as<uint64_t>(), P.rows, c.data->ptr); break;
      case 8: launch_expand<uint64_t>(dense->ptr, c.valid_words(), prefix->as<uint64_t>(), P.rows, c.data->ptr); break;
      case 4: launch_expand<uint32_t>(dense->ptr, c.valid_words(), prefix->as<uint64_t>(), P.rows, c.data->ptr); break;
      default: launch_expand<uint8_t>(dense->ptr, c.valid_words(), prefix->as<uint64_t>(), P.rows, c.data->ptr); break;
    }
  }
  lap(4);
  if (keep) *keep = sources;   // the copies' sources: the caller's until the stream has passed them
  else DFGPU_HIP(hipStreamSynchronize(st));
  return c;
}

}  // namespace
}  // namespace dfgpu

using namespace dfgpu;

extern "C" {

int dfgpu_parquet_inspect_chunk(const uint8_t* chunk, int64_t chunk_bytes, const dfgpu_parquet_column* column, dfgpu_parquet_chunk_info* out) {
  return guarded([&] {
    DFGPU_CHECK(chunk && column && out && chunk_bytes >= 0, "null argument");
    ChunkPlan P = plan_chunk(chunk, chunk_bytes, *column);
    *out = P.info;
    out->dictionary_values = P.dict_count;
  });
}

int dfgpu_parquet_decode_chunk(const uint8_t* chunk, int64_t chunk_bytes, const dfgpu_parquet_column* column, dfgpu_table_t* out) {
  return guarded([&] {
    require_init();
    DFGPU_CHECK(chunk && column && out && chunk_bytes >= 0, "null argument");
    auto t = std::make_unique<Table>();
    t->cols.push_back(decode_chunk(chunk, chunk_bytes, *column));
    t->nrows = t->cols[0].length;
    *out = wrap(t.release());
  });
}

}  // extern "C"

// ------------------------------------------------------------------------------ a scan's chunks in one call
// One worker per host thread asked for: a chunk's host half (page headers, decompression, levels, run headers) runs on that
// thread, its device half on that thread's stream — the same shape as a scan node decoding from its partition threads, without
// a call through the boundary (and, for the Python face, without the interpreter) per chunk.
// A unit of host work = a column chunk.  (Groups of about 1 MiB of a big chunk's data pages as the unit — the dictionary page riding
// along where needed, the groups' columns put together afterwards — were measured: 10.5-22 ms against 8.7-8.8 ms for ZSTD at SF1
// on 16 threads, 10.4 against 6.9-7.5 uncompressed: more launches, more concatenation, the dictionary parsed per group.  Dropped.)
struct ScanTask {
  int64_t chunk = 0;            // index into the caller's chunk list
};
struct ScanShared {
  const dfgpu_parquet_chunk* chunks;
  std::vector<ScanTask> tasks;
  dfgpu_cache_t cache;
  std::vector<std::unique_ptr<Table>>* out;   // per task
  std::vector<int64_t> order;   // largest task first: the slowest worker ends as early as a greedy schedule lets it
  std::atomic<int64_t> next{0};
  std::mutex mu;
  std::string error;
  dfgpu_metrics metrics{};   // the workers' metrics, folded into the caller's
  double plan_ms = 0, worker_ms_max = 0, drain_ms_max = 0, settle_ms = 0, upload_ms[6] = {0, 0, 0, 0, 0, 0};   // DFGPU_TRACE_SCAN
};
// The scan's host threads live as long as the process: a thread made for one scan starts with a cold allocator arena and a cold
// stack — its first megabytes of page buffers are page-faulted in, 16-32 threads at a time on one address space — and a HIP stream
// to find; measured as host halves that took 3-7 times their single-threaded time.  Workers sleep on a condition variable between
// scans.
struct ScanPool {
  std::mutex mu;
  std::condition_variable wake, done_cv;
  std::vector<std::thread> workers;
  std::function<void()> job;
  int64_t generation = 0;
  int to_start = 0, running = 0;
  void loop() {
    int64_t seen = 0;
    for (;;) {
      std::function<void()> mine;
      {
        std::unique_lock<std::mutex> lk(mu);
        wake.wait(lk, [&] { return generation != seen && to_start > 0; });
        seen = generation;
        to_start--;
        mine = job;
      }
      mine();
      {
        std::lock_guard<std::mutex> lk(mu);
        if (--running == 0) done_cv.notify_all();
      }
    }
  }
  // fn() on `n` pool threads at once; returns when all of them have returned.  One scan at a time uses the pool.
  void run(int n, std::function<void()> fn) {
    std::unique_lock<std::mutex> lk(mu);
    done_cv.wait(lk, [&] { return running == 0; });
    while ((int)workers.size() < n) {
      workers.emplace_back([this] { loop(); });
      workers.back().detach();
    }
    job = std::move(fn);
    generation++;
    to_start = n;
    running = n;
    wake.notify_all();
    done_cv.wait(lk, [&] { return running == 0; });
  }
};
static ScanPool& scan_pool() {
  static ScanPool* p = new ScanPool();   // never destroyed: its threads outlive static destructors
  return *p;
}

// one chunk, launched: `keep` holds what its uploads read from (null when decoded with a wait inside), `to_cache` says whether the
// result is to be put into the cache once complete
static std::unique_ptr<Table> scan_one_task(const ScanShared& sh, const ScanTask& task, std::shared_ptr<void>& keep, std::function<void(Column&)>& fin, bool& to_cache) {
  to_cache = false;
  keep.reset();
  fin = nullptr;
  const dfgpu_parquet_chunk& ch = sh.chunks[task.chunk];
  DFGPU_CHECK(ch.bytes && ch.n_bytes >= 0, "parquet: null chunk");
  auto t = std::make_unique<Table>();
  dfgpu_parquet_column col = ch.column;
  try {
    t->cols.push_back(decode_chunk(ch.bytes, ch.n_bytes, col, &keep, &fin));
  } catch (const Error& e) {
    // a string chunk whose writer fell back to PLAIN pages: Utf8 bytes instead of dictionary indices (the caller brings the
    // chunks of the column to one kind)
    if (col.physical_type != DFGPU_PARQUET_BYTE_ARRAY || col.field.type == DFGPU_UTF8 || std::string(e.what()).find("read the column as Utf8") == std::string::npos) throw;
    col.field.type = DFGPU_UTF8;
    t->cols.push_back(decode_chunk(ch.bytes, ch.n_bytes, col, &keep, &fin));
  }
  t->nrows = t->cols[0].length;
  t->device = current_device();
  to_cache = sh.cache && ch.cache_key && ch.cache_key_bytes > 0;
  return t;
}
static void scan_worker(ScanShared& sh, int device, bool own_thread) {
  const auto t_w0 = std::chrono::steady_clock::now();
  t_plan_ms = 0;
  if (own_thread) {
    use_device(device);
    thread_metrics() = dfgpu_metrics{};
  }
  // a few tasks in flight per worker (parquet.in_flight, 4): while task i crosses PCIe and is decoded — on the device path that includes its
  // pages' decompression, one wave per page — the host halves of the next ones run; task i's host
  // buffers go (and a whole chunk enters the cache: complete, other threads may take it at once) when its event has passed
  struct InFlight {
    std::shared_ptr<void> keep;
    std::function<void(Column&)> fin;   // (a chunk decoded on the device: its pages' error flags and NULL count, once the stream is past it)
    int64_t k = -1;
    bool to_cache = false;
    hipEvent_t ev = nullptr;
  };
  constexpr int MAX_FLIGHT = 8;
  InFlight fl[MAX_FLIGHT];
  const int depth = (int)std::max<int64_t>(2, std::min<int64_t>(MAX_FLIGHT, option_int("parquet.in_flight", 4)));
  int cur = 0;
  double settle_ms = 0;
  for (double& x : t_upload_ms) x = 0;
  auto settle = [&](InFlight& f) {
    if (f.k < 0) return;
    const auto ts = std::chrono::steady_clock::now();
    if (f.ev) (void)hipEventSynchronize(f.ev);
    settle_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ts).count();
    f.keep.reset();
    if (f.fin) {
      auto fin = std::move(f.fin);
      f.fin = nullptr;
      fin((*sh.out)[(size_t)f.k]->cols[0]);
    }
    if (f.to_cache) {
      const dfgpu_parquet_chunk& ch = sh.chunks[sh.tasks[(size_t)f.k].chunk];
      if (dfgpu_cache_put(sh.cache, ch.cache_key, ch.cache_key_bytes, wrap_quiet((*sh.out)[(size_t)f.k].get())) != 0) throw Error(dfgpu_last_error());
    }
    f.k = -1;
  };
  try {
    // blocking events: a worker that waits for its chunk to cross PCIe sleeps instead of spinning — the host cores (16 CPUs' worth
    // of quota on the benchmark boxes) belong to the workers that are decompressing
    for (int i = 0; i < depth; i++) DFGPU_HIP(hipEventCreateWithFlags(&fl[i].ev, hipEventDisableTiming | hipEventBlockingSync));
    for (;;) {
      const int64_t at = sh.next.fetch_add(1);
      if (at >= (int64_t)sh.tasks.size()) break;
      const int64_t k = sh.order[(size_t)at];
      {
        std::lock_guard<std::mutex> lk(sh.mu);
        if (!sh.error.empty()) break;
      }
      InFlight& f = fl[cur];
      settle(f);   // (the task before the previous one)
      (*sh.out)[(size_t)k] = scan_one_task(sh, sh.tasks[(size_t)k], f.keep, f.fin, f.to_cache);
      f.k = k;
      DFGPU_HIP(hipEventRecord(f.ev, rt().stream));
      cur = (cur + 1) % depth;
    }
    for (int i = 0; i < depth; i++) settle(fl[(cur + i) % depth]);   // (oldest first)
  } catch (const std::exception& e) {
    (void)hipStreamSynchronize(rt().stream);   // nothing may still read the host buffers that go with `fl`
    std::lock_guard<std::mutex> lk(sh.mu);
    if (sh.error.empty()) sh.error = e.what();
  }
  for (int i = 0; i < MAX_FLIGHT; i++)
    if (fl[i].ev) (void)hipEventDestroy(fl[i].ev);
  const auto t_w1 = std::chrono::steady_clock::now();
  if (own_thread) call_epilogue();   // drains this thread's stream; the blocks it freed join the pool
  {
    const auto t_w2 = std::chrono::steady_clock::now();
    std::lock_guard<std::mutex> lk(sh.mu);
    sh.plan_ms += t_plan_ms;
    sh.settle_ms += settle_ms;
    for (int i = 0; i < 6; i++) sh.upload_ms[i] += t_upload_ms[i];
    sh.worker_ms_max = std::max(sh.worker_ms_max, std::chrono::duration<double, std::milli>(t_w1 - t_w0).count());
    sh.drain_ms_max = std::max(sh.drain_ms_max, std::chrono::duration<double, std::milli>(t_w2 - t_w1).count());
  }
  if (own_thread) {
    std::lock_guard<std::mutex> lk(sh.mu);
    const dfgpu_metrics& m = thread_metrics();
    sh.metrics.h2d_bytes += m.h2d_bytes;
    sh.metrics.d2h_bytes += m.d2h_bytes;
    sh.metrics.hbm_bytes_algorithmic += m.hbm_bytes_algorithmic;
  }
}

extern "C" int dfgpu_parquet_read_chunks(const dfgpu_parquet_chunk* chunks, int32_t n_row_groups, int32_t n_columns, int32_t threads, dfgpu_cache_t cache,
                                         dfgpu_table_t* out, int64_t* chunks_from_cache) {
  return guarded([&] {
    require_init();
    DFGPU_CHECK(chunks && out && n_row_groups >= 1 && n_columns >= 1, "bad argument");
    const int64_t n = (int64_t)n_row_groups * n_columns;
    std::vector<std::unique_ptr<Table>> parts((size_t)n);
    ScanShared sh;
    sh.chunks = chunks;
    sh.cache = cache;
    const int device = current_device();
    const bool trace = trace_on("scan");
    const int64_t dev_allocs0 = rt().driver_allocs.load(), dev_ns0 = rt().driver_alloc_ns.load(), pin_allocs0 = g_pinned_driver_allocs.load(), pin_ns0 = g_pinned_driver_ns.load();
    const auto t0 = std::chrono::steady_clock::now();
    auto since = [&](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); };
    // chunks the cache holds are taken from it; the others become tasks
    int64_t from_cache = 0;
    for (int64_t k = 0; k < n; k++) {
      const dfgpu_parquet_chunk& ch = chunks[k];
      if (cache && ch.cache_key && ch.cache_key_bytes > 0) {
        dfgpu_table_t got = nullptr;
        if (dfgpu_cache_get(cache, ch.cache_key, ch.cache_key_bytes, &got) != 0) throw Error(dfgpu_last_error());
        if (got) {
          parts[(size_t)k].reset(unwrap_quiet(got));
          from_cache++;
          continue;
        }
      }
      ScanTask task;
      task.chunk = k;
      sh.tasks.push_back(task);
    }
    const int64_t n_tasks = (int64_t)sh.tasks.size();
    std::vector<std::unique_ptr<Table>> done((size_t)n_tasks);
    sh.out = &done;
    sh.order.resize((size_t)n_tasks);
    std::iota(sh.order.begin(), sh.order.end(), (int64_t)0);
    std::stable_sort(sh.order.begin(), sh.order.end(), [&](int64_t a, int64_t b) { return chunks[sh.tasks[(size_t)a].chunk].n_bytes > chunks[sh.tasks[(size_t)b].chunk].n_bytes; });
    const int T = (int)std::max<int64_t>(1, std::min<int64_t>(threads, n_tasks));
    if (n_tasks == 0) {
      // everything came from the cache
    } else if (T <= 1) {
      scan_worker(sh, device, false);
    } else {
      scan_pool().run(T, [&] { scan_worker(sh, device, true); });
      dfgpu_metrics& m = thread_metrics();
      m.h2d_bytes += sh.metrics.h2d_bytes;
      m.d2h_bytes += sh.metrics.d2h_bytes;
      m.hbm_bytes_algorithmic += sh.metrics.hbm_bytes_algorithmic;
    }
    if (!sh.error.empty()) throw Error(sh.error);
    if (chunks_from_cache) *chunks_from_cache = from_cache;
    DFGPU_HIP(hipStreamSynchronize(rt().stream));
    for (int64_t t = 0; t < n_tasks; t++) parts[(size_t)sh.tasks[(size_t)t].chunk] = std::move(done[(size_t)t]);
    const double workers_ms = since(t0);
    const auto t1 = std::chrono::steady_clock::now();
    // a string column whose chunks came out in both kinds (dictionary indices here, Utf8 bytes there) becomes Utf8 everywhere
    for (int j = 0; j < n_columns; j++) {
      bool any_utf8 = false, any_dict = false;
      for (int g = 0; g < n_row_groups; g++) {
        const Column& c = parts[(size_t)g * n_columns + j]->cols[0];
        (c.field.type == DFGPU_UTF8 ? any_utf8 : any_dict) = true;
      }
      if (!(any_utf8 && any_dict)) continue;
      for (int g = 0; g < n_row_groups; g++) {
        Column& c = parts[(size_t)g * n_columns + j]->cols[0];
        if (c.field.type != DFGPU_UTF8) {
          auto fresh = std::make_unique<Table>(*parts[(size_t)g * n_columns + j]);   // (the cached chunk keeps its own form)
          fresh->cols[0] = dictionary_decode(c);
          parts[(size_t)g * n_columns + j] = std::move(fresh);
        }
      }
    }
    // row groups side by side, then one below the other
    std::vector<std::unique_ptr<Table>> groups;
    for (int g = 0; g < n_row_groups; g++) {
      auto t = std::make_unique<Table>();
      t->device = device;
      t->nrows = parts[(size_t)g * n_columns]->nrows;
      for (int j = 0; j < n_columns; j++) {
        Table& p = *parts[(size_t)g * n_columns + j];
        DFGPU_CHECK(p.nrows == t->nrows, "parquet: the column chunks of a row group differ in row count");
        t->cols.push_back(p.cols[0]);
      }
      groups.push_back(std::move(t));
    }
    if (n_row_groups == 1) {
      *out = wrap(groups[0].release());
      return;
    }
    std::vector<dfgpu_table_t> hs;
    for (auto& g : groups) hs.push_back(wrap_quiet(g.get()));
    dfgpu_table_t whole = nullptr;
    if (dfgpu_table_concat(hs.data(), (int)hs.size(), &whole) != 0) throw Error(dfgpu_last_error());
    *out = whole;
    if (trace) {
      DFGPU_HIP(hipStreamSynchronize(rt().stream));
      fprintf(stderr, "[scan] %lld chunks (%lld to decode) on %d threads: workers %.3f ms (slowest %.3f + drain %.3f; host halves %.3f ms in total), assembly %.3f ms\n",
              (long long)n, (long long)n_tasks, T, workers_ms, sh.worker_ms_max, sh.drain_ms_max, sh.plan_ms, since(t1));
      fprintf(stderr, "[scan]   summed over workers: waits for chunks in flight %.3f ms; uploads: dictionary %.3f, staging %.3f, pages %.3f, runs %.3f, kernels + validity %.3f ms\n",
              sh.settle_ms, sh.upload_ms[0], sh.upload_ms[1], sh.upload_ms[2], sh.upload_ms[3], sh.upload_ms[4]);
      fprintf(stderr, "[scan]   pool misses: %lld hipMalloc (%.3f ms), %lld hipHostMalloc (%.3f ms)\n", (long long)(rt().driver_allocs.load() - dev_allocs0),
              (double)(rt().driver_alloc_ns.load() - dev_ns0) * 1e-6, (long long)(g_pinned_driver_allocs.load() - pin_allocs0), (double)(g_pinned_driver_ns.load() - pin_ns0) * 1e-6);
    }
  });
}
