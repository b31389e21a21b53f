// ipc.hip — scan -> device for Arrow IPC files and streams, below the C ABI (the reference's
// `DataSourceExec` over an `ArrowSource`, datasource-arrow/src/source.rs:260-330: arrow-ipc's FileReader / StreamReader).
//
// An IPC file already holds Arrow buffers, so there is nothing to decode: the host walks the encapsulated messages (a
// hand-written reader of the few flatbuffers tables the format uses: Message, Schema, Field, Int / FloatingPoint / Decimal / Date,
// DictionaryEncoding, RecordBatch, BodyCompression — format/Message.fbs, Schema.fbs, File.fbs of apache/arrow; the flatbuffers
// wire format is vtable-relative offsets, nothing else is needed), lays Arrow C Data structs over the file's own bytes where the
// page cache maps them, and hands each record batch — the projected columns only — to the import path of table.hip (pinned,
// side-stream H2D copies).  Per-buffer compression (ZSTD; LZ4_FRAME when liblz4 is present) is undone on the host first, as the
// reference's reader does.  Dictionary batches are kept and attached to the columns that refer to them.  Flat columns of the types
// the device knows (Int32/64, UInt8/32/64, Float64, Date32, Decimal128, Boolean, Utf8 / LargeUtf8 / Utf8View, dictionary-encoded
// Utf8); anything else is an error and the rule keeps the CPU scan.
#include <dlfcn.h>

#include <cstdio>
#include <unordered_map>

#include "internal.hpp"

namespace dfgpu {
namespace {

// ------------------------------------------------------------------------------------------------ flatbuffers, read-only
struct Fb {
  const uint8_t* base = nullptr;
  const uint8_t* end = nullptr;
  template <typename T>
  T rd(const uint8_t* p) const {
    DFGPU_CHECK(p >= base && p + sizeof(T) <= end, "arrow ipc: metadata points outside its message");
    T v;
    std::memcpy(&v, p, sizeof(T));
    return v;
  }
  const uint8_t* root() const { return base + rd<uint32_t>(base); }
  // position of field `i` inside table `t`, or null when absent (default value applies)
  const uint8_t* field(const uint8_t* t, int i) const {
    const uint8_t* vt = t - rd<int32_t>(t);
    const uint16_t vsize = rd<uint16_t>(vt);
    if (4 + 2 * i + 2 > vsize) return nullptr;
    const uint16_t off = rd<uint16_t>(vt + 4 + 2 * i);
    return off ? t + off : nullptr;
  }
  template <typename T>
  T scalar(const uint8_t* t, int i, T dflt) const {
    const uint8_t* p = field(t, i);
    return p ? rd<T>(p) : dflt;
  }
  const uint8_t* indirect(const uint8_t* p) const { return p + rd<uint32_t>(p); }   // table / string / vector behind an offset
  const uint8_t* table(const uint8_t* t, int i) const {
    const uint8_t* p = field(t, i);
    return p ? indirect(p) : nullptr;
  }
  std::string str(const uint8_t* t, int i) const {
    const uint8_t* p = field(t, i);
    if (!p) return "";
    const uint8_t* s = indirect(p);
    const uint32_t n = rd<uint32_t>(s);
    DFGPU_CHECK(s + 4 + n <= end, "arrow ipc: string overruns its message");
    return std::string((const char*)s + 4, n);
  }
  // vector: element count and first element
  const uint8_t* vec(const uint8_t* t, int i, uint32_t& n) const {
    n = 0;
    const uint8_t* p = field(t, i);
    if (!p) return nullptr;
    const uint8_t* v = indirect(p);
    n = rd<uint32_t>(v);
    return v + 4;
  }
};

enum { TY_INT = 2, TY_FLOAT = 3, TY_UTF8 = 5, TY_BOOL = 6, TY_DECIMAL = 7, TY_DATE = 8, TY_LARGE_UTF8 = 20, TY_UTF8_VIEW = 24 };
enum { HDR_SCHEMA = 1, HDR_DICTIONARY = 2, HDR_BATCH = 3 };

struct IpcField {
  std::string name, format;     // Arrow C Data format string of the column as it is stored (indices for a dictionary column)
  bool nullable = true;
  int n_buffers = 2;            // buffers this field takes from the RecordBatch's list (a Utf8View's data buffers come on top)
  bool view = false;
  int64_t dict_id = -1;         // >= 0: dictionary-encoded; `value_format` = the dictionary's format
  std::string value_format;
};
struct IpcBatch {
  const uint8_t* meta = nullptr;   // the RecordBatch flatbuffers table's Message
  int64_t meta_len = 0;
  const uint8_t* body = nullptr;
  int64_t body_len = 0;
  int64_t rows = 0;
};
struct IpcFile {
  const uint8_t* data = nullptr;
  int64_t nbytes = 0;
  std::vector<IpcField> fields;
  std::vector<IpcBatch> batches;
  std::vector<std::pair<int64_t, IpcBatch>> dictionaries;   // (dictionary id, its batch), in file order
  bool is_file = false;
};

std::string int_format(int bits, bool is_signed, const std::string& name) {
  if (bits == 8 && !is_signed) return "C";
  if (bits == 32) return is_signed ? "i" : "I";
  if (bits == 64) return is_signed ? "l" : "L";
  if (bits == 8) return "c";
  if (bits == 16) return is_signed ? "s" : "S";
  throw Error("arrow ipc: column '" + name + "': integer width " + std::to_string(bits));
}

IpcField parse_field(const Fb& fb, const uint8_t* f) {
  IpcField out;
  out.name = fb.str(f, 0);
  out.nullable = fb.scalar<uint8_t>(f, 1, 0) != 0;
  const int ty = fb.scalar<uint8_t>(f, 2, 0);
  const uint8_t* t = fb.table(f, 3);
  uint32_t n_children = 0;
  (void)fb.vec(f, 5, n_children);
  DFGPU_CHECK(n_children == 0, "arrow ipc: column '" + out.name + "' is nested (struct / list / map): not supported on the GPU scan path");
  std::string fmt;
  switch (ty) {
    case TY_INT: fmt = int_format(t ? fb.scalar<int32_t>(t, 0, 0) : 0, t ? fb.scalar<uint8_t>(t, 1, 0) != 0 : false, out.name); break;
    case TY_FLOAT: {
      const int prec = t ? fb.scalar<int16_t>(t, 0, 0) : 0;
      DFGPU_CHECK(prec == 2, "arrow ipc: column '" + out.name + "': only Float64 is supported on the GPU scan path");
      fmt = "g";
      break;
    }
    case TY_BOOL: fmt = "b"; break;
    case TY_DECIMAL: {
      const int p = t ? fb.scalar<int32_t>(t, 0, 0) : 0, s = t ? fb.scalar<int32_t>(t, 1, 0) : 0, bits = t ? fb.scalar<int32_t>(t, 2, 128) : 128;
      DFGPU_CHECK(bits == 128, "arrow ipc: column '" + out.name + "': only Decimal128 is supported on the GPU scan path");
      fmt = "d:" + std::to_string(p) + "," + std::to_string(s);
      break;
    }
    case TY_DATE: {
      const int unit = t ? fb.scalar<int16_t>(t, 0, 1) : 1;   // DateUnit: DAY = 0, MILLISECOND = 1 (the default)
      DFGPU_CHECK(unit == 0, "arrow ipc: column '" + out.name + "': only Date32 is supported on the GPU scan path");
      fmt = "tdD";
      break;
    }
    case TY_UTF8: fmt = "u"; out.n_buffers = 3; break;
    case TY_LARGE_UTF8: fmt = "U"; out.n_buffers = 3; break;
    case TY_UTF8_VIEW: fmt = "vu"; out.n_buffers = 2; out.view = true; break;
    default: throw Error("arrow ipc: column '" + out.name + "': type id " + std::to_string(ty) + " is not supported on the GPU scan path");
  }
  if (const uint8_t* d = fb.table(f, 4)) {   // DictionaryEncoding {id, indexType: Int, isOrdered}
    out.dict_id = fb.scalar<int64_t>(d, 0, 0);
    out.value_format = fmt;
    DFGPU_CHECK(fmt == "u" || fmt == "U", "arrow ipc: column '" + out.name + "': only string dictionaries are supported on the GPU scan path");
    const uint8_t* it = fb.table(d, 1);
    out.format = it ? int_format(fb.scalar<int32_t>(it, 0, 32), fb.scalar<uint8_t>(it, 1, 1) != 0, out.name) : "i";
    out.n_buffers = 2;
  } else {
    out.format = fmt;
  }
  return out;
}

std::unique_ptr<IpcFile> open_ipc(const uint8_t* data, int64_t nbytes) {
  auto f = std::make_unique<IpcFile>();
  f->data = data;
  f->nbytes = nbytes;
  const uint8_t* p = data;
  const uint8_t* end = data + nbytes;
  if (nbytes >= 12 && std::memcmp(p, "ARROW1", 6) == 0) {
    f->is_file = true;
    p += 8;   // magic + 2 bytes of padding; the body of a file is a stream, the footer behind it only indexes it
  }
  bool have_schema = false;
  for (;;) {
    if (end - p < 8) break;
    uint32_t cont, mlen;
    std::memcpy(&cont, p, 4);
    if (cont == 0xFFFFFFFFu) {
      std::memcpy(&mlen, p + 4, 4);
      p += 8;
    } else {   // pre-0.15 framing: the length alone
      mlen = cont;
      p += 4;
    }
    if (mlen == 0) break;   // end of stream (in a file: the footer follows)
    DFGPU_CHECK((int64_t)mlen <= end - p, "arrow ipc: message metadata overruns the file");
    Fb fb{p, p + mlen};
    const uint8_t* msg = fb.root();
    const int hdr = fb.scalar<uint8_t>(msg, 1, 0);
    const uint8_t* h = fb.table(msg, 2);
    const int64_t body_len = fb.scalar<int64_t>(msg, 3, 0);
    const uint8_t* body = p + mlen;
    DFGPU_CHECK(body_len >= 0 && body_len <= end - body, "arrow ipc: message body overruns the file");
    if (hdr == HDR_SCHEMA) {
      DFGPU_CHECK(h != nullptr && !have_schema, "arrow ipc: malformed or repeated schema message");
      DFGPU_CHECK(fb.scalar<int16_t>(h, 0, 0) == 0, "arrow ipc: big-endian files are not supported");
      uint32_t n = 0;
      const uint8_t* v = fb.vec(h, 1, n);
      for (uint32_t i = 0; i < n; i++) f->fields.push_back(parse_field(fb, fb.indirect(v + 4 * i)));
      have_schema = true;
    } else if (hdr == HDR_BATCH || hdr == HDR_DICTIONARY) {
      DFGPU_CHECK(have_schema && h != nullptr, "arrow ipc: a batch before the schema");
      IpcBatch b;
      b.meta = p;
      b.meta_len = mlen;
      b.body = body;
      b.body_len = body_len;
      const uint8_t* rb = hdr == HDR_BATCH ? h : fb.table(h, 1);
      DFGPU_CHECK(rb != nullptr, "arrow ipc: dictionary batch without data");
      b.rows = fb.scalar<int64_t>(rb, 0, 0);
      if (hdr == HDR_BATCH) {
        f->batches.push_back(b);
      } else {
        DFGPU_CHECK(fb.scalar<uint8_t>(h, 2, 0) == 0, "arrow ipc: delta dictionaries are not supported on the GPU scan path");
        f->dictionaries.emplace_back(fb.scalar<int64_t>(h, 0, 0), b);
      }
    }
    p = body + body_len;
  }
  DFGPU_CHECK(have_schema, "arrow ipc: no schema message (not an Arrow IPC file or stream)");
  return f;
}

// ------------------------------------------------------------------------------------------------ batch -> Arrow C Data structs
struct Owned {                      // what the structs built below point into, released with the root array
  std::vector<std::vector<uint8_t>> decompressed;
  std::vector<std::unique_ptr<ArrowArray>> arrays;
  std::vector<std::unique_ptr<ArrowSchema>> schemas;
  std::vector<std::vector<const void*>> buffer_lists;
  std::vector<std::vector<ArrowArray*>> child_lists;
  std::vector<std::vector<ArrowSchema*>> schema_child_lists;
  std::vector<std::string> strings;
};
void release_owned_array(ArrowArray* a) {
  if (!a || !a->release) return;
  delete (Owned*)a->private_data;   // children are plain members of Owned: nothing of theirs to release
  a->release = nullptr;
}
void release_plain_schema(ArrowSchema* s) {
  if (s) s->release = nullptr;      // owned by the array's Owned block
}

using ZstdFn = size_t (*)(void*, size_t, const void*, size_t);
using ZstdErrFn = unsigned (*)(size_t);
void ipc_decompress(int codec, const uint8_t* src, int64_t n, uint8_t* dst, int64_t dst_len) {
  if (codec == 1) {   // ZSTD
    static void* h = dlopen("libzstd.so.1", RTLD_NOW | RTLD_LOCAL);
    DFGPU_CHECK(h != nullptr, "arrow ipc: ZSTD-compressed buffers need libzstd.so.1");
    static ZstdFn dec = (ZstdFn)dlsym(h, "ZSTD_decompress");
    static ZstdErrFn iserr = (ZstdErrFn)dlsym(h, "ZSTD_isError");
    const size_t r = dec(dst, (size_t)dst_len, src, (size_t)n);
    DFGPU_CHECK(!iserr(r) && (int64_t)r == dst_len, "arrow ipc: corrupt ZSTD buffer");
    return;
  }
  // LZ4_FRAME
  static void* h = dlopen("liblz4.so.1", RTLD_NOW | RTLD_LOCAL);
  DFGPU_CHECK(h != nullptr, "arrow ipc: LZ4_FRAME-compressed buffers need liblz4.so.1 (not present): the CPU scan stays");
  using CreateFn = size_t (*)(void**, unsigned);
  using FreeFn = size_t (*)(void*);
  using DecFn = size_t (*)(void*, void*, size_t*, const void*, size_t*, const void*);
  static CreateFn create = (CreateFn)dlsym(h, "LZ4F_createDecompressionContext");
  static FreeFn destroy = (FreeFn)dlsym(h, "LZ4F_freeDecompressionContext");
  static DecFn dec = (DecFn)dlsym(h, "LZ4F_decompress");
  static ZstdErrFn iserr = (ZstdErrFn)dlsym(h, "LZ4F_isError");
  DFGPU_CHECK(create && destroy && dec && iserr, "arrow ipc: liblz4 lacks the frame API");
  void* ctx = nullptr;
  DFGPU_CHECK(!iserr(create(&ctx, 100)), "arrow ipc: LZ4F context");
  size_t out_pos = 0, in_pos = 0;
  while (in_pos < (size_t)n && out_pos < (size_t)dst_len) {
    size_t out_n = (size_t)dst_len - out_pos, in_n = (size_t)n - in_pos;
    const size_t r = dec(ctx, dst + out_pos, &out_n, src + in_pos, &in_n, nullptr);
    if (iserr(r)) {
      destroy(ctx);
      throw Error("arrow ipc: corrupt LZ4 frame");
    }
    out_pos += out_n;
    in_pos += in_n;
    if (r == 0) break;
  }
  destroy(ctx);
  DFGPU_CHECK((int64_t)out_pos == dst_len, "arrow ipc: LZ4 frame shorter than its declared length");
}

struct BatchCursor {
  const IpcBatch* b;
  Fb fb;
  const uint8_t* rb;            // RecordBatch table
  const uint8_t* nodes;         // FieldNode structs {length, null_count}
  const uint8_t* buffers;       // Buffer structs {offset, length}
  uint32_t n_nodes = 0, n_buffers = 0;
  const uint8_t* variadic = nullptr;
  uint32_t n_variadic = 0;
  int codec = -1;               // -1 = uncompressed
  uint32_t node_at = 0, buf_at = 0, var_at = 0;
};
BatchCursor cursor_of(const IpcBatch& b, bool dictionary) {
  BatchCursor c{&b, Fb{b.meta, b.meta + b.meta_len}, nullptr, nullptr, nullptr};
  const uint8_t* msg = c.fb.root();
  const uint8_t* h = c.fb.table(msg, 2);
  c.rb = dictionary ? c.fb.table(h, 1) : h;
  c.nodes = c.fb.vec(c.rb, 1, c.n_nodes);
  c.buffers = c.fb.vec(c.rb, 2, c.n_buffers);
  if (const uint8_t* comp = c.fb.table(c.rb, 3)) {
    c.codec = c.fb.scalar<int8_t>(comp, 0, 0);
    DFGPU_CHECK(c.fb.scalar<int8_t>(comp, 1, 0) == 0, "arrow ipc: unknown body compression method");
  }
  c.variadic = c.fb.vec(c.rb, 4, c.n_variadic);
  return c;
}
// the next buffer of the batch as host bytes (decompressed into `own` when the body is compressed)
const void* next_buffer(BatchCursor& c, Owned& own, int64_t* out_len = nullptr) {
  DFGPU_CHECK(c.buf_at < c.n_buffers, "arrow ipc: a record batch lists fewer buffers than its schema needs");
  const uint8_t* e = c.buffers + (size_t)c.buf_at * 16;
  c.buf_at++;
  const int64_t off = c.fb.rd<int64_t>(e), len = c.fb.rd<int64_t>(e + 8);
  DFGPU_CHECK(off >= 0 && len >= 0 && off + len <= c.b->body_len, "arrow ipc: a buffer lies outside its message body");
  if (out_len) *out_len = len;
  if (len == 0) return nullptr;
  const uint8_t* p = c.b->body + off;
  if (c.codec < 0) return p;
  DFGPU_CHECK(len >= 8, "arrow ipc: compressed buffer without its length prefix");
  int64_t raw;
  std::memcpy(&raw, p, 8);
  if (raw == -1) {   // stored uncompressed
    if (out_len) *out_len = len - 8;
    return p + 8;
  }
  DFGPU_CHECK(raw >= 0 && raw < ((int64_t)1 << 40), "arrow ipc: bad uncompressed buffer length");
  own.decompressed.emplace_back((size_t)raw + 64);
  if (raw) ipc_decompress(c.codec, p + 8, len - 8, own.decompressed.back().data(), raw);
  if (out_len) *out_len = raw;
  return own.decompressed.back().data();
}

// bytes per value of a fixed-width Arrow C format string; 0 = bit-packed (Boolean); -1 = not fixed width
int format_width(const std::string& fmt) {
  if (fmt == "b") return 0;
  if (fmt == "c" || fmt == "C") return 1;
  if (fmt == "s" || fmt == "S" || fmt == "e") return 2;
  if (fmt == "i" || fmt == "I" || fmt == "f" || fmt == "tdD") return 4;
  if (fmt == "l" || fmt == "L" || fmt == "g" || fmt == "tdm" || fmt.rfind("ts", 0) == 0 || fmt.rfind("tt", 0) == 0 || fmt.rfind("tD", 0) == 0) return 8;
  if (fmt.rfind("d:", 0) == 0) return fmt.find(",256") != std::string::npos ? 32 : 16;
  return -1;
}
// Everything dfgpu_table_import is about to read through these pointers must lie INSIDE the buffers the file really holds: the node's
// row count against every buffer's (decompressed) length, the last string offset against the data buffer, view references against
// their data buffers, dictionary indices against the dictionary.  arrow-ipc's reader validates the same (ArrayData::validate_full
// for untrusted input); a truncated or corrupt file must fail here, not read beyond a host mapping.
void validate_column(const IpcField& f, int64_t length, int64_t nulls, const std::vector<const void*>& bl, const std::vector<int64_t>& lens, int64_t dict_len) {
  // (length is untrusted: bounded before any size product below, so that no `length * width` can wrap past a buffer-length check)
  DFGPU_CHECK(length >= 0 && length < ((int64_t)1 << 48) && nulls >= 0 && nulls <= length, "arrow ipc: column '" + f.name + "': bad field node (length / null count)");
  const std::string what = "arrow ipc: column '" + f.name + "': ";
  if (nulls > 0) DFGPU_CHECK(lens[0] >= (length + 7) / 8, what + "validity buffer shorter than its rows");
  const uint8_t* valid = nulls > 0 ? static_cast<const uint8_t*>(bl[0]) : nullptr;
  auto is_valid = [&](int64_t i) { return !valid || ((valid[i >> 3] >> (i & 7)) & 1); };
  if (length == 0) return;
  const std::string& fmt = f.format;
  if (f.view && f.dict_id < 0) {
    DFGPU_CHECK(lens.size() >= 2 && lens[1] >= length * 16, what + "view buffer shorter than its rows");
    const uint8_t* v = static_cast<const uint8_t*>(bl[1]);
    for (int64_t i = 0; i < length; i++) {
      if (!is_valid(i)) continue;
      int32_t n, bi, off;
      std::memcpy(&n, v + i * 16, 4);
      DFGPU_CHECK(n >= 0, what + "negative view length");
      if (n <= 12) continue;
      std::memcpy(&bi, v + i * 16 + 8, 4);
      std::memcpy(&off, v + i * 16 + 12, 4);
      DFGPU_CHECK(bi >= 0 && (size_t)bi + 2 < lens.size() && off >= 0 && (int64_t)off + n <= lens[(size_t)bi + 2], what + "a view points outside its data buffers");
    }
    return;
  }
  if (fmt == "u" || fmt == "z" || fmt == "U" || fmt == "Z") {
    const int ow = (fmt == "u" || fmt == "z") ? 4 : 8;
    DFGPU_CHECK(lens.size() >= 3 && lens[1] >= (length + 1) * ow, what + "offsets buffer shorter than its rows");
    const uint8_t* o = static_cast<const uint8_t*>(bl[1]);
    int64_t prev = 0;
    for (int64_t i = 0; i <= length; i++) {
      int64_t cur;
      if (ow == 4) {
        int32_t t;
        std::memcpy(&t, o + i * 4, 4);
        cur = t;
      } else {
        std::memcpy(&cur, o + i * 8, 8);
      }
      DFGPU_CHECK(cur >= prev && (i > 0 || cur >= 0), what + "string offsets are not non-decreasing");
      prev = cur;
    }
    DFGPU_CHECK(prev <= lens[2], what + "the last string offset lies beyond the data buffer");
    return;
  }
  const int w = format_width(fmt);
  if (w < 0) return;   // (a layout this reader does not import: dfgpu_table_import names it)
  DFGPU_CHECK(lens.size() >= 2 && lens[1] >= (w == 0 ? (length + 7) / 8 : length * w), what + "values buffer shorter than its rows");
  if (f.dict_id >= 0) {   // the values are indices into the dictionary
    const uint8_t* p = static_cast<const uint8_t*>(bl[1]);
    const bool is_unsigned = fmt == "C" || fmt == "S" || fmt == "I" || fmt == "L";
    for (int64_t i = 0; i < length; i++) {
      if (!is_valid(i)) continue;
      int64_t idx = 0;
      switch (w) {
        case 1: idx = is_unsigned ? (int64_t)p[i] : (int64_t)(int8_t)p[i]; break;
        case 2: { uint16_t t; std::memcpy(&t, p + i * 2, 2); idx = is_unsigned ? (int64_t)t : (int64_t)(int16_t)t; } break;
        case 4: { uint32_t t; std::memcpy(&t, p + i * 4, 4); idx = is_unsigned ? (int64_t)t : (int64_t)(int32_t)t; } break;
        default: std::memcpy(&idx, p + i * 8, 8); break;
      }
      DFGPU_CHECK(idx >= 0 && idx < dict_len, what + "a dictionary index lies outside its dictionary");
    }
  }
}

// one column of the batch as an ArrowArray over host bytes; `take` = false skips its buffers (a column outside the projection).
// `rows` = the batch's declared row count the node must agree with (-1: a dictionary batch); `dict_len` = length of the column's dictionary
ArrowArray* column_array(BatchCursor& c, const IpcField& f, Owned& own, bool take, int64_t rows = -1, int64_t dict_len = 0) {
  DFGPU_CHECK(c.node_at < c.n_nodes, "arrow ipc: a record batch lists fewer field nodes than its schema has columns");
  const uint8_t* node = c.nodes + (size_t)c.node_at * 16;
  c.node_at++;
  const int64_t length = c.fb.rd<int64_t>(node), nulls = c.fb.rd<int64_t>(node + 8);
  int n_buf = f.n_buffers;
  int64_t n_data = 0;
  if (f.view && f.dict_id < 0) {
    DFGPU_CHECK(c.var_at < c.n_variadic, "arrow ipc: Utf8View column without its variadic buffer count");
    n_data = c.fb.rd<int64_t>(c.variadic + (size_t)c.var_at * 8);
    c.var_at++;
    n_buf += (int)n_data;
  }
  if (!take) {
    c.buf_at += (uint32_t)n_buf;
    return nullptr;
  }
  own.buffer_lists.emplace_back();
  std::vector<const void*>& bl = own.buffer_lists.back();
  std::vector<int64_t> data_lens, lens;
  for (int i = 0; i < n_buf; i++) {
    int64_t len = 0;
    bl.push_back(next_buffer(c, own, &len));
    lens.push_back(len);
    if (i >= 2) data_lens.push_back(len);
  }
  DFGPU_CHECK(rows < 0 || length == rows, "arrow ipc: column '" + f.name + "': its field node's length differs from the record batch's");
  validate_column(f, length, nulls, bl, lens, dict_len);
  if (nulls == 0) bl[0] = nullptr;
  if (f.view && f.dict_id < 0) {
    // the C Data Interface carries a view array's data-buffer lengths as one more (last) buffer of int64
    own.decompressed.emplace_back(data_lens.size() * 8 + 8);
    std::memcpy(own.decompressed.back().data(), data_lens.data(), data_lens.size() * 8);
    bl.push_back(own.decompressed.back().data());
  }
  own.arrays.push_back(std::make_unique<ArrowArray>());
  ArrowArray* a = own.arrays.back().get();
  std::memset(a, 0, sizeof(*a));
  a->length = length;
  a->null_count = nulls;
  a->n_buffers = (int64_t)bl.size();
  a->buffers = bl.data();
  a->release = [](ArrowArray* x) { x->release = nullptr; };
  return a;
}

ArrowSchema* plain_schema(Owned& own, const std::string& format, const std::string& name, bool nullable) {
  own.strings.push_back(format);
  const char* fmt = own.strings.back().c_str();
  own.strings.push_back(name);
  const char* nm = own.strings.back().c_str();
  own.schemas.push_back(std::make_unique<ArrowSchema>());
  ArrowSchema* s = own.schemas.back().get();
  std::memset(s, 0, sizeof(*s));
  s->format = fmt;
  s->name = nm;
  s->flags = nullable ? 2 : 0;
  s->release = release_plain_schema;
  return s;
}

}  // namespace
}  // namespace dfgpu

using namespace dfgpu;

extern "C" {

int dfgpu_ipc_open(const uint8_t* data, int64_t nbytes, dfgpu_ipc_t* out) {
  return guarded([&] {
    DFGPU_CHECK(data && nbytes >= 8 && out, "dfgpu_ipc_open: empty input");
    *out = reinterpret_cast<dfgpu_ipc_t>(open_ipc(data, nbytes).release());
  });
}
int dfgpu_ipc_close(dfgpu_ipc_t h) {
  return guarded([&] { delete reinterpret_cast<IpcFile*>(h); });
}
int dfgpu_ipc_info(dfgpu_ipc_t h, int64_t* n_batches, int32_t* n_columns, int32_t* is_file_format) {
  return guarded([&] {
    DFGPU_CHECK(h != nullptr, "null ipc handle");
    const IpcFile& f = *reinterpret_cast<IpcFile*>(h);
    if (n_batches) *n_batches = (int64_t)f.batches.size();
    if (n_columns) *n_columns = (int32_t)f.fields.size();
    if (is_file_format) *is_file_format = f.is_file ? 1 : 0;
  });
}
int dfgpu_ipc_column(dfgpu_ipc_t h, int32_t i, const char** name, const char** format, int32_t* nullable, int32_t* dictionary_encoded) {
  return guarded([&] {
    DFGPU_CHECK(h != nullptr, "null ipc handle");
    const IpcFile& f = *reinterpret_cast<IpcFile*>(h);
    DFGPU_CHECK(i >= 0 && i < (int32_t)f.fields.size(), "ipc column index out of range");
    const IpcField& c = f.fields[(size_t)i];
    if (name) *name = c.name.c_str();
    if (format) *format = c.dict_id >= 0 ? c.value_format.c_str() : c.format.c_str();
    if (nullable) *nullable = c.nullable ? 1 : 0;
    if (dictionary_encoded) *dictionary_encoded = c.dict_id >= 0 ? 1 : 0;
  });
}
int dfgpu_ipc_batch_rows(dfgpu_ipc_t h, int64_t i, int64_t* rows) {
  return guarded([&] {
    DFGPU_CHECK(h && rows, "null argument");
    const IpcFile& f = *reinterpret_cast<IpcFile*>(h);
    DFGPU_CHECK(i >= 0 && i < (int64_t)f.batches.size(), "ipc batch index out of range");
    *rows = f.batches[(size_t)i].rows;
  });
}

int dfgpu_ipc_read_batch(dfgpu_ipc_t h, int64_t i, const int* columns, int ncols, dfgpu_table_t* out) {
  return guarded([&] {
    require_init();
    DFGPU_CHECK(h && out, "null argument");
    const IpcFile& f = *reinterpret_cast<IpcFile*>(h);
    DFGPU_CHECK(i >= 0 && i < (int64_t)f.batches.size(), "ipc batch index out of range");
    std::vector<int> want;
    if (columns) want.assign(columns, columns + ncols);
    else for (int c = 0; c < (int)f.fields.size(); c++) want.push_back(c);
    std::vector<int> slot(f.fields.size(), -1);
    for (size_t k = 0; k < want.size(); k++) {
      DFGPU_CHECK(want[k] >= 0 && want[k] < (int)f.fields.size(), "ipc column index out of range");
      DFGPU_CHECK(slot[(size_t)want[k]] < 0, "ipc projection names a column twice");
      slot[(size_t)want[k]] = (int)k;
    }
    auto own = std::make_unique<Owned>();
    own->strings.reserve(4 * f.fields.size() + 8);   // c_str() pointers into it must stay put
    // ---- dictionaries of the projected columns (the last batch of an id before this record batch wins: replacement dictionaries)
    std::unordered_map<int64_t, ArrowArray*> dict_arrays;
    for (int c : want) {
      const IpcField& fld = f.fields[(size_t)c];
      if (fld.dict_id < 0 || dict_arrays.count(fld.dict_id)) continue;
      const IpcBatch* db = nullptr;
      for (const auto& d : f.dictionaries)
        if (d.first == fld.dict_id && d.second.meta < f.batches[(size_t)i].meta) db = &d.second;
      DFGPU_CHECK(db != nullptr, "arrow ipc: column '" + fld.name + "' refers to a dictionary that precedes no record batch");
      BatchCursor dc = cursor_of(*db, true);
      IpcField values;
      values.name = "";
      values.format = fld.value_format;
      values.n_buffers = 3;
      dict_arrays[fld.dict_id] = column_array(dc, values, *own, true);
    }
    // ---- the record batch: every column's node and buffers are walked in schema order, the projected ones are kept
    BatchCursor bc = cursor_of(f.batches[(size_t)i], false);
    std::vector<ArrowArray*> kids(want.size(), nullptr);
    std::vector<ArrowSchema*> skids(want.size(), nullptr);
    for (size_t c = 0; c < f.fields.size(); c++) {
      const IpcField& fld = f.fields[c];
      ArrowArray* a = column_array(bc, fld, *own, slot[c] >= 0, f.batches[(size_t)i].rows, fld.dict_id >= 0 && slot[c] >= 0 ? dict_arrays[fld.dict_id]->length : 0);
      if (slot[c] < 0) continue;
      ArrowSchema* s = plain_schema(*own, fld.format, fld.name, fld.nullable);
      if (fld.dict_id >= 0) {
        a->dictionary = dict_arrays[fld.dict_id];
        s->dictionary = plain_schema(*own, fld.value_format, "", true);
      }
      kids[(size_t)slot[c]] = a;
      skids[(size_t)slot[c]] = s;
    }
    own->child_lists.push_back(kids);
    own->schema_child_lists.push_back(skids);
    own->buffer_lists.push_back({nullptr});
    ArrowArray root;
    std::memset(&root, 0, sizeof root);
    root.length = f.batches[(size_t)i].rows;
    root.n_buffers = 1;
    root.buffers = own->buffer_lists.back().data();
    root.n_children = (int64_t)kids.size();
    root.children = own->child_lists.back().data();
    root.release = release_owned_array;
    ArrowSchema rs;
    std::memset(&rs, 0, sizeof rs);
    rs.format = "+s";
    rs.name = "";
    rs.n_children = (int64_t)skids.size();
    rs.children = own->schema_child_lists.back().data();
    rs.release = release_plain_schema;
    root.private_data = own.release();
    // the import consumes both structs (copies on a side stream, pinning large buffers on the fly) and releases them
    if (dfgpu_table_import(&root, &rs, out) != 0) throw Error(dfgpu_last_error());
  });
}

}  // extern "C"
