// interop.hip — device-resident hand-off across the C ABI (no kernels here, only ownership):
//   dfgpu_table_retain            a second owner of the same buffers
//   dfgpu_table_export_device /
//   dfgpu_table_import_device     Arrow C Device Data Interface (ARROW_DEVICE_ROCM): what adjacent GPU plan nodes pass each other
//                                 instead of exporting to the host and importing again (execution_plan.rs:696-700)
//   dfgpu_cache_*                 the device-resident scan cache (HBM twin of MemorySourceConfig, datasource/src/memory.rs:58)
#include <list>
#include <unordered_map>

#include "internal.hpp"

namespace dfgpu {

int64_t table_device_bytes(const Table& t) {
  int64_t b = 0;
  for (const Column& c : t.cols) {
    if (c.field.type == DFGPU_UTF8) {
      b += (int64_t)(c.length + 1) * 8 + (c.data ? (int64_t)c.data->bytes : 0);
    } else {
      b += (int64_t)data_bytes(c.field.type, c.length);
    }
    if (c.validity) b += (int64_t)bitmap_bytes(c.length);
  }
  return b;
}

namespace {

// ------------------------------------------------------------------------------------------------ export
struct DevSchemaPrivate {
  std::string format, name;
  std::vector<ArrowSchema*> children;
  ArrowSchema* dictionary = nullptr;
};
void release_dev_schema(ArrowSchema* s) {
  if (!s || !s->release) return;
  auto* p = (DevSchemaPrivate*)s->private_data;
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
DevSchemaPrivate* fill_dev_schema(ArrowSchema* s, const std::string& fmt, const std::string& name, bool nullable) {
  auto* p = new DevSchemaPrivate{fmt, name, {}, nullptr};
  std::memset(s, 0, sizeof(*s));
  s->format = p->format.c_str();
  s->name = p->name.c_str();
  s->flags = nullable ? 2 : 0;
  s->release = release_dev_schema;
  s->private_data = p;
  return p;
}

// private data of every ArrowArray this file exports.  The ROOT's `table` is what dfgpu_table_import_device hands back.
struct DevArrayPrivate {
  uint64_t magic = 0x4446475055444556ull;  // "DFGPUDEV"
  std::unique_ptr<Table> table;            // root only: a clone of the exported table (shares its buffers)
  std::vector<BufPtr> keep;                // children: the buffers the pointers below point into
  std::vector<const void*> buffer_ptrs;
  std::vector<ArrowArray*> children;
  ArrowArray* dictionary = nullptr;
};
void release_dev_array(ArrowArray* a) {
  if (!a || !a->release) return;
  auto* p = (DevArrayPrivate*)a->private_data;
  for (ArrowArray* c : p->children) {
    if (c->release) c->release(c);
    delete c;
  }
  if (p->dictionary) {
    if (p->dictionary->release) p->dictionary->release(p->dictionary);
    delete p->dictionary;
  }
  delete p;
  a->release = nullptr;
}
ArrowArray* new_dev_array(DevArrayPrivate* p, int64_t length, int64_t null_count) {
  auto* a = new ArrowArray();
  std::memset(a, 0, sizeof(*a));
  a->length = length;
  a->null_count = null_count;
  a->n_buffers = (int64_t)p->buffer_ptrs.size();
  a->buffers = p->buffer_ptrs.data();
  a->n_children = (int64_t)p->children.size();
  a->children = p->children.empty() ? nullptr : p->children.data();
  a->dictionary = p->dictionary;
  a->release = release_dev_array;
  a->private_data = p;
  return a;
}

// a dictionary's strings as a LargeUtf8 array in HBM (uploaded once per export)
ArrowArray* export_device_dictionary(const DictValues& dv) {
  const size_t n = dv.values.size();
  std::vector<int64_t> offs(n + 1, 0);
  std::string bytes;
  std::vector<uint64_t> vbits((n + 63) / 64, 0);
  int64_t nulls = 0;
  for (size_t i = 0; i < n; i++) {
    offs[i] = (int64_t)bytes.size();
    bytes += dv.values[i];
    if (dv.valid[i]) vbits[i >> 6] |= 1ull << (i & 63); else nulls++;
  }
  offs[n] = (int64_t)bytes.size();
  auto* p = new DevArrayPrivate();
  BufPtr bo = make_buf(offs.size() * 8), bd = make_buf(bytes.size() + 8), bv;
  DFGPU_HIP(hipMemcpyAsync(bo->ptr, offs.data(), offs.size() * 8, hipMemcpyHostToDevice, rt().stream));
  if (!bytes.empty()) DFGPU_HIP(hipMemcpyAsync(bd->ptr, bytes.data(), bytes.size(), hipMemcpyHostToDevice, rt().stream));
  if (nulls) {
    bv = make_buf(vbits.size() * 8);
    DFGPU_HIP(hipMemcpyAsync(bv->ptr, vbits.data(), vbits.size() * 8, hipMemcpyHostToDevice, rt().stream));
  }
  DFGPU_HIP(hipStreamSynchronize(rt().stream));  // the host vectors go out of scope
  p->keep = {bo, bd};
  if (bv) p->keep.push_back(bv);
  p->buffer_ptrs = {bv ? bv->ptr : nullptr, bo->ptr, bd->ptr};
  return new_dev_array(p, (int64_t)n, nulls);
}

void export_device(Table* t, ArrowDeviceArray* out, ArrowSchema* out_schema) {
  DFGPU_CHECK(out && out_schema, "null argument");
  auto* rp = new DevArrayPrivate();
  rp->table = std::make_unique<Table>(*t);
  DevSchemaPrivate* sp = fill_dev_schema(out_schema, "+s", "", false);
  try {
    for (Column& c : rp->table->cols) {
      if (c.validity && c.null_count < 0) count_nulls(c);
      auto* cp = new DevArrayPrivate();
      const void* valid = (c.validity && c.null_count != 0) ? c.validity->ptr : nullptr;
      if (c.validity) cp->keep.push_back(c.validity);
      if (c.data) cp->keep.push_back(c.data);
      std::string fmt;
      if (c.field.type == DFGPU_UTF8) {
        cp->keep.push_back(c.offsets);
        cp->buffer_ptrs = {valid, c.offsets->ptr, c.ptr()};
        fmt = "U";  // the 64-bit offsets as they are stored
      } else {
        cp->buffer_ptrs = {valid, c.ptr()};
        fmt = c.dict ? c.dict->index_format : format_of(c.field);
      }
      auto* cs = new ArrowSchema();
      DevSchemaPrivate* csp = fill_dev_schema(cs, fmt, c.name, true);
      sp->children.push_back(cs);
      if (c.dict) {
        cp->dictionary = export_device_dictionary(*c.dict);
        auto* ds = new ArrowSchema();
        fill_dev_schema(ds, "U", "", true);
        csp->dictionary = ds;
        cs->dictionary = ds;
      }
      rp->children.push_back(new_dev_array(cp, c.length, valid ? c.null_count : 0));
    }
  } catch (...) {
    ArrowArray tmp{};
    tmp.release = release_dev_array;
    tmp.private_data = rp;
    release_dev_array(&tmp);
    release_dev_schema(out_schema);
    throw;
  }
  out_schema->n_children = (int64_t)sp->children.size();
  out_schema->children = sp->children.data();
  // whatever produced the table is complete before the consumer sees the pointers: no event to wait on
  DFGPU_HIP(hipStreamSynchronize(rt().stream));
  rp->buffer_ptrs = {nullptr};
  std::memset(out, 0, sizeof(*out));
  ArrowArray* root = new_dev_array(rp, t->nrows, 0);
  out->array = *root;
  delete root;  // the struct was copied out; private data now belongs to out->array
  out->device_id = t->device;
  out->device_type = ARROW_DEVICE_ROCM;
  out->sync_event = nullptr;
}

// ------------------------------------------------------------------------------------------------ import
// a foreign producer's array: released when the last column that points into it goes
struct ForeignArray {
  ArrowArray array;
  explicit ForeignArray(const ArrowArray& a) : array(a) {}
  ~ForeignArray() {
    if (array.release) array.release(&array);
  }
};

std::shared_ptr<const DictValues> import_device_dictionary(const ArrowArray* d, const ArrowSchema* ds, const char* index_format, const char* name) {
  const std::string vf(ds->format), xf(index_format);
  DFGPU_CHECK(vf == "u" || vf == "U", std::string("device dictionary column '") + name + "': only Utf8 / LargeUtf8 values are supported");
  DFGPU_CHECK(xf == "C" || xf == "i" || xf == "I" || xf == "l" || xf == "L", std::string("device dictionary column '") + name + "': unsupported index type");
  DFGPU_CHECK(d && d->n_buffers == 3 && d->offset == 0, "malformed device dictionary array");
  const int64_t n = d->length;
  const size_t ow = vf == "u" ? 4 : 8;
  std::vector<char> offs((size_t)(n + 1) * ow);
  d2h(offs.data(), d->buffers[1], offs.size());
  auto off_at = [&](int64_t i) -> int64_t { return ow == 4 ? (int64_t)((const int32_t*)offs.data())[i] : ((const int64_t*)offs.data())[i]; };
  std::string bytes((size_t)off_at(n), '\0');
  if (!bytes.empty()) d2h(&bytes[0], d->buffers[2], bytes.size());
  std::vector<uint8_t> vb;
  if (d->buffers[0]) {
    vb.resize((size_t)(n + 7) / 8);
    d2h(vb.data(), d->buffers[0], vb.size());
  }
  auto dv = std::make_shared<DictValues>();
  dv->index_format = xf;
  dv->value_format = vf;
  bool sorted = true;
  for (int64_t i = 0; i < n; i++) {
    const bool ok = vb.empty() || ((vb[(size_t)i >> 3] >> (i & 7)) & 1);
    dv->values.emplace_back(ok ? bytes.substr((size_t)off_at(i), (size_t)(off_at(i + 1) - off_at(i))) : std::string());
    dv->valid.push_back(ok ? 1 : 0);
    if (!ok || (i > 0 && !(dv->values[(size_t)i - 1] < dv->values[(size_t)i]))) sorted = false;
  }
  dv->sorted = sorted;
  return dv;
}

std::unique_ptr<Table> import_foreign(ArrowDeviceArray* in, ArrowSchema* schema) {
  DFGPU_CHECK(in->device_type == ARROW_DEVICE_ROCM, "dfgpu_table_import_device: the array is not in ROCm device memory (device_type " +
                                                        std::to_string(in->device_type) + "); host arrays go through dfgpu_table_import");
  const int device = (int)in->device_id;
  bool known = false;
  for (int d : initialised_devices()) known |= d == device;
  DFGPU_CHECK(known, "dfgpu_table_import_device: device " + std::to_string(device) + " was not given to dfgpu_init");
  use_device(device);
  DFGPU_CHECK(std::string(schema->format) == "+s", "dfgpu_table_import_device expects a struct array (RecordBatch)");
  DFGPU_CHECK(in->array.n_children == schema->n_children && in->array.offset == 0, "device array / schema mismatch (or a sliced struct array)");
  if (in->sync_event) DFGPU_HIP(hipStreamWaitEvent(rt().stream, *(hipEvent_t*)in->sync_event, 0));
  auto keep = std::make_shared<ForeignArray>(in->array);  // takes over the producer's release callback
  in->array.release = nullptr;
  auto t = std::make_unique<Table>();
  t->nrows = keep->array.length;
  t->device = device;
  for (int64_t i = 0; i < keep->array.n_children; i++) {
    const ArrowArray* a = keep->array.children[i];
    const ArrowSchema* s = schema->children[i];
    const std::string name = s->name ? s->name : "";
    DFGPU_CHECK(a->offset == 0, "device column '" + name + "': sliced arrays (offset != 0) cannot be wrapped without a copy");
    DFGPU_CHECK(a->length == t->nrows, "device column '" + name + "': length differs from the batch");
    Column c;
    c.name = name;
    c.length = a->length;
    const std::string fmt(s->format);
    DFGPU_CHECK(fmt != "u" && fmt != "vu", "device column '" + name + "': Utf8 with 32-bit offsets / Utf8View has no zero-copy device form here (LargeUtf8 does)");
    c.field = parse_format(fmt == "U" ? "U" : s->format, true);
    const size_t vbytes = bitmap_bytes(a->length);
    if (a->buffers[0] && a->null_count != 0) {
      // the library reads bitmaps as whole 64-bit words: a producer's bitmap allocation is at least 8-byte padded in every Arrow
      // implementation that follows the spec's 8-byte (recommended 64-byte) padding
      c.validity = std::make_shared<DevBuf>(const_cast<void*>(a->buffers[0]), vbytes, keep);
      c.null_count = a->null_count;
    }
    if (c.field.type == DFGPU_UTF8) {
      DFGPU_CHECK(a->n_buffers == 3, "malformed device string column '" + name + "'");
      c.offsets = std::make_shared<DevBuf>(const_cast<void*>(a->buffers[1]), (size_t)(a->length + 1) * 8, keep);
      int64_t total = 0;
      if (a->length) d2h(&total, (const char*)a->buffers[1] + (size_t)a->length * 8, 8);
      c.data = std::make_shared<DevBuf>(const_cast<void*>(a->buffers[2]), (size_t)total, keep);
    } else {
      DFGPU_CHECK(a->n_buffers == 2, "malformed device column '" + name + "'");
      c.data = std::make_shared<DevBuf>(const_cast<void*>(a->buffers[1]), data_bytes(c.field.type, a->length), keep);
      if (a->dictionary) {
        DFGPU_CHECK(s->dictionary != nullptr, "device column '" + name + "': dictionary array without a dictionary schema");
        c.dict = import_device_dictionary(a->dictionary, s->dictionary, s->format, name.c_str());
      }
    }
    t->cols.push_back(std::move(c));
  }
  return t;
}

// ------------------------------------------------------------------------------------------------ cache
struct Cache {
  std::mutex mu;
  int64_t budget = 0, bytes = 0;
  int64_t hits = 0, misses = 0, insertions = 0, evictions = 0;
  struct Entry {
    std::string key;
    std::unique_ptr<Table> table;
    int64_t bytes;
  };
  std::list<Entry> lru;  // front = most recently used
  std::unordered_map<std::string, std::list<Entry>::iterator> index;
};

}  // namespace
}  // namespace dfgpu

using namespace dfgpu;

extern "C" {

int dfgpu_table_retain(dfgpu_table_t th, dfgpu_table_t* out) {
  return guarded([&] {
    DFGPU_CHECK(out != nullptr, "null argument");
    Table* t = unwrap_quiet(th);
    *out = wrap_quiet(new Table(*t));
  });
}

int dfgpu_table_export_device(dfgpu_table_t th, struct ArrowDeviceArray* out_array, struct ArrowSchema* out_schema) {
  return guarded([&] {
    require_init();
    export_device(unwrap_quiet(th), out_array, out_schema);
  });
}

int dfgpu_table_import_device(struct ArrowDeviceArray* array, struct ArrowSchema* schema, dfgpu_table_t* out) {
  int rc = guarded([&] {
    require_init();
    DFGPU_CHECK(array && schema && out, "null argument");
    if (array->array.release == release_dev_array && ((DevArrayPrivate*)array->array.private_data)->table) {
      // one of ours: the same buffers, dictionaries, statistics and names (the schema adds nothing)
      const Table& src = *((DevArrayPrivate*)array->array.private_data)->table;
      use_device(src.device);
      *out = wrap_quiet(new Table(src));
      return;
    }
    *out = wrap_quiet(import_foreign(array, schema).release());
  });
  // the call consumes both structures whether or not it succeeded (a wrapped foreign array lives on inside the table)
  if (array && array->array.release) array->array.release(&array->array);
  if (schema && schema->release) schema->release(schema);
  return rc;
}

int dfgpu_cache_create(int64_t budget_bytes, dfgpu_cache_t* out) {
  return guarded([&] {
    DFGPU_CHECK(out != nullptr && budget_bytes >= 0, "bad argument");
    auto* c = new Cache();
    c->budget = budget_bytes;
    *out = reinterpret_cast<dfgpu_cache_t>(c);
  });
}
int dfgpu_cache_free(dfgpu_cache_t h) {
  return guarded([&] { delete reinterpret_cast<Cache*>(h); });
}
int dfgpu_cache_get(dfgpu_cache_t h, const void* key, int64_t key_bytes, dfgpu_table_t* out) {
  return guarded([&] {
    DFGPU_CHECK(h && key && key_bytes >= 0 && out, "bad argument");
    Cache& c = *reinterpret_cast<Cache*>(h);
    *out = nullptr;
    std::lock_guard<std::mutex> lk(c.mu);
    auto it = c.index.find(std::string((const char*)key, (size_t)key_bytes));
    if (it == c.index.end()) {
      c.misses++;
      return;
    }
    c.hits++;
    c.lru.splice(c.lru.begin(), c.lru, it->second);
    *out = wrap_quiet(new Table(*it->second->table));
  });
}
int dfgpu_cache_put(dfgpu_cache_t h, const void* key, int64_t key_bytes, dfgpu_table_t th) {
  return guarded([&] {
    DFGPU_CHECK(h && key && key_bytes >= 0, "bad argument");
    Cache& c = *reinterpret_cast<Cache*>(h);
    Table* t = unwrap_quiet(th);
    const int64_t nb = table_device_bytes(*t);
    std::vector<std::unique_ptr<Table>> dropped;  // freed outside the lock
    {
      std::lock_guard<std::mutex> lk(c.mu);
      if (c.budget <= 0 || nb > c.budget) return;
      std::string k((const char*)key, (size_t)key_bytes);
      if (c.index.count(k)) return;
      c.lru.push_front(Cache::Entry{k, std::make_unique<Table>(*t), nb});
      c.index[k] = c.lru.begin();
      c.bytes += nb;
      c.insertions++;
      while (c.bytes > c.budget && c.lru.size() > 1) {
        Cache::Entry& old = c.lru.back();
        c.bytes -= old.bytes;
        c.index.erase(old.key);
        dropped.push_back(std::move(old.table));
        c.lru.pop_back();
        c.evictions++;
      }
    }
  });
}
int dfgpu_cache_clear(dfgpu_cache_t h) {
  return guarded([&] {
    DFGPU_CHECK(h != nullptr, "null cache");
    Cache& c = *reinterpret_cast<Cache*>(h);
    std::list<Cache::Entry> gone;
    {
      std::lock_guard<std::mutex> lk(c.mu);
      gone.swap(c.lru);
      c.index.clear();
      c.bytes = 0;
    }
  });
}
int dfgpu_cache_get_stats(dfgpu_cache_t h, dfgpu_cache_stats* out) {
  return guarded([&] {
    DFGPU_CHECK(h && out, "null argument");
    Cache& c = *reinterpret_cast<Cache*>(h);
    std::lock_guard<std::mutex> lk(c.mu);
    *out = dfgpu_cache_stats{(int64_t)c.lru.size(), c.bytes, c.budget, c.hits, c.misses, c.insertions, c.evictions};
  });
}

}  // extern "C"
