// exchange.hip — the multi-GPU exchange below the C ABI: RepartitionExec(Partitioning::Hash) across GPUs
// (physical-plan/src/repartition/mod.rs:1097-1150: hash(keys; seed 0) % n routes a row), the build-side collection of
// PartitionMode::CollectLeft (hash_join/exec.rs:1325-1328) as an all-gather, and that all-gather pruned by the
// destinations' probe-key bounds (hash_join/shared_bounds.rs:277-284 turned around).  In the reference these are
// in-process tokio channels between partition tasks; here a partition is a GPU and the channel is RCCL over xGMI.
//
// Transport: RCCL point-to-point — one ncclGroup of ncclSend / ncclRecv per column, every peer pair at once.  xGMI is a
// full mesh of point-to-point links (7 x ~153 GB/s per GPU), so an all-to-all(v) made of direct sends puts exactly one
// peer's slice on each link and needs no ring; the all-gather of a build side is the same pattern with every peer
// receiving the same slice.  librccl is opened with dlopen when the first communicator is created: the library loads on
// hosts without RCCL, and a process that already carries a copy (torch's) shares it by soname.
// Two deployments (include/dfgpu.h): one process per GPU (ncclCommInitRank, the bootstrap id travels through the host
// engine) or one process driving several GPUs (ncclCommInitAll over the devices given to dfgpu_init; every call then
// takes one table per local rank).  A third transport moves the same slices through host memory with collectives the
// embedding engine supplies (dfgpu_comm_init_host): inter-node engines, and tests where two ranks share one GPU.
//
// What crosses: value buffers as raw bytes per (column, peer); validity bitmaps as whole 64-bit words per (column,
// peer), re-based to the receiver's row offsets with a bit-granular placement kernel; dictionaries of dictionary-encoded
// string columns are compared by digest first and merged (ascending, identical on every rank) only when they differ,
// after which every rank rewrites its indices — so hash routing and later joins / group-bys on the indices mean the
// strings on every GPU.
#include <dlfcn.h>

#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <deque>
#include <thread>

#include "device.hpp"
#include "internal.hpp"

namespace dfgpu {

// ------------------------------------------------------------------------------------------------ RCCL by dlopen
typedef struct ncclComm* ncclComm_t;
struct ncclUniqueId { char internal[128]; };
enum { NCCL_SUCCESS = 0, NCCL_INT8 = 0, NCCL_INT64 = 4 };  // ncclDataType_t values of rccl.h (ncclInt8 = 0, ncclInt64 = 4)

struct RcclApi {
  void* handle = nullptr;
  int (*GetUniqueId)(ncclUniqueId*) = nullptr;
  int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  int (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*CommCount)(const ncclComm_t, int*) = nullptr;
  int (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
static RcclApi& rccl() {
  static RcclApi api;
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  if (api.handle) return api;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (api.handle) break;
  }
  DFGPU_CHECK(api.handle != nullptr, std::string("RCCL is not available: ") + (dlerror() ? dlerror() : "librccl.so.1 not found") +
                                         " (multi-GPU exchanges have no fallback transport unless the host supplies one: dfgpu_comm_init_host)");
  auto sym = [&](const char* s) {
    void* p = dlsym(api.handle, s);
    DFGPU_CHECK(p != nullptr, std::string("librccl lacks ") + s);
    return p;
  };
  api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
  api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
  api.CommInitAll = reinterpret_cast<decltype(api.CommInitAll)>(sym("ncclCommInitAll"));
  api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
  api.CommCount = reinterpret_cast<decltype(api.CommCount)>(sym("ncclCommCount"));
  api.CommUserRank = reinterpret_cast<decltype(api.CommUserRank)>(sym("ncclCommUserRank"));
  api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
  api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
  api.Send = reinterpret_cast<decltype(api.Send)>(sym("ncclSend"));
  api.Recv = reinterpret_cast<decltype(api.Recv)>(sym("ncclRecv"));
  api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
  api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
  return api;
}
#define DFGPU_NCCL(expr)                                                                                         \
  do {                                                                                                           \
    int _r = (expr);                                                                                             \
    if (_r != NCCL_SUCCESS) throw ::dfgpu::Error(std::string("RCCL error ") + rccl().GetErrorString(_r) + " at " #expr); \
  } while (0)

// ------------------------------------------------------------------------------------------------ communicator
struct Comm {
  int world = 1;
  int first_rank = 0;            // global rank of local rank 0
  std::vector<int> devices;      // one per local rank
  std::vector<ncclComm_t> nccl;  // RCCL transport: one communicator per local rank
  bool host = false;             // host transport
  dfgpu_host_transport ht{};
  std::mutex mu;
  dfgpu_exchange_stats stats{};
  // streamed exchanges open on this communicator: their collectives run on library threads, so while one is open the communicator
  // takes no other collective and cannot be freed (a HashStream holds a plain pointer to it)
  std::atomic<int> open_streams{0};
  int n_local() const { return (int)devices.size(); }
};
// the communicator behind a handle, free to start a collective of its own
static Comm& idle_comm(dfgpu_comm_t h, const char* who) {
  Comm& c = *reinterpret_cast<Comm*>(h);
  DFGPU_CHECK(c.open_streams.load() == 0, std::string(who) + ": a streamed exchange is open on this communicator (dfgpu_exchange_hash_stream_free it first)");
  return c;
}
// a send / receive is cut into messages of at most this many bytes (every rank derives the same cuts from the row counts)
static int64_t max_message_bytes() {
  static const int64_t v = [] {
    return (int64_t)1 << 30;
  }();
  return v;
}

// one local rank's share of an all-to-all(v): slice `send[p]` goes to global rank p, `recv[p]` receives from it
struct Xfer {
  std::vector<const void*> send;
  std::vector<int64_t> send_bytes;
  std::vector<void*> recv;
  std::vector<int64_t> recv_bytes;
  explicit Xfer(int world) : send(world, nullptr), send_bytes(world, 0), recv(world, nullptr), recv_bytes(world, 0) {}
};

// device buffers of every local rank exchange their slices; enqueued on each device's library stream (stream-ordered
// after the kernels that produced the slices, before the kernels that read what arrives)
static void alltoallv_one(Comm& c, std::vector<Xfer>& x, bool own_group);
// several all-to-all(v)s (the columns of a table) as ONE ncclGroup: every slice of every column is in flight at once
static void alltoallv(Comm& c, std::vector<std::vector<Xfer>>& xs) {
  if (xs.empty()) return;
  const bool grouped = c.world > 1 && !c.host;
  int64_t moved = 0;   // bytes this process sends (the profile's "exchange_alltoall" row: HIP events on the library stream around the sends / receives)
  for (auto& x : xs)
    for (auto& xf : x)
      for (int64_t b : xf.send_bytes) moved += b;
  ProfileScope ps("exchange_alltoall", moved);
  if (grouped) DFGPU_NCCL(rccl().GroupStart());
  for (auto& x : xs) alltoallv_one(c, x, !grouped);
  if (grouped) DFGPU_NCCL(rccl().GroupEnd());
  std::lock_guard<std::mutex> lk(c.mu);
  c.stats.collectives += 1;
}
static void alltoallv_one(Comm& c, std::vector<Xfer>& x, bool own_group) {
  const int L = c.n_local();
  DFGPU_CHECK((int)x.size() == L, "internal: one transfer set per local rank");
  int64_t sent = 0, received = 0, messages = 0;
  // slices that stay on their GPU (or move between two GPUs of this process over the host transport) are plain copies
  for (int l = 0; l < L; l++) {
    const int me = c.first_rank + l;
    if (x[l].send_bytes[me] > 0) {
      DFGPU_CHECK(x[l].send_bytes[me] == x[l].recv_bytes[me], "internal: self slice sizes differ");
      use_device(c.devices[l]);
      if (x[l].send[me] != x[l].recv[me])
        DFGPU_HIP(hipMemcpyAsync(x[l].recv[me], x[l].send[me], (size_t)x[l].send_bytes[me], hipMemcpyDeviceToDevice, rt().stream));
    }
    for (int p = 0; p < c.world; p++)
      if (p != me) {
        sent += x[l].send_bytes[p];
        received += x[l].recv_bytes[p];
      }
  }
  if (c.world > 1) {
    if (!c.host) {
      RcclApi& R = rccl();
      const int64_t cut = max_message_bytes();
      if (own_group) DFGPU_NCCL(R.GroupStart());
      for (int l = 0; l < L; l++) {
        const int me = c.first_rank + l;
        hipStream_t st = rt_of(c.devices[l]).stream;
        for (int p = 0; p < c.world; p++) {
          if (p == me) continue;
          for (int64_t o = 0; o < x[l].send_bytes[p]; o += cut) {
            DFGPU_NCCL(R.Send((const char*)x[l].send[p] + o, (size_t)std::min(cut, x[l].send_bytes[p] - o), NCCL_INT8, p, c.nccl[l], st));
            messages++;
          }
          for (int64_t o = 0; o < x[l].recv_bytes[p]; o += cut)
            DFGPU_NCCL(R.Recv((char*)x[l].recv[p] + o, (size_t)std::min(cut, x[l].recv_bytes[p] - o), NCCL_INT8, p, c.nccl[l], st));
        }
      }
      if (own_group) DFGPU_NCCL(R.GroupEnd());
    } else {
      DFGPU_CHECK(L == 1, "the host transport carries one rank per process");
      use_device(c.devices[0]);
      Runtime& r = rt();
      const int me = c.first_rank;
      int64_t st = 0, rt_ = 0;
      for (int p = 0; p < c.world; p++)
        if (p != me) {
          st += x[0].send_bytes[p];
          rt_ += x[0].recv_bytes[p];
        }
      char *hs = nullptr, *hr = nullptr;
      DFGPU_HIP(hipHostMalloc((void**)&hs, (size_t)std::max<int64_t>(st, 1), hipHostMallocDefault));
      DFGPU_HIP(hipHostMalloc((void**)&hr, (size_t)std::max<int64_t>(rt_, 1), hipHostMallocDefault));
      std::vector<const void*> sp(c.world, nullptr);
      std::vector<void*> rp(c.world, nullptr);
      std::vector<int64_t> sb(x[0].send_bytes), rb(x[0].recv_bytes);
      sb[me] = rb[me] = 0;
      int64_t so = 0, ro = 0;
      for (int p = 0; p < c.world; p++) {
        if (p == me) continue;
        sp[p] = hs + so;
        rp[p] = hr + ro;
        if (sb[p]) DFGPU_HIP(hipMemcpyAsync(hs + so, x[0].send[p], (size_t)sb[p], hipMemcpyDeviceToHost, r.stream));
        so += sb[p];
        ro += rb[p];
        messages += sb[p] > 0;
      }
      DFGPU_HIP(hipStreamSynchronize(r.stream));
      const int rc = c.ht.alltoallv(c.ht.ctx, sp.data(), sb.data(), rp.data(), rb.data());
      if (rc == 0)
        for (int p = 0; p < c.world; p++)
          if (p != me && rb[p]) DFGPU_HIP(hipMemcpyAsync(x[0].recv[p], rp[p], (size_t)rb[p], hipMemcpyHostToDevice, r.stream));
      (void)hipStreamSynchronize(r.stream);
      (void)hipHostFree(hs);
      (void)hipHostFree(hr);
      DFGPU_CHECK(rc == 0, "host transport: alltoallv failed");
    }
  }
  std::lock_guard<std::mutex> lk(c.mu);
  c.stats.bytes_sent_to_peers += sent;
  c.stats.bytes_received_from_peers += received;
  c.stats.messages += messages;
}

// small host-side metadata: every local rank contributes `bytes` bytes, everyone learns all `world` contributions (rank order)
static std::vector<uint8_t> allgather_meta(Comm& c, const std::vector<std::vector<uint8_t>>& mine, int64_t bytes) {
  const int L = c.n_local();
  std::vector<uint8_t> all((size_t)c.world * bytes);
  if (L == c.world) {  // every rank lives in this process
    for (int l = 0; l < L; l++) std::memcpy(all.data() + (size_t)(c.first_rank + l) * bytes, mine[l].data(), (size_t)bytes);
    return all;
  }
  if (c.host) {
    DFGPU_CHECK(c.ht.allgather(c.ht.ctx, mine[0].data(), bytes, all.data()) == 0, "host transport: allgather failed");
    return all;
  }
  RcclApi& R = rccl();
  std::vector<BufPtr> in(L), out(L);
  for (int l = 0; l < L; l++) {
    use_device(c.devices[l]);
    in[l] = make_buf((size_t)bytes + 16);
    out[l] = make_buf((size_t)c.world * bytes + 16);
    DFGPU_HIP(hipMemcpyAsync(in[l]->ptr, mine[l].data(), (size_t)bytes, hipMemcpyHostToDevice, rt().stream));
  }
  DFGPU_NCCL(R.GroupStart());
  for (int l = 0; l < L; l++) DFGPU_NCCL(R.AllGather(in[l]->ptr, out[l]->ptr, (size_t)bytes, NCCL_INT8, c.nccl[l], rt_of(c.devices[l]).stream));
  DFGPU_NCCL(R.GroupEnd());
  use_device(c.devices[0]);
  d2h(all.data(), out[0]->ptr, all.size());
  for (int l = 1; l < L; l++) {
    use_device(c.devices[l]);
    DFGPU_HIP(hipStreamSynchronize(rt().stream));
  }
  return all;
}
// variable-size contributions: sizes first, then the payloads padded to the largest
static std::vector<std::vector<uint8_t>> allgather_blobs(Comm& c, const std::vector<std::vector<uint8_t>>& mine) {
  const int L = c.n_local();
  std::vector<std::vector<uint8_t>> sz(L, std::vector<uint8_t>(8));
  for (int l = 0; l < L; l++) {
    const int64_t n = (int64_t)mine[l].size();
    std::memcpy(sz[l].data(), &n, 8);
  }
  const std::vector<uint8_t> all_sz = allgather_meta(c, sz, 8);
  int64_t mx = 1;
  std::vector<int64_t> sizes(c.world);
  for (int r = 0; r < c.world; r++) {
    std::memcpy(&sizes[r], all_sz.data() + (size_t)r * 8, 8);
    mx = std::max(mx, sizes[r]);
  }
  std::vector<std::vector<uint8_t>> padded(L);
  for (int l = 0; l < L; l++) {
    padded[l] = mine[l];
    padded[l].resize((size_t)mx, 0);
  }
  const std::vector<uint8_t> all = allgather_meta(c, padded, mx);
  std::vector<std::vector<uint8_t>> out(c.world);
  for (int r = 0; r < c.world; r++) out[r].assign(all.begin() + (size_t)r * mx, all.begin() + (size_t)r * mx + (size_t)sizes[r]);
  return out;
}

// ------------------------------------------------------------------------------------------------ dictionaries
static uint64_t fnv1a(const uint8_t* p, size_t n, uint64_t h = 0xcbf29ce484222325ull) {
  for (size_t i = 0; i < n; i++) h = (h ^ p[i]) * 0x100000001b3ull;
  return h;
}
static void put_u32(std::vector<uint8_t>& b, uint32_t v) { b.insert(b.end(), (uint8_t*)&v, (uint8_t*)&v + 4); }
static std::vector<uint8_t> serialize_dict(const DictValues& d) {
  std::vector<uint8_t> b;
  put_u32(b, (uint32_t)d.index_format.size());
  b.insert(b.end(), d.index_format.begin(), d.index_format.end());
  put_u32(b, (uint32_t)d.value_format.size());
  b.insert(b.end(), d.value_format.begin(), d.value_format.end());
  put_u32(b, (uint32_t)d.values.size());
  for (size_t i = 0; i < d.values.size(); i++) {
    b.push_back(d.valid[i]);
    put_u32(b, (uint32_t)d.values[i].size());
    b.insert(b.end(), d.values[i].begin(), d.values[i].end());
  }
  return b;
}
static DictValues deserialize_dict(const std::vector<uint8_t>& b) {
  DictValues d;
  size_t o = 0;
  auto u32 = [&] {
    DFGPU_CHECK(o + 4 <= b.size(), "exchange: truncated dictionary");
    uint32_t v;
    std::memcpy(&v, b.data() + o, 4);
    o += 4;
    return v;
  };
  auto str = [&](uint32_t n) {
    DFGPU_CHECK(o + n <= b.size(), "exchange: truncated dictionary");
    std::string s((const char*)b.data() + o, n);
    o += n;
    return s;
  };
  d.index_format = str(u32());
  d.value_format = str(u32());
  const uint32_t n = u32();
  for (uint32_t i = 0; i < n; i++) {
    DFGPU_CHECK(o + 1 <= b.size(), "exchange: truncated dictionary");
    d.valid.push_back(b[o++]);
    d.values.push_back(str(u32()));
  }
  return d;
}

// After this every rank's column `ci` (of every local table) is encoded with one dictionary: untouched when all ranks
// already agree (digest), else the ascending merge of all ranks' values (NULL value first), identical everywhere.
static void unify_dictionaries(Comm& c, std::vector<Table>& t) {
  const int L = c.n_local();
  const size_t ncols = t[0].cols.size();
  bool any_dict = false;
  for (int l = 0; l < L; l++)
    for (const Column& col : t[l].cols) any_dict |= col.dict != nullptr;
  if (!any_dict) return;  // the schema is the same on every rank (the planner's): no rank holds an encoded column
  // round 1: per column {encoded?, digest}
  std::vector<std::vector<uint8_t>> meta(L, std::vector<uint8_t>(ncols * 9, 0));
  std::vector<std::vector<std::vector<uint8_t>>> ser(L, std::vector<std::vector<uint8_t>>(ncols));
  for (int l = 0; l < L; l++)
    for (size_t ci = 0; ci < ncols; ci++) {
      if (!t[l].cols[ci].dict) continue;
      ser[l][ci] = serialize_dict(*t[l].cols[ci].dict);
      const uint64_t h = fnv1a(ser[l][ci].data(), ser[l][ci].size());
      meta[l][ci * 9] = 1;
      std::memcpy(&meta[l][ci * 9 + 1], &h, 8);
    }
  const std::vector<uint8_t> all = allgather_meta(c, meta, (int64_t)ncols * 9);
  for (size_t ci = 0; ci < ncols; ci++) {
    bool any = false, every = true, same = true;
    uint64_t h0 = 0;
    for (int r = 0; r < c.world; r++) {
      const uint8_t* m = all.data() + ((size_t)r * ncols + ci) * 9;
      any |= m[0] != 0;
      every &= m[0] != 0;
      uint64_t h;
      std::memcpy(&h, m + 1, 8);
      if (r == 0) h0 = h;
      same &= h == h0;
    }
    if (!any) continue;
    DFGPU_CHECK(every, "exchange: column " + t[0].cols[ci].name + " is dictionary-encoded on some ranks only");
    if (same) continue;
    // round 2: everyone's dictionary -> the merged one
    std::vector<std::vector<uint8_t>> mine(L);
    for (int l = 0; l < L; l++) mine[l] = ser[l][ci];
    const auto blobs = allgather_blobs(c, mine);
    std::map<std::string, int> keys;  // ascending; the NULL value's marker sorts first
    bool has_null = false;
    DictValues first = deserialize_dict(blobs[0]);
    for (int r = 0; r < c.world; r++) {
      const DictValues d = r == 0 ? first : deserialize_dict(blobs[r]);
      for (size_t k = 0; k < d.values.size(); k++) {
        if (d.valid[k]) keys.emplace(d.values[k], 0);
        else has_null = true;
      }
    }
    auto merged = std::make_shared<DictValues>();
    merged->index_format = first.index_format;
    merged->value_format = first.value_format;
    if (has_null) {
      merged->values.push_back(std::string());
      merged->valid.push_back(0);
    }
    for (auto& kv : keys) {
      merged->values.push_back(kv.first);
      merged->valid.push_back(1);
    }
    merged->sorted = !has_null;
    for (int l = 0; l < L; l++) {
      use_device(c.devices[l]);
      t[l].cols[ci] = remap_to_dictionary(t[l].cols[ci], merged);
    }
  }
}

// ------------------------------------------------------------------------------------------------ the exchange proper
__global__ __launch_bounds__(BLOCK) void k_offsets_to_lengths(const int64_t* __restrict__ off, int64_t n, uint32_t* __restrict__ len) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) len[i] = (uint32_t)(off[i + 1] - off[i]);
}
__global__ __launch_bounds__(BLOCK) void k_fill_words(uint64_t* __restrict__ p, int64_t nw, uint64_t v) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < nw; i += (int64_t)gridDim.x * BLOCK) p[i] = v;
}
// rows whose key lies in [lo, hi] (NULL keys never): the slice of a build side one destination can match
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_range_mask(const T* __restrict__ key, const uint64_t* __restrict__ valid, int64_t n, long long lo, long long hi,
                                                      uint64_t* __restrict__ mask) {
  const int64_t n_words = (n + 63) >> 6;
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * BLOCK) >> 6;
  for (int64_t w = wave; w < n_words; w += n_waves) {
    const int64_t i = (w << 6) + lane_id();
    bool in = false;
    if (i < n) {
      const long long v = (long long)key[i];
      in = v >= lo && v <= hi;
    }
    uint64_t word = ballot64(in);
    if (valid) word &= valid[w];
    if (lane_id() == 0) mask[w] = word;
  }
}

// parts[l][p] = the rows local rank l sends to global rank p (every part of one local rank has the schema of proto[l]);
// returns, per local rank, the rows it receives from all ranks in rank order.
// `poisoned`: this rank takes part in the metadata all-gather only to tell the others that it has FAILED (row counts of -1): every
// rank — this one included — then abandons the exchange with the same error instead of waiting for rows that never come.
struct ExchangeAbandoned : Error {
  using Error::Error;
};
static std::vector<Table> exchange_parts(Comm& c, const std::vector<std::vector<Table>>& parts, const std::vector<Table>& proto, bool poisoned = false) {
  const int L = c.n_local(), W = c.world;
  const size_t ncols = proto[0].cols.size();
  // Utf8 columns travel as two buffers: one 32-bit length per row (scanned into Arrow offsets at the receiver) and the bytes; the
  // receiver must know how many bytes every source sends it
  std::vector<size_t> utf8_cols;
  for (size_t ci = 0; ci < ncols; ci++)
    if (proto[0].cols[ci].field.type == DFGPU_UTF8) utf8_cols.push_back(ci);
  const size_t NU = utf8_cols.size();
  // what every rank must know: rows[src][dst], per column whether any rank carries a validity bitmap + the type (a schema check),
  // and string bytes[src][dst] per Utf8 column
  const int64_t meta_strings = (int64_t)W * 8 + (int64_t)ncols * 2;
  const int64_t meta_bytes = meta_strings + (int64_t)NU * W * 8;
  std::vector<std::vector<uint8_t>> meta(L, std::vector<uint8_t>((size_t)meta_bytes, 0));
  // string bytes of part (l, p), column utf8_cols[u]: [first, last) of its data buffer
  std::vector<std::vector<std::vector<std::pair<int64_t, int64_t>>>> str_range(L, std::vector<std::vector<std::pair<int64_t, int64_t>>>(W, std::vector<std::pair<int64_t, int64_t>>(NU, {0, 0})));
  for (int l = 0; l < L && NU; l++) {
    use_device(c.devices[l]);
    for (int p = 0; p < W; p++)
      for (size_t u = 0; u < NU; u++) {
        const Column& pc = parts[l][p].cols[utf8_cols[u]];
        const int64_t n = parts[l][p].nrows;
        if (n == 0 || !pc.offsets) continue;
        const uint64_t first = read_u64(reinterpret_cast<const uint64_t*>(str_offsets(pc))), last = read_u64(reinterpret_cast<const uint64_t*>(str_offsets(pc)) + n);
        str_range[l][p][u] = {(int64_t)first, (int64_t)last};
        const int64_t nb = (int64_t)(last - first);
        std::memcpy(&meta[l][(size_t)meta_strings + (u * W + (size_t)p) * 8], &nb, 8);
      }
  }
  for (int l = 0; l < L; l++) {
    DFGPU_CHECK(proto[l].cols.size() == ncols, "exchange: the local tables differ in their column count");
    for (int p = 0; p < W; p++) {
      const int64_t n = poisoned ? -1 : parts[l][p].nrows;
      std::memcpy(&meta[l][(size_t)p * 8], &n, 8);
    }
    for (size_t ci = 0; ci < ncols; ci++) {
      bool v = false;
      for (int p = 0; p < W; p++) v |= parts[l][p].nrows > 0 && parts[l][p].cols[ci].validity != nullptr;
      meta[l][(size_t)W * 8 + ci * 2] = v;
      meta[l][(size_t)W * 8 + ci * 2 + 1] = (uint8_t)proto[l].cols[ci].field.type;
    }
  }
  const std::vector<uint8_t> all = allgather_meta(c, meta, meta_bytes);
  auto rows = [&](int src, int dst) {
    int64_t n;
    std::memcpy(&n, all.data() + (size_t)src * meta_bytes + (size_t)dst * 8, 8);
    return n;
  };
  auto str_bytes = [&](int src, int dst, size_t u) {
    int64_t n;
    std::memcpy(&n, all.data() + (size_t)src * meta_bytes + (size_t)meta_strings + (u * W + (size_t)dst) * 8, 8);
    return n;
  };
  for (int r = 0; r < W; r++)
    if (rows(r, 0) < 0) throw ExchangeAbandoned("exchange abandoned on every rank: rank " + std::to_string(r) + " reported an error (or its consumer left the stream early)");
  std::vector<uint8_t> any_valid(ncols, 0);
  for (size_t ci = 0; ci < ncols; ci++)
    for (int r = 0; r < W; r++) {
      const uint8_t* m = all.data() + (size_t)r * meta_bytes + (size_t)W * 8 + ci * 2;
      any_valid[ci] |= m[0];
      DFGPU_CHECK(m[1] == (uint8_t)proto[0].cols[ci].field.type, "exchange: column " + proto[0].cols[ci].name + " has different types on different ranks");
    }

  std::vector<Table> out(L);
  std::vector<std::vector<int64_t>> off(L, std::vector<int64_t>(W + 1, 0));  // receive offsets in rows
  int64_t rows_to_peers = 0, rows_from_peers = 0;
  for (int l = 0; l < L; l++) {
    const int me = c.first_rank + l;
    use_device(c.devices[l]);
    for (int p = 0; p < W; p++) {
      off[l][p + 1] = off[l][p] + rows(p, me);
      if (p != me) {
        rows_to_peers += rows(me, p);
        rows_from_peers += rows(p, me);
      }
    }
    out[l] = Table();
    out[l].nrows = off[l][W];
    for (size_t ci = 0; ci < ncols; ci++) {
      if (proto[l].cols[ci].field.type == DFGPU_UTF8) {  // offsets and bytes are made after the lengths arrived
        Column n = alloc_string_column(proto[l].cols[ci], out[l].nrows);
        n.validity.reset();
        n.null_count = 0;
        if (any_valid[ci]) {
          n.validity = make_zero_buf(bitmap_bytes(out[l].nrows));
          n.null_count = -1;
        }
        out[l].cols.push_back(std::move(n));
        continue;
      }
      Column n = alloc_like(proto[l].cols[ci], out[l].nrows);
      if (proto[l].cols[ci].field.type == DFGPU_BOOL && out[l].nrows) DFGPU_HIP(hipMemsetAsync(n.data->ptr, 0, bitmap_bytes(out[l].nrows), rt().stream));
      if (any_valid[ci]) {
        n.validity = make_zero_buf(bitmap_bytes(out[l].nrows));
        n.null_count = -1;
      }
      out[l].cols.push_back(std::move(n));
    }
  }

  struct Placement { size_t ci; bool validity; std::vector<BufPtr> stage; std::vector<std::vector<int64_t>> soff; };
  std::vector<Placement> placements;
  std::vector<BufPtr> keep;  // staging that must outlive the enqueued transfers
  std::vector<std::vector<Xfer>> xs;
  // per local rank and Utf8 column: the received lengths (rows of the result), the byte offsets of every source's share
  std::vector<std::vector<BufPtr>> str_len(L, std::vector<BufPtr>(NU));
  for (size_t u = 0; u < NU; u++) {
    const size_t ci = utf8_cols[u];
    std::vector<Xfer> xl(L, Xfer(W)), xb(L, Xfer(W));
    for (int l = 0; l < L; l++) {
      const int me = c.first_rank + l;
      use_device(c.devices[l]);
      str_len[l][u] = make_buf((size_t)std::max<int64_t>(out[l].nrows, 1) * 4 + 16);
      int64_t total = 0;
      for (int p = 0; p < W; p++) total += str_bytes(p, me, u);
      out[l].cols[ci].data = make_buf((size_t)total + 16);
      int64_t at = 0;
      for (int p = 0; p < W; p++) {
        const Column& pc = parts[l][p].cols[ci];
        const int64_t n = parts[l][p].nrows;
        if (n) {  // this part's lengths, from its offsets
          BufPtr lens = make_buf((size_t)n * 4 + 16);
          k_offsets_to_lengths<<<grid_for(n, BLOCK), BLOCK, 0, rt().stream>>>(str_offsets(pc), n, lens->as<uint32_t>());
          DFGPU_HIP(hipGetLastError());
          keep.push_back(lens);
          xl[l].send[p] = lens->ptr;
          xb[l].send[p] = (const char*)pc.ptr() + str_range[l][p][u].first;
        }
        xl[l].send_bytes[p] = n * 4;
        xb[l].send_bytes[p] = str_range[l][p][u].second - str_range[l][p][u].first;
        xl[l].recv[p] = (char*)str_len[l][u]->ptr + (size_t)off[l][p] * 4;
        xl[l].recv_bytes[p] = rows(p, me) * 4;
        xb[l].recv[p] = (char*)out[l].cols[ci].data->ptr + at;
        xb[l].recv_bytes[p] = str_bytes(p, me, u);
        at += str_bytes(p, me, u);
      }
    }
    xs.push_back(std::move(xl));
    xs.push_back(std::move(xb));
  }
  for (size_t ci = 0; ci < ncols; ci++) {
    const int type = proto[0].cols[ci].field.type;
    // ---- values: byte-addressable types travel straight from the partition slices into the result column
    if (type != DFGPU_BOOL && type != DFGPU_UTF8) {
      const int w = type_width(type);
      std::vector<Xfer> x(L, Xfer(W));
      for (int l = 0; l < L; l++) {
        const int me = c.first_rank + l;
        for (int p = 0; p < W; p++) {
          x[l].send[p] = parts[l][p].cols[ci].ptr();
          x[l].send_bytes[p] = parts[l][p].nrows * w;
          x[l].recv[p] = (char*)out[l].cols[ci].data->ptr + (size_t)off[l][p] * w;
          x[l].recv_bytes[p] = rows(p, me) * w;
        }
      }
      xs.push_back(std::move(x));
    }
    // ---- bit-packed buffers (validity, Boolean values): whole words per peer into staging, placed at the receiver's bit offsets afterwards
    for (int what = 0; what < 2; what++) {
      const bool validity = what == 0;
      if (validity ? !any_valid[ci] : type != DFGPU_BOOL) continue;
      std::vector<Xfer> x(L, Xfer(W));
      Placement pl{ci, validity, std::vector<BufPtr>(L), std::vector<std::vector<int64_t>>(L, std::vector<int64_t>(W + 1, 0))};
      for (int l = 0; l < L; l++) {
        const int me = c.first_rank + l;
        use_device(c.devices[l]);
        for (int p = 0; p < W; p++) pl.soff[l][p + 1] = pl.soff[l][p] + (int64_t)bitmap_bytes(rows(p, me));
        pl.stage[l] = make_buf((size_t)pl.soff[l][W] + 16);
        for (int p = 0; p < W; p++) {
          const Column& pc = parts[l][p].cols[ci];
          const int64_t n = parts[l][p].nrows;
          const void* src = validity ? (const void*)pc.valid_words() : pc.ptr();
          if (validity && !src && n) {  // an all-valid part still owes the receiver its bits
            BufPtr ones = make_buf(bitmap_bytes(n));
            k_fill_words<<<grid_for((n + 63) / 64, BLOCK), BLOCK, 0, rt().stream>>>(ones->as<uint64_t>(), (n + 63) / 64, ~0ull);
            keep.push_back(ones);
            src = ones->ptr;
          }
          x[l].send[p] = src;
          x[l].send_bytes[p] = (int64_t)bitmap_bytes(n);
          x[l].recv[p] = (char*)pl.stage[l]->ptr + pl.soff[l][p];
          x[l].recv_bytes[p] = (int64_t)bitmap_bytes(rows(p, me));
        }
      }
      xs.push_back(std::move(x));
      placements.push_back(std::move(pl));
    }
  }
  alltoallv(c, xs);
  for (const Placement& pl : placements)
    for (int l = 0; l < L; l++) {
      const int me = c.first_rank + l;
      use_device(c.devices[l]);
      Column& oc = out[l].cols[pl.ci];
      uint64_t* dst = pl.validity ? oc.validity->as<uint64_t>() : oc.data->as<uint64_t>();
      for (int p = 0; p < W; p++)
        if (rows(p, me)) bitmap_place((const uint64_t*)((const char*)pl.stage[l]->ptr + pl.soff[l][p]), off[l][p], rows(p, me), dst);
    }
  for (size_t u = 0; u < NU; u++)
    for (int l = 0; l < L; l++) {  // the received lengths, in row order, become the Arrow offsets
      use_device(c.devices[l]);
      Column& oc = out[l].cols[utf8_cols[u]];
      if (out[l].nrows) scan_u32(str_len[l][u]->as<uint32_t>(), out[l].nrows, oc.offsets->as<uint64_t>());
      else DFGPU_HIP(hipMemsetAsync(oc.offsets->ptr, 0, 8, rt().stream));
    }
  for (int l = 0; l < L; l++) {  // the caller may free the parts as soon as this returns
    use_device(c.devices[l]);
    DFGPU_HIP(hipStreamSynchronize(rt().stream));
  }
  std::lock_guard<std::mutex> lk(c.mu);
  c.stats.rows_sent_to_peers += rows_to_peers;
  c.stats.rows_received_from_peers += rows_from_peers;
  return out;
}

static std::vector<Table> local_inputs(Comm& c, const dfgpu_table_t* inputs) {
  std::vector<Table> t;
  for (int l = 0; l < c.n_local(); l++) {
    const Table* p = unwrap(inputs[l]);
    DFGPU_CHECK(p->device == c.devices[l], "exchange: input table " + std::to_string(l) + " does not live on the device of local rank " + std::to_string(l));
    t.push_back(*p);  // shallow: the columns share their buffers
  }
  return t;
}
// ---- range exchange (the distributed ORDER BY): destination of a row = number of splitters <= its key
constexpr int MAX_RANGE_RANKS = 64;
struct Splitters {
  long long s[MAX_RANGE_RANKS];
  int n;  // world - 1
};
__global__ __launch_bounds__(BLOCK) void k_range_masks(KeyCol k, int64_t n, Splitters sp, int world, int descending, int null_dest, uint64_t* __restrict__ masks,
                                                       int64_t n_words) {
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * BLOCK) >> 6;
  for (int64_t w = wave; w < n_words; w += n_waves) {
    const int64_t i = (w << 6) + lane_id();
    int dest = -1;
    if (i < n) {
      if (k.valid && !bit_at(k.valid, i)) {
        dest = null_dest;
      } else {
        uint64_t lo, hi;
        load_words(k, i, lo, hi);
        const long long v = (long long)lo;
        int d = 0;
        for (int q = 0; q < sp.n; q++) d += sp.s[q] <= v ? 1 : 0;
        dest = descending ? world - 1 - d : d;
      }
    }
    for (int d = 0; d < world; d++) {
      const uint64_t b = ballot64(dest == d);
      if (lane_id() == 0) masks[(int64_t)d * n_words + w] = b;
    }
  }
}
__global__ __launch_bounds__(BLOCK) void k_sample_keys(KeyCol k, int64_t n, int64_t every, int samples, long long* __restrict__ out, int* __restrict__ n_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= samples) return;
  const int64_t r = (int64_t)i * every;
  if (r >= n || (k.valid && !bit_at(k.valid, r))) return;
  uint64_t lo, hi;
  load_words(k, r, lo, hi);
  out[atomicAdd(n_out, 1)] = (long long)lo;
}

// ------------------------------------------------------------------------------------------------ the streaming exchange
// rows [begin, end) of a table as a zero-copy view (begin a multiple of 64, so validity words and bit-packed values start on a word);
// Utf8 columns cannot be viewed this way (their offsets start at 0 by contract): the caller keeps such tables whole
static bool sliceable(const Table& t) {
  for (const Column& c : t.cols)
    if (c.field.type == DFGPU_UTF8) return false;
  return true;
}
static Table slice_rows(const Table& t, int64_t begin, int64_t end) {
  Table out;
  out.device = t.device;
  out.nrows = end - begin;
  for (const Column& c : t.cols) {
    Column v = c;
    v.length = end - begin;
    v.stats.reset();   // (the view's rows are not the source's)
    v.data_offset = c.data_offset + (c.field.type == DFGPU_BOOL ? (size_t)(begin / 8) : (size_t)begin * type_width(c.field.type));
    if (c.validity) {
      v.validity = std::make_shared<DevBuf>((char*)c.validity->ptr + begin / 8, bitmap_bytes(end - begin), std::static_pointer_cast<void>(c.validity));
      v.null_count = -1;
    }
    out.cols.push_back(std::move(v));
  }
  return out;
}

// RepartitionExec streams: every input partition hashes a batch, hands the pieces to the output partitions' channels and goes on with
// the next batch while the consumers work (repartition/mod.rs:154-360, 1097-1150).  Here the three stages of a chunk — partition
// kernel, all-to-all(v), the consumer — run on three host threads with a stream each: while chunk k is being consumed (a join
// builder's push, a probe), chunk k + 1 crosses the links and chunk k + 2 is being partitioned.  Every rank cuts its input into the
// SAME number of row ranges (the all-to-all(v)s of the ranks must pair up); a rank with fewer rows sends empty pieces.
struct HashStream {
  Comm* c = nullptr;
  std::vector<Table> inputs;   // shallow copies, dictionaries unified
  std::vector<int> keys;
  int n_chunks = 1;
  std::vector<int> devices;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<std::vector<std::vector<Table>>> parted;   // partitioner -> exchanger
  std::deque<std::vector<Table>> received;               // exchanger -> consumer
  bool parted_done = false, received_done = false, cancel = false;
  std::string error;
  std::thread partitioner, exchanger;
  int handed = 0;
  static constexpr size_t DEPTH = 1;   // chunks waiting between two stages (each stage also holds the one it works on)
};
static void hash_stream_partition(HashStream* s) {
  try {
    for (int k = 0; k < s->n_chunks; k++) {
      {
        std::unique_lock<std::mutex> lk(s->mu);
        s->cv.wait(lk, [&] { return s->cancel || s->parted.size() < HashStream::DEPTH; });
        if (s->cancel) break;
      }
      std::vector<std::vector<Table>> parts(s->inputs.size());
      for (size_t l = 0; l < s->inputs.size(); l++) {
        use_device(s->devices[l]);
        const Table& t = s->inputs[l];
        const int64_t n = t.nrows;
        auto cut = [&](int q) { return q >= s->n_chunks ? n : (n * q / s->n_chunks) / 64 * 64; };
        parts[l] = partition_table(s->n_chunks == 1 ? t : slice_rows(t, cut(k), cut(k + 1)), s->keys, s->c->world);
      }
      call_epilogue();   // this thread's streams drained: the pieces are complete before another thread sees them
      std::lock_guard<std::mutex> lk(s->mu);
      s->parted.push_back(std::move(parts));
      s->cv.notify_all();
    }
  } catch (const std::exception& e) {
    std::lock_guard<std::mutex> lk(s->mu);
    if (s->error.empty()) s->error = e.what();
  }
  call_epilogue();
  std::lock_guard<std::mutex> lk(s->mu);
  s->parted_done = true;
  s->cv.notify_all();
}
static void hash_stream_exchange(HashStream* s) {
  int exchanged = 0;         // collectives this rank has completed: its peers are about to enter number `exchanged`
  bool abandoned = false;    // the error came out of a collective every rank abandoned together: nobody waits for another one
  try {
    for (;;) {
      std::vector<std::vector<Table>> parts;
      {
        std::unique_lock<std::mutex> lk(s->mu);
        s->cv.wait(lk, [&] { return s->cancel || !s->parted.empty() || s->parted_done; });
        if (s->cancel || (s->parted.empty() && s->parted_done)) break;
        parts = std::move(s->parted.front());
        s->parted.pop_front();
        s->cv.notify_all();
      }
      std::vector<Table> res = exchange_parts(*s->c, parts, s->inputs);
      exchanged++;
      parts.clear();
      call_epilogue();
      std::unique_lock<std::mutex> lk(s->mu);
      s->cv.wait(lk, [&] { return s->cancel || s->received.size() < HashStream::DEPTH; });
      if (s->cancel) break;
      s->received.push_back(std::move(res));
      s->cv.notify_all();
    }
  } catch (const ExchangeAbandoned& e) {
    abandoned = true;
    std::lock_guard<std::mutex> lk(s->mu);
    if (s->error.empty()) s->error = e.what();
  } catch (const std::exception& e) {
    std::lock_guard<std::mutex> lk(s->mu);
    if (s->error.empty()) s->error = e.what();
  }
  // This rank stops before the last chunk — an error on one of its threads, or a consumer that left early — while its peers are about to
  // enter the next chunk's collective: it enters that one too, poisoned, so that every rank abandons the stream with an error at the
  // same chunk instead of waiting forever (round-5 advice).  Not after an abandoned collective: there everybody has left already.
  bool stopped_early;
  {
    std::lock_guard<std::mutex> lk(s->mu);
    stopped_early = (s->cancel || !s->error.empty()) && exchanged < s->n_chunks;
  }
  if (stopped_early && !abandoned && s->c->world > 1) {
    try {
      std::vector<std::vector<Table>> none(s->inputs.size());
      for (size_t l = 0; l < s->inputs.size(); l++) {   // (pieces of no rows with the input's columns: only the metadata round runs)
        Table empty;
        empty.device = s->inputs[l].device;
        empty.nrows = 0;
        for (const Column& col : s->inputs[l].cols) {
          Column v = col;
          v.length = 0;
          v.validity = nullptr;
          v.stats.reset();
          empty.cols.push_back(std::move(v));
        }
        none[l].assign((size_t)s->c->world, empty);
      }
      (void)exchange_parts(*s->c, none, s->inputs, /*poisoned=*/true);
    } catch (const std::exception& e) {
      std::lock_guard<std::mutex> lk(s->mu);
      if (s->error.empty()) s->error = e.what();
    }
  }
  call_epilogue();
  std::lock_guard<std::mutex> lk(s->mu);
  s->received_done = true;
  s->cv.notify_all();
}

static void hand_out(std::vector<Table>& res, dfgpu_table_t* outs) {
  for (size_t l = 0; l < res.size(); l++) outs[l] = wrap(new Table(std::move(res[l])));
}

}  // namespace dfgpu

using namespace dfgpu;

extern "C" {

int dfgpu_comm_unique_id(uint8_t* out_id) {
  return guarded([&] {
    DFGPU_CHECK(out_id != nullptr, "null argument");
    ncclUniqueId id;
    DFGPU_NCCL(rccl().GetUniqueId(&id));
    std::memcpy(out_id, id.internal, DFGPU_COMM_ID_BYTES);
  });
}

int dfgpu_comm_init_rank(const uint8_t* id, int world, int rank, dfgpu_comm_t* out) {
  return guarded([&] {
    require_init();
    DFGPU_CHECK(id && out && world >= 1 && rank >= 0 && rank < world, "dfgpu_comm_init_rank: bad arguments");
    auto c = std::make_unique<Comm>();
    c->world = world;
    c->first_rank = rank;
    c->devices = {current_device()};
    c->nccl.resize(1);
    ncclUniqueId uid;
    std::memcpy(uid.internal, id, DFGPU_COMM_ID_BYTES);
    (void)rt();  // hipSetDevice on this thread: RCCL binds the communicator to the current device
    DFGPU_NCCL(rccl().CommInitRank(&c->nccl[0], world, uid, rank));
    *out = reinterpret_cast<dfgpu_comm_t>(c.release());
  });
}

int dfgpu_comm_init_all(dfgpu_comm_t* out) {
  return guarded([&] {
    require_init();
    DFGPU_CHECK(out != nullptr, "null argument");
    auto c = std::make_unique<Comm>();
    c->devices = initialised_devices();
    c->world = (int)c->devices.size();
    c->first_rank = 0;
    c->nccl.resize(c->devices.size());
    DFGPU_NCCL(rccl().CommInitAll(c->nccl.data(), c->world, c->devices.data()));
    *out = reinterpret_cast<dfgpu_comm_t>(c.release());
  });
}

int dfgpu_comm_init_host(const dfgpu_host_transport* t, int world, int rank, dfgpu_comm_t* out) {
  return guarded([&] {
    require_init();
    DFGPU_CHECK(t && t->alltoallv && t->allgather && out && world >= 1 && rank >= 0 && rank < world, "dfgpu_comm_init_host: bad arguments");
    auto c = std::make_unique<Comm>();
    c->world = world;
    c->first_rank = rank;
    c->devices = {current_device()};
    c->host = true;
    c->ht = *t;
    *out = reinterpret_cast<dfgpu_comm_t>(c.release());
  });
}

int dfgpu_comm_free(dfgpu_comm_t h) {
  return guarded([&] {
    if (!h) return;
    Comm* c = reinterpret_cast<Comm*>(h);
    DFGPU_CHECK(c->open_streams.load() == 0, "dfgpu_comm_free: a streamed exchange is still open on this communicator (its threads hold it)");
    for (ncclComm_t n : c->nccl)
      if (n) (void)rccl().CommDestroy(n);
    delete c;
  });
}

int dfgpu_comm_info(dfgpu_comm_t h, int* world, int* first_rank, int* n_local) {
  return guarded([&] {
    DFGPU_CHECK(h != nullptr, "null communicator");
    Comm* c = reinterpret_cast<Comm*>(h);
    if (world) *world = c->world;
    if (first_rank) *first_rank = c->first_rank;
    if (n_local) *n_local = c->n_local();
  });
}

// what the TRANSPORT itself says about the communicator (a first run on real multi-GPU hardware diagnoses itself with it):
// is_rccl = 1 RCCL / 0 host transport; rccl_ranks = ncclCommCount of local rank 0's communicator (= world when every rank joined
// the same communicator), rccl_rank = its ncclCommUserRank; -1 where the transport is not RCCL
int dfgpu_comm_transport_info(dfgpu_comm_t h, int* is_rccl, int* rccl_ranks, int* rccl_rank) {
  return guarded([&] {
    DFGPU_CHECK(h != nullptr, "null communicator");
    Comm* c = reinterpret_cast<Comm*>(h);
    const bool r = !c->host && !c->nccl.empty();
    if (is_rccl) *is_rccl = r ? 1 : 0;
    int n = -1, me = -1;
    if (r) {
      DFGPU_NCCL(rccl().CommCount(c->nccl[0], &n));
      DFGPU_NCCL(rccl().CommUserRank(c->nccl[0], &me));
    }
    if (rccl_ranks) *rccl_ranks = n;
    if (rccl_rank) *rccl_rank = me;
  });
}

int dfgpu_comm_stats(dfgpu_comm_t h, dfgpu_exchange_stats* out, int reset) {
  return guarded([&] {
    DFGPU_CHECK(h != nullptr, "null communicator");
    Comm* c = reinterpret_cast<Comm*>(h);
    std::lock_guard<std::mutex> lk(c->mu);
    if (out) *out = c->stats;
    if (reset) c->stats = dfgpu_exchange_stats{};
  });
}

int dfgpu_exchange_hash(dfgpu_comm_t h, const dfgpu_table_t* inputs, const int* key_cols, int nkeys, dfgpu_table_t* outs) {
  return guarded([&] {
    require_init();
    DFGPU_CHECK(h && inputs && outs && key_cols && nkeys >= 1, "dfgpu_exchange_hash: bad arguments");
    Comm& c = idle_comm(h, "dfgpu_exchange_hash");
    DFGPU_CHECK(c.world <= 64, "dfgpu_exchange_hash supports up to 64 ranks");
    std::vector<Table> t = local_inputs(c, inputs);
    unify_dictionaries(c, t);
    std::vector<std::vector<Table>> parts(c.n_local());
    const std::vector<int> keys(key_cols, key_cols + nkeys);
    for (int l = 0; l < c.n_local(); l++) {
      use_device(c.devices[l]);
      parts[l] = partition_table(t[l], keys, c.world);
    }
    std::vector<Table> res = exchange_parts(c, parts, t);
    hand_out(res, outs);
  });
}

int dfgpu_exchange_hash_stream_open(dfgpu_comm_t h, const dfgpu_table_t* inputs, const int* key_cols, int nkeys, int n_chunks, dfgpu_exchange_stream_t* out) {
  return guarded([&] {
    require_init();
    DFGPU_CHECK(h && inputs && out && key_cols && nkeys >= 1 && n_chunks >= 1, "dfgpu_exchange_hash_stream_open: bad arguments");
    Comm& c = idle_comm(h, "dfgpu_exchange_hash_stream_open");
    DFGPU_CHECK(c.world <= 64, "dfgpu_exchange_hash supports up to 64 ranks");
    auto s = std::make_unique<HashStream>();
    s->c = &c;
    s->inputs = local_inputs(c, inputs);
    unify_dictionaries(c, s->inputs);
    s->keys.assign(key_cols, key_cols + nkeys);
    s->devices = c.devices;
    // Utf8 columns travel whole (no zero-copy row ranges of them): ONE chunk.  Every rank sees the same schema, so every rank decides the same.
    bool can_cut = true;
    for (const Table& t : s->inputs) can_cut = can_cut && sliceable(t);
    s->n_chunks = can_cut ? n_chunks : 1;
    (void)rt().stream;   // (the calling thread keeps the device's first stream: the workers get streams of their own)
    HashStream* raw = s.get();
    c.open_streams++;
    s->partitioner = std::thread(hash_stream_partition, raw);
    s->exchanger = std::thread(hash_stream_exchange, raw);
    *out = reinterpret_cast<dfgpu_exchange_stream_t>(s.release());
  });
}
int dfgpu_exchange_hash_stream_next(dfgpu_exchange_stream_t sh, dfgpu_table_t* outs, int* done) {
  return guarded([&] {
    DFGPU_CHECK(sh && outs && done, "dfgpu_exchange_hash_stream_next: bad arguments");
    HashStream& s = *reinterpret_cast<HashStream*>(sh);
    std::vector<Table> res;
    {
      std::unique_lock<std::mutex> lk(s.mu);
      s.cv.wait(lk, [&] { return !s.received.empty() || s.received_done; });
      if (s.received.empty()) {
        DFGPU_CHECK(s.error.empty(), "exchange stream: " + s.error);
        *done = 1;
        return;
      }
      res = std::move(s.received.front());
      s.received.pop_front();
      s.cv.notify_all();
    }
    *done = 0;
    s.handed++;
    hand_out(res, outs);
  });
}
int dfgpu_exchange_hash_stream_free(dfgpu_exchange_stream_t sh) {
  return guarded([&] {
    if (!sh) return;
    HashStream* s = reinterpret_cast<HashStream*>(sh);
    {
      std::lock_guard<std::mutex> lk(s->mu);
      s->cancel = true;
      s->cv.notify_all();
    }
    if (s->partitioner.joinable()) s->partitioner.join();
    if (s->exchanger.joinable()) s->exchanger.join();
    s->c->open_streams--;
    delete s;
  });
}

int dfgpu_exchange_broadcast(dfgpu_comm_t h, const dfgpu_table_t* inputs, dfgpu_table_t* outs) {
  return guarded([&] {
    require_init();
    DFGPU_CHECK(h && inputs && outs, "dfgpu_exchange_broadcast: bad arguments");
    Comm& c = idle_comm(h, "dfgpu_exchange_broadcast");
    std::vector<Table> t = local_inputs(c, inputs);
    unify_dictionaries(c, t);
    std::vector<std::vector<Table>> parts(c.n_local());
    for (int l = 0; l < c.n_local(); l++) parts[l].assign(c.world, t[l]);  // every peer receives the whole local table
    std::vector<Table> res = exchange_parts(c, parts, t);
    hand_out(res, outs);
  });
}

// The exchange of a distributed ORDER BY (SURVEY §8e "all-to-all for sample-sort"; SortPreservingMergeExec over partitions that sit
// on different GPUs, sorts/sort_preserving_merge.rs:91): every rank samples its rows' FIRST sort key, the samples of all ranks give
// world - 1 splitters (quantiles), and a row goes to the rank whose key range holds its key — rank 0 the smallest keys (the largest
// under DESC), NULL keys first or last as the sort options say.  Rows with equal first keys land on ONE rank, so after a local sort
// by the full key list the ranks' outputs, read in rank order, are the globally sorted result: no rank ever holds or re-sorts
// everything.  Integer-like first key (Int32 / Int64 / Date32 / UInt8 / UInt32).
int dfgpu_exchange_range(dfgpu_comm_t h, const dfgpu_table_t* inputs, int key_col, int descending, int nulls_first, dfgpu_table_t* outs) {
  return guarded([&] {
    require_init();
    DFGPU_CHECK(h && inputs && outs, "dfgpu_exchange_range: bad arguments");
    Comm& c = idle_comm(h, "dfgpu_exchange_range");
    DFGPU_CHECK(c.world <= MAX_RANGE_RANKS, "dfgpu_exchange_range: too many ranks");
    std::vector<Table> t = local_inputs(c, inputs);
    unify_dictionaries(c, t);
    const int L = c.n_local();
    constexpr int S = 1024;
    std::vector<std::vector<uint8_t>> mine((size_t)L);
    for (int l = 0; l < L; l++) {
      use_device(c.devices[l]);
      DFGPU_CHECK(key_col >= 0 && key_col < (int)t[l].cols.size(), "dfgpu_exchange_range: key column out of range");
      const Column& kc = t[l].cols[(size_t)key_col];
      DFGPU_CHECK(is_integer_like(kc.field.type) && kc.field.type != DFGPU_UINT64 && !kc.dict, "dfgpu_exchange_range: the first sort key must be an integer-like column");
      const int64_t n = t[l].nrows;
      std::vector<long long> samples;
      if (n > 0) {
        BufPtr d = make_buf((size_t)S * 8), cnt = make_zero_buf(4);
        const KeyCol k{kc.ptr(), kc.valid_words(), kc.field.type, type_width(kc.field.type)};
        k_sample_keys<<<S / BLOCK, BLOCK, 0, rt().stream>>>(k, n, std::max<int64_t>(1, n / S), S, d->as<long long>(), cnt->as<int>());
        DFGPU_HIP(hipGetLastError());
        int got = 0;
        d2h(&got, cnt->ptr, 4);
        samples.resize((size_t)got);
        if (got) d2h(samples.data(), d->ptr, (size_t)got * 8);
      }
      mine[(size_t)l].assign((const uint8_t*)samples.data(), (const uint8_t*)samples.data() + samples.size() * 8);
    }
    const std::vector<std::vector<uint8_t>> all = allgather_blobs(c, mine);
    std::vector<long long> pool;
    for (const auto& b : all) {
      const size_t m = b.size() / 8;
      const size_t at = pool.size();
      pool.resize(at + m);
      if (m) std::memcpy(pool.data() + at, b.data(), m * 8);
    }
    std::sort(pool.begin(), pool.end());
    Splitters sp{};
    sp.n = c.world - 1;
    for (int q = 0; q < sp.n; q++) sp.s[q] = pool.empty() ? 0 : pool[std::min(pool.size() - 1, (size_t)((q + 1) * pool.size() / (size_t)c.world))];
    // ---- every local table split by destination
    std::vector<std::vector<Table>> parts((size_t)L);
    for (int l = 0; l < L; l++) {
      use_device(c.devices[l]);
      const int64_t n = t[l].nrows, n_words = (n + 63) / 64;
      std::vector<int> allc(t[l].cols.size());
      for (size_t i = 0; i < allc.size(); i++) allc[i] = (int)i;
      BufPtr masks = make_buf((size_t)std::max<int64_t>(1, n_words * c.world) * 8);
      if (n) {
        const Column& kc = t[l].cols[(size_t)key_col];
        const KeyCol k{kc.ptr(), kc.valid_words(), kc.field.type, type_width(kc.field.type)};
        // NULLs: first in the output -> the rank that comes first; the key order among ranks is reversed under DESC
        const int null_dest = nulls_first ? 0 : c.world - 1;
        ProfileScope ps("exchange_range_masks", n * k.width);
        k_range_masks<<<grid_for(n_words, BLOCK / WAVE), BLOCK, 0, rt().stream>>>(k, n, sp, c.world, descending, null_dest, masks->as<uint64_t>(), n_words);
        DFGPU_HIP(hipGetLastError());
      }
      for (int d = 0; d < c.world; d++) parts[(size_t)l].push_back(compact_table(t[l], allc, masks->as<uint64_t>() + (int64_t)d * n_words, nullptr));
    }
    std::vector<Table> res = exchange_parts(c, parts, t);
    hand_out(res, outs);
  });
}

// CollectLeft with build-side emission across ranks: OR of the visited marks (and null-aware flags) of the replicated join tables
int dfgpu_exchange_join_visited(dfgpu_comm_t h, const dfgpu_join_t* joins) {
  return guarded([&] {
    require_init();
    DFGPU_CHECK(h && joins, "null argument");
    Comm& c = *reinterpret_cast<Comm*>(h);
    const int L = c.n_local();
    std::vector<std::vector<uint8_t>> mine((size_t)L);
    for (int l = 0; l < L; l++) {
      DFGPU_CHECK(joins[l] != nullptr, "null join table");
      mine[(size_t)l] = join_visited_export(joins[l]);
    }
    const std::vector<std::vector<uint8_t>> all = allgather_blobs(c, mine);
    std::vector<uint8_t> merged = all[0];
    for (int r = 1; r < c.world; r++) {
      DFGPU_CHECK(all[(size_t)r].size() == merged.size(), "join visited merge: the ranks' build sides differ in size (the build side must be replicated)");
      for (size_t i = 0; i < merged.size(); i++) merged[i] |= all[(size_t)r][i];
    }
    for (int l = 0; l < L; l++) join_visited_merge(joins[l], merged.data(), merged.size());
    std::lock_guard<std::mutex> lk(c.mu);
    c.stats.collectives += 1;
  });
}

int dfgpu_exchange_broadcast_pruned(dfgpu_comm_t h, const dfgpu_table_t* builds, int build_key, const dfgpu_table_t* probes, int probe_key,
                                    dfgpu_table_t* outs) {
  return guarded([&] {
    require_init();
    DFGPU_CHECK(h && builds && probes && outs, "dfgpu_exchange_broadcast_pruned: bad arguments");
    Comm& c = idle_comm(h, "dfgpu_exchange_broadcast_pruned");
    const int L = c.n_local(), W = c.world;
    std::vector<Table> b = local_inputs(c, builds);
    unify_dictionaries(c, b);
    // every rank's closed key ranges of both sides: {probe lo, hi, any, build lo, hi, any}
    std::vector<std::vector<uint8_t>> meta(L, std::vector<uint8_t>(48));
    std::vector<ColStats> bstats(L);
    for (int l = 0; l < L; l++) {
      Table* pt = unwrap(probes[l]);
      Table* bt = unwrap(builds[l]);
      DFGPU_CHECK(build_key >= 0 && build_key < (int)bt->cols.size() && probe_key >= 0 && probe_key < (int)pt->cols.size(), "key column out of range");
      DFGPU_CHECK(!bt->cols[build_key].dict && !pt->cols[probe_key].dict, "pruned broadcast: integer key columns only");
      const ColStats ps = column_stats(pt->cols[probe_key], pt->nrows);
      bstats[l] = column_stats(bt->cols[build_key], bt->nrows);
      const long long v[6] = {ps.min, ps.max, ps.valid > 0, bstats[l].min, bstats[l].max, bstats[l].valid > 0};
      std::memcpy(meta[l].data(), v, 48);
    }
    const std::vector<uint8_t> all = allgather_meta(c, meta, 48);
    auto range = [&](int r, int side, long long& lo, long long& hi) {
      long long v[6];
      std::memcpy(v, all.data() + (size_t)r * 48, 48);
      lo = v[side * 3];
      hi = v[side * 3 + 1];
      return v[side * 3 + 2] != 0;
    };
    std::vector<std::vector<Table>> parts(L);
    for (int l = 0; l < L; l++) {
      use_device(c.devices[l]);
      const Column& kc = b[l].cols[build_key];
      std::vector<int> allc(b[l].cols.size());
      for (size_t i = 0; i < allc.size(); i++) allc[i] = (int)i;
      for (int p = 0; p < W; p++) {
        // what destination p can match lies inside [min, max] of ITS probe keys
        long long plo, phi;
        const bool pany = range(p, 0, plo, phi);
        const long long lo = std::max(plo, (long long)bstats[l].min), hi = std::min(phi, (long long)bstats[l].max);
        if (!pany || bstats[l].valid == 0 || lo > hi) {
          Table none;  // nothing of this shard can match on rank p
          none.nrows = 0;
          for (const Column& col : b[l].cols) none.cols.push_back(alloc_like(col, 0));
          parts[l].push_back(std::move(none));
        } else if (lo == bstats[l].min && hi == bstats[l].max && !kc.validity) {
          parts[l].push_back(b[l]);  // the bounds cover the whole shard: a view, no filter pass
        } else {
          BufPtr mask = make_buf(bitmap_bytes(b[l].nrows));
          const int g = grid_for((b[l].nrows + 63) / 64, BLOCK / WAVE);
          switch (kc.field.type) {
            case DFGPU_INT64: k_range_mask<int64_t><<<g, BLOCK, 0, rt().stream>>>((const int64_t*)kc.ptr(), kc.valid_words(), b[l].nrows, lo, hi, mask->as<uint64_t>()); break;
            case DFGPU_INT32: case DFGPU_DATE32: k_range_mask<int32_t><<<g, BLOCK, 0, rt().stream>>>((const int32_t*)kc.ptr(), kc.valid_words(), b[l].nrows, lo, hi, mask->as<uint64_t>()); break;
            case DFGPU_UINT32: k_range_mask<uint32_t><<<g, BLOCK, 0, rt().stream>>>((const uint32_t*)kc.ptr(), kc.valid_words(), b[l].nrows, lo, hi, mask->as<uint64_t>()); break;
            case DFGPU_UINT8: k_range_mask<uint8_t><<<g, BLOCK, 0, rt().stream>>>((const uint8_t*)kc.ptr(), kc.valid_words(), b[l].nrows, lo, hi, mask->as<uint64_t>()); break;
            default: throw Error("pruned broadcast: integer key columns only");
          }
          DFGPU_HIP(hipGetLastError());
          parts[l].push_back(compact_table(b[l], allc, mask->as<uint64_t>(), nullptr));
        }
      }
    }
    std::vector<Table> res = exchange_parts(c, parts, b);
    hand_out(res, outs);
  });
}

}  // extern "C"
