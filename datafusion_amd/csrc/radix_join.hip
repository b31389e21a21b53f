// radix_join.hip — the general hash join as LDS-staged hash-table partitions (BASELINE north_star).
//
// Reference semantics: JoinHashMap::update_from_iter + get_matched_indices (joins/join_hash_map.rs:307-338,389-484),
// chains of equal hashes (joins/chain.rs:29-69), key re-check equal_rows_arr (joins/utils.rs:2191-2260).  The reference keeps
// ONE table over the whole build side and walks it with dependent random loads; on MI355X a random access costs a 128-byte
// line of HBM (profiles/r2_fetch_calib.md) once the table outgrows the 256 MiB Infinity Cache, so here both sides are radix
// partitioned on the top bits of a mixed key until a build partition fits the LDS of one workgroup, and every partition
// pair is joined with a chained table that lives in LDS — the table walk never leaves the CU:
//   1. records  : (mixed key u64, row id u32) per row whose key can match (NULL keys dropped under NullEqualsNothing).
//                 One integer key column <= 64 bits: mixed key = fmix64(value) — a bijection, so equal mixed keys ARE equal
//                 keys and no re-check is needed.  Anything else (several columns, Decimal128, NULL == NULL): mixed key =
//                 the 64-bit row hash (create_hashes semantics, device.hpp hash_row) and hash-equal pairs are re-checked on
//                 the real key columns (equal_rows_arr), which is also what the `force_hash_collisions` test mode exercises.
//   2. partition: stable LSD radix passes (sort.hip k_rs_scatter2: ballot ranking, LDS-staged contiguous runs) over the top
//                 B bits, B chosen so that a build partition averages 1024 rows; partition p = key >> (64 - B).
//   3. join     : a task = (partition, <= 8192 probe rows of it).  The workgroup loads the build partition into LDS in chunks
//                 of 2048 rows (keys, row ids, u32 heads + u16 next chains: 44 KB, 3 workgroups per CU), streams its probe rows
//                 against it and counts matches; task counts are prefix-summed and a second identical walk emits the
//                 (build row, probe row) pairs at exact offsets (no atomics, exact allocation for M:N outputs).  Partitions
//                 that exceed a chunk (skew, duplicates, forced collisions) take several chunks per task — correct at any skew.
// The pairs feed the same per-JoinType finishing as the JoinFilter path (visited bytes, probe hit counts, unmatched rows),
// so all ten join types, NullEqualsNull and residual filters work on top of it.  Output order is partition order: valid where
// the plan does not observe HashJoinExec's probe-side order (it is chosen by table_mode 4, or by `auto` under probe_mode 4).
#include <algorithm>
#include <cstdlib>
#include <functional>

#include "device.hpp"
#include "internal.hpp"

namespace dfgpu {

static bool force_fused_off() { return !option_on("join.radix_fused_emit", true); }   // (A/B switch: the pairs + gathers of round 5)
constexpr int RJ_CAP = 2048;     // build rows per LDS chunk
constexpr int RJ_HEADS = 4096;   // chain heads (power of two, 2 x RJ_CAP)
constexpr int RJ_TASK_ROWS = 8192;
constexpr int RJ_STAGE = 512;    // matches of one 256-row probe tile staged in LDS before they are written (8 KB: 3 workgroups per CU still fit)

struct RadixTask {
  uint32_t part;
  uint32_t rows;
  uint64_t q0;
};

// ---------------------------------------------------------------------------------------------- records
// EXACT: ks.n == 1, integer key.  `mask` / `prefix` (optional): rows that take part and their dense positions.
template <bool EXACT>
__global__ __launch_bounds__(BLOCK) void k_rj_records(KeySet ks, int64_t n, int force_collisions, const uint64_t* __restrict__ mask, const uint64_t* __restrict__ prefix,
                                                     uint64_t* __restrict__ key, uint32_t* __restrict__ rid) {
  const int64_t n_words = (n + 63) >> 6;
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * BLOCK) >> 6;
  for (int64_t w = wave; w < n_words; w += n_waves) {
    const int64_t i = (w << 6) + lane_id();
    uint64_t m = ~0ull;
    const int64_t rem = n - (w << 6);
    if (rem < 64) m = (~0ull) >> (64 - rem);
    if (mask) m &= mask[w];
    if (!((m >> lane_id()) & 1ull)) continue;
    const int64_t d = mask ? (int64_t)(prefix[w] + mbcnt(m)) : i;
    uint64_t k;
    if (EXACT) {
      uint64_t lo, hi;
      load_words(ks.c[0], i, lo, hi);
      k = fmix64(lo);
    } else {
      bool any_null;
      k = hash_row(ks, i, SEED_JOIN, any_null);
    }
    if (force_collisions) k = 0;
    key[d] = k;
    rid[d] = (uint32_t)i;
  }
}
// AND of the key columns' validity words (rows with a NULL key never match under NullEqualsNothing)
__global__ __launch_bounds__(BLOCK) void k_rj_valid_mask(KeySet ks, int64_t n_words, uint64_t* __restrict__ mask) {
  for (int64_t w = (int64_t)blockIdx.x * BLOCK + threadIdx.x; w < n_words; w += (int64_t)gridDim.x * BLOCK) {
    uint64_t m = ~0ull;
    for (int c = 0; c < ks.n; c++)
      if (ks.c[c].valid) m &= ks.c[c].valid[w];
    mask[w] = m;
  }
}
// first position of every non-empty partition of a sorted record array (empty ones are filled in on the host)
__global__ __launch_bounds__(BLOCK) void k_rj_starts(const uint64_t* __restrict__ key, int64_t n, int shift, unsigned long long* __restrict__ starts) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    const uint64_t p = shift >= 64 ? 0ull : key[i] >> shift;
    const uint64_t q = i == 0 ? ~0ull : (shift >= 64 ? 0ull : key[i - 1] >> shift);
    if (p != q) starts[p] = (unsigned long long)i;
  }
}

struct RadixSide {
  BufPtr key, rid, starts;
  std::vector<uint64_t> h_starts;  // P + 1 entries
  int64_t n = 0;                   // records (rows that take part)
};
struct RadixTable {
  int bits = 0;
  bool exact = false;
  RadixSide build;
};

static KeySet rj_keyset(const Table& t, const std::vector<int>& cols) {
  KeySet ks{};
  DFGPU_CHECK((int)cols.size() <= MAX_KEYS && !cols.empty(), "bad number of join key columns");
  ks.n = (int)cols.size();
  for (int i = 0; i < ks.n; i++) {
    DFGPU_CHECK(cols[i] >= 0 && cols[i] < (int)t.cols.size(), "join key column index out of range");
    const Column& c = t.cols[cols[i]];
    DFGPU_CHECK(c.field.type != DFGPU_BOOL, "Boolean join keys are not supported on the GPU path");
    ks.c[i] = KeyCol{c.ptr(), c.valid_words(), c.field.type, type_width(c.field.type)};
  }
  return ks;
}

// records of one side, partitioned by the top `bits` bits of the mixed key
static RadixSide rj_partition(const Table& t, const std::vector<int>& key_cols, int bits, bool exact, bool null_equals_null, bool force_collisions, const char* what) {
  Runtime& r = rt();
  RadixSide s;
  const int64_t n = t.nrows;
  DFGPU_CHECK(n < 0xFFFFFFFFll, "the radix join addresses rows with 32 bits");
  const int64_t P = (int64_t)1 << bits;
  KeySet ks = rj_keyset(t, key_cols);
  bool nullable = false;
  for (int i = 0; i < ks.n; i++) nullable |= ks.c[i].valid != nullptr;
  const int64_t n_words = (n + 63) / 64;
  BufPtr mask, prefix;
  s.n = n;
  // NULL keys match nothing under NullEqualsNothing; an EXACT table is only built when NULL == NULL cannot match either
  // (no NULL build key), so NULL probe keys are dropped there too (the value under a NULL is arbitrary)
  if (nullable && (!null_equals_null || exact) && n) {
    mask = make_buf(bitmap_bytes(n));
    prefix = make_buf((size_t)(n_words + 1) * 8);
    k_rj_valid_mask<<<grid_for(n_words, BLOCK), BLOCK, 0, r.stream>>>(ks, n_words, mask->as<uint64_t>());
    scan_mask_popcounts(mask->as<uint64_t>(), nullptr, n, prefix->as<uint64_t>());
    s.n = (int64_t)read_u64(prefix->as<uint64_t>() + n_words);
  }
  s.key = make_buf((size_t)std::max<int64_t>(s.n, 1) * 8);
  s.rid = make_buf((size_t)std::max<int64_t>(s.n, 1) * 4);
  int64_t key_bytes = 0;
  for (int i = 0; i < ks.n; i++) key_bytes += n * ks.c[i].width;
  if (n) {
    ProfileScope ps(what[0] == 'b' ? "radix_join_build_records" : "radix_join_probe_records", key_bytes + s.n * 12);
    const int g = grid_for(n_words, BLOCK / WAVE);
    const uint64_t* mk = mask ? mask->as<uint64_t>() : nullptr;
    const uint64_t* pf = prefix ? prefix->as<uint64_t>() : nullptr;
    if (exact) k_rj_records<true><<<g, BLOCK, 0, r.stream>>>(ks, n, force_collisions, mk, pf, s.key->as<uint64_t>(), s.rid->as<uint32_t>());
    else k_rj_records<false><<<g, BLOCK, 0, r.stream>>>(ks, n, force_collisions, mk, pf, s.key->as<uint64_t>(), s.rid->as<uint32_t>());
    DFGPU_HIP(hipGetLastError());
  }
  if (bits > 0 && s.n > 1) radix_sort_pairs(s.key, s.rid, s.n, 64 - bits, bits);
  // partition boundaries
  s.h_starts.assign((size_t)P + 1, ~0ull);
  if (s.n) {
    BufPtr st = make_buf((size_t)(P + 1) * 8);
    DFGPU_HIP(hipMemsetAsync(st->ptr, 0xFF, (size_t)(P + 1) * 8, r.stream));
    k_rj_starts<<<grid_for(s.n, BLOCK), BLOCK, 0, r.stream>>>(s.key->as<uint64_t>(), s.n, 64 - bits, st->as<unsigned long long>());
    DFGPU_HIP(hipGetLastError());
    d2h(s.h_starts.data(), st->ptr, (size_t)(P + 1) * 8);
  }
  s.h_starts[(size_t)P] = (uint64_t)s.n;
  for (int64_t p = P - 1; p >= 0; p--)
    if (s.h_starts[(size_t)p] == ~0ull) s.h_starts[(size_t)p] = s.h_starts[(size_t)p + 1];
  s.starts = make_buf((size_t)(P + 1) * 8);
  DFGPU_HIP(hipMemcpyAsync(s.starts->ptr, s.h_starts.data(), (size_t)(P + 1) * 8, hipMemcpyHostToDevice, r.stream));
  DFGPU_HIP(hipStreamSynchronize(r.stream));
  return s;
}

// ---------------------------------------------------------------------------------------------- the LDS join
struct RjVerify {
  KeySet bkeys, pkeys;
  int null_equals_null;
};
// The output columns of an INNER join written by the emit walk itself (round 6): a pair's build row and probe row are in registers the
// moment the pair is found — its columns are gathered and written right there, and the (build row, probe row) list — 16 bytes a pair
// written, then read back by one gather per column — never exists (M:N at SF100 sizes, 300 M pairs: 4.8 GB of pairs each way).  A column
// that IS the join key (EXACT records: mixed key = fmix64(value), a bijection) comes out of the record's key by unfmix64: no access at all.
constexpr int RJ_MAX_COLS = 12;
struct RjCols {
  const void* src[RJ_MAX_COLS];
  void* dst[RJ_MAX_COLS];
  int width[RJ_MAX_COLS];
  int n_build;   // the first n_build entries read the build row, the others the probe row
  int n;
  unsigned key_cols;   // bit c: entry c is the join key column (EXACT): the value is unfmix64(record key)
};
__device__ __forceinline__ uint64_t unfmix64(uint64_t x) {   // fmix64's inverse (the multipliers' inverses mod 2^64; x ^= x >> 33 undoes itself)
  x ^= x >> 33; x *= 0x9cb4b2f8129337dbULL;
  x ^= x >> 33; x *= 0x4f74430c22a54005ULL;
  x ^= x >> 33;
  return x;
}
__device__ __forceinline__ void rj_emit_row(const RjCols& cols, uint64_t key, uint32_t brow, uint32_t prow, unsigned long long o) {
  for (int c = 0; c < cols.n; c++) {
    const int w = cols.width[c];
    if ((cols.key_cols >> c) & 1u) {
      const uint64_t v = unfmix64(key);
      if (w == 8) reinterpret_cast<uint64_t*>(cols.dst[c])[o] = v;
      else if (w == 4) reinterpret_cast<uint32_t*>(cols.dst[c])[o] = (uint32_t)v;
      else reinterpret_cast<uint8_t*>(cols.dst[c])[o] = (uint8_t)v;
      continue;
    }
    const int64_t s = c < cols.n_build ? (int64_t)brow : (int64_t)prow;
    switch (w) {
      case 16: reinterpret_cast<uint4*>(cols.dst[c])[o] = reinterpret_cast<const uint4*>(cols.src[c])[s]; break;
      case 8: reinterpret_cast<uint64_t*>(cols.dst[c])[o] = reinterpret_cast<const uint64_t*>(cols.src[c])[s]; break;
      case 4: reinterpret_cast<uint32_t*>(cols.dst[c])[o] = reinterpret_cast<const uint32_t*>(cols.src[c])[s]; break;
      default: reinterpret_cast<uint8_t*>(cols.dst[c])[o] = reinterpret_cast<const uint8_t*>(cols.src[c])[s]; break;
    }
  }
}
// EMIT: 0 = count the task's matches, 1 = write (build row, probe row) pairs, 2 = write the output columns (RjCols)
template <bool EXACT, int EMIT>
__global__ __launch_bounds__(BLOCK) void k_rj_join(const uint64_t* __restrict__ bkey, const uint32_t* __restrict__ brid, const uint64_t* __restrict__ bstart,
                                                  const uint64_t* __restrict__ pkey, const uint32_t* __restrict__ prid, const RadixTask* __restrict__ tasks,
                                                  int64_t n_tasks, RjVerify v, unsigned long long* __restrict__ task_counts,
                                                  const uint64_t* __restrict__ task_off, int64_t* __restrict__ out_b, int64_t* __restrict__ out_p, RjCols cols = RjCols{}) {
  __shared__ uint64_t s_key[RJ_CAP];
  __shared__ uint32_t s_rid[RJ_CAP];
  __shared__ uint32_t s_head[RJ_HEADS];
  __shared__ uint16_t s_next[RJ_CAP];
  __shared__ unsigned long long s_wtot[BLOCK / WAVE];
  __shared__ uint64_t s_ok[EMIT ? RJ_STAGE : 1];   // the emit walks' stage: a tile's matches (record key, build row, probe row)
  __shared__ uint32_t s_ob[EMIT ? RJ_STAGE : 1], s_op[EMIT ? RJ_STAGE : 1];
  const unsigned lane = lane_id();
  const int wave = threadIdx.x >> 6;
  for (int64_t t = blockIdx.x; t < n_tasks; t += gridDim.x) {
    const RadixTask task = tasks[t];
    const uint64_t b0 = bstart[task.part], b1 = bstart[task.part + 1];
    unsigned long long mine = 0;                                   // COUNT: this thread's matches over the whole task
    unsigned long long run = EMIT ? task_off[t] : 0ull;             // EMIT: next free output position of the task (uniform)
    for (uint64_t c0 = b0; c0 < b1; c0 += RJ_CAP) {
      const int nbk = (int)((b1 - c0) < (uint64_t)RJ_CAP ? (b1 - c0) : (uint64_t)RJ_CAP);
      for (int i = threadIdx.x; i < RJ_HEADS; i += BLOCK) s_head[i] = 0;
      __syncthreads();
      for (int i = threadIdx.x; i < nbk; i += BLOCK) {
        const uint64_t k = bkey[c0 + i];
        s_key[i] = k;
        s_rid[i] = brid[c0 + i];
        const uint32_t old = atomicExch(&s_head[(uint32_t)k & (RJ_HEADS - 1)], (uint32_t)i + 1u);
        s_next[i] = (uint16_t)old;
      }
      __syncthreads();
      for (uint32_t r0 = 0; r0 < task.rows; r0 += BLOCK) {
        const uint32_t j = r0 + threadIdx.x;
        const bool in = j < task.rows;
        uint64_t k = 0;
        uint32_t pr = 0;
        if (in) {
          k = pkey[task.q0 + j];
          pr = prid[task.q0 + j];
        }
        uint32_t cnt = 0;
        if (in) {
          uint32_t cur = s_head[(uint32_t)k & (RJ_HEADS - 1)];
          while (cur) {
            if (s_key[cur - 1] == k && (EXACT || keys_equal(v.bkeys, (int64_t)s_rid[cur - 1], v.pkeys, (int64_t)pr, v.null_equals_null != 0, true))) cnt++;
            cur = s_next[cur - 1];
          }
        }
        if (!EMIT) {
          mine += cnt;
          continue;
        }
        // exclusive offsets of this tile's matches: wave scan + wave totals
        const unsigned long long inc = wave_inclusive_sum<unsigned long long>((unsigned long long)cnt);
        if (lane == 63) s_wtot[wave] = inc;
        __syncthreads();
        unsigned long long base = run, tot = 0;
#pragma unroll
        for (int w = 0; w < BLOCK / WAVE; w++) {
          if (w < wave) base += s_wtot[w];
          tot += s_wtot[w];
        }
        unsigned long long o = base + inc - cnt;
        if (tot <= (unsigned long long)RJ_STAGE) {
          // the tile's matches are listed in LDS first, then every OUTPUT row gets a thread: consecutive threads write consecutive rows
          // (a thread writing its own two or three matches one after the other left 8-byte stores 16-24 bytes apart: 5.6 ms for 300 M
          // pairs against 3.x staged) and the gathers of a row's columns are issued by as many threads as there are rows
          if (cnt) {
            uint32_t lo = (uint32_t)(o - run);
            uint32_t cur = s_head[(uint32_t)k & (RJ_HEADS - 1)];
            while (cur) {
              if (s_key[cur - 1] == k && (EXACT || keys_equal(v.bkeys, (int64_t)s_rid[cur - 1], v.pkeys, (int64_t)pr, v.null_equals_null != 0, true))) {
                s_ok[lo] = k;
                s_ob[lo] = s_rid[cur - 1];
                s_op[lo] = pr;
                lo++;
              }
              cur = s_next[cur - 1];
            }
          }
          __syncthreads();
          for (uint32_t q = threadIdx.x; q < (uint32_t)tot; q += BLOCK) {
            if (EMIT == 2) {
              rj_emit_row(cols, s_ok[q], s_ob[q], s_op[q], run + q);
            } else {
              out_b[run + q] = (int64_t)s_ob[q];
              out_p[run + q] = (int64_t)s_op[q];
            }
          }
        } else if (cnt) {   // (more matches than the stage holds — heavy duplicates: every thread writes its own)
          uint32_t cur = s_head[(uint32_t)k & (RJ_HEADS - 1)];
          while (cur) {
            if (s_key[cur - 1] == k && (EXACT || keys_equal(v.bkeys, (int64_t)s_rid[cur - 1], v.pkeys, (int64_t)pr, v.null_equals_null != 0, true))) {
              if (EMIT == 2) {
                rj_emit_row(cols, k, s_rid[cur - 1], pr, o);
              } else {
                out_b[o] = (int64_t)s_rid[cur - 1];
                out_p[o] = (int64_t)pr;
              }
              o++;
            }
            cur = s_next[cur - 1];
          }
        }
        run += tot;
        __syncthreads();  // s_wtot and the stage are reused by the next tile
      }
      __syncthreads();  // the next chunk overwrites the table
    }
    if (!EMIT) {
      mine = wave_sum<unsigned long long>(mine);
      if (lane == 0) s_wtot[wave] = mine;
      __syncthreads();
      if (threadIdx.x == 0) {
        unsigned long long tot = 0;
        for (int w = 0; w < BLOCK / WAVE; w++) tot += s_wtot[w];
        task_counts[t] = tot;
      }
      __syncthreads();
    }
  }
}

// ---------------------------------------------------------------------------------------------- host
std::shared_ptr<RadixTable> radix_join_build(const Table& build, const std::vector<int>& key_cols, bool null_equals_null, bool force_collisions) {
  auto rt_ = std::make_shared<RadixTable>();
  KeySet ks = rj_keyset(build, key_cols);
  bool nullable = false;
  for (int i = 0; i < ks.n; i++) nullable |= ks.c[i].valid != nullptr;
  rt_->exact = ks.n == 1 && is_integer_like(ks.c[0].type) && !(null_equals_null && nullable) && !force_collisions;
  // ~1024 build rows per partition on average, up to three 6-bit passes (268 M build rows): a partition that needs several LDS
  // chunks per task costs far more than a third pass over HBM (profiles/r2_radix_sweep.md)
  const int max_bits = 18;
  int bits = 0;
  while (bits < max_bits && ((int64_t)1024 << bits) < build.nrows) bits++;
  if (force_collisions) bits = 0;
  rt_->bits = bits;
  rt_->build = rj_partition(build, key_cols, bits, rt_->exact, null_equals_null, force_collisions, "build");
  return rt_;
}

int64_t radix_join_table_bytes(const RadixTable& t) { return t.build.n * 12 + (((int64_t)1 << t.bits) + 1) * 8; }
int radix_join_bits(const RadixTable& t) { return t.bits; }

// inner pairs (build row, probe row) of all key-equal rows, in partition order — or, with `columns` (an INNER join's output columns: a
// callback that allocates them for m rows and fills in the destinations), the joined rows themselves
static void radix_join_run(const RadixTable& t, const Table& build, const std::vector<int>& build_keys, const Table& probe, const std::vector<int>& probe_keys,
                           bool null_equals_null, bool force_collisions, BufPtr& out_b, BufPtr& out_p, int64_t& m, const std::function<RjCols(int64_t)>* columns) {
  Runtime& r = rt();
  m = 0;
  out_b = make_buf(8);
  out_p = make_buf(8);
  if (t.build.n == 0 || probe.nrows == 0) return;
  // the probe key types must allow the build side's choice of record key
  KeySet pk = rj_keyset(probe, probe_keys), bk = rj_keyset(build, build_keys);
  RadixSide ps = rj_partition(probe, probe_keys, t.bits, t.exact, null_equals_null, force_collisions, "probe");
  if (ps.n == 0) return;
  // tasks: every partition with rows on both sides, its probe rows cut into pieces
  std::vector<RadixTask> tasks;
  const int64_t P = (int64_t)1 << t.bits;
  for (int64_t p = 0; p < P; p++) {
    const uint64_t b0 = t.build.h_starts[(size_t)p], b1 = t.build.h_starts[(size_t)p + 1], q0 = ps.h_starts[(size_t)p], q1 = ps.h_starts[(size_t)p + 1];
    if (b0 == b1 || q0 == q1) continue;
    for (uint64_t q = q0; q < q1; q += RJ_TASK_ROWS) tasks.push_back(RadixTask{(uint32_t)p, (uint32_t)std::min<uint64_t>(RJ_TASK_ROWS, q1 - q), q});
  }
  if (tasks.empty()) return;
  const int64_t nt = (int64_t)tasks.size();
  BufPtr d_tasks = make_buf((size_t)nt * sizeof(RadixTask));
  DFGPU_HIP(hipMemcpyAsync(d_tasks->ptr, tasks.data(), (size_t)nt * sizeof(RadixTask), hipMemcpyHostToDevice, r.stream));
  BufPtr d_counts = make_buf((size_t)nt * 8), d_off = make_buf((size_t)nt * 8);
  RjVerify v{bk, pk, null_equals_null ? 1 : 0};
  const int grid = (int)std::min<int64_t>(nt, (int64_t)r.num_cus * 12);
  const int64_t rec_bytes = (t.build.n + ps.n) * 12;
  {
    ProfileScope psc("radix_join_count", rec_bytes);
    if (t.exact)
      k_rj_join<true, 0><<<grid, BLOCK, 0, r.stream>>>(t.build.key->as<uint64_t>(), t.build.rid->as<uint32_t>(), t.build.starts->as<uint64_t>(), ps.key->as<uint64_t>(),
                                                           ps.rid->as<uint32_t>(), d_tasks->as<RadixTask>(), nt, v, d_counts->as<unsigned long long>(), nullptr, nullptr, nullptr);
    else
      k_rj_join<false, 0><<<grid, BLOCK, 0, r.stream>>>(t.build.key->as<uint64_t>(), t.build.rid->as<uint32_t>(), t.build.starts->as<uint64_t>(), ps.key->as<uint64_t>(),
                                                            ps.rid->as<uint32_t>(), d_tasks->as<RadixTask>(), nt, v, d_counts->as<unsigned long long>(), nullptr, nullptr, nullptr);
    DFGPU_HIP(hipGetLastError());
  }
  std::vector<uint64_t> counts((size_t)nt), off((size_t)nt);
  d2h(counts.data(), d_counts->ptr, (size_t)nt * 8);
  uint64_t tot = 0;
  for (int64_t i = 0; i < nt; i++) {
    off[(size_t)i] = tot;
    tot += counts[(size_t)i];
  }
  m = (int64_t)tot;
  if (m == 0) return;
  DFGPU_HIP(hipMemcpyAsync(d_off->ptr, off.data(), (size_t)nt * 8, hipMemcpyHostToDevice, r.stream));
  if (columns) {
    const RjCols cols = (*columns)(m);
    int64_t row_bytes = 0;
    for (int c = 0; c < cols.n; c++) row_bytes += cols.width[c];
    ProfileScope pse("radix_join_emit_columns", rec_bytes + m * row_bytes * 2);
    DFGPU_CHECK(t.exact, "internal: the fused column emit needs exact record keys");
    k_rj_join<true, 2><<<grid, BLOCK, 0, r.stream>>>(t.build.key->as<uint64_t>(), t.build.rid->as<uint32_t>(), t.build.starts->as<uint64_t>(), ps.key->as<uint64_t>(),
                                                     ps.rid->as<uint32_t>(), d_tasks->as<RadixTask>(), nt, v, nullptr, d_off->as<uint64_t>(), nullptr, nullptr, cols);
    DFGPU_HIP(hipGetLastError());
    DFGPU_HIP(hipStreamSynchronize(r.stream));  // `off` and `tasks` are locals the copies read
    return;
  }
  out_b = make_buf((size_t)m * 8);
  out_p = make_buf((size_t)m * 8);
  {
    ProfileScope pse("radix_join_emit", rec_bytes + m * 16);
    if (t.exact)
      k_rj_join<true, 1><<<grid, BLOCK, 0, r.stream>>>(t.build.key->as<uint64_t>(), t.build.rid->as<uint32_t>(), t.build.starts->as<uint64_t>(), ps.key->as<uint64_t>(),
                                                          ps.rid->as<uint32_t>(), d_tasks->as<RadixTask>(), nt, v, nullptr, d_off->as<uint64_t>(), out_b->as<int64_t>(), out_p->as<int64_t>());
    else
      k_rj_join<false, 1><<<grid, BLOCK, 0, r.stream>>>(t.build.key->as<uint64_t>(), t.build.rid->as<uint32_t>(), t.build.starts->as<uint64_t>(), ps.key->as<uint64_t>(),
                                                           ps.rid->as<uint32_t>(), d_tasks->as<RadixTask>(), nt, v, nullptr, d_off->as<uint64_t>(), out_b->as<int64_t>(), out_p->as<int64_t>());
    DFGPU_HIP(hipGetLastError());
  }
  DFGPU_HIP(hipStreamSynchronize(r.stream));  // `off` and `tasks` are locals the copies read
}

void radix_join_pairs(const RadixTable& t, const Table& build, const std::vector<int>& build_keys, const Table& probe, const std::vector<int>& probe_keys,
                      bool null_equals_null, bool force_collisions, BufPtr& out_b, BufPtr& out_p, int64_t& m) {
  radix_join_run(t, build, build_keys, probe, probe_keys, null_equals_null, force_collisions, out_b, out_p, m, nullptr);
}

// INNER join, output columns written by the emit walk (no pair list, no gathers).  false: this table / these columns do not qualify
// (record keys that are hashes, nullable or variable-width payload, too many columns) — the caller takes the pairs.
bool radix_join_inner_columns(const RadixTable& t, const Table& build, const std::vector<int>& build_keys, const Table& probe, const std::vector<int>& probe_keys,
                              const std::vector<int>& bout, const std::vector<int>& pout, Table& out) {
  if (!t.exact || force_fused_off() || (int)(bout.size() + pout.size()) > RJ_MAX_COLS || bout.size() + pout.size() == 0) return false;
  auto fixed = [](const Column& c) { return !c.has_nulls() && c.field.type != DFGPU_UTF8 && c.field.type != DFGPU_BOOL; };
  for (int c : bout)
    if (!fixed(build.cols[(size_t)c])) return false;
  for (int c : pout)
    if (!fixed(probe.cols[(size_t)c])) return false;
  if (probe.cols[(size_t)probe_keys[0]].has_nulls() || build.cols[(size_t)build_keys[0]].has_nulls()) return false;   // (NULL keys are dropped from the records: fine, but keep it simple)
  out = Table{};
  out.device = probe.device;
  std::function<RjCols(int64_t)> columns = [&](int64_t m) {
    RjCols rc{};
    out.nrows = m;
    for (int c : bout) {
      const Column& sc = build.cols[(size_t)c];
      out.cols.push_back(alloc_like(sc, m));
      rc.src[rc.n] = sc.ptr();
      rc.dst[rc.n] = out.cols.back().data->ptr;
      rc.width[rc.n] = type_width(sc.field.type);
      if (c == build_keys[0]) rc.key_cols |= 1u << rc.n;
      rc.n++;
    }
    rc.n_build = rc.n;
    for (int c : pout) {
      const Column& sc = probe.cols[(size_t)c];
      out.cols.push_back(alloc_like(sc, m));
      rc.src[rc.n] = sc.ptr();
      rc.dst[rc.n] = out.cols.back().data->ptr;
      rc.width[rc.n] = type_width(sc.field.type);
      if (c == probe_keys[0]) rc.key_cols |= 1u << rc.n;
      rc.n++;
    }
    return rc;
  };
  BufPtr ob, op;
  int64_t m = 0;
  radix_join_run(t, build, build_keys, probe, probe_keys, false, false, ob, op, m, &columns);
  if (m == 0) {   // no pair: the callback never ran
    out.nrows = 0;
    out.cols.clear();
    for (int c : bout) out.cols.push_back(alloc_like(build.cols[(size_t)c], 0));
    for (int c : pout) out.cols.push_back(alloc_like(probe.cols[(size_t)c], 0));
  }
  return true;
}

}  // namespace dfgpu
