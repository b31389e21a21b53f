// tpch.hip — deterministic TPC-H-shaped synthetic generator, on device.
// tpchgen-cli / dbgen are unavailable (no network), so the workload tables are produced by a
// counter-based PRNG: every value is a pure function of (table seed, row, stream), any order
// range can be generated independently (one range per GPU) and datafusion_amd/tpch.py holds a
// bit-identical numpy mirror used by the CPU oracle.  Schemas: benchmarks/src/tpch/mod.rs:93-122
// restricted to the columns TPC-H Q1/Q3 read; 1-byte string columns are stored as UInt8 codes.
#include "device.hpp"
#include "internal.hpp"

#include <cmath>

namespace dfgpu {

constexpr uint64_t SEED_BASE = 0xDF55;
constexpr uint64_t T_CUSTOMER = 1, T_ORDERS = 2, T_LINEITEM = 3;
constexpr int32_t DATE_START = 8035;  // 1992-01-01
constexpr int32_t DATE_END = 10440;   // 1998-08-02
constexpr int32_t DATE_CUTOFF = 9298; // 1995-06-17

__host__ __device__ __forceinline__ uint64_t rnd(uint64_t table, uint64_t row, uint64_t stream) {
  return hash_u64(row * 16 + stream, SEED_BASE + table);
}
__host__ __device__ __forceinline__ int64_t order_key(int64_t i) { return (i >> 3) * 32 + (i & 7) + 1; }
__host__ __device__ __forceinline__ int32_t order_date(int64_t i) { return DATE_START + (int32_t)(rnd(T_ORDERS, (uint64_t)i, 1) % (uint64_t)(DATE_END - DATE_START + 1)); }
__host__ __device__ __forceinline__ uint32_t line_count(int64_t i) { return 1u + (uint32_t)(rnd(T_ORDERS, (uint64_t)i, 2) % 7u); }

__global__ __launch_bounds__(BLOCK) void k_gen_orders(int64_t begin, int64_t n, int64_t n_customers, int64_t* okey, int64_t* ocust, int32_t* odate, int32_t* oprio) {
  for (int64_t k = (int64_t)blockIdx.x * BLOCK + threadIdx.x; k < n; k += (int64_t)gridDim.x * BLOCK) {
    int64_t i = begin + k;
    okey[k] = order_key(i);
    int64_t c = 1 + (int64_t)(rnd(T_ORDERS, (uint64_t)i, 0) % (uint64_t)n_customers);
    if (c % 3 == 0) c = c > 1 ? c - 1 : c + 1;  // customers with custkey % 3 == 0 place no orders
    ocust[k] = c;
    odate[k] = order_date(i);
    oprio[k] = 0;
  }
}
__global__ __launch_bounds__(BLOCK) void k_gen_linecounts(int64_t begin, int64_t n, uint32_t* counts) {
  for (int64_t k = (int64_t)blockIdx.x * BLOCK + threadIdx.x; k < n; k += (int64_t)gridDim.x * BLOCK) counts[k] = line_count(begin + k);
}

struct LineCols {
  int64_t* orderkey;
  void* quantity;
  void* extprice;
  void* discount;
  void* tax;
  uint8_t* returnflag;
  uint8_t* linestatus;
  int32_t* shipdate;
};
template <bool FLOAT>
__device__ __forceinline__ void put_money(void* col, int64_t r, int64_t cents) {
  if (FLOAT) reinterpret_cast<double*>(col)[r] = (double)cents / 100.0;
  else reinterpret_cast<i128*>(col)[r] = (i128)cents;
}
template <bool FLOAT>
__global__ __launch_bounds__(BLOCK) void k_gen_lineitem(int64_t begin, int64_t n_orders, const uint64_t* __restrict__ offsets, int64_t n_parts, LineCols c) {
  for (int64_t k = (int64_t)blockIdx.x * BLOCK + threadIdx.x; k < n_orders; k += (int64_t)gridDim.x * BLOCK) {
    int64_t i = begin + k;
    int64_t r0 = (int64_t)offsets[k];
    uint32_t cnt = line_count(i);
    int64_t ok = order_key(i);
    int32_t od = order_date(i);
    for (uint32_t j = 0; j < cnt; j++) {
      int64_t r = r0 + j;
      uint64_t line = (uint64_t)i * 8 + j;  // unique line id
      int64_t qty = 1 + (int64_t)(rnd(T_LINEITEM, line, 0) % 50);
      int64_t part = 1 + (int64_t)(rnd(T_LINEITEM, line, 1) % (uint64_t)n_parts);
      int64_t price = 90000 + ((part / 10) % 20001) + 100 * (part % 1000);  // retail price in cents
      int64_t disc = (int64_t)(rnd(T_LINEITEM, line, 2) % 11);
      int64_t tax = (int64_t)(rnd(T_LINEITEM, line, 3) % 9);
      int32_t ship = od + 1 + (int32_t)(rnd(T_LINEITEM, line, 4) % 121);
      int32_t receipt = ship + 1 + (int32_t)(rnd(T_LINEITEM, line, 5) % 30);
      c.orderkey[r] = ok;
      put_money<FLOAT>(c.quantity, r, qty * 100);
      put_money<FLOAT>(c.extprice, r, qty * price);
      put_money<FLOAT>(c.discount, r, disc);
      put_money<FLOAT>(c.tax, r, tax);
      c.returnflag[r] = receipt <= DATE_CUTOFF ? ((rnd(T_LINEITEM, line, 6) & 1) ? 'R' : 'A') : 'N';
      c.linestatus[r] = ship > DATE_CUTOFF ? 'O' : 'F';
      c.shipdate[r] = ship;
    }
  }
}
__global__ __launch_bounds__(BLOCK) void k_gen_customer(int64_t begin, int64_t n, int64_t* ckey, uint8_t* seg) {
  for (int64_t k = (int64_t)blockIdx.x * BLOCK + threadIdx.x; k < n; k += (int64_t)gridDim.x * BLOCK) {
    int64_t i = begin + k;
    ckey[k] = i + 1;
    seg[k] = (uint8_t)(rnd(T_CUSTOMER, (uint64_t)i, 0) % 5);  // 0 AUTOMOBILE 1 BUILDING 2 FURNITURE 3 HOUSEHOLD 4 MACHINERY
  }
}

static dfgpu_field fld(int type, int p = 0, int s = 0) {
  dfgpu_field f{};
  f.type = type;
  f.precision = p;
  f.scale = s;
  return f;
}
static int64_t n_orders_for(double sf) { return (int64_t)std::llround(1500000.0 * sf); }
static int64_t n_customers_for(double sf) { return std::max<int64_t>(1, (int64_t)std::llround(150000.0 * sf)); }
static int64_t n_parts_for(double sf) { return std::max<int64_t>(1, (int64_t)std::llround(200000.0 * sf)); }

}  // namespace dfgpu

using namespace dfgpu;

extern "C" {

int dfgpu_tpch_orders(double sf, int64_t begin, int64_t end, dfgpu_table_t* out) {
  return guarded([&] {
    require_init();
    int64_t total = n_orders_for(sf);
    if (end < 0 || end > total) end = total;
    DFGPU_CHECK(begin >= 0 && begin <= end, "bad order range");
    int64_t n = end - begin;
    auto t = std::make_unique<Table>();
    t->nrows = n;
    t->cols.push_back(alloc_column(fld(DFGPU_INT64), "o_orderkey", n));
    t->cols.push_back(alloc_column(fld(DFGPU_INT64), "o_custkey", n));
    t->cols.push_back(alloc_column(fld(DFGPU_DATE32), "o_orderdate", n));
    t->cols.push_back(alloc_column(fld(DFGPU_INT32), "o_shippriority", n));
    if (n)
      k_gen_orders<<<grid_for(n, BLOCK), BLOCK, 0, rt().stream>>>(begin, n, n_customers_for(sf), t->cols[0].data->as<int64_t>(), t->cols[1].data->as<int64_t>(),
                                                                  t->cols[2].data->as<int32_t>(), t->cols[3].data->as<int32_t>());
    DFGPU_HIP(hipGetLastError());
    *out = wrap(t.release());
  });
}

int dfgpu_tpch_lineitem(double sf, int64_t begin, int64_t end, int32_t float_money, dfgpu_table_t* out) {
  return guarded([&] {
    require_init();
    Runtime& r = rt();
    int64_t total = n_orders_for(sf);
    if (end < 0 || end > total) end = total;
    DFGPU_CHECK(begin >= 0 && begin <= end, "bad order range");
    int64_t n = end - begin;
    BufPtr counts = make_buf((size_t)(n ? n : 1) * 4);
    BufPtr offsets = make_buf((size_t)(n + 1) * 8);
    if (n) k_gen_linecounts<<<grid_for(n, BLOCK), BLOCK, 0, r.stream>>>(begin, n, counts->as<uint32_t>());
    scan_u32(counts->as<uint32_t>(), n, offsets->as<uint64_t>());
    int64_t rows = (int64_t)read_u64(offsets->as<uint64_t>() + n);
    dfgpu_field money = float_money ? fld(DFGPU_FLOAT64) : fld(DFGPU_DECIMAL128, 15, 2);
    auto t = std::make_unique<Table>();
    t->nrows = rows;
    t->cols.push_back(alloc_column(fld(DFGPU_INT64), "l_orderkey", rows));
    t->cols.push_back(alloc_column(money, "l_quantity", rows));
    t->cols.push_back(alloc_column(money, "l_extendedprice", rows));
    t->cols.push_back(alloc_column(money, "l_discount", rows));
    t->cols.push_back(alloc_column(money, "l_tax", rows));
    t->cols.push_back(alloc_column(fld(DFGPU_UINT8), "l_returnflag", rows));
    t->cols.push_back(alloc_column(fld(DFGPU_UINT8), "l_linestatus", rows));
    t->cols.push_back(alloc_column(fld(DFGPU_DATE32), "l_shipdate", rows));
    LineCols c{t->cols[0].data->as<int64_t>(), t->cols[1].data->ptr, t->cols[2].data->ptr, t->cols[3].data->ptr, t->cols[4].data->ptr,
               t->cols[5].data->as<uint8_t>(), t->cols[6].data->as<uint8_t>(), t->cols[7].data->as<int32_t>()};
    if (n) {
      if (float_money) k_gen_lineitem<true><<<grid_for(n, BLOCK), BLOCK, 0, r.stream>>>(begin, n, offsets->as<uint64_t>(), n_parts_for(sf), c);
      else k_gen_lineitem<false><<<grid_for(n, BLOCK), BLOCK, 0, r.stream>>>(begin, n, offsets->as<uint64_t>(), n_parts_for(sf), c);
    }
    DFGPU_HIP(hipGetLastError());
    DFGPU_HIP(hipStreamSynchronize(r.stream));
    *out = wrap(t.release());
  });
}

int dfgpu_tpch_customer(double sf, int64_t begin, int64_t end, dfgpu_table_t* out) {
  return guarded([&] {
    require_init();
    int64_t total = n_customers_for(sf);
    if (end < 0 || end > total) end = total;
    DFGPU_CHECK(begin >= 0 && begin <= end, "bad customer range");
    int64_t n = end - begin;
    auto t = std::make_unique<Table>();
    t->nrows = n;
    t->cols.push_back(alloc_column(fld(DFGPU_INT64), "c_custkey", n));
    t->cols.push_back(alloc_column(fld(DFGPU_UINT8), "c_mktsegment", n));
    if (n) k_gen_customer<<<grid_for(n, BLOCK), BLOCK, 0, rt().stream>>>(begin, n, t->cols[0].data->as<int64_t>(), t->cols[1].data->as<uint8_t>());
    DFGPU_HIP(hipGetLastError());
    *out = wrap(t.release());
  });
}

}  // extern "C"
