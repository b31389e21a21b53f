// expr.hip — K9: PhysicalExpr evaluation on device.  Column / Literal / CastExpr /
// BinaryExpr(+,-,*, comparisons, AND/OR) / NOT / IS [NOT] NULL over fixed-width Arrow columns.
//
// Mirrors BinaryExpr::evaluate (physical-expr/src/expressions/binary.rs:536-656): children
// are evaluated to ColumnarValues (array or scalar datum), arithmetic dispatches to wrapping
// kernels (arrow-arith add_wrapping/sub_wrapping/mul_wrapping, :625-637), comparisons to
// `apply_cmp` (physical-expr-common/src/datum.rs:60-100) and produce bit-packed Boolean
// arrays via one wave64 ballot per 64 rows; AND/OR follow Kleene logic (:543-603).
// Decimal128 result types follow arrow-rs (clamped to precision 38, never an error):
// add/sub -> scale max(s1,s2), mul -> scale s1+s2 (expr-common/src/type_coercion/binary.rs:168-186).
#include <cmath>

#include "device.hpp"
#include "internal.hpp"

namespace dfgpu {

// ----------------------------------------------------------------------------- typing
static dfgpu_field mkfield(int type, int p = 0, int s = 0) {
  dfgpu_field f{};
  f.type = type;
  f.precision = p;
  f.scale = s;
  f.nullable = 1;
  return f;
}
bool same_field_type(const dfgpu_field& a, const dfgpu_field& b) {
  if (a.type != b.type) return false;
  if (a.type == DFGPU_DECIMAL128) return a.precision == b.precision && a.scale == b.scale;
  return true;
}
dfgpu_field arith_result_type(int op, const dfgpu_field& l, const dfgpu_field& r) {
  if (l.type == DFGPU_DECIMAL128 && r.type == DFGPU_DECIMAL128) {
    int p1 = l.precision, s1 = l.scale, p2 = r.precision, s2 = r.scale;
    if (op == DFGPU_EXPR_DIV) {  // arrow-arith decimal_op Op::Div: "follow postgres and MySQL adding a fixed scale increment of 4"
      const int s = std::min(38, s1 + 4);
      return mkfield(DFGPU_DECIMAL128, std::min(38, s - s1 + s2 + p1), s);
    }
    if (op == DFGPU_EXPR_MOD) {
      const int s = std::max(s1, s2);
      return mkfield(DFGPU_DECIMAL128, std::min(38, s + std::min(p1 - s1, p2 - s2)), s);
    }
    if (op == DFGPU_EXPR_MUL) return mkfield(DFGPU_DECIMAL128, std::min(38, p1 + p2 + 1), std::min(38, s1 + s2));
    int s = std::max(s1, s2);
    return mkfield(DFGPU_DECIMAL128, std::min(38, s + std::max(p1 - s1, p2 - s2) + 1), s);
  }
  DFGPU_CHECK(same_field_type(l, r), "arithmetic operand types differ: " + type_name(l) + " vs " + type_name(r) + " (the planner inserts casts)");
  DFGPU_CHECK(l.type == DFGPU_INT32 || l.type == DFGPU_INT64 || l.type == DFGPU_FLOAT64,
              "arithmetic on " + type_name(l) + " is not supported on the GPU path");
  return mkfield(l.type);
}

static dfgpu_field node_type(const dfgpu_expr& e, int idx, const Table& in) {
  DFGPU_CHECK(idx >= 0 && idx < e.n_nodes, "expression node index out of range");
  const dfgpu_expr_node& n = e.nodes[idx];
  switch (n.op) {
    case DFGPU_EXPR_COLUMN:
      DFGPU_CHECK(n.column >= 0 && n.column < (int)in.cols.size(), "Column index out of range");
      return in.cols[n.column].field;
    case DFGPU_EXPR_LITERAL:
    case DFGPU_EXPR_CAST:
      return n.field;
    case DFGPU_EXPR_ADD: case DFGPU_EXPR_SUB: case DFGPU_EXPR_MUL: case DFGPU_EXPR_DIV: case DFGPU_EXPR_MOD:
      return arith_result_type(n.op, node_type(e, n.left, in), node_type(e, n.right, in));
    case DFGPU_EXPR_DATE_PART:
      DFGPU_CHECK(node_type(e, n.left, in).type == DFGPU_DATE32, "date_part: the GPU path takes a Date32 argument");
      DFGPU_CHECK(n.column >= DFGPU_DATE_PART_YEAR && n.column <= DFGPU_DATE_PART_DAY, "date_part: the GPU path extracts YEAR, MONTH or DAY");
      return mkfield(DFGPU_INT32);
    case DFGPU_EXPR_SUBSTR: {
      // a Utf8 column stays Utf8; a dictionary-encoded column keeps its index type (eval_node checks that it IS one)
      dfgpu_field t = node_type(e, n.left, in);
      DFGPU_CHECK(t.type == DFGPU_UTF8 || t.type == DFGPU_INT32 || t.type == DFGPU_UINT32 || t.type == DFGPU_UINT8 || t.type == DFGPU_INT64 || t.type == DFGPU_UINT64,
                  "substr: the argument must be a string column (Utf8 or dictionary-encoded)");
      return t;
    }
    case DFGPU_EXPR_EQ: case DFGPU_EXPR_NE: case DFGPU_EXPR_LT: case DFGPU_EXPR_LE: case DFGPU_EXPR_GT: case DFGPU_EXPR_GE:
    case DFGPU_EXPR_AND: case DFGPU_EXPR_OR: case DFGPU_EXPR_NOT: case DFGPU_EXPR_IS_NULL: case DFGPU_EXPR_IS_NOT_NULL:
    case DFGPU_EXPR_LIKE: case DFGPU_EXPR_ILIKE:
      return mkfield(DFGPU_BOOL);
    case DFGPU_EXPR_CASE: {
      DFGPU_CHECK(node_type(e, n.column, in).type == DFGPU_BOOL, "CASE WHEN condition must be Boolean");
      dfgpu_field t = node_type(e, n.left, in);
      if (n.right >= 0) {
        dfgpu_field f = node_type(e, n.right, in);
        DFGPU_CHECK(same_field_type(t, f), "CASE branch types differ: " + type_name(t) + " vs " + type_name(f) + " (the planner inserts casts)");
      }
      t.nullable = 1;
      return t;
    }
  }
  throw Error("unsupported expression op " + std::to_string(n.op));
}
dfgpu_field expr_type(const dfgpu_expr& e, const Table& input) { return node_type(e, e.root, input); }

// ---------------------------------------------------------------------------- kernels
// operand: array pointer or broadcast scalar
template <typename T>
struct Operand {
  const T* p;
  T s;
  __device__ __forceinline__ T at(int64_t i) const { return p ? p[i] : s; }
};

template <typename T>
__device__ __forceinline__ bool cmp_op(int op, T x, T y) {
  switch (op) {
    case DFGPU_EXPR_EQ: return x == y;
    case DFGPU_EXPR_NE: return x != y;
    case DFGPU_EXPR_LT: return x < y;
    case DFGPU_EXPR_LE: return x <= y;
    case DFGPU_EXPR_GT: return x > y;
    default: return x >= y;
  }
}
// arrow-ord compares floats by total order (f64::total_cmp)
__device__ __forceinline__ int64_t f64_total_key(double d) {
  int64_t b = __double_as_longlong(d);
  return b ^ (int64_t)((uint64_t)(b >> 63) >> 1);
}

constexpr int CMP_UNROLL = 4;
// One wave64 produces one 64-bit mask word per 64 rows via ballot.  MA/MB = optional decimal
// rescale multipliers (10^k) applied with wrapping multiply before the comparison.
template <typename T, bool IS_F64>
__global__ __launch_bounds__(BLOCK) void k_cmp(int op, Operand<T> a, Operand<T> b, T ma, T mb, int64_t n, uint64_t* __restrict__ out) {
  const int64_t n_words = (n + 63) >> 6;
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * BLOCK) >> 6;
  for (int64_t w0 = wave * CMP_UNROLL; w0 < n_words; w0 += n_waves * CMP_UNROLL) {
    T x[CMP_UNROLL], y[CMP_UNROLL];
#pragma unroll
    for (int j = 0; j < CMP_UNROLL; j++) {
      int64_t i = ((w0 + j) << 6) + lane_id();
      bool in = i < n;
      x[j] = in ? a.at(i) : T{};
      y[j] = in ? b.at(i) : T{};
    }
    // the iteration's mask words leave in ONE store (lane j holds word j: 32 contiguous bytes) instead of one 8-byte store each
    uint64_t mine = 0;
#pragma unroll
    for (int j = 0; j < CMP_UNROLL; j++) {
      int64_t i = ((w0 + j) << 6) + lane_id();
      bool r;
      if constexpr (IS_F64) r = cmp_op<int64_t>(op, f64_total_key((double)x[j]), f64_total_key((double)y[j]));
      else r = cmp_op<T>(op, (T)(x[j] * ma), (T)(y[j] * mb));
      const uint64_t word = ballot64(i < n && r);
      mine = (int)lane_id() == j ? word : mine;
    }
    if ((int)lane_id() < CMP_UNROLL && w0 + (int64_t)lane_id() < n_words) out[w0 + lane_id()] = mine;
  }
}

template <typename T, typename UT>
__global__ __launch_bounds__(BLOCK) void k_arith(int op, Operand<T> a, Operand<T> b, UT ma, UT mb, int64_t n, T* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    UT x = (UT)a.at(i) * ma, y = (UT)b.at(i) * mb;
    out[i] = (T)(op == DFGPU_EXPR_ADD ? x + y : op == DFGPU_EXPR_SUB ? x - y : x * y);
  }
}
__global__ __launch_bounds__(BLOCK) void k_arith_f64(int op, Operand<double> a, Operand<double> b, int64_t n, double* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    double x = a.at(i), y = b.at(i);
    out[i] = op == DFGPU_EXPR_ADD ? x + y : op == DFGPU_EXPR_SUB ? x - y : x * y;
  }
}

// arrow-arith div / rem over the rows that are valid on both sides (try_binary visits no others).  flags[0] |= 1: a zero
// divisor, |= 2: an overflow (MIN / -1, or a decimal operand times its power of ten leaving 128 bits: mul_checked)
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_divmod(int is_mod, Operand<T> a, Operand<T> b, T ma, T mb, const uint64_t* __restrict__ valid, int64_t n,
                                                  T* __restrict__ out, unsigned* __restrict__ flags) {
  unsigned bad = 0;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    T v = 0;
    if (!valid || bit_at(valid, i)) {
      T x, y;
      const bool ox = __builtin_mul_overflow(a.at(i), ma, &x), oy = __builtin_mul_overflow(b.at(i), mb, &y);
      const bool ovf = ox || oy;
      if (ovf) bad |= 2u;
      else if (y == 0) bad |= 1u;
      else if (y == (T)-1 && x == (T)((T)1 << (sizeof(T) * 8 - 1))) bad |= 2u;
      else v = is_mod ? (T)(x % y) : (T)(x / y);
    }
    out[i] = v;
  }
  if (bad) atomicOr(flags, bad);
}
__global__ __launch_bounds__(BLOCK) void k_divmod_f64(int is_mod, Operand<double> a, Operand<double> b, int64_t n, double* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    const double x = a.at(i), y = b.at(i);
    out[i] = is_mod ? fmod(x, y) : x / y;
  }
}
__global__ __launch_bounds__(BLOCK) void k_date_part(const int32_t* __restrict__ days, int part, int64_t n, int32_t* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) out[i] = date32_part(days[i], part);
}

template <typename S, typename D>
__global__ __launch_bounds__(BLOCK) void k_cast(const S* __restrict__ in, int64_t n, D mul, D* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) out[i] = (D)in[i] * mul;
}

// word-wise bitmap kernels. mode: 0 and, 1 or, 2 not(a), 3 copy-not-valid (is_null), 4 fill(v)
// arrow-cast cast_decimal_to_decimal, scale reduction: x / 10^k rounded half away from zero; a result of more than `limit` - 1
// (= 10^precision - 1) in magnitude does not fit the target precision: flag (the reference's cast is not `safe`: an error)
__global__ __launch_bounds__(BLOCK) void k_cast_dec_down(const i128* __restrict__ in, const uint64_t* __restrict__ valid, int64_t n, i128 div, i128 limit,
                                                         i128* __restrict__ out, unsigned* __restrict__ flags) {
  unsigned bad = 0;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    const i128 x = in[i];
    i128 d = x / div;
    const i128 r = x % div, half = div / 2;
    if (x >= 0 ? r >= half : -r >= half) d += x >= 0 ? 1 : -1;
    if ((!valid || bit_at(valid, i)) && (d >= limit || d <= -limit)) bad = 1;
    out[i] = d;
  }
  if (bad) atomicOr(flags, 1u);
}
// arrow-cast cast_floating_point_to_decimal128: (v * 10^scale).round() as i128 (f64::round: half away from zero)
// With safe = false (the CastExpr default) a non-NULL row whose product is not finite or leaves the i128 range is the error "Cannot cast to
// Decimal128(p, s). Overflowing on v" (flags |= 1), one beyond the declared precision "v is too large to store in a Decimal128 of precision p"
// (flags |= 2); try_unary visits no NULL row.
__global__ __launch_bounds__(BLOCK) void k_cast_f64_dec(const double* __restrict__ in, const uint64_t* __restrict__ valid, int64_t n, double mul, i128 limit,
                                                        i128* __restrict__ out, unsigned* __restrict__ flags) {
  unsigned bad = 0;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    const double r = round(in[i] * mul);
    const bool live = !valid || bit_at(valid, i);
    const bool fits = r >= -0x1p127 && r < 0x1p127;   // false for NaN
    const i128 d = fits ? (i128)r : (i128)0;
    if (live && !fits) bad |= 1u;
    if (live && fits && (d >= limit || d <= -limit)) bad |= 2u;
    out[i] = d;
  }
  if (bad) atomicOr(flags, bad);
}
__global__ __launch_bounds__(BLOCK) void k_cast_div(const i128* __restrict__ in, int64_t n, double div, double* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) out[i] = (double)in[i] / div;
}
__global__ __launch_bounds__(BLOCK) void k_bitmap(int mode, const uint64_t* a, const uint64_t* b, uint64_t fill, int64_t n_words, uint64_t* out) {
  for (int64_t w = (int64_t)blockIdx.x * BLOCK + threadIdx.x; w < n_words; w += (int64_t)gridDim.x * BLOCK) {
    uint64_t r;
    switch (mode) {
      case 0: r = a[w] & b[w]; break;
      case 1: r = a[w] | b[w]; break;
      case 2: r = ~a[w]; break;
      default: r = fill; break;
    }
    out[w] = r;
  }
}
// Kleene AND / OR with validity (arrow and_kleene / or_kleene)
__global__ __launch_bounds__(BLOCK) void k_kleene(int is_or, const uint64_t* av, const uint64_t* avalid, const uint64_t* bv,
                                                  const uint64_t* bvalid, int64_t n_words, uint64_t* out_v, uint64_t* out_valid) {
  for (int64_t w = (int64_t)blockIdx.x * BLOCK + threadIdx.x; w < n_words; w += (int64_t)gridDim.x * BLOCK) {
    uint64_t va = avalid ? avalid[w] : ~0ull, vb = bvalid ? bvalid[w] : ~0ull;
    uint64_t at = av[w] & va, af = ~av[w] & va, bt = bv[w] & vb, bf = ~bv[w] & vb;
    uint64_t val, valid;
    if (is_or) { val = at | bt; valid = at | bt | (af & bf); }
    else { val = at & bt; valid = (at & bt) | af | bf; }
    out_v[w] = val;
    out_valid[w] = valid;
  }
}

// ------------------------------------------------------------------------------ host
// CASE WHEN: out[i] = (cond bit i set and valid) ? a : b, values by element width
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_select(const uint64_t* __restrict__ cond, const uint64_t* __restrict__ cond_valid, Operand<T> a, Operand<T> b, int64_t n,
                                                  T* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    uint64_t w = cond[i >> 6];
    if (cond_valid) w &= cond_valid[i >> 6];
    out[i] = ((w >> (i & 63)) & 1ull) ? a.at(i) : b.at(i);
  }
}
// the same per 64-row word for bit-packed values (Boolean data, validity bitmaps); a / b null = the fill word
__global__ __launch_bounds__(BLOCK) void k_select_bits(const uint64_t* __restrict__ cond, const uint64_t* __restrict__ cond_valid, const uint64_t* a, uint64_t afill,
                                                       const uint64_t* b, uint64_t bfill, int64_t n_words, uint64_t* __restrict__ out) {
  for (int64_t w = (int64_t)blockIdx.x * BLOCK + threadIdx.x; w < n_words; w += (int64_t)gridDim.x * BLOCK) {
    uint64_t m = cond[w];
    if (cond_valid) m &= cond_valid[w];
    out[w] = (m & (a ? a[w] : afill)) | (~m & (b ? b[w] : bfill));
  }
}

static i128 pow10_i128(int k) {
  i128 m = 1;
  for (int i = 0; i < k; i++) m *= 10;
  return m;
}
static i128 scalar_i128(const Datum& d) { return (i128)(((u128)d.lit_hi << 64) | d.lit_lo); }
static void set_scalar_i128(Datum& d, i128 v) {
  d.lit_lo = (uint64_t)(u128)v;
  d.lit_hi = (uint64_t)((u128)v >> 64);
}
static Datum make_scalar(const dfgpu_field& f, i128 v, bool is_null) {
  Datum d;
  d.scalar = true;
  d.scalar_null = is_null;
  d.col.field = f;
  set_scalar_i128(d, v);
  return d;
}

static BufPtr combine_validity(const Datum& a, const Datum& b, int64_t n, bool& all_null) {
  all_null = (a.scalar && a.scalar_null) || (b.scalar && b.scalar_null);
  if (all_null) return make_zero_buf(bitmap_bytes(n));
  const BufPtr& va = a.scalar ? nullptr : a.col.validity;
  const BufPtr& vb = b.scalar ? nullptr : b.col.validity;
  if (va && vb) {
    BufPtr o = make_buf(bitmap_bytes(n));
    int64_t nw = (n + 63) / 64;
    k_bitmap<<<grid_for(nw, BLOCK), BLOCK, 0, rt().stream>>>(0, va->as<uint64_t>(), vb->as<uint64_t>(), 0, nw, o->as<uint64_t>());
    return o;
  }
  return va ? va : vb;
}

template <typename T>
static Operand<T> operand(const Datum& d) {
  Operand<T> o{};
  if (d.scalar) {
    o.p = nullptr;
    if constexpr (std::is_same<T, double>::value) {
      double v;
      std::memcpy(&v, &d.lit_lo, 8);
      o.s = v;
    } else {
      o.s = (T)scalar_i128(d);
    }
  } else {
    o.p = reinterpret_cast<const T*>(d.col.ptr());
  }
  return o;
}

static Datum eval_node(const dfgpu_expr& e, int idx, const Table& in);

static Datum eval_cast(const dfgpu_expr_node& n, const Datum& src) {
  const dfgpu_field& from = src.col.field;
  const dfgpu_field& to = n.field;
  if (src.scalar) {
    // constant folding on the host
    if (src.scalar_null) return make_scalar(to, 0, true);
    if (to.type == DFGPU_DECIMAL128 && from.type == DFGPU_FLOAT64) {
      double v;
      std::memcpy(&v, &src.lit_lo, 8);
      const double r = std::round(v * std::pow(10.0, to.scale));
      DFGPU_CHECK(r >= -0x1p127 && r < 0x1p127, "Arrow error: Cast error: Cannot cast to " + type_name(to) + ". Overflowing on " + std::to_string(v));
      const i128 d = (i128)r, limit = pow10_i128(to.precision);
      DFGPU_CHECK(d < limit && d > -limit, "Arrow error: Invalid argument error: a value is too large to store in a " + type_name(to));
      return make_scalar(to, d, false);
    }
    if (to.type == DFGPU_DECIMAL128) {
      int fs = from.type == DFGPU_DECIMAL128 ? from.scale : 0;
      if (to.scale < fs) {
        const i128 x = scalar_i128(src), div = pow10_i128(fs - to.scale), half = div / 2, r = x % div;
        i128 d = x / div;
        if (x >= 0 ? r >= half : -r >= half) d += x >= 0 ? 1 : -1;
        const i128 limit = pow10_i128(to.precision);   // the same precision check the column path makes (k_cast_dec_down)
        DFGPU_CHECK(d < limit && d > -limit, "Arrow error: Invalid argument error: a value is too large to store in a " + type_name(to));
        return make_scalar(to, d, false);
      }
      return make_scalar(to, (i128)((u128)scalar_i128(src) * (u128)pow10_i128(to.scale - fs)), false);
    }
    if (to.type == DFGPU_FLOAT64) {
      double v = from.type == DFGPU_FLOAT64 ? 0 : (double)scalar_i128(src);
      if (from.type == DFGPU_DECIMAL128) v /= std::pow(10.0, from.scale);
      Datum d = make_scalar(to, 0, false);
      if (from.type == DFGPU_FLOAT64) return src;
      std::memcpy(&d.lit_lo, &v, 8);
      return d;
    }
    Datum d = src;
    d.col.field = to;
    return d;
  }
  const int64_t len = src.col.length;
  Datum out;
  out.col = alloc_column(to, src.col.name, len);
  out.col.validity = src.col.validity;
  out.col.null_count = src.col.null_count;
  int g = grid_for(len, BLOCK);
  hipStream_t st = rt().stream;
  auto unsupported = [&]() { throw Error("cast " + type_name(from) + " -> " + type_name(to) + " is not supported on the GPU path"); };
  int ft = from.type == DFGPU_DATE32 ? DFGPU_INT32 : from.type;
  if (same_field_type(from, to)) return src;
  ProfileScope ps("cast", len * (type_width(from.type) + type_width(to.type)));
  if (to.type == DFGPU_DECIMAL128 && ft == DFGPU_FLOAT64) {
    BufPtr flags = make_zero_buf(4);
    k_cast_f64_dec<<<g, BLOCK, 0, st>>>((const double*)src.col.ptr(), src.col.valid_words(), len, std::pow(10.0, to.scale), pow10_i128(to.precision),
                                        out.col.data->as<i128>(), flags->as<unsigned>());
    DFGPU_HIP(hipGetLastError());
    unsigned hf = 0;
    d2h(&hf, flags->ptr, 4);
    DFGPU_CHECK(!(hf & 1u), "Arrow error: Cast error: Cannot cast to " + type_name(to) + ". Overflowing on a value of the Float64 input");
    DFGPU_CHECK(!(hf & 2u), "Arrow error: Invalid argument error: a value is too large to store in a " + type_name(to));
  } else if (to.type == DFGPU_DECIMAL128 && from.type == DFGPU_DECIMAL128 && to.scale < from.scale) {
    BufPtr flags = make_zero_buf(4);
    k_cast_dec_down<<<g, BLOCK, 0, st>>>((const i128*)src.col.ptr(), src.col.valid_words(), len, pow10_i128(from.scale - to.scale), pow10_i128(to.precision),
                                         out.col.data->as<i128>(), flags->as<unsigned>());
    DFGPU_HIP(hipGetLastError());
    unsigned hf = 0;
    d2h(&hf, flags->ptr, 4);
    DFGPU_CHECK(hf == 0, "Arrow error: Invalid argument error: a value is too large to store in a " + type_name(to));
  } else if (to.type == DFGPU_DECIMAL128) {
    int fs = from.type == DFGPU_DECIMAL128 ? from.scale : 0;
    DFGPU_CHECK(to.scale >= fs, "decimal scale-down cast not supported on the GPU path");
    i128 mul = pow10_i128(to.scale - fs);
    i128* o = out.col.data->as<i128>();
    switch (ft) {
      case DFGPU_INT32: k_cast<int32_t, i128><<<g, BLOCK, 0, st>>>((const int32_t*)src.col.ptr(), len, mul, o); break;
      case DFGPU_INT64: k_cast<int64_t, i128><<<g, BLOCK, 0, st>>>((const int64_t*)src.col.ptr(), len, mul, o); break;
      case DFGPU_UINT8: k_cast<uint8_t, i128><<<g, BLOCK, 0, st>>>((const uint8_t*)src.col.ptr(), len, mul, o); break;
      case DFGPU_DECIMAL128: k_cast<i128, i128><<<g, BLOCK, 0, st>>>((const i128*)src.col.ptr(), len, mul, o); break;
      default: unsupported();
    }
  } else if (to.type == DFGPU_INT64) {
    int64_t* o = out.col.data->as<int64_t>();
    switch (ft) {
      case DFGPU_INT32: k_cast<int32_t, int64_t><<<g, BLOCK, 0, st>>>((const int32_t*)src.col.ptr(), len, 1, o); break;
      case DFGPU_UINT8: k_cast<uint8_t, int64_t><<<g, BLOCK, 0, st>>>((const uint8_t*)src.col.ptr(), len, 1, o); break;
      case DFGPU_UINT32: k_cast<uint32_t, int64_t><<<g, BLOCK, 0, st>>>((const uint32_t*)src.col.ptr(), len, 1, o); break;
      default: unsupported();
    }
  } else if (to.type == DFGPU_FLOAT64) {
    double* o = out.col.data->as<double>();
    switch (ft) {
      case DFGPU_INT32: k_cast<int32_t, double><<<g, BLOCK, 0, st>>>((const int32_t*)src.col.ptr(), len, 1.0, o); break;
      case DFGPU_INT64: k_cast<int64_t, double><<<g, BLOCK, 0, st>>>((const int64_t*)src.col.ptr(), len, 1.0, o); break;
      // arrow-cast cast_decimal_to_float: x as f64 / 10_f64.powi(scale)
      case DFGPU_DECIMAL128: k_cast_div<<<g, BLOCK, 0, st>>>((const i128*)src.col.ptr(), len, std::pow(10.0, from.scale), o); break;
      default: unsupported();
    }
  } else if ((to.type == DFGPU_INT32 || to.type == DFGPU_DATE32) && ft == DFGPU_INT32) {
    Datum d = src;
    d.col.field = to;
    return d;
  } else {
    unsupported();
  }
  DFGPU_HIP(hipGetLastError());
  return out;
}

static Datum to_bool_array(const Datum& d, int64_t n) {
  if (!d.scalar) return d;
  Datum o;
  o.col = alloc_column(mkfield(DFGPU_BOOL), "", n);
  int64_t nw = (n + 63) / 64;
  uint64_t fill = (!d.scalar_null && d.lit_lo) ? ~0ull : 0ull;
  if (nw) k_bitmap<<<grid_for(nw, BLOCK), BLOCK, 0, rt().stream>>>(4, nullptr, nullptr, fill, nw, o.col.data->as<uint64_t>());
  if (d.scalar_null) {
    o.col.validity = make_zero_buf(bitmap_bytes(n));
    o.col.null_count = n;
  }
  return o;
}

static void throw_divmod(unsigned flags) {
  if (flags & 1u) throw Error("Arrow error: Divide by zero error");
  if (flags & 2u) throw Error("Arrow error: Arithmetic overflow: Overflow happened on a division");
}
// BinaryExpr Divide / Modulo (binary.rs:636-637 -> arrow-arith div / rem)
static Datum eval_divmod(int op, const Datum& a, const Datum& b, int64_t nrows) {
  Runtime& r = rt();
  const dfgpu_field& lt = a.col.field;
  const dfgpu_field& rtp = b.col.field;
  const bool is_mod = op == DFGPU_EXPR_MOD;
  const dfgpu_field out_field = arith_result_type(op, lt, rtp);
  i128 ma = 1, mb = 1;
  if (out_field.type == DFGPU_DECIMAL128) {
    if (is_mod) {
      ma = pow10_i128(out_field.scale - lt.scale);
      mb = pow10_i128(out_field.scale - rtp.scale);
    } else {
      const int mul_pow = out_field.scale - lt.scale + rtp.scale;  // >= 0 unless the result scale was capped at 38
      if (mul_pow >= 0) ma = pow10_i128(mul_pow);
      else mb = pow10_i128(-mul_pow);
    }
  }
  if (a.scalar && b.scalar) {
    const bool isnull = a.scalar_null || b.scalar_null;
    if (isnull) return make_scalar(out_field, 0, true);
    if (lt.type == DFGPU_FLOAT64) {
      double x, y;
      std::memcpy(&x, &a.lit_lo, 8);
      std::memcpy(&y, &b.lit_lo, 8);
      const double v = is_mod ? std::fmod(x, y) : x / y;
      Datum d = make_scalar(out_field, 0, false);
      std::memcpy(&d.lit_lo, &v, 8);
      return d;
    }
    i128 x, y;
    const bool ox = __builtin_mul_overflow(scalar_i128(a), ma, &x), oy = __builtin_mul_overflow(scalar_i128(b), mb, &y);
    unsigned flags = (ox || oy) ? 2u : 0u;
    if (!flags && y == 0) flags = 1u;
    const i128 lowest = out_field.type == DFGPU_INT32 ? (i128)INT32_MIN : out_field.type == DFGPU_INT64 ? (i128)INT64_MIN : (i128)((u128)1 << 127);
    if (!flags && y == -1 && x == lowest) flags = 2u;
    throw_divmod(flags);
    return make_scalar(out_field, is_mod ? x % y : x / y, false);
  }
  Datum o;
  o.col = alloc_column(out_field, "", nrows);
  bool all_null = false;
  o.col.validity = combine_validity(a, b, nrows, all_null);
  o.col.null_count = o.col.validity ? -1 : 0;
  if (nrows == 0) return o;
  const int g = grid_for(nrows, BLOCK);
  const uint64_t* valid = o.col.valid_words();
  BufPtr flags = make_zero_buf(4);
  {
    ProfileScope ps("arith", nrows * type_width(out_field.type) * ((a.scalar ? 0 : 1) + (b.scalar ? 0 : 1) + 1));
    switch (out_field.type) {
      case DFGPU_INT32: k_divmod<int32_t><<<g, BLOCK, 0, r.stream>>>(is_mod, operand<int32_t>(a), operand<int32_t>(b), 1, 1, valid, nrows, o.col.data->as<int32_t>(), flags->as<unsigned>()); break;
      case DFGPU_INT64: k_divmod<int64_t><<<g, BLOCK, 0, r.stream>>>(is_mod, operand<int64_t>(a), operand<int64_t>(b), 1, 1, valid, nrows, o.col.data->as<int64_t>(), flags->as<unsigned>()); break;
      case DFGPU_DECIMAL128: k_divmod<i128><<<g, BLOCK, 0, r.stream>>>(is_mod, operand<i128>(a), operand<i128>(b), ma, mb, valid, nrows, o.col.data->as<i128>(), flags->as<unsigned>()); break;
      case DFGPU_FLOAT64: k_divmod_f64<<<g, BLOCK, 0, r.stream>>>(is_mod, operand<double>(a), operand<double>(b), nrows, o.col.data->as<double>()); break;
      default: throw Error("arithmetic result type not supported");
    }
    DFGPU_HIP(hipGetLastError());
  }
  if (out_field.type != DFGPU_FLOAT64) {
    unsigned hf = 0;
    d2h(&hf, flags->ptr, 4);
    throw_divmod(hf);
  }
  return o;
}

static Datum eval_binary(const dfgpu_expr_node& n, const Datum& a, const Datum& b, int64_t nrows) {
  Runtime& r = rt();
  if (n.op == DFGPU_EXPR_DIV || n.op == DFGPU_EXPR_MOD) return eval_divmod(n.op, a, b, nrows);
  const dfgpu_field& lt = a.col.field;
  const dfgpu_field& rtp = b.col.field;
  const int op = n.op;
  // dictionary-encoded strings are compared through their indices: that is only the strings' comparison when both
  // sides share one dictionary (and, for an ordering, when it is sorted); a literal was bound to an index by the caller
  if (!a.scalar && !b.scalar && (a.col.dict || b.col.dict) && op >= DFGPU_EXPR_EQ && op <= DFGPU_EXPR_GE) {
    DFGPU_CHECK(a.col.dict && b.col.dict, "comparison of a dictionary-encoded column with a plain column is not supported on the GPU path");
    DFGPU_CHECK(same_dictionary(a.col.dict, b.col.dict), "comparison of two dictionary-encoded columns with different dictionaries is not supported on the GPU path");
    DFGPU_CHECK(op == DFGPU_EXPR_EQ || op == DFGPU_EXPR_NE || a.col.dict->sorted, "ordering comparison over an unsorted dictionary is not supported on the GPU path");
  }
  if (op == DFGPU_EXPR_AND || op == DFGPU_EXPR_OR) {
    DFGPU_CHECK(lt.type == DFGPU_BOOL && rtp.type == DFGPU_BOOL, "AND/OR operands must be Boolean");
    Datum x = to_bool_array(a, nrows), y = to_bool_array(b, nrows);
    Datum o;
    o.col = alloc_column(mkfield(DFGPU_BOOL), "", nrows);
    int64_t nw = (nrows + 63) / 64;
    if (nw == 0) return o;
    ProfileScope ps("bool_and_or", nw * 24);
    if (!x.col.validity && !y.col.validity) {
      k_bitmap<<<grid_for(nw, BLOCK), BLOCK, 0, r.stream>>>(op == DFGPU_EXPR_OR ? 1 : 0, x.col.data->as<uint64_t>(), y.col.data->as<uint64_t>(), 0, nw, o.col.data->as<uint64_t>());
    } else {
      o.col.validity = make_buf(bitmap_bytes(nrows));
      o.col.null_count = -1;
      k_kleene<<<grid_for(nw, BLOCK), BLOCK, 0, r.stream>>>(op == DFGPU_EXPR_OR, x.col.data->as<uint64_t>(), x.col.valid_words(), y.col.data->as<uint64_t>(),
                                                            y.col.valid_words(), nw, o.col.data->as<uint64_t>(), o.col.validity->as<uint64_t>());
    }
    DFGPU_HIP(hipGetLastError());
    return o;
  }
  const bool is_cmp = op >= DFGPU_EXPR_EQ && op <= DFGPU_EXPR_GE;
  // decimal operand rescale multipliers
  i128 ma = 1, mb = 1;
  dfgpu_field out_field;
  if (is_cmp) {
    out_field = mkfield(DFGPU_BOOL);
    if (lt.type == DFGPU_DECIMAL128 && rtp.type == DFGPU_DECIMAL128) {
      int s = std::max(lt.scale, rtp.scale);
      ma = pow10_i128(s - lt.scale);
      mb = pow10_i128(s - rtp.scale);
    } else {
      int l = lt.type == DFGPU_DATE32 ? DFGPU_INT32 : lt.type, rr = rtp.type == DFGPU_DATE32 ? DFGPU_INT32 : rtp.type;
      DFGPU_CHECK(l == rr, "comparison operand types differ: " + type_name(lt) + " vs " + type_name(rtp));
    }
  } else {
    out_field = arith_result_type(op, lt, rtp);
    if (out_field.type == DFGPU_DECIMAL128 && op != DFGPU_EXPR_MUL) {
      ma = pow10_i128(out_field.scale - lt.scale);
      mb = pow10_i128(out_field.scale - rtp.scale);
    }
  }
  if (a.scalar && b.scalar) {
    // both literal: fold on the host (BinaryExpr with two scalar datums)
    bool isnull = a.scalar_null || b.scalar_null;
    if (lt.type == DFGPU_FLOAT64) {
      double x, y;
      std::memcpy(&x, &a.lit_lo, 8);
      std::memcpy(&y, &b.lit_lo, 8);
      if (is_cmp) {
        bool rr = op == DFGPU_EXPR_EQ ? x == y : op == DFGPU_EXPR_NE ? x != y : op == DFGPU_EXPR_LT ? x < y : op == DFGPU_EXPR_LE ? x <= y : op == DFGPU_EXPR_GT ? x > y : x >= y;
        return make_scalar(out_field, rr, isnull);
      }
      double v = op == DFGPU_EXPR_ADD ? x + y : op == DFGPU_EXPR_SUB ? x - y : x * y;
      Datum d = make_scalar(out_field, 0, isnull);
      std::memcpy(&d.lit_lo, &v, 8);
      return d;
    }
    u128 x = (u128)scalar_i128(a) * (u128)ma, y = (u128)scalar_i128(b) * (u128)mb;
    if (is_cmp) {
      i128 sx = (i128)x, sy = (i128)y;
      bool rr = op == DFGPU_EXPR_EQ ? sx == sy : op == DFGPU_EXPR_NE ? sx != sy : op == DFGPU_EXPR_LT ? sx < sy : op == DFGPU_EXPR_LE ? sx <= sy : op == DFGPU_EXPR_GT ? sx > sy : sx >= sy;
      return make_scalar(out_field, rr, isnull);
    }
    u128 v = op == DFGPU_EXPR_ADD ? x + y : op == DFGPU_EXPR_SUB ? x - y : x * y;
    if (out_field.type == DFGPU_INT32) v = (u128)(i128)(int32_t)(uint32_t)v;
    if (out_field.type == DFGPU_INT64) v = (u128)(i128)(int64_t)(uint64_t)v;
    return make_scalar(out_field, (i128)v, isnull);
  }
  Datum o;
  o.col = alloc_column(out_field, "", nrows);
  bool all_null = false;
  o.col.validity = combine_validity(a, b, nrows, all_null);
  o.col.null_count = o.col.validity ? -1 : 0;
  if (nrows == 0) return o;
  int ptype = lt.type == DFGPU_DATE32 ? DFGPU_INT32 : lt.type;
  hipStream_t st = r.stream;
  if (is_cmp) {
    int64_t nw = (nrows + 63) / 64;
    int g = grid_for(nw, (BLOCK / WAVE) * CMP_UNROLL);
    int64_t bytes = nrows / 8 + (a.scalar ? 0 : nrows * type_width(lt.type)) + (b.scalar ? 0 : nrows * type_width(rtp.type));
    ProfileScope ps("cmp", bytes);
    uint64_t* out = o.col.data->as<uint64_t>();
    switch (ptype) {
      case DFGPU_INT32: k_cmp<int32_t, false><<<g, BLOCK, 0, st>>>(op, operand<int32_t>(a), operand<int32_t>(b), 1, 1, nrows, out); break;
      case DFGPU_INT64: k_cmp<int64_t, false><<<g, BLOCK, 0, st>>>(op, operand<int64_t>(a), operand<int64_t>(b), 1, 1, nrows, out); break;
      case DFGPU_UINT8: k_cmp<uint8_t, false><<<g, BLOCK, 0, st>>>(op, operand<uint8_t>(a), operand<uint8_t>(b), 1, 1, nrows, out); break;
      case DFGPU_UINT32: k_cmp<uint32_t, false><<<g, BLOCK, 0, st>>>(op, operand<uint32_t>(a), operand<uint32_t>(b), 1, 1, nrows, out); break;
      case DFGPU_UINT64: k_cmp<uint64_t, false><<<g, BLOCK, 0, st>>>(op, operand<uint64_t>(a), operand<uint64_t>(b), 1, 1, nrows, out); break;
      case DFGPU_DECIMAL128: k_cmp<i128, false><<<g, BLOCK, 0, st>>>(op, operand<i128>(a), operand<i128>(b), ma, mb, nrows, out); break;
      case DFGPU_FLOAT64: k_cmp<double, true><<<g, BLOCK, 0, st>>>(op, operand<double>(a), operand<double>(b), 1.0, 1.0, nrows, out); break;
      default: throw Error("comparison on " + type_name(lt) + " is not supported on the GPU path");
    }
  } else {
    int g = grid_for(nrows, BLOCK);
    int w = type_width(out_field.type);
    ProfileScope ps("arith", nrows * w * ((a.scalar ? 0 : 1) + (b.scalar ? 0 : 1) + 1));
    switch (out_field.type) {
      case DFGPU_INT32: k_arith<int32_t, uint32_t><<<g, BLOCK, 0, st>>>(op, operand<int32_t>(a), operand<int32_t>(b), 1u, 1u, nrows, o.col.data->as<int32_t>()); break;
      case DFGPU_INT64: k_arith<int64_t, uint64_t><<<g, BLOCK, 0, st>>>(op, operand<int64_t>(a), operand<int64_t>(b), 1ull, 1ull, nrows, o.col.data->as<int64_t>()); break;
      case DFGPU_DECIMAL128: k_arith<i128, u128><<<g, BLOCK, 0, st>>>(op, operand<i128>(a), operand<i128>(b), (u128)ma, (u128)mb, nrows, o.col.data->as<i128>()); break;
      case DFGPU_FLOAT64: k_arith_f64<<<g, BLOCK, 0, st>>>(op, operand<double>(a), operand<double>(b), nrows, o.col.data->as<double>()); break;
      default: throw Error("arithmetic result type not supported");
    }
  }
  DFGPU_HIP(hipGetLastError());
  return o;
}

// CaseExpr, one WHEN (expressions/case.rs `case_when_no_expr`): THEN where the condition is TRUE, ELSE where it is
// FALSE or NULL
static Datum eval_case(const Datum& c, const Datum& a, const Datum& b, int64_t nrows) {
  if (c.scalar) return (!c.scalar_null && (c.lit_lo & 1)) ? a : b;
  const dfgpu_field f = a.col.field;
  const uint64_t* cw = c.col.data->as<uint64_t>();
  const uint64_t* cv = c.col.valid_words();
  hipStream_t st = rt().stream;
  const int64_t nw = (nrows + 63) / 64;
  Datum o;
  o.col = alloc_column(f, "", nrows);
  auto bits_of = [](const Datum& d, bool validity, const uint64_t*& p, uint64_t& fill) {
    p = nullptr;
    if (validity) {
      if (d.scalar) fill = d.scalar_null ? 0ull : ~0ull;
      else { p = d.col.valid_words(); fill = ~0ull; }
    } else {
      if (d.scalar) fill = (d.lit_lo & 1) ? ~0ull : 0ull;
      else p = d.col.data->as<uint64_t>();
    }
  };
  if (nrows) {
    if (f.type == DFGPU_BOOL) {
      const uint64_t *pa, *pb;
      uint64_t fa = 0, fb = 0;
      bits_of(a, false, pa, fa);
      bits_of(b, false, pb, fb);
      k_select_bits<<<grid_for(nw, BLOCK), BLOCK, 0, st>>>(cw, cv, pa, fa, pb, fb, nw, o.col.data->as<uint64_t>());
    } else {
      const int g = grid_for(nrows, BLOCK);
      switch (type_width(f.type)) {
        case 16: k_select<i128><<<g, BLOCK, 0, st>>>(cw, cv, operand<i128>(a), operand<i128>(b), nrows, o.col.data->as<i128>()); break;
        case 8: k_select<uint64_t><<<g, BLOCK, 0, st>>>(cw, cv, operand<uint64_t>(a), operand<uint64_t>(b), nrows, o.col.data->as<uint64_t>()); break;
        case 4: k_select<uint32_t><<<g, BLOCK, 0, st>>>(cw, cv, operand<uint32_t>(a), operand<uint32_t>(b), nrows, o.col.data->as<uint32_t>()); break;
        default: k_select<uint8_t><<<g, BLOCK, 0, st>>>(cw, cv, operand<uint8_t>(a), operand<uint8_t>(b), nrows, o.col.data->as<uint8_t>()); break;
      }
    }
  }
  const bool a_nulls = a.scalar ? a.scalar_null : a.col.has_nulls();
  const bool b_nulls = b.scalar ? b.scalar_null : b.col.has_nulls();
  if (a_nulls || b_nulls) {
    const uint64_t *pa, *pb;
    uint64_t fa = ~0ull, fb = ~0ull;
    bits_of(a, true, pa, fa);
    bits_of(b, true, pb, fb);
    o.col.validity = make_buf(bitmap_bytes(nrows));
    if (nw) k_select_bits<<<grid_for(nw, BLOCK), BLOCK, 0, st>>>(cw, cv, pa, fa, pb, fb, nw, o.col.validity->as<uint64_t>());
    o.col.null_count = -1;
  }
  // dictionary-encoded branches: the indices are what was selected; a NULL literal on the other side keeps the encoding
  const std::shared_ptr<const DictValues> da = a.scalar ? nullptr : a.col.dict, db = b.scalar ? nullptr : b.col.dict;
  DFGPU_CHECK(!(da && db && da != db), "CASE over two differently encoded dictionary columns is not supported on the GPU path");
  DFGPU_CHECK(!((da && b.scalar && !b.scalar_null) || (db && a.scalar && !a.scalar_null)),
              "CASE mixing a dictionary-encoded column with a literal index is not supported on the GPU path");
  DFGPU_CHECK(!((da && !b.scalar && !db) || (db && !a.scalar && !da)), "CASE mixing a dictionary-encoded column with a plain column");
  o.col.dict = da ? da : db;
  return o;
}

static Datum eval_node(const dfgpu_expr& e, int idx, const Table& in) {
  DFGPU_CHECK(idx >= 0 && idx < e.n_nodes, "expression node index out of range");
  const dfgpu_expr_node& n = e.nodes[idx];
  const int64_t nrows = in.nrows;
  switch (n.op) {
    case DFGPU_EXPR_CASE: {
      Datum c = eval_node(e, n.column, in);
      Datum a = eval_node(e, n.left, in);
      Datum b = n.right >= 0 ? eval_node(e, n.right, in) : make_scalar(a.col.field, 0, true);
      return eval_case(c, a, b, nrows);
    }
    case DFGPU_EXPR_COLUMN: {
      DFGPU_CHECK(n.column >= 0 && n.column < (int)in.cols.size(), "Column index out of range");
      Datum d;
      d.col = in.cols[n.column];
      return d;
    }
    case DFGPU_EXPR_LITERAL: {
      Datum d;
      d.scalar = true;
      d.scalar_null = n.is_null != 0;
      d.col.field = n.field;
      d.lit_lo = n.lit_lo;
      d.lit_hi = n.lit_hi;
      if (n.field.type == DFGPU_UTF8 && !d.scalar_null) {
        DFGPU_CHECK(e.string_pool != nullptr, "a string literal without dfgpu_expr.string_pool");
        d.str.assign(e.string_pool + n.lit_lo, (size_t)n.lit_hi);
      }
      return d;
    }
    case DFGPU_EXPR_CAST:
      return eval_cast(n, eval_node(e, n.left, in));
    case DFGPU_EXPR_DATE_PART: {
      Datum a = eval_node(e, n.left, in);
      if (a.scalar) return make_scalar(mkfield(DFGPU_INT32), a.scalar_null ? 0 : (i128)date32_part((int32_t)scalar_i128(a), n.column), a.scalar_null);
      Datum o;
      o.col = alloc_column(mkfield(DFGPU_INT32), "", nrows);
      o.col.validity = a.col.validity;
      o.col.null_count = a.col.null_count;
      if (nrows) {
        ProfileScope ps("date_part", nrows * 8);
        k_date_part<<<grid_for(nrows, BLOCK), BLOCK, 0, rt().stream>>>((const int32_t*)a.col.ptr(), n.column, nrows, o.col.data->as<int32_t>());
        DFGPU_HIP(hipGetLastError());
      }
      return o;
    }
    case DFGPU_EXPR_SUBSTR: {
      Datum a = eval_node(e, n.left, in);
      DFGPU_CHECK(!a.scalar, "substr of a literal is folded by the planner");
      DFGPU_CHECK(a.col.dict || a.col.field.type == DFGPU_UTF8, "substr: the argument must be a string column (Utf8 or dictionary-encoded)");
      Datum o;
      o.col = substr_column(a.col, (int64_t)n.column, n.is_null == 0, (int64_t)n.lit_lo);
      o.col.name.clear();
      return o;
    }
    case DFGPU_EXPR_NOT: {
      Datum a = to_bool_array(eval_node(e, n.left, in), nrows);
      DFGPU_CHECK(a.col.field.type == DFGPU_BOOL, "NOT operand must be Boolean");
      Datum o;
      o.col = alloc_column(mkfield(DFGPU_BOOL), "", nrows);
      o.col.validity = a.col.validity;
      o.col.null_count = a.col.null_count;
      int64_t nw = (nrows + 63) / 64;
      if (nw) k_bitmap<<<grid_for(nw, BLOCK), BLOCK, 0, rt().stream>>>(2, a.col.data->as<uint64_t>(), nullptr, 0, nw, o.col.data->as<uint64_t>());
      return o;
    }
    case DFGPU_EXPR_IS_NULL:
    case DFGPU_EXPR_IS_NOT_NULL: {
      Datum a = eval_node(e, n.left, in);
      bool want_null = n.op == DFGPU_EXPR_IS_NULL;
      if (a.scalar) return make_scalar(mkfield(DFGPU_BOOL), a.scalar_null == want_null, false);
      Datum o;
      o.col = alloc_column(mkfield(DFGPU_BOOL), "", nrows);
      int64_t nw = (nrows + 63) / 64;
      if (nw) {
        if (!a.col.validity) k_bitmap<<<grid_for(nw, BLOCK), BLOCK, 0, rt().stream>>>(4, nullptr, nullptr, want_null ? 0ull : ~0ull, nw, o.col.data->as<uint64_t>());
        else if (want_null) k_bitmap<<<grid_for(nw, BLOCK), BLOCK, 0, rt().stream>>>(2, a.col.valid_words(), nullptr, 0, nw, o.col.data->as<uint64_t>());
        else o.col.data = a.col.validity;
      }
      return o;
    }
    default: {
      Datum a = eval_node(e, n.left, in);
      Datum b = eval_node(e, n.right, in);
      // `dictionary-encoded column = 'literal'`: the literal becomes the index of that string in the column's dictionary (an
      // index no row holds when the string is absent) — what a caller does beforehand for a plain column reference
      // (dfgpu_table_dictionary_lookup) and cannot do for a computed column such as substr(c_phone, 1, 2)
      if (n.op == DFGPU_EXPR_EQ || n.op == DFGPU_EXPR_NE) {
        Datum* colside = (!a.scalar && a.col.dict && b.scalar && b.col.field.type == DFGPU_UTF8) ? &a : (!b.scalar && b.col.dict && a.scalar && a.col.field.type == DFGPU_UTF8) ? &b : nullptr;
        if (colside) {
          Datum& lit = colside == &a ? b : a;
          const DictValues& dv = *colside->col.dict;
          i128 code = (i128)dv.values.size();
          for (size_t k = 0; k < dv.values.size() && !lit.scalar_null; k++)
            if (dv.valid[k] && dv.values[k] == lit.str) {
              code = (i128)k;
              break;
            }
          DFGPU_CHECK(type_width(colside->col.field.type) > 1 || code <= 255, "the dictionary index type cannot hold the marker of an absent string");
          Datum bound = make_scalar(colside->col.field, code, lit.scalar_null);
          return colside == &a ? eval_binary(n, a, bound, nrows) : eval_binary(n, bound, b, nrows);
        }
      }
      if ((n.op == DFGPU_EXPR_LIKE || n.op == DFGPU_EXPR_ILIKE) && !a.scalar && a.col.dict && b.scalar && b.col.field.type == DFGPU_UTF8) {
        DFGPU_CHECK(!b.scalar_null, "LIKE NULL is folded by the planner");
        Datum o;
        o.col = dictionary_like_column(a.col, b.str, n.op == DFGPU_EXPR_ILIKE);
        return o;
      }
      if (n.op == DFGPU_EXPR_LIKE || n.op == DFGPU_EXPR_ILIKE || a.col.field.type == DFGPU_UTF8 || b.col.field.type == DFGPU_UTF8)
        return string_binary(n.op, a, b, nrows);
      return eval_binary(n, a, b, nrows);
    }
  }
}

Datum evaluate(const dfgpu_expr& e, const Table& input) {
  (void)expr_type(e, input);  // type-check first (errors surface before any launch)
  return eval_node(e, e.root, input);
}

template <typename T>
__global__ __launch_bounds__(BLOCK) void k_fill(T v, int64_t n, T* out) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) out[i] = v;
}

// ColumnarValue::into_array (scalar -> array of n copies)
Column datum_to_column(const Datum& d, int64_t n, const std::string& name) {
  if (!d.scalar) {
    Column c = d.col;
    c.name = name;
    return c;
  }
  if (d.col.field.type == DFGPU_BOOL) {
    Column c = to_bool_array(d, n).col;
    c.name = name;
    return c;
  }
  Column c = alloc_column(d.col.field, name, n);
  hipStream_t st = rt().stream;
  int g = grid_for(n, BLOCK);
  if (n) {
    switch (type_width(d.col.field.type)) {
      case 16: k_fill<i128><<<g, BLOCK, 0, st>>>(scalar_i128(d), n, c.data->as<i128>()); break;
      case 8: k_fill<uint64_t><<<g, BLOCK, 0, st>>>(d.lit_lo, n, c.data->as<uint64_t>()); break;
      case 4: k_fill<uint32_t><<<g, BLOCK, 0, st>>>((uint32_t)d.lit_lo, n, c.data->as<uint32_t>()); break;
      case 1: k_fill<uint8_t><<<g, BLOCK, 0, st>>>((uint8_t)d.lit_lo, n, c.data->as<uint8_t>()); break;
    }
  }
  if (d.scalar_null) {
    c.validity = make_zero_buf(bitmap_bytes(n));
    c.null_count = n;
  }
  return c;
}

}  // namespace dfgpu

using namespace dfgpu;

extern "C" {

int dfgpu_expr_type(const dfgpu_expr* e, dfgpu_table_t input, dfgpu_field* out) {
  return guarded([&] { *out = expr_type(*e, *unwrap(input)); });
}

int dfgpu_project(dfgpu_table_t input, const dfgpu_expr* exprs, const char* const* names, int n, dfgpu_table_t* out) {
  return guarded([&] {
    require_init();
    Table* t = unwrap(input);
    auto o = std::make_unique<Table>();
    o->nrows = t->nrows;
    for (int i = 0; i < n; i++) {
      Datum d = evaluate(exprs[i], *t);
      o->cols.push_back(datum_to_column(d, t->nrows, names && names[i] ? names[i] : ""));
    }
    *out = wrap(o.release());
  });
}

}  // extern "C"
