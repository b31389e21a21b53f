// rowprog.hpp — fused expression evaluation: a PhysicalExpr forest compiled (on the host,
// rowprog.hip) into a short register program that a consuming kernel runs per row, so that
// FilterExec predicates, ProjectionExec expressions and aggregate arguments never touch HBM as
// intermediate columns.  The reference evaluates BinaryExpr trees column-at-a-time, one freshly
// allocated 8192-row array per node (physical-expr/src/expressions/binary.rs:536-656; SURVEY §8
// row a12 "no fusion"); at whole-partition granularity that is one full HBM round trip per node.
//
// Execution model (CDNA4): the program lives in the kernel-argument segment, so instruction fetch
// and operand indices are wave-uniform scalar loads and the dispatch is scalar branches; the
// register file is a per-lane array indexed by those uniform indices, which the compiler lowers
// to VGPR-relative addressing (s_set_gpr_idx_on / v_mov) — no scratch, no LDS.  Input columns are
// loaded by a statically unrolled prologue straight into registers 0..n_cols-1 so all of a row's
// HBM loads are in flight together before the first dependent instruction.
//
// Value representation: every register is 128 bits (lo, hi) + one null bit.
//   INT32 / DATE32 / INT64   sign-extended to 128 bits
//   UINT8 / UINT32 / UINT64  zero-extended
//   DECIMAL128               as is (two's complement)
//   FLOAT64                  IEEE bits in lo
//   BOOL                     0 / 1 in lo
#pragma once
#include "device.hpp"

namespace dfgpu {

enum RpOp : uint8_t {
  RP_LIT = 0,     // dst <- literal[aux]; a != 0: NULL literal
  RP_ADD = 1,     // 128-bit wrapping (arrow-arith add_wrapping / sub_wrapping / mul_wrapping)
  RP_SUB = 2,
  RP_MUL = 3,
  RP_SEXT32 = 4,  // re-normalise an Int32 result: sign-extend bits 31..0
  RP_SEXT64 = 5,  // same for Int64
  RP_FADD = 6,
  RP_FSUB = 7,
  RP_FMUL = 8,
  RP_I2F = 9,     // signed 64-bit integer -> f64
  RP_F64ORD = 10, // f64 bits -> order-preserving i64 key (f64::total_cmp order)
  RP_CMP = 11,    // signed 128-bit compare, aux = dfgpu_expr_op (EQ..GE) -> bool
  RP_FCMP = 12,   // f64 total-order compare
  RP_AND = 13,    // Kleene
  RP_OR = 14,
  RP_NOT = 15,
  RP_IS_NULL = 16,
  RP_IS_NOT_NULL = 17,
  RP_MOV = 18,
  // CASE WHEN c THEN x ELSE y END = MERGE(GATE(x, t), GATE(y, NOT t)) with t = c AND (c IS NOT NULL):
  RP_GATE = 19,   // dst <- b is TRUE ? a : (0, not NULL)        (b: a non-NULL Boolean)
  RP_MERGE = 20,  // dst <- a | b bitwise, NULL if either is     (at most one side is non-zero / NULL)
  RP_DATE_PART = 21  // dst <- date_part(aux: 0 YEAR / 1 MONTH / 2 DAY, Date32 a) as Int32
};
// how a column is widened into a register
enum RpLoad : uint8_t { RPL_I32 = 0, RPL_I64 = 1, RPL_U8 = 2, RPL_U32 = 3, RPL_U64 = 4, RPL_I128 = 5, RPL_F64 = 6, RPL_BOOL = 7 };

struct RpIns {
  uint8_t op, dst, a, b;
  uint32_t aux;
};
constexpr int RP_MAX_INS = 56;
constexpr int RP_MAX_COLS = 10;
constexpr int RP_MAX_LITS = 12;
constexpr int RP_NREG = 32;       // compiler limit; kernels are instantiated for 16 and 32 registers

struct RowProgram {
  RpIns ins[RP_MAX_INS];
  int n_ins;
  int n_cols;
  const void* col_data[RP_MAX_COLS];
  const uint64_t* col_valid[RP_MAX_COLS];  // optional validity bitmap per input column
  uint8_t col_kind[RP_MAX_COLS];           // RpLoad
  uint64_t lit_lo[RP_MAX_LITS], lit_hi[RP_MAX_LITS];
};

// Four dword planes: arrays of <= 32 dwords are what the AMDGPU backend keeps in VGPRs and indexes
// with s_set_gpr_idx (wider elements or longer arrays fall back to scratch memory).
// RpRegs is a view over four separate local arrays (RP_DECLARE_REGS) — one struct holding the arrays
// would be a single stack object that dynamic indexing keeps in scratch.
struct RpRegs {
  uint32_t *w0, *w1, *w2, *w3;
  uint32_t nulls;  // bit r = register r is NULL
  __device__ __forceinline__ uint64_t lo(int i) const { return ((uint64_t)w1[i] << 32) | w0[i]; }
  __device__ __forceinline__ uint64_t hi(int i) const { return ((uint64_t)w3[i] << 32) | w2[i]; }
  __device__ __forceinline__ void set(int i, uint64_t l, uint64_t h) {
    w0[i] = (uint32_t)l;
    w1[i] = (uint32_t)(l >> 32);
    w2[i] = (uint32_t)h;
    w3[i] = (uint32_t)(h >> 32);
  }
};

#define RP_DECLARE_REGS(r, NREG)                                            \
  uint32_t r##_w0[NREG], r##_w1[NREG], r##_w2[NREG], r##_w3[NREG];         \
  RpRegs r{r##_w0, r##_w1, r##_w2, r##_w3, 0u}

__device__ __forceinline__ bool rp_is_null(const RpRegs& r, int reg) { return (r.nulls >> reg) & 1u; }
__device__ __forceinline__ void rp_set_null(RpRegs& r, int reg, bool isnull) {
  r.nulls = (r.nulls & ~(1u << reg)) | ((isnull ? 1u : 0u) << reg);
}
// predicate semantics of FilterExec: NULL => row dropped (arrow-select filter)
__device__ __forceinline__ bool rp_true(const RpRegs& r, int reg) { return !rp_is_null(r, reg) && (r.w0[reg] & 1u); }

// Prologue: issue every input column's load for `row` (static register indices => the loads are
// independent and stay in flight together), then widen.
__device__ __forceinline__ void rp_load_row(const RowProgram& p, int64_t row, RpRegs& r) {
  uint32_t nulls = r.nulls;
#pragma unroll
  for (int c = 0; c < RP_MAX_COLS; c++) {
    if (c < p.n_cols) {
      const void* d = p.col_data[c];
      uint64_t lo = 0, hi = 0;
      switch (p.col_kind[c]) {
        case RPL_I32: lo = (uint64_t)(uint32_t)((const int32_t*)d)[row]; break;
        case RPL_U32: lo = ((const uint32_t*)d)[row]; break;
        case RPL_I64: case RPL_U64: case RPL_F64: lo = ((const uint64_t*)d)[row]; break;
        case RPL_U8: lo = ((const uint8_t*)d)[row]; break;
        case RPL_I128: { const uint64_t* q = (const uint64_t*)d + 2 * row; lo = q[0]; hi = q[1]; break; }
        default: lo = (((const uint64_t*)d)[row >> 6] >> (row & 63)) & 1ull; break;  // RPL_BOOL
      }
      r.set(c, lo, hi);
      const uint64_t* v = p.col_valid[c];
      bool isnull = v ? !bit_at(v, row) : false;
      nulls = (nulls & ~(1u << c)) | ((isnull ? 1u : 0u) << c);
    }
  }
#pragma unroll
  for (int c = 0; c < RP_MAX_COLS; c++) {
    if (c < p.n_cols) {
      int k = p.col_kind[c];
      if (k == RPL_I32) { uint32_t s = (uint32_t)((int32_t)r.w0[c] >> 31); r.w1[c] = s; r.w2[c] = s; r.w3[c] = s; }
      else if (k == RPL_I64) { uint32_t s = (uint32_t)((int32_t)r.w1[c] >> 31); r.w2[c] = s; r.w3[c] = s; }
    }
  }
  r.nulls = nulls;
}

// Software-pipelined variant of rp_load_row: rp_issue_row starts the loads of a (future) row into a
// statically indexed staging set, rp_commit_row widens a completed set into the register file.  A kernel
// issues row i+stride before interpreting row i, so HBM latency overlaps the interpreter even at the low
// occupancy its VGPR footprint allows.
struct RpRaw {
  uint64_t lo[RP_MAX_COLS];
  uint64_t hi[RP_MAX_COLS];
  uint32_t nulls;
};
__device__ __forceinline__ void rp_issue_row(const RowProgram& p, int64_t row, RpRaw& w) {
  // raw loads only: anything computed from a loaded value here would put the s_waitcnt at issue time.
  // (Validity bits are the exception — a nullable column costs the wait; TPC-H columns are non-null.)
  uint32_t nulls = 0;
#pragma unroll
  for (int c = 0; c < RP_MAX_COLS; c++) {
    if (c < p.n_cols) {
      const void* d = p.col_data[c];
      uint64_t lo = 0, hi = 0;
      switch (p.col_kind[c]) {
        case RPL_I32: case RPL_U32: lo = ((const uint32_t*)d)[row]; break;
        case RPL_I64: case RPL_U64: case RPL_F64: lo = ((const uint64_t*)d)[row]; break;
        case RPL_U8: lo = ((const uint8_t*)d)[row]; break;
        case RPL_I128: { const uint64_t* q = (const uint64_t*)d + 2 * row; lo = q[0]; hi = q[1]; break; }
        default: lo = ((const uint64_t*)d)[row >> 6]; hi = (uint64_t)(row & 63); break;  // RPL_BOOL: word + bit position
      }
      w.lo[c] = lo;
      w.hi[c] = hi;
      const uint64_t* v = p.col_valid[c];
      if (v) nulls |= (bit_at(v, row) ? 0u : 1u) << c;
    }
  }
  w.nulls = nulls;
}
__device__ __forceinline__ void rp_commit_row(const RowProgram& p, const RpRaw& w, RpRegs& r) {
#pragma unroll
  for (int c = 0; c < RP_MAX_COLS; c++) {
    if (c < p.n_cols) {
      uint64_t lo = w.lo[c], hi = w.hi[c];
      switch (p.col_kind[c]) {
        case RPL_I32: lo = (uint64_t)(int64_t)(int32_t)(uint32_t)lo; hi = (uint64_t)((int64_t)lo >> 63); break;
        case RPL_I64: hi = (uint64_t)((int64_t)lo >> 63); break;
        case RPL_BOOL: lo = (lo >> hi) & 1ull; hi = 0; break;
        default: break;
      }
      r.set(c, lo, hi);
    }
  }
  const uint32_t colmask = (1u << p.n_cols) - 1u;
  r.nulls = (r.nulls & ~colmask) | w.nulls;
}

__device__ __forceinline__ int64_t rp_f64_ordered(uint64_t bits) {
  int64_t b = (int64_t)bits;
  return b ^ (int64_t)((uint64_t)(b >> 63) >> 1);
}
__device__ __forceinline__ bool rp_cmp128(uint32_t op, i128 x, i128 y) {
  switch (op) {
    case DFGPU_EXPR_EQ: return x == y;
    case DFGPU_EXPR_NE: return x != y;
    case DFGPU_EXPR_LT: return x < y;
    case DFGPU_EXPR_LE: return x <= y;
    case DFGPU_EXPR_GT: return x > y;
    default: return x >= y;
  }
}

// run instructions [k0, k1) for the current row
__device__ __forceinline__ void rp_exec(const RowProgram& p, int k0, int k1, RpRegs& r) {
  for (int k = k0; k < k1; k++) {
    const RpIns in = p.ins[k];
    const int ra = in.a, rb = in.b, rd = in.dst;
    uint64_t alo = r.lo(ra), ahi = r.hi(ra);
    uint64_t blo = r.lo(rb), bhi = r.hi(rb);
    bool an = (r.nulls >> ra) & 1u, bn = (r.nulls >> rb) & 1u;
    uint64_t olo = 0, ohi = 0;
    bool on = an | bn;
    switch (in.op) {
      case RP_LIT: olo = p.lit_lo[in.aux]; ohi = p.lit_hi[in.aux]; on = in.a != 0; break;
      case RP_ADD: { u128 v = (((u128)ahi << 64) | alo) + (((u128)bhi << 64) | blo); olo = (uint64_t)v; ohi = (uint64_t)(v >> 64); break; }
      case RP_SUB: { u128 v = (((u128)ahi << 64) | alo) - (((u128)bhi << 64) | blo); olo = (uint64_t)v; ohi = (uint64_t)(v >> 64); break; }
      case RP_MUL: { u128 v = (((u128)ahi << 64) | alo) * (((u128)bhi << 64) | blo); olo = (uint64_t)v; ohi = (uint64_t)(v >> 64); break; }
      case RP_SEXT32: { int64_t s = (int64_t)(int32_t)(uint32_t)alo; olo = (uint64_t)s; ohi = (uint64_t)(s >> 63); on = an; break; }
      case RP_SEXT64: olo = alo; ohi = (uint64_t)((int64_t)alo >> 63); on = an; break;
      case RP_FADD: olo = (uint64_t)__double_as_longlong(__longlong_as_double((long long)alo) + __longlong_as_double((long long)blo)); break;
      case RP_FSUB: olo = (uint64_t)__double_as_longlong(__longlong_as_double((long long)alo) - __longlong_as_double((long long)blo)); break;
      case RP_FMUL: olo = (uint64_t)__double_as_longlong(__longlong_as_double((long long)alo) * __longlong_as_double((long long)blo)); break;
      case RP_I2F: olo = (uint64_t)__double_as_longlong((double)(int64_t)alo); on = an; break;
      case RP_F64ORD: { int64_t s = rp_f64_ordered(alo); olo = (uint64_t)s; ohi = (uint64_t)(s >> 63); on = an; break; }
      case RP_DATE_PART: { int64_t s = (int64_t)date32_part((int32_t)(uint32_t)alo, (int)in.aux); olo = (uint64_t)s; ohi = (uint64_t)(s >> 63); on = an; break; }
      case RP_CMP: olo = rp_cmp128(in.aux, (i128)(((u128)ahi << 64) | alo), (i128)(((u128)bhi << 64) | blo)) ? 1ull : 0ull; break;
      case RP_FCMP: olo = rp_cmp128(in.aux, (i128)rp_f64_ordered(alo), (i128)rp_f64_ordered(blo)) ? 1ull : 0ull; break;
      case RP_AND: {  // and_kleene: false AND x = false
        bool at = !an && (alo & 1), af = !an && !(alo & 1), bt = !bn && (blo & 1), bf = !bn && !(blo & 1);
        olo = (at && bt) ? 1ull : 0ull;
        on = !((at && bt) || af || bf);
        break;
      }
      case RP_OR: {  // or_kleene: true OR x = true
        bool at = !an && (alo & 1), af = !an && !(alo & 1), bt = !bn && (blo & 1), bf = !bn && !(blo & 1);
        olo = (at || bt) ? 1ull : 0ull;
        on = !(at || bt || (af && bf));
        break;
      }
      case RP_NOT: olo = (alo & 1) ^ 1ull; on = an; break;
      case RP_IS_NULL: olo = an ? 1ull : 0ull; on = false; break;
      case RP_IS_NOT_NULL: olo = an ? 0ull : 1ull; on = false; break;
      case RP_GATE: { const bool t = (blo & 1) != 0; olo = t ? alo : 0ull; ohi = t ? ahi : 0ull; on = t && an; break; }
      case RP_MERGE: olo = alo | blo; ohi = ahi | bhi; break;
      default: olo = alo; ohi = ahi; on = an; break;  // RP_MOV
    }
    r.set(rd, olo, ohi);
    r.nulls = (r.nulls & ~(1u << rd)) | ((on ? 1u : 0u) << rd);
  }
}


// ------------------------------------------------------------------ tile programs (LDS register file)
// Measured on MI355X (profiles/r1_q1_pmc.md): with the per-lane register file in VGPRs, every operand access
// is s_set_gpr_idx_on + v_mov + s_set_gpr_idx_off, and the Q1 node issued 925 SALU + 564 VALU instructions
// per 64 rows — instruction-issue bound at 12 % of HBM peak.  A TileProgram keeps the register file in LDS
// instead: register u of lane t lives at a fixed stride, so an operand is ONE ds_read with a wave-uniform
// offset, literals are scalar (kernel-argument) operands, and VGPRs are free to stage the next row's loads.
//   operand byte: bit 7 = literal (bits 0-6 literal index), else a lane register: ids [0, n_wide) are 16-byte
//   registers (Decimal128 / UInt64 values), ids [n_wide, n_wide + n_narrow) are 8-byte registers holding the
//   value sign-extended to 64 bits (ints, dates, bytes, bools, f64 bits).
constexpr uint8_t TP_LIT = 0x80;
struct TileProgram {
  RpIns ins[RP_MAX_INS];  // dst / a / b are operand bytes (dst never a literal); no RP_LIT instructions
  int n_ins;
  int n_pred_end;         // [0, n_pred_end): predicate; [n_pred_end, n_ins): outputs
  int n_cols;
  const void* col_data[RP_MAX_COLS];
  const uint64_t* col_valid[RP_MAX_COLS];
  uint8_t col_kind[RP_MAX_COLS];  // RpLoad
  uint8_t col_reg[RP_MAX_COLS];   // lane register receiving the column
  uint64_t lit_lo[RP_MAX_LITS], lit_hi[RP_MAX_LITS];
  uint32_t lit_nulls;             // bit k = literal k is NULL
  int n_wide, n_narrow;
};
// per-lane view of the LDS register file
struct TileRegs {
  char* wide;      // this lane's slot of wide register 0; register u at + u * BLOCK * 16
  char* narrow;    // this lane's slot of narrow register 0; register u at + (u - n_wide) * BLOCK * 8
  uint32_t nulls;  // bit u = lane register u is NULL
};
__host__ __device__ inline size_t tile_regfile_bytes(int n_wide, int n_narrow) { return (size_t)BLOCK * ((size_t)n_wide * 16 + (size_t)n_narrow * 8); }

__device__ __forceinline__ void tp_fetch(const TileProgram& p, const TileRegs& t, uint32_t opnd, uint64_t& lo, uint64_t& hi, bool& isnull) {
  if (opnd & TP_LIT) {  // wave-uniform: scalar loads from the kernel-argument segment
    const uint32_t k = opnd & 0x7Fu;
    lo = p.lit_lo[k];
    hi = p.lit_hi[k];
    isnull = (p.lit_nulls >> k) & 1u;
  } else if ((int)opnd < p.n_wide) {
    const uint4 v = *reinterpret_cast<const uint4*>(t.wide + (size_t)opnd * (BLOCK * 16));
    lo = ((uint64_t)v.y << 32) | v.x;
    hi = ((uint64_t)v.w << 32) | v.z;
    isnull = (t.nulls >> opnd) & 1u;
  } else {
    lo = *reinterpret_cast<const uint64_t*>(t.narrow + (size_t)((int)opnd - p.n_wide) * (BLOCK * 8));
    hi = (uint64_t)((int64_t)lo >> 63);
    isnull = (t.nulls >> opnd) & 1u;
  }
}
__device__ __forceinline__ void tp_store(const TileProgram& p, TileRegs& t, uint32_t reg, uint64_t lo, uint64_t hi, bool isnull) {
  if ((int)reg < p.n_wide) {
    *reinterpret_cast<uint4*>(t.wide + (size_t)reg * (BLOCK * 16)) = make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
  } else {
    *reinterpret_cast<uint64_t*>(t.narrow + (size_t)((int)reg - p.n_wide) * (BLOCK * 8)) = lo;
  }
  t.nulls = (t.nulls & ~(1u << reg)) | ((isnull ? 1u : 0u) << reg);
}
// staged column values -> lane registers (widening as rp_commit_row)
__device__ __forceinline__ void tp_commit_row(const TileProgram& p, const RpRaw& w, TileRegs& t) {
#pragma unroll
  for (int c = 0; c < RP_MAX_COLS; c++) {
    if (c < p.n_cols) {
      uint64_t lo = w.lo[c], hi = w.hi[c];
      switch (p.col_kind[c]) {
        case RPL_I32: lo = (uint64_t)(int64_t)(int32_t)(uint32_t)lo; hi = (uint64_t)((int64_t)lo >> 63); break;
        case RPL_I64: hi = (uint64_t)((int64_t)lo >> 63); break;
        case RPL_BOOL: lo = (lo >> hi) & 1ull; hi = 0; break;
        default: break;
      }
      tp_store(p, t, p.col_reg[c], lo, hi, (w.nulls >> c) & 1u);
    }
  }
}
// issue the loads of `row` (as rp_issue_row, for a TileProgram)
__device__ __forceinline__ void tp_issue_row(const TileProgram& p, int64_t row, RpRaw& w) {
  uint32_t nulls = 0;
#pragma unroll
  for (int c = 0; c < RP_MAX_COLS; c++) {
    if (c < p.n_cols) {
      const void* d = p.col_data[c];
      uint64_t lo = 0, hi = 0;
      switch (p.col_kind[c]) {
        case RPL_I32: case RPL_U32: lo = ((const uint32_t*)d)[row]; break;
        case RPL_I64: case RPL_U64: case RPL_F64: lo = ((const uint64_t*)d)[row]; break;
        case RPL_U8: lo = ((const uint8_t*)d)[row]; break;
        case RPL_I128: { const uint64_t* q = (const uint64_t*)d + 2 * row; lo = q[0]; hi = q[1]; break; }
        default: lo = ((const uint64_t*)d)[row >> 6]; hi = (uint64_t)(row & 63); break;  // RPL_BOOL: word + bit position
      }
      w.lo[c] = lo;
      w.hi[c] = hi;
      const uint64_t* v = p.col_valid[c];
      if (v) nulls |= (bit_at(v, row) ? 0u : 1u) << c;
    }
  }
  w.nulls = nulls;
}
// run instructions [k0, k1) for the current row
__device__ __forceinline__ void tp_exec(const TileProgram& p, int k0, int k1, TileRegs& t) {
  for (int k = k0; k < k1; k++) {
    const RpIns in = p.ins[k];
    uint64_t alo, ahi, blo = 0, bhi = 0;
    bool an, bn = false;
    tp_fetch(p, t, in.a, alo, ahi, an);
    const bool unary = in.op == RP_SEXT32 || in.op == RP_SEXT64 || in.op == RP_I2F || in.op == RP_F64ORD || in.op == RP_DATE_PART || (in.op >= RP_NOT && in.op <= RP_MOV);
    if (!unary) tp_fetch(p, t, in.b, blo, bhi, bn);
    uint64_t olo = 0, ohi = 0;
    bool on = an | bn;
    switch (in.op) {
      case RP_ADD: { u128 v = (((u128)ahi << 64) | alo) + (((u128)bhi << 64) | blo); olo = (uint64_t)v; ohi = (uint64_t)(v >> 64); break; }
      case RP_SUB: { u128 v = (((u128)ahi << 64) | alo) - (((u128)bhi << 64) | blo); olo = (uint64_t)v; ohi = (uint64_t)(v >> 64); break; }
      case RP_MUL: { u128 v = (((u128)ahi << 64) | alo) * (((u128)bhi << 64) | blo); olo = (uint64_t)v; ohi = (uint64_t)(v >> 64); break; }
      case RP_SEXT32: { int64_t s = (int64_t)(int32_t)(uint32_t)alo; olo = (uint64_t)s; ohi = (uint64_t)(s >> 63); on = an; break; }
      case RP_SEXT64: olo = alo; ohi = (uint64_t)((int64_t)alo >> 63); on = an; break;
      case RP_FADD: olo = (uint64_t)__double_as_longlong(__longlong_as_double((long long)alo) + __longlong_as_double((long long)blo)); break;
      case RP_FSUB: olo = (uint64_t)__double_as_longlong(__longlong_as_double((long long)alo) - __longlong_as_double((long long)blo)); break;
      case RP_FMUL: olo = (uint64_t)__double_as_longlong(__longlong_as_double((long long)alo) * __longlong_as_double((long long)blo)); break;
      case RP_I2F: olo = (uint64_t)__double_as_longlong((double)(int64_t)alo); on = an; break;
      case RP_F64ORD: { int64_t s = rp_f64_ordered(alo); olo = (uint64_t)s; ohi = (uint64_t)(s >> 63); on = an; break; }
      case RP_DATE_PART: { int64_t s = (int64_t)date32_part((int32_t)(uint32_t)alo, (int)in.aux); olo = (uint64_t)s; ohi = (uint64_t)(s >> 63); on = an; break; }
      case RP_CMP: olo = rp_cmp128(in.aux, (i128)(((u128)ahi << 64) | alo), (i128)(((u128)bhi << 64) | blo)) ? 1ull : 0ull; break;
      case RP_FCMP: olo = rp_cmp128(in.aux, (i128)rp_f64_ordered(alo), (i128)rp_f64_ordered(blo)) ? 1ull : 0ull; break;
      case RP_AND: {
        bool at = !an && (alo & 1), af = !an && !(alo & 1), bt = !bn && (blo & 1), bf = !bn && !(blo & 1);
        olo = (at && bt) ? 1ull : 0ull;
        on = !((at && bt) || af || bf);
        break;
      }
      case RP_OR: {
        bool at = !an && (alo & 1), af = !an && !(alo & 1), bt = !bn && (blo & 1), bf = !bn && !(blo & 1);
        olo = (at || bt) ? 1ull : 0ull;
        on = !(at || bt || (af && bf));
        break;
      }
      case RP_NOT: olo = (alo & 1) ^ 1ull; on = an; break;
      case RP_IS_NULL: olo = an ? 1ull : 0ull; on = false; break;
      case RP_IS_NOT_NULL: olo = an ? 0ull : 1ull; on = false; break;
      case RP_GATE: { const bool g = (blo & 1) != 0; olo = g ? alo : 0ull; ohi = g ? ahi : 0ull; on = g && an; break; }
      case RP_MERGE: olo = alo | blo; ohi = ahi | bhi; break;
      default: olo = alo; ohi = ahi; on = an; break;  // RP_MOV
    }
    tp_store(p, t, in.dst, olo, ohi, on);
  }
}
// FilterExec semantics of a predicate operand: NULL => dropped
__device__ __forceinline__ bool tp_true(const TileProgram& p, const TileRegs& t, uint32_t opnd) {
  uint64_t lo, hi;
  bool n;
  tp_fetch(p, t, opnd, lo, hi, n);
  return !n && (lo & 1ull);
}

}  // namespace dfgpu
