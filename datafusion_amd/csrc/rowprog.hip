// rowprog.hip — host compiler: dfgpu_expr forest -> RowProgram (see rowprog.hpp).
//
// Value numbering gives common-subexpression elimination across the predicate and all outputs
// (the reference's planner does the same at plan level: `__common_expr_1` in
// sqllogictest/test_files/tpch/plans/q1.slt.part:45-46); literal-only subtrees are folded with
// the column-at-a-time evaluator's own scalar path so both evaluators agree bit for bit.
#include <cstdio>
#include <cstdlib>
#include <string>

#include "rowprog_host.hpp"

namespace dfgpu {

static dfgpu_field mk(int type, int p = 0, int s = 0) {
  dfgpu_field f{};
  f.type = type;
  f.precision = p;
  f.scale = s;
  f.nullable = 1;
  return f;
}
static i128 pow10(int k) {
  i128 m = 1;
  for (int i = 0; i < k; i++) m *= 10;
  return m;
}
static int physical_type(int t) { return t == DFGPU_DATE32 ? DFGPU_INT32 : t; }

int RowProgramCompiler::emit(uint8_t op, int a, int b, uint32_t aux, int slot, bool lit_null, bool wide) {
  auto key = std::make_tuple((int)op, a, b, aux, slot);
  auto it = cse_.find(key);
  if (it != cse_.end()) return it->second;
  Val v;
  v.op = op;
  v.a = a;
  v.b = b;
  v.aux = aux;
  v.slot = slot;
  v.lit_null = lit_null;
  v.wide = wide;
  v.seg = (op == RP_LIT) ? 0 : (op == 0xFF ? -1 : seg_);
  vals_.push_back(v);
  int id = (int)vals_.size() - 1;
  cse_[key] = id;
  return id;
}

RpValue RowProgramCompiler::column(int idx) {
  DFGPU_CHECK(idx >= 0 && idx < (int)in_.cols.size(), "Column index out of range");
  int slot = -1;
  for (size_t s = 0; s < slot_col_.size(); s++)
    if (slot_col_[s] == idx) slot = (int)s;
  if (slot < 0) {
    slot_col_.push_back(idx);
    slot = (int)slot_col_.size() - 1;
    if (slot >= RP_MAX_COLS) fail("more than " + std::to_string(RP_MAX_COLS) + " input columns");
  }
  const int ct = in_.cols[idx].field.type;
  if (ct == DFGPU_UTF8) fail("string columns are evaluated column-at-a-time");
  return RpValue{emit(0xFF, -1, -1, 0, slot, false, ct == DFGPU_DECIMAL128 || ct == DFGPU_UINT64), in_.cols[idx].field};
}

RpValue RowProgramCompiler::literal(const dfgpu_field& f, uint64_t lo, uint64_t hi, bool is_null) {
  if (is_null) lo = hi = 0;
  if (f.type == DFGPU_FLOAT64 || f.type == DFGPU_BOOL) hi = 0;
  // a NULL literal never shares a slot with the value literal of the same bits (zero): the LDS register-file program
  // keeps NULL-ness per slot (TileProgram::lit_nulls)
  int slot = -1;
  for (size_t i = 0; i < lits_.size(); i++)
    if (lits_[i].first == lo && lits_[i].second == hi && lit_null_[i] == is_null) slot = (int)i;
  if (slot < 0) {
    lits_.push_back({lo, hi});
    lit_null_.push_back(is_null);
    slot = (int)lits_.size() - 1;
    if (slot >= RP_MAX_LITS) fail("more than " + std::to_string(RP_MAX_LITS) + " distinct literals");
  }
  return RpValue{emit(RP_LIT, is_null ? 1 : 0, -1, (uint32_t)slot, slot, is_null), f};
}
RpValue RowProgramCompiler::literal_i128(const dfgpu_field& f, i128 v) {
  return literal(f, (uint64_t)(u128)v, (uint64_t)((u128)v >> 64), false);
}

RpValue RowProgramCompiler::rescale(RpValue x, int by_digits) {
  if (by_digits == 0) return x;
  if (is_literal(x)) {
    const Val& v = vals_[x.id];
    if (v.lit_null) return x;
    u128 bits = ((u128)lits_[v.slot].second << 64) | lits_[v.slot].first;
    return literal_i128(x.type, (i128)(bits * (u128)pow10(by_digits)));
  }
  RpValue m = literal_i128(mk(DFGPU_DECIMAL128, 38, 0), pow10(by_digits));
  return RpValue{emit(RP_MUL, x.id, m.id, 0, -1, false, true), x.type};
}

// fold `op` over literal operands through expr.hip's scalar path (no kernel is launched for scalars)
static Datum fold_with_evaluator(const std::vector<dfgpu_expr_node>& nodes) {
  Table empty;
  dfgpu_expr e{nodes.data(), (int)nodes.size(), (int)nodes.size() - 1};
  Datum d = evaluate(e, empty);
  DFGPU_CHECK(d.scalar, "constant folding did not produce a scalar");
  return d;
}

RpValue RowProgramCompiler::lower_cast(const dfgpu_field& to, RpValue x) {
  const dfgpu_field from = x.type;
  if (same_field_type(from, to)) return x;
  if (is_literal(x)) {
    const Val& v = vals_[x.id];
    std::vector<dfgpu_expr_node> nodes(2);
    nodes[0] = dfgpu_expr_node{};
    nodes[0].op = DFGPU_EXPR_LITERAL;
    nodes[0].left = nodes[0].right = -1;
    nodes[0].field = from;
    nodes[0].is_null = v.lit_null;
    nodes[0].lit_lo = lits_[v.slot].first;
    nodes[0].lit_hi = lits_[v.slot].second;
    nodes[1] = dfgpu_expr_node{};
    nodes[1].op = DFGPU_EXPR_CAST;
    nodes[1].left = 0;
    nodes[1].right = -1;
    nodes[1].field = to;
    Datum d = fold_with_evaluator(nodes);
    return literal(to, d.lit_lo, d.lit_hi, d.scalar_null);
  }
  const int ft = physical_type(from.type);
  auto unsupported = [&]() -> RpValue { throw Error("cast " + type_name(from) + " -> " + type_name(to) + " is not supported on the GPU path"); };
  if (to.type == DFGPU_DECIMAL128) {
    if (!(ft == DFGPU_INT32 || ft == DFGPU_INT64 || ft == DFGPU_UINT8 || ft == DFGPU_DECIMAL128)) return unsupported();
    int fs = from.type == DFGPU_DECIMAL128 ? from.scale : 0;
    if (to.scale < fs) {  // rounding + precision check raise errors a row program cannot: column-at-a-time (expr.hip)
      fail("decimal scale-down casts are evaluated column-at-a-time");
      return literal(to, 0, 0, true);
    }
    RpValue r = rescale(x, to.scale - fs);
    r.type = to;
    return r;
  }
  if (to.type == DFGPU_INT64) {
    if (!(ft == DFGPU_INT32 || ft == DFGPU_UINT8 || ft == DFGPU_UINT32)) return unsupported();
    return RpValue{x.id, to};  // the widened register already holds the value
  }
  if (to.type == DFGPU_FLOAT64) {
    if (!(ft == DFGPU_INT32 || ft == DFGPU_INT64)) return unsupported();
    return RpValue{emit(RP_I2F, x.id, x.id, 0), to};
  }
  if ((to.type == DFGPU_INT32 || to.type == DFGPU_DATE32) && ft == DFGPU_INT32) return RpValue{x.id, to};
  return unsupported();
}

RpValue RowProgramCompiler::lower_binary(int op, RpValue a, RpValue b) {
  const dfgpu_field lt = a.type, rtp = b.type;
  if (op == DFGPU_EXPR_AND || op == DFGPU_EXPR_OR) {
    DFGPU_CHECK(lt.type == DFGPU_BOOL && rtp.type == DFGPU_BOOL, "AND/OR operands must be Boolean");
    return RpValue{emit(op == DFGPU_EXPR_AND ? RP_AND : RP_OR, a.id, b.id, 0), mk(DFGPU_BOOL)};
  }
  const bool is_cmp = op >= DFGPU_EXPR_EQ && op <= DFGPU_EXPR_GE;
  if (lt.type == DFGPU_UTF8 || rtp.type == DFGPU_UTF8) {  // already failed in column() / literal: keep lowering harmless
    fail("string operands are evaluated column-at-a-time");
    return literal(mk(DFGPU_BOOL), 0, 0, true);
  }
  if (is_literal(a) && is_literal(b)) {
    std::vector<dfgpu_expr_node> nodes(3);
    const RpValue* side[2] = {&a, &b};
    for (int i = 0; i < 2; i++) {
      const Val& v = vals_[side[i]->id];
      nodes[i] = dfgpu_expr_node{};
      nodes[i].op = DFGPU_EXPR_LITERAL;
      nodes[i].left = nodes[i].right = -1;
      nodes[i].field = side[i]->type;
      nodes[i].is_null = v.lit_null;
      nodes[i].lit_lo = lits_[v.slot].first;
      nodes[i].lit_hi = lits_[v.slot].second;
    }
    nodes[2] = dfgpu_expr_node{};
    nodes[2].op = op;
    nodes[2].left = 0;
    nodes[2].right = 1;
    Datum d = fold_with_evaluator(nodes);
    return literal(d.col.field, d.lit_lo, d.lit_hi, d.scalar_null);
  }
  if (is_cmp) {
    if (lt.type == DFGPU_DECIMAL128 && rtp.type == DFGPU_DECIMAL128) {
      int s = std::max(lt.scale, rtp.scale);
      a = rescale(a, s - lt.scale);
      b = rescale(b, s - rtp.scale);
      return RpValue{emit(RP_CMP, a.id, b.id, (uint32_t)op), mk(DFGPU_BOOL)};
    }
    int l = physical_type(lt.type), r = physical_type(rtp.type);
    DFGPU_CHECK(l == r, "comparison operand types differ: " + type_name(lt) + " vs " + type_name(rtp));
    switch (l) {
      case DFGPU_INT32: case DFGPU_INT64: case DFGPU_UINT8: case DFGPU_UINT32: case DFGPU_UINT64: case DFGPU_DECIMAL128:
        return RpValue{emit(RP_CMP, a.id, b.id, (uint32_t)op), mk(DFGPU_BOOL)};
      case DFGPU_FLOAT64:
        return RpValue{emit(RP_FCMP, a.id, b.id, (uint32_t)op), mk(DFGPU_BOOL)};
    }
    throw Error("comparison on " + type_name(lt) + " is not supported on the GPU path");
  }
  dfgpu_field out = arith_result_type(op, lt, rtp);
  const uint8_t iop = op == DFGPU_EXPR_ADD ? RP_ADD : op == DFGPU_EXPR_SUB ? RP_SUB : RP_MUL;
  switch (out.type) {
    case DFGPU_DECIMAL128: {
      if (op != DFGPU_EXPR_MUL) {
        a = rescale(a, out.scale - lt.scale);
        b = rescale(b, out.scale - rtp.scale);
      }
      return RpValue{emit(iop, a.id, b.id, 0, -1, false, true), out};
    }
    case DFGPU_INT32: return RpValue{emit(RP_SEXT32, emit(iop, a.id, b.id, 0), -2, 0), out};
    case DFGPU_INT64: return RpValue{emit(RP_SEXT64, emit(iop, a.id, b.id, 0), -2, 0), out};
    case DFGPU_FLOAT64: {
      const uint8_t fop = op == DFGPU_EXPR_ADD ? RP_FADD : op == DFGPU_EXPR_SUB ? RP_FSUB : RP_FMUL;
      return RpValue{emit(fop, a.id, b.id, 0), out};
    }
  }
  throw Error("arithmetic result type not supported");
}

RpValue RowProgramCompiler::lower(const dfgpu_expr& e, int idx) {
  DFGPU_CHECK(idx >= 0 && idx < e.n_nodes, "expression node index out of range");
  const dfgpu_expr_node& n = e.nodes[idx];
  switch (n.op) {
    case DFGPU_EXPR_COLUMN: return column(n.column);
    case DFGPU_EXPR_LITERAL:
      if (n.field.type == DFGPU_UTF8) fail("string literals are evaluated column-at-a-time");
      return literal(n.field, n.lit_lo, n.lit_hi, n.is_null != 0);
    case DFGPU_EXPR_LIKE: case DFGPU_EXPR_ILIKE: {
      lower(e, n.left);
      fail("LIKE is evaluated column-at-a-time");
      return literal(mk(DFGPU_BOOL), 0, 0, true);
    }
    case DFGPU_EXPR_SUBSTR: {
      lower(e, n.left);
      fail("substr is evaluated column-at-a-time");
      return literal(mk(DFGPU_INT32), 0, 0, true);
    }
    case DFGPU_EXPR_CAST: return lower_cast(n.field, lower(e, n.left));
    case DFGPU_EXPR_NOT: {
      RpValue a = lower(e, n.left);
      DFGPU_CHECK(a.type.type == DFGPU_BOOL, "NOT operand must be Boolean");
      return RpValue{emit(RP_NOT, a.id, -2, 0), mk(DFGPU_BOOL)};
    }
    case DFGPU_EXPR_IS_NULL:
    case DFGPU_EXPR_IS_NOT_NULL: {
      RpValue a = lower(e, n.left);
      return RpValue{emit(n.op == DFGPU_EXPR_IS_NULL ? RP_IS_NULL : RP_IS_NOT_NULL, a.id, -2, 0), mk(DFGPU_BOOL)};
    }
    case DFGPU_EXPR_DATE_PART: {
      RpValue a = lower(e, n.left);
      DFGPU_CHECK(a.type.type == DFGPU_DATE32, "date_part: the GPU path takes a Date32 argument");
      DFGPU_CHECK(n.column >= DFGPU_DATE_PART_YEAR && n.column <= DFGPU_DATE_PART_DAY, "date_part: the GPU path extracts YEAR, MONTH or DAY");
      return RpValue{emit(RP_DATE_PART, a.id, -2, (uint32_t)n.column), mk(DFGPU_INT32)};
    }
    case DFGPU_EXPR_DIV: case DFGPU_EXPR_MOD: {
      // a zero divisor is an error the row programs cannot raise: division stays column-at-a-time (expr.hip eval_divmod)
      RpValue a = lower(e, n.left);
      RpValue b = lower(e, n.right);
      fail("division is evaluated column-at-a-time");
      return literal(arith_result_type(n.op, a.type, b.type), 0, 0, true);
    }
    case DFGPU_EXPR_ADD: case DFGPU_EXPR_SUB: case DFGPU_EXPR_MUL:
    case DFGPU_EXPR_EQ: case DFGPU_EXPR_NE: case DFGPU_EXPR_LT: case DFGPU_EXPR_LE: case DFGPU_EXPR_GT: case DFGPU_EXPR_GE:
    case DFGPU_EXPR_AND: case DFGPU_EXPR_OR: {
      RpValue a = lower(e, n.left);
      RpValue b = lower(e, n.right);
      return lower_binary(n.op, a, b);
    }
    case DFGPU_EXPR_CASE: {
      // binary ops only (the register allocator and the three program forms stay two-operand):
      //   t = c AND (c IS NOT NULL)   TRUE exactly where the WHEN condition is TRUE, never NULL
      //   result = MERGE(GATE(then, t), GATE(else, NOT t))
      RpValue c = lower(e, n.column);
      DFGPU_CHECK(c.type.type == DFGPU_BOOL, "CASE WHEN condition must be Boolean");
      RpValue a = lower(e, n.left);
      RpValue b = n.right >= 0 ? lower(e, n.right) : literal(a.type, 0, 0, true);
      DFGPU_CHECK(same_field_type(a.type, b.type), "CASE branch types differ: " + type_name(a.type) + " vs " + type_name(b.type) + " (the planner inserts casts)");
      const bool wide = a.type.type == DFGPU_DECIMAL128 || a.type.type == DFGPU_UINT64;
      const int t = emit(RP_AND, c.id, emit(RP_IS_NOT_NULL, c.id, -2, 0), 0);
      const int nt = emit(RP_NOT, t, -2, 0);
      const int ga = emit(RP_GATE, a.id, t, 0, -1, false, wide);
      const int gb = emit(RP_GATE, b.id, nt, 0, -1, false, wide);
      dfgpu_field out = a.type;
      out.nullable = 1;
      return RpValue{emit(RP_MERGE, ga, gb, 0, -1, false, wide), out};
    }
  }
  throw Error("unsupported expression op " + std::to_string(n.op));
}

void RowProgramCompiler::set_predicate(const dfgpu_expr& e) {
  DFGPU_CHECK(pred_ < 0 && outs_.empty(), "the predicate must be set first and once");
  seg_ = 1;
  RpValue p = lower(e, e.root);
  seg_ = 2;
  DFGPU_CHECK(p.type.type == DFGPU_BOOL, "Cannot create filter with non-boolean predicate");
  pred_ = p.id;
}

int RowProgramCompiler::add_output(const dfgpu_expr& e) {
  outs_.push_back(lower(e, e.root));
  return (int)outs_.size() - 1;
}

void RowProgramCompiler::convert_output(int out, RpOp op, const dfgpu_field& new_type) {
  outs_[out] = RpValue{emit(op, outs_[out].id, -2, 0), new_type};
}

bool RowProgramCompiler::finish(CompiledProgram& cp, std::string& why) {
  if (failed_) {
    why = why_;
    return false;
  }
  const int nv = (int)vals_.size();
  // program order: literals, predicate segment, output segment (each in emission order, which is
  // topological because operands are always emitted before their users)
  std::vector<int> order;
  for (int seg = 0; seg <= 2; seg++)
    for (int v = 0; v < nv; v++)
      if (vals_[v].seg == seg) order.push_back(v);
  // a value first emitted in the output segment may be needed by ... nothing earlier: fine.  A value
  // emitted in the predicate segment and reused by outputs stays live (last_use below).
  std::vector<int> pos(nv, -1);
  for (size_t i = 0; i < order.size(); i++) pos[order[i]] = (int)i;
  const int n_ins = (int)order.size();
  if (n_ins > RP_MAX_INS) {
    why = "program needs " + std::to_string(n_ins) + " instructions";
    return false;
  }
  const int INF = 1 << 30;
  std::vector<int> last_use(nv, -1);
  for (int v = 0; v < nv; v++) {
    const Val& x = vals_[v];
    if (x.op == 0xFF || x.op == RP_LIT) continue;
    if (x.a >= 0) last_use[x.a] = std::max(last_use[x.a], pos[v]);
    if (x.b >= 0) last_use[x.b] = std::max(last_use[x.b], pos[v]);
  }
  if (pred_ >= 0) last_use[pred_] = INF;
  for (const RpValue& o : outs_) last_use[o.id] = INF;
  // registers: columns own 0..n_cols-1, literals are pinned (loaded once per thread), the rest by linear scan
  const int n_cols = (int)slot_col_.size();
  std::vector<int> reg(nv, -1);
  std::vector<bool> busy(RP_NREG, false);
  for (int v = 0; v < nv; v++)
    if (vals_[v].op == 0xFF) {
      reg[v] = vals_[v].slot;
      busy[reg[v]] = true;
    }
  auto alloc = [&]() -> int {
    for (int r = n_cols; r < RP_NREG; r++)
      if (!busy[r]) {
        busy[r] = true;
        return r;
      }
    return -1;
  };
  cp = CompiledProgram{};
  cp.n_regs = n_cols;
  RowProgram& P = cp.prog;
  int n_prologue = 0, n_pred_end = 0;
  for (int i = 0; i < n_ins; i++) {
    const int v = order[i];
    const Val& x = vals_[v];
    if (x.seg == 0) n_prologue = i + 1;
    if (x.seg <= 1) n_pred_end = i + 1;
    // operands die here unless used later; literals and columns never die
    auto release = [&](int o) {
      if (o < 0) return;
      if (vals_[o].op == 0xFF || vals_[o].op == RP_LIT) return;
      if (last_use[o] == i && reg[o] >= 0) busy[reg[o]] = false;
    };
    if (x.op != RP_LIT) {
      release(x.a);
      if (x.b != x.a) release(x.b);
    }
    int r = alloc();
    if (r < 0) {
      why = "expression forest needs more than " + std::to_string(RP_NREG) + " registers";
      return false;
    }
    reg[v] = r;
    cp.n_regs = std::max(cp.n_regs, r + 1);
    RpIns ins{};
    ins.op = x.op;
    ins.dst = (uint8_t)r;
    if (x.op == RP_LIT) {
      ins.a = x.lit_null ? 1 : 0;
      ins.b = 0;
      ins.aux = (uint32_t)x.slot;
    } else {
      ins.a = (uint8_t)reg[x.a];
      ins.b = (uint8_t)(x.b >= 0 ? reg[x.b] : reg[x.a]);
      ins.aux = x.aux;
    }
    P.ins[i] = ins;
    if (last_use[v] < 0 && last_use[v] != INF) busy[r] = false;  // dead value (cannot happen for reachable nodes)
  }
  if (n_pred_end < n_prologue) n_pred_end = n_prologue;
  P.n_ins = n_ins;
  P.n_cols = n_cols;
  for (int s = 0; s < n_cols; s++) {
    const Column& c = in_.cols[slot_col_[s]];
    P.col_data[s] = c.ptr();
    P.col_valid[s] = c.valid_words();
    int k = 0;
    switch (c.field.type) {
      case DFGPU_INT32: case DFGPU_DATE32: k = RPL_I32; break;
      case DFGPU_INT64: k = RPL_I64; break;
      case DFGPU_UINT8: k = RPL_U8; break;
      case DFGPU_UINT32: k = RPL_U32; break;
      case DFGPU_UINT64: k = RPL_U64; break;
      case DFGPU_DECIMAL128: k = RPL_I128; break;
      case DFGPU_FLOAT64: k = RPL_F64; break;
      case DFGPU_BOOL: k = RPL_BOOL; break;
      default: throw Error("column type " + type_name(c.field) + " is not supported on the GPU path");
    }
    P.col_kind[s] = (uint8_t)k;
    cp.input_bytes_per_row += c.field.type == DFGPU_BOOL ? 1 : type_width(c.field.type);
  }
  for (size_t i = 0; i < lits_.size(); i++) {
    P.lit_lo[i] = lits_[i].first;
    P.lit_hi[i] = lits_[i].second;
  }
  cp.n_prologue = n_prologue;
  cp.n_pred_end = n_pred_end;
  cp.pred_reg = pred_ >= 0 ? reg[pred_] : -1;
  for (const RpValue& o : outs_) {
    cp.out_regs.push_back(reg[o.id]);
    cp.out_types.push_back(o.type);
  }
  // ---- the same forest as a TileProgram: literals become operands, lane registers are allocated per width class
  {
    TileProgram& T = cp.tile;
    T = TileProgram{};
    std::vector<int> treg(nv, -1);      // class-local register index
    std::vector<bool> wbusy(64, false), nbusy(64, false);
    int n_wide = 0, n_narrow = 0;
    auto talloc = [&](bool wide) -> int {
      std::vector<bool>& busy_c = wide ? wbusy : nbusy;
      for (int r2 = 0; r2 < 64; r2++)
        if (!busy_c[r2]) {
          busy_c[r2] = true;
          int& hi = wide ? n_wide : n_narrow;
          hi = std::max(hi, r2 + 1);
          return r2;
        }
      return -1;
    };
    for (int v = 0; v < nv; v++)
      if (vals_[v].op == 0xFF) treg[v] = talloc(vals_[v].wide);
    struct Pending { int v; };
    std::vector<Pending> emitted;
    for (int i = 0; i < n_ins; i++) {
      const int v = order[i];
      const Val& x = vals_[v];
      if (x.op == RP_LIT) continue;
      auto release = [&](int o) {
        if (o < 0) return;
        if (vals_[o].op == 0xFF || vals_[o].op == RP_LIT) return;
        if (last_use[o] == i && treg[o] >= 0) (vals_[o].wide ? wbusy : nbusy)[treg[o]] = false;
      };
      release(x.a);
      if (x.b != x.a) release(x.b);
      treg[v] = talloc(x.wide);
      emitted.push_back({v});
    }
    const bool fits = n_wide + n_narrow <= 32 && (int)emitted.size() <= RP_MAX_INS;
    if (fits) {
      auto opnd = [&](int v) -> uint8_t {
        if (vals_[v].op == RP_LIT) return (uint8_t)(TP_LIT | vals_[v].slot);
        return (uint8_t)(vals_[v].wide ? treg[v] : n_wide + treg[v]);
      };
      T.n_ins = 0;
      T.n_pred_end = 0;
      for (const Pending& e : emitted) {
        const Val& x = vals_[e.v];
        RpIns ins{};
        ins.op = x.op;
        ins.dst = opnd(e.v);
        ins.a = opnd(x.a);
        ins.b = x.b >= 0 ? opnd(x.b) : ins.a;
        ins.aux = x.aux;
        T.ins[T.n_ins++] = ins;
        if (x.seg <= 1) T.n_pred_end = T.n_ins;
      }
      T.n_cols = n_cols;
      for (int v = 0; v < nv; v++)
        if (vals_[v].op == 0xFF) T.col_reg[vals_[v].slot] = opnd(v);
      for (int sl = 0; sl < n_cols; sl++) {
        T.col_data[sl] = P.col_data[sl];
        T.col_valid[sl] = P.col_valid[sl];
        T.col_kind[sl] = P.col_kind[sl];
      }
      for (size_t i = 0; i < lits_.size(); i++) {
        T.lit_lo[i] = lits_[i].first;
        T.lit_hi[i] = lits_[i].second;
      }
      for (int v = 0; v < nv; v++)
        if (vals_[v].op == RP_LIT && vals_[v].lit_null) T.lit_nulls |= 1u << vals_[v].slot;
      T.n_wide = n_wide;
      T.n_narrow = n_narrow;
      cp.tile_pred = pred_ >= 0 ? opnd(pred_) : -1;
      for (const RpValue& o : outs_) cp.tile_outs.push_back(opnd(o.id));
    } else {
      T.n_wide = T.n_narrow = -1;  // no tile form
    }
  }
  // ---- the same forest as HIP source (one `const i128 V<id>` + `const bool N<id>` per value)
  {
    auto V = [](int v) { return "V" + std::to_string(v); };
    auto N = [](int v) { return "N" + std::to_string(v); };
    auto hex = [](uint64_t x) {
      char b[32];
      snprintf(b, sizeof b, "0x%016llxull", (unsigned long long)x);
      return std::string(b);
    };
    static const char* cmp_ops[] = {"==", "!=", "<", "<=", ">", ">="};
    auto cmp_str = [&](uint32_t aux) -> const char* {
      switch (aux) {
        case DFGPU_EXPR_EQ: return cmp_ops[0];
        case DFGPU_EXPR_NE: return cmp_ops[1];
        case DFGPU_EXPR_LT: return cmp_ops[2];
        case DFGPU_EXPR_LE: return cmp_ops[3];
        case DFGPU_EXPR_GT: return cmp_ops[4];
        default: return cmp_ops[5];
      }
    };
    std::string loads, pred_src, outs_src;
    std::vector<bool> maybe_null((size_t)nv, false);
    for (int v = 0; v < nv; v++) {
      const Val& x = vals_[v];
      if (x.op != 0xFF) continue;
      maybe_null[(size_t)v] = P.col_valid[x.slot] != nullptr;
      const std::string sl = std::to_string(x.slot), c = "C" + sl;
      std::string decl, widen;
      switch (P.col_kind[x.slot]) {
        case RPL_I32: decl = "const I32 " + c + " = ((const I32*)a.col[" + sl + "])[i];"; widen = "(i128)" + c; break;
        case RPL_U32: decl = "const U32 " + c + " = ((const U32*)a.col[" + sl + "])[i];"; widen = "(i128)" + c; break;
        case RPL_I64: decl = "const I64 " + c + " = ((const I64*)a.col[" + sl + "])[i];"; widen = "(i128)" + c; break;
        case RPL_U64: case RPL_F64: decl = "const U64 " + c + " = ((const U64*)a.col[" + sl + "])[i];"; widen = "(i128)(u128)" + c; break;
        case RPL_U8: decl = "const U8 " + c + " = ((const U8*)a.col[" + sl + "])[i];"; widen = "(i128)" + c; break;
        case RPL_I128: decl = "const i128 " + c + " = ((const i128*)a.col[" + sl + "])[i];"; widen = c; break;
        default: decl = "const U64 " + c + " = (((const U64*)a.col[" + sl + "])[i >> 6] >> (i & 63)) & 1ull;"; widen = "(i128)" + c; break;
      }
      loads += "    " + decl + "\n    const i128 " + V(v) + " = " + widen + ";\n";
      if (P.col_valid[x.slot]) loads += "    const bool " + N(v) + " = !((a.valid[" + sl + "][i >> 6] >> (i & 63)) & 1ull);\n";
      else loads += "    const bool " + N(v) + " = false;\n";
    }
    for (int i = 0; i < n_ins; i++) {
      const int v = order[i];
      const Val& x = vals_[v];
      std::string st;
      const std::string A = x.a >= 0 ? V(x.a) : "", B = x.b >= 0 ? V(x.b) : A;
      const std::string NA = x.a >= 0 ? N(x.a) : "false", NB = x.b >= 0 ? N(x.b) : NA;
      std::string val, nul = "(" + NA + " || " + NB + ")";
      switch (x.op) {
        case RP_LIT:
          val = "(i128)(((u128)" + hex(lits_[x.slot].second) + " << 64) | (u128)" + hex(lits_[x.slot].first) + ")";
          nul = x.lit_null ? "true" : "false";
          break;
        case RP_ADD: val = "(i128)((u128)" + A + " + (u128)" + B + ")"; break;
        case RP_SUB: val = "(i128)((u128)" + A + " - (u128)" + B + ")"; break;
        case RP_MUL: val = "(i128)((u128)" + A + " * (u128)" + B + ")"; break;
        case RP_SEXT32: val = "(i128)(I32)(U32)(U64)" + A; nul = NA; break;
        case RP_SEXT64: val = "(i128)(I64)(U64)" + A; nul = NA; break;
        case RP_FADD: val = "f2v(v2f(" + A + ") + v2f(" + B + "))"; break;
        case RP_FSUB: val = "f2v(v2f(" + A + ") - v2f(" + B + "))"; break;
        case RP_FMUL: val = "f2v(v2f(" + A + ") * v2f(" + B + "))"; break;
        case RP_I2F: val = "f2v((double)(I64)(U64)" + A + ")"; nul = NA; break;
        case RP_F64ORD: val = "(i128)f64ord((U64)" + A + ")"; nul = NA; break;
        case RP_DATE_PART: val = "(i128)date32_part((I32)(U32)(U64)" + A + ", " + std::to_string(x.aux) + ")"; nul = NA; break;
        case RP_CMP: val = "(i128)(" + A + " " + cmp_str(x.aux) + " " + B + ")"; break;
        case RP_FCMP: val = "(i128)(f64ord((U64)" + A + ") " + cmp_str(x.aux) + " f64ord((U64)" + B + "))"; break;
        case RP_AND:
          val = "(i128)(kt(" + A + "," + NA + ") && kt(" + B + "," + NB + "))";
          nul = "!((kt(" + A + "," + NA + ") && kt(" + B + "," + NB + ")) || kf(" + A + "," + NA + ") || kf(" + B + "," + NB + "))";
          break;
        case RP_OR:
          val = "(i128)(kt(" + A + "," + NA + ") || kt(" + B + "," + NB + "))";
          nul = "!(kt(" + A + "," + NA + ") || kt(" + B + "," + NB + ") || (kf(" + A + "," + NA + ") && kf(" + B + "," + NB + ")))";
          break;
        case RP_NOT: val = "(i128)((" + A + " & 1) ^ 1)"; nul = NA; break;
        case RP_IS_NULL: val = "(i128)(" + NA + ")"; nul = "false"; break;
        case RP_IS_NOT_NULL: val = "(i128)(!" + NA + ")"; nul = "false"; break;
        case RP_GATE: val = "((" + B + " & 1) ? " + A + " : (i128)0)"; nul = "((" + B + " & 1) && " + NA + ")"; break;
        case RP_MERGE: val = "(" + A + " | " + B + ")"; break;
        default: val = A; nul = NA; break;  // RP_MOV
      }
      switch (x.op) {
        case RP_LIT: maybe_null[(size_t)v] = x.lit_null; break;
        case RP_IS_NULL: case RP_IS_NOT_NULL: maybe_null[(size_t)v] = false; break;
        case RP_GATE: maybe_null[(size_t)v] = maybe_null[(size_t)x.a]; break;
        default: maybe_null[(size_t)v] = (x.a >= 0 && maybe_null[(size_t)x.a]) || (x.b >= 0 && maybe_null[(size_t)x.b]); break;
      }
      st = "    const i128 " + V(v) + " = " + val + ";\n    const bool " + N(v) + " = " + nul + ";\n";
      if (x.seg <= 1) pred_src += st;  // literals (seg 0) are declared with the predicate: visible to both segments
      else outs_src += st;
    }
    cp.src_loads = loads;
    cp.src_pred = pred_src;
    cp.src_outs = outs_src;
    cp.src_pred_val = pred_;
    cp.src_maybe_null = maybe_null;
    for (const RpValue& o : outs_) cp.src_out_vals.push_back(o.id);
  }
  if (trace_on("rowprog")) {
    static const char* names[] = {"lit", "add", "sub", "mul", "sext32", "sext64", "fadd", "fsub", "fmul", "i2f", "f64ord", "cmp", "fcmp", "and", "or", "not",
                                  "is_null", "is_not_null", "mov", "gate", "merge", "?", "?"};
    fprintf(stderr, "[rowprog] cols=%d regs=%d ins=%d (prologue %d, predicate end %d, pred reg %d)\n", n_cols, cp.n_regs, n_ins, n_prologue, n_pred_end, cp.pred_reg);
    for (int i = 0; i < n_ins; i++)
      fprintf(stderr, "  %2d: r%-2d = %-8s r%-2d r%-2d aux=%u\n", i, P.ins[i].dst, P.ins[i].op < 23 ? names[P.ins[i].op] : "?", P.ins[i].a, P.ins[i].b, P.ins[i].aux);
    for (size_t o = 0; o < cp.out_regs.size(); o++) fprintf(stderr, "  out%zu = r%d (%s)\n", o, cp.out_regs[o], type_name(cp.out_types[o]).c_str());
    fprintf(stderr, "[tileprog] wide=%d narrow=%d ins=%d (predicate end %d, pred operand %d)\n", cp.tile.n_wide, cp.tile.n_narrow, cp.tile.n_ins, cp.tile.n_pred_end, cp.tile_pred);
    for (int i = 0; i < cp.tile.n_ins; i++)
      fprintf(stderr, "  %2d: %3d = %-8s %3d %3d aux=%u\n", i, cp.tile.ins[i].dst, cp.tile.ins[i].op < 23 ? names[cp.tile.ins[i].op] : "?", cp.tile.ins[i].a, cp.tile.ins[i].b, cp.tile.ins[i].aux);
    for (size_t o = 0; o < cp.tile_outs.size(); o++) fprintf(stderr, "  out%zu = %d\n", o, cp.tile_outs[o]);
  }
  return true;
}

}  // namespace dfgpu
