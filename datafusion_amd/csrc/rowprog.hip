// rowprog.hip — host compiler: dfgpu_expr forest -> RowProgram (see rowprog.hpp).
//
// Value numbering gives common-subexpression elimination across the predicate and all outputs
// (the reference's planner does the same at plan level: `__common_expr_1` in
// sqllogictest/test_files/tpch/plans/q1.slt.part:45-46); literal-only subtrees are folded with
// the column-at-a-time evaluator's own scalar path so both evaluators agree bit for bit.
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

int RowProgramCompiler::emit(uint8_t op, int a, int b, uint32_t aux, int slot, bool lit_null) {
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
  return RpValue{emit(0xFF, -1, -1, 0, slot), in_.cols[idx].field};
}

RpValue RowProgramCompiler::literal(const dfgpu_field& f, uint64_t lo, uint64_t hi, bool is_null) {
  if (is_null) lo = hi = 0;
  if (f.type == DFGPU_FLOAT64 || f.type == DFGPU_BOOL) hi = 0;
  int slot = -1;
  for (size_t i = 0; i < lits_.size(); i++)
    if (lits_[i].first == lo && lits_[i].second == hi) slot = (int)i;
  if (slot < 0) {
    lits_.push_back({lo, hi});
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
  return RpValue{emit(RP_MUL, x.id, m.id, 0), x.type};
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
    DFGPU_CHECK(to.scale >= fs, "decimal scale-down cast not supported on the GPU path");
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
      return RpValue{emit(iop, a.id, b.id, 0), out};
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
    case DFGPU_EXPR_LITERAL: return literal(n.field, n.lit_lo, n.lit_hi, n.is_null != 0);
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
    case DFGPU_EXPR_ADD: case DFGPU_EXPR_SUB: case DFGPU_EXPR_MUL:
    case DFGPU_EXPR_EQ: case DFGPU_EXPR_NE: case DFGPU_EXPR_LT: case DFGPU_EXPR_LE: case DFGPU_EXPR_GT: case DFGPU_EXPR_GE:
    case DFGPU_EXPR_AND: case DFGPU_EXPR_OR: {
      RpValue a = lower(e, n.left);
      RpValue b = lower(e, n.right);
      return lower_binary(n.op, a, b);
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
  return true;
}

}  // namespace dfgpu
