// rowprog_host.hpp — host-side compiler of PhysicalExpr trees (dfgpu_expr) into a RowProgram
// (rowprog.hpp).  Typing rules are those of expr.hip (the column-at-a-time evaluator), which
// stays the fallback whenever a forest does not fit the register file.
#pragma once
#include <map>
#include <string>
#include <tuple>

#include "internal.hpp"
#include "rowprog.hpp"

namespace dfgpu {

// shared with expr.hip
dfgpu_field arith_result_type(int op, const dfgpu_field& l, const dfgpu_field& r);
bool same_field_type(const dfgpu_field& a, const dfgpu_field& b);

struct RpValue {
  int id = -1;          // SSA value id inside the compiler
  dfgpu_field type{};   // logical type carried by that value
};

struct CompiledProgram {
  RowProgram prog{};
  int n_prologue = 0;   // [0, n_prologue): literal loads, run once per thread
  int n_pred_end = 0;   // [n_prologue, n_pred_end): predicate; [n_pred_end, n_ins): outputs
  int pred_reg = -1;
  int n_regs = 0;       // registers used (highest index + 1): picks the 16- or 32-register kernel instance
  std::vector<int> out_regs;
  std::vector<dfgpu_field> out_types;
  int64_t input_bytes_per_row = 0;  // widths of the referenced input columns (algorithmic bytes)
  // the same forest for the LDS register file interpreter (rowprog.hpp, TileProgram)
  TileProgram tile{};
  int tile_pred = -1;          // operand byte of the predicate, -1 = none
  std::vector<int> tile_outs;  // operand byte per output
  // the same forest as HIP source for runtime-specialised kernels (jit.hip): statements over `i` (row) and
  // `a.col[s]` / `a.valid[s]` defining, per value v, `const i128 V<v>` and `const bool N<v>` (NULL flag)
  std::string src_loads;       // column loads + widening
  std::string src_pred;        // predicate segment
  std::string src_outs;        // output segment
  int src_pred_val = -1;       // value id of the predicate (V<id> / N<id>), -1 = none
  std::vector<int> src_out_vals;
  std::vector<bool> src_maybe_null;  // per value id: can N<id> ever be true?
};

class RowProgramCompiler {
 public:
  explicit RowProgramCompiler(const Table& in) : in_(in) {}
  // the FilterExec predicate evaluated before anything else (optional, at most once, first)
  void set_predicate(const dfgpu_expr& e);
  // an output expression; returns its index in CompiledProgram::out_regs
  int add_output(const dfgpu_expr& e);
  // post-process output `out` with a unary conversion (RP_I2F / RP_F64ORD), in place
  void convert_output(int out, RpOp op, const dfgpu_field& new_type);
  dfgpu_field output_type(int out) const { return outs_[out].type; }
  // false (with a reason) when the forest does not fit: too many columns / registers / instructions
  bool finish(CompiledProgram& cp, std::string& why);

 private:
  struct Val {
    uint8_t op;       // RpOp, or 0xFF = input column
    int a = -1, b = -1;
    uint32_t aux = 0;
    int seg = 2;      // 0 prologue literal, 1 predicate, 2 outputs
    int slot = -1;    // column slot / literal index
    bool lit_null = false;
    bool wide = false;  // needs a 16-byte register (Decimal128 / UInt64); else the value fits 64 bits sign-extended
  };
  const Table& in_;
  std::vector<Val> vals_;
  std::map<std::tuple<int, int, int, uint32_t, int>, int> cse_;
  std::vector<int> slot_col_;                             // slot -> table column
  std::vector<std::pair<uint64_t, uint64_t>> lits_;
  std::vector<bool> lit_null_;                            // per literal slot: SQL NULL
  std::vector<RpValue> outs_;
  int pred_ = -1;
  int seg_ = 2;
  bool failed_ = false;
  std::string why_;

  int emit(uint8_t op, int a, int b, uint32_t aux, int slot = -1, bool lit_null = false, bool wide = false);
  RpValue column(int idx);
  RpValue literal(const dfgpu_field& f, uint64_t lo, uint64_t hi, bool is_null);
  RpValue literal_i128(const dfgpu_field& f, i128 v);
  bool is_literal(const RpValue& v) const { return v.id >= 0 && vals_[v.id].op == RP_LIT; }
  RpValue lower(const dfgpu_expr& e, int idx);
  RpValue lower_cast(const dfgpu_field& to, RpValue x);
  RpValue lower_binary(int op, RpValue a, RpValue b);
  RpValue rescale(RpValue x, int by_digits);
  void fail(const std::string& why) {
    if (!failed_) why_ = why;
    failed_ = true;
  }
};

}  // namespace dfgpu
