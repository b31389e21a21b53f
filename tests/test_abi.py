"""CPU-side checks of the drop-in boundary: libdfgpu.so loads without a GPU, exports every
symbol include/dfgpu.h declares, and refuses to run (loudly) when no MI355X is present."""
import ctypes as C
import os
import re

import pytest

from datafusion_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ABI = int(re.search(r"#define DFGPU_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "dfgpu.h")).read()).group(1))


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "dfgpu.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(dfgpu_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 40
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/dfgpu.h but not exported"
    assert sorted(_lib.SYMBOLS) == names, "python binding list out of sync with the header"


def test_abi_version():
    assert _lib.load().dfgpu_abi_version() == ABI


def test_struct_layouts_match_header(tmp_path):
    """sizes the C compiler gives the ABI structs of include/dfgpu.h (LP64) == the ctypes mirrors"""
    import subprocess
    names = {"dfgpu_field": _lib.Field, "dfgpu_expr_node": _lib.ExprNode, "dfgpu_expr": _lib.Expr, "dfgpu_join_options": _lib.JoinOptions,
             "dfgpu_join_info": _lib.JoinInfo, "dfgpu_kernel_stat": _lib.KernelStat, "dfgpu_agg_spec": _lib.AggSpec,
             "dfgpu_column_view": _lib.ColumnView, "dfgpu_join_filter": _lib.JoinFilter, "dfgpu_parquet_column": _lib.ParquetColumn,
             "dfgpu_parquet_chunk_info": _lib.ParquetChunkInfo}
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "dfgpu.h"\nint main(void){' +
                   "".join(f'printf("{n} %zu\\n", sizeof({n}));' for n in names) +
                   'printf("ArrowSchema %zu\\nArrowArray %zu\\n", sizeof(struct ArrowSchema), sizeof(struct ArrowArray));return 0;}')
    exe = tmp_path / "sizes"
    # the header is plain C: it must compile as strict C99 without a single diagnostic
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    for n, t in names.items():
        assert int(got[n]) == C.sizeof(t), (n, got[n], C.sizeof(t))
    from datafusion_amd.table import ArrowArray, ArrowSchema
    assert int(got["ArrowSchema"]) == C.sizeof(ArrowSchema) == 72 and int(got["ArrowArray"]) == C.sizeof(ArrowArray) == 80


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = _lib.load()
    n = C.c_int(-1)
    assert lib.dfgpu_device_count(C.byref(n)) == 0 and n.value == 0
    assert lib.dfgpu_init((C.c_int * 1)(0), 1) != 0
    assert b"no HIP device" in lib.dfgpu_last_error()
    # operators refuse to run before init
    out = C.c_void_p()
    assert lib.dfgpu_tpch_orders(C.c_double(0.001), C.c_int64(0), C.c_int64(-1), C.byref(out)) != 0
    assert b"dfgpu_init" in lib.dfgpu_last_error()


def test_expression_lowering():
    import pyarrow as pa
    from datafusion_amd.expr import col, lit, lower
    e = (col("l_extendedprice") * (lit(1, pa.decimal128(20, 0)) - col("l_discount")))
    l = lower(e, ["l_orderkey", "l_extendedprice", "l_discount"])
    assert l.c.n_nodes == 5 and l.c.root == 4
    ops = [l.nodes[i].op for i in range(5)]
    assert ops == [1, 2, 1, 11, 12]
    assert l.nodes[0].column == 1 and l.nodes[2].column == 2
    assert l.nodes[1].lit_lo == 1 and l.nodes[1].field.precision == 20
    neg = lower(lit(-5, pa.int64()), [])
    assert neg.nodes[0].lit_lo == 2**64 - 5 and neg.nodes[0].lit_hi == 2**64 - 1


def test_c_program_links_against_the_library(tmp_path):
    """a C translation unit (no C++, no Python) binds the boundary: links libdfgpu.so, reads the ABI version, gets a readable
    error from an operator called before dfgpu_init, and drives the host half of the Parquet scan — what a cgo / Rust `extern "C"`
    binding does"""
    import subprocess
    lib_dir = os.path.dirname(_lib.LIB_PATH)
    src = tmp_path / "bind.c"
    src.write_text(r"""
#include <stdio.h>
#include <string.h>
#include "dfgpu.h"
int main(void) {
  if (dfgpu_abi_version() != DFGPU_ABI_VERSION) return 10;
  dfgpu_table_t t = 0;
  if (dfgpu_tpch_orders(0.001, 0, -1, &t) == 0) return 11;            /* not initialised: must fail */
  if (!strstr(dfgpu_last_error(), "dfgpu_init")) return 12;
  /* host half of the scan on a truncated chunk: an error, not a crash */
  unsigned char junk[4] = {0x15, 0x00, 0x15, 0x02};
  dfgpu_parquet_column col;
  dfgpu_parquet_chunk_info info;
  memset(&col, 0, sizeof col);
  col.physical_type = DFGPU_PARQUET_INT64; col.codec = DFGPU_PARQUET_UNCOMPRESSED; col.num_values = 10;
  col.field.type = DFGPU_INT64; col.name = "k";
  if (dfgpu_parquet_inspect_chunk(junk, 4, &col, &info) == 0) return 13;
  if (!strstr(dfgpu_last_error(), "parquet")) return 14;
  printf("abi %d ok\n", dfgpu_abi_version());
  return 0;
}
""")
    exe = tmp_path / "bind"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L", lib_dir, "-ldfgpu", f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert out.stdout.strip() == f"abi {ABI} ok"


def test_case_and_in_list_lowering():
    """CaseExpr -> one DFGPU_EXPR_CASE node per WHEN (condition in `column`, THEN left, ELSE right, later WHENs nested in ELSE);
    InListExpr -> the OR of equalities, NOT of it when negated (include/dfgpu.h, expr.py)"""
    import pyarrow as pa
    from datafusion_amd.expr import OP_CASE, OP_NOT, case, col, lit, lower
    e = case([(col("a") > lit(0), lit(1)), (col("a") < lit(0), lit(-1))], lit(0))
    l = lower(e, ["a"])
    root = l.nodes[l.c.root]
    assert root.op == OP_CASE and l.nodes[root.column].op == 24 and l.nodes[root.left].lit_lo == 1     # WHEN a > 0 THEN 1
    inner = l.nodes[root.right]
    assert inner.op == OP_CASE and l.nodes[inner.column].op == 22 and l.nodes[inner.right].op == 2 and l.nodes[inner.right].lit_lo == 0
    no_else = lower(case([(col("a") > lit(0), col("a"))]), ["a"])
    assert no_else.nodes[no_else.c.root].right == -1
    inl = lower(col("a").in_list([lit(1), lit(2), lit(3)]), ["a"])
    ops = [inl.nodes[i].op for i in range(inl.c.n_nodes)]
    assert ops.count(20) == 3 and ops.count(31) == 2 and inl.nodes[inl.c.root].op == 31
    neg = lower(col("a").in_list([lit(1)], negated=True), ["a"])
    assert neg.nodes[neg.c.root].op == OP_NOT and neg.c.n_nodes == 4
    with pytest.raises(ValueError):
        col("a").in_list([])


class _StubTable:
    """what bind_string_literals needs of a DeviceTable: schema (index types of dictionary columns), index_of, dictionary_code"""

    def __init__(self, columns):
        import pyarrow as pa
        self.names = [n for n, _, _ in columns]
        self.schema = pa.schema([pa.field(n, t) for n, t, _ in columns])
        self.dicts = {n: d for n, _, d in columns}

    def index_of(self, c):
        return c if isinstance(c, int) else self.names.index(c)

    def dictionary_code(self, column, value):
        d = self.dicts[self.names[self.index_of(column)]]
        return d.index(value) if value in d else None


def test_string_literals_are_bound_to_dictionary_indices_on_the_host():
    """expr.bind_string_literals: `dict_col = 'x'` -> index comparison; an absent string -> comparison with -1 over the widened
    index (constant FALSE for `=`, TRUE for `!=` on non-NULL rows); a NULL string literal -> NULL index literal; binding reaches
    into AND / OR / NOT / CASE / IN lists"""
    import pyarrow as pa
    from datafusion_amd.expr import BinaryExpr, CaseExpr, CastExpr, Literal, NotExpr, bind_string_literals, case, col, lit
    t = _StubTable([("seg", pa.uint8(), ["AUTOMOBILE", "BUILDING", "FURNITURE"]), ("prio", pa.int32(), ["1-URGENT", "2-HIGH"]), ("k", pa.int64(), None)])
    s = lambda v: lit(v, pa.string())   # noqa: E731
    e = bind_string_literals(col("seg").eq(s("BUILDING")), t)
    assert isinstance(e, BinaryExpr) and e.op == "=" and e.left.index == 0 and e.right.value == 1 and e.right.type == pa.uint8()
    e = bind_string_literals(s("2-HIGH").ne(col("prio")), t)                      # literal on the left
    assert e.op == "!=" and e.left.index == 1 and e.right.value == 1 and e.right.type == pa.int32()
    e = bind_string_literals(col("seg").eq(s("NOT THERE")), t)
    assert isinstance(e.left, CastExpr) and e.left.cast_type == pa.int64() and e.right.value == -1
    e = bind_string_literals(col("prio").ne(s("NOT THERE")), t)
    assert isinstance(e.left, CastExpr) and e.op == "!=" and e.right.value == -1
    e = bind_string_literals(col("seg").eq(Literal(None, pa.string())), t)
    assert e.right.value is None and e.right.type == pa.uint8()
    e = bind_string_literals(col("seg").eq(s("BUILDING")).and_(col("k") > lit(5)).or_(col("prio").eq(s("1-URGENT")).not_()), t)
    assert e.op == "or" and e.left.left.right.value == 1 and isinstance(e.right, NotExpr) and e.right.arg.right.value == 0
    e = bind_string_literals(case([(col("prio").eq(s("1-URGENT")).or_(col("prio").eq(s("2-HIGH"))), lit(1))], lit(0)), t)
    assert isinstance(e, CaseExpr) and e.when_then[0][0].left.right.value == 0 and e.when_then[0][0].right.right.value == 1
    e = bind_string_literals(col("seg").in_list([s("FURNITURE"), s("AUTOMOBILE")], negated=True), t)
    assert isinstance(e, NotExpr) and e.arg.op == "or" and e.arg.left.right.value == 2 and e.arg.right.right.value == 0
    plain = bind_string_literals(col("k") > lit(5), t)           # nothing to bind: the same tree (rebuilt)
    assert plain.op == ">" and plain.left.name == "k" and plain.right.value == 5


def test_join_filter_string_literals_bind_through_the_source_columns():
    """expr.IntermediateSchema: a JoinFilter's intermediate column k is column (index, side) of the build / probe table; string
    literals in the filter are bound with THAT column's dictionary (Q19: p_brand = 'Brand#12' AND p_container IN (...) AND l_quantity ...)"""
    import pyarrow as pa
    from datafusion_amd.expr import IntermediateSchema, bind_string_literals, col, lit, lower
    build = _StubTable([("l_partkey", pa.int64(), None), ("l_quantity", pa.decimal128(15, 2), None)])
    probe = _StubTable([("p_partkey", pa.int64(), None), ("p_brand", pa.uint8(), ["Brand#12", "Brand#23"]), ("p_size", pa.int32(), None),
                        ("p_container", pa.uint8(), ["LG BOX", "SM BOX", "SM CASE"])])
    view = IntermediateSchema(build, probe, [(1, "Left"), (1, "Right"), (2, "Right"), (3, "Right")])
    assert view.schema.names == ["f0", "f1", "f2", "f3"] and view.schema.field("f3").type == pa.uint8() and view.schema.field("f0").type == pa.decimal128(15, 2)
    assert view.dictionary_code("f1", "Brand#23") == 1 and view.dictionary_code(3, "SM CASE") == 2 and view.dictionary_code("f3", "JUMBO JAR") is None
    s = lambda v: lit(v, pa.string())   # noqa: E731
    e = col("f1").eq(s("Brand#12")).and_(col("f3").in_list([s("SM CASE"), s("SM BOX")])).and_(col("f2") <= lit(5, pa.int32()))
    b = bind_string_literals(e, view)
    assert b.left.left.right.value == 0 and b.left.left.left.index == 1                     # f1 = index of 'Brand#12'
    assert b.left.right.op == "or" and b.left.right.left.right.value == 2 and b.left.right.right.right.value == 1 and b.left.right.left.left.index == 3
    lowered = lower(e, view.names, view)                                                   # lowers without a string node left
    assert all(lowered.nodes[i].op != 2 or lowered.nodes[i].field.type != 0 for i in range(lowered.c.n_nodes))


def test_rust_shim_sys_rs_matches_the_header():
    """shim/src/sys.rs (the Rust `extern "C"` block) is generated from include/dfgpu.h: a change of the C ABI that is not carried over fails here"""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "gen_shim_sys.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def _shim(name):
    return open(os.path.join(ROOT, "shim", "src", name)).read()


def test_rust_shim_binds_only_declared_entry_points_and_the_join_sequence_the_c_driver_runs():
    """No rustc in this image, so the shim is kept right by inspection: every `sys::dfgpu_*` it calls exists in the generated binding
    (= in the header), `hash_join.rs` performs the sequence tests/c/plan_driver.c executes on the GPU (builder push -> finish once ->
    probe / probe_with_filter per partition -> emit_unmatched once), and the round-2 review's defects stay fixed."""
    sys_rs = _shim("sys.rs")
    declared = set(re.findall(r"pub fn (dfgpu_\w+)\(", sys_rs))
    used = {}
    for f in ("lib.rs", "table.rs", "device.rs", "expr.rs", "hash_join.rs", "operators.rs", "rule.rs", "ffi.rs", "scan.rs"):
        used[f] = set(re.findall(r"sys::(dfgpu_[a-z0-9_]+)\(", _shim(f)))
        assert used[f] <= declared, (f, used[f] - declared)
    hj = _shim("hash_join.rs")
    assert {"dfgpu_join_builder_create", "dfgpu_join_builder_push", "dfgpu_join_builder_finish", "dfgpu_join_builder_free", "dfgpu_join_probe",
            "dfgpu_join_probe_with_filter", "dfgpu_join_emit_unmatched", "dfgpu_join_free", "dfgpu_column_minmax"} <= used["hash_join.rs"]
    driver = open(os.path.join(ROOT, "tests", "c", "plan_driver.c")).read()
    for fn in ("dfgpu_join_builder_push", "dfgpu_join_builder_finish", "dfgpu_join_probe_with_filter", "dfgpu_join_emit_unmatched", "dfgpu_table_export_batch",
               "dfgpu_table_export_device", "dfgpu_table_import_device"):
        assert fn + "(" in driver, fn
    # the defects of the round-2 review (VERDICT weak 2 / ADVICE high) by their fingerprints
    assert "null_aware: self.null_aware as i32" in hj and "null_aware: 0" not in hj
    assert "OnceCell" in hj and "remaining.fetch_sub" in hj                       # ONE shared build, the last partition emits
    assert "build_join_schema" in hj and "reorder" in hj                           # per-join-type column mapping, interleaved projections
    assert ".expect(\"the rule admits column keys only\")" not in hj             # expression keys decline instead of panicking
    ops_rs = _shim("operators.rs")
    assert "RepartitionState" in ops_rs and "get_or_try_init" in ops_rs            # every input partition executed once
    rule = _shim("rule.rs")
    assert "all_hash_repartitions_offloadable" in rule and "!needs_order" in rule  # co-partitioning stays consistent; order_insensitive from the ancestors
    for f in ("hash_join.rs", "operators.rs", "rule.rs", "expr.rs"):
        assert ".as_any()" not in _shim(f), f                                      # DataFusion 55: `dyn ExecutionPlan` / `dyn PhysicalExpr` downcast_ref
    ffi = _shim("ffi.rs")
    assert "FFI_PhysicalOptimizerRule::new(" in ffi and "FFI_QueryPlanner::new_with_ffi_codecs(" in ffi and "datafusion_gpu_amd_get_module" in ffi


def test_rust_shim_targets_the_reference_workspace_versions():
    """shim/Cargo.toml pins the datafusion / arrow versions of the reference workspace (skipped on the GPU box, where /root/reference does not exist)"""
    ref = "/root/reference/Cargo.toml"
    if not os.path.exists(ref):
        pytest.skip("no reference checkout here")
    text = open(ref).read()
    df = re.search(r'^version = "([0-9.]+)"', text[text.index("[workspace.package]"):], flags=re.M).group(1)
    arrow = re.search(r'^arrow = \{ version = "([0-9.]+)"', text, flags=re.M).group(1)
    cargo = open(os.path.join(ROOT, "shim", "Cargo.toml")).read()
    assert re.search(rf'^datafusion = (\{{ version = )?"{re.escape(df)}"', cargo, flags=re.M) and f'datafusion-ffi = "{df}"' in cargo, df
    assert f'arrow = {{ version = "{arrow}"' in cargo, arrow
