"""CPU-side checks of the drop-in boundary: libdfgpu.so loads without a GPU, exports every
symbol include/dfgpu.h declares, and refuses to run (loudly) when no MI355X is present."""
import ctypes as C
import os
import re

import pytest

from datafusion_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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
    assert _lib.load().dfgpu_abi_version() == 6


def test_struct_layouts_match_header(tmp_path):
    """sizes the C compiler gives the ABI structs of include/dfgpu.h (LP64) == the ctypes mirrors"""
    import subprocess
    names = {"dfgpu_field": _lib.Field, "dfgpu_expr_node": _lib.ExprNode, "dfgpu_expr": _lib.Expr, "dfgpu_join_options": _lib.JoinOptions,
             "dfgpu_join_info": _lib.JoinInfo, "dfgpu_kernel_stat": _lib.KernelStat, "dfgpu_agg_spec": _lib.AggSpec,
             "dfgpu_column_view": _lib.ColumnView}
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "dfgpu.h"\nint main(void){' +
                   "".join(f'printf("{n} %zu\\n", sizeof({n}));' for n in names) +
                   'printf("ArrowSchema %zu\\nArrowArray %zu\\n", sizeof(struct ArrowSchema), sizeof(struct ArrowArray));return 0;}')
    exe = tmp_path / "sizes"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
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
    assert lib.dfgpu_init(0) != 0
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
