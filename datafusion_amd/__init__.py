"""datafusion_amd — MI355X-native (gfx950) execution backend for DataFusion's vectorized
physical operators: hash-join build/probe, hash aggregation, filter/projection expression
evaluation, sort/TopK and the hash-repartition exchange, as hand-written HIP kernels behind
the C ABI of include/dfgpu.h (libdfgpu.so).  This package is the host-side mirror of the
reference's operator interface; it holds no CPU implementation of any operator."""
from . import _lib  # noqa: F401
from ._lib import DfgpuError  # noqa: F401

__all__ = ["_lib", "DfgpuError"]
