"""ctypes binding of libdfgpu.so (include/dfgpu.h).

The library is the product: there is NO CPU fallback.  If the shared object is missing the
import fails loudly; if no MI355X is visible `dfgpu_init` fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdfgpu.so")


class DfgpuError(RuntimeError):
    """DataFusionError::External equivalent raised for any non-zero return code."""


class Field(C.Structure):
    _fields_ = [("type", C.c_int32), ("precision", C.c_int32), ("scale", C.c_int32), ("nullable", C.c_int32)]


class ColumnView(C.Structure):
    _fields_ = [("field", Field), ("length", C.c_int64), ("null_count", C.c_int64), ("data", C.c_void_p),
                ("validity", C.c_void_p), ("name", C.c_char_p), ("offsets", C.c_void_p)]


class ExprNode(C.Structure):
    _fields_ = [("op", C.c_int32), ("column", C.c_int32), ("left", C.c_int32), ("right", C.c_int32),
                ("field", Field), ("is_null", C.c_int32), ("_pad", C.c_int32), ("lit_lo", C.c_uint64),
                ("lit_hi", C.c_uint64)]


class Expr(C.Structure):
    _fields_ = [("nodes", C.POINTER(ExprNode)), ("n_nodes", C.c_int32), ("root", C.c_int32), ("string_pool", C.c_char_p)]


class Metrics(C.Structure):
    """dfgpu_metrics"""
    _fields_ = [(n, C.c_int64) for n in ("calls", "elapsed_ns", "kernel_ns", "h2d_bytes", "d2h_bytes", "hbm_bytes_algorithmic", "rows_in", "rows_out")]


class JoinFilter(C.Structure):
    """dfgpu_join_filter"""
    _fields_ = [("expression", Expr), ("column_index", C.POINTER(C.c_int32)), ("column_side", C.POINTER(C.c_int32)), ("n_columns", C.c_int32)]


class JoinOptions(C.Structure):
    _fields_ = [("perfect_hash_join_small_build_threshold", C.c_int64),
                ("perfect_hash_join_min_key_density", C.c_double), ("table_mode", C.c_int32),
                ("force_hash_collisions", C.c_int32), ("probe_mode", C.c_int32), ("null_aware", C.c_int32)]


class JoinInfo(C.Structure):
    _fields_ = [("build_rows", C.c_int64), ("table_bytes", C.c_int64), ("used_array_map", C.c_int32),
                ("build_keys_unique", C.c_int32), ("probe_rows", C.c_int64), ("output_rows", C.c_int64),
                ("table_kind", C.c_int32), ("build_keys_ascending", C.c_int32)]


class AggSpec(C.Structure):
    _fields_ = [("func", C.c_int32), ("has_arg", C.c_int32), ("arg", Expr), ("name", C.c_char_p), ("return_field", Field)]


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("calls", C.c_int64), ("total_ms", C.c_double),
                ("algorithmic_bytes", C.c_int64)]


class CacheStats(C.Structure):
    """dfgpu_cache_stats"""
    _fields_ = [(n, C.c_int64) for n in ("entries", "bytes", "budget_bytes", "hits", "misses", "insertions", "evictions")]


class ExchangeStats(C.Structure):
    _fields_ = [("bytes_sent_to_peers", C.c_int64), ("bytes_received_from_peers", C.c_int64), ("rows_sent_to_peers", C.c_int64),
                ("rows_received_from_peers", C.c_int64), ("messages", C.c_int64), ("collectives", C.c_int64)]


ALLTOALLV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_void_p), C.POINTER(C.c_int64))
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)


class HostTransport(C.Structure):
    """dfgpu_host_transport"""
    _fields_ = [("ctx", C.c_void_p), ("alltoallv", ALLTOALLV_FN), ("allgather", ALLGATHER_FN)]


class ParquetColumn(C.Structure):
    _fields_ = [("physical_type", C.c_int32), ("type_length", C.c_int32), ("codec", C.c_int32), ("max_definition_level", C.c_int32),
                ("max_repetition_level", C.c_int32), ("_pad", C.c_int32), ("num_values", C.c_int64), ("field", Field), ("name", C.c_char_p)]


class ParquetChunk(C.Structure):
    _fields_ = [("bytes", C.c_void_p), ("n_bytes", C.c_int64), ("column", ParquetColumn), ("cache_key", C.c_char_p), ("cache_key_bytes", C.c_int64)]


class ParquetChunkInfo(C.Structure):
    _fields_ = [("n_pages", C.c_int32), ("n_dictionary_pages", C.c_int32), ("n_data_pages_v1", C.c_int32), ("n_data_pages_v2", C.c_int32),
                ("n_plain_pages", C.c_int32), ("n_dictionary_encoded_pages", C.c_int32), ("n_runs_rle", C.c_int64), ("n_runs_bitpacked", C.c_int64),
                ("values", C.c_int64), ("nulls", C.c_int64), ("uncompressed_bytes", C.c_int64), ("compressed_bytes", C.c_int64),
                ("dictionary_values", C.c_int64)]


# every symbol include/dfgpu.h declares (tests/test_abi.py checks the library exports them all)
SYMBOLS = [
    "dfgpu_abi_version", "dfgpu_set_option", "dfgpu_init", "dfgpu_set_device", "dfgpu_get_device", "dfgpu_shutdown", "dfgpu_device_count", "dfgpu_last_error", "dfgpu_sync",
    "dfgpu_stream", "dfgpu_mem_stats", "dfgpu_mem_trim", "dfgpu_table_import", "dfgpu_table_export",
    "dfgpu_table_alloc", "dfgpu_table_dictionary_lookup", "dfgpu_table_free", "dfgpu_table_num_rows", "dfgpu_table_num_columns", "dfgpu_table_column",
    "dfgpu_table_select", "dfgpu_table_hstack", "dfgpu_table_concat", "dfgpu_table_slice", "dfgpu_expr_type",
    "dfgpu_filter", "dfgpu_project", "dfgpu_join_build", "dfgpu_join_probe", "dfgpu_join_probe_filtered", "dfgpu_join_probe_with_filter", "dfgpu_column_minmax", "dfgpu_join_emit_unmatched",
    "dfgpu_join_get_info", "dfgpu_join_free", "dfgpu_agg_create", "dfgpu_agg_update", "dfgpu_agg_update_filtered", "dfgpu_agg_fused_updates", "dfgpu_set_fusion", "dfgpu_jit_stats", "dfgpu_agg_emit",
    "dfgpu_agg_free", "dfgpu_sort", "dfgpu_partition", "dfgpu_hash_columns", "dfgpu_tpch_orders",
    "dfgpu_tpch_lineitem", "dfgpu_tpch_customer", "dfgpu_profile_enable", "dfgpu_profile_reset",
    "dfgpu_profile_count", "dfgpu_profile_get", "dfgpu_profile_launches", "dfgpu_join_probe_bounded", "dfgpu_parquet_decode_chunk", "dfgpu_parquet_inspect_chunk", "dfgpu_parquet_read_chunks",
    "dfgpu_comm_unique_id", "dfgpu_comm_init_rank", "dfgpu_comm_init_all", "dfgpu_comm_init_host", "dfgpu_comm_free", "dfgpu_comm_info", "dfgpu_comm_transport_info",
    "dfgpu_exchange_hash", "dfgpu_exchange_hash_stream_open", "dfgpu_exchange_hash_stream_next", "dfgpu_exchange_hash_stream_free", "dfgpu_exchange_broadcast", "dfgpu_exchange_broadcast_pruned", "dfgpu_comm_stats",
    "dfgpu_mem_set_limit", "dfgpu_mem_limit", "dfgpu_mem_try_reserve", "dfgpu_mem_reservation_size", "dfgpu_mem_release",
    "dfgpu_table_export_batch", "dfgpu_host_register", "dfgpu_host_unregister", "dfgpu_table_export_into",
    "dfgpu_column_inlist", "dfgpu_metrics_reset", "dfgpu_metrics_get", "dfgpu_table_dictionary_like", "dfgpu_table_dictionary_encode", "dfgpu_table_dictionary_decode", "dfgpu_table_dictionary_size", "dfgpu_join_builder_create", "dfgpu_join_builder_push", "dfgpu_join_builder_finish", "dfgpu_join_builder_free", "dfgpu_join_estimate_bytes",
    "dfgpu_table_retain", "dfgpu_table_export_device", "dfgpu_table_import_device",
    "dfgpu_cache_create", "dfgpu_cache_free", "dfgpu_cache_get", "dfgpu_cache_put", "dfgpu_cache_clear", "dfgpu_cache_get_stats", "dfgpu_jit_cache_stats", "dfgpu_join_contains", "dfgpu_exchange_join_visited", "dfgpu_exchange_range",
    "dfgpu_agg_create_grouping_sets", "dfgpu_ipc_open", "dfgpu_ipc_close", "dfgpu_ipc_info", "dfgpu_ipc_column", "dfgpu_ipc_batch_rows", "dfgpu_ipc_read_batch",
]

_lib = None
_initialised_device = None


def load() -> C.CDLL:
    """dlopen libdfgpu.so; raises if it has not been built (python __graft_entry__.py build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `make -C datafusion_amd/csrc` "
                "(or __graft_entry__.build()). There is no CPU fallback.")
        _lib = C.CDLL(LIB_PATH)
        _lib.dfgpu_last_error.restype = C.c_char_p
        _lib.dfgpu_stream.restype = C.c_void_p
        for name in SYMBOLS:
            getattr(_lib, name)  # AttributeError here = header/library mismatch
    return _lib


def check(rc: int):
    if rc != 0:
        raise DfgpuError(load().dfgpu_last_error().decode("utf-8", "replace"))


def init(device: int | None = None) -> C.CDLL:
    """bind this process to one GPU (LOCAL_RANK under torchrun, else 0)"""
    global _initialised_device
    lib = load()
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    if _initialised_device is None:
        check(lib.dfgpu_init((C.c_int * 1)(device), 1))
        _initialised_device = device
    elif _initialised_device != device:
        raise DfgpuError(f"already bound to device {_initialised_device}")
    return lib
