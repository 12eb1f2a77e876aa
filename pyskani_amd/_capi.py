"""ctypes binding of libpyskani_amd.so (the C-ABI declared in include/pyskani_amd.h).

The library is the product path: if it is missing this module raises ImportError — there is
no CPU fallback (the CPU oracle under oracle/ is test infrastructure and is never imported here).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PSK_LIB_PATH") or os.path.join(_HERE, "libpyskani_amd.so")   # override: A/B builds only

PSK_OK, PSK_EINVAL, PSK_ENOMEM, PSK_EHIP, PSK_ENOMODEL, PSK_EKEY, PSK_ELIMIT, PSK_ERCCL = range(8)
COMM_ID_BYTES = 128
ABI_VERSION = 5      # PSK_ABI_VERSION of include/pyskani_amd.h this binding was written against


class Params(C.Structure):
    _fields_ = [("c", C.c_int32), ("marker_c", C.c_int32), ("k", C.c_int32)]


class QueryOpts(C.Structure):
    _fields_ = [("learned_ani", C.c_int32), ("median", C.c_int32), ("robust", C.c_int32),
                ("faster_small", C.c_int32), ("cutoff", C.c_double), ("min_aligned_frac", C.c_double),
                ("model", C.c_void_p)]


class Hit(C.Structure):
    _fields_ = [("ani", C.c_float), ("af_query", C.c_float), ("af_ref", C.c_float),
                ("ref_index", C.c_uint32), ("n_chunks", C.c_uint32), ("n_intervals", C.c_uint32),
                ("n_anchors", C.c_uint64), ("covered_query", C.c_uint64), ("covered_ref", C.c_uint64),
                ("sum_chain_anchors", C.c_uint64), ("sum_chunk_seeds", C.c_uint64),
                ("ani_raw", C.c_float), ("ani_std", C.c_float), ("learned", C.c_uint32), ("reserved", C.c_uint32)]


class HitMin(C.Structure):
    """psk_hit_min: what the reference's Hit holds (hit.rs:77-104) in 20 bytes; `query` = index of the query within the call, bit 31 = learned"""
    _fields_ = [("ani", C.c_float), ("af_query", C.c_float), ("af_ref", C.c_float), ("ref_index", C.c_uint32), ("query", C.c_uint32)]


class TreeNode(C.Structure):
    _fields_ = [("feature", C.c_int32), ("threshold", C.c_float), ("left", C.c_int32), ("right", C.c_int32),
                ("value", C.c_float), ("missing", C.c_int32), ("is_leaf", C.c_int32), ("reserved", C.c_int32)]


FEATURE_NAMES = ["ani100", "std100", "q90_query", "q50_query", "q10_query", "q90_ref", "q50_ref", "q10_ref", "avg_chain_len",
                 "af_query", "af_ref", "n_chunks", "total_len_query", "total_len_ref", "n_contigs_query", "n_contigs_ref"]
DEFAULT_FEATURES = FEATURE_NAMES[:9]


class Seed(C.Structure):
    _fields_ = [("kmer", C.c_uint32), ("pos", C.c_uint32), ("contig", C.c_uint32), ("canon", C.c_uint32)]


# every symbol include/pyskani_amd.h declares
SYMBOLS = [
    "psk_last_error", "psk_version", "psk_abi_version", "psk_free", "psk_ctx_create", "psk_ctx_destroy",
    "psk_ctx_synchronize", "psk_pack2bit_host", "psk_ctx_small_query_stats", "psk_ctx_set_timing", "psk_ctx_timing", "psk_db_add_batch", "psk_device_alloc", "psk_device_free", "psk_memcpy_h2d",
    "psk_sketch_host", "psk_sketch_many_host", "psk_sketch_batch_device", "psk_sketch_free", "psk_sketch_free_many", "psk_sketch_info",
    "psk_sketch_export", "psk_sketch_contig_lens", "psk_sketch_import", "psk_db_create", "psk_db_destroy", "psk_db_add", "psk_db_size",
    "psk_db_name", "psk_db_sketch", "psk_screen", "psk_chain", "psk_query", "psk_query_host", "psk_query_many", "psk_query_many_min", "psk_gather_hits_min",
    "psk_sketch_pack_size", "psk_sketch_pack", "psk_sketch_unpack", "psk_sketch_pack_many", "psk_ctx_clock_probe", "psk_ctx_work", "psk_ctx_join_work",
    "psk_comm_unique_id", "psk_comm_create", "psk_comm_destroy", "psk_comm_info", "psk_gather_hits", "psk_gather_sketches",
    "psk_model_create", "psk_model_load_json", "psk_model_load_file", "psk_model_free", "psk_model_info", "psk_model_predict",
]

_lib = None


def load():
    """Load the shared library (no GPU is touched until a context is created)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C pyskani_amd/csrc`. pyskani_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    vp, u32, u64 = C.c_void_p, C.c_uint32, C.c_uint64
    lib.psk_abi_version.restype = C.c_int
    if lib.psk_abi_version() != ABI_VERSION:      # (an argument list that moved would be a memory error, not a link error)
        raise ImportError(f"{LIB_PATH} speaks C-ABI revision {lib.psk_abi_version()}, this binding {ABI_VERSION}: rebuild with `make -C pyskani_amd/csrc`")
    lib.psk_last_error.restype = C.c_char_p
    lib.psk_version.restype = C.c_char_p
    lib.psk_free.argtypes = [vp]
    lib.psk_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.psk_ctx_destroy.argtypes = [vp]
    lib.psk_ctx_destroy.restype = None
    lib.psk_ctx_synchronize.argtypes = [vp]
    lib.psk_ctx_small_query_stats.argtypes = [vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]
    lib.psk_pack2bit_host.argtypes = [vp, u64, vp, C.c_int]
    lib.psk_pack2bit_host.restype = None
    lib.psk_ctx_set_timing.argtypes = [vp, C.c_int]
    lib.psk_ctx_timing.argtypes = [vp, C.c_char_p, C.POINTER(C.c_double), C.POINTER(u64)]
    lib.psk_db_add_batch.argtypes = [vp, C.POINTER(C.c_char_p), C.POINTER(vp), u32]
    lib.psk_device_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    lib.psk_device_free.argtypes = [vp, vp]
    lib.psk_memcpy_h2d.argtypes = [vp, vp, vp, C.c_size_t]
    lib.psk_sketch_host.argtypes = [vp, C.POINTER(Params), C.POINTER(C.c_char_p), C.POINTER(u64), u32, C.c_int, C.POINTER(vp)]
    lib.psk_sketch_many_host.argtypes = [vp, C.POINTER(Params), C.POINTER(C.c_char_p), C.POINTER(u64), C.POINTER(u32), u32, C.c_int, C.POINTER(vp)]
    lib.psk_sketch_batch_device.argtypes = [vp, C.POINTER(Params), vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(u32), u32, C.c_int, C.POINTER(vp)]
    lib.psk_sketch_free.argtypes = [vp]
    lib.psk_sketch_free.restype = None
    lib.psk_sketch_free_many.argtypes = [C.POINTER(vp), u32]
    lib.psk_sketch_free_many.restype = None
    lib.psk_sketch_info.argtypes = [vp, C.POINTER(Params), C.POINTER(u64), C.POINTER(u64), C.POINTER(u64), C.POINTER(u32)]
    lib.psk_sketch_export.argtypes = [vp, vp, vp]
    lib.psk_sketch_contig_lens.argtypes = [vp, vp]
    lib.psk_sketch_import.argtypes = [vp, C.POINTER(Params), vp, u32, vp, u64, vp, u64, C.c_int, C.POINTER(vp)]
    lib.psk_db_create.argtypes = [vp, C.POINTER(Params), C.POINTER(vp)]
    lib.psk_db_destroy.argtypes = [vp]
    lib.psk_db_destroy.restype = None
    lib.psk_db_add.argtypes = [vp, C.c_char_p, vp]
    lib.psk_db_size.argtypes = [vp]
    lib.psk_db_size.restype = u32
    lib.psk_db_name.argtypes = [vp, u32]
    lib.psk_db_name.restype = C.c_char_p
    lib.psk_db_sketch.argtypes = [vp, u32]
    lib.psk_db_sketch.restype = vp
    lib.psk_screen.argtypes = [vp, vp, C.c_double, C.c_int, vp, vp]
    lib.psk_chain.argtypes = [vp, C.POINTER(vp), u32, vp, C.POINTER(QueryOpts), C.POINTER(Hit)]
    lib.psk_query.argtypes = [vp, vp, C.POINTER(QueryOpts), C.POINTER(C.POINTER(Hit)), C.POINTER(u64)]
    lib.psk_query_host.argtypes = [vp, C.POINTER(C.c_char_p), C.POINTER(u64), u32, C.c_int, C.POINTER(QueryOpts), C.POINTER(C.POINTER(Hit)), C.POINTER(u64)]
    lib.psk_query_many.argtypes = [vp, C.POINTER(vp), u32, C.POINTER(QueryOpts), C.POINTER(C.POINTER(Hit)), C.POINTER(u64)]
    lib.psk_query_many_min.argtypes = [vp, C.POINTER(vp), u32, C.POINTER(QueryOpts), C.POINTER(C.POINTER(HitMin)), C.POINTER(u64)]
    lib.psk_gather_hits_min.argtypes = [vp, vp, u64, C.POINTER(C.POINTER(HitMin)), C.POINTER(u64), C.POINTER(u64)]
    lib.psk_sketch_pack_size.argtypes = [vp, C.POINTER(u64)]
    lib.psk_sketch_pack.argtypes = [vp, vp, u64]
    lib.psk_sketch_unpack.argtypes = [vp, vp, u64, C.POINTER(u64), u32, C.POINTER(vp)]
    lib.psk_sketch_pack_many.argtypes = [C.POINTER(vp), u32, vp, C.POINTER(u64), u64]
    lib.psk_ctx_clock_probe.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.psk_ctx_work.argtypes = [vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64), C.c_int]
    lib.psk_ctx_join_work.argtypes = [vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64), C.POINTER(u64), C.c_int]
    lib.psk_comm_unique_id.argtypes = [vp]
    lib.psk_comm_create.argtypes = [vp, C.c_int, C.c_int, vp, C.POINTER(vp)]
    lib.psk_comm_destroy.argtypes = [vp]
    lib.psk_comm_destroy.restype = None
    lib.psk_comm_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(u64), C.POINTER(u64)]
    lib.psk_gather_hits.argtypes = [vp, vp, u64, C.POINTER(C.POINTER(Hit)), C.POINTER(u64), C.POINTER(u64)]
    lib.psk_gather_sketches.argtypes = [vp, C.POINTER(vp), u32, C.POINTER(C.POINTER(vp)), C.POINTER(u32)]
    lib.psk_model_create.argtypes = [vp, C.POINTER(TreeNode), u64, C.POINTER(u32), u32, C.c_float, C.c_float, C.POINTER(C.c_int32), u32, C.POINTER(vp)]
    lib.psk_model_load_json.argtypes = [vp, C.c_char_p, C.c_size_t, C.POINTER(vp)]
    lib.psk_model_load_file.argtypes = [vp, C.c_char_p, C.POINTER(vp)]
    lib.psk_model_free.argtypes = [vp]
    lib.psk_model_free.restype = None
    lib.psk_model_info.argtypes = [vp, C.POINTER(u32), C.POINTER(u64), C.POINTER(u32)]
    lib.psk_model_predict.argtypes = [vp, vp, u32, vp]
    _lib = lib
    return lib


_EXC = {PSK_EINVAL: ValueError, PSK_ENOMEM: MemoryError, PSK_EHIP: RuntimeError,
        PSK_ENOMODEL: RuntimeError, PSK_EKEY: KeyError, PSK_ELIMIT: OverflowError, PSK_ERCCL: RuntimeError}


def check(status):
    """Map a psk_status to the Python exception type the reference raises for that failure."""
    if status != PSK_OK:
        msg = load().psk_last_error().decode("utf-8", "replace")
        raise _EXC.get(status, RuntimeError)(msg)


def hit_records(hits_p, lo, hi, dtype):
    """psk_hit records [lo, hi) behind a `POINTER(Hit)` as a numpy array that owns its memory: one memmove into a fresh array (a ctypes
    array type per length, which `(Hit * n).from_address(...)` creates, costs more than the copy: 4.6 ms against 0.7 for 10^5 hits)."""
    import numpy as np
    n = max(0, hi - lo)
    out = np.empty(n, dtype)
    if n:
        C.memmove(out.ctypes.data, C.cast(hits_p, C.c_void_p).value + lo * dtype.itemsize, n * dtype.itemsize)
    return out
