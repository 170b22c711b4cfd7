"""ctypes loader for libisle_hip.so (the C ABI declared in include/isle_hip.h)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

# name -> (restype, argtypes); every symbol include/isle_hip.h declares.
_P, _I, _U64, _F = C.c_void_p, C.c_int, C.c_uint64, C.c_float
SYMBOLS = {
    "isle_hip_create": (_P, [_I]),
    "isle_hip_destroy": (None, [_P]),
    "isle_hip_last_error": (C.c_char_p, [_P]),
    "isle_hip_switch_info": (_I, [_I, _P, _P, _P]),
    "isle_hip_comm_unique_id": (_I, [_P]),
    "isle_hip_comm_init": (_I, [_P, _I, _I, _P]),
    "isle_hip_comm_init_host": (_I, [_P, _I, _I, _P, _P]),
    "isle_hip_plan_shards": (_I, [_U64, _P, _I, _P]),
    "isle_hip_upload_csc_u64": (_I, [_P, _U64, _U64, _U64, _P, _P, _P, _U64, _U64]),
    "isle_hip_upload_csc_u32": (_I, [_P, _U64, _U64, _U64, _P, _P, _P, _U64, _U64]),
    "isle_hip_upload_counts_u32": (_I, [_P, _U64, _U64, _U64, _P, _P, _P, _U64, _U64]),
    "isle_hip_ingest_tdf": (_I, [_P, _P, _U64, _U64, _U64, _U64, _P, _P]),
    "isle_hip_get_A": (_I, [_P, _P, _P, _P]),
    "isle_hip_threshold": (_I, [_P, _U64, C.c_double, _U64, _P, _P, _P, _P]),
    "isle_hip_get_B": (_I, [_P, _P, _P, _P, _P, _P]),
    "isle_hip_shape": (_I, [_P, _P, _P, _P, _P, _P]),
    "isle_hip_catchwords": (_I, [_P, _I, _P, _U64, C.c_double, _P, _P, _P]),
    "isle_hip_topic_model": (_I, [_P, _I, _U64, _P, _P, _P, _P, _P]),
    "isle_hip_get_doc_topic_sums": (_I, [_P, _P, _P, _P]),
    "isle_hip_edge_topics": (_I, [_P, _P, _I, _F, _P]),
    "isle_hip_frobenius": (_I, [_P, _P]),
    "isle_hip_gram_apply": (_I, [_P, _P, _I, _P]),
    "isle_hip_operator_form": (_I, [_P, _P]),
    "isle_hip_infer": (_I, [_P, C.c_uint64, _I, _P, C.c_uint64, C.c_uint64, _P, _P, _P, _I, C.c_float, C.c_float, _P, _P, _P, _P, _P]),
    "isle_hip_block_ks": (_I, [_P, _I, _I, _I, _I, _F, _U64, _P, _P, _P, _P]),
    "isle_hip_block_ks_dense": (_I, [_P, _P, _U64, _I, _I, _I, _I, _F, _U64, _P, _P, _P, _P, _P, _P, _P]),
    "isle_hip_get_U": (_I, [_P, _P]),
    "isle_hip_set_U": (_I, [_P, _P, _I]),
    "isle_hip_eig_sym": (_I, [_P, _P, _I, _P, _P]),
    "isle_hip_kmeanspp_projected": (_I, [_P, _I, _P, _U64, _P, _P, _P, _P]),
    "isle_hip_get_min_dist": (_I, [_P, _P]),
    "isle_hip_host_rand": (_I, [_U64, _I, _P]),
    "isle_hip_lloyds_projected": (_I, [_P, _I, _P, _I, _P, _P]),
    "isle_hip_lift_centers": (_I, [_P, _P, _I, _I, _P]),
    "isle_hip_lloyds_sparse": (_I, [_P, _I, _P, _P, _P, _I, _P]),
    "isle_hip_timing_enable": (_I, [_P, _I]),
    "isle_hip_timing_reset": (_I, [_P]),
    "isle_hip_timing_get": (_I, [_P, _P, _P]),
    "isle_hip_synchronize": (_I, [_P]),
}


class IsleHipError(RuntimeError):
    pass


def library_path():
    # ISLE_HIP_LIB: developer override for A/B runs of differently compiled builds of the same sources (tools/*_probe.py)
    return os.environ.get("ISLE_HIP_LIB") or os.path.join(_HERE, "libisle_hip.so")


def load_library():
    """Loads libisle_hip.so and binds every declared symbol.  Raises if the library is missing:
    the hot path has no other implementation to fall back to."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise IsleHipError(
            "%s not found — build it with `make -C isle_amd/csrc` (or __graft_entry__.build()). "
            "There is no CPU fallback for the ISLE hot path." % path)
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib
