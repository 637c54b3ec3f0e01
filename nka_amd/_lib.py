"""Loader for libnka_hip.so (built in tree by nka_amd/csrc/Makefile)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_i32p = C.POINTER(C.c_int32)
_dp = C.POINTER(C.c_double)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p)
HOST_DOT_FN = C.CFUNCTYPE(C.c_double, C.c_void_p, C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_double))
HOST_ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int32)

# name -> (restype, argtypes): every symbol include/nka_hip.h (core), nka_hip_ext.h and nka_hip_vec.h declare
SIGNATURES = {
    "nka_hip_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int64, C.c_int32, C.c_double, C.c_int32, C.c_int32, C.c_void_p]),
    "nka_hip_destroy": (C.c_int, [C.c_void_p]),
    "nka_hip_clone": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "nka_hip_capture_safe": (C.c_int, [C.c_void_p]),
    "nka_hip_list_bound": (C.c_int, [C.c_void_p]),
    "nka_hip_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "nka_hip_accel_update": (C.c_int, [C.c_void_p, C.c_void_p]),
    "nka_hip_accel_update_host": (C.c_int, [C.c_void_p, C.c_void_p]),
    "nka_hip_accel_update_swap": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "nka_hip_restart": (C.c_int, [C.c_void_p]),
    "nka_hip_relax": (C.c_int, [C.c_void_p]),
    "nka_hip_set_vec_tol": (C.c_int, [C.c_void_p, C.c_double]),
    "nka_hip_num_vec": (C.c_int, [C.c_void_p]),
    "nka_hip_max_vec": (C.c_int, [C.c_void_p]),
    "nka_hip_vec_len": (C.c_int64, [C.c_void_p]),
    "nka_hip_vec_tol": (C.c_double, [C.c_void_p]),
    "nka_hip_defined": (C.c_int, [C.c_void_p]),
    "nka_hip_flavor": (C.c_int, [C.c_void_p]),
    "nka_hip_get_state": (C.c_int, [C.c_void_p, _i32p, _i32p, _i32p, _i32p, _i32p, _i32p, _i32p, _dp, _dp]),
    "nka_hip_get_reductions": (C.c_int, [C.c_void_p, _dp]),
    "nka_hip_get_w": (C.c_int, [C.c_void_p, C.c_int32, _dp]),
    "nka_hip_get_v": (C.c_int, [C.c_void_p, C.c_int32, _dp]),
    "nka_hip_set_allreduce": (C.c_int, [C.c_void_p, ALLREDUCE_FN, C.c_void_p]),
    "nka_hip_comm_unique_id": (C.c_int, [C.c_void_p]),
    "nka_hip_comm_init_rank": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]),
    "nka_hip_comm_destroy": (C.c_int, [C.c_void_p]),
    "nka_hip_comm_library": (C.c_int, [C.c_char_p, C.c_int32]),
    "nka_hip_comm_info": (C.c_int, [C.c_void_p, _i32p, _i32p]),
    "nka_hip_allreduce_now": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "nka_hip_p2p_export": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "nka_hip_p2p_attach": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]),
    "nka_hip_p2p_detach": (C.c_int, [C.c_void_p]),
    "nka_hip_p2p_mailbox": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "nka_hip_p2p_attach_local": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int32, C.c_int32]),
    "nka_hip_state_digest": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "nka_hip_set_host_dot": (C.c_int, [C.c_void_p, HOST_DOT_FN, C.c_void_p]),
    "nka_hip_set_timing": (C.c_int, [C.c_void_p, C.c_int32]),
    "nka_hip_get_timing": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_float)]),
    "nka_hip_set_timing_stride": (C.c_int, [C.c_void_p, C.c_int32]),
    "nka_hip_set_sum_order": (C.c_int, [C.c_void_p, C.c_int32]),
    "nka_hip_set_shard": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "nka_hip_last_error": (C.c_char_p, []),
    "nka_hip_invalidate_pointer_cache": (None, []),
    "nka_hip_device_info": (C.c_int, [C.c_void_p, C.c_char_p, _i32p]),
    "nka_hip_vec_workspace_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32, C.c_void_p]),
    "nka_hip_vec_workspace_destroy": (C.c_int, [C.c_void_p]),
    "nka_hip_vec_alloc": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_void_p)]),
    "nka_hip_vec_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "nka_hip_vec_copy": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "nka_hip_vec_setval": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_double]),
    "nka_hip_vec_scale": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_double]),
    "nka_hip_vec_update1": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_double, C.c_void_p]),
    "nka_hip_vec_update2": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_double, C.c_void_p, C.c_double]),
    "nka_hip_vec_update3": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_double, C.c_void_p, C.c_double, C.c_void_p]),
    "nka_hip_vec_update4": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_double, C.c_void_p, C.c_double, C.c_void_p, C.c_double]),
    "nka_hip_vec_dot": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, _dp]),
    "nka_hip_vec_norm2": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, _dp]),
    "nka_hip_vec_dot_many": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(C.c_void_p), C.c_int32, _dp]),
    "nka_hip_vec_dot_pair_many": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p),
                                            C.c_int32, _dp, _dp, _dp]),
    "nka_hip_vec_update_many": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, _dp, C.POINTER(C.c_void_p), _dp,
                                          C.POINTER(C.c_void_p), C.c_int32]),
    "nka_hip_vec_axpy_many": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, _dp, C.POINTER(C.c_void_p), C.c_int32]),
    "nka_hip_vec_update_norm2": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_double, C.c_void_p, C.c_int32, _dp]),
    "nka_hip_vec_scale_dot_pair_many": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_double, C.c_int32,
                                                  C.c_int32, C.c_double, C.c_void_p, C.POINTER(C.c_void_p), C.c_int32,
                                                  _dp, _dp, _dp]),
    "nka_hip_vec_update_many_keep": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, _dp, C.POINTER(C.c_void_p), _dp,
                                               C.POINTER(C.c_void_p), C.c_int32, C.c_void_p, C.c_void_p]),
    "nka_hip_vec_axpy_many_keep": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, _dp, C.POINTER(C.c_void_p), C.c_int32,
                                             C.c_void_p, C.c_void_p]),
    "nka_hip_vec_dot_pair_many_scaled": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_double, C.c_int32, C.c_double,
                                                   C.c_void_p, C.POINTER(C.c_void_p), C.c_int32, _dp, _dp, _dp]),
    "nka_hip_vec_diff_norm_dot_pair_many": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_double, C.c_void_p,
                                                      C.POINTER(C.c_void_p), C.c_int32, _dp, _dp, _dp, _dp]),
    "nka_hip_vec_update_many_keep_pend": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, _dp, C.POINTER(C.c_void_p), _dp,
                                                    C.POINTER(C.c_void_p), C.c_int32, C.c_void_p, C.c_void_p,
                                                    C.c_double, C.c_int32, C.c_double, C.c_int32]),
    "nka_hip_vec_axpy_many_keep_pend": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, _dp, C.POINTER(C.c_void_p), C.c_int32,
                                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int32, C.c_double]),
    "nka_hip_vec_set_sum_order": (C.c_int, [C.c_void_p, C.c_int32]),
    "nka_hip_vec_get_sum_order": (C.c_int, [C.c_void_p]),
    "nka_hip_vec_set_allreduce": (C.c_int, [C.c_void_p, ALLREDUCE_FN, C.c_void_p]),
    "nka_hip_vec_set_host_allreduce": (C.c_int, [C.c_void_p, HOST_ALLREDUCE_FN, C.c_void_p]),
    "nka_hip_vec_comm_init_rank": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]),
    "nka_hip_vec_comm_destroy": (C.c_int, [C.c_void_p]),
    "nka_hip_vec_allreduce_now": (C.c_int, [C.c_void_p, _dp, C.c_int32]),
    "nka_hip_vec_h2d": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "nka_hip_vec_d2h": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
}


# include/nka_example_dev.h (device-resident example system)
SIGNATURES.update({
    "nka_ex_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_double, C.c_int32, C.c_void_p]),
    "nka_ex_destroy": (C.c_int, [C.c_void_p]),
    "nka_ex_residual": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "nka_ex_pc_ssor": (C.c_int, [C.c_void_p, C.c_int32, C.c_double, C.c_void_p]),
    "nka_ex_update_solution": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "nka_ex_residual_grid": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "nka_ex_pc_ssor_grid": (C.c_int, [C.c_void_p, C.c_int32, C.c_double, C.c_void_p]),
})


# include/nka_hip_diag.h: only in the diagnostic build libnka_hip_diag.so (-DNKA_DIAGNOSTIC)
DIAG_SIGNATURES = {
    "nka_hip_set_tuning": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int32]),
    "nka_hip_set_grid": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "nka_hip_debug_time_pa": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_float)]),
    "nka_hip_debug_chain_sum": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_int32,
                                          C.POINTER(C.c_double), C.POINTER(C.c_float)]),
    "nka_hip_get_stamps": (C.c_int, [C.c_void_p, _dp]),
    "nka_hip_vec_set_tuning": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int32]),
}
_DIAG = None


def diag_lib_path() -> str:
    return os.environ.get("NKA_HIP_DIAG_LIB") or os.path.join(HERE, "libnka_hip_diag.so")


def load_diag() -> C.CDLL:
    """The diagnostic build (product + the A/B switches of include/nka_hip_diag.h): tests that hold every kernel
    variant to the same bits and the measurement tools use it; a second, independent copy of the library in the
    process (its own handles: never mix them with the product's)."""
    global _DIAG
    if _DIAG is None:
        path = diag_lib_path()
        if not os.path.exists(path):
            raise RuntimeError(f"{path} not found: `make -C nka_amd/csrc diag` (or __graft_entry__.build())")
        L = C.CDLL(path)
        for name, (res, args) in list(SIGNATURES.items()) + list(DIAG_SIGNATURES.items()):
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _DIAG = L
    return _DIAG


_EXTRA = {}


def load_diag_at(path: str) -> C.CDLL:
    """Another build of the diagnostic ABI (e.g. libnka_hip_diag_ft.so, `make -C nka_amd/csrc ftemporal`) as one more
    independent copy of the library in the process: tools/ab_libs.py alternates two builds on the same inputs."""
    path = os.path.abspath(path)
    if path not in _EXTRA:
        if not os.path.exists(path):
            raise RuntimeError(f"{path} not found")
        L = C.CDLL(path)
        for name, (res, args) in list(SIGNATURES.items()) + list(DIAG_SIGNATURES.items()):
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _EXTRA[path] = L
    return _EXTRA[path]


def lib_path() -> str:
    # NKA_HIP_LIB: load another build of the same ABI (kernel tuning experiments)
    return os.environ.get("NKA_HIP_LIB") or os.path.join(HERE, "libnka_hip.so")


def build(force: bool = False) -> str:
    """Compile the HIP library for gfx950 in tree (hipcc cross-compiles without a GPU)."""
    args = ["make", "-s", "-C", os.path.join(HERE, "csrc"), "-j6"]
    if force:
        subprocess.run(args + ["clean"], check=True)
    subprocess.run(args, check=True)
    return lib_path()


def load() -> C.CDLL:
    """Load libnka_hip.so.  Fails loudly when it is missing: there is no CPU path."""
    global _LIB
    if _LIB is None:
        path = lib_path()
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(nka_amd has no CPU fallback)")
        L = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB
