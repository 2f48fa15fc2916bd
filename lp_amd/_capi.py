"""ctypes binding of liblpipm.so (include/lpipm.h).  No fallback: if the HIP library is missing or
cannot be loaded this raises -- the product path never routes through a CPU implementation."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "liblpipm.so")

# lpipm_status
OK, UNCONSTRAINED, NUMERICAL_PROBLEM, INVALID_PARAMETER, INCOMPATIBLE_DIMENSIONS, INFEASIBLE, \
    UNBOUNDED, ITERATION_LIMIT = range(8)
ERR_HIP, ERR_NO_PROBLEM, ERR_UNSUPPORTED, ERR_BAD_ARGUMENT = 100, 101, 102, 103


class Opts(C.Structure):  # lpipm_opts
    _fields_ = [("tol", C.c_double), ("alpha0", C.c_double), ("max_iter", C.c_uint64),
                ("ip", C.c_int32), ("solver_type", C.c_int32), ("disp", C.c_int32)]


class IterRow(C.Structure):  # lpipm_iter_row
    _fields_ = [(k, C.c_double) for k in ("alpha", "rho_p", "rho_d", "rho_A", "rho_g", "rho_mu", "obj")]


class PhaseTimes(C.Structure):  # lpipm_phase_times
    _fields_ = [("adat_ms", C.c_double), ("potrf_ms", C.c_double), ("trsv_ms", C.c_double),
                ("gemv_ms", C.c_double), ("vec_ms", C.c_double), ("total_ms", C.c_double),
                ("adat_launches", C.c_uint64), ("iterations", C.c_uint64),
                ("gemv_passes", C.c_uint64)]


# every symbol include/lpipm.h declares: name -> (restype, argtypes)
_dp, _u64, _vp = C.POINTER(C.c_double), C.c_uint64, C.c_void_p
_dpp = C.POINTER(_dp)
SYMBOLS = {
    "lpipm_default_opts": (None, [C.POINTER(Opts)]),
    "lpipm_strerror": (C.c_char_p, [C.c_int]),
    "lpipm_last_error_detail": (C.c_char_p, []),
    "lpipm_device_count": (C.c_int, []),
    "lpipm_problem_build": (C.c_int, [_u64, _u64, _dp, _dp, _u64, _dp, _dp, _dp, _dp, _dp, _dp, C.POINTER(_u64)]),
    "lpipm_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "lpipm_destroy": (None, [_vp]),
    "lpipm_upload": (C.c_int, [_vp, _u64, _u64, _dp, _u64, _dp, _dp, C.c_double]),
    "lpipm_upload_slack": (C.c_int, [_vp, _u64, _u64, _dp, _u64, _dp, _dp, C.c_double, _u64]),
    "lpipm_upload_ub_eq": (C.c_int, [_vp, _u64, _u64, _dp, _u64, _dp, _u64, _dp, _u64, _dp, _dp, C.c_double]),
    "lpipm_solve": (C.c_int, [_vp, C.POINTER(Opts), _dp, _dp, C.POINTER(_u64), C.POINTER(IterRow)]),
    "lpipm_solve_device": (C.c_int, [_vp, C.POINTER(Opts), _vp, _dp, C.POINTER(_u64), C.POINTER(IterRow)]),
    "lpipm_solve_batch": (C.c_int, [_vp, _u64, C.POINTER(_u64), C.POINTER(_u64), _dpp, _dpp, _dpp, _dp,
                                    C.POINTER(Opts), _dpp, _dp, C.POINTER(_u64), C.POINTER(C.c_int32)]),
    "lpipm_set_collective": (C.c_int, [_vp, C.c_int, C.c_int, C.c_void_p, _vp]),
    "lpipm_set_collective_on_stream": (C.c_int, [_vp, C.c_int]),
    "lpipm_upload_nsplit": (C.c_int, [_vp, _u64, _u64, _u64, _dp, _u64, _dp, _dp, C.c_double]),
    "lpipm_upload_lockstep": (C.c_int, [_vp, _u64, _u64, _u64, _dpp, _dpp, _dpp, _dp]),
    "lpipm_solve_lockstep": (C.c_int, [_vp, C.POINTER(Opts), _dpp, _dp, C.POINTER(_u64), C.POINTER(C.c_int32)]),
    "lpipm_solve_lockstep_device": (C.c_int, [_vp, C.POINTER(Opts), _vp, _u64, _dp, C.POINTER(_u64), C.POINTER(C.c_int32)]),
    "lpipm_solve_batch_device": (C.c_int, [_vp, _u64, C.POINTER(_u64), C.POINTER(_u64), _dpp, _dpp, _dpp, _dp,
                                           C.POINTER(Opts), _vp, _u64, _dp, C.POINTER(_u64), C.POINTER(C.c_int32)]),
    "lpipm_set_batch_lockstep": (C.c_int, [_vp, C.c_int]),
    "lpipm_set_batch_concurrency": (C.c_int, [_vp, C.c_int]),
    "lpipm_set_profiling": (C.c_int, [_vp, C.c_int]),
    "lpipm_get_phase_times": (C.c_int, [_vp, C.POINTER(PhaseTimes)]),
    "lpipm_k_adat": (C.c_int, [_vp, _dp, _dp, C.c_int, _dp]),
    "lpipm_k_potrf": (C.c_int, [_vp, _u64, _dp, C.POINTER(C.c_int32), C.c_int, _dp]),
    "lpipm_k_chol_solve": (C.c_int, [_vp, _u64, C.c_int, _dp, _dp, C.c_int, _dp]),
    "lpipm_k_symv_residual": (C.c_int, [_vp, _u64, _dp, C.c_int, _dp, _dp, _dp]),
    "lpipm_k_qr_solve": (C.c_int, [_vp, _u64, _dp, C.c_int, _dp, _dp, C.POINTER(C.c_int32), _dp]),
    "lpipm_k_gemv_n": (C.c_int, [_vp, C.c_int, _dp, _dp, C.c_int, _dp]),
    "lpipm_k_gemv_t": (C.c_int, [_vp, C.c_int, _dp, _dp, C.c_int, _dp]),
    "lpipm_k_iteration": (C.c_int, [_vp, C.POINTER(Opts), C.c_int, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp,
                                    C.POINTER(C.c_int32)]),
    "lpipm_k_gemv_dual": (C.c_int, [_vp, _dp, _dp, _dp, _dp, C.c_int, _dp]),
    "lpipm_k_mfma_f64_probe": (C.c_int, [_vp, C.c_int, _dp, _dp]),
    "lpipm_solve_f32": (C.c_int, [_vp, _u64, _u64, C.POINTER(C.c_float), _u64, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float,
                                  C.POINTER(Opts), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_uint64), C.c_void_p]),
    "lpipm_k_generic_solve_f64": (C.c_int, [_vp, _u64, _u64, _dp, _u64, _dp, _dp, C.c_double, C.POINTER(Opts), _dp, _dp,
                                            C.POINTER(C.c_uint64), C.c_void_p]),
    "lpipm_synth_planted_lp": (C.c_int, [_u64, _u64, _u64, _dp, _dp, _dp, _dp]),
}

ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p)   # lpipm_allreduce_fn

_lib = None


def lib():
    """The loaded library.  Raises (never falls back) when liblpipm.so is absent or unloadable."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C lp_amd/csrc`).  There is no CPU fallback.")
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 (same SONAME as
        # /opt/rocm's).  If liblpipm.so pulled the system one in first, a later `import torch` would
        # be bound to it and fail to see the GPU ("No HIP GPUs are available").  Loading torch first
        # makes both sides share torch's runtime (device pointers of torch tensors stay valid here).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the header and the library ever diverge
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def strerror(code: int) -> str:
    return lib().lpipm_strerror(int(code)).decode()


def last_error_detail() -> str:
    return lib().lpipm_last_error_detail().decode()
