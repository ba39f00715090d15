"""ctypes binding of libfilterhip.so (include/filterhip.h) -- the only way Python reaches the
HIP kernels.  There is no fallback: if the shared library is missing or a call fails this
module raises.

Device memory, streams and torch.distributed come from PyTorch-ROCm (plumbing); torch is
imported BEFORE the library is loaded so that both resolve the same libamdhip64.so.7 and
device pointers / streams are interchangeable.
"""
import ctypes
import os

import torch  # noqa: F401  (must precede CDLL: shares the HIP runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfilterhip.so")

FK_OK = 0
FK_ERR_BAD_ARG, FK_ERR_UNSUPPORTED, FK_ERR_LAUNCH, FK_ERR_WORKSPACE = -1, -2, -3, -4
FK_LAYOUT_AOS, FK_LAYOUT_SOA = 0, 1
FK_MODEL_SHARED, FK_MODEL_PER_TRACK, FK_MODEL_PER_TRACK_STEP, FK_MODEL_PER_STEP = 0, 1, 2, 3
FK_STATUS_NOT_PD, FK_STATUS_NONFINITE, FK_STATUS_OVERRUN, FK_STATUS_INTERNAL, FK_STATUS_BAD_WEIGHTS = 1, 2, 4, 8, 16
FK_UKF_FLAG_PAIR_WEIGHTS = 1

c_i32, c_i64, c_f64, c_vp, c_sz = ctypes.c_int32, ctypes.c_int64, ctypes.c_double, ctypes.c_void_p, ctypes.c_size_t


class fk_kf_desc(ctypes.Structure):
    _fields_ = [("n", c_i32), ("m", c_i32), ("nu", c_i32), ("model_mode", c_i32),
                ("N", c_i64), ("T", c_i64), ("layout", c_i32), ("update_first", c_i32),
                ("alpha_sq", c_f64), ("flags", c_i32), ("reserved", c_i32)]


FK_KF_FLAG_R_JOSEPH_DIAG = 1
FK_KF_FLAG_COV_INTERLEAVED = 2
# a caller-supplied inverse (KalmanFilter.inv / rts_smoother(inv=...)): fk_kf_update_f64 / fk_kf_rts_f64 cut at the callable
FK_KF_FLAG_S_ONLY, FK_KF_FLAG_SI_GIVEN, FK_KF_FLAG_PP_ONLY, FK_KF_FLAG_PPINV_GIVEN = 4, 8, 16, 32

FK_ABI_VERSION = 4          # include/filterhip.h: the library must report exactly this


class fk_kf_extras(ctypes.Structure):
    _fields_ = [("y", c_vp), ("K", c_vp), ("S", c_vp), ("SI", c_vp), ("log_likelihood", c_vp), ("mahalanobis", c_vp)]


class fk_ukf_desc(ctypes.Structure):
    _fields_ = [("n", c_i32), ("m", c_i32), ("N", c_i64), ("T", c_i64), ("layout", c_i32),
                ("flags", c_i32), ("scale", c_f64)]


class fk_imm_desc(ctypes.Structure):
    _fields_ = [("n", c_i32), ("m", c_i32), ("n_models", c_i32), ("layout", c_i32), ("N", c_i64), ("T", c_i64),
                ("phase", c_i32), ("flags", c_i32)]


class FilterHipError(RuntimeError):
    """code: the FK_ERR_* value the library returned (None: raised by the Python layer)"""

    def __init__(self, msg, code=None):
        super().__init__(msg)
        self.code = code


# every symbol include/filterhip.h declares, with its signature
SIGNATURES = {
    "fk_kf_batch_filter_f64": (ctypes.c_int, [ctypes.POINTER(fk_kf_desc)] + [c_vp] * 16),
    "fk_kf_batch_filter_ex_f64": (ctypes.c_int, [ctypes.POINTER(fk_kf_desc)] + [c_vp] * 14 +
                                  [ctypes.POINTER(fk_kf_extras), c_vp, c_vp]),
    "fk_kf_predict_f64": (ctypes.c_int, [ctypes.POINTER(fk_kf_desc)] + [c_vp] * 8),
    "fk_kf_update_f64": (ctypes.c_int, [ctypes.POINTER(fk_kf_desc)] + [c_vp] * 12),
    "fk_kf_rts_f64": (ctypes.c_int, [ctypes.POINTER(fk_kf_desc)] + [c_vp] * 8 + [c_i32, c_vp, c_vp]),
    "fk_ut_sigma_points_f64": (ctypes.c_int, [c_i32, c_i64, c_i32, c_f64, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "fk_ut_transform_f64": (ctypes.c_int, [c_i32, c_i32, c_i64, c_i32] + [c_vp] * 7),
    "fk_ut_cross_variance_f64": (ctypes.c_int, [c_i32, c_i32, c_i32, c_i64, c_i32] + [c_vp] * 7),
    "fk_ut_linear_map_f64": (ctypes.c_int, [c_i32, c_i32, c_i32, c_i64, c_i32] + [c_vp] * 4),
    "fk_ukf_correct_f64": (ctypes.c_int, [c_i32, c_i32, c_i64, c_i32] + [c_vp] * 9),
    "fk_ukf_linear_batch_f64": (ctypes.c_int, [ctypes.POINTER(fk_ukf_desc)] + [c_vp] * 14),
    "fk_ukf_linear_rts_f64": (ctypes.c_int, [ctypes.POINTER(fk_ukf_desc)] + [c_vp] * 11),
    "fk_ukf_linear_supported": (ctypes.c_int, [ctypes.c_int32] * 4),
    "fk_kf_steadystate_f64": (ctypes.c_int, [ctypes.POINTER(fk_kf_desc)] + [c_vp] * 12),
    "fk_kf_update_correlated_f64": (ctypes.c_int, [ctypes.POINTER(fk_kf_desc)] + [c_vp] * 13),
    "fk_ukf_rts_correct_f64": (ctypes.c_int, [c_i32, c_i64, c_i32] + [c_vp] * 10),
    "fk_imm_batch_f64": (ctypes.c_int, [ctypes.POINTER(fk_imm_desc)] + [c_vp] * 17),
    "fk_imm_batch_ex_f64": (ctypes.c_int, [ctypes.POINTER(fk_imm_desc)] + [c_vp] * 8 + [ctypes.c_int32] + [c_vp] * 13),
    "fk_resample_systematic_f64": (ctypes.c_int, [c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "fk_resample_stratified_f64": (ctypes.c_int, [c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "fk_resample_multinomial_f64": (ctypes.c_int, [c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "fk_resample_residual_fill_f64": (ctypes.c_int, [c_i64, c_i64] + [c_vp] * 6),
    "fk_resample_residual_draw_f64": (ctypes.c_int, [c_i64, c_i64] + [c_vp] * 6),
    "fk_resample_gather_mean_f64": (ctypes.c_int, [c_i64, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "fk_cumsum_exact_f64": (ctypes.c_int, [c_i64, c_i64, c_vp, c_vp, c_i32, c_vp]),
    "fk_resample_workspace_bytes": (c_sz, [c_i64, c_i64]),
    "fk_multinomial_workspace_bytes": (c_sz, [c_i64, c_i64]),
    "fk_abi_version": (ctypes.c_int, []),
    "fk_build_arch": (ctypes.c_char_p, []),
    "fk_last_error": (ctypes.c_char_p, []),
    "fk_chunk_plan": (ctypes.c_int, [c_i64, c_i64, c_i32, c_i64, c_i32, c_vp, c_vp, c_vp]),
}

_lib = None


def lib():
    """Load libfilterhip.so (once).  Raises FilterHipError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FilterHipError(
                f"{LIB_PATH} not found: build it with `make -C filterpy_amd/csrc` "
                "(or __graft_entry__.build()). filterpy_amd has no CPU fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        # the version FIRST: a stale library (an older checkout's build) must say "rebuild", not die on a missing symbol
        try:
            handle.fk_abi_version.restype, handle.fk_abi_version.argtypes = ctypes.c_int, []
            have = handle.fk_abi_version()
        except AttributeError:
            have = None
        if have != FK_ABI_VERSION:
            raise FilterHipError(f"{LIB_PATH} reports ABI version {have}, this package binds version {FK_ABI_VERSION} "
                                 "(include/filterhip.h): rebuild the library (`make -C filterpy_amd/csrc` or __graft_entry__.build())")
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError:
                raise FilterHipError(f"{LIB_PATH} lacks {name}, which include/filterhip.h declares: rebuild the library") from None
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def check(rc, what):
    if rc != FK_OK:
        msg = lib().fk_last_error().decode(errors="replace")
        raise FilterHipError(f"{what} failed with code {rc}: {msg}", code=rc)
