"""ctypes binding of libgprc_native.so (include/gprc_native.h).

This is the only place the host mirror touches native code.  There is NO CPU fallback: if the
shared library is missing the import of the binding fails loudly, and if no MI355X is visible
every compute entry point raises `GprcError` (GPRC_ERR_NO_DEVICE).
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libgprc_native" + os.environ.get("GPRC_LIB_SUFFIX", "") + ".so")

# kernel ids (include/gprc_native.h gprc_kernel_id)
CONSTANT, LINEAR, POLYNOMIAL, SQREXP, GAMMAEXP, RATQUAD = range(6)

OK = 0
GPC_REFERENCE_STOP = 1
ERR_ARG, ERR_HIP, ERR_NOMEM, ERR_NOT_PD, ERR_DIVERGED, ERR_MAXITER, ERR_NO_DEVICE = -1, -2, -3, -4, -5, -6, -7


class GprcError(RuntimeError):
    """A negative gprc_status from the native library."""

    def __init__(self, status: int, message: str):
        super().__init__(f"gprc native error {status}: {message}")
        self.status = status
        self.message = message


class NotPositiveDefinite(ArithmeticError):
    """LAPACK-style info > 0: what R's chol() reports (reference R/GPRclass.R:142)."""

    def __init__(self, info: int):
        super().__init__(f"the leading minor of order {info} is not positive definite")
        self.info = info


_dp = C.POINTER(C.c_double)
_i64 = C.c_int64
_vp = C.c_void_p

# name -> (restype, argtypes); must list every GPRC_API symbol of the header
PROTOTYPES = {
    "gprc_abi_version": (C.c_int, []),
    "gprc_last_error": (C.c_char_p, []),
    "gprc_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "gprc_ctx_create": (C.c_int, [C.c_int, _vp, C.POINTER(_vp)]),
    "gprc_ctx_destroy": (C.c_int, [_vp]),
    "gprc_ctx_synchronize": (C.c_int, [_vp]),
    "gprc_ctx_trim": (C.c_int, [_vp]),
    "gprc_kernel_matrix": (C.c_int, [_vp, C.c_int, _dp, C.c_int, _vp, _i64, _i64, _vp, _i64, _vp, _i64]),
    "gprc_kernel_colwise": (C.c_int, [_vp, C.c_int, _dp, C.c_int, _vp, _vp, _i64, _i64, _vp]),
    "gprc_gpr_fit": (C.c_int, [_vp, C.c_int, _dp, C.c_int, _vp, _i64, _i64, _vp, C.c_double, C.POINTER(_vp)]),
    "gprc_gpr_fit_retry": (C.c_int, [_vp, C.c_int, _dp, C.c_int, _vp, _i64, _i64, _vp, C.c_double, C.POINTER(_vp),
                                     C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "gprc_gpr_log_marginal": (C.c_int, [_vp, C.c_int, _dp, C.c_int, _vp, _i64, _i64, _vp, C.c_double, C.POINTER(C.c_double)]),
    "gprc_fit_gradient": (C.c_int, [_vp, C.c_int, _dp, C.c_int, _vp, _i64, _i64, _vp, _dp]),
    "gprc_gpr_predict": (C.c_int, [_vp, _vp, _i64, C.c_int, _vp, _vp]),
    "gprc_model_dims": (C.c_int, [_vp, C.POINTER(_i64), C.POINTER(_i64)]),
    "gprc_model_get_L": (C.c_int, [_vp, _vp, _i64]),
    "gprc_gpr_get_alpha": (C.c_int, [_vp, _vp]),
    "gprc_gpr_get_logp": (C.c_int, [_vp, C.POINTER(C.c_double)]),
    "gprc_gpr_get_noise": (C.c_int, [_vp, C.POINTER(C.c_double)]),
    "gprc_model_free": (C.c_int, [_vp]),
    "gprc_gpc_fit": (C.c_int, [_vp, C.c_int, _dp, C.c_int, _vp, _i64, _i64, _vp, C.c_double, C.c_int, C.c_int, C.POINTER(_vp),
                               C.POINTER(C.c_int)]),
    "gprc_gpc_predict_latent": (C.c_int, [_vp, _vp, _i64, _vp, _vp]),
    "gprc_gpc_predict_class": (C.c_int, [_vp, _vp, _i64, _vp]),
    "gprc_class_probability": (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    "gprc_gpc_get_f_hat": (C.c_int, [_vp, _vp]),
    "gprc_gpc_get_logq": (C.c_int, [_vp, C.POINTER(C.c_double)]),
    "gprc_panel_width": (_i64, []),
    "gprc_pad": (_i64, [_i64]),
    "gprc_panel_count": (_i64, [_i64]),
    "gprc_panel_offset": (_i64, [_i64, _i64]),
    "gprc_panel_elems": (_i64, [_i64, _i64]),
    "gprc_packed_size": (_i64, [_i64]),
    "gprc_winv_size": (_i64, [_i64]),
    "gprc_dev_fill_panel": (C.c_int, [_vp, C.c_int, _dp, C.c_int, _vp, _i64, _i64, _i64, C.c_double, _vp, _i64]),
    "gprc_dev_factor_panel": (C.c_int, [_vp, _vp, _i64, _i64, _vp, _vp]),
    "gprc_dev_factor_subpanel": (C.c_int, [_vp, _vp, _i64, _i64, C.c_int, C.c_int, _vp, _vp]),
    "gprc_dev_factor_all": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _vp]),
    "gprc_factor_service": (C.c_int, [C.c_int]),
    "gprc_solve_inv_size": (_i64, [_i64]),
    "gprc_dev_solve_prepare": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _i64, _i64]),
    "gprc_dev_update_trailing": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _i64, _i64]),
    "gprc_dev_update_range": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64]),
    "gprc_trsv_work_size": (_i64, [_i64]),
    "gprc_dev_trsv": (C.c_int, [_vp, _vp, _vp, _i64, _vp, C.c_int, _vp]),
    "gprc_dev_trsv_step": (C.c_int, [_vp, _vp, _vp, _i64, _vp, C.c_int, _i64, _vp]),
    "gprc_dev_fill_cross": (C.c_int, [_vp, C.c_int, _dp, C.c_int, _vp, _i64, _i64, _i64, _vp, _i64, _i64, _vp, _i64]),
    "gprc_rowreduce_splits": (_i64, [_i64]),
    "gprc_dev_row_reduce": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp]),
    "gprc_dev_logp": (C.c_int, [_vp, _vp, _i64, _i64, _vp, _vp, _vp]),
    "gprc_gpr_model_from_device": (C.c_int, [_vp, C.c_int, _dp, C.c_int, _vp, _i64, _i64, _vp, _vp, _vp, _vp, C.c_double,
                                             C.c_double, C.POINTER(_vp)]),
    "gprc_mgpu_create": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(_vp)]),
    "gprc_mgpu_destroy": (C.c_int, [_vp]),
    "gprc_mgpu_ranks": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "gprc_mgpu_calibrate": (C.c_int, [_vp, _i64, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    "gprc_mgpu_exchange_mode": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "gprc_mgpu_stats": (C.c_int, [_vp, C.POINTER(C.c_double), C.c_int]),
    "gprc_mgpu_gpr_fit": (C.c_int, [_vp, C.c_int, _dp, C.c_int, _vp, _i64, _i64, _vp, C.c_double, C.POINTER(_vp)]),
    "gprc_mgpu_gpr_fit_retry": (C.c_int, [_vp, C.c_int, _dp, C.c_int, _vp, _i64, _i64, _vp, C.c_double, C.POINTER(_vp),
                                          C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "gprc_mgpu_gpr_predict": (C.c_int, [_vp, _vp, _i64, _vp, _vp]),
    "gprc_mgpu_gpr_get_alpha": (C.c_int, [_vp, _vp]),
    "gprc_mgpu_gpr_get_logp": (C.c_int, [_vp, C.POINTER(C.c_double)]),
    "gprc_mgpu_gpr_get_noise": (C.c_int, [_vp, C.POINTER(C.c_double)]),
    "gprc_mgpu_model_rank": (C.c_int, [_vp, C.c_int, C.POINTER(_vp)]),
    "gprc_mgpu_model_free": (C.c_int, [_vp]),
    "gprc_mvn_factor": (C.c_int, [_vp, _vp, _i64, _i64, C.c_double, _vp, C.POINTER(C.c_int)]),
    "gprc_mvn_sample": (C.c_int, [_vp, _vp, _i64, _i64, _vp, C.c_double, _vp, _i64, _vp, C.POINTER(C.c_int)]),
    "gprc_sym_eigen": (C.c_int, [_vp, _vp, _i64, _i64, _vp, _vp, C.POINTER(C.c_int)]),
    "gprc_combine_all": (C.c_int, [_vp, _vp, C.POINTER(_i64), C.c_int, _vp]),
    "gprc_prof_enable": (C.c_int, [C.c_int]),
    "gprc_prof_panel_trace": (C.c_int, [_vp, C.c_int, C.POINTER(_i64), C.c_int]),
    "gprc_prof_service_trace": (C.c_int, [_vp, C.POINTER(_i64), C.c_int]),
    "gprc_prof_wait_timeout": (C.c_int, [C.POINTER(C.c_int), C.c_int]),
    "gprc_prof_reset": (C.c_int, []),
    "gprc_prof_kinds": (C.c_int, []),
    "gprc_prof_summary": (C.c_int, [C.c_int, C.POINTER(_i64), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "gprc_dev_solve_rows": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _i64, _i64]),
}

_lib = None
_lib_lock = threading.Lock()


def _share_hip_runtime_with_torch() -> None:
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME libamdhip64.so.7) and link it by the UNVERSIONED
    name.  If this library (linked against /opt/rocm's libamdhip64.so.7) is loaded first and torch later, the process
    ends up with two HIP runtimes and device pointers / streams stop being interchangeable.  Loading torch's copy
    first makes both resolve to the same object (ours by SONAME, torch's by path).  torch itself is not imported."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return  # torch already brought its runtime in; our NEEDED libamdhip64.so.7 matches its SONAME
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib() -> C.CDLL:
    """Load the shared library (once).  Raises ImportError with build instructions if absent."""
    global _lib
    with _lib_lock:
        if _lib is None:
            _share_hip_runtime_with_torch()
            if not os.path.exists(LIB_PATH):
                raise ImportError(
                    f"{LIB_PATH} not found: build it with gaussian-process-regression_amd/csrc/build.sh "
                    "(or __graft_entry__.build()).  The gprc MI355X path has no CPU fallback.")
            handle = C.CDLL(LIB_PATH)
            for name, (res, args) in PROTOTYPES.items():
                fn = getattr(handle, name)
                fn.restype = res
                fn.argtypes = args
            if handle.gprc_abi_version() != 1:
                raise ImportError("libgprc_native.so ABI version mismatch")
            _lib = handle
    return _lib


def last_error() -> str:
    return (lib().gprc_last_error() or b"").decode("utf-8", "replace")


def check(rc: int) -> int:
    """0 -> 0; info > 0 -> NotPositiveDefinite; < 0 -> GprcError."""
    if rc == 0:
        return 0
    if rc > 0:
        raise NotPositiveDefinite(rc)
    raise GprcError(rc, last_error())


def device_count() -> int:
    c = C.c_int(0)
    check(lib().gprc_device_count(C.byref(c)))
    return c.value


INFO_WAIT_TIMEOUT = -99   # include/gprc_native.h: a device-side dependency wait ran out
PROF_KINDS = ["fill", "potf2_inv", "trsm_panel", "gemm_inner_k128", "trailing_update", "solve_update_k512", "trsv",
              "row_reduce", "cov_syrk", "deriv_rowsum", "jacobi_sweep", "solve_left", "trailing_left", "panel_fused", "solve_panel"]


def prof_summary():
    """{kind: dict(count, ms, flops, bytes)} of the launches seen since gprc_prof_reset()."""
    out = {}
    for kind, name in enumerate(PROF_KINDS):
        cnt, ms, fl, by = _i64(), C.c_double(), C.c_double(), C.c_double()
        check(lib().gprc_prof_summary(kind, C.byref(cnt), C.byref(ms), C.byref(fl), C.byref(by)))
        out[name] = dict(count=cnt.value, ms=ms.value, flops=fl.value, bytes=by.value)
    return out


def params_array(params):
    p = np.ascontiguousarray(np.asarray(params, dtype=np.float64).ravel())
    return p, p.ctypes.data_as(_dp), int(p.size)


def ptr(a) -> int:
    """Address of a host numpy array or a torch tensor (host or device), as an integer."""
    if a is None:
        return 0
    if isinstance(a, np.ndarray):
        return a.ctypes.data
    if hasattr(a, "data_ptr"):
        return a.data_ptr()
    raise TypeError(f"cannot take the address of {type(a)!r}")


class Context:
    """gprc_ctx: one GPU + one HIP stream.  `stream` is a raw hipStream_t (int) or None."""

    def __init__(self, device: int = 0, stream: int | None = None):
        self._h = _vp()
        check(lib().gprc_ctx_create(int(device), _vp(stream) if stream else None, C.byref(self._h)))
        self.device = int(device)

    @property
    def handle(self):
        if not self._h:
            raise GprcError(ERR_ARG, "context already destroyed")
        return self._h

    def synchronize(self):
        check(lib().gprc_ctx_synchronize(self.handle))

    def close(self):
        if self._h:
            lib().gprc_ctx_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx: dict[int, Context] = {}


def default_context(device: int = 0) -> Context:
    """Process-wide context per device (created on first use; raises GprcError without a GPU)."""
    ctx = _default_ctx.get(device)
    if ctx is None:
        ctx = Context(device)
        _default_ctx[device] = ctx
    return ctx
