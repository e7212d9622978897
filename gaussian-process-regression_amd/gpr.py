"""GPR: host mirror of the reference's R6 class `GPR` and its six kernel subclasses
(R/GPRclass.R:116-351).  Construction = kernel fill + Cholesky + alpha + logp on the GPU
(`GPR$initialize`, :127-154); `predict` = `GPR$predict` (:155-170).  Same argument names, orders,
defaults, return shapes, warnings and error texts; the plot methods (:171-228) are rendering code and
stay with the reference.
"""
from __future__ import annotations

import ctypes as C
import warnings

import numpy as np

from . import _native as nat
from .covfunc import (as_points, cov_func, constant, linear, polynomial, sqrexp, gammaexp, rationalquadratic,
                      require_tagged)

__all__ = ["GPR", "GPR_constant", "GPR_linear", "GPR_polynomial", "GPR_sqrexp", "GPR_gammaexp", "GPR_rationalquadratic"]


def _r_num(x) -> str:
    """sprintf("%s", <double>) as R prints it (15 significant digits)."""
    return f"{float(x):.15g}"


def _is_numeric_vector(y):
    y = np.asarray(y)
    return y.ndim == 1 and y.dtype.kind in "fiub"


class _ReadOnly:
    """Active binding: readable, `stop("`$name` is read only")` on assignment (R/GPRclass.R:230-280)."""

    def __init__(self, name, getter):
        self.name, self.getter = name, getter

    def __get__(self, obj, objtype=None):
        return self if obj is None else self.getter(obj)

    def __set__(self, obj, value):
        raise AttributeError(f"`${self.name}` is read only")


class GPR:
    """GPR$new(X, y, noise = 0, k, cov_names)  --  R/GPRclass.R:127-128."""

    def __init__(self, X, y, noise=0, k=None, cov_names=None, *, ctx=None, devices=None, rccl=False, exchange="broadcast",
                 lookahead=True):
        if k is None:
            # the reference default: k = fit(X, y, noise, cov_names)$func (R/GPRclass.R:127); all six kernels of the
            # default list are covered (Brent for the one-parameter kernels and the polynomial degree loop, vmmin/BFGS
            # with the reference's dens_deriv for gammaexp and rationalquadratic).
            from .fit import fit as _fit
            k = _fit(X, y, noise, cov_names, ctx=ctx)["func"]
        Xa = np.asarray(X)
        if Xa.dtype.kind not in "fiub" or not _is_numeric_vector(y):
            raise TypeError("is.numeric(X), is.vector(y), is.numeric(y) are not all TRUE")   # :129
        if np.ndim(noise) != 0 or not isinstance(noise, (int, float, np.floating, np.integer)) or not noise >= 0:
            raise ValueError("is.numeric(noise), length(noise) == 1, noise >= 0 are not all TRUE")  # :130
        Xm = as_points(Xa)                                                                  # :132
        y = np.ascontiguousarray(np.asarray(y, dtype=np.float64))
        if y.size != Xm.shape[1]:
            raise ValueError("length(y) == ncol(X) is not TRUE")                            # :133
        k = require_tagged(k, "GPR")
        d, n = Xm.shape
        self._ctx = ctx or nat.default_context()
        self._X, self._y, self._k = Xm, y, k
        self._L = None
        self._model = C.c_void_p()
        self._mgpu = self._mmodel = None
        _, pp, npar = nat.params_array(k.native_params(d))
        noise_used, attempts = C.c_double(), C.c_int()
        if devices is not None and (len(devices) > 1 or rccl):
            # the reference's GPR$new over several GPUs from this ONE process (gprc_mgpu_*; the R side: options(gprc.devices = ...)).
            # `devices` may list a device several times: virtual ranks sharing a GPU (peer copies only).
            # exchange: the form of the one exchange step -- "broadcast" (rooted), "scatter_allgather" (large-message form) or
            # "auto" (both timed at creation, the faster kept); lookahead=False: factor panel p+1 only after update p.
            self._mgpu = _mgpu_for(tuple(int(v) for v in devices), bool(rccl), exchange, bool(lookahead))
            self._mmodel = C.c_void_p()
            rc = nat.lib().gprc_mgpu_gpr_fit_retry(self._mgpu, k.gprc_kernel[0], pp, npar, Xm.ctypes.data, d, n, y.ctypes.data, float(noise),
                                                   C.byref(self._mmodel), C.byref(noise_used), C.byref(attempts))
        else:
            rc = nat.lib().gprc_gpr_fit_retry(self._ctx.handle, k.gprc_kernel[0], pp, npar, Xm.ctypes.data, d, n, y.ctypes.data,
                                              float(noise), C.byref(self._model), C.byref(noise_used), C.byref(attempts))
        if rc == nat.ERR_NOT_PD:
            raise ArithmeticError("Inputs lead to non positive definite covariance matrix. "
                                  "Try using a larger noise or a smaller lengthscale.")      # :149
        nat.check(rc)
        if attempts.value > 1:                                                               # :144
            warnings.warn(f"Noise got changed to {_r_num(noise_used.value)} to avoid errors in cholesky decomposition")
        self._noise = noise_used.value
        alpha = np.empty(n)
        lp = C.c_double()
        if self._mmodel is not None:
            nat.check(nat.lib().gprc_mgpu_gpr_get_alpha(self._mmodel, alpha.ctypes.data))
            nat.check(nat.lib().gprc_mgpu_gpr_get_logp(self._mmodel, C.byref(lp)))
            nat.check(nat.lib().gprc_mgpu_model_rank(self._mmodel, 0, C.byref(self._model)))   # rank 0's replica: $L, full covariance
        else:
            nat.check(nat.lib().gprc_gpr_get_alpha(self._model, alpha.ctypes.data))
            nat.check(nat.lib().gprc_gpr_get_logp(self._model, C.byref(lp)))
        self._alpha, self._logp = alpha, lp.value

    @classmethod
    def new(cls, *args, **kwargs):
        """`GPR$new(...)` spelling of the constructor."""
        return cls(*args, **kwargs)

    def predict(self, X_star, pointwise_var=True):
        """GPR$predict(X_star, pointwise_var = TRUE)  --  R/GPRclass.R:155-170.
        pointwise: n* x 2 array cbind(mean, variance); otherwise [mean (n* x 1), covariance (n* x n*)]."""
        d = self._X.shape[0]
        Xs = np.asarray(X_star)
        if Xs.dtype.kind not in "fiub" or Xs.size % d:
            raise ValueError("is.numeric(X_star), length(X_star) %% nrow(self$X) == 0 are not all TRUE")  # :156
        Xs = as_points(Xs, d=d, what="X_star") if Xs.ndim <= 1 else as_points(Xs, what="X_star")
        if Xs.shape[0] != d:
            raise ValueError("X_star must have nrow(X) rows")
        ns = Xs.shape[1]
        mean = np.empty(ns)
        if pointwise_var:
            var = np.empty(ns)
            if self._mmodel is not None:   # the test points are sliced over the ranks
                nat.check(nat.lib().gprc_mgpu_gpr_predict(self._mmodel, Xs.ctypes.data, ns, mean.ctypes.data, var.ctypes.data))
            else:
                nat.check(nat.lib().gprc_gpr_predict(self._model, Xs.ctypes.data, ns, 1, mean.ctypes.data, var.ctypes.data))
            return np.column_stack([mean, var])                                              # :165
        cov = np.empty((ns, ns), order="F")
        nat.check(nat.lib().gprc_gpr_predict(self._model, Xs.ctypes.data, ns, 0, mean.ctypes.data, cov.ctypes.data))
        return [mean.reshape(-1, 1), cov]                                                    # :168

    def posterior_draws(self, n=5, limits=None, length_out=200, *, z=None, rng=None):
        """The data behind GPR$plot_posterior_draws(n = 5, limits = expand_range(X), length.out = 200)  --
        R/GPRclass.R:190-199 (one-dimensional inputs only, :192-195): test points, cbind(mean, diag(cov)) and n draws
        from the posterior, multivariate_normal(n, mean, cov).  Returns dict(x, y (length.out x 2), z (length.out x n)).
        The ggplot layer itself is not part of this package."""
        from .sampling import expand_range, multivariate_normal
        if self._X.shape[0] > 1:
            raise ValueError("No plot method for multidimensional data.")                    # :192-193
        lo, hi = expand_range(self._X) if limits is None else (float(limits[0]), float(limits[1]))
        testpoints = np.linspace(lo, hi, int(length_out))                                    # seq(..., length.out)
        mean, cov = self.predict(testpoints, pointwise_var=False)                            # :197
        y = np.column_stack([mean[:, 0], np.diag(cov)])                                      # :198
        zz = multivariate_normal(n, mean[:, 0], cov, z=z, rng=rng)                           # :199
        return {"x": testpoints, "y": y, "z": zz}

    def posterior_variance(self, where, limits=None, length_out=200):
        """The data behind GPR$plot_posterior_variance(where, limits, length.out)  --  R/GPRclass.R:211-221: posterior
        covariance between each test point and each point of `where`.  Returns dict(x, y (length.out x length(where)))."""
        from .sampling import expand_range
        if self._X.shape[0] > 1:
            raise ValueError("No plot method for multidimensional data.")
        where = np.atleast_1d(np.asarray(where, dtype=np.float64))
        lo, hi = expand_range(self._X) if limits is None else (float(limits[0]), float(limits[1]))
        testpoints = np.linspace(lo, hi, int(length_out))
        cov = self.predict(np.concatenate([where, testpoints]), pointwise_var=False)[1]
        return {"x": testpoints, "y": cov[where.size:, :where.size]}                         # [(len + 1):(len + length(testpoints)), 1:len]

    def _get_L(self):
        if self._L is None:  # lazy: n x n doubles cross PCIe only when `$L` is read
            n = self._X.shape[1]
            L = np.empty((n, n), order="F")
            nat.check(nat.lib().gprc_model_get_L(self._model, L.ctypes.data, n))
            self._L = L
        return self._L

    X = _ReadOnly("X", lambda s: s._X)
    k = _ReadOnly("k", lambda s: s._k)
    y = _ReadOnly("y", lambda s: s._y)
    noise = _ReadOnly("noise", lambda s: s._noise)
    L = _ReadOnly("L", _get_L)
    alpha = _ReadOnly("alpha", lambda s: s._alpha)
    logp = _ReadOnly("logp", lambda s: s._logp)

    def close(self):
        if getattr(self, "_mmodel", None):
            nat.lib().gprc_mgpu_model_free(self._mmodel)     # owns the per-rank replicas, including the borrowed self._model
            self._mmodel = None
            self._model = C.c_void_p()
        if getattr(self, "_model", None):
            nat.lib().gprc_model_free(self._model)
            self._model = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_mgpu_cache = {}


MGPU_RCCL, MGPU_NO_LOOKAHEAD, MGPU_SCATTER_ALLGATHER, MGPU_AUTO_EXCHANGE = 1, 2, 4, 8   # include/gprc_native.h


def _mgpu_for(devices, rccl, exchange="broadcast", lookahead=True):
    """One gprc_mgpu per (devices, flags) for the life of the process: creating streams / RCCL communicators is not free, and
    fitted models keep handles into it."""
    try:
        ex = {"broadcast": 0, "scatter_allgather": MGPU_SCATTER_ALLGATHER, "auto": MGPU_AUTO_EXCHANGE}[exchange]
    except KeyError:
        raise ValueError('exchange must be "broadcast", "scatter_allgather" or "auto"') from None
    flags = (MGPU_RCCL if rccl else 0) | (0 if lookahead else MGPU_NO_LOOKAHEAD) | ex
    key = (devices, flags)
    h = _mgpu_cache.get(key)
    if h is None:
        h = C.c_void_p()
        arr = (C.c_int * len(devices))(*devices)
        nat.check(nat.lib().gprc_mgpu_create(arr, len(devices), flags, C.byref(h)))
        _mgpu_cache[key] = h
    return h


def mgpu_stats(handle):
    """gprc_mgpu_stats as a dict: what the last fit / predict on this gprc_mgpu did (a schedule-rehearsal record)."""
    g = C.c_int()
    nat.check(nat.lib().gprc_mgpu_ranks(handle, C.byref(g)))
    out = (C.c_double * (10 + 3 * g.value))()
    nat.check(nat.lib().gprc_mgpu_stats(handle, out, len(out)))
    names = ["ranks", "panels", "exchange_mode", "exchange_ops", "bytes_in_per_rank", "event_pairs", "far_passes", "lookahead_updates",
             "fit_ms", "predict_ms"]
    rec = {k: (float(out[i]) if k.endswith("_ms") else int(out[i])) for i, k in enumerate(names)}
    rec["exchange"] = ["copies/broadcast", "rccl/broadcast", "copies/scatter_allgather", "rccl/scatter_allgather"][rec["exchange_mode"]]
    rec["per_rank"] = [{"rank": r, "fill_sweep_ms": round(out[10 + 3 * r], 3), "alpha_logp_ms": round(out[11 + 3 * r], 3),
                        "predict_ms": round(out[12 + 3 * r], 3)} for r in range(g.value)]
    return rec


def _fitted(X, y, noise, kernel, ctx=None):
    """fit(X, y, noise, "<kernel>")$par -- the default of every kernel-specific constructor's parameters
    (R/GPRclass.R:286,297,308-309,320,332-333,344-345).  The reference evaluates it once per missing argument; the
    result is the same, so one call serves both."""
    from .fit import fit as _fit
    return _fit(X, y, noise, [kernel], ctx=ctx)["par"]


def _len(x):
    return np.size(x)


class GPR_constant(GPR):
    """GPR.constant$new(X, y, noise, c)  --  R/GPRclass.R:284-292."""

    def __init__(self, X, y, noise, c=None, **kw):
        if c is None:
            c = _fitted(X, y, noise, "constant", kw.get("ctx"))[0]
        if not (np.isreal(c) and np.all(np.asarray(c) > 0)):
            raise ValueError("is.numeric(c), c > 0 are not all TRUE")
        super().__init__(X, y, noise, cov_func(constant, c=c), **kw)


class GPR_linear(GPR):
    """GPR.linear$new(X, y, noise, sigma)  --  R/GPRclass.R:295-303."""

    def __init__(self, X, y, noise, sigma=None, **kw):
        if sigma is None:
            sigma = _fitted(X, y, noise, "linear", kw.get("ctx"))[0]          # a scalar: passes :298 only for one-dimensional X
        if _len(sigma) != as_points(X).shape[0]:
            raise ValueError("length(sigma) == nrow(X) is not TRUE")
        super().__init__(X, y, noise, cov_func(linear, sigma=sigma), **kw)


class GPR_polynomial(GPR):
    """GPR.polynomial$new(X, y, noise, sigma, p)  --  R/GPRclass.R:306-315."""

    def __init__(self, X, y, noise, sigma=None, p=None, **kw):
        if sigma is None or p is None:
            par = _fitted(X, y, noise, "polynomial", kw.get("ctx"))
            sigma, p = (par[0] if sigma is None else sigma), (par[1] if p is None else p)
        if _len(sigma) != 1 or _len(p) != 1:
            raise ValueError("length(sigma) == 1, length(p) == 1 are not all TRUE")
        super().__init__(X, y, noise, cov_func(polynomial, sigma=sigma, p=p), **kw)


class GPR_sqrexp(GPR):
    """GPR.sqrexp$new(X, y, noise, l)  --  R/GPRclass.R:318-327."""

    def __init__(self, X, y, noise, l=None, **kw):
        if l is None:
            l = _fitted(X, y, noise, "sqrexp", kw.get("ctx"))[0]
        if _len(l) != 1:
            raise ValueError("length(l) == 1 is not TRUE")
        super().__init__(X, y, noise, cov_func(sqrexp, l=l), **kw)


class GPR_gammaexp(GPR):
    """GPR.gammaexp$new(X, y, noise, gamma, l)  --  R/GPRclass.R:330-339."""

    def __init__(self, X, y, noise, gamma=None, l=None, **kw):
        if gamma is None or l is None:   # as written in :332-333: gamma <- $par[[1]], l <- $par[[2]], although fit's par is (l, gamma)
            par = _fitted(X, y, noise, "gammaexp", kw.get("ctx"))
            gamma, l = (par[0] if gamma is None else gamma), (par[1] if l is None else l)
        if _len(gamma) != 1 or _len(l) != 1:
            raise ValueError("length(gamma) == 1, length(l) == 1 are not all TRUE")
        super().__init__(X, y, noise, cov_func(gammaexp, l=l, gamma=gamma), **kw)


class GPR_rationalquadratic(GPR):
    """GPR.rationalquadratic$new(X, y, noise, alpha, l)  --  R/GPRclass.R:342-351."""

    def __init__(self, X, y, noise, alpha=None, l=None, **kw):
        if alpha is None or l is None:   # as written in :344-345: alpha <- $par[[1]], l <- $par[[2]], although fit's par is (l, alpha)
            par = _fitted(X, y, noise, "rationalquadratic", kw.get("ctx"))
            alpha, l = (par[0] if alpha is None else alpha), (par[1] if l is None else l)
        if _len(alpha) != 1 or _len(l) != 1:
            raise ValueError("length(alpha) == 1, length(l) == 1 are not all TRUE")
        super().__init__(X, y, noise, cov_func(rationalquadratic, l=l, alpha=alpha), **kw)


# `GPR.sqrexp$new(...)` reads `GPR.sqrexp.new(...)` here
GPR.constant = GPR_constant
GPR.linear = GPR_linear
GPR.polynomial = GPR_polynomial
GPR.sqrexp = GPR_sqrexp
GPR.gammaexp = GPR_gammaexp
GPR.rationalquadratic = GPR_rationalquadratic
