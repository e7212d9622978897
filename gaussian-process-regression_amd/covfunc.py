"""Covariance-function layer: the reference's operator API (R/GPRclass.R:353-357, 378-403, 424-427).

`cov_func(func, ...)` fixes the parameters of one of the six kernel generics and returns a callable
`k(x, y)` obeying the reference's closure contract (two d x m matrices -> the m kernel values of their
columns).  The returned closure carries `gprc_kernel = (id, params)`, which is what routes
`covariance_matrix`, `GPR` and `GPC` to the fused HIP fill kernel.  Arbitrary user closures are part of
the reference API but not of this hot path: they stay on the reference's R implementation, and are
rejected here loudly rather than evaluated on the CPU.
"""
from __future__ import annotations

import numpy as np

from . import _native as nat

__all__ = ["cov_func", "covariance_matrix", "constant", "linear", "polynomial", "sqrexp", "gammaexp",
           "rationalquadratic", "CovFunc", "KernelGeneric", "as_points"]


def as_points(X, d=None, what="X"):
    """Numeric d x m matrix in the reference's layout.  A bare vector becomes 1 x n (R/GPRclass.R:132)
    or, when `d` is given, d x (len/d) filled column by column (R/GPRclass.R:157-159)."""
    X = np.asarray(X)
    if X.dtype.kind not in "fiub":
        raise TypeError(f"is.numeric({what}) is not TRUE")
    X = X.astype(np.float64, copy=False)
    if X.ndim == 0:
        X = X.reshape(1)
    if X.ndim == 1:
        if d is None:
            X = X.reshape(1, -1)
        else:
            if X.size % d:
                raise ValueError(f"length({what}) %% nrow(X) == 0 is not TRUE")
            X = X.reshape(d, -1, order="F")
    elif X.ndim != 2:
        raise ValueError(f"{what} must be a vector or a d x n matrix")
    return np.asfortranarray(X)


class KernelGeneric:
    """One of the reference's S3 kernel generics (`sqrexp(x, y, l)` ...).  Calling it evaluates the
    column-wise kernel on the GPU; its main use is as the first argument of `cov_func`."""

    def __init__(self, name, kid, arg_names):
        self.__name__ = name
        self.kernel_id = kid
        self.arg_names = tuple(arg_names)

    def bind(self, args, kwargs):
        """-> dict name->value following R's argument matching (named first, then positional)."""
        vals = dict(kwargs)
        for k in vals:
            if k not in self.arg_names:
                raise TypeError(f"unused argument ({k}) for kernel {self.__name__}")
        free = [a for a in self.arg_names if a not in vals]
        if len(args) > len(free):
            raise TypeError(f"too many arguments for kernel {self.__name__}")
        for a, v in zip(free, args):
            vals[a] = v
        missing = [a for a in self.arg_names if a not in vals]
        if missing:
            raise TypeError(f'argument "{missing[0]}" is missing, with no default')
        return vals

    def param_vector(self, vals):
        return np.concatenate([np.atleast_1d(np.asarray(vals[a], dtype=np.float64)).ravel() for a in self.arg_names])

    def __call__(self, x, y, *args, **kwargs):
        return CovFunc(self, self.bind(args, kwargs))(x, y)

    def __repr__(self):
        return f"<gprc kernel {self.__name__}({', '.join(('x', 'y') + self.arg_names)})>"


# argument order = the reference signatures R/GPRclass.R:381,385,389,393,397,401
constant = KernelGeneric("constant", nat.CONSTANT, ("c",))
linear = KernelGeneric("linear", nat.LINEAR, ("sigma",))
polynomial = KernelGeneric("polynomial", nat.POLYNOMIAL, ("sigma", "p"))
sqrexp = KernelGeneric("sqrexp", nat.SQREXP, ("l",))
gammaexp = KernelGeneric("gammaexp", nat.GAMMAEXP, ("l", "gamma"))
rationalquadratic = KernelGeneric("rationalquadratic", nat.RATQUAD, ("l", "alpha"))


class CovFunc:
    """The closure `function(x, y) func(x, y, ...)` (R/GPRclass.R:426), tagged for native dispatch."""

    def __init__(self, func: KernelGeneric, values: dict):
        self.func = func
        self.values = dict(values)
        self.params = func.param_vector(values)
        self.gprc_kernel = (func.kernel_id, self.params)

    def native_params(self, d):
        """Parameter vector checked against the input dimension (linear: sigma of length 1 or d)."""
        if self.func.kernel_id == nat.LINEAR and self.params.size not in (1, d):
            raise ValueError("length(sigma) == nrow(X) is not TRUE")
        return self.params

    def __call__(self, x, y, ctx=None):
        x = np.asarray(x, dtype=np.float64)
        y = np.asarray(y, dtype=np.float64)
        scalar = x.ndim <= 1 and y.ndim <= 1  # the .numeric methods: two vectors -> one value
        if scalar:
            x = x.reshape(-1, 1)
            y = y.reshape(-1, 1)
        x, y = np.asfortranarray(x), np.asfortranarray(y)
        if x.shape != y.shape:
            raise ValueError("kernel arguments must have the same shape")
        d, m = x.shape
        out = np.empty(m)
        ctx = ctx or nat.default_context()
        _, pp, npar = nat.params_array(self.native_params(d))
        nat.check(nat.lib().gprc_kernel_colwise(ctx.handle, self.func.kernel_id, pp, npar, x.ctypes.data, y.ctypes.data, d, m,
                                                 out.ctypes.data))
        return float(out[0]) if scalar else out

    def __repr__(self):
        args = ", ".join(f"{k} = {v}" for k, v in self.values.items())
        return f"cov_func({self.func.__name__}, {args})"


def cov_func(func, *args, **kwargs):
    """cov_func(func, ...) (R/GPRclass.R:424-427): a covariance function with fixed parameters."""
    if not isinstance(func, KernelGeneric):
        raise TypeError("cov_func: `func` must be one of constant, linear, polynomial, sqrexp, gammaexp, "
                        "rationalquadratic on the MI355X path (arbitrary R closures stay on the reference's R path)")
    return CovFunc(func, func.bind(args, kwargs))


def require_tagged(k, who):
    if not callable(k):
        raise TypeError("is.function(k) is not TRUE")
    if not hasattr(k, "gprc_kernel"):
        raise TypeError(f"{who}: covariance function is not a cov_func() of a gprc kernel; untagged closures are "
                        "evaluated by the reference's R code path only -- there is no CPU fallback here")
    return k


def covariance_matrix(A, B, covariance_function, ctx=None):
    """covariance_matrix(A, B, k) (R/GPRclass.R:355-357): ncol(A) x ncol(B), [i,j] = k(A[,i], B[,j])."""
    k = require_tagged(covariance_function, "covariance_matrix")
    A, B = as_points(A, what="A"), as_points(B, what="B")
    if A.shape[0] != B.shape[0]:
        raise ValueError("A and B must have the same number of rows")
    d, nA = A.shape
    nB = B.shape[1]
    out = np.empty((nA, nB), order="F")
    if nA == 0 or nB == 0:
        return out
    ctx = ctx or nat.default_context()
    _, pp, npar = nat.params_array(k.native_params(d))
    nat.check(nat.lib().gprc_kernel_matrix(ctx.handle, k.gprc_kernel[0], pp, npar, A.ctypes.data, d, nA, B.ctypes.data, nB,
                                            out.ctypes.data, nA))
    return out
