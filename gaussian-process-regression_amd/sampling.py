"""multivariate_normal() and the numeric half of the posterior-draw plots (reference
R/GPRclass.R:360-376, :190-199).

The factorisation of the covariance -- t(chol(.)), or eigen() when the Cholesky fails, which for a posterior covariance
is the normal case -- and `mean + L %*% Z` run on the MI355X (gprc_mvn_sample).  The standard normal matrix Z is drawn
on the host, as R's rnorm is; pass `z` to supply it, or `rng` (a numpy Generator) to control it.  R's Mersenne-Twister
stream is not reproduced (parity of the draws themselves is unpinned and, through eigen()'s sign freedom, not even
defined); what is pinned is L %*% t(L) = covariance (projected on the PSD cone) and the reference's acceptance rule.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as nat

__all__ = ["multivariate_normal", "expand_range", "mvn_factor", "sym_eigen"]


def _sq(cov, what="covariance"):
    cov = np.asfortranarray(np.asarray(cov, dtype=np.float64))
    if cov.ndim != 2 or cov.shape[0] != cov.shape[1]:
        raise ValueError(f"{what} must be a square matrix")
    return cov


def sym_eigen(A, *, vectors=True, ctx=None):
    """eigen(A, symmetric = TRUE): (values in decreasing order, vectors as columns or None).  Lower triangle read."""
    A = _sq(A, "A")
    m = A.shape[0]
    ctx = ctx or nat.default_context()
    val = np.empty(m)
    vec = np.empty((m, m), order="F") if vectors else None
    sweeps = C.c_int()
    nat.check(nat.lib().gprc_sym_eigen(ctx.handle, A.ctypes.data, m, m, val.ctypes.data, vec.ctypes.data if vectors else None,
                                       C.byref(sweeps)))
    return val, vec


def mvn_factor(covariance, tol=1e-6, *, ctx=None):
    """(L, method): L = t(chol(covariance)) [method "chol"] or eigen$vectors %*% diag(sqrt(pmax(eigen$values, 0)))
    [method "eigen"]  --  R/GPRclass.R:362-368.  Raises ValueError when an eigenvalue is below -tol * |largest| (:366)."""
    cov = _sq(covariance)
    m = cov.shape[0]
    ctx = ctx or nat.default_context()
    L = np.empty((m, m), order="F")
    method = C.c_int()
    try:
        nat.check(nat.lib().gprc_mvn_factor(ctx.handle, cov.ctypes.data, m, m, float(tol), L.ctypes.data, C.byref(method)))
    except nat.GprcError as e:
        if e.status == nat.ERR_NOT_PD:
            raise ValueError("all(eigval > -tol * abs(eigval[1])) is not TRUE") from e
        raise
    return L, {1: "chol", 2: "eigen"}[method.value]


def multivariate_normal(n, mean, covariance, tol=1e-6, *, z=None, rng=None, ctx=None):
    """multivariate_normal(n, mean, covariance, tol = 1e-6)  --  R/GPRclass.R:360-370.
    Returns a length(mean) x n array, one draw per column: drop(mean) + L %*% matrix(rnorm(n * length(mean)), nrow = length(mean))."""
    cov = _sq(covariance)
    mean = np.ascontiguousarray(np.asarray(mean, dtype=np.float64).ravel())
    m = mean.size
    if m != cov.shape[0]:
        raise ValueError("length(mean) == nrow(covariance) is not TRUE")               # :361
    n = int(n)
    if z is None:
        rng = rng if rng is not None else np.random.default_rng()
        z = rng.standard_normal((n, m)).T                                              # column-major m x n fill order, as matrix(rnorm(.), nrow = m)
    z = np.asfortranarray(np.asarray(z, dtype=np.float64).reshape(m, n))
    ctx = ctx or nat.default_context()
    out = np.empty((m, n), order="F")
    method = C.c_int()
    try:
        nat.check(nat.lib().gprc_mvn_sample(ctx.handle, cov.ctypes.data, m, m, mean.ctypes.data, float(tol), z.ctypes.data, n,
                                            out.ctypes.data, C.byref(method)))
    except nat.GprcError as e:
        if e.status == nat.ERR_NOT_PD:
            raise ValueError("all(eigval > -tol * abs(eigval[1])) is not TRUE") from e
        raise
    return out


def expand_range(x):
    """expand_range(x)  --  R/GPRclass.R:372-376: the range of x widened by 20 % about its midpoint."""
    x = np.asarray(x, dtype=np.float64)
    lo, hi = float(x.min()), float(x.max())
    mid = (lo + hi) / 2.0
    return mid - 1.2 * (mid - lo), mid + 1.2 * (hi - mid)
