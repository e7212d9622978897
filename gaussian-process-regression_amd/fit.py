"""fit(): hyper-parameter selection by maximising the log marginal likelihood (reference R/fit.R:110-169) --
the immediate caller of the hot path.

What runs where
  * the objective `dens(v)` (R/fit.R:117-124: K, Cholesky, alpha, log marginal likelihood) is one native call,
    gprc_gpr_log_marginal; the reference's O(n^4) `min(det(leading minors)) > 0` guard (:119) is the Cholesky's
    own positive-definiteness test;
  * the optimiser driver is host code, as in the reference.  R's `optim(method = "Brent")` is `optimize()`, Brent's
    fmin; it is restated here (golden section + successive parabolic interpolation, tol = sqrt(.Machine$double.eps))
    so that the one-parameter kernels (sqrexp, constant, linear) and the polynomial degree loop follow the same
    iterates.  `optim_until_error` (R/fit.R:47-69) is kept: a failing evaluation yields the sentinel -10000.
  * the two-parameter kernels gammaexp / rationalquadratic go through `optim(method = "BFGS")` with the gradient
    `dens_deriv` (R/fit.R:126-139).  That gradient is reproduced AS WRITTEN, by the native gprc_fit_gradient: it
    inverts the noise-free K, ignores `noise`, applies `diag(.) %*%` where a trace is meant and binds the parameter
    vector in `deriv`'s argument order, which is the reverse of the kernels' own.  R's BFGS is `vmmin` (Nash's
    variable-metric Algorithm 21); it is restated below.  Consequences that follow from the reference's code and are
    kept: gammaexp's gradient has a NaN component (0 * log 0 on the diagonal), vmmin then takes its "uphill" exit at
    once and fit() returns the start values (1, 1); a failing gradient (K not invertible) aborts optim and
    optim_until_error falls back to the best objective value seen so far.
Parity status: unpinned (the reference holds no numeric expectations for fit(); tests/testthat/test-fit.R:12-17 only
checks which kernel NAME wins); cross-checked against scipy's bounded Brent on the same native objective, and the
BFGS branch against the same driver running on the CPU oracle's objective and gradient.
"""
from __future__ import annotations

import ctypes as C
import math
import sys

import numpy as np

from . import _native as nat
from .covfunc import CovFunc, as_points, constant, linear, polynomial, sqrexp, gammaexp, rationalquadratic

__all__ = ["fit", "dens", "dens_deriv", "cov_dict", "brent_fmin", "vmmin"]

# R/fit.R:2-33: name -> (kernel generic, display name, start values)
cov_dict = {
    "sqrexp": (sqrexp, "Squared Exponential", (1.0,)),
    "gammaexp": (gammaexp, "Gamma Exponential", (1.0, 1.0)),
    "constant": (constant, "Constant", (1.0,)),
    "linear": (linear, "Linear", (1.0,)),
    "polynomial": (polynomial, "Polynomial", (1.0, 2.0)),
    "rationalquadratic": (rationalquadratic, "Rational Quadratic", (1.0, 1.0)),
}
SENTINEL = -10000.0  # R/fit.R:50


def dens(X, y, noise, name, v, ctx=None):
    """dens(v) of R/fit.R:117-124 for kernel `name` with parameter vector v (in the generic's argument order).
    Raises nat.NotPositiveDefinite when K + noise*I is not positive definite (the reference's stopifnot / chol error)."""
    func = cov_dict[name][0]
    Xm = as_points(X)
    y = np.ascontiguousarray(np.asarray(y, dtype=np.float64))
    d, n = Xm.shape
    ctx = ctx or nat.default_context()
    _, pp, npar = nat.params_array(np.atleast_1d(np.asarray(v, dtype=np.float64)))
    out = C.c_double()
    nat.check(nat.lib().gprc_gpr_log_marginal(ctx.handle, func.kernel_id, pp, npar, Xm.ctypes.data, d, n, y.ctypes.data,
                                              float(noise), C.byref(out)))
    return out.value


def dens_deriv(X, y, name, v, ctx=None):
    """dens_deriv(v) of R/fit.R:126-139 (quirks included, see the module docstring) for kernel `name`.
    Raises nat.NotPositiveDefinite when the noise-free K cannot be inverted (the reference's solve(K) error)."""
    func = cov_dict[name][0]
    Xm = as_points(X)
    y = np.ascontiguousarray(np.asarray(y, dtype=np.float64))
    d, n = Xm.shape
    ctx = ctx or nat.default_context()
    p, pp, npar = nat.params_array(np.atleast_1d(np.asarray(v, dtype=np.float64)))
    g = np.empty(p.size)
    nat.check(nat.lib().gprc_fit_gradient(ctx.handle, func.kernel_id, pp, npar, Xm.ctypes.data, d, n, y.ctypes.data,
                                          g.ctypes.data_as(C.POINTER(C.c_double))))
    return g


def vmmin(b0, fn, gr, maxit=100, abstol=-math.inf, reltol=math.sqrt(np.finfo(float).eps)):
    """R's optim(method = "BFGS") core: Nash's variable-metric minimiser (Compact Numerical Methods, Algorithm 21)
    with R's constants (stepredn 0.2, acctol 1e-4, reltest 10).  Minimises fn; returns (par, value, fncount, grcount,
    fail).  Restated from the published algorithm; control flow details that matter for the reference's behaviour:
    a non-finite initial value is an error; a search direction that is not downhill (including a NaN gradient
    projection) resets B once and then terminates."""
    stepredn, acctol, reltest = 0.2, 1e-4, 10.0
    b = np.array(b0, dtype=np.float64)
    n = b.size
    if maxit <= 0:
        return b, float(fn(b)), 0, 0, 0
    f = float(fn(b))
    if not math.isfinite(f):
        raise ArithmeticError("initial value in 'vmmin' is not finite")
    fmin = f
    funcount = gradcount = 1
    g = np.array(gr(b), dtype=np.float64)
    it = 1
    ilast = gradcount
    B = np.zeros((n, n))
    count = 0
    while True:
        if ilast == gradcount:
            B = np.eye(n)
        X = b.copy()
        c = g.copy()
        t = np.empty(n)
        gradproj = 0.0
        for i in range(n):
            sacc = 0.0
            for j in range(i + 1):
                sacc -= B[i, j] * g[j]
            for j in range(i + 1, n):
                sacc -= B[j, i] * g[j]
            t[i] = sacc
            gradproj += sacc * g[i]
        if gradproj < 0.0:  # downhill
            steplength = 1.0
            accpoint = False
            while True:
                count = 0
                for i in range(n):
                    b[i] = X[i] + steplength * t[i]
                    if reltest + X[i] == reltest + b[i]:
                        count += 1
                if count < n:
                    f = float(fn(b))
                    funcount += 1
                    accpoint = math.isfinite(f) and f <= fmin + gradproj * steplength * acctol
                    if not accpoint:
                        steplength *= stepredn
                if count == n or accpoint:
                    break
            enough = f > abstol and abs(f - fmin) > reltol * (abs(fmin) + reltol)
            if not enough:
                count = n
                fmin = f
            if count < n:  # making progress
                fmin = f
                g = np.array(gr(b), dtype=np.float64)
                gradcount += 1
                it += 1
                D1 = 0.0
                for i in range(n):
                    t[i] = steplength * t[i]
                    c[i] = g[i] - c[i]
                    D1 += t[i] * c[i]
                if D1 > 0:
                    D2 = 0.0
                    for i in range(n):
                        sacc = 0.0
                        for j in range(i + 1):
                            sacc += B[i, j] * c[j]
                        for j in range(i + 1, n):
                            sacc += B[j, i] * c[j]
                        X[i] = sacc
                        D2 += sacc * c[i]
                    D2 = 1.0 + D2 / D1
                    for i in range(n):
                        for j in range(i + 1):
                            B[i, j] += (D2 * t[i] * t[j] - X[i] * t[j] - t[i] * X[j]) / D1
                else:
                    ilast = gradcount
            else:  # no progress
                if ilast < gradcount:
                    count = 0
                    ilast = gradcount
        else:  # uphill search direction (or NaN): reset B, or stop if it has just been reset
            count = 0
            if ilast == gradcount:
                count = n
            else:
                ilast = gradcount
        if it >= maxit:
            break
        if gradcount - ilast > 2 * n:
            ilast = gradcount  # periodic restart
        if not (count != n or ilast != gradcount):
            break
    return b, fmin, funcount, gradcount, (0 if it < maxit else 1)


def brent_fmin(f, ax, bx, tol):
    """Brent's fmin, the algorithm behind R's optimize() / optim(method = "Brent"): minimum of f on [ax, bx].
    Restated from the published algorithm (Brent 1973, ch. 5; netlib fmin)."""
    c = (3.0 - math.sqrt(5.0)) * 0.5
    eps = math.sqrt(np.finfo(float).eps)
    a, b = ax, bx
    v = a + c * (b - a)
    w = x = v
    d = e = 0.0
    fx = f(x)
    fv = fw = fx
    tol3 = tol / 3.0
    while True:
        xm = (a + b) * 0.5
        tol1 = eps * abs(x) + tol3
        t2 = tol1 * 2.0
        if abs(x - xm) <= t2 - (b - a) * 0.5:
            break
        p = q = r = 0.0
        if abs(e) > tol1:  # fit a parabola
            r = (x - w) * (fx - fv)
            q = (x - v) * (fx - fw)
            p = (x - v) * q - (x - w) * r
            q = (q - r) * 2.0
            if q > 0.0:
                p = -p
            else:
                q = -q
            r = e
            e = d
        if abs(p) >= abs(q * 0.5 * r) or p <= q * (a - x) or p >= q * (b - x):  # golden-section step
            e = (b - x) if x < xm else (a - x)
            d = c * e
        else:  # parabolic-interpolation step
            d = p / q
            u = x + d
            if u - a < t2 or b - u < t2:  # f must not be evaluated too close to a or b
                d = tol1 if x < xm else -tol1
        if abs(d) >= tol1:  # f must not be evaluated too close to x
            u = x + d
        elif d > 0.0:
            u = x + tol1
        else:
            u = x - tol1
        fu = f(u)
        if fu <= fx:
            if u < x:
                b = x
            else:
                a = x
            v, w, x = w, x, u
            fv, fw, fx = fw, fx, fu
        else:
            if u < x:
                a = u
            else:
                b = u
            if fu <= fw or w == x:
                v, fv = w, fw
                w, fw = u, fu
            elif fu <= fv or v == x or v == w:
                v, fv = u, fu
    return x


def _optim_brent_until_error(f, lower, upper):
    """optim_until_error(start, f, method = "Brent", lower, upper, control = list(fnscale = -1))  (R/fit.R:47-69):
    failing evaluations return the sentinel; optimize() minimises f / fnscale = -f with tol = sqrt(eps)."""
    def f_new(par):
        try:
            return f(par)
        except (nat.NotPositiveDefinite, ArithmeticError):
            return SENTINEL
    tol = math.sqrt(np.finfo(float).eps)  # optim's default reltol
    xmin = brent_fmin(lambda p_: -f_new(p_), lower, upper, tol)
    return xmin, f_new(xmin)


def _optim_bfgs_until_error(start, f, gr):
    """optim_until_error(start, f, gr = dens_deriv, method = "BFGS", control = list(fnscale = -1))  (R/fit.R:47-69,
    :157-158).  Only f is wrapped by the sentinel (:49-55); an error inside gr aborts optim (:56), after which the best
    (par, value) among the successful evaluations so far is returned (:63-65), or f(start) if there was none (:58-61).
    `f`/`gr` here take the objective/gradient failures as nat.NotPositiveDefinite / ArithmeticError."""
    seen = []  # (par, value) of every successful evaluation, in order

    def f_new(par):
        try:
            out = float(f(par))
        except (nat.NotPositiveDefinite, ArithmeticError):
            return SENTINEL
        if out != SENTINEL:
            seen.append((np.array(par, dtype=np.float64), out))
        return out

    try:  # optim minimises fn / fnscale and gr / fnscale with fnscale = -1
        par, val, *_ = vmmin(np.asarray(start, dtype=np.float64), lambda p_: -f_new(p_), lambda p_: -np.asarray(gr(p_)))
        return tuple(float(x) for x in par), -val
    except (nat.NotPositiveDefinite, ArithmeticError):
        if not seen:
            try:
                value = float(f(np.asarray(start, dtype=np.float64)))
            except (nat.NotPositiveDefinite, ArithmeticError):
                value = SENTINEL
            return tuple(float(x) for x in start), value
        best = max(range(len(seen)), key=lambda i: (seen[i][1], -i))  # which.max: the first maximum
        return tuple(float(x) for x in seen[best][0]), seen[best][1]


def fit(X, y, noise, cov_names=None, *, ctx=None):
    """fit(X, y, noise, cov_names = as.list(cov_df$name))  --  R/fit.R:110-169.
    Returns dict(par, cov, score, func); `func` is a tagged cov_func closure, ready for GPR / GPC."""
    names = list(cov_dict) if cov_names is None else list(cov_names)
    for nm in names:
        if nm not in cov_dict:
            raise KeyError(nm)
    Xm = as_points(X)
    ctx = ctx or nat.default_context()
    params, score = [], []
    for nm in names:
        if nm == "polynomial":  # R/fit.R:146-155: Brent over sigma in [0, 5] for every degree 1..10
            best = None
            for deg in range(1, 11):
                par, val = _optim_brent_until_error(lambda sig, deg=deg: dens(Xm, y, noise, nm, [sig, float(deg)], ctx), 0.0, 5.0)
                if best is None or val > best[1]:   # which.max: first maximum wins
                    best = ((par, float(deg)), val)
            params.append(best[0])
            score.append(best[1])
        elif len(cov_dict[nm][2]) == 2:  # gammaexp, rationalquadratic: BFGS with dens_deriv  (R/fit.R:125-140, 144)
            par, val = _optim_bfgs_until_error(cov_dict[nm][2], lambda v, nm=nm: dens(Xm, y, noise, nm, v, ctx),
                                               lambda v, nm=nm: dens_deriv(Xm, y, nm, v, ctx))
            params.append(par)
            score.append(val)
        else:                   # one parameter: Brent on [0, 10]  (R/fit.R:143, 157-158)
            par, val = _optim_brent_until_error(lambda v: dens(Xm, y, noise, nm, [v], ctx), 0.0, 10.0)
            params.append((par,))
            score.append(val)
    w = int(np.argmax(score))
    name, par = names[w], params[w]
    sys.stderr.write("The optimal covariance function is %s, with parameters %s\n" % (name, ", ".join("%.15g" % p for p in par)))  # :166
    func = cov_dict[name][0]
    return {"par": tuple(par), "cov": name, "score": list(score), "func": CovFunc(func, func.bind(par, {}))}
