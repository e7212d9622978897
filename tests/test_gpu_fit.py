"""fit() (reference R/fit.R:110-169): the native objective `dens` against the oracle, the four
reference expectations of tests/testthat/test-fit.R:12-15 that involve Brent-optimised kernels, and the optimum
against an independent optimiser on the oracle's objective.  Parity status of fit(): unpinned by the reference (it
only checks the winning kernel NAME)."""
import numpy as np
import pytest
from scipy.optimize import minimize_scalar

from conftest import nerr
from gprc_amd import GPR, NotPositiveDefinite, dens, fit
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

X = np.round(np.arange(0, 1.1001, 0.1), 10).reshape(1, -1)   # tests/testthat/test-fit.R:2
x = X[0]


def test_dens_is_the_log_marginal_likelihood():
    rng = np.random.default_rng(8)
    Xr = rng.uniform(-1, 1, (3, 150))
    yr = rng.normal(size=150)
    for name, kid, v in [("sqrexp", orc.SQREXP, [0.8]), ("constant", orc.CONSTANT, [2.0]), ("linear", orc.LINEAR, [0.7]),
                         ("polynomial", orc.POLYNOMIAL, [0.5, 3.0]), ("gammaexp", orc.GAMMAEXP, [0.9, 1.5]),
                         ("rationalquadratic", orc.RATQUAD, [1.1, 1.5])]:
        ref = orc.gpr_fit(kid, v, Xr, yr, 0.1)
        assert ref["attempts"] == 1
        assert abs(dens(Xr, yr, 0.1, name, v) - ref["logp"]) <= 1e-10 * abs(ref["logp"]), name
    with pytest.raises(NotPositiveDefinite):                       # the reference's stopifnot(min(det(minors)) > 0)
        dens(np.array([[2.0, 0.1, 0.5]]), np.array([1.0, 2.0, 3.0]), 0.0, "polynomial", [-1.0, 1.0])


def test_reference_fit_expectations_brent_kernels():
    names = ["linear", "constant", "polynomial", "sqrexp"]        # the Brent-optimised subset of test-fit.R's list
    assert fit(X, 3 * x, 0.05, names)["cov"] == "linear"                       # test-fit.R:12
    assert fit(X, np.full(12, 5.0), 0.05, names)["cov"] == "constant"          # :13
    assert fit(X, 3 * x ** 2 - 2 * x, 0.05, names)["cov"] == "polynomial"      # :14
    # :15 expects "sqrexp" for y = 5 exp(-x^2).  Following R/fit.R line by line, the degree-2 polynomial reaches a
    # HIGHER log marginal likelihood (-3.39 at sigma = 4.67) than the best sqrexp (-10.59 at l = 0.75) -- confirmed below
    # with an independent optimiser on the oracle's objective -- so which.max(score) is "polynomial".  Whether the
    # reference's own test passes cannot be checked without R (its GPC tests are stale); the winner
    # here is asserted against the independently optimised scores instead.
    y4 = 5 * np.exp(-x ** 2)
    r4 = fit(X, y4, 0.05, names)

    def best(kid, pars_of, lo, hi):
        def neg(v):
            try:
                f = orc.gpr_fit(kid, pars_of(v), X, y4, 0.05)
                return -f["logp"] if f["attempts"] == 1 else 1e4
            except ArithmeticError:
                return 1e4
        return -minimize_scalar(neg, bounds=(lo, hi), method="bounded", options={"xatol": 1e-10}).fun
    indep = {"linear": best(orc.LINEAR, lambda v: [v], 0, 10), "constant": best(orc.CONSTANT, lambda v: [v], 0, 10),
             "sqrexp": best(orc.SQREXP, lambda v: [v], 0, 10),
             "polynomial": max(best(orc.POLYNOMIAL, lambda v, p=p: [v, float(p)], 0, 5) for p in range(1, 11))}
    assert r4["cov"] == max(indep, key=indep.get) == "polynomial"
    for nm, sc in zip(names, r4["score"]):
        assert abs(sc - indep[nm]) <= 1e-6 * abs(indep[nm]), nm


def test_fit_optimum_matches_independent_optimiser():
    y = 5 * np.exp(-x ** 2)
    r = fit(X, y, 0.05, ["sqrexp"])
    assert r["cov"] == "sqrexp" and len(r["par"]) == 1 and len(r["score"]) == 1

    def neg(l):
        try:
            return -orc.gpr_fit(orc.SQREXP, [l], X, y, 0.05)["logp"] if l > 0 else 1e4
        except ArithmeticError:
            return 1e4
    ref = minimize_scalar(neg, bounds=(0, 10), method="bounded", options={"xatol": 1e-10})
    assert abs(r["par"][0] - ref.x) <= 1e-5 * ref.x and abs(r["score"][0] + ref.fun) <= 1e-9 * abs(ref.fun)
    # the returned closure is a tagged cov_func: it drives GPR directly, and GPR's default k = fit(...)$func works
    g = GPR(X, y, 0.05, r["func"])
    g2 = GPR(X, y, 0.05, cov_names=["sqrexp"])
    assert nerr(g2.alpha, g.alpha) == 0.0 and g2.k.gprc_kernel[1][0] == r["par"][0]


def test_fit_gradient_matches_the_reference_formula():
    """gprc_fit_gradient against the oracle's line-by-line restatement of R/fit.R:126-139 (noise-free K, the
    diag(.) %*% quirk, deriv's own parameter order).  The native path inverts K through its Cholesky factor, the
    reference through LU: same numbers on well-conditioned K, which is what these cases are."""
    from gprc_amd import dens_deriv
    rng = np.random.default_rng(21)
    Xr = rng.uniform(-1, 1, (3, 200))
    yr = rng.normal(size=200)
    for name, kid, v in [("sqrexp", orc.SQREXP, [0.15]), ("rationalquadratic", orc.RATQUAD, [0.2, 0.7]),
                         ("rationalquadratic", orc.RATQUAD, [0.3, 2.5])]:
        ref = orc.fit_gradient(kid, v, Xr, yr)
        got = dens_deriv(Xr, yr, name, v)
        assert got.shape == ref.shape and np.max(np.abs(got - ref)) <= 1e-9 * np.max(np.abs(ref)), (name, got, ref)
    with np.errstate(all="ignore"):
        ref = orc.fit_gradient(orc.GAMMAEXP, [0.2, 1.5], Xr, yr)
        got = dens_deriv(Xr, yr, "gammaexp", [0.2, 1.5])
    assert np.isnan(ref[0]) and np.isnan(got[0])                      # 0 * log(0) on the diagonal, R/fit.R:12
    assert abs(got[1] - ref[1]) <= 1e-9 * abs(ref[1])
    Xp, yp = np.array([[0.2, 0.9, 1.7]]), np.array([1.0, -0.5, 0.3])  # polynomial p = 2: K has rank 3, n = 3
    ref = orc.fit_gradient(orc.POLYNOMIAL, [0.5, 2.0], Xp, yp)
    got = dens_deriv(Xp, yp, "polynomial", [0.5, 2.0])
    assert np.max(np.abs(got - ref)) <= 1e-6 * np.max(np.abs(ref))   # cond(K) ~ 1e3..1e4 at n = 3
    with pytest.raises(NotPositiveDefinite):                          # duplicate points: K singular, solve(K) fails in R
        dens_deriv(np.array([[0.0, 0.0, 1.0]]), np.array([1.0, 2.0, 3.0]), "sqrexp", [1.0])
    with pytest.raises(Exception, match="sqrexp, gammaexp, polynomial, rationalquadratic"):
        dens_deriv(Xp, yp, "linear", [1.0])


def test_fit_bfgs_kernels_follow_the_reference_driver():
    """gammaexp / rationalquadratic: optim(method = "BFGS") with dens_deriv (R/fit.R:125-140,144,157-158).  The
    native fit() against the same vmmin + optim_until_error driver fed by the ORACLE's objective and gradient."""
    from gprc_amd.fit import _optim_bfgs_until_error

    def oracle_driver(kid, Xd, yd, noise):
        def dens_o(v):
            f = orc.gpr_fit(kid, list(v), Xd, yd, noise)
            if f["attempts"] != 1:
                raise ArithmeticError("not positive definite")
            return f["logp"]
        with np.errstate(all="ignore"):
            return _optim_bfgs_until_error((1.0, 1.0), dens_o, lambda v: orc.fit_gradient(kid, list(v), Xd, yd))

    for seed, d, n in [(1, 2, 40), (5, 3, 60)]:                       # the first stalls at the start, the second moves
        rng = np.random.default_rng(seed)
        Xd = rng.uniform(-3, 3, (d, n))
        yd = np.sin(Xd.sum(0)) + 0.1 * rng.normal(size=n)
        ref_par, ref_val = oracle_driver(orc.RATQUAD, Xd, yd, 0.1)
        r = fit(Xd, yd, 0.1, ["rationalquadratic"])
        assert r["cov"] == "rationalquadratic" and len(r["par"]) == 2
        assert np.allclose(r["par"], ref_par, rtol=1e-6, atol=1e-9), (seed, r["par"], ref_par)
        assert abs(r["score"][0] - ref_val) <= 1e-8 * abs(ref_val)
    # gammaexp: the NaN gradient component makes vmmin stop at once -- fit() returns the start values (1, 1)
    r = fit(Xd, yd, 0.1, ["gammaexp"])
    assert r["par"] == (1.0, 1.0)
    assert abs(r["score"][0] - orc.gpr_fit(orc.GAMMAEXP, [1.0, 1.0], Xd, yd, 0.1)["logp"]) <= 1e-10 * abs(r["score"][0])
    # the full default list (R/fit.R:110) now runs end to end and the closure drives GPR
    full = fit(Xd, yd, 0.1)
    assert full["cov"] in ("sqrexp", "gammaexp", "constant", "linear", "polynomial", "rationalquadratic") and len(full["score"]) == 6
    assert GPR(Xd, yd, 0.1, full["func"]).predict(Xd[:, :5]).shape == (5, 2)


def test_kernel_specific_constructors_default_to_fit():
    """GPR.<kernel>$new(X, y, noise) with the parameters left out: each defaults to fit(X, y, noise, "<kernel>")$par
    (R/GPRclass.R:286-345), including the as-written assignment gamma <- par[[1]], l <- par[[2]] for gammaexp."""
    y = 5 * np.exp(-x ** 2)
    g = GPR.sqrexp.new(X, y, 0.05)
    assert g.k.gprc_kernel[1][0] == fit(X, y, 0.05, ["sqrexp"])["par"][0]
    gp = GPR.polynomial.new(X, y, 0.05)
    assert tuple(gp.k.gprc_kernel[1]) == fit(X, y, 0.05, ["polynomial"])["par"]
    gc = GPR.constant.new(X, y, 0.05)
    assert gc.k.gprc_kernel[1][0] == fit(X, y, 0.05, ["constant"])["par"][0]
    gl = GPR.linear.new(X, y, 0.05)                                    # one-dimensional X: the scalar sigma passes :298
    assert gl.k.gprc_kernel[1][0] == fit(X, y, 0.05, ["linear"])["par"][0]
    gg = GPR.gammaexp.new(X, y, 0.05)                                  # fit returns the start values (1, 1) for gammaexp
    assert tuple(gg.k.gprc_kernel[1]) == (1.0, 1.0)
    gs = GPR.sqrexp.new(X, y, 0.05, l=0.5)                             # explicit parameters never trigger fit()
    assert gs.k.gprc_kernel[1][0] == 0.5
    with pytest.raises(ValueError, match="length\\(sigma\\) == nrow\\(X\\)"):
        GPR.linear.new(np.vstack([x, x ** 2]), y, 0.05)                # d = 2: the fitted scalar sigma fails :298, as in R
