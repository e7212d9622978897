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
