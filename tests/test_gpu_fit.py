"""fit() (reference R/fit.R:110-169): the native objective `dens` against the oracle, the reference's
own fixture tests/testthat/test-fit.R:12-17 with its full six-kernel list (recorded outcomes), and the optimum
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


def test_reference_fit_fixture_full_six_kernel_list():
    """The one fixture the reference holds for fit(): tests/testthat/test-fit.R:1-17 -- six 12-point targets, noise
    0.05, the FULL six-kernel list, and the NAME of the kernel expected to win each.  The native fit() is run on all
    six lines and must reproduce the committed record tests/golden/fit_six_kernels.json (made by
    tests/golden/make_fit_record.py: the same host driver on the CPU oracle's objective and gradient): winner,
    parameters and the whole score vector.
      :12-14 (linear, constant, polynomial)  HOLD.
      :15-17 (sqrexp, gammaexp, rationalquadratic)  DO NOT: the degree-2 / degree-4 polynomial reaches a higher log
              marginal likelihood.  That is not an optimiser artefact: the record carries, for each of these lines, the
              SUPREMUM of the expected kernel's log marginal likelihood over its valid parameter domain found by an
              unrelated optimiser, and it lies 7-12 nats BELOW the polynomial's score -- no search strategy can make
              the expected kernel win which.max(score) (R/fit.R:163).  Whether the reference's own test passes cannot
              be observed without R (its GPC tests are demonstrably stale); parity of fit() stays unpinned on these
              three lines and the outcome is recorded rather than hidden."""
    import json
    import os
    from conftest import ROOT
    rec = json.load(open(os.path.join(ROOT, "tests", "golden", "fit_six_kernels.json")))
    names = rec["cov_names"]
    assert names == ["linear", "constant", "polynomial", "sqrexp", "gammaexp", "rationalquadratic"]        # test-fit.R:12-17
    ys = [3 * x, np.full(12, 5.0), 3 * x ** 2 - 2 * x, 5 * np.exp(-x ** 2), 5 * np.exp(-x ** 5), 5 / (1 + x ** 2)]
    holds = []
    for case, y in zip(rec["cases"], ys):
        r = fit(X, y, rec["noise"], names)
        assert r["cov"] == case["winner"], case["line"]
        assert np.allclose(r["par"], case["par"], rtol=1e-6, atol=1e-9), (case["line"], r["par"], case["par"])
        for nm, sc in zip(names, r["score"]):
            assert abs(sc - case["score"][nm]) <= 1e-7 * max(1.0, abs(case["score"][nm])), (case["line"], nm, sc, case["score"][nm])
        holds.append(r["cov"] == case["expected_by_test_fit_R"])
        if not holds[-1]:
            sup = case["supremum_of_expected_kernel"]["logp"]
            assert r["cov"] == "polynomial" and sup < case["score"]["polynomial"] - 5.0     # the expected kernel cannot win
            assert case["score"][case["expected_by_test_fit_R"]] <= sup + 1e-6                # fit()'s own score respects the supremum
    assert holds == [True, True, True, False, False, False]


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


def test_fit_gradient_triangular_solve_is_bit_identical():
    """dens_deriv's diag(K^-1) (R/fit.R:131: solve(K)) comes from L^-1 computed panel-wise on rows of the identity.  The rows are zero
    left of their own column, so the solve skips those products (n^3 / 3 instead of n^3 flops); everything skipped is a product
    with an exact zero, hence the SAME BITS as the dense solve (GPRC_FITGRAD_DENSE=1, in a child process: the switch is read once).
    n = 3000: six panels, several row chunks (GPRC_CHUNK_BYTES keeps a chunk at 1024 rows) and panel groups."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import numpy as np\nimport gprc_amd\nfrom gprc_amd import dens_deriv\n"
            "rng = np.random.default_rng(8); X = rng.uniform(-1, 1, (4, 3000)); y = rng.normal(size=3000)\n"
            "for name, v in (('sqrexp', [0.35]), ('rationalquadratic', [0.4, 1.5])):\n"
            "    print('GRAD', name, dens_deriv(X, y, name, v).tobytes().hex())\n") % (ROOT,)
    outs = []
    for extra in ({}, {"GPRC_FITGRAD_DENSE": "1"}, {"GPRC_CHUNK_BYTES": str(1024 * 3072 * 8)}):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, **extra))
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([ln for ln in r.stdout.splitlines() if ln.startswith("GRAD")])
    assert len(outs[0]) == 2 and outs[0] == outs[1] == outs[2]
