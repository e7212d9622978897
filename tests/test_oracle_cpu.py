"""The CPU oracle against (1) the reference's own closed-form known answers
(tests/testthat/test-gpr.R:6-27), (2) the committed golden vectors of the independent numpy/scipy
restatement, (3) itself (blocked tier vs textbook tier)."""
import math

import numpy as np
import pytest

from conftest import TOL, nerr, oracle_params
from oracle import oracle as orc


def _run(kid, par, X, y, noise, xs):
    f = orc.gpr_fit(kid, par, np.array(X, float), np.array(y, float), noise)
    m, v = orc.gpr_predict(kid, par, np.array(X, float), f["L"], f["alpha"], np.array(xs, float))
    return m, v


def test_reference_known_answers():
    # expect_equivalent tolerance in the reference is 1.5e-8; the oracle meets them to rounding
    e = math.exp
    m, v = _run(orc.POLYNOMIAL, [0.25, 1], [[-0.5, 0.5]], [4, 4], 0.5, [0])          # test-gpr.R:6-9
    assert abs(m[0] - 2) < 1e-14 and abs(v[0] - 1 / 8) < 1e-14
    m, v = _run(orc.CONSTANT, [1], [[1, 2]], [1, 3], 1, [3])                          # :12-15
    assert abs(m[0] - 4 / 3) < 1e-14 and abs(v[0] - 1 / 3) < 1e-14
    m, v = _run(orc.CONSTANT, [1], [[100, 54]], [5, 0], 1, [math.pi])                 # :16-19
    assert abs(m[0] - 5 / 3) < 1e-14 and abs(v[0] - 1 / 3) < 1e-14
    m, v = _run(orc.SQREXP, [1], [[1, 2]], [0, 1], 1, [0])                            # :23-27
    assert abs(m[0] - (2 * e(-2) - e(-1)) / (4 - e(-1))) < 1e-15
    assert abs(v[0] - (1 - (2 * e(-1) - 2 * e(-3) + 2 * e(-4)) / (4 - e(-1)))) < 1e-15


def test_golden_closed_forms(golden):
    for c in golden.of_type("gpr_closed_form"):
        kid = orc.KERNEL_IDS[c["kernel"]]
        par = oracle_params(c["kernel"], c["params"])
        m, v = _run(kid, par, golden.get(c, "X"), golden.get(c, "y"), c["noise"], golden.get(c, "Xs"))
        assert abs(m[0] - golden.get(c, "mean")[0]) < 1e-14, c["name"]
        assert abs(v[0] - golden.get(c, "var")[0]) < 1e-14, c["name"]


def test_golden_gpr(golden):
    cases = golden.of_type("gpr")
    assert len(cases) >= 40
    for c in cases:
        kid = orc.KERNEL_IDS[c["kernel"]]
        par = oracle_params(c["kernel"], c["params"])
        X, y, Xs = golden.get(c, "X"), golden.get(c, "y"), golden.get(c, "Xs")
        f = orc.gpr_fit(kid, par, X, y, c["noise"])
        assert f["noise"] == golden.get(c, "noise")[0], c["name"]
        assert f["attempts"] == int(golden.get(c, "attempts")[0]), c["name"]
        assert nerr(f["alpha"], golden.get(c, "alpha")) <= TOL, c["name"]
        assert nerr([f["logp"]], golden.get(c, "logp")) <= TOL, c["name"]
        assert nerr(np.diag(f["L"]), golden.get(c, "diagL")) <= TOL, c["name"]
        if "L" in c["outputs"]:
            assert nerr(f["L"], golden.get(c, "L")) <= TOL, c["name"]
            assert nerr(orc.kernel_matrix(kid, par, X, X), golden.get(c, "K")) <= 1e-14, c["name"]
        m, v = orc.gpr_predict(kid, par, X, f["L"], f["alpha"], Xs)
        assert nerr(m, golden.get(c, "mean")) <= TOL, c["name"]
        assert nerr(v, golden.get(c, "var")) <= TOL, c["name"]
        if "cov" in c["outputs"]:
            m2, cov = orc.gpr_predict(kid, par, X, f["L"], f["alpha"], Xs, pointwise=False)
            assert nerr(cov, golden.get(c, "cov")) <= TOL, c["name"]
            assert nerr(m2, m) == 0.0


def test_jitter_sequence(golden):
    # exactly singular (duplicate points, noise 0): WHICH pivot trips first depends on summation order
    # (LAPACK: 3, textbook recurrence: 6) so only the robust facts are pinned: the first attempt fails and
    # noise + 0.01 succeeds (R/GPRclass.R:141-148)
    c = [c for c in golden.cases if c["name"] == "gpr_jitter_duplicates"][0]
    f = orc.gpr_fit(orc.SQREXP, [1.0], golden.get(c, "X"), golden.get(c, "y"), 0.0)
    assert f["info_first"] > 0 and f["attempts"] == 2 and f["noise"] == 0.01
    # robustly indefinite: first pivot negative until the 4th attempt (noise 0, 0.01, 0.02, 0.03)
    c = [c for c in golden.cases if c["name"] == "gpr_jitter_linear_negative"][0]
    f = orc.gpr_fit(orc.LINEAR, [-1.0], golden.get(c, "X"), golden.get(c, "y"), 0.0)
    assert f["info_first"] == 1 == int(golden.get(c, "info_first")[0])
    assert f["attempts"] == 4 == int(golden.get(c, "attempts")[0]) and f["noise"] == 0.03 == golden.get(c, "noise")[0]


def test_all_attempts_fail(golden):
    c = golden.of_type("gpr_notpd")[0]
    X, y = golden.get(c, "X"), golden.get(c, "y")
    K = orc.kernel_matrix(orc.POLYNOMIAL, [-1.0, 1.0], X, X)
    _, info = orc.potrf_lower(K)
    assert info == 2 == int(golden.get(c, "info_first")[0])   # LAPACK info: second leading minor
    with pytest.raises(ArithmeticError):                      # all ten jitter steps fail -> the reference stop()s
        orc.gpr_fit(orc.POLYNOMIAL, [-1.0, 1.0], X, y, 0.0)


def test_golden_gpc(golden):
    for c in golden.of_type("gpc"):
        par = oracle_params(c["kernel"], c["params"])
        X, y, Xs = golden.get(c, "X"), golden.get(c, "y"), golden.get(c, "Xs")
        g = orc.gpc_fit(orc.SQREXP, par, X, y, c["epsilon"])
        assert g["iters"] == int(golden.get(c, "iters")[0]), c["name"]
        assert nerr(g["f_hat"], golden.get(c, "f_hat")) <= 1e-9, c["name"]
        assert nerr([g["logq"]], golden.get(c, "logq")) <= 1e-9, c["name"]
        fs, vf = orc.gpc_predict_latent(orc.SQREXP, par, X, y, g["f_hat"], g["L"], Xs)
        assert nerr(fs, golden.get(c, "fs_bar")) <= 1e-9, c["name"]
        assert nerr(vf, golden.get(c, "Vfs")) <= 1e-9, c["name"]


def test_reference_gpc_sign_expectations(golden):
    # tests/testthat/test-gpc.R:9-10,17-18,26-27: P(class +1) < 0.5 at the first probe, > 0.5 at the second
    for name in ("gpc_ref_step", "gpc_ref_unbalanced", "gpc_ref_raster"):
        c = [c for c in golden.cases if c["name"] == name][0]
        prob = golden.get(c, "prob")
        assert prob[0] < 0.5 < prob[1], name


def test_blocked_tier_matches_textbook():
    rng = np.random.default_rng(3)
    d, n, ns = 8, 700, 90
    X = rng.uniform(-1, 1, (d, n))
    y = rng.normal(size=n)
    Xs = rng.uniform(-1, 1, (d, ns))
    f = orc.gpr_fit(orc.RATQUAD, [1.0, 1.5], X, y, 0.1)
    m, v = orc.gpr_predict(orc.RATQUAD, [1.0, 1.5], X, f["L"], f["alpha"], Xs)
    b = orc.gpr_fit_predict_blocked(orc.RATQUAD, [1.0, 1.5], X, y, 0.1, Xs)
    assert b["info"] == 0
    assert nerr(b["L"], f["L"]) <= 1e-12 and nerr(b["alpha"], f["alpha"]) <= 1e-11
    assert nerr(b["mean"], m) <= 1e-11 and nerr(b["var"], v) <= 1e-11
    assert abs(b["logp"] - f["logp"]) <= 1e-9 * abs(f["logp"])


def test_potrf_info_matches_lapack():
    import scipy.linalg as sl
    rng = np.random.default_rng(5)
    A = rng.normal(size=(40, 40))
    A = A @ A.T + 40 * np.eye(40)
    A[25, 25] = -1.0  # leading minor of order 26 is not PD
    for blocked in (False, True):
        _, info = orc.potrf_lower(A, blocked=blocked)
        assert info == 26
    _, info = sl.lapack.dpotrf(A, lower=1)
    assert info == 26


def test_kernel_colwise_matches_matrix_diagonal():
    rng = np.random.default_rng(9)
    A = rng.uniform(-1, 1, (3, 11))
    B = rng.uniform(-1, 1, (3, 11))
    for kid, par in [(orc.CONSTANT, [2.0]), (orc.LINEAR, [0.3, 0.6, 0.9]), (orc.POLYNOMIAL, [0.5, 2.0]), (orc.SQREXP, [0.7]),
                     (orc.GAMMAEXP, [0.7, 1.3]), (orc.RATQUAD, [0.7, 2.5])]:
        assert np.array_equal(orc.kernel_colwise(kid, par, A, B), np.diag(orc.kernel_matrix(kid, par, A, B)))


def test_fit_gradient_restatement_against_literal_numpy():
    """oracle_fit_gradient (R/fit.R:126-139) against a line-by-line numpy transcription of the same R expressions
    (np.linalg.inv for solve).  Unpinned by the reference: it holds no numbers for fit()."""
    rng = np.random.default_rng(4)
    d, n = 2, 30
    X = rng.uniform(-3, 3, (d, n))
    y = rng.normal(size=n)

    def literal(kid, v, deriv):
        K = orc.kernel_matrix(kid, v, X, X)                                             # :132
        Kd = np.array([[deriv(X[:, i], X[:, j], *v) for j in range(n)] for i in range(n)]).reshape(n, n, len(v))  # :133
        Ki = np.linalg.inv(K)                                                           # :136
        al = Ki @ y                                                                     # :137
        return np.array([0.5 * np.sum(np.diag(np.outer(al, al) - Ki) @ Kd[:, :, i]) for i in range(len(v))])  # :138

    def d_sqrexp(x, yy, l):                                                             # R/fit.R:4-7
        r = np.sqrt(np.sum((x - yy) ** 2))
        return [r ** 2 / l ** 3 * np.exp(-r ** 2 / (l ** 2 * 2))]

    def d_ratquad(x, yy, alpha, l):                                                     # R/fit.R:25-31
        r = np.sum((x - yy) ** 2)
        q = r / (2 * l ** 2 * alpha) + 1
        return [(q ** (-alpha) * (r - (2 * l ** 2 * alpha + r) * np.log(q))) / (2 * l ** 2 * alpha + r),
                (r * q ** (-alpha - 1)) / l ** 3]

    def d_poly(x, yy, sigma, p):                                                        # R/fit.R:20-23
        s = x @ yy + sigma
        return [p * s ** (p - 1), s ** p * np.log(s)]

    for kid, v, dv in [(orc.SQREXP, [0.5], d_sqrexp), (orc.RATQUAD, [0.7, 1.3], d_ratquad), (orc.RATQUAD, [1.0, 1.0], d_ratquad)]:
        got, ref = orc.fit_gradient(kid, v, X, y), literal(kid, v, dv)
        assert np.max(np.abs(got - ref)) <= 1e-9 * np.max(np.abs(ref)), (kid, v)
    Xp = np.array([[0.2, 0.9, 1.7]])                                                    # polynomial: K has rank p + 1, keep n = 3
    yp = np.array([1.0, -0.5, 0.3])
    X, y, n = Xp, yp, 3
    got, ref = orc.fit_gradient(orc.POLYNOMIAL, [0.5, 2.0], Xp, yp), literal(orc.POLYNOMIAL, [0.5, 2.0], d_poly)
    assert np.max(np.abs(got - ref)) <= 1e-7 * np.max(np.abs(ref))
    # gammaexp: -exp(.) * (r/l)^gamma * log(r/l) is 0 * -Inf = NaN at r = 0, i.e. on every diagonal entry (R/fit.R:12)
    g = orc.fit_gradient(orc.GAMMAEXP, [1.0, 1.0], rng.uniform(-3, 3, (2, 10)), rng.normal(size=10))
    assert math.isnan(g[0]) and math.isfinite(g[1])


def test_sampling_restatement():
    """oracle_sym_eigen / oracle_mvn_factor (R/GPRclass.R:360-370) against numpy's LAPACK: eigenvalues, the
    reconstruction, L L^T = covariance on both branches, and the acceptance rule :366."""
    rng = np.random.default_rng(12)
    for m, r in [(1, 1), (2, 2), (31, 7), (64, 64)]:
        B = rng.normal(size=(m, r))
        A = B @ B.T
        val, vec = orc.sym_eigen(A)
        w = np.linalg.eigvalsh(A)[::-1]
        assert np.max(np.abs(val - w)) <= 1e-12 * max(w[0], 1e-300)
        assert np.max(np.abs(vec @ np.diag(val) @ vec.T - A)) <= 1e-12 * w[0] and np.max(np.abs(vec.T @ vec - np.eye(m))) <= 1e-12
        L, method = orc.mvn_factor(A)
        assert method == (1 if r == m else 2)
        assert np.max(np.abs(L @ L.T - A)) <= 1e-12 * w[0]
        if method == 1:
            assert np.max(np.abs(L - np.linalg.cholesky(A))) <= 1e-12 * math.sqrt(w[0]) and np.all(np.triu(L, 1) == 0)
    A = np.diag([2.0, 1.0, -1e-3])                         # eigenvalue below -tol * |largest|: stopifnot fails
    with pytest.raises(ArithmeticError):
        orc.mvn_factor(A, 1e-6)
    L, method = orc.mvn_factor(np.diag([2.0, 1.0, -1e-9]), 1e-6)   # within tol: clipped to zero (pmax)
    assert method == 2 and np.max(np.abs(L @ L.T - np.diag([2.0, 1.0, 0.0]))) <= 1e-15
    Z = rng.normal(size=(3, 4))
    out, _ = orc.multivariate_normal([1.0, 2.0, 3.0], np.diag([4.0, 9.0, 16.0]), Z)
    assert np.allclose(out, np.array([[1.0], [2.0], [3.0]]) + np.diag([2.0, 3.0, 4.0]) @ Z, rtol=0, atol=1e-15)


def test_fit_record_is_what_the_host_driver_yields_on_the_oracle():
    """tests/golden/fit_six_kernels.json (the recorded outcome of tests/testthat/test-fit.R:12-17 with the full six-kernel
    list) against a fresh run of tests/golden/make_fit_record.py's recipe: the package's Brent / vmmin /
    optim_until_error driver on the oracle's dens and dens_deriv.  Pins the host driver and the record to each other on
    the CPU; tests/test_gpu_fit.py then holds the native objective to the same record."""
    import importlib.util
    import json
    import os
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("make_fit_record", os.path.join(ROOT, "tests", "golden", "make_fit_record.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rec = json.load(open(os.path.join(ROOT, "tests", "golden", "fit_six_kernels.json")))
    x, ys = mod.targets()
    with mod.oracle_backed_fit() as fit:
        for case, y in zip(rec["cases"], ys):
            r = fit(x.reshape(1, -1), y, rec["noise"], rec["cov_names"])
            assert r["cov"] == case["winner"] and case["holds"] == (case["winner"] == case["expected_by_test_fit_R"])
            assert np.allclose(r["par"], case["par"], rtol=1e-9, atol=1e-12)
            assert np.allclose(r["score"], [case["score"][nm] for nm in rec["cov_names"]], rtol=1e-10, atol=1e-10)
    assert [c["holds"] for c in rec["cases"]] == [True, True, True, False, False, False]


def test_oracle_is_clean_under_address_and_ub_sanitizers(tmp_path):
    """tests/oracle_sanitize_driver.c: the oracle's translation unit compiled with -fsanitize=address,undefined
    (-fno-sanitize-recover) and driven through the closed-form case, a ragged blocked fit + predict, the jitter loop, a GPC fit
    and the eigen helpers.  GPU sanitizers are not available on this pool, so the checker itself is what gets sanitised
    (SURVEY section 5)."""
    import os
    import subprocess
    from conftest import ROOT
    exe = str(tmp_path / "oracle_sanitize")
    subprocess.check_call(["gcc", "-std=gnu11", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fopenmp",
                           "-ffp-contract=off", "-Wall", "-o", exe, os.path.join(ROOT, "tests", "oracle_sanitize_driver.c"), "-lm"])
    r = subprocess.run([exe], env=dict(os.environ, OMP_NUM_THREADS="2", ASAN_OPTIONS="detect_leaks=1"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "oracle_sanitize_driver: ok" in r.stdout, r.stdout + r.stderr[-3000:]
