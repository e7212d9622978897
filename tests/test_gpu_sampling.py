"""Sampling: multivariate_normal of R/GPRclass.R:360-370 on the device -- Cholesky branch, the
eigen() fallback (cyclic Jacobi), the acceptance rule, and the data behind plot_posterior_draws / _variance (:190-221).
The draws for a given Z are compared exactly on the Cholesky branch; on the eigen branch eigenvector signs are free
(in LAPACK and R too), so the comparison is on L L^T, i.e. on the distribution."""
import numpy as np
import pytest

from conftest import nerr
from gprc_amd import GPR, cov_func, sqrexp, multivariate_normal, mvn_factor, sym_eigen, expand_range
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def test_sym_eigen_matches_lapack():
    rng = np.random.default_rng(3)
    for m, r in [(1, 1), (2, 2), (5, 5), (64, 64), (129, 20), (300, 300)]:   # odd m exercises the dummy player
        B = rng.normal(size=(m, r))
        A = B @ B.T
        val, vec = sym_eigen(A)
        w = np.linalg.eigvalsh(A)[::-1]
        assert np.all(np.diff(val) <= 0)                                    # decreasing, as eigen() returns them
        assert np.max(np.abs(val - w)) <= 1e-12 * w[0], m
        assert np.max(np.abs(vec @ np.diag(val) @ vec.T - A)) <= 1e-12 * w[0], m
        assert np.max(np.abs(vec.T @ vec - np.eye(m))) <= 1e-12, m
        assert np.max(np.abs(val - orc.sym_eigen(A)[0])) <= 1e-12 * w[0]
    A = rng.normal(size=(40, 40))
    lower = np.tril(A) + np.tril(A, -1).T                                   # only the lower triangle is read
    assert np.max(np.abs(sym_eigen(A, vectors=False)[0] - np.linalg.eigvalsh(lower)[::-1])) <= 1e-12 * np.abs(lower).max() * 40


def test_mvn_cholesky_branch_is_exact():
    rng = np.random.default_rng(5)
    m, n = 200, 7
    B = rng.normal(size=(m, m))
    cov = B @ B.T + m * np.eye(m)
    mean = rng.normal(size=m)
    Z = rng.normal(size=(m, n))
    L, method = mvn_factor(cov)
    Lo, mo = orc.mvn_factor(cov)
    assert method == "chol" and mo == 1 and np.all(np.triu(L, 1) == 0)
    assert nerr(L, Lo) <= 1e-12
    out = multivariate_normal(n, mean, cov, z=Z)
    ref, _ = orc.multivariate_normal(mean, cov, Z)
    assert out.shape == (m, n) and nerr(out, ref) <= 1e-12
    # the reference's own use: two clouds around (+-0.5, +-0.5) with covariance diag(0.1, 0.1)  (tests/testthat/test-gpc.R:31)
    z2 = rng.normal(size=(2, 50))
    pts = multivariate_normal(50, [0.5, 0.5], np.diag([0.1, 0.1]), z=z2)
    assert np.allclose(pts, 0.5 + np.sqrt(0.1) * z2, rtol=0, atol=1e-15)
    with pytest.raises(ValueError, match="length\\(mean\\)"):
        multivariate_normal(1, [0.0, 0.0, 0.0], np.eye(2))


def test_mvn_eigen_branch_and_acceptance_rule():
    rng = np.random.default_rng(6)
    m = 150
    B = rng.normal(size=(m, 12))
    cov = B @ B.T                                                          # rank 12: chol() fails, eigen() takes over
    L, method = mvn_factor(cov)
    assert method == "eigen"
    lam = np.linalg.eigvalsh(cov)[-1]
    assert np.max(np.abs(L @ L.T - cov)) <= 1e-12 * lam
    Lo, mo = orc.mvn_factor(cov)
    assert mo == 2 and np.max(np.abs(L @ L.T - Lo @ Lo.T)) <= 1e-12 * lam
    assert np.max(np.abs(np.linalg.norm(L, axis=0) ** 2 - np.linalg.eigvalsh(cov)[::-1].clip(0))) <= 1e-11 * lam   # columns scaled by sqrt(lambda_k), decreasing
    Z = rng.normal(size=(m, 4))
    mean = rng.normal(size=m)
    out = multivariate_normal(4, mean, cov, z=Z)
    assert np.max(np.abs(out - (mean[:, None] + L @ Z))) <= 1e-12 * np.abs(out).max()
    with pytest.raises(ValueError, match="eigval"):                        # stopifnot(all(eigval > -tol * abs(eigval[1])))
        multivariate_normal(1, np.zeros(3), np.diag([2.0, 1.0, -1e-3]))
    L, method = mvn_factor(np.diag([2.0, 1.0, -1e-9]))                     # inside tol: pmax(eigval, 0)
    assert method == "eigen" and np.max(np.abs(L @ L.T - np.diag([2.0, 1.0, 0.0]))) <= 1e-15
    # statistical sanity with the host generator: sample covariance of many draws approaches cov
    draws = multivariate_normal(20000, np.zeros(m), cov, rng=np.random.default_rng(0))
    emp = draws @ draws.T / draws.shape[1]
    assert np.max(np.abs(emp - cov)) <= 0.08 * lam


def test_posterior_draws_and_variance_data():
    x = np.linspace(-2, 2, 25)
    y = np.sin(2 * x)
    gp = GPR(x.reshape(1, -1), y, 0.01, cov_func(sqrexp, l=0.8))
    rng = np.random.default_rng(1)
    Z = rng.normal(size=(200, 5))
    d = gp.posterior_draws(5, z=Z)
    lo, hi = expand_range(x)
    assert d["x"].shape == (200,) and abs(d["x"][0] - lo) < 1e-15 and abs(d["x"][-1] - hi) < 1e-15
    mean, cov = gp.predict(d["x"], pointwise_var=False)
    assert nerr(d["y"][:, 0], mean[:, 0]) == 0.0 and nerr(d["y"][:, 1], np.diag(cov)) == 0.0
    L, method = mvn_factor(cov)
    assert method == "eigen"                                               # a posterior covariance is numerically rank deficient
    assert d["z"].shape == (200, 5) and np.max(np.abs(d["z"] - (mean + L @ Z))) <= 1e-10
    inside = np.abs(d["z"] - mean) <= 6 * np.sqrt(np.maximum(d["y"][:, 1:2], 0)) + 1e-6
    assert inside.all()
    v = gp.posterior_variance([0.0, 1.0], length_out=50)
    xs = np.concatenate([[0.0, 1.0], v["x"]])
    full = gp.predict(xs, pointwise_var=False)[1]
    assert v["y"].shape == (50, 2) and nerr(v["y"], full[2:, :2]) == 0.0
    with pytest.raises(ValueError, match="multidimensional"):
        GPR(np.zeros((2, 3)) + np.arange(3), np.arange(3.0), 0.1, cov_func(sqrexp, l=1.0)).posterior_draws()
