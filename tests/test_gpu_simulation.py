"""The simulate_* harness (R/simulation.R): the device-built test grid against a literal restatement
of combine_all's rep(each, times), and the three simulations against the oracle run on the same inputs."""
import numpy as np
import pytest

from conftest import nerr
from gprc_amd import (GPC, GPR, combine_all, cov_func, iid_noise, simulate_classification, simulate_regression,
                      simulate_regression_gp, sqrexp)
from gprc_amd.simulation import Summary
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def r_combine_all(lst):
    """R/simulation.R:338-349 line by line: out[k, ] <- rep(lst[[k]], each = rev_prods[l + 1 - k], times = prods[k])."""
    l = len(lst)
    lengths = [len(a) for a in lst]
    prods = np.concatenate([[1], np.cumprod(lengths)])
    rev_prods = np.concatenate([[1], np.cumprod(lengths[::-1])])
    out = np.zeros((l, prods[l]))
    for k in range(1, l + 1):
        out[k - 1] = np.tile(np.repeat(lst[k - 1], rev_prods[l - k]), prods[k - 1])
    return out


def test_combine_all_matches_the_reference_indexing():
    for lst in ([np.array([1.0, 2.0, 3.0])], [np.arange(3.0), np.arange(10.0, 14.0)],
                [np.linspace(-1, 1, 4)] * 8, [np.arange(5.0), np.array([7.0]), np.arange(2.0)]):
        got = combine_all(lst)
        assert got.shape == r_combine_all(lst).shape and np.array_equal(got, r_combine_all(lst))
    assert combine_all([np.linspace(-1, 1, 4)] * 8).shape == (8, 65536)     # the C4 test grid of BASELINE.md
    with pytest.raises(ValueError):
        combine_all([])


def test_simulate_regression_reference_example():
    """example 1 of R/simulation.R:388-394: f(x) = 0.1 x^3 on [-6, 6], X = seq(-5, 5, by = 0.2), noise = 1."""
    f = lambda x: 0.1 * x ** 3
    X = np.round(np.arange(-5, 5.0001, 0.2), 10).reshape(1, -1)
    rng = np.random.default_rng(2)
    eps = rng.normal(0, 2, X.shape[1])
    k = cov_func(sqrexp, l=1.5)
    s = simulate_regression(f, np.array([[-6.0, 6.0]]), X, observation_noise=lambda M: eps, noise=1, k=k)
    assert isinstance(s, Summary) and list(s) == ["Min.", "1st Qu.", "Median", "Mean", "3rd Qu.", "Max."]
    tp = s.data["test_points"]
    assert tp.shape == (1, 10000) and tp[0, 0] == -6.0 and tp[0, -1] == 6.0
    y = f(X[0]) + eps
    fit = orc.gpr_fit(orc.SQREXP, [1.5], X, y, 1.0)
    mean, var = orc.gpr_predict(orc.SQREXP, [1.5], X, fit["L"], fit["alpha"], tp)
    assert nerr(s.data["predictions"][:, 0], mean) <= 1e-10 and nerr(s.data["predictions"][:, 1], var) <= 1e-10
    ref = Summary.of(np.abs(mean - f(tp[0])))
    assert all(abs(s[key] - ref[key]) <= 1e-9 * max(1.0, abs(ref[key])) for key in ref)
    assert s.data["x"][0] == -6.0 and abs(s.data["x"][1] - s.data["x"][0] - 0.05) < 1e-12   # seq(by = 0.05)
    # random training points (runif per axis) in 3-D, D > 1 slice plot data, default grid rule 10000^(1/3) -> 22 per axis
    s3 = simulate_regression(lambda x: 0.1 * np.sum(x ** 3), [-1, 1, -1, 1, -1, 1], training_size=30, noise=0.1,
                             k=cov_func(sqrexp, l=1.0), rng=np.random.default_rng(4))
    assert s3.data["test_points"].shape == (3, 22 ** 3) and s3.data["model"].X.shape == (3, 30)
    assert np.all(np.abs(s3.data["model"].X) <= 1) and s3["Max."] < 0.5
    with pytest.raises(ValueError, match="inside limits"):
        simulate_regression(f, [-1, 1], np.array([[0.0, 2.0]]), noise=1, k=k)


def test_simulate_regression_gp_follows_the_recipe():
    k = cov_func(sqrexp, l=1.0)
    rng = np.random.default_rng(9)
    Z = rng.normal(size=(300, 1))
    s = simulate_regression_gp(k, np.array([[-5.0, 5.0]]), training_size=10, random_training=False, noise=0.1, k=k, z=Z)
    d = s.data
    assert d["testpoints"].shape == (1, 300) and d["training_set"].tolist() == [30 * i - 1 for i in range(1, 11)]   # (1:10)*floor(300/10)
    K = orc.kernel_matrix(orc.SQREXP, [1.0], d["testpoints"], d["testpoints"])
    L, method = orc.mvn_factor(K)
    assert method == 2                                                      # a smooth prior on 300 points: eigen branch
    assert np.max(np.abs(K - L @ L.T)) <= 1e-10
    X = d["testpoints"][:, d["training_set"]]
    fit = orc.gpr_fit(orc.SQREXP, [1.0], X, d["f"][d["training_set"]], 0.1)
    mean, var = orc.gpr_predict(orc.SQREXP, [1.0], X, fit["L"], fit["alpha"], d["testpoints"])
    assert nerr(d["prediction"][:, 0], mean) <= 1e-10 and nerr(d["variance"], var) <= 1e-10
    assert abs(s["Mean"] - np.mean(np.abs(mean - d["f"]))) <= 1e-9
    s2 = simulate_regression_gp(k, [-5, 5], training_size=10, noise=0.1, k=k, rng=np.random.default_rng(3))
    assert len(set(s2.data["training_set"].tolist())) == 10                # sample.int without replacement
    with pytest.raises(ValueError):
        simulate_regression_gp(k, [-5, 5], training_size=400, test_size=300, noise=0.1, k=k)


def test_simulate_classification_reference_example():
    """example of R/simulation.R:306-309: labels sign(sum|x| > 2.5) on [-4, 4]^2, k = sqrexp l = 1."""
    f = lambda x: float(np.sum(np.abs(x)) > 2.5) - float(not (np.sum(np.abs(x)) > 2.5))
    k = cov_func(sqrexp, l=1.0)
    s = simulate_classification(f, [-4, 4, -4, 4], training_size=50, k=k, rng=np.random.default_rng(7))
    d = s.data
    assert d["test_points"].shape == (2, 10000) and set(np.unique(d["residual"])) <= {-2.0, 0.0, 2.0}
    Xtr, ytr = d["model"].X, d["model"].y
    ref = orc.gpc_fit(orc.SQREXP, [1.0], Xtr, ytr, 1e-5)
    fs, vf = orc.gpc_predict_latent(orc.SQREXP, [1.0], Xtr, ytr, ref["f_hat"], ref["L"], d["test_points"])
    lat = d["model"].predict_latent(d["test_points"])
    assert nerr(lat[0], fs) <= 1e-9 and nerr(lat[1], vf) <= 1e-9
    assert np.mean(np.abs(d["residual"])) / 2 < 0.25 and s["Max."] in (0.0, 2.0)


def test_config1_simulate_regression_n256():
    """BASELINE.json configs[0] exactly: simulate_regression with n = 256 training points drawn by runif on [-1, 1] (d = 1),
    y = f(x) + iid N(0, 0.1^2) noise, GPR$new(noise = 0.1, k = sqrexp l = 1), predict on the 10000-point equispaced grid
    (R/simulation.R:86-103) -- on the HIP path, against the oracle on the same inputs at the north star's 1e-10; and the same
    model through GPR(...)/predict directly (alpha, logp, L)."""
    f = lambda x: 0.1 * x ** 3
    k = cov_func(sqrexp, l=1.0)
    s = simulate_regression(f, np.array([[-1.0, 1.0]]), training_size=256, observation_noise=iid_noise(lambda m, sd: np.random.default_rng(5).normal(0, sd, m), 0.1),
                            noise=0.1, k=k, rng=np.random.default_rng(20261004))
    d = s.data
    X, y = d["model"].X, d["model"].y
    assert X.shape == (1, 256) and d["test_points"].shape == (1, 10000) and d["predictions"].shape == (10000, 2)
    assert d["test_points"][0, 0] == -1.0 and d["test_points"][0, -1] == 1.0
    fit = orc.gpr_fit(orc.SQREXP, [1.0], X, y, 0.1)
    mean, var = orc.gpr_predict(orc.SQREXP, [1.0], X, fit["L"], fit["alpha"], d["test_points"])
    assert nerr(d["predictions"][:, 0], mean) <= 1e-10 and nerr(d["predictions"][:, 1], var) <= 1e-10
    g = GPR(X, y, 0.1, k)
    assert nerr(g.alpha, fit["alpha"]) <= 1e-10 and abs(g.logp - fit["logp"]) <= 1e-10 * abs(fit["logp"]) and nerr(g.L, fit["L"]) <= 1e-10
    assert np.array_equal(g.predict(d["test_points"]), d["predictions"])      # the harness adds nothing to the numbers
    assert s["Max."] < 0.2 and g.noise == 0.1
