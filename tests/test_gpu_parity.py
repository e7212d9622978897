"""GPU parity tests proper: the HIP path, called through the C ABI by the host mirror of the reference API,
against (1) the reference's closed-form known answers, (2) the committed golden vectors, (3) the CPU oracle
on seeded inputs.  Tolerance: normwise 1e-10 in fp64 (BASELINE.json north_star); kernel fills 1e-13."""
import ctypes as C
import math
import os
import warnings

import numpy as np
import pytest

from conftest import TOL, nerr, oracle_params
import gprc_amd
from gprc_amd import (GPR, GPC, GprcError, NotPositiveDefinite, cov_func, covariance_matrix, constant, linear, polynomial, sqrexp,
                      gammaexp, rationalquadratic)
from gprc_amd import _native as nat
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

GENERIC = {"constant": constant, "linear": linear, "polynomial": polynomial, "sqrexp": sqrexp, "gammaexp": gammaexp,
           "rationalquadratic": rationalquadratic}


def kfun(kind, params):
    return cov_func(GENERIC[kind], **params)


# ---- the reference's own test file, line by line (tests/testthat/test-gpr.R:5-28) -------------------
def test_predict_works_reference_known_answers():
    X1 = np.array([[-1 / 2, 1 / 2]])
    y1 = np.array([4.0, 4.0])
    GPRobj1 = GPR.polynomial.new(X1, y1, 1 / 2, 1 / 4, 1)
    np.testing.assert_allclose(GPRobj1.predict(0).ravel(), [2, 1 / 8], rtol=0, atol=1.5e-8)
    X2 = np.array([[1.0, 2.0]])
    y2 = np.array([1.0, 3.0])
    GPRobj2 = GPR.constant.new(X2, y2, 1, 1)
    np.testing.assert_allclose(GPRobj2.predict(3).ravel(), [4 / 3, 1 / 3], rtol=0, atol=1.5e-8)
    X3 = np.array([[100.0, 54.0]])
    y3 = np.array([5.0, 0.0])
    GPRobj3 = GPR.constant.new(X3, y3, 1, 1)
    np.testing.assert_allclose(GPRobj3.predict(math.pi).ravel(), [5 / 3, 1 / 3], rtol=0, atol=1.5e-8)
    X4 = np.array([[1.0, 2.0]])
    y4 = np.array([0.0, 1.0])
    GPRobj4 = GPR.sqrexp.new(X4, y4, 1, 1)
    cov = 1 - (2 * math.exp(-1) - 2 * math.exp(-3) + 2 * math.exp(-4)) / (4 - math.exp(-1))
    got = GPRobj4.predict(0).ravel()
    np.testing.assert_allclose(got, [(2 * math.exp(-2) - math.exp(-1)) / (4 - math.exp(-1)), cov], rtol=0, atol=1.5e-8)
    # far tighter than the reference's 1.5e-8
    assert abs(got[0] - (-0.026763669631486305)) < 1e-15 and abs(got[1] - 0.8147594463104431) < 1e-15


def test_gpc_works_reference_sign_tests():
    # tests/testthat/test-gpc.R:5-27 with the constructor's real argument order (X, y, k, epsilon);
    # kappa(x,y) = exp(-3 (x-y)^2) == sqrexp with l = sqrt(1/6)
    X = np.round(np.arange(-1, 1.0001, 0.1), 10).reshape(1, -1)
    y = 2.0 * (X[0] > 0) - 1
    gc = GPC.new(X, y, cov_func(sqrexp, l=math.sqrt(1 / 6)), 1e-5)
    assert gc.predict_class(-0.2)[0] < 0.5 < gc.predict_class(0.2)[0]
    X = np.concatenate([np.round(np.arange(-1, -0.0999, 0.1), 10), np.round(np.arange(0, 1.0001, 0.2), 10)]).reshape(1, -1)
    y = 2.0 * (X[0] > 0) - 1
    gc = GPC.new(X, y, cov_func(sqrexp, l=math.sqrt(1 / 6)), 1e-5)
    assert gc.predict_class(-0.2)[0] < 0.5 < gc.predict_class(0.2)[0]
    s = np.arange(-1, 1.0001, 0.5)
    X = np.stack([np.repeat(s, len(s)), np.tile(s, len(s))])
    y = 2.0 * (X[0] > X[1]) - 1
    gc = GPC.new(X, y, cov_func(sqrexp, l=1), 1e-5)
    assert gc.predict_class(np.array([[0.0], [1.0]]))[0] < 0.5 < gc.predict_class(np.array([[-0.3], [-0.9]]))[0]


def test_gpc_works_reference_cluster_test():
    """tests/testthat/test-gpc.R:30-36 (the fourth sign test): two 2-D Gaussian clusters drawn with the package's own
    multivariate_normal(n, c(+-0.5, +-0.5), diag(c(0.1, 0.1))) -- here on the device, Cholesky branch -- labelled +1 / -1,
    k = sqrexp(l = 1); the class probability must be < 0.5 at (-0.2, -0.2) and > 0.5 at (0.2, 0.2).  The reference draws
    unseeded (R's Mersenne-Twister stream is not reproducible here), so the expectation has to hold for any draw: six
    seeded draws are run, each also against the oracle on the same points."""
    from gprc_amd import multivariate_normal
    n = 10
    for seed in range(6):
        rng = np.random.default_rng(seed)
        X = np.hstack([multivariate_normal(n, [0.5, 0.5], np.diag([0.1, 0.1]), rng=rng),
                       multivariate_normal(n, [-0.5, -0.5], np.diag([0.1, 0.1]), rng=rng)])       # cbind(...)  :32
        y = np.repeat([1.0, -1.0], n)                                                               # :33
        assert X.shape == (2, 2 * n) and X[:, :n].mean() > 0 > X[:, n:].mean()
        gc = GPC.new(X, y, cov_func(sqrexp, l=1), 1e-5)                                             # :34-35 (argument order fixed)
        assert gc.predict_class(np.array([[-0.2], [-0.2]]))[0] < 0.5 < gc.predict_class(np.array([[0.2], [0.2]]))[0]   # :36-37
        ref = orc.gpc_fit(orc.SQREXP, [1.0], X, y, 1e-5)
        fs, vf = orc.gpc_predict_latent(orc.SQREXP, [1.0], X, y, ref["f_hat"], ref["L"], np.array([[-0.2, 0.2], [-0.2, 0.2]]))
        got = gc.predict_latent(np.array([[-0.2, 0.2], [-0.2, 0.2]]))
        assert gc.iterations == ref["iters"] and nerr(got[0], fs) <= 1e-9 and nerr(got[1], vf) <= 1e-9


# ---- committed golden vectors ------------------------------------------------------------------------
def test_golden_closed_forms(golden):
    for c in golden.of_type("gpr_closed_form"):
        g = GPR(golden.get(c, "X"), golden.get(c, "y"), c["noise"], kfun(c["kernel"], c["params"]))
        pr = g.predict(golden.get(c, "Xs"))
        assert pr.shape == (1, 2)
        assert abs(pr[0, 0] - golden.get(c, "mean")[0]) < 1e-14 and abs(pr[0, 1] - golden.get(c, "var")[0]) < 1e-14, c["name"]


def test_golden_gpr(golden):
    for c in golden.of_type("gpr"):
        X, y, Xs = golden.get(c, "X"), golden.get(c, "y"), golden.get(c, "Xs")
        if c["name"] == "gpr_jitter_duplicates":
            # K is EXACTLY singular (duplicate points, noise 0): whether the first chol() attempt fails is decided
            # by the sign of the last rounding error -- LAPACK trips at minor 3, the textbook recurrence at 6, the
            # right-looking MFMA order may pass.  The reference's own outcome is platform-dependent here, so values
            # are not compared; the path must still return a finite, documented outcome.
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                g = GPR(X, y, c["noise"], kfun(c["kernel"], c["params"]))
            assert g.noise in (0.0, 0.01) and np.isfinite(g.alpha).all()
            continue
        with warnings.catch_warnings(record=True) as wl:
            warnings.simplefilter("always")
            g = GPR(X, y, c["noise"], kfun(c["kernel"], c["params"]))
        attempts = int(golden.get(c, "attempts")[0])
        assert g.noise == golden.get(c, "noise")[0], c["name"]
        assert (len(wl) == 1) == (attempts > 1), c["name"]       # R/GPRclass.R:144 warns iff the noise changed
        if attempts > 1:
            assert str(wl[0].message) == f"Noise got changed to {g.noise:.15g} to avoid errors in cholesky decomposition"
        assert nerr(g.alpha, golden.get(c, "alpha")) <= TOL, c["name"]
        assert nerr([g.logp], golden.get(c, "logp")) <= TOL, c["name"]
        assert nerr(np.diag(g.L), golden.get(c, "diagL")) <= TOL, c["name"]
        if "L" in c["outputs"]:
            assert nerr(g.L, golden.get(c, "L")) <= TOL, c["name"]
            assert np.array_equal(np.triu(g.L, 1), np.zeros_like(g.L))       # t(chol(.)) has a zero upper triangle
            assert nerr(covariance_matrix(X, X, g.k), golden.get(c, "K")) <= 1e-13, c["name"]
        pr = g.predict(Xs)
        assert pr.shape == (Xs.shape[1], 2)
        assert nerr(pr[:, 0], golden.get(c, "mean")) <= TOL and nerr(pr[:, 1], golden.get(c, "var")) <= TOL, c["name"]
        if "cov" in c["outputs"]:
            mean, cov = g.predict(Xs, pointwise_var=False)
            assert mean.shape == (Xs.shape[1], 1) and cov.shape == (Xs.shape[1],) * 2
            assert nerr(cov, golden.get(c, "cov")) <= TOL and nerr(mean[:, 0], golden.get(c, "mean")) <= TOL, c["name"]


def test_golden_not_positive_definite(golden):
    c = golden.of_type("gpr_notpd")[0]
    X, y = golden.get(c, "X"), golden.get(c, "y")
    ctx = nat.default_context()
    _, pp, npar = nat.params_array(oracle_params(c["kernel"], c["params"]))
    model = C.c_void_p()
    Xf = np.asfortranarray(X)
    rc = nat.lib().gprc_gpr_fit(ctx.handle, nat.POLYNOMIAL, pp, npar, Xf.ctypes.data, 1, X.shape[1], y.ctypes.data, 0.0, C.byref(model))
    assert rc == int(golden.get(c, "info_first")[0]) == 2          # LAPACK info through the C ABI
    assert "leading minor of order 2" in nat.last_error()
    with pytest.raises(NotPositiveDefinite) as ei:
        nat.check(rc)
    assert ei.value.info == 2
    with pytest.raises(ArithmeticError, match="Inputs lead to non positive definite covariance matrix"):
        GPR(X, y, 0.0, kfun(c["kernel"], c["params"]))               # R/GPRclass.R:149


GPC_TOL = 1e-12   # observed on MI355X (tools/measure_gpc_errors.py): <= 8e-15 against the golden set and the oracle; the gate is ~100x that


def test_golden_gpc(golden):
    for c in golden.of_type("gpc"):
        X, y, Xs = golden.get(c, "X"), golden.get(c, "y"), golden.get(c, "Xs")
        gc = GPC(X, y, kfun(c["kernel"], c["params"]), c["epsilon"])
        assert gc.iterations == int(golden.get(c, "iters")[0]), c["name"]
        assert nerr(gc.f_hat, golden.get(c, "f_hat")) <= GPC_TOL, c["name"]
        assert nerr([gc.logq], golden.get(c, "logq")) <= GPC_TOL, c["name"]
        assert nerr(np.diag(gc.L), golden.get(c, "diagL")) <= GPC_TOL, c["name"]
        fs, vf = gc.predict_latent(Xs)
        assert nerr(fs, golden.get(c, "fs_bar")) <= GPC_TOL and nerr(vf, golden.get(c, "Vfs")) <= GPC_TOL, c["name"]
        assert nerr(gc.predict_class(Xs), golden.get(c, "prob")) <= 1e-7, c["name"]   # QUADPACK on both sides


# ---- against the oracle on seeded inputs --------------------------------------------------------------
KERNELS = [("constant", dict(c=1.7)), ("linear", dict(sigma=0.7)), ("polynomial", dict(sigma=0.5, p=3.0)),
           ("polynomial", dict(sigma=0.25, p=2.0)), ("sqrexp", dict(l=1.3)), ("gammaexp", dict(l=0.9, gamma=1.5)),
           ("gammaexp", dict(l=0.9, gamma=2.0)), ("rationalquadratic", dict(l=1.1, alpha=1.5)),
           ("rationalquadratic", dict(l=0.8, alpha=0.7)),    # general alpha: exp(-alpha log q)
           ("rationalquadratic", dict(l=0.9, alpha=3.5)), ("rationalquadratic", dict(l=1.2, alpha=0.5))]   # half-integer alpha: the rsqrt form


# the fill's other power forms (kernel matrices only: some of these are too ill-conditioned for the GPR-level comparisons below)
FILL_EXTRA = [("gammaexp", dict(l=1.1, gamma=1.0)), ("gammaexp", dict(l=0.8, gamma=0.5)),   # gamma = 0.5, 1, 1.5: the square-root form
              ("gammaexp", dict(l=0.9, gamma=1.3)),                                          # general gamma: exp(gamma / 2 log(s / l^2))
              ("polynomial", dict(sigma=0.5, p=5.0)),                                        # integer degree 3..8: multiplications
              ("polynomial", dict(sigma=40.0, p=2.5))]                                       # any other degree: R_pow (sigma keeps the base positive up to d = 37)


@pytest.mark.parametrize("d,nA,nB", [(1, 1, 1), (1, 5, 3), (2, 130, 67), (8, 257, 300), (20, 64, 129), (37, 300, 10)])
def test_covariance_matrix_all_kernels(d, nA, nB):
    rng = np.random.default_rng(100 + d)
    A, B = rng.uniform(-1, 1, (d, nA)), rng.uniform(-1, 1, (d, nB))
    for kind, par in KERNELS + FILL_EXTRA + [("linear", dict(sigma=list(rng.uniform(0.2, 1.5, d))))]:
        k = kfun(kind, par)
        ref = orc.kernel_matrix(orc.KERNEL_IDS[kind], oracle_params(kind, par), A, B)
        got = covariance_matrix(A, B, k)
        assert got.shape == (nA, nB) and got.flags.f_contiguous
        assert nerr(got, ref) <= 1e-13, (kind, par)
        m = min(nA, nB)
        assert nerr(k(A[:, :m], B[:, :m]), np.diag(ref)[:m]) <= 1e-13           # the closure contract
    assert covariance_matrix(A, B[:, :0], kfun("sqrexp", dict(l=1.0))).shape == (nA, 0)
    Ksym = covariance_matrix(A, A, kfun("gammaexp", dict(l=0.9, gamma=1.5)))
    assert np.array_equal(Ksym, Ksym.T)                                           # bitwise symmetric, like the reference


def test_kernel_scalar_methods():
    # the .numeric methods (R/GPRclass.R:383-403): two vectors -> one number
    x, y = np.array([0.3, -0.2, 0.9]), np.array([0.1, 0.4, -0.5])
    assert abs(sqrexp(x, y, l=0.7) - math.exp(-((x - y) ** 2).sum() / (2 * 0.49))) < 1e-15
    assert abs(polynomial(x, y, 0.5, 3) - (x @ y + 0.5) ** 3) < 1e-15
    assert abs(linear(x, y, sigma=[1.0, 2.0, 3.0]) - (np.array([1.0, 2.0, 3.0]) * x * y).sum()) < 1e-15
    assert constant(x, y, c=4.0) == 4.0


@pytest.mark.parametrize("n,d,ns", [(1, 1, 4), (127, 3, 129), (513, 2, 50), (1500, 8, 333),
                                    (512, 2, 128), (1024, 3, 127), (1025, 1, 129), (2049, 4, 257)])   # at / around whole panels: no padding, one padded row
def test_gpr_against_oracle(n, d, ns):
    rng = np.random.default_rng(n)
    X = rng.uniform(-1, 1, (d, n))
    y = 0.1 * (X ** 3).sum(0) + rng.normal(0, 0.1, n)
    Xs = rng.uniform(-1, 1, (d, ns))
    kinds = KERNELS if n <= 600 else [("sqrexp", dict(l=1.0)), ("rationalquadratic", dict(l=1.0, alpha=1.5))]
    for kind, par in kinds:
        noise = 0.1
        f = orc.gpr_fit(orc.KERNEL_IDS[kind], oracle_params(kind, par), X, y, noise)
        g = GPR(X, y, noise, kfun(kind, par))
        assert g.noise == f["noise"]
        assert nerr(g.L, f["L"]) <= TOL and nerr(g.alpha, f["alpha"]) <= TOL, (kind, n)
        assert abs(g.logp - f["logp"]) <= TOL * abs(f["logp"])
        mr, vr = orc.gpr_predict(orc.KERNEL_IDS[kind], oracle_params(kind, par), X, f["L"], f["alpha"], Xs)
        pr = g.predict(Xs)
        assert nerr(pr[:, 0], mr) <= TOL and nerr(pr[:, 1], vr) <= TOL, (kind, n)
        if ns <= 130:
            mean, cov = g.predict(Xs, pointwise_var=False)
            _, cr = orc.gpr_predict(orc.KERNEL_IDS[kind], oracle_params(kind, par), X, f["L"], f["alpha"], Xs, pointwise=False)
            assert nerr(cov, cr) <= TOL and nerr(np.diag(cov), pr[:, 1]) <= TOL


def test_predict_input_forms_and_edges():
    rng = np.random.default_rng(1)
    X = rng.uniform(-1, 1, (2, 40))
    y = rng.normal(size=40)
    g = GPR(X, y, 0.1, cov_func(sqrexp, l=1.0))
    Xs = rng.uniform(-1, 1, (2, 6))
    a = g.predict(Xs)
    b = g.predict(Xs.reshape(-1, order="F"))                     # bare vector -> d x len/d (R/GPRclass.R:157-159)
    assert np.array_equal(a, b)
    assert g.predict(np.zeros((2, 0))).shape == (0, 2)           # empty X_star
    with pytest.raises(ValueError):
        g.predict(np.zeros(3))                                   # length %% nrow(X) != 0
    for name in ("X", "k", "y", "noise", "L", "alpha", "logp"):
        with pytest.raises(AttributeError, match="read only"):
            setattr(g, name, 0)
    assert np.array_equal(g.X, X) and np.array_equal(g.y, y)
    g1 = GPR(np.array([0.1, 0.5, 0.9]), np.array([1.0, 2.0, 3.0]), 0.5, cov_func(sqrexp, l=1.0))  # vector X -> 1 x n
    assert g1.X.shape == (1, 3) and g1.predict(0.3).shape == (1, 2)


def test_chunked_predict_is_bitwise_chunk_invariant(monkeypatch):
    rng = np.random.default_rng(2)
    d, n, ns = 4, 700, 900
    X = rng.uniform(-1, 1, (d, n))
    y = rng.normal(size=n)
    Xs = rng.uniform(-1, 1, (d, ns))
    g = GPR(X, y, 0.1, cov_func(sqrexp, l=1.0))
    whole = g.predict(Xs)
    monkeypatch.setenv("GPRC_CHUNK_BYTES", str(256 * 1024 * 8))  # 256 rows per chunk at n_pad = 1024
    ctx2 = nat.Context(0)
    g2 = GPR(X, y, 0.1, cov_func(sqrexp, l=1.0), ctx=ctx2)
    parts = g2.predict(Xs)
    assert np.array_equal(whole, parts)
    g2.close()
    ctx2.close()


def test_device_pointers_are_used_in_place():
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(3)
    d, n, ns = 5, 300, 77
    X = np.asfortranarray(rng.uniform(-1, 1, (d, n)))
    y = rng.normal(size=n)
    Xs = np.asfortranarray(rng.uniform(-1, 1, (d, ns)))
    host = GPR(X, y, 0.1, cov_func(sqrexp, l=1.0)).predict(Xs)
    dev = torch.device("cuda:0")
    Xd, yd = torch.from_numpy(X.T.copy()).to(dev), torch.from_numpy(y).to(dev)
    Xsd = torch.from_numpy(Xs.T.copy()).to(dev)
    mean, var = torch.empty(ns, dtype=torch.float64, device=dev), torch.empty(ns, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    ctx = nat.default_context()
    _, pp, npar = nat.params_array([1.0])
    model = C.c_void_p()
    nat.check(nat.lib().gprc_gpr_fit(ctx.handle, nat.SQREXP, pp, npar, Xd.data_ptr(), d, n, yd.data_ptr(), 0.1, C.byref(model)))
    nat.check(nat.lib().gprc_gpr_predict(model, Xsd.data_ptr(), ns, 1, mean.data_ptr(), var.data_ptr()))
    nat.lib().gprc_model_free(model)
    assert np.array_equal(mean.cpu().numpy(), host[:, 0]) and np.array_equal(var.cpu().numpy(), host[:, 1])


def test_gpc_against_oracle_random():
    rng = np.random.default_rng(4)
    X = rng.uniform(-1, 1, (3, 600))                              # two panels
    y = np.sign(X.sum(0) + 0.2 * rng.normal(size=600))
    y[y == 0] = 1.0
    Xs = rng.uniform(-1, 1, (3, 41))
    # the reference's stop rule (R/GPCclass.R:90-91) fires on this problem: the objective improves by > 10
    # after iteration 1.  Parity includes that error ...
    with pytest.raises(ArithmeticError, match="Apparently does not converge."):
        orc.gpc_fit(orc.SQREXP, [1.0], X, y, 1e-5)
    with pytest.raises(ArithmeticError, match="Apparently does not converge."):
        GPC(X, y, cov_func(sqrexp, l=1.0), 1e-5)
    # ... and with the rule switched off on both sides the numbers agree
    oc = orc.gpc_fit(orc.SQREXP, [1.0], X, y, 1e-5, divergence_stop=False)
    gc = GPC(X, y, cov_func(sqrexp, l=1.0), 1e-5, reference_stop=False)
    assert gc.iterations == oc["iters"]
    assert nerr(gc.f_hat, oc["f_hat"]) <= GPC_TOL and abs(gc.logq - oc["logq"]) <= GPC_TOL * abs(oc["logq"])
    assert nerr(gc.L, oc["L"]) <= GPC_TOL
    fs, vf = gc.predict_latent(Xs)
    ofs, ovf = orc.gpc_predict_latent(orc.SQREXP, [1.0], X, y, oc["f_hat"], oc["L"], Xs)
    assert nerr(fs, ofs) <= GPC_TOL and nerr(vf, ovf) <= GPC_TOL
    for name in ("X", "k", "y", "f_hat", "L", "logq"):
        with pytest.raises(AttributeError, match="read only"):
            setattr(gc, name, 0)


def test_distributed_driver_single_rank_matches_host_path():
    """The bench/multi-GPU driver (world 1, with and without look-ahead) gives the host path's numbers."""
    pytest.importorskip("torch")
    from gprc_amd.distributed import DistributedGPR, HipOps, SingleComm
    rng = np.random.default_rng(5)
    d, n, ns = 8, 1700, 260
    X = rng.uniform(-1, 1, (n, d))
    y = rng.normal(size=n)
    Xs = rng.uniform(-1, 1, (ns, d))
    g = GPR(X.T, y, 0.1, cov_func(sqrexp, l=1.0))
    ref = g.predict(Xs.T)
    for look in (True, False):
        ops = HipOps(0, nat.SQREXP, [1.0], d, n, 0.1)
        eng = DistributedGPR(ops, SingleComm(), lookahead=look)
        ypad = np.zeros(ops.geom.n_pad)
        ypad[:n] = y
        Xd, yd, Xsd = ops.from_host(X), ops.from_host(ypad), ops.from_host(Xs)
        mean, var = ops.zeros(ns), ops.zeros(ns)
        for _ in range(2):                                        # twice: buffers are reused across steps
            assert eng.fit(Xd, yd) == 0
            eng.predict_local(Xd, yd, Xsd, ns, mean, var)
        assert np.array_equal(ops.to_host(eng.alpha)[:n], g.alpha)
        assert np.array_equal(ops.to_host(mean), ref[:, 0]) and np.array_equal(ops.to_host(var), ref[:, 1])
        assert abs(float(ops.to_host(eng.scal)[0]) - g.logp) <= 1e-12 * abs(g.logp)
        ops.close()


def test_class_probability_kernel_against_quadrature():
    """the batched device quadrature that replaces stats::integrate in GPC$predict_class, against
    (a) an independent brute-force numpy quadrature (4e5-point trapezoid in t-space, spectrally accurate for these
    integrands) and (b) QUADPACK QAGI through scipy, the reference's own method.  sd is the reference's `sd = Vfs`."""
    from scipy import integrate, stats
    mus = np.array([-30.0, -8.0, -2.5, -0.3, 0.0, 0.2, 1.7, 6.0, 25.0])
    sds = np.array([1e-3, 0.05, 0.43, 0.999, 1.0, 1.8, 7.0, 40.0])
    M, S = np.meshgrid(mus, sds, indexing="ij")
    mu, sd = M.ravel(), S.ravel()
    out = np.empty(mu.size)
    ctx = nat.default_context()
    nat.check(nat.lib().gprc_class_probability(ctx.handle, mu.ctypes.data, sd.ctypes.data, mu.size, out.ctypes.data))
    t = np.linspace(-12, 12, 400001)
    wt = np.exp(-0.5 * t * t) / math.sqrt(2 * math.pi) * (t[1] - t[0])
    ref = np.array([(wt / (1 + np.exp(-(m + s * t)))).sum() for m, s in zip(mu, sd)])
    assert np.abs(out - ref).max() <= 1e-11
    assert ((out >= 0) & (out <= 1)).all()
    # QAGI on (-Inf, Inf) misses a narrow peak far from the origin (mu = +-8, sd = 0.05 returns ~0): a known failure
    # mode of integrate() that the reference shares -- there the device kernel is right and the reference is not.
    sel = [i for i in range(mu.size) if 0.43 <= sd[i] <= 7.0 and abs(mu[i]) <= 8]     # where QAGI itself is reliable
    qp = np.array([integrate.quad(lambda z: stats.norm.pdf(z, mu[i], sd[i]) / (1 + np.exp(-z)), -np.inf, np.inf)[0] for i in sel])
    assert np.abs(out[sel] - qp).max() <= 1e-7
    # degenerate sd: dnorm undefined / point mass -> NaN, as integrate() refuses it in the reference
    bad_mu, bad_sd, bad = np.array([0.0, 1.0]), np.array([0.0, -0.5]), np.empty(2)
    nat.check(nat.lib().gprc_class_probability(ctx.handle, bad_mu.ctypes.data, bad_sd.ctypes.data, 2, bad.ctypes.data))
    assert np.isnan(bad).all()


def test_class_probability_divergence_is_pinned():
    """Where predict_class(integrator="native") -- the Python default -- does NOT reproduce the reference
    (R/GPCclass.R:116-117): for a narrow peak far from the origin, stats::integrate's dqagi on (-Inf, Inf) never samples
    the peak and returns ~0 with a tiny error estimate (no warning), so the reference reports P ~ 0.  The device kernel
    returns the value of the integral.  Both behaviours are asserted so the divergence cannot drift unnoticed:
    quadpack (the reference's method, R's default tolerances) ~ 0; native == brute-force quadrature ~ sigmoid(mu)."""
    from gprc_amd.gpc import class_probability_quadpack
    mu = np.array([8.0, 8.0, 25.0, 6.0, -8.0, -30.0])
    sd = np.array([0.05, 1e-3, 0.05, 1e-3, 0.05, 1e-3])
    out = np.empty(mu.size)
    nat.check(nat.lib().gprc_class_probability(nat.default_context().handle, mu.ctypes.data, sd.ctypes.data, mu.size, out.ctypes.data))
    t = np.linspace(-12, 12, 400001)
    wt = np.exp(-0.5 * t * t) / math.sqrt(2 * math.pi) * (t[1] - t[0])
    brute = np.array([(wt / (1 + np.exp(-(m + s * t)))).sum() for m, s in zip(mu, sd)])
    assert np.abs(out - brute).max() <= 1e-11                      # native: the integral
    qp = class_probability_quadpack(mu, sd)                        # the reference's method
    assert (qp[:4] < 1e-4).all() and (out[:4] > 0.997).all()       # positive side: reference ~0, truth ~1 -- they DIFFER
    assert abs(out[0] - 0.99966423) < 1e-7 and qp[0] < 1e-30        # the case of the round-1 review: (8, 0.05)
    assert (qp[4:] < 1e-4).all() and (out[4:] < 1e-3).all()        # negative side: both ~0 (the reference is right by accident)
    # where dqagi does resolve the peak the two agree to integrate()'s own tolerance and better
    mu2, sd2 = np.array([8.0, -2.5, 0.2, 6.0]), np.array([0.43, 0.43, 1.8, 7.0])
    out2 = np.empty(4)
    nat.check(nat.lib().gprc_class_probability(nat.default_context().handle, mu2.ctypes.data, sd2.ctypes.data, 4, out2.ctypes.data))
    assert np.abs(out2 - class_probability_quadpack(mu2, sd2)).max() <= 1e-7


def test_predict_class_native_matches_reference_method(golden):
    for c in golden.of_type("gpc"):
        gc = GPC(golden.get(c, "X"), golden.get(c, "y"), kfun(c["kernel"], c["params"]), c["epsilon"])
        Xs = golden.get(c, "Xs")
        native = gc.predict_class(Xs)
        quadpack = gc.predict_class(Xs, integrator="quadpack")
        assert nerr(native, quadpack) <= 1e-7 and nerr(native, golden.get(c, "prob")) <= 1e-7, c["name"]


def test_wide_inputs_and_non_finite_values():
    """d far beyond the LDS staging width (16 coordinates per pass), linear's per-coordinate sigma at d = 100 and its
    documented limit, and NaN / Inf inputs: the reference's chol() fails on them through all ten jitter attempts."""
    rng = np.random.default_rng(17)
    d, n, ns = 100, 150, 40
    X = rng.uniform(-1, 1, (d, n)) / np.sqrt(d)
    y = rng.normal(size=n)
    Xs = rng.uniform(-1, 1, (d, ns)) / np.sqrt(d)
    sig = rng.uniform(0.5, 1.5, d)
    for name, kid, kw, par in [("sqrexp", orc.SQREXP, dict(l=0.7), [0.7]), ("rationalquadratic", orc.RATQUAD, dict(l=0.7, alpha=1.5), [0.7, 1.5]),
                               ("gammaexp", orc.GAMMAEXP, dict(l=0.7, gamma=1.3), [0.7, 1.3]), ("polynomial", orc.POLYNOMIAL, dict(sigma=0.5, p=3.0), [0.5, 3.0]),
                               ("linear", orc.LINEAR, dict(sigma=sig), list(sig))]:
        g = GPR(X, y, 0.1, cov_func(GENERIC[name], **kw))
        ref = orc.gpr_fit(kid, par, X, y, 0.1)
        mean, var = orc.gpr_predict(kid, par, X, ref["L"], ref["alpha"], Xs)
        pr = g.predict(Xs)
        assert nerr(g.alpha, ref["alpha"]) <= TOL and nerr(pr[:, 0], mean) <= TOL and nerr(pr[:, 1], var) <= TOL, name
    with pytest.raises(GprcError, match="256"):
        GPR(np.zeros((300, 4)) + np.arange(4), np.arange(4.0), 0.1, cov_func(linear, sigma=np.ones(300)))
    Xbad = rng.uniform(-1, 1, (2, 30))
    for bad in (np.nan, np.inf):
        Xb = Xbad.copy()
        Xb[1, 7] = bad
        with pytest.raises(ArithmeticError, match="non positive definite"):
            GPR(Xb, rng.normal(size=30), 0.1, cov_func(sqrexp, l=1.0))
    yb = rng.normal(size=30)
    yb[3] = np.nan                                                       # NaN target: the factor is fine, alpha and the mean are NaN (as in R)
    g = GPR(Xbad, yb, 0.1, cov_func(sqrexp, l=1.0))
    assert np.isnan(g.alpha).all() or np.isnan(g.alpha).any()
    assert np.isnan(g.predict(Xbad[:, :3])[:, 0]).all() and np.isfinite(g.predict(Xbad[:, :3])[:, 1]).all()


def test_left_and_right_looking_solve_schedules_are_bit_identical(monkeypatch):
    """The predict solve vt := vt L^-T has two schedules (right-looking per panel, left-looking with the C tile held in
    the accumulators across all earlier panels, for single panels or groups of panels); same products in the same
    order, so identical bits."""
    rng = np.random.default_rng(23)
    d, n, ns = 3, 1700, 300                       # 4 panels: K = 512, 1024, 1536 in the left-looking passes
    X = rng.uniform(-1, 1, (d, n))
    y = rng.normal(size=n)
    Xs = rng.uniform(-1, 1, (d, ns))
    g = GPR(X, y, 0.1, cov_func(sqrexp, l=0.7))
    out = {}
    for mode in ("right", "left", "2", "3"):              # "2": groups {0,1},{2,3}; "3": {0,1,2},{3}
        monkeypatch.setenv("GPRC_SOLVE", mode)
        out[mode] = (g.predict(Xs), g.predict(Xs, pointwise_var=False)[1])
    for mode in ("left", "2", "3"):
        assert np.array_equal(out[mode][0], out["right"][0]) and np.array_equal(out[mode][1], out["right"][1]), mode
    ref = orc.gpr_fit(orc.SQREXP, [0.7], X, y, 0.1)
    mean, var = orc.gpr_predict(orc.SQREXP, [0.7], X, ref["L"], ref["alpha"], Xs)
    assert nerr(out["left"][0][:, 0], mean) <= TOL and nerr(out["left"][0][:, 1], var) <= TOL


def test_factor_schedules_are_bit_identical(monkeypatch):
    """The one-GPU Cholesky sweep groups panels and applies all earlier panels to a group in one left-looking pass; any
    grouping -- one group (= right-looking), one panel per group (= left-looking), mixed -- gives the same bits."""
    rng = np.random.default_rng(29)
    d, n = 3, 2300                                 # 5 panels; lower tiles per panel: 74, 58, 42, 26, 10
    X = rng.uniform(-1, 1, (d, n))
    y = rng.normal(size=n)
    res = {}
    for mode in ("right", "1", "60", "100"):       # "1": every panel its own group; "60": {0},{1,2},{3,4}; "100": {0,1},{2,3,4}
        monkeypatch.setenv("GPRC_FACTOR", mode)
        g = GPR(X, y, 0.1, cov_func(rationalquadratic, l=0.8, alpha=1.5))
        res[mode] = (g.L.copy(), g.alpha.copy(), g.logp)
        g.close()
    for mode in ("1", "60", "100"):
        assert np.array_equal(res[mode][0], res["right"][0]) and np.array_equal(res[mode][1], res["right"][1]), mode
        assert res[mode][2] == res["right"][2]
    ref = orc.gpr_fit(orc.RATQUAD, [0.8, 1.5], X, y, 0.1)
    assert nerr(res["1"][0], ref["L"]) <= TOL and nerr(res["1"][1], ref["alpha"]) <= TOL


def test_two_contexts_fit_concurrently_on_one_gpu():
    """Two host threads, each with its own context (own stream), fit and predict at the same time on the one GPU: every fit runs
    its own factor service (a persistent launch waiting on counters that the same fit's other kernels feed) beside the other's.
    The results are those of the same fits run one after the other, bit for bit."""
    import threading
    rng = np.random.default_rng(77)
    probs = []
    for n, l in ((3000, 0.7), (2600, 0.9)):
        X = rng.uniform(-1, 1, (3, n)); y = rng.normal(size=n); Xs = rng.uniform(-1, 1, (3, 400))
        probs.append((X, y, Xs, l))
    ref = []
    for X, y, Xs, l in probs:
        g = GPR(X, y, 0.1, cov_func(sqrexp, l=l))
        ref.append((g.alpha.copy(), g.logp, g.predict(Xs).copy()))
        g.close()
    out, errs = [None, None], []

    def work(i):
        try:
            X, y, Xs, l = probs[i]
            ctx = nat.Context(0)
            res = []
            for _ in range(6):
                g = GPR(X, y, 0.1, cov_func(sqrexp, l=l), ctx=ctx)
                res.append((g.alpha.copy(), g.logp, g.predict(Xs).copy()))
                g.close()
            out[i] = res
            ctx.close()
        except Exception as e:   # noqa: BLE001
            errs.append(repr(e))

    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in ts]
    [t.join(timeout=300) for t in ts]
    assert not errs and all(o is not None for o in out), errs
    for i in range(2):
        for alpha, logp, pred in out[i]:
            assert np.array_equal(alpha, ref[i][0]) and logp == ref[i][1] and np.array_equal(pred, ref[i][2]), i


def test_factor_service_switch_changes_no_bit():
    """gprc_factor_service(0) makes every later factorisation of the process use one fused launch per panel (what the library
    falls back to by itself where kernels cannot run concurrently, e.g. under rocprofv3 --pmc); same factor, same alpha, same
    prediction, bit for bit; the switch reports its previous state."""
    L = nat.lib()
    rng = np.random.default_rng(78)
    X = rng.uniform(-1, 1, (3, 2900)); y = rng.normal(size=2900); Xs = rng.uniform(-1, 1, (3, 300))
    assert L.gprc_factor_service(-1) == 1
    g = GPR(X, y, 0.1, cov_func(sqrexp, l=0.8)); a = (g.L.copy(), g.alpha.copy(), g.logp, g.predict(Xs).copy()); g.close()
    try:
        assert L.gprc_factor_service(0) == 1 and L.gprc_factor_service(-1) == 0
        g = GPR(X, y, 0.1, cov_func(sqrexp, l=0.8)); b = (g.L.copy(), g.alpha.copy(), g.logp, g.predict(Xs).copy()); g.close()
    finally:
        assert L.gprc_factor_service(1) == 0
    assert L.gprc_factor_service(-1) == 1
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2] and np.array_equal(a[3], b[3])


def test_a_model_may_outlive_its_context():
    """A host language's garbage collector may finalise a context before the models fitted on it (Python does so at interpreter
    exit when both hang off module globals): gprc_model_free then must leave the dead context's stream and block pool alone.  In a
    child process, because the failure mode was a segmentation fault."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "import gprc_amd\n"
        "from gprc_amd import GPR, GPC, cov_func, sqrexp, _native as nat\n"
        "rng = np.random.default_rng(1); X = rng.uniform(-1, 1, (2, 700)); y = rng.normal(size=700)\n"
        "ctx = nat.Context(0)\n"
        "g = GPR(X, y, 0.1, cov_func(sqrexp, l=0.7), ctx=ctx)\n"
        "c = GPC(X, np.where(y > 0, 1.0, -1.0), cov_func(sqrexp, l=0.7), 1e-5, ctx=ctx, reference_stop=False)\n"
        "ctx.close()\n"
        "g.close(); c.close()\n"
        "g2 = GPR(X, y, 0.1, cov_func(sqrexp, l=0.7))\n"
        "print('OUTLIVED-OK', g2.logp)\n") % (os.path.dirname(here),)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OUTLIVED-OK" in r.stdout, (r.returncode, r.stderr[-1500:])


def test_not_positive_definite_through_the_factor_service():
    """A singular matrix over several panels (constant kernel, noise 0: K = 1 1^T, second pivot exactly 0): the factor service and
    everything that waits on it run to completion on garbage (every flag is published whatever the numbers are), the call returns
    LAPACK's info = 2 promptly, and GPR$new's jitter loop (R/GPRclass.R:140-149) then succeeds with a changed noise."""
    import time
    rng = np.random.default_rng(3)
    n = 2600
    X = rng.uniform(-1, 1, (2, n)); y = rng.normal(size=n)
    ctx = nat.default_context()
    _, pp, npar = nat.params_array([1.0])
    model = C.c_void_p()
    Xf = np.asfortranarray(X)
    t0 = time.time()
    rc = nat.lib().gprc_gpr_fit(ctx.handle, nat.CONSTANT, pp, npar, Xf.ctypes.data, 2, n, y.ctypes.data, 0.0, C.byref(model))
    assert rc == 2 and time.time() - t0 < 5.0
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        g = GPR(X, y, 0.0, cov_func(constant, c=1.0))
    assert g.noise > 0 and any("Noise got changed" in str(x.message) for x in w)
    assert np.all(np.isfinite(g.alpha))
    g.close()


def _child(code, env=None, timeout=600):
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    pre = "import sys; sys.path.insert(0, %r)\nimport numpy as np, hashlib\nimport gprc_amd\nfrom gprc_amd import GPR, GPC, cov_func, sqrexp, _native as nat\n" % (os.path.dirname(here),)
    r = subprocess.run([sys.executable, "-c", pre + code], capture_output=True, text=True, timeout=timeout, env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    return r


def test_service_timeout_refills_and_matches_the_service_result():
    """The fallback include/gprc_native.h promises, exercised without a profiler: GPRC_TEST_SERVICE_TIMEOUT makes the first service gate
    of a (child) process give up at once, info becomes GPRC_INFO_WAIT_TIMEOUT, and the fit entry points -- GPR, the GPC's IRLS loop --
    switch the service off (one line on stderr), rebuild the matrix and factor again.  Same bits as a fit under the service."""
    code = (
        "rng = np.random.default_rng(5); X = rng.uniform(-1, 1, (3, 2900)); y = rng.normal(size=2900); Xs = rng.uniform(-1, 1, (3, 200))\n"
        "assert nat.lib().gprc_factor_service(-1) == 1\n"
        "g = GPR(X, y, 0.1, cov_func(sqrexp, l=0.8))\n"
        "assert nat.lib().gprc_factor_service(-1) == 0, 'the forced timeout should have switched the service off'\n"
        "h = hashlib.sha256(g.alpha.tobytes() + np.float64(g.logp).tobytes() + g.predict(Xs).tobytes()).hexdigest()\n"
        "print('DIGEST', h)\n")
    r = _child(code, {"GPRC_TEST_SERVICE_TIMEOUT": "1"})
    assert "service is now OFF" in r.stderr
    import hashlib
    rng = np.random.default_rng(5); X = rng.uniform(-1, 1, (3, 2900)); y = rng.normal(size=2900); Xs = rng.uniform(-1, 1, (3, 200))
    assert nat.lib().gprc_factor_service(-1) == 1
    g = GPR(X, y, 0.1, cov_func(sqrexp, l=0.8))
    want = hashlib.sha256(g.alpha.tobytes() + np.float64(g.logp).tobytes() + g.predict(Xs).tobytes()).hexdigest()
    g.close()
    assert ("DIGEST " + want) in r.stdout


def test_predict_chunk_shrinks_to_the_memory_that_is_there():
    """gprc_gpr_predict sizes its K*^T chunk from hipMemGetInfo and, when an allocation still fails, halves the chunk and tries again
    (down to 256 rows) instead of failing the call (GPRC_CHUNK_BYTES is a wish, not a promise).  A child process hogs the GPU down to
    ~3 GiB free, then predicts 65536 points against n = 8192 -- 4.3 GB in one chunk -- (a) sized by the free-memory figure, (b) with that
    figure ignored (GPRC_IGNORE_MEMINFO: the first hipMalloc fails, the halved chunk fits).  Chunking never changes a bit."""
    code = (
        "import torch\n"
        "rng = np.random.default_rng(6); X = rng.uniform(-1, 1, (8, 8192)); y = rng.normal(size=8192); Xs = rng.uniform(-1, 1, (8, 65536))\n"
        "g = GPR(X, y, 0.1, cov_func(sqrexp, l=1.0))\n"
        "free, total = torch.cuda.mem_get_info()\n"
        "hog = torch.empty(int(free - (3 << 30)), dtype=torch.uint8, device='cuda'); torch.cuda.synchronize()\n"
        "pr = g.predict(Xs)\n"
        "print('FREE_GIB', torch.cuda.mem_get_info()[0] / 2**30)\n"
        "print('DIGEST', hashlib.sha256(pr.tobytes()).hexdigest())\n")
    rng = np.random.default_rng(6); X = rng.uniform(-1, 1, (8, 8192)); y = rng.normal(size=8192); Xs = rng.uniform(-1, 1, (8, 65536))
    import hashlib
    g = GPR(X, y, 0.1, cov_func(sqrexp, l=1.0))
    want = hashlib.sha256(g.predict(Xs).tobytes()).hexdigest()
    g.close()
    nat.check(nat.lib().gprc_ctx_trim(nat.default_context().handle))      # this process's own 4.3 GB chunk goes back first
    for env in ({}, {"GPRC_IGNORE_MEMINFO": "1"}):
        r = _child(code, env)
        assert ("DIGEST " + want) in r.stdout, (env, r.stdout[-500:])


def test_device_data_on_the_contexts_own_stream_needs_no_synchronisation():
    """The ordering rule of include/gprc_native.h for DEVICE pointers: the library reads and writes them on the context's stream, so
    data produced on THAT stream is ordered by the stream itself.  Here the context is created on a torch stream, the inputs are
    uploaded, the outputs allocated and the results read back on that stream -- with NO torch.cuda.synchronize() anywhere between
    torch's work and the library's (the other tests, whose contexts own a private non-blocking stream, must synchronise)."""
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(12)
    d, n, ns = 4, 3300, 900
    X = np.asfortranarray(rng.uniform(-1, 1, (d, n))); y = rng.normal(size=n); Xs = np.asfortranarray(rng.uniform(-1, 1, (d, ns)))
    host = GPR(X, y, 0.1, cov_func(sqrexp, l=0.9))
    want = host.predict(Xs)
    st = torch.cuda.Stream()
    ctx = nat.Context(0, st.cuda_stream)
    _, pp, npar = nat.params_array([0.9])
    for rep in range(3):
        with torch.cuda.stream(st):
            big = torch.randn(1 << 26, device="cuda")                       # ~0.3 s of queued work in front of the uploads
            for _ in range(20):
                big = big * 1.0000001
            Xd = torch.from_numpy(X.T.copy()).pin_memory().to("cuda", non_blocking=True)
            yd = torch.from_numpy(y).pin_memory().to("cuda", non_blocking=True)
            Xsd = torch.from_numpy(Xs.T.copy()).pin_memory().to("cuda", non_blocking=True)
            mean = torch.full((ns,), float("nan"), dtype=torch.float64, device="cuda")
            var = torch.full((ns,), float("nan"), dtype=torch.float64, device="cuda")
            model = C.c_void_p()
            nat.check(nat.lib().gprc_gpr_fit(ctx.handle, nat.SQREXP, pp, npar, Xd.data_ptr(), d, n, yd.data_ptr(), 0.1, C.byref(model)))
            nat.check(nat.lib().gprc_gpr_predict(model, Xsd.data_ptr(), ns, 1, mean.data_ptr(), var.data_ptr()))
            got = torch.stack([mean, var], 1).cpu().numpy()
        nat.lib().gprc_model_free(model)
        assert np.array_equal(got, want), rep
    ctx.close()
    host.close()
