"""Host-side logic that needs no GPU: cov_func argument matching and tagging, argument validation in the
reference's stopifnot() order, read-only bindings, packed-layout index arithmetic, and the loud failure of
the compute path when no MI355X is present (no CPU fallback)."""
import math

import numpy as np
import pytest

import gprc_amd
from gprc_amd import (GPR, GPC, GprcError, cov_func, covariance_matrix, constant, linear, polynomial, sqrexp, gammaexp,
                      rationalquadratic)
from gprc_amd import _native as nat
from gprc_amd.covfunc import as_points
from gprc_amd.gpr import _ReadOnly, _r_num
from conftest import gpu_available


def test_cov_func_tags_and_argument_matching():
    k = cov_func(sqrexp, l=0.1)                                  # man/cov_func.Rd example
    assert k.gprc_kernel[0] == nat.SQREXP and k.gprc_kernel[1].tolist() == [0.1]
    k = cov_func(rationalquadratic, l=1, alpha=0.5)
    assert k.gprc_kernel[0] == nat.RATQUAD and k.gprc_kernel[1].tolist() == [1.0, 0.5]
    # R argument matching: named first, then positional in signature order (x, y, l, gamma)
    assert cov_func(gammaexp, 0.9, 1.5).params.tolist() == [0.9, 1.5]
    assert cov_func(gammaexp, gamma=1.5, l=0.9).params.tolist() == [0.9, 1.5]
    assert cov_func(gammaexp, 0.9, gamma=1.5).params.tolist() == [0.9, 1.5]
    assert cov_func(polynomial, sigma=0.25, p=1).params.tolist() == [0.25, 1.0]
    assert cov_func(linear, sigma=[0.1, 0.2, 0.3]).params.tolist() == [0.1, 0.2, 0.3]
    assert cov_func(constant, 2).params.tolist() == [2.0]
    with pytest.raises(TypeError):
        cov_func(sqrexp)                                         # argument "l" is missing
    with pytest.raises(TypeError):
        cov_func(sqrexp, 1.0, 2.0)
    with pytest.raises(TypeError):
        cov_func(sqrexp, gamma=2.0)                              # unused argument
    with pytest.raises(TypeError):
        cov_func(lambda x, y: np.exp(-3 * (x - y) ** 2))         # arbitrary closures stay on the R path


def test_untagged_closures_are_rejected_loudly():
    kappa = lambda x, y: np.exp(-3 * (x - y) ** 2)               # noqa: E731  (tests/testthat/test-gpc.R:7)
    with pytest.raises(TypeError, match="no CPU fallback"):
        covariance_matrix(np.zeros((1, 3)), np.zeros((1, 3)), kappa)
    with pytest.raises(TypeError, match="no CPU fallback"):
        GPR(np.zeros((1, 3)), np.zeros(3), 0.1, kappa)
    with pytest.raises(TypeError):
        GPC(np.zeros((1, 3)), np.array([1.0, -1.0, 1.0]), 1e-5, kappa)   # the stale argument order of test-gpc.R:8


def test_argument_validation_mirrors_stopifnot():
    k = cov_func(sqrexp, l=1.0)
    X = np.zeros((2, 4))
    y = np.zeros(4)
    with pytest.raises(TypeError):
        GPR(np.array(["a", "b"]), y, 0.1, k)                     # is.numeric(X)
    with pytest.raises(TypeError):
        GPR(X, np.zeros((4, 1)), 0.1, k)                         # is.vector(y)
    with pytest.raises(ValueError):
        GPR(X, y, -1.0, k)                                       # noise >= 0
    with pytest.raises(ValueError):
        GPR(X, y, [0.1, 0.2], k)                                 # length(noise) == 1
    with pytest.raises(ValueError):
        GPR(X, np.zeros(3), 0.1, k)                              # length(y) == ncol(X)
    with pytest.raises(TypeError):
        GPR(X, y, 0.1, 3.0)                                      # is.function(k)
    if not gpu_available():
        with pytest.raises(GprcError, match="no CPU fallback"):
            GPR(X, y, 0.1)                                       # default k = fit(...)$func runs the native objective
    with pytest.raises(TypeError):
        GPC(X, y, k, epsilon=0.0)                                # epsilon > 0
    with pytest.raises(ValueError):
        gprc_amd.GPR_linear(X, y, 0.1, sigma=[1.0, 2.0, 3.0])    # length(sigma) == nrow(X)
    with pytest.raises(ValueError):
        gprc_amd.GPR_sqrexp(X, y, 0.1, l=[1.0, 2.0])             # length(l) == 1
    with pytest.raises(ValueError):
        gprc_amd.GPR_constant(X, y, 0.1, c=-1.0)                 # c > 0
    assert GPR.sqrexp is gprc_amd.GPR_sqrexp and GPR.rationalquadratic is gprc_amd.GPR_rationalquadratic


def test_vector_inputs_become_matrices():
    assert as_points([1.0, 2.0, 3.0]).shape == (1, 3)                        # R/GPRclass.R:132
    Xs = as_points(np.arange(6.0), d=2, what="X_star")                       # R/GPRclass.R:157-159: column by column
    assert Xs.shape == (2, 3) and Xs[:, 1].tolist() == [2.0, 3.0]
    with pytest.raises(ValueError):
        as_points(np.arange(5.0), d=2, what="X_star")


def test_read_only_bindings():
    class T:
        X = _ReadOnly("X", lambda s: 42)
    t = T()
    assert t.X == 42
    with pytest.raises(AttributeError, match=r"`\$X` is read only"):        # R/GPRclass.R:234
        t.X = 1
    for name in ("X", "k", "y", "noise", "L", "alpha", "logp"):
        assert isinstance(GPR.__dict__[name], _ReadOnly)
    for name in ("X", "k", "y", "f_hat", "L", "logq"):
        assert isinstance(GPC.__dict__[name], _ReadOnly)


def test_r_number_formatting():
    assert _r_num(0.01) == "0.01" and _r_num(0.060000000000000005) == "0.06" and _r_num(1.0) == "1"


def test_packed_layout_arithmetic():
    L = nat.lib()
    NB = L.gprc_panel_width()
    assert NB % 128 == 0
    for n in (1, 2, 127, 512, 513, 8192, 65536):
        n_pad = L.gprc_pad(n)
        assert n_pad % NB == 0 and n <= n_pad < n + NB
        P = L.gprc_panel_count(n_pad)
        off = 0
        for p in range(P):
            assert L.gprc_panel_offset(n_pad, p) == off
            assert L.gprc_panel_elems(n_pad, p) == (n_pad - p * NB) * NB
            off += (n_pad - p * NB) * NB
        assert L.gprc_packed_size(n_pad) == off
        assert L.gprc_winv_size(n_pad) == n_pad * 128
    # the factor at n = 65536 takes ~half of a dense matrix
    n_pad = L.gprc_pad(65536)
    assert L.gprc_packed_size(n_pad) * 8 < 0.51 * 8 * 65536 ** 2


@pytest.mark.skipif(gpu_available(), reason="checks the no-GPU failure mode")
def test_compute_fails_loudly_without_gpu():
    assert gprc_amd.device_count() == 0
    k = cov_func(sqrexp, l=1.0)
    with pytest.raises(GprcError) as ei:
        GPR(np.zeros((1, 3)), np.zeros(3), 0.1, k)
    assert ei.value.status == nat.ERR_NO_DEVICE and "no CPU fallback" in ei.value.message
    with pytest.raises(GprcError):
        covariance_matrix(np.zeros((1, 3)), np.zeros((1, 3)), k)
    with pytest.raises(GprcError):
        k(np.zeros((1, 3)), np.zeros((1, 3)))


def test_brent_fmin_restatement_against_scipy():
    """R's optim(method = "Brent") is Brent's fmin; the restatement in gprc_amd.fit must find the same minima as
    scipy's independent implementation of the same published algorithm."""
    import math
    from scipy.optimize import minimize_scalar
    from gprc_amd.fit import brent_fmin
    tol = math.sqrt(np.finfo(float).eps)
    cases = [(lambda x: (x - 2.3) ** 2 + 1, 0, 10), (lambda x: math.cos(x) + 0.1 * x, 0, 10), (lambda x: -x * math.exp(-x), 0, 5),
             (lambda x: abs(x - 7.123) ** 1.5, 0, 10), (lambda x: -1.0 / (1 + (x - 0.4) ** 2), 0, 10)]
    for f, a, b in cases:
        mine = brent_fmin(f, a, b, tol)
        ref = minimize_scalar(f, bounds=(a, b), method="bounded", options={"xatol": 1e-10}).x
        assert abs(mine - ref) <= 2e-6 * max(1.0, abs(ref)), (mine, ref)
    calls = []
    brent_fmin(lambda x: calls.append(x) or (x - 1) ** 2, 0, 10, tol)
    assert abs(calls[0] - (0 + (3 - math.sqrt(5)) / 2 * 10)) < 1e-15      # first probe: the golden-section point


def test_vmmin_reproduces_the_documented_optim_example():
    """?optim's own example: optim(c(-1.2, 1), fr, grr, method = "BFGS") on the Rosenbrock banana reports
    $value 9.594956e-18, $counts function 110 / gradient 43, $convergence 0 -- the known answer for the restatement."""
    from gprc_amd.fit import vmmin
    fr = lambda v: 100 * (v[1] - v[0] ** 2) ** 2 + (1 - v[0]) ** 2
    grr = lambda v: np.array([-400 * v[0] * (v[1] - v[0] ** 2) - 2 * (1 - v[0]), 200 * (v[1] - v[0] ** 2)])
    par, val, nf, ng, fail = vmmin([-1.2, 1.0], fr, grr)
    assert (nf, ng, fail) == (110, 43, 0)
    assert abs(val - 9.594956e-18) <= 1e-23 and np.allclose(par, [1.0, 1.0], atol=1e-7)
    # a search direction that is not downhill -- here because one gradient component is NaN, as gammaexp's is in
    # R/fit.R:12 -- terminates at once at the start value (the "uphill" exit right after the initial reset)
    par, val, nf, ng, fail = vmmin([1.0, 1.0], fr, lambda v: np.array([np.nan, 1.0]))
    assert par.tolist() == [1.0, 1.0] and (nf, ng, fail) == (1, 1, 0)
    with pytest.raises(ArithmeticError, match="not finite"):
        vmmin([1.0, 1.0], lambda v: math.inf, grr)


def test_optim_until_error_bfgs_semantics():
    """R/fit.R:47-69 with method = "BFGS": the sentinel wraps f only; an error in gr aborts optim and the best
    successful evaluation so far is returned; with none, (start, f(start))."""
    from gprc_amd.fit import SENTINEL, _optim_bfgs_until_error
    f = lambda v: -((v[0] - 2.0) ** 2 + (v[1] + 1.0) ** 2)                 # maximum 0 at (2, -1)
    gr = lambda v: np.array([-2 * (v[0] - 2.0), -2 * (v[1] + 1.0)])
    par, val = _optim_bfgs_until_error((1.0, 1.0), f, gr)
    assert np.allclose(par, [2.0, -1.0], atol=1e-6) and abs(val) < 1e-10
    calls = {"g": 0}

    def gr_fails_later(v):
        calls["g"] += 1
        if calls["g"] > 1:
            raise ArithmeticError("computationally singular")
        return gr(v)
    seen = []
    par, val = _optim_bfgs_until_error((1.0, 1.0), lambda v: (seen.append((tuple(v), f(v))) or seen[-1][1]), gr_fails_later)
    best = max(seen, key=lambda t: t[1])
    assert par == best[0] and val == best[1] and len(seen) >= 2           # best-so-far, not the start value

    def gr_fails(v):
        raise ArithmeticError("computationally singular")
    par, val = _optim_bfgs_until_error((1.0, 1.0), f, gr_fails)            # f(start) was recorded before gr failed
    assert par == (1.0, 1.0) and val == f((1.0, 1.0))

    def f_fails(v):
        raise ArithmeticError("not positive definite")
    par, val = _optim_bfgs_until_error((1.0, 1.0), f_fails, gr_fails)      # nothing recorded: (start, sentinel)
    assert par == (1.0, 1.0) and val == SENTINEL


def test_simulation_host_pieces():
    """The host-side pieces of the simulate_* harness (R/simulation.R): summary(), limits handling, apply(., 2, f),
    iid_noise, the fractional length.out rule of the test grid."""
    from gprc_amd.simulation import Summary, _apply_cols, _limits, iid_noise
    s = Summary.of([1, 2, 3, 4, 10])                                       # R: summary(c(1, 2, 3, 4, 10))
    assert s == {"Min.": 1.0, "1st Qu.": 2.0, "Median": 3.0, "Mean": 4.0, "3rd Qu.": 4.0, "Max.": 10.0}
    assert Summary.of([0.5, 1.5, 4.0, 2.5])["1st Qu."] == 1.25            # quantile type 7
    assert _limits([-1, 1, -2, 2]).tolist() == [[-1, 1], [-2, 2]]          # matrix(limits, ncol = 2, byrow = TRUE)
    with pytest.raises(ValueError):
        _limits([1, 2, 3])
    M = np.array([[1.0, 2.0], [3.0, 4.0]])
    assert _apply_cols(M, lambda x: np.sum(x) ** 2).tolist() == [16.0, 36.0]
    assert _apply_cols(M[:1], lambda x: 0.1 * x ** 3).tolist() == [0.1 * 1.0 ** 3, 0.1 * 2.0 ** 3]
    noise = iid_noise(lambda n, sd: np.full(n, sd), sd=0.25)              # iid_noise(rnorm, sd = ...) shape contract
    assert noise(np.zeros((3, 7))).tolist() == [0.25] * 7
    if not gpu_available():
        with pytest.raises(GprcError, match="no CPU fallback"):
            gprc_amd.combine_all([[0.0, 1.0], [2.0, 3.0]])


def test_bench_refuses_a_world_size_it_was_not_asked_for():
    """bench.py --gpus N must never emit a line for a different N (round-1 finding: --gpus 8 under WORLD_SIZE=1 measured
    one GPU).  The check runs before torch or the GPU are touched."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "refusing" in r.stderr and not r.stdout.strip()
