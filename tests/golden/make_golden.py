"""Generates tests/golden/gprc_golden.npz -- golden input/output vectors for the GP predict hot path.

The reference (R package gprc) cannot run in the build container (no R toolchain), so these vectors
come from an INDEPENDENT numpy/scipy restatement of the reference formulas written for this script
(LAPACK dpotrf / dtrtrs through scipy, numpy broadcasting for the kernels) -- deliberately sharing no
code with oracle/gprc_oracle.c or the HIP kernels.  The four closed-form cases of the reference's own
test file (tests/testthat/test-gpr.R:6-27) are stored with their exact expected values.

Run from the repo root:  python tests/golden/make_golden.py
Formulas restated: R/GPRclass.R:127-170 (GPR fit / predict), :355-357 (covariance_matrix), :382-402
(kernels); R/GPCclass.R:63, 66-118 (Laplace IRLS, latent predict, class-probability integral).
"""
import json
import math
import os

import numpy as np
import scipy.integrate as si
import scipy.linalg as sl
import scipy.stats as st

HERE = os.path.dirname(os.path.abspath(__file__))


# ---- kernels on d x m column-point matrices (numpy restatement) -----------------------------------
def _sqdist(A, B):
    return ((A[:, :, None] - B[:, None, :]) ** 2).sum(0)


def kmat(kind, par, A, B):
    A, B = np.atleast_2d(A), np.atleast_2d(B)
    if kind == "constant":
        return np.full((A.shape[1], B.shape[1]), par["c"])
    if kind == "linear":
        sg = np.broadcast_to(np.asarray(par["sigma"], float).reshape(-1, 1), (A.shape[0], 1))
        return ((sg * A)[:, :, None] * B[:, None, :]).sum(0)
    if kind == "polynomial":
        return ((A[:, :, None] * B[:, None, :]).sum(0) + par["sigma"]) ** par["p"]
    D = _sqdist(A, B)
    if kind == "sqrexp":
        return np.exp(-D / (2 * par["l"] ** 2))
    if kind == "gammaexp":
        return np.exp(-(np.sqrt(D) / par["l"]) ** par["gamma"])
    if kind == "rationalquadratic":
        return (1 + D / (2 * par["alpha"] * par["l"] ** 2)) ** (-par["alpha"])
    raise ValueError(kind)


def kdiag(kind, par, A):
    return np.array([kmat(kind, par, A[:, [i]], A[:, [i]])[0, 0] for i in range(A.shape[1])])


def gpr_case(kind, par, X, y, noise, Xs, with_cov):
    n = X.shape[1]
    K = kmat(kind, par, X, X)
    new_noise, attempts, L, info_first = noise, 0, None, 0
    for i in range(1, 11):  # R/GPRclass.R:141-148
        attempts = i
        try:
            L = sl.cholesky(K + new_noise * np.eye(n), lower=True)
            break
        except sl.LinAlgError as e:
            if i == 1:
                info_first = int(str(e).split("-th")[0].split()[-1]) if "-th" in str(e) else -1
            new_noise = 0.01 * i + noise
    alpha = sl.solve_triangular(L.T, sl.solve_triangular(L, y, lower=True), lower=False)
    logp = -0.5 * y @ alpha - np.log(np.diag(L)).sum() - n / 2 * math.log(2 * math.pi)
    Ks = kmat(kind, par, X, Xs)
    mean = Ks.T @ alpha
    v = sl.solve_triangular(L, Ks, lower=True)
    var = kdiag(kind, par, Xs) - (v * v).sum(0)
    out = dict(alpha=alpha, logp=np.array([logp]), mean=mean, var=var, noise=np.array([new_noise]),
               attempts=np.array([attempts]), info_first=np.array([info_first]), diagL=np.diag(L).copy())
    if n <= 64:
        out["L"] = L
        out["K"] = K
    if with_cov:
        out["cov"] = kmat(kind, par, Xs, Xs) - v.T @ v
    return out


def sigmoid(x):
    return 1 / (1 + np.exp(-x))


def gpc_case(kind, par, X, y, eps, Xs):
    n = X.shape[1]
    K = kmat(kind, par, X, X)
    f = np.zeros(n)
    it = 0
    while True:  # R/GPCclass.R:76-97
        it += 1
        P = sigmoid(f)
        W = (1 - P) * P
        sw = np.sqrt(W)
        L = sl.cholesky(np.eye(n) + np.outer(sw, sw) * K, lower=True)
        b = W * f + (y + 1) / 2 - P
        t = sl.solve_triangular(L.T, sl.solve_triangular(L, sw * (K @ b), lower=True), lower=False)
        a = b - sw * t
        f = K @ a
        obj = -(a * f).sum() / 2 - np.log(1 + np.exp(-y * f)).sum()
        if it > 1:
            if abs(obj - last) < eps:
                break
            if least + 10 < obj:
                raise ArithmeticError("Apparently does not converge.")
        else:
            least = obj
        last = obj
    P = sigmoid(f)
    W = (1 - P) * P
    sw = np.sqrt(W)
    L = sl.cholesky(np.eye(n) + np.outer(sw, sw) * K, lower=True)
    logq = obj - np.diag(L).sum()  # sic, R/GPCclass.R:103
    Ks = kmat(kind, par, X, Xs)
    fs = Ks.T @ ((y + 1) / 2 - P)
    v = sl.solve_triangular(L, sw[:, None] * Ks, lower=True)
    Vfs = kdiag(kind, par, Xs) - (v * v).sum(0)
    prob = np.array([si.quad(lambda z: sigmoid(z) * st.norm.pdf(z, loc=fs[i], scale=Vfs[i]), -np.inf, np.inf)[0]
                     for i in range(len(fs))])  # sd = Vfs (sic), R/GPCclass.R:117
    return dict(f_hat=f, logq=np.array([logq]), iters=np.array([it]), fs_bar=fs, Vfs=Vfs, prob=prob, diagL=np.diag(L).copy())


def main():
    rng = np.random.Generator(np.random.Philox(20261004))
    arrays, manifest = {}, []

    def add(name, meta, inputs, outputs):
        for k, v in {**inputs, **outputs}.items():
            arrays[f"{name}/{k}"] = np.asarray(v, dtype=np.float64)
        manifest.append(dict(name=name, inputs=sorted(inputs), outputs=sorted(outputs), **meta))

    # 1. the reference's own closed-form known answers (tests/testthat/test-gpr.R:6-27), exact values
    e = math.exp
    closed = [
        ("ref_polynomial", "polynomial", dict(sigma=0.25, p=1.0), [[-0.5, 0.5]], [4, 4], 0.5, [[0.0]], (2.0, 1 / 8)),
        ("ref_constant_a", "constant", dict(c=1.0), [[1, 2]], [1, 3], 1.0, [[3.0]], (4 / 3, 1 / 3)),
        ("ref_constant_b", "constant", dict(c=1.0), [[100, 54]], [5, 0], 1.0, [[math.pi]], (5 / 3, 1 / 3)),
        ("ref_sqrexp", "sqrexp", dict(l=1.0), [[1, 2]], [0, 1], 1.0, [[0.0]],
         ((2 * e(-2) - e(-1)) / (4 - e(-1)), 1 - (2 * e(-1) - 2 * e(-3) + 2 * e(-4)) / (4 - e(-1)))),
    ]
    for name, kind, par, X, y, noise, Xs, (m, v) in closed:
        add(name, dict(type="gpr_closed_form", kernel=kind, params=par, noise=noise, source="tests/testthat/test-gpr.R:6-27"),
            dict(X=np.array(X, float), y=np.array(y, float), Xs=np.array(Xs, float)), dict(mean=[m], var=[v]))

    # 2. random GPR cases, all six kernels (+ vector sigma), several shapes, both variance modes
    kernel_sets = [
        ("constant", dict(c=1.7)), ("linear", dict(sigma=0.7)), ("linearvec", None),
        ("polynomial", dict(sigma=0.5, p=3.0)), ("polynomial1", dict(sigma=0.25, p=1.0)), ("sqrexp", dict(l=1.3)),
        ("gammaexp", dict(l=0.9, gamma=1.5)), ("gammaexp2", dict(l=1.2, gamma=2.0)),
        ("rationalquadratic", dict(l=1.1, alpha=1.5)), ("rationalquadratic05", dict(l=0.8, alpha=0.5)),
    ]
    shapes = [(2, 1, 3), (17, 2, 5), (64, 1, 10), (256, 8, 33)]
    for label, par in kernel_sets:
        kind = {"linearvec": "linear", "polynomial1": "polynomial", "gammaexp2": "gammaexp",
                "rationalquadratic05": "rationalquadratic"}.get(label, label)
        for n, d, ns in shapes:
            X = rng.uniform(-1, 1, (d, n))
            y = 0.1 * (X ** 3).sum(0) + rng.normal(0, 0.1, n)
            Xs = rng.uniform(-1, 1, (d, ns))
            p = dict(sigma=list(rng.uniform(0.2, 1.5, d))) if par is None else par
            noise = 0.1 if n != 17 else 0.01
            out = gpr_case(kind, p, X, y, noise, Xs, with_cov=(ns <= 10))
            add(f"gpr_{label}_n{n}_d{d}", dict(type="gpr", kernel=kind, params=p, noise=noise), dict(X=X, y=y, Xs=Xs), out)
    X = rng.uniform(-1, 1, (8, 600))   # two 512-column panels on the GPU side
    y = 0.1 * (X ** 3).sum(0) + rng.normal(0, 0.1, 600)
    Xs = rng.uniform(-1, 1, (8, 64))
    add("gpr_sqrexp_n600_d8", dict(type="gpr", kernel="sqrexp", params=dict(l=1.0), noise=0.1), dict(X=X, y=y, Xs=Xs),
        gpr_case("sqrexp", dict(l=1.0), X, y, 0.1, Xs, with_cov=False))

    # 3. non-PD at noise 0 (duplicate points): info index, jitter sequence, final noise (R/GPRclass.R:139-151)
    Xd = np.array([[0.0, 0.5, 0.5, 1.0, 2.0, 2.0]])
    yd = np.arange(6.0)
    add("gpr_jitter_duplicates", dict(type="gpr", kernel="sqrexp", params=dict(l=1.0), noise=0.0), dict(X=Xd, y=yd, Xs=np.array([[0.25, 1.5]])),
        gpr_case("sqrexp", dict(l=1.0), Xd, yd, 0.0, np.array([[0.25, 1.5]]), with_cov=True))

    # 3b. robustly indefinite inputs (margins >= 0.0025, independent of summation order): a negative-sigma linear
    #     kernel whose first pivot turns positive at the 4th jitter step, and a negative-sigma polynomial whose
    #     second leading minor is negative for every jitter step (the reference stop()s, R/GPRclass.R:149)
    Xn = np.array([[0.15, 0.05]])
    yn = np.array([1.0, -1.0])
    add("gpr_jitter_linear_negative", dict(type="gpr", kernel="linear", params=dict(sigma=-1.0), noise=0.0),
        dict(X=Xn, y=yn, Xs=np.array([[0.1, 0.3]])), gpr_case("linear", dict(sigma=-1.0), Xn, yn, 0.0, np.array([[0.1, 0.3]]), with_cov=True))
    add("gpr_notpd_polynomial_negative", dict(type="gpr_notpd", kernel="polynomial", params=dict(sigma=-1.0, p=1.0), noise=0.0),
        dict(X=np.array([[2.0, 0.1, 0.5]]), y=np.array([1.0, 2.0, 3.0])), dict(info_first=[2], all_fail=[1]))

    # 4. GPC: the three deterministic problems of tests/testthat/test-gpc.R:5-27 (argument order corrected;
    #    kappa(x,y) = exp(-3 (x-y)^2) is sqrexp with l = sqrt(1/6)), plus one random 2-D problem
    l6 = math.sqrt(1 / 6)
    X1 = np.round(np.arange(-1, 1.0001, 0.1), 10).reshape(1, -1)
    y1 = 2.0 * (X1[0] > 0) - 1
    X2 = np.concatenate([np.round(np.arange(-1, -0.0999, 0.1), 10), np.round(np.arange(0, 1.0001, 0.2), 10)]).reshape(1, -1)
    y2 = 2.0 * (X2[0] > 0) - 1
    s = np.arange(-1, 1.0001, 0.5)
    X3 = np.stack([np.repeat(s, len(s)), np.tile(s, len(s))])
    y3 = 2.0 * (X3[0] > X3[1]) - 1
    gpc = [("gpc_ref_step", dict(l=l6), X1, y1, np.array([[-0.2, 0.2]])),
           ("gpc_ref_unbalanced", dict(l=l6), X2, y2, np.array([[-0.2, 0.2]])),
           ("gpc_ref_raster", dict(l=1.0), X3, y3, np.array([[0.0, -0.3], [1.0, -0.9]]))]
    X4 = rng.uniform(-1, 1, (2, 150))
    y4 = np.sign(X4.sum(0) + 0.3 * rng.normal(size=150))
    y4[y4 == 0] = 1
    gpc.append(("gpc_random_2d", dict(l=1.0), X4, y4, rng.uniform(-1, 1, (2, 12))))
    for name, par, X, y, Xs in gpc:
        add(name, dict(type="gpc", kernel="sqrexp", params=par, epsilon=1e-5, source="tests/testthat/test-gpc.R:5-27 (corrected order)"),
            dict(X=X, y=y, Xs=Xs), gpc_case("sqrexp", par, X, y, 1e-5, Xs))

    arrays["manifest"] = np.frombuffer(json.dumps(manifest).encode(), dtype=np.uint8)
    out = os.path.join(HERE, "gprc_golden.npz")
    np.savez_compressed(out, **arrays)
    print(f"wrote {out}: {len(manifest)} cases, {os.path.getsize(out) / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
