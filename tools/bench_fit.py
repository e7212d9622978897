"""Measures fit()'s two per-trial device calls: the objective dens (R/fit.R:117-124) and the
gradient dens_deriv (R/fit.R:126-139), with the CPU oracle's versions timed beside them on a bounded sample.

    python tools/bench_fit.py [n] [d]          -> one JSON line (commit under profiles/)

Algorithmic work: dens = n^3/3 (Cholesky) + n^2 fill; dens_deriv = n^3/3 (Cholesky) + n^3/3 (the triangular inverse L^-1,
row chunk by row chunk, for diag(K^-1); n^3 before round 3, when the solve ignored the zeros of the identity's rows) + 2 n^2
derivative evaluations."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gprc_amd as g
from gprc_amd import _native as nat
from oracle import oracle as orc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
d = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rng = np.random.default_rng(20261004)
X = rng.uniform(-1, 1, (d, n))
y = 0.1 * (X ** 3).sum(0) + 0.1 * rng.normal(size=n)
v = [0.3, 1.5]                                   # rationalquadratic (l, alpha): noise-free K stays positive definite


def timed(fn, reps):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    return (time.perf_counter() - t0) / reps, out


L = nat.lib()
L.gprc_prof_enable(1); L.gprc_prof_reset()
t_dens, logp = timed(lambda: g.dens(X, y, 0.1, "rationalquadratic", v), 3)
prof_dens = {k: r for k, r in nat.prof_summary().items() if r["count"]}
L.gprc_prof_reset()
t_grad, grad = timed(lambda: g.dens_deriv(X, y, "rationalquadratic", v), 3)
prof = {k: r for k, r in nat.prof_summary().items() if r["count"]}
L.gprc_prof_enable(0)

orc.set_threads(1)                               # the CPU lines below are single-thread figures
# the reference's own regime: X <- seq(-5, 5, by = 0.2) (51 points, d = 1): per-call latency, dominated by launch + PCIe
Xs_, ys_ = np.arange(-5, 5.0001, 0.2).reshape(1, -1), np.sin(np.arange(-5, 5.0001, 0.2))
t_small, _ = timed(lambda: g.dens(Xs_, ys_, 0.1, "sqrexp", [1.0]), 50)
t_small_o, _ = timed(lambda: orc.gpr_fit(orc.SQREXP, [1.0], Xs_, ys_, 0.1)["logp"], 50)

nc = 1024                                        # CPU sample: the oracle's gradient is an unblocked O(n^3) LU
t_cd, ref_logp = timed(lambda: orc.gpr_fit(orc.RATQUAD, v, X[:, :nc], y[:nc], 0.1)["logp"], 1)
t_cg, ref_grad = timed(lambda: orc.fit_gradient(orc.RATQUAD, v, X[:, :nc], y[:nc]), 1)
chk_l = g.dens(X[:, :nc], y[:nc], 0.1, "rationalquadratic", v)
chk_g = g.dens_deriv(X[:, :nc], y[:nc], "rationalquadratic", v)
print(json.dumps({
    "what": "fit() per-trial calls, rationalquadratic, host pointers (PCIe-inclusive: X, y in; scalars out)",
    "n": n, "d": d,
    "dens_ms": round(t_dens * 1e3, 2), "dens_tflops": round((n ** 3 / 3) / t_dens * 1e-12, 2),
    "dens_deriv_ms": round(t_grad * 1e3, 2), "dens_deriv_tflops": round((2 * n ** 3 / 3) / t_grad * 1e-12, 2),
    "deriv_rowsum_kernel_ms": round(prof["deriv_rowsum"]["ms"] / prof["deriv_rowsum"]["count"], 3) if "deriv_rowsum" in prof else None,
    "dens_deriv_kernels_ms_per_call": {k: {"launches": r["count"] // 4, "ms": round(r["ms"] / 4, 2), "tflops": round(r["flops"] / r["ms"] * 1e-9, 1) if r["ms"] else None}
                                       for k, r in prof.items()},
    "small_n": {"n": 51, "d": 1, "dens_ms": round(t_small * 1e3, 3), "cpu_oracle_dens_ms": round(t_small_o * 1e3, 3)},
    "cpu_oracle": {"n": nc, "threads": 1, "dens_ms": round(t_cd * 1e3, 1), "dens_deriv_ms": round(t_cg * 1e3, 1)},
    "parity_at_cpu_sample": {"dens_rel": abs(chk_l - ref_logp) / abs(ref_logp),
                             "grad_rel": float(np.max(np.abs(chk_g - ref_grad)) / np.max(np.abs(ref_grad)))},
}))
