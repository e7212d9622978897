"""Measures multivariate_normal() on both branches, host pointers, CPU beside it.

    python tools/bench_sampling.py [m] [rank]     -> one JSON line (commit under profiles/)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gprc_amd as g
from gprc_amd import _native as nat
from oracle import oracle as orc

m = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 64
rng = np.random.default_rng(20261004)
B = rng.normal(size=(m, rank))
low = B @ B.T                       # rank deficient: the eigen branch
full = low + m * np.eye(m)          # positive definite: the Cholesky branch
Z = rng.normal(size=(m, 8))
mean = np.zeros(m)


def timed(fn, reps=2):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    return (time.perf_counter() - t0) / reps, out


L = nat.lib()
t_chol, _ = timed(lambda: g.multivariate_normal(8, mean, full, z=Z))
L.gprc_prof_enable(1); L.gprc_prof_reset()
t_eig, _ = timed(lambda: g.multivariate_normal(8, mean, low, z=Z), 1)
jac = nat.prof_summary()["jacobi_sweep"]
L.gprc_prof_enable(0)
Lf, method = g.mvn_factor(low)
mc = min(m, 384)                    # CPU sample: the oracle's Jacobi is a scalar O(sweeps * m^3) loop
t_cpu, _ = timed(lambda: orc.mvn_factor(low[:mc, :mc]), 1)
t_np, _ = timed(lambda: np.linalg.eigh(low), 1)
print(json.dumps({
    "what": "multivariate_normal(8, mean, cov): factor + mean + L %*% Z, host pointers",
    "m": m, "rank_of_low": rank,
    "cholesky_branch_ms": round(t_chol * 1e3, 2),
    "eigen_branch_ms": round(t_eig * 1e3, 2), "method": method,
    "jacobi": {"sweeps": jac["count"] // 2, "ms_per_sweep": round(jac["ms"] / max(jac["count"], 1), 2),
               "gbs": round(jac["bytes"] / jac["ms"] * 1e-6, 1) if jac["ms"] else None},
    "reconstruction_err": float(np.max(np.abs(Lf @ Lf.T - low)) / np.linalg.eigvalsh(low)[-1]),
    "cpu_oracle_eigen_branch": {"m": mc, "threads": 1, "ms": round(t_cpu * 1e3, 1)},
    "numpy_lapack_eigh_ms": round(t_np * 1e3, 1),
}))
