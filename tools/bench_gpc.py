"""Config 5 of BASELINE.json for the record: simulate_classification recipe, n = 16384, d = 4, sqexp, GPC$new
(Laplace IRLS) + latent predict + class probabilities on one MI355X.  Prints one JSON line with per-kernel-kind
event timings.  (Not the bench.py metric; kept under tools/ and profiles/ as supporting evidence.)"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (one HIP runtime, see _native._share_hip_runtime_with_torch)
import gprc_amd
from gprc_amd import GPC, cov_func, sqrexp, _native as nat

n, d, ns = int(os.environ.get("GPC_N", 16384)), 4, 10000
rng = np.random.Generator(np.random.Philox(20261004))
X = rng.uniform(-1, 1, (d, n))
y = np.where(X.sum(0) > 0, 1.0, -1.0)
per = 10
Xs = np.stack(np.meshgrid(*[np.linspace(-1, 1, per)] * d, indexing="ij"), -1).reshape(-1, d).T.copy()
k = cov_func(sqrexp, l=1.0)
GPC(X[:, :2048], y[:2048], k, 1e-5, reference_stop=False)  # warm-up
nat.lib().gprc_prof_reset(); nat.lib().gprc_prof_enable(1)
t0 = time.perf_counter(); gc = GPC(X, y, k, 1e-5, reference_stop=False); t1 = time.perf_counter()
fs, vf = gc.predict_latent(Xs); t2 = time.perf_counter()
p = gc.predict_class(Xs); t3 = time.perf_counter()
nat.lib().gprc_prof_enable(0)
prof = {kname: {"launches": r["count"], "ms": round(r["ms"], 2)} for kname, r in nat.prof_summary().items() if r["count"]}
print(json.dumps({"workload": f"c5: GPC n={n} d={d} sqexp, n*={Xs.shape[1]}", "irls_iterations": gc.iterations,
                  "fit_ms": round((t1 - t0) * 1e3, 1), "ms_per_irls_iteration": round((t1 - t0) * 1e3 / (gc.iterations + 1), 1),
                  "potrf_tflops_per_iteration": round(n ** 3 / 3 / ((t1 - t0) / (gc.iterations + 1)) * 1e-12, 2),
                  "predict_latent_ms": round((t2 - t1) * 1e3, 1), "predict_class_ms": round((t3 - t2) * 1e3, 1),
                  "train_accuracy_proxy": float((np.sign(fs[:10]) != 0).mean()), "kernels": prof}))
