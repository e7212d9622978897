"""The predict's fused chunk fill by kernel: GPR fit at n, predict n* points, library event timings of the fill kernels.  python tools/predict_fill_check.py [n] [n*]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
import gprc_amd
from gprc_amd import GPR, cov_func, sqrexp, rationalquadratic, _native as nat
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
rng = np.random.default_rng(3)
X = rng.uniform(-1, 1, (8, n)); y = rng.normal(size=n); Xs = rng.uniform(-1, 1, (8, ns))
L = nat.lib()
for name, k in (("sqrexp", cov_func(sqrexp, l=1.0)), ("rationalquadratic alpha=1.5", cov_func(rationalquadratic, l=1.0, alpha=1.5)),
                ("rationalquadratic alpha=1.7", cov_func(rationalquadratic, l=1.0, alpha=1.7))):
    g = GPR(X, y, 0.1, k)
    g.predict(Xs[:, :256])
    L.gprc_prof_reset(); L.gprc_prof_enable(1)
    g.predict(Xs)
    L.gprc_prof_enable(0)
    r = nat.prof_summary()["fill"]
    print(f"{name}: fill {r['count']} launches, {r['ms']:.2f} ms, {r['bytes'] / r['ms'] / 1e6:.0f} GB/s", flush=True)
    g.close()
