"""Timeline of the factor service: per panel, the stamps gprc_prof_service_trace returns, in microseconds from the first panel's
chain start, during ONE fit at n (default 8192).
    GPRC_SERVICE_TRACE=1 python tools/service_trace.py [n]"""
import ctypes as C, os, sys
os.environ.setdefault("GPRC_SERVICE_TRACE", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gprc_amd
from gprc_amd import GPR, cov_func, sqrexp, _native as nat
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
P = (n + 511) // 512
rng = np.random.default_rng(3)
X = rng.uniform(-1, 1, (8, n)); y = rng.normal(size=n)
names = ["chain0", "chain1", "la0", "la1", "d_go", "d_last", "d_done", "strips0", "strips1", "upd0", "upd_la", "upd_pub", "upd_d2", "upd_end", "arrive"]
for rep in range(3):
    g = GPR(X, y, 0.1, cov_func(sqrexp, l=1.0)); g.close()
t = (C.c_int64 * (16 * P))()
nat.check(nat.lib().gprc_prof_service_trace(nat.default_context().handle, t, P))
t = np.array(list(t), dtype=np.int64).reshape(P, 16)
t0 = t[0, 0]
print("panel " + " ".join(f"{nm:>8s}" for nm in names))
for p in range(P):
    print(f"{p:5d} " + " ".join((f"{(t[p, k] - t0) / 100.0:8.1f}" if t[p, k] else "       -") for k in range(15)))
per = np.diff(t[:, 0]) / 100.0
print("chain start to chain start (us):", " ".join(f"{v:.0f}" for v in per))
