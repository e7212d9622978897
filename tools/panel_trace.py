"""Where the time of one fused panel launch goes: stage stamps of the factor role (the critical chain) of panel GPRC_PANEL_TRACE
(default 0) during ONE fit at n (default 8192), printed in microseconds.
    GPRC_PANEL_TRACE=0 GPRC_SERVICE=0 python tools/panel_trace.py [n]     (the fused launch per panel, not the factor service)"""
import ctypes as C, os, sys
os.environ.setdefault("GPRC_PANEL_TRACE", "0")
os.environ.setdefault("GPRC_SERVICE", "0")

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gprc_amd
from gprc_amd import GPR, cov_func, sqrexp, _native as nat
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
rng = np.random.default_rng(3)
X = rng.uniform(-1, 1, (8, n)); y = rng.normal(size=n)
for rep in range(3):
    g = GPR(X, y, 0.1, cov_func(sqrexp, l=1.0)); g.close()
    t = (C.c_int64 * 24)()
    nat.check(nat.lib().gprc_prof_panel_trace(nat.default_context().handle, 0, t, 24))
    t = np.array(list(t), dtype=np.int64)
    us = (t - t[0]) / 100.0
    names = ["start"] + [f"{k}{j}" for j in range(4) for k in ("potf2_", "pubW_", "seeE_", "trsm_", "pubR_", "upd_")][:23]
    print("rep", rep, " ".join(f"{nm}={v:.1f}" for nm, v in zip(names, us) if v >= 0 and nm[:3] != "xxx"))
    d = np.diff(us[:20])
    print("   deltas:", " ".join(f"{names[i+1]}:{d[i]:.1f}" for i in range(len(d))))
