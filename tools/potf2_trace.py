"""Where the 46 us of one diagonal-block factorisation (potf2_blocked_body<8>, second block of panel 0) go: stamps in microseconds.
    GPRC_POTF2_TRACE=1 GPRC_PANEL_TRACE=0 GPRC_SERVICE=0 python tools/potf2_trace.py [n]"""
import ctypes as C, os, sys
os.environ.setdefault("GPRC_POTF2_TRACE", "1"); os.environ.setdefault("GPRC_PANEL_TRACE", "0")
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
    us = (t[:20] - t[0]) / 100.0
    names = ["entry", "loaded", "diag0"] + [f"{ph}{s}" for s in range(8) for ph in ("B", "C")] + ["exit"]
    print("rep", rep, " ".join(f"{nm}={v:.2f}" for nm, v in zip(names, us)))
    print("   deltas:", " ".join(f"{names[i + 1]}:{us[i + 1] - us[i]:.2f}" for i in range(19)))
