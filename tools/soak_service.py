"""Soak: the same fit + predict repeated many times (single thread, then two threads with their own contexts) must give the same
bits every time -- a race in the factor service's hand-offs or in the vector solves' gates would show as an occasional difference.
    python tools/soak_service.py [reps]"""
import os, sys, threading, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gprc_amd
from gprc_amd import GPR, GPC, cov_func, sqrexp, _native as nat
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:16]


def run(n, l, reps, ctx=None, seed=0):
    rng = np.random.default_rng(seed)
    X = rng.uniform(-1, 1, (3, n)); y = rng.normal(size=n); Xs = rng.uniform(-1, 1, (3, 256))
    seen = {}
    for r in range(reps):
        g = GPR(X, y, 0.1, cov_func(sqrexp, l=l), ctx=ctx)
        d = digest(g.alpha, np.array([g.logp]), g.predict(Xs))
        g.close()
        seen[d] = seen.get(d, 0) + 1
        if len(seen) > 1 and seen[d] == 1:
            print(f"   n={n}: repetition {r} gave a new result", flush=True)
    return seen


bad = 0
for n, l in (() if os.environ.get("SOAK_THREADS_ONLY") else ((1500, 0.6), (3000, 0.7), (5200, 0.8), (9100, 0.9), (21000, 1.0))):
    s = run(n, l, reps if n < 20000 else max(5, reps // 10))
    print(f"n={n}: {len(s)} distinct result(s) over {sum(s.values())} fits", flush=True)
    bad += len(s) != 1
out = {}


def worker(i):
    ctx = nat.Context(0)
    out[i] = run((2600, 4100)[i], 0.7 + 0.1 * i, reps, ctx=ctx, seed=10 + i)
    ctx.close()


ts = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
[t.start() for t in ts]; [t.join() for t in ts]
for i in range(2):
    print(f"thread {i}: {len(out[i])} distinct result(s) over {sum(out[i].values())} fits")
    bad += len(out[i]) != 1
rng = np.random.default_rng(5)
X = rng.uniform(-1, 1, (2, 3000)); yl = np.where(X.sum(0) > 0, 1.0, -1.0)
seen = {}
for r in range(max(5, reps // 10)):
    gc = GPC(X, yl, cov_func(sqrexp, l=1.0), 1e-5, reference_stop=False)
    d = digest(gc.predict_latent(X[:, :200])[0])
    seen[d] = seen.get(d, 0) + 1
print(f"GPC n=3000: {len(seen)} distinct result(s) over {sum(seen.values())} fits")
bad += len(seen) != 1
print("SOAK", "OK" if bad == 0 else "FAILED")
sys.exit(1 if bad else 0)
