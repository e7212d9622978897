"""Observed normwise errors of the GPC path against the golden set and the oracle (what tests/test_gpu_parity.py gates): the gate is
then set to ~10x the worst observation.  python tools/measure_gpc_errors.py"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import Golden, nerr, oracle_params
import gprc_amd
from gprc_amd import GPC, cov_func, sqrexp
from oracle import oracle as orc
import test_gpu_parity as tp

worst = {}
def see(name, e):
    worst[name] = max(worst.get(name, 0.0), float(e))
g = Golden()
for c in g.of_type("gpc"):
    X, y, Xs = g.get(c, "X"), g.get(c, "y"), g.get(c, "Xs")
    gc = GPC(X, y, tp.kfun(c["kernel"], c["params"]), c["epsilon"])
    see("golden f_hat", nerr(gc.f_hat, g.get(c, "f_hat"))); see("golden logq", nerr([gc.logq], g.get(c, "logq")))
    see("golden diagL", nerr(np.diag(gc.L), g.get(c, "diagL")))
    fs, vf = gc.predict_latent(Xs)
    see("golden fs_bar", nerr(fs, g.get(c, "fs_bar"))); see("golden Vfs", nerr(vf, g.get(c, "Vfs")))
rng = np.random.default_rng(4)
X = rng.uniform(-1, 1, (3, 600)); y = np.sign(X.sum(0) + 0.2 * rng.normal(size=600)); y[y == 0] = 1.0
Xs = rng.uniform(-1, 1, (3, 41))
oc = orc.gpc_fit(orc.SQREXP, [1.0], X, y, 1e-5, divergence_stop=False)
gc = GPC(X, y, cov_func(sqrexp, l=1.0), 1e-5, reference_stop=False)
see("oracle f_hat", nerr(gc.f_hat, oc["f_hat"])); see("oracle logq", abs(gc.logq - oc["logq"]) / abs(oc["logq"])); see("oracle L", nerr(gc.L, oc["L"]))
fs, vf = gc.predict_latent(Xs)
ofs, ovf = orc.gpc_predict_latent(orc.SQREXP, [1.0], X, y, oc["f_hat"], oc["L"], Xs)
see("oracle fs_bar", nerr(fs, ofs)); see("oracle Vfs", nerr(vf, ovf))
print(json.dumps(worst, indent=1))
