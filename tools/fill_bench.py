"""Kernel-fill rate by kernel (gprc_dev_fill_panel into a resident packed matrix): GB/s written.   python tools/fill_bench.py [n] [d]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import gprc_amd
from gprc_amd import _native as nat
from gprc_amd.distributed import Geometry
L = nat.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
d = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ctx = nat.Context(0, torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(1)
X = torch.from_numpy(np.ascontiguousarray(rng.uniform(-1, 1, (n, d)))).cuda()
g = Geometry(n)
K = torch.zeros(g.packed_size, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
for name, kid, params in (("sqrexp l=1", 3, [1.0]), ("rationalquadratic l=1 alpha=1.5 (rsqrt form)", 5, [1.0, 1.5]), ("rationalquadratic l=1 alpha=1.7 (exp/log form)", 5, [1.0, 1.7]),
                          ("gammaexp l=1 gamma=1.5", 4, [1.0, 1.5]), ("polynomial sigma=0.5 p=3", 2, [0.5, 3.0]), ("linear sigma=0.7", 1, [0.7])):
    par, pp, npar = nat.params_array(params)
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for p in range(g.P):
            nat.check(L.gprc_dev_fill_panel(ctx.handle, kid, pp, npar, X.data_ptr(), d, n, g.n_pad, 0.1, K.data_ptr(), p))
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print(f"{name}: {best * 1e3:.2f} ms for {g.packed_size * 8 / 1e9:.2f} GB = {g.packed_size * 8 / best / 1e9:.0f} GB/s", flush=True)
# the predict's cross fill K(X*, X) (unfused form: gprc_dev_fill_cross), 16384 x n chunk
m = 16384
Xs = torch.from_numpy(np.ascontiguousarray(rng.uniform(-1, 1, (m, d)))).cuda()
ld = m + 128
vt = torch.empty(ld * g.n_pad, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
for name, kid, params in (("sqrexp l=1", 3, [1.0]), ("rationalquadratic alpha=1.5", 5, [1.0, 1.5]), ("rationalquadratic alpha=1.7", 5, [1.0, 1.7])):
    par, pp, npar = nat.params_array(params)
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        nat.check(L.gprc_dev_fill_cross(ctx.handle, kid, pp, npar, Xs.data_ptr(), d, m, m, X.data_ptr(), n, g.n_pad, vt.data_ptr(), ld))
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print(f"cross fill {name}: {best * 1e3:.2f} ms for {m * g.n_pad * 8 / 1e9:.2f} GB = {m * g.n_pad * 8 / best / 1e9:.0f} GB/s", flush=True)
