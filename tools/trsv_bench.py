"""One vector solve (forward + backward timed separately) with the packed factor of an n x n sqexp kernel matrix.
    python tools/trsv_bench.py 8192 16384 65536"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import gprc_amd
from gprc_amd import _native as nat
from gprc_amd.distributed import Geometry
L = nat.lib()
ctx = nat.Context(0, torch.cuda.current_stream().cuda_stream)
for n in [int(a) for a in sys.argv[1:]] or [8192, 16384]:
    rng = np.random.default_rng(1)
    X = torch.from_numpy(np.ascontiguousarray(rng.uniform(-1, 1, (n, 8)))).cuda()
    g = Geometry(n)
    par, pp, npar = nat.params_array([1.0])
    a = torch.zeros(g.packed_size, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    for p in range(g.P):
        nat.check(L.gprc_dev_fill_panel(ctx.handle, 3, pp, npar, X.data_ptr(), 8, n, g.n_pad, 0.1, a.data_ptr(), p))
    w = torch.zeros(g.winv_size, dtype=torch.float64, device="cuda"); info = torch.zeros(4, dtype=torch.int32, device="cuda")
    inv = torch.zeros(int(L.gprc_solve_inv_size(g.n_pad)), dtype=torch.float64, device="cuda")
    work = torch.zeros(g.trsv_work, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    nat.check(L.gprc_dev_factor_all(ctx.handle, a.data_ptr(), g.n_pad, w.data_ptr(), info.data_ptr(), inv.data_ptr()))
    torch.cuda.synchronize()
    assert int(info[0]) == 0
    b0 = torch.from_numpy(rng.normal(size=g.n_pad)).cuda()
    best = [1e9, 1e9]
    for rep in range(6):
        x = b0.clone(); torch.cuda.synchronize()
        for tr in (0, 1):
            t0 = time.perf_counter()
            nat.check(L.gprc_dev_trsv(ctx.handle, a.data_ptr(), inv.data_ptr(), g.n_pad, x.data_ptr(), tr, work.data_ptr()))
            torch.cuda.synchronize()
            if rep: best[tr] = min(best[tr], time.perf_counter() - t0)
    gb = 8 * 0.5 * n * n / 1e9
    assert bool(torch.isfinite(x).all())
    print(f"n={n} forward {best[0] * 1e3:.3f} ms ({gb / best[0] / 1e3:.2f} TB/s)  backward {best[1] * 1e3:.3f} ms ({gb / best[1] / 1e3:.2f} TB/s)", flush=True)
    del a, inv, w
