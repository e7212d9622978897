"""Where the persistent sweep kernel's workgroups spend their time (library built with -DGPRC_SWEEP_PROF, GPRC_LIB_SUFFIX=_sprof):
per item the mean microseconds in ticket take / dependency wait / tile / publish, and per strip.
    GPRC_LIB_SUFFIX=_sprof python tools/sweep_prof.py 8192 16384"""
import ctypes as C, os, sys, time
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
    K = torch.zeros(g.packed_size, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    for p in range(g.P):
        nat.check(L.gprc_dev_fill_panel(ctx.handle, 3, pp, npar, X.data_ptr(), 8, n, g.n_pad, 0.1, K.data_ptr(), p))
    torch.cuda.synchronize()
    a = torch.empty_like(K); w = torch.zeros(g.winv_size, dtype=torch.float64, device="cuda"); info = torch.zeros(4, dtype=torch.int32, device="cuda")
    out = (C.c_uint64 * 8)()
    for rep in range(3):
        a.copy_(K); torch.cuda.synchronize()
        L.gprc_debug_sweep_prof(out, 1)
        t0 = time.perf_counter()
        nat.check(L.gprc_dev_factor_all(ctx.handle, a.data_ptr(), g.n_pad, w.data_ptr(), info.data_ptr(), None))
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
    L.gprc_debug_sweep_prof(out, 0)
    v = [float(x) for x in out]
    tiles, strips = max(v[5], 1.0), max(v[6], 1.0)
    print(f"n={n}: factor {ms:.2f} ms; {int(v[5])} tiles, {int(v[6])} strips; per tile: take {v[0] / 100 / (tiles + strips):.2f} us, wait {v[1] / 100 / tiles:.2f}, "
          f"tile {v[2] / 100 / tiles:.2f}, publish {v[3] / 100 / tiles:.2f}; per strip {v[4] / 100 / strips:.1f} us; workgroup-time total {v[7] / 100 / 1e3:.1f} ms "
          f"(take {v[0] / v[7]:.3f} wait {v[1] / v[7]:.3f} tile {v[2] / v[7]:.3f} publish {v[3] / v[7]:.3f} strip {v[4] / v[7]:.3f})", flush=True)
