"""gprc_dev_factor_all alone (packed sqexp kernel matrix resident, refilled by a device copy each repetition) under the schedule the
environment selects: milliseconds and TFLOP/s (n^3/3) per size.
    python tools/factor_bench.py 8192 16384 24576            # default schedule
    GPRC_SERVICE=0 python tools/factor_bench.py ...   # grouped left-looking, one fused launch per panel
    GPRC_BENCH_INV=1 ...                              # also the explicit diagonal-block inverses the vector solves need (what a fit asks for)"""
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
    K = torch.zeros(g.packed_size, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()                             # the library's stream is not ordered with torch's
    for p in range(g.P):
        nat.check(L.gprc_dev_fill_panel(ctx.handle, 3, pp, npar, X.data_ptr(), 8, n, g.n_pad, 0.1, K.data_ptr(), p))
    torch.cuda.synchronize()
    a = torch.empty_like(K); w = torch.zeros(g.winv_size, dtype=torch.float64, device="cuda"); info = torch.zeros(4, dtype=torch.int32, device="cuda")
    inv = torch.empty(int(L.gprc_solve_inv_size(g.n_pad)), dtype=torch.float64, device="cuda") if os.environ.get("GPRC_BENCH_INV") else None
    best = 1e9
    for rep in range(6):
        a.copy_(K); torch.cuda.synchronize()
        t0 = time.perf_counter()
        nat.check(L.gprc_dev_factor_all(ctx.handle, a.data_ptr(), g.n_pad, w.data_ptr(), info.data_ptr(), inv.data_ptr() if inv is not None else None))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if rep: best = min(best, dt)
    if int(info[0]) < 0:                                  # a device-side wait gave up: who, and on what (kernels_chol.hip: wait_diag)
        import ctypes
        rec = (ctypes.c_int * 392)(); L.gprc_prof_wait_timeout(rec, 392)
        print(f"n={n} info={int(info[0])} waits that gave up [site (+10: bystander), workgroup, grid, needed, saw, word, threads, sy]:", flush=True)
        for k in range(min(rec[0], 48)): print("   ", list(rec)[8 * (k + 1): 8 * (k + 2)], flush=True)
        continue
    assert int(info[0]) == 0, int(info[0])
    print(f"n={n} factor_all {best * 1e3:.3f} ms  {n ** 3 / 3 / best * 1e-12:.2f} TFLOP/s", flush=True)
