"""Where the split panel chain spends its time (library built with -DGPRC_CHAIN_PROF=<panel>, GPRC_LIB_SUFFIX=_cprof): stamps of chain
helper 0 and of the factor role during that panel of one factorisation at n, in microseconds from the panel's first W.
    GPRC_EXTRA_FLAGS=-DGPRC_CHAIN_PROF=12 GPRC_LIB_SUFFIX=_cprof bash gaussian-process-regression_amd/csrc/build.sh
    GPRC_CHAIN_SPLIT=1 GPRC_LIB_SUFFIX=_cprof python tools/chain_prof.py 8192"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import gprc_amd
from gprc_amd import _native as nat
from gprc_amd.distributed import Geometry
L = nat.lib()
ctx = nat.Context(0, torch.cuda.current_stream().cuda_stream)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
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
inv = torch.empty(int(L.gprc_solve_inv_size(g.n_pad)), dtype=torch.float64, device="cuda")
out = (C.c_uint64 * 64)()
names = ["W seen", "ops in", "mfma done", "counted", "S seen", "ops in", "mfma done", "counted"]
for rep in range(4):
    a.copy_(K); torch.cuda.synchronize()
    nat.check(L.gprc_dev_factor_all(ctx.handle, a.data_ptr(), g.n_pad, w.data_ptr(), info.data_ptr(), inv.data_ptr()))
    torch.cuda.synchronize()
    assert L.gprc_debug_chain_prof(out) == 0
    v = np.array(list(out), dtype=np.int64)
    t0 = v[34]                      # W_0 published
    ev = [(v[32 + 4 * j + k], f"factor j={j} " + ["U seen", "potf2 done", "W published"][k]) for j in range(4) for k in range(3)]
    ev += [(v[8 * j + k], f"  helper j={j} " + names[k]) for j in range(4) for k in range(8) if v[8 * j + k]]
    ev.sort()
    print(f"rep {rep}  XCC ids: factor role {v[56]}, helpers {list(v[57:61])}")
    prev = None
    for t, nm in ev:
        print(f"  {(t - t0) / 100.0:8.2f}  {'' if prev is None else f'+{(t - prev) / 100.0:6.2f}'}  {nm}")
        prev = t
