"""Per-launch rate of the caller's-stream update kernels during gprc_dev_factor_all (factor service on) against the plain trailing
kernel on the same matrix and panel: python tools/update_bench.py [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gprc_amd
from gprc_amd import _native as nat
from gprc_amd.distributed import Geometry
L = nat.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
st = torch.cuda.Stream()
ctx = nat.Context(0, st.cuda_stream)
g = Geometry(n)
rng = np.random.default_rng(1)
with torch.cuda.stream(st):
    X = torch.from_numpy(np.ascontiguousarray(rng.uniform(-1, 1, (n, 8)))).cuda()
    par, pp, npar = nat.params_array([1.0])
    K = torch.zeros(g.packed_size, dtype=torch.float64, device="cuda")
    for p in range(g.P):
        nat.check(L.gprc_dev_fill_panel(ctx.handle, 3, pp, npar, X.data_ptr(), 8, n, g.n_pad, 0.1, K.data_ptr(), p))
    a = torch.empty_like(K); w = torch.zeros(g.winv_size, dtype=torch.float64, device="cuda"); info = torch.zeros(4, dtype=torch.int32, device="cuda")
    for rep in range(2):
        a.copy_(K); st.synchronize()
        L.gprc_prof_reset(); L.gprc_prof_enable(1)
        nat.check(L.gprc_dev_factor_all(ctx.handle, a.data_ptr(), g.n_pad, w.data_ptr(), info.data_ptr(), None))
        st.synchronize(); L.gprc_prof_enable(0)
    for name, r in nat.prof_summary().items():
        if r["count"]:
            print(f"service sweep  {name}: {r['count']} launches, {r['ms']:.3f} ms total, {r['flops'] / r['ms'] / 1e9:.2f} TFLOP/s")
    # the plain trailing kernel: panel p applied to everything behind it, for a few p (the data is a finished factor: timing only)
    for p in (0, g.P // 4, g.P // 2, 3 * g.P // 4):
        for _ in range(2):
            nat.check(L.gprc_dev_update_trailing(ctx.handle, a.data_ptr(), g.n_pad, p, p + 1, g.P, 1))
        st.synchronize()
        L.gprc_prof_reset(); L.gprc_prof_enable(1)
        for _ in range(6):
            nat.check(L.gprc_dev_update_trailing(ctx.handle, a.data_ptr(), g.n_pad, p, p + 1, g.P, 1))
        st.synchronize(); L.gprc_prof_enable(0)
        r = nat.prof_summary()["trailing_update"]
        print(f"plain trailing p={p}: {r['ms'] / r['count']:.3f} ms per launch, {r['flops'] / r['ms'] / 1e9:.2f} TFLOP/s")
    rnd = (torch.rand_like(a) - 0.5) * 0.02
    for _ in range(2):
        nat.check(L.gprc_dev_update_trailing(ctx.handle, rnd.data_ptr(), g.n_pad, 0, 1, g.P, 1))
    st.synchronize()
    L.gprc_prof_reset(); L.gprc_prof_enable(1)
    for _ in range(6):
        nat.check(L.gprc_dev_update_trailing(ctx.handle, rnd.data_ptr(), g.n_pad, 0, 1, g.P, 1))
    st.synchronize(); L.gprc_prof_enable(0)
    r = nat.prof_summary()["trailing_update"]
    print(f"plain trailing p=0 on uniform(-0.01, 0.01) data: {r['ms'] / r['count']:.3f} ms per launch, {r['flops'] / r['ms'] / 1e9:.2f} TFLOP/s")
