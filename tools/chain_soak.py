"""Repeats gprc_dev_factor_all over a few sizes and compares every factor with the first one of its size bit for bit; stops at the first
device-side wait timeout and prints who gave up (kernels_chol.hip: wait_diag).
    GPRC_CHAIN_SPLIT=1 python tools/chain_soak.py <rounds> [sizes...]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import gprc_amd
from gprc_amd import _native as nat
from gprc_amd.distributed import Geometry
L = nat.lib()
ctx = nat.Context(0, torch.cuda.current_stream().cuda_stream)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 50
sizes = [int(a) for a in sys.argv[2:]] or [16384, 8192, 12288, 20480]
state = {}
for n in sizes:
    rng = np.random.default_rng(1)
    X = torch.from_numpy(np.ascontiguousarray(rng.uniform(-1, 1, (n, 8)))).cuda()
    g = Geometry(n)
    par, pp, npar = nat.params_array([1.0])
    K = torch.zeros(g.packed_size, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    for p in range(g.P):
        nat.check(L.gprc_dev_fill_panel(ctx.handle, 3, pp, npar, X.data_ptr(), 8, n, g.n_pad, 0.1, K.data_ptr(), p))
    torch.cuda.synchronize()
    state[n] = dict(g=g, K=K, a=torch.empty_like(K), w=torch.zeros(g.winv_size, dtype=torch.float64, device="cuda"),
                    inv=torch.empty(int(L.gprc_solve_inv_size(g.n_pad)), dtype=torch.float64, device="cuda"), ref=None)
info = torch.zeros(4, dtype=torch.int32, device="cuda")
t0 = time.time(); count = 0; bad = 0
for r in range(rounds):
    for n in sizes:
        st = state[n]
        st["a"].copy_(st["K"]); info.zero_(); torch.cuda.synchronize()
        nat.check(L.gprc_dev_factor_all(ctx.handle, st["a"].data_ptr(), st["g"].n_pad, st["w"].data_ptr(), info.data_ptr(), st["inv"].data_ptr()))
        torch.cuda.synchronize()
        count += 1
        if int(info[0]) < 0:
            rec = (ctypes.c_int * 392)(); L.gprc_prof_wait_timeout(rec, 392)
            print(f"round {r} n={n}: info={int(info[0])}; waits that gave up [site (+10: bystander), workgroup, grid, needed, saw, word, threads, sy]:", flush=True)
            for k in range(min(rec[0], 48)): print("   ", list(rec)[8 * (k + 1): 8 * (k + 2)], flush=True)
            sys.exit(3)
        assert int(info[0]) == 0, int(info[0])
        if st["ref"] is None:
            st["ref"] = (st["a"].clone(), st["w"].clone(), st["inv"].clone())
        elif not (torch.equal(st["a"], st["ref"][0]) and torch.equal(st["w"], st["ref"][1]) and torch.equal(st["inv"], st["ref"][2])):
            bad += 1
            print(f"round {r} n={n}: factor differs from the first one", flush=True)
    if r % 10 == 9:
        print(f"round {r + 1}: {count} factorisations, {bad} differing, {time.time() - t0:.0f} s", flush=True)
print(f"done: {count} factorisations, {bad} differing")
sys.exit(1 if bad else 0)
