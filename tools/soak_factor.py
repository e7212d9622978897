"""Two host threads, each with its own context, factor the same matrix again and again through gprc_dev_factor_all and compare every
word with their first result: where (panel, 128-row block, 128-column block) does a factorisation first differ, and what is info?
    python tools/soak_factor.py [reps] [n0] [n1]"""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import gprc_amd
from gprc_amd import _native as nat
from gprc_amd.distributed import Geometry
L = nat.lib()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
sizes = [int(a) for a in sys.argv[2:4]] or [2600, 4100]
NULL = None


def worker(i, n):
    st = torch.cuda.Stream()
    ctx = nat.Context(0, st.cuda_stream)
    g = Geometry(n)
    rng = np.random.default_rng(10 + i)
    X = torch.from_numpy(np.ascontiguousarray(rng.uniform(-1, 1, (n, 3)))).cuda()
    par, pp, npar = nat.params_array([0.7 + 0.1 * i])
    with torch.cuda.stream(st):
        K = torch.zeros(g.packed_size, dtype=torch.float64, device="cuda")
        st.synchronize()
        for p in range(g.P):
            nat.check(L.gprc_dev_fill_panel(ctx.handle, 3, pp, npar, X.data_ptr(), 3, n, g.n_pad, 0.1, K.data_ptr(), p))
        st.synchronize()
        ref = None
        bad = 0
        for r in range(reps):
            a = K.clone(); w = torch.zeros(g.winv_size, dtype=torch.float64, device="cuda"); info = torch.zeros(4, dtype=torch.int32, device="cuda")
            st.synchronize()
            nat.check(L.gprc_dev_factor_all(ctx.handle, a.data_ptr(), g.n_pad, w.data_ptr(), info.data_ptr(), NULL))
            st.synchronize()
            if ref is None:
                ref = a.clone(); refinfo = int(info[0])
                continue
            if not torch.equal(a, ref) or int(info[0]) != refinfo:
                bad += 1
                diff = (a != ref) | (torch.isnan(a) != torch.isnan(ref))
                idx = torch.nonzero(diff).flatten()
                first = int(idx[0]) if len(idx) else -1
                # locate: packed layout, panel p at offset NB * (p n_pad - NB p (p - 1) / 2), column-major with ld = n_pad - p NB
                where = "?"
                blocks = set()
                for e in idx[:: max(1, len(idx) // 2000)].tolist():
                    for p in range(g.P):
                        off = 512 * (p * g.n_pad - 512 * p * (p - 1) // 2); ld = g.n_pad - p * 512
                        if off <= e < off + ld * 512:
                            c, rr = divmod(e - off, ld)
                            blocks.add((p, rr // 128, c // 128))
                            break
                print(f"thread {i} n={n} rep {r}: info {int(info[0])}, {len(idx)} words differ; (panel, row block, column block) touched: {sorted(blocks)[:24]}", flush=True)
        print(f"thread {i} n={n}: {bad} of {reps - 1} factorisations differ from the first", flush=True)
    ctx.close()


ts = [threading.Thread(target=worker, args=(i, n)) for i, n in enumerate(sizes)]
[t.start() for t in ts]; [t.join() for t in ts]
