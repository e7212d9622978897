"""Replays ONE rank's share of an N-rank fit + predict step on a single MI355X, with the exchange step stubbed out:
panels this rank would RECEIVE are preloaded from a reference factorisation (as if their broadcast had arrived), so
every kernel the rank launches sees valid data and real sizes, and what is timed is that rank's compute + launch
critical path -- the step time an N-GPU run would reach with infinitely fast links.  (The real N-GPU numbers are the
driver's; this tool is for finding per-rank bottlenecks.)

    python tools/rank_replay.py [--n 65536] [--worlds 1,2,4,8] [--ranks all|0]     -> one JSON line
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import gprc_amd
from gprc_amd import _native as nat
from gprc_amd.distributed import DistributedGPR, HipOps, SingleComm
import bench


class ReplayComm:
    """world > 1 but nothing moves: the received panels are already in place."""

    def __init__(self, rank, world):
        self.rank, self.world = rank, world

    def broadcast(self, t, src):
        pass

    def min_positive(self, value):
        return value

    def barrier(self):
        pass


ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=65536)
ap.add_argument("--d", type=int, default=8)
ap.add_argument("--nstar", type=int, default=65536)
ap.add_argument("--worlds", default="1,2,4,8")
ap.add_argument("--ranks", default="0")
ap.add_argument("--reps", type=int, default=2)
args = ap.parse_args()
n, d = args.n, args.d
X, y, Xs = bench.synth(n, d, None if args.nstar == 65536 and d == 8 else args.nstar)
ns = Xs.shape[0]
kid, params = bench.KERNEL_IDS["sqrexp"], [1.0]

# reference factor on one rank
ops = HipOps(0, kid, params, d, n, 0.1)
ref = DistributedGPR(ops, SingleComm())
g = ops.geom
ypad = np.zeros(g.n_pad); ypad[:n] = y
Xd, yd = ops.from_host(X), ops.from_host(ypad)
assert ref.fit(Xd, yd) == 0
torch.cuda.synchronize()
with ops.on(False):
    L_ref, W_ref = ref.packed.clone(), ref.winv.clone()
torch.cuda.synchronize()
del ref
torch.cuda.empty_cache()

rows = []
for world in [int(w) for w in args.worlds.split(",")]:
    ranks = range(world) if args.ranks == "all" else [int(r) for r in args.ranks.split(",") if int(r) < world]
    for rank in ranks:
        comm = ReplayComm(rank, world) if world > 1 else SingleComm()
        eng = DistributedGPR(ops, comm)
        lo, hi = eng.slice_bounds(ns, world)[rank]
        Xl = ops.from_host(Xs[lo:hi]); mean = ops.zeros(hi - lo); var = ops.zeros(hi - lo)
        fit_ms, pred_ms = [], []
        for rep in range(args.reps + 1):
            with ops.on(False):                                     # on the library's own stream, not torch's default one
                eng.packed.copy_(L_ref); eng.winv.copy_(W_ref)      # "received" panels; own panels are refilled by fit()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            info = eng.fit(Xd, yd)
            ops.synchronize()
            t1 = time.perf_counter()
            eng.predict_local(Xd, yd, Xl, hi - lo, mean, var)
            ops.synchronize()
            t2 = time.perf_counter()
            if rep:
                fit_ms.append((t1 - t0) * 1e3); pred_ms.append((t2 - t1) * 1e3)
        # the replayed rank reproduces the reference factor bit for bit (compared panel by panel: the whole buffer has
        # more than 2^31 elements at n = 65536)
        ok = info == 0 and all(bool(torch.equal(eng.packed[g.panel_slice(q)], L_ref[g.panel_slice(q)])) for q in range(g.P))
        rows.append({"world": world, "rank": rank, "fit_ms": round(min(fit_ms), 1), "predict_ms": round(min(pred_ms), 1),
                     "step_ms": round(min(fit_ms) + min(pred_ms), 1), "lookahead": eng.lookahead, "factor_bits_equal": ok})
        print(rows[-1], file=sys.stderr, flush=True)
        del eng, Xl, mean, var
        torch.cuda.empty_cache()
base = next(r["step_ms"] for r in rows if r["world"] == 1) if any(r["world"] == 1 for r in rows) else None
print(json.dumps({"what": "per-rank compute replay on one GPU, exchange stubbed (links infinitely fast)", "n": n, "d": d, "n_star": ns,
                  "rows": rows,
                  "compute_bound_speedup": {str(w): round(base / max(r["step_ms"] for r in rows if r["world"] == w), 2)
                                            for w in sorted({r["world"] for r in rows})} if base else None}))
