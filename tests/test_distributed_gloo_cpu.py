"""The N > 1 path on CPU: world_size 2 (and 3) with the gloo backend.  The driver under test is the product
code gprc_amd.distributed.DistributedGPR; the per-stage arithmetic is injected from tests/dist_ops_oracle.py.
Checks: every rank ends with the full replicated factor, results equal the single-process oracle, each panel
is filled/updated only by its owner, and the look-ahead order (panel p+1 updated and factored before the rest
of update p) is respected."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, TOL, nerr

sys.path.insert(0, os.path.join(ROOT, "tests"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _problem(n, d, ns, seed=11):
    rng = np.random.default_rng(seed)
    X = rng.uniform(-1, 1, (n, d))       # point-major
    y = 0.1 * (X ** 3).sum(1) + rng.normal(0, 0.1, n)
    Xs = rng.uniform(-1, 1, (ns, d))
    return X, y, Xs


def _worker(rank, world, port, n, d, ns, out_dir, bcast="broadcast"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    # every rank is a process of its own on the same few cores: without a cap each one starts a BLAS / OpenMP pool as wide as the
    # machine (8 ranks x 8 threads on 8 cores took 5 minutes for n = 4200)
    threads = max(1, (os.cpu_count() or 1) // world)
    torch.set_num_threads(threads)
    from threadpoolctl import threadpool_limits
    threadpool_limits(threads)
    if bcast == "whole_panel":                               # the unpipelined exchange: factor the panel, then one broadcast
        os.environ["GPRC_PIPE_BCAST"] = "0"
        bcast = "broadcast"
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import gprc_amd  # noqa: F401
    from gprc_amd.distributed import DistributedGPR, TorchComm
    from dist_ops_oracle import OracleOps
    from oracle import oracle as orc
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        X, y, Xs = _problem(n, d, ns)
        ops = OracleOps(orc.SQREXP, [1.0], d, n, 0.1)
        comm = TorchComm(strategy=bcast)
        if bcast == "auto":                                  # the timing is arbitrary on CPU: only the protocol is checked
            comm.calibrate(lambda c: torch.zeros(c, dtype=torch.float64), lambda: None, count=1 << 18, reps=1)
            assert comm.calibration["agree"] and comm.choice in ("broadcast", "scatter_allgather")
        eng = DistributedGPR(ops, comm)
        g = ops.geom
        ypad = np.zeros(g.n_pad)
        ypad[:n] = y
        Xd, yd = ops.from_host(X), ops.from_host(ypad)
        info = eng.fit(Xd, yd)
        lo, hi = eng.slice_bounds(ns, world)[rank]
        mean, var = ops.zeros(max(hi - lo, 1)), ops.zeros(max(hi - lo, 1))
        eng.predict_local(Xd, yd, ops.from_host(Xs[lo:hi] if hi > lo else np.zeros((1, d))), hi - lo, mean, var)
        sizes = [b - a for a, b in eng.slice_bounds(ns, world)]
        mean_all = comm.gather_concat(mean[: hi - lo], sizes)
        var_all = comm.gather_concat(var[: hi - lo], sizes)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), info=info, L=ops.dense_L(eng.packed)[:n, :n], alpha=eng.alpha.numpy()[:n],
                 logp=float(eng.scal[0]), mean=mean_all.numpy(), var=var_all.numpy(),
                 log=np.array([repr(e) for e in ops.log]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,bcast", [(2, 1300, "broadcast"), (3, 2100, "broadcast"), (2, 1300, "scatter_allgather"),
                                           (4, 2100, "scatter_allgather"), (2, 1300, "auto"), (4, 700, "broadcast"),
                                           (2, 1300, "whole_panel"), (4, 2600, "scatter_allgather"), (3, 2600, "whole_panel"),
                                           (8, 3600, "broadcast"), (8, 3600, "scatter_allgather")])
def test_block_cyclic_fit_and_sliced_predict(tmp_path, monkeypatch, world, n, bcast):
    """bcast: how a factored panel reaches the other ranks -- one rooted broadcast, or scatter + all-gather (the
    large-message form for point-to-point links); "auto" runs the calibration that picks one.  Same results.
    world = 8, n = 3600: the rank count BASELINE.json config 4 is quoted on (8 panels, one per rank: the smallest problem that gives every
    rank a panel -- eight processes share this container's eight cores); rank 0 owning a second panel is what 4 ranks x 6 panels cover."""
    from oracle import oracle as orc
    d, ns = 3, (3 if n < 1000 else 37)   # n = 700: 2 panels and 3 test points on 4 ranks -- ranks that own nothing
    port = _free_port()
    for var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):   # read when the children load their libraries
        monkeypatch.setenv(var, str(max(1, (os.cpu_count() or 1) // world)))
    mp.spawn(_worker, args=(world, port, n, d, ns, str(tmp_path), bcast), nprocs=world, join=True)
    X, y, Xs = _problem(n, d, ns)
    ref = orc.gpr_fit(orc.SQREXP, [1.0], X.T, y, 0.1)
    mr, vr = orc.gpr_predict(orc.SQREXP, [1.0], X.T, ref["L"], ref["alpha"], Xs.T)
    outs = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]
    P = -(-n // 512)
    for r, o in enumerate(outs):
        assert int(o["info"]) == 0
        assert nerr(o["L"], ref["L"]) <= TOL                 # replicated factor on every rank
        assert nerr(o["alpha"], ref["alpha"]) <= TOL
        assert abs(float(o["logp"]) - ref["logp"]) <= TOL * abs(ref["logp"])
        assert nerr(o["mean"], mr) <= TOL and nerr(o["var"], vr) <= TOL
        log = [eval(e) for e in o["log"]]
        fills = [e[1] for e in log if e[0] == "fill"]
        assert fills == list(range(r, P, world))              # F1: own panels only
        assert all(e[1] % world == r for e in log if e[0] == "factor")
        assert [e[1] for e in log if e[0] == "fwd"] == list(range(P))   # the forward solve rides along, one step per panel
        assert all(e[2] % world == r for e in log if e[0] == "update")
        # look-ahead: when this rank owns panel p+1, its update by panel p and its factorisation come before
        # the rest of trailing update p
        for p in range(P - 1):
            if (p + 1) % world != r:
                continue
            idx_fac = log.index(("factor", p + 1, True))
            idx_upd = log.index(("update", p, p + 1, True))
            rest = [i for i, e in enumerate(log) if e[0] == "update" and e[1] == p and e[2] > p + 1]
            assert idx_upd < idx_fac and all(i > idx_fac for i in rest)
            if bcast != "whole_panel":                        # pipelined exchange: a hand-over to the comm stream per quarter
                # (the "factor" entry is logged by the last quarter's factor step; its hand-over follows immediately)
                assert sum(1 for e in log[idx_upd:idx_fac + 2] if e == ("comm_after_side",)) == 4
    # every panel is updated by every earlier panel exactly once, somewhere
    ups = sorted((e[1], e[2]) for o in outs for e in map(eval, o["log"]) if e[0] == "update")
    assert ups == [(p, q) for p in range(P) for q in range(p + 1, P)]


def test_slice_bounds_cover_everything():
    from gprc_amd.distributed import DistributedGPR, owned_after
    for ns in (0, 1, 7, 64, 65536):
        for w in (1, 2, 3, 8):
            b = DistributedGPR.slice_bounds(ns, w)
            assert b[0][0] == 0 and b[-1][1] == ns and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
    assert owned_after(0, 0, 2) == 2 and owned_after(0, 1, 2) == 1 and owned_after(3, 1, 4) == 5 and owned_after(3, 0, 4) == 4
