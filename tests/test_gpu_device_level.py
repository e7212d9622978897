"""Device-level entry points of the C ABI (gprc_dev_*), called directly on torch-owned device buffers: the alternative
schedules they offer are bit-identical to the basic one -- factor_panel in quarters, trailing updates by a range of
panels, the whole-sweep call, the solve in panel steps."""
import ctypes as C

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import gprc_amd  # noqa: E402,F401
from gprc_amd import _native as nat  # noqa: E402
from gprc_amd.distributed import Geometry  # noqa: E402

pytestmark = pytest.mark.gpu


def _filled(n, d=3, seed=31):
    L = nat.lib()
    rng = np.random.default_rng(seed)
    X = torch.from_numpy(np.ascontiguousarray(rng.uniform(-1, 1, (n, d)))).cuda()
    g = Geometry(n)
    ctx = nat.Context(0, torch.cuda.current_stream().cuda_stream)
    par, pp, npar = nat.params_array([0.6])
    packed = torch.zeros(g.packed_size, dtype=torch.float64, device="cuda")
    for p in range(g.P):
        nat.check(L.gprc_dev_fill_panel(ctx.handle, 3, pp, npar, X.data_ptr(), d, n, g.n_pad, 0.1, packed.data_ptr(), p))
    torch.cuda.synchronize()
    return L, ctx, g, packed


def _clone(t):
    """A copy that is COMPLETE before the library touches it: the context below runs on the library's own (non-blocking) stream, which
    is not ordered with torch's; without the synchronisation a large copy is still in flight when the first kernel starts (seen
    once at n = 20987: the copy overwrote the tail panels' first updates)."""
    c = t.clone()
    torch.cuda.synchronize()
    return c


def _new(g):
    r = (torch.zeros(g.winv_size, dtype=torch.float64, device="cuda"), torch.zeros(4, dtype=torch.int32, device="cuda"))
    torch.cuda.synchronize()
    return r


def _new_inv(g):
    r = torch.full((int(nat.lib().gprc_solve_inv_size(g.n_pad)),), float("nan"), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    return r


NULL = None


def test_sweep_variants_are_bit_identical():
    n = 2900                                             # 6 panels
    L, ctx, g, K = _filled(n)
    P = g.P

    def basic():                                         # factor_panel + one update per panel
        a = _clone(K); w, info = _new(g)
        for p in range(P):
            nat.check(L.gprc_dev_factor_panel(ctx.handle, a.data_ptr(), g.n_pad, p, w.data_ptr(), info.data_ptr()))
            if p + 1 < P:
                nat.check(L.gprc_dev_update_trailing(ctx.handle, a.data_ptr(), g.n_pad, p, p + 1, P, 1))
        torch.cuda.synchronize()
        return a, w, int(info[0])

    def quarters_and_ranges(batch):                      # factor in 4 x (part 1, part 2); far panels updated in batches
        a = _clone(K); w, info = _new(g)
        far_from = 0
        for p in range(P):
            if p > 0:                                    # bring panel p up to date, whatever has not been applied yet
                nat.check(L.gprc_dev_update_range(ctx.handle, a.data_ptr(), g.n_pad, far_from, p, p, p + 1, 1))
            for j in range(4):
                for part in (1, 2):
                    nat.check(L.gprc_dev_factor_subpanel(ctx.handle, a.data_ptr(), g.n_pad, p, j, part, w.data_ptr(), info.data_ptr()))
            if p + 1 - far_from >= batch or p + 2 >= P:
                nat.check(L.gprc_dev_update_range(ctx.handle, a.data_ptr(), g.n_pad, far_from, p + 1, p + 1, P, 1))
                far_from = p + 1
        torch.cuda.synchronize()
        return a, w, int(info[0])

    def whole():
        a = _clone(K); w, info = _new(g)
        nat.check(L.gprc_dev_factor_all(ctx.handle, a.data_ptr(), g.n_pad, w.data_ptr(), info.data_ptr(), NULL))
        torch.cuda.synchronize()
        return a, w, int(info[0])

    ref = basic()
    assert ref[2] == 0
    for name, got in (("batch 1", quarters_and_ranges(1)), ("batch 3", quarters_and_ranges(3)), ("factor_all", whole())):
        assert got[2] == 0 and torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), name
    with pytest.raises(nat.GprcError, match="behind the source range"):
        nat.check(L.gprc_dev_update_range(ctx.handle, _clone(K).data_ptr(), g.n_pad, 0, 3, 2, P, 1))
    ctx.close()


def test_grouped_left_looking_with_the_service_inside_the_groups(monkeypatch):
    """The whole-sweep call takes the panels in groups: a left-looking pass brings a group's columns up to date, inside the group
    the factor service runs (its launch restricted to the group: the block behind the group is left to that block's own pass).
    Any grouping gives the bits of the launch-per-panel sweep; n = 9100: 18 panels, groups of 1, 2-3, ~6 and all."""
    n = 9100
    L, ctx, g, K = _filled(n, seed=6)
    P = g.P
    a = _clone(K); w, info = _new(g)
    for p in range(P):
        nat.check(L.gprc_dev_factor_panel(ctx.handle, a.data_ptr(), g.n_pad, p, w.data_ptr(), info.data_ptr()))
        if p + 1 < P:
            nat.check(L.gprc_dev_update_trailing(ctx.handle, a.data_ptr(), g.n_pad, p, p + 1, P, 1))
    torch.cuda.synchronize()
    assert int(info[0]) == 0
    inv_ref = None
    for want in ("1", "500", "1500", "right"):
        monkeypatch.setenv("GPRC_FACTOR", want)
        b = _clone(K); w2, info2 = _new(g); inv2 = _new_inv(g)
        nat.check(L.gprc_dev_factor_all(ctx.handle, b.data_ptr(), g.n_pad, w2.data_ptr(), info2.data_ptr(), inv2.data_ptr()))
        torch.cuda.synchronize()
        assert int(info2[0]) == 0, want
        assert torch.equal(b, a) and torch.equal(w2, w), want
        blk = torch.arange(512, device="cuda") // 128
        written = (blk[None, :] <= blk[:, None])
        T = inv2.view(P, 512, 512)[:, written]
        if inv_ref is None:
            inv_ref = T
        assert torch.equal(T, inv_ref), want
    ctx.close()


def test_factor_service_over_many_panel_counts(monkeypatch):
    """Every panel count from 2 to 14 and a few larger ones, including 41 (n_pad = 20992: the first size that the default schedule
    takes in groups), each under the default grouping and under small groups: the whole-sweep call against the launch-per-panel
    sweep, bitwise.  (The tile enumeration of the service's update kernel, its ticketed head and the group boundaries all depend
    on the panel count.)"""
    sizes = [512 * P - 37 for P in range(2, 15)] + [512 * 19, 512 * 27 - 300, 512 * 41 - 5]
    for n in sizes:
        L, ctx, g, K = _filled(n, seed=n % 97)
        P = g.P
        a = _clone(K); w, info = _new(g)
        for p in range(P):
            nat.check(L.gprc_dev_factor_panel(ctx.handle, a.data_ptr(), g.n_pad, p, w.data_ptr(), info.data_ptr()))
            if p + 1 < P:
                nat.check(L.gprc_dev_update_trailing(ctx.handle, a.data_ptr(), g.n_pad, p, p + 1, P, 1))
        torch.cuda.synchronize()
        assert int(info[0]) == 0, n
        for want in (None, "400"):
            if want is None:
                monkeypatch.delenv("GPRC_FACTOR", raising=False)
            else:
                monkeypatch.setenv("GPRC_FACTOR", want)
            b = _clone(K); w2, info2 = _new(g)
            nat.check(L.gprc_dev_factor_all(ctx.handle, b.data_ptr(), g.n_pad, w2.data_ptr(), info2.data_ptr(), NULL))
            torch.cuda.synchronize()
            assert int(info2[0]) == 0, (n, want)
            assert torch.equal(b, a) and torch.equal(w2, w), (n, want)
            del b, w2
        ctx.close()
        del a, w, K
        torch.cuda.empty_cache()


@pytest.mark.parametrize("n", [600, 1100, 1536, 9100])
def test_factor_service_is_bit_identical_to_the_launch_per_panel_sweep(n):
    """gprc_dev_factor_all at these sizes is the factor service (one persistent launch carrying every panel's dependent chain,
    the rows of the next diagonal block and that block's update; the caller's stream the rest) -- against factor_panel +
    update_trailing per panel: every word of the factor and of the block inverses equal, twice in a row (the service's
    flags are per call).  2, 3, 3 (no padding) and 18 panels."""
    L, ctx, g, K = _filled(n, seed=5)
    P = g.P
    a = _clone(K); w, info = _new(g)
    for p in range(P):
        nat.check(L.gprc_dev_factor_panel(ctx.handle, a.data_ptr(), g.n_pad, p, w.data_ptr(), info.data_ptr()))
        if p + 1 < P:
            nat.check(L.gprc_dev_update_trailing(ctx.handle, a.data_ptr(), g.n_pad, p, p + 1, P, 1))
    torch.cuda.synchronize()
    assert int(info[0]) == 0
    for rep in range(2):
        b = _clone(K); w2, info2 = _new(g)
        inv2 = _new_inv(g) if rep else None      # with and without the explicit diagonal inverses riding along
        nat.check(L.gprc_dev_factor_all(ctx.handle, b.data_ptr(), g.n_pad, w2.data_ptr(), info2.data_ptr(), inv2.data_ptr() if rep else NULL))
        torch.cuda.synchronize()
        assert int(info2[0]) == 0
        assert torch.equal(b, a) and torch.equal(w2, w), (n, rep)
    # the inverses the service leaves are the ones gprc_dev_solve_prepare computes (same block arithmetic; entries below T's
    # block diagonal are never written: compare through a mask), and they invert the diagonal blocks
    inv1 = _new_inv(g)
    nat.check(L.gprc_dev_solve_prepare(ctx.handle, a.data_ptr(), w.data_ptr(), g.n_pad, inv1.data_ptr(), 0, P))
    torch.cuda.synchronize()
    T1 = inv1.view(P, 512, 512); T2 = inv2.view(P, 512, 512)               # [p, r, c] = T[c + r NB] = inv(L_pp)[r, c]
    blk = torch.arange(512, device="cuda") // 128
    written = (blk[None, :] <= blk[:, None])                                # inverse entry (r, c): block row >= block column
    assert torch.equal(T1[:, written], T2[:, written])
    for p in (0, P - 1):
        ld = g.n_pad - p * 512
        Lpp = torch.tril(a[g.panel_slice(p)].view(512, ld).t()[:512, :])
        inv_pp = torch.where(written, T1[p], torch.zeros((), dtype=torch.float64, device="cuda"))
        inv_pp = torch.tril(inv_pp)
        err = (inv_pp @ Lpp - torch.eye(512, dtype=torch.float64, device="cuda")).abs().max().item()
        assert err <= 1e-11, (p, err)
    ctx.close()


@pytest.mark.parametrize("n,reps", [(2100, 1), (600, 1), (9100, 2)])
def test_solve_in_panel_steps_is_bit_identical(n, reps):
    """gprc_dev_trsv (the whole solve: one launch per panel, the diagonal step through the explicit inverses of the diagonal
    blocks, spread over 16 workgroups of the panel's launch) against gprc_dev_trsv_step, the per-panel form the multi-rank sweep
    runs beside the factorisation: every word equal.  n = 600: two panels; n = 9100: 18 panels.  And it is a solve."""
    L, ctx, g, a = _filled(n, seed=32)
    w, info = _new(g)
    inv = _new_inv(g)
    nat.check(L.gprc_dev_factor_all(ctx.handle, a.data_ptr(), g.n_pad, w.data_ptr(), info.data_ptr(), inv.data_ptr()))
    rng = np.random.default_rng(1)
    b0 = torch.from_numpy(np.concatenate([rng.normal(size=n), np.zeros(g.n_pad - n)])).cuda()
    work = torch.zeros(g.trsv_work, dtype=torch.float64, device="cuda")
    work2 = torch.zeros(g.trsv_work, dtype=torch.float64, device="cuda")
    for transpose in (0, 1):
        steps = _clone(b0)
        order = range(g.P) if not transpose else range(g.P - 1, -1, -1)
        for p in order:
            nat.check(L.gprc_dev_trsv_step(ctx.handle, a.data_ptr(), inv.data_ptr(), g.n_pad, steps.data_ptr(), transpose, p, work2.data_ptr()))
        for _ in range(reps):
            whole = _clone(b0)
            nat.check(L.gprc_dev_trsv(ctx.handle, a.data_ptr(), inv.data_ptr(), g.n_pad, whole.data_ptr(), transpose, work.data_ptr()))
            torch.cuda.synchronize()
            assert torch.equal(whole, steps)
    # and it is a solve: L (L^T x) = b
    x = _clone(b0)
    nat.check(L.gprc_dev_trsv(ctx.handle, a.data_ptr(), inv.data_ptr(), g.n_pad, x.data_ptr(), 0, work.data_ptr()))
    nat.check(L.gprc_dev_trsv(ctx.handle, a.data_ptr(), inv.data_ptr(), g.n_pad, x.data_ptr(), 1, work.data_ptr()))
    torch.cuda.synchronize()
    from oracle import oracle as orc
    Xh = np.random.default_rng(32).uniform(-1, 1, (n, 3))
    Kh = orc.kernel_matrix(orc.SQREXP, [0.6], Xh.T, Xh.T) + 0.1 * np.eye(n)
    assert np.max(np.abs(Kh @ x.cpu().numpy()[:n] - b0.cpu().numpy()[:n])) <= 1e-10 * np.abs(b0.cpu().numpy()).max() * n
    with pytest.raises(nat.GprcError, match="bad arguments"):
        nat.check(L.gprc_dev_trsv(ctx.handle, a.data_ptr(), NULL, g.n_pad, x.data_ptr(), 0, work.data_ptr()))
    ctx.close()


def test_solve_in_panel_steps_is_bit_identical_at_a_large_size():
    """From n_pad = 32768 on, the forward product takes the rows far from the diagonal 128 per workgroup (the better HBM access
    pattern) instead of 32: another thread layout of the same summation tree.  Whole solve against per-panel steps at n = 33000
    (65 panels): every word equal, both directions; and L^T (L x) reproduces the right-hand side's solve (residual through the
    packed factor itself: x from the solves, then forward-substituted back by the per-step launches of the OTHER direction)."""
    n = 33000
    L, ctx, g, a = _filled(n, seed=33)
    w, info = _new(g)
    inv = _new_inv(g)
    nat.check(L.gprc_dev_factor_all(ctx.handle, a.data_ptr(), g.n_pad, w.data_ptr(), info.data_ptr(), inv.data_ptr()))
    torch.cuda.synchronize()
    assert int(info[0]) == 0
    b0 = torch.from_numpy(np.concatenate([np.random.default_rng(2).normal(size=n), np.zeros(g.n_pad - n)])).cuda()
    work = torch.zeros(g.trsv_work, dtype=torch.float64, device="cuda")
    work2 = torch.zeros(g.trsv_work, dtype=torch.float64, device="cuda")
    for transpose in (0, 1):
        steps = _clone(b0)
        for p in (range(g.P) if not transpose else range(g.P - 1, -1, -1)):
            nat.check(L.gprc_dev_trsv_step(ctx.handle, a.data_ptr(), inv.data_ptr(), g.n_pad, steps.data_ptr(), transpose, p, work2.data_ptr()))
        whole = _clone(b0)
        nat.check(L.gprc_dev_trsv(ctx.handle, a.data_ptr(), inv.data_ptr(), g.n_pad, whole.data_ptr(), transpose, work.data_ptr()))
        torch.cuda.synchronize()
        assert torch.equal(whole, steps), transpose
        assert bool(torch.isfinite(whole).all())
    ctx.close()


def test_k_chunked_left_looking_passes_are_bit_identical():
    """GPRC_KCHUNK cuts the long-K left-looking passes (predict solve and Cholesky trailing update) into several launches
    over K ranges; same products in the same order, so the factor and the prediction must not change by a bit.  The
    switch is read once per process: the chunked run happens in a child process and the two digests are compared.
    n = 3100 (7 panels) with GPRC_FACTOR=1 (every panel its own left-looking group) and GPRC_SOLVE=left, so passes with
    K up to 6 panels exist and chunks of 1 and 4 panels cut them unevenly."""
    import hashlib
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import hashlib, numpy as np\n"
        "from gprc_amd import GPR, cov_func, sqrexp\n"
        "rng = np.random.default_rng(41)\n"
        "X = rng.uniform(-1, 1, (3, 3100)); y = rng.normal(size=3100); Xs = rng.uniform(-1, 1, (3, 900))\n"
        "g = GPR(X, y, 0.1, cov_func(sqrexp, l=0.7))\n"
        "h = hashlib.sha256(); [h.update(np.ascontiguousarray(a).tobytes()) for a in (g.alpha, g.L, g.predict(Xs))]\n"
        "print('DIGEST', h.hexdigest())\n") % (os.path.dirname(here), here)
    digests = {}
    for kc in ("0", "1", "4"):
        env = dict(os.environ, GPRC_KCHUNK=kc, GPRC_FACTOR="1", GPRC_SOLVE="left")
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        digests[kc] = [ln.split()[1] for ln in r.stdout.splitlines() if ln.startswith("DIGEST")][0]
    assert digests["1"] == digests["0"] and digests["4"] == digests["0"], digests


def test_256x128_macro_tile_is_bit_identical():
    """GPRC_TILE256=1 runs the predict's left-looking passes on 256 x 128 macro-tiles (8 waves sharing one B strip) instead of
    128 x 128 tiles: same k order per output element, so not a bit may change.  Child processes (the switch is read once);
    ns = 1024 rows (m_pad a multiple of 256), GPRC_SOLVE=left / 2 so that left-looking passes with K up to 5 panels run."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import hashlib, numpy as np\n"
        "from gprc_amd import GPR, cov_func, sqrexp\n"
        "rng = np.random.default_rng(43)\n"
        "X = rng.uniform(-1, 1, (3, 2900)); y = rng.normal(size=2900); Xs = rng.uniform(-1, 1, (3, 1024))\n"
        "g = GPR(X, y, 0.1, cov_func(sqrexp, l=0.7))\n"
        "h = hashlib.sha256(); [h.update(np.ascontiguousarray(a).tobytes()) for a in (g.alpha, g.predict(Xs), g.predict(Xs[:, :512]))]\n"
        "print('DIGEST', h.hexdigest())\n") % (os.path.dirname(here), here)
    digests = {}
    for t256, solve in (("0", "left"), ("1", "left"), ("1", "2")):
        env = dict(os.environ, GPRC_TILE256=t256, GPRC_SOLVE=solve)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        digests[(t256, solve)] = [ln.split()[1] for ln in r.stdout.splitlines() if ln.startswith("DIGEST")][0]
    assert len(set(digests.values())) == 1, digests


def test_fused_in_panel_predict_solve_is_bit_identical():
    """The predict's in-panel solve runs as one launch per panel (solve_panel_fused_kernel: a strip's four sub-steps back to
    back); GPRC_SOLVE_PANEL=steps selects the seven-launch form.  Same tiles in the same order per strip: identical bits,
    for the pointwise predict (with the fused sums of squares), the full covariance and GPC's latent predict."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import hashlib, numpy as np\n"
        "from gprc_amd import GPR, GPC, cov_func, sqrexp\n"
        "rng = np.random.default_rng(47)\n"
        "X = rng.uniform(-1, 1, (3, 1900)); y = rng.normal(size=1900); Xs = rng.uniform(-1, 1, (3, 700))\n"
        "g = GPR(X, y, 0.1, cov_func(sqrexp, l=0.7))\n"
        "gc = GPC(X[:, :600], np.sign(X[0, :600] + 0.3 * X[1, :600]), cov_func(sqrexp, l=0.8), 1e-5, reference_stop=False)\n"
        "parts = [g.predict(Xs), g.predict(Xs[:, :150], pointwise_var=False)[1], np.column_stack(gc.predict_latent(Xs))]\n"
        "h = hashlib.sha256(); [h.update(np.ascontiguousarray(a).tobytes()) for a in parts]\n"
        "print('DIGEST', h.hexdigest())\n") % (os.path.dirname(here), here)
    digests = {}
    for mode in ("fused", "steps"):
        env = dict(os.environ, GPRC_SOLVE_PANEL=mode)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        digests[mode] = [ln.split()[1] for ln in r.stdout.splitlines() if ln.startswith("DIGEST")][0]
    assert digests["fused"] == digests["steps"], digests


def test_two_contexts_factor_many_times_and_every_word_agrees():
    """Two host threads, each with its own context and stream, factor again and again at the same time on the one GPU.  Their factor
    services do not start in step then (a service workgroup needs a whole CU and gets it when the other context's kernels let go
    of one), which is what exposed the round-2 look-ahead counter -- one sum over the sub-steps of the four look-ahead strips: a
    strip that started late was read by the next diagonal block's update before it had finished (1 factorisation in ~1000 wrong,
    profiles/r03_la_counter_race.txt).  400 factorisations per thread, every word of factor and block inverses against the first."""
    import threading
    L = nat.lib()
    errs, done = [], []

    def work(i, n):
        try:
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
                for r in range(400):
                    a = K.clone()
                    w = torch.zeros(g.winv_size, dtype=torch.float64, device="cuda")
                    info = torch.zeros(4, dtype=torch.int32, device="cuda")
                    st.synchronize()
                    nat.check(L.gprc_dev_factor_all(ctx.handle, a.data_ptr(), g.n_pad, w.data_ptr(), info.data_ptr(), NULL))
                    st.synchronize()
                    if int(info[0]) != 0:
                        errs.append(f"thread {i}: info {int(info[0])} at repetition {r}")
                        break
                    if ref is None:
                        ref = (a.clone(), w.clone())
                    elif not (torch.equal(a, ref[0]) and torch.equal(w, ref[1])):
                        errs.append(f"thread {i}: repetition {r} differs from the first")
                        break
            ctx.close()
            done.append(i)
        except Exception as e:   # noqa: BLE001
            errs.append(repr(e))

    ts = [threading.Thread(target=work, args=(i, n)) for i, n in enumerate((2600, 4100))]
    [t.start() for t in ts]
    [t.join(timeout=300) for t in ts]
    assert not errs and sorted(done) == [0, 1], errs


def test_one_launch_per_panel_under_the_service_is_still_bit_identical(tmp_path):
    """GPRC_SWEEP=0 keeps round 2's form of the caller's-stream work (one trailing_service_kernel per panel instead of the persistent
    sweep kernel).  The switch is read once per process, so a child process factors with it and compares every word with the
    factor_panel / update_trailing sweep, as test_factor_service_is_bit_identical_to_the_launch_per_panel_sweep does for the default."""
    import os
    import subprocess
    import sys
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import torch\n"
        "import tests.test_gpu_device_level as t\n"
        "from gprc_amd import _native as nat\n"
        "for n in (1536, 9100):\n"
        "    L, ctx, g, K = t._filled(n, seed=5)\n"
        "    a = t._clone(K); w, info = t._new(g)\n"
        "    for p in range(g.P):\n"
        "        nat.check(L.gprc_dev_factor_panel(ctx.handle, a.data_ptr(), g.n_pad, p, w.data_ptr(), info.data_ptr()))\n"
        "        if p + 1 < g.P:\n"
        "            nat.check(L.gprc_dev_update_trailing(ctx.handle, a.data_ptr(), g.n_pad, p, p + 1, g.P, 1))\n"
        "    torch.cuda.synchronize()\n"
        "    b = t._clone(K); w2, info2 = t._new(g)\n"
        "    nat.check(L.gprc_dev_factor_all(ctx.handle, b.data_ptr(), g.n_pad, w2.data_ptr(), info2.data_ptr(), None))\n"
        "    torch.cuda.synchronize()\n"
        "    assert int(info[0]) == 0 and int(info2[0]) == 0\n"
        "    assert torch.equal(b, a) and torch.equal(w2, w), n\n"
        "    ctx.close()\n"
        "print('SWEEP0_OK')\n"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GPRC_SWEEP="0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "SWEEP0_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("env,sizes", [({"GPRC_CHAIN_SPLIT": "0"}, (1024, 2900, 9100)), ({"GPRC_CHAIN_SPLIT": "1", "GPRC_FACTOR": "40"}, (1024, 2900, 9100)),
                                       ({"GPRC_CHAIN_SPLIT": "1", "GPRC_SWEEP": "0"}, (1024, 2900, 9100)),
                                       ({"GPRC_SERVICE_SHARE": "1"}, (11000,)), ({}, (13500,)), ({"GPRC_SERVICE_SHARE": "1", "GPRC_CHAIN_SPLIT": "0", "GPRC_FACTOR": "300"}, (11000,))],
                         ids=["unsplit", "split-in-groups", "split-launch-per-panel", "shared-service", "shared-service-default", "shared-unsplit-in-groups"])
def test_both_forms_of_the_panel_chain_are_bit_identical(env, sizes):
    """The split chain (four helper workgroups share the chain's solve / update tiles in 32-row slices, apply every update of a panel's
    diagonal blocks and run the step from W_3 to the next panel's first potf2; the look-ahead strips' early k-chunks ride on the
    next-diagonal-block roles) is the default below n_pad = 20480.  GPRC_CHAIN_SPLIT is read once per process, so child processes
    force the other form, the split form inside the groups of the left-looking schedule (GPRC_FACTOR=40: groups of a few panels, i.e.
    group boundaries where the hand-over to the helpers does not happen) and the split form beside one update launch per panel -- each
    compared word for word, with the explicit inverses, against the factor_panel / update_trailing sweep.  Likewise the SHARED service
    (the 4-wave roles in a launch of their own with a GEMM team's LDS, one sweep workgroup beside each: default from n_pad = 13312, forced
    from 10752), alone, at its default size, and unsplit inside groups."""
    import os
    import subprocess
    import sys
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import torch\n"
        "import tests.test_gpu_device_level as t\n"
        "from gprc_amd import _native as nat\n"
        "for n in %r:\n"
        "    L, ctx, g, K = t._filled(n, seed=7)\n"
        "    a = t._clone(K); w, info = t._new(g)\n"
        "    for p in range(g.P):\n"
        "        nat.check(L.gprc_dev_factor_panel(ctx.handle, a.data_ptr(), g.n_pad, p, w.data_ptr(), info.data_ptr()))\n"
        "        if p + 1 < g.P:\n"
        "            nat.check(L.gprc_dev_update_trailing(ctx.handle, a.data_ptr(), g.n_pad, p, p + 1, g.P, 1))\n"
        "    inv = t._new_inv(g)\n"
        "    nat.check(L.gprc_dev_solve_prepare(ctx.handle, a.data_ptr(), w.data_ptr(), g.n_pad, inv.data_ptr(), 0, g.P))\n"
        "    torch.cuda.synchronize()\n"
        "    for rep in range(3):\n"
        "        b = t._clone(K); w2, info2 = t._new(g); inv2 = t._new_inv(g)\n"
        "        nat.check(L.gprc_dev_factor_all(ctx.handle, b.data_ptr(), g.n_pad, w2.data_ptr(), info2.data_ptr(), inv2.data_ptr()))\n"
        "        torch.cuda.synchronize()\n"
        "        assert int(info[0]) == 0 and int(info2[0]) == 0, (n, int(info2[0]))\n"
        "        assert torch.equal(b, a) and torch.equal(w2, w), n\n"
        "        assert torch.equal(torch.nan_to_num(inv2), torch.nan_to_num(inv)), n   # (entries below the block diagonal are never written)\n"
        "    ctx.close()\n"
        "print('CHAIN_OK')\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), tuple(sizes))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "CHAIN_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
