"""Oracle-backed stand-in for gprc_amd.distributed.HipOps -- TEST INFRASTRUCTURE.

Same method surface as HipOps, but every stage is computed on the CPU with the oracle / numpy on torch CPU
tensors, so the multi-rank driver (panel ownership, broadcast schedule, look-ahead order, test-point
slicing) can be exercised with the gloo backend on a box without GPUs.  It lives under tests/ so that the
product package never reaches the oracle."""
from contextlib import contextmanager

import numpy as np
import scipy.linalg as sl
import torch

from gprc_amd.distributed import Geometry
from oracle import oracle as orc


class OracleOps:
    torch = torch

    def __init__(self, kernel_id, params, d, n, noise):
        self.kernel_id, self.params = int(kernel_id), list(np.atleast_1d(params).astype(float))
        self.d, self.n, self.noise = int(d), int(n), float(noise)
        self.geom = Geometry(n)
        self.log = []   # (stage, args) trace for schedule assertions
        self._dense = None   # (signature of the packed buffer, its dense factor): see dense_L

    # memory / streams (no-ops on the CPU)
    def zeros(self, count, dtype=None):
        return torch.zeros(int(count), dtype=dtype or torch.float64)

    def from_host(self, a):
        return torch.from_numpy(np.ascontiguousarray(a))

    def to_host(self, t):
        return t.numpy()

    @contextmanager
    def on(self, side):
        yield

    @contextmanager
    def on_comm(self):
        yield

    def comm_after_side(self):
        self.log.append(("comm_after_side",))

    def fork_side(self):
        self.log.append(("fork",))

    def join_side(self):
        self.log.append(("join",))

    def synchronize(self):
        pass

    def close(self):
        pass

    # helpers
    def _panel(self, packed, p):
        g = self.geom
        ld = g.n_pad - p * g.NB
        return packed.numpy()[g.panel_slice(p)].reshape(g.NB, ld).T  # (ld, NB) matrix-oriented view

    def _points(self, X):
        return X.numpy().reshape(self.n, self.d).T  # d x n

    def dense_L(self, packed):
        # built once per finished factor (the solves, logp, the predict and the test's own read all ask for it: five n_pad^2 arrays per
        # rank, each a fresh 100-MB mapping -- most of the 8-rank cases' five minutes was the kernel zeroing pages); a sampled checksum of
        # the packed buffer tells a changed factor from the one already expanded
        g = self.geom
        flat = packed.numpy()
        sig = (id(packed), float(flat[::61].sum()), float(np.abs(flat[::127]).sum()))   # the exchange writes received panels without telling us
        if self._dense is None or self._dense[0] != sig:
            L = np.zeros((g.n_pad, g.n_pad))
            for p in range(g.P):
                c0 = p * g.NB
                L[c0:, c0:c0 + g.NB] = self._panel(packed, p)
                L[c0:c0 + g.NB, c0:c0 + g.NB] = np.tril(L[c0:c0 + g.NB, c0:c0 + g.NB])
            self._dense = (sig, L)
        return self._dense[1]

    # stages
    def fill_panel(self, X, packed, p):
        g = self.geom
        Xp = self._points(X)
        c0, c1 = p * g.NB, (p + 1) * g.NB
        blk = np.zeros((g.n_pad - c0, g.NB))
        rows = np.arange(c0, g.n_pad)
        cols = np.arange(c0, c1)
        vr, vc = rows < self.n, cols < self.n
        if vr.any() and vc.any():
            blk[np.ix_(vr, vc)] = orc.kernel_matrix(self.kernel_id, self.params, Xp[:, rows[vr]], Xp[:, cols[vc]])
        for j, c in enumerate(cols):
            blk[c - c0, j] = blk[c - c0, j] + self.noise if c < self.n else 1.0
        self._panel(packed, p)[...] = blk
        self.log.append(("fill", p))

    def factor_panel(self, packed, p, winv, info, side):
        g = self.geom
        pan = self._panel(packed, p)
        D, inf = orc.potrf_lower(pan[:g.NB, :g.NB])
        if inf and int(info[0]) == 0:
            info[0] = p * g.NB + inf
        pan[:g.NB, :g.NB] = D
        if pan.shape[0] > g.NB and not inf:
            pan[g.NB:, :] = sl.solve_triangular(np.tril(D), pan[g.NB:, :].T, lower=True).T
        w = winv.numpy()[g.winv_slice(p)].reshape(g.NB // 128, 128, 128)
        for j in range(g.NB // 128):
            blk = np.tril(D[j * 128:(j + 1) * 128, j * 128:(j + 1) * 128])
            w[j] = (np.linalg.inv(blk) if not inf else np.eye(128)).T  # stored column-major
        self.log.append(("factor", p, bool(side)))

    def factor_subpanel(self, packed, p, j, part, winv, info, side):
        """Sub-step j of panel p: part 1 = factor the 128 x 128 diagonal block, solve the rows below it (those columns
        are then final); part 2 = update the remaining columns of the panel with them."""
        g = self.geom
        pan = self._panel(packed, p)
        c0, c1 = 128 * j, 128 * (j + 1)
        if part in (0, 1):
            D, inf = orc.potrf_lower(pan[c0:c1, c0:c1])
            if inf and int(info[0]) == 0:
                info[0] = p * g.NB + c0 + inf
            pan[c0:c1, c0:c1] = D
            if not inf:
                pan[c1:, c0:c1] = sl.solve_triangular(np.tril(D), pan[c1:, c0:c1].T, lower=True).T
            w = winv.numpy()[g.winv_slice(p)].reshape(g.NB // 128, 128, 128)
            w[j] = (np.linalg.inv(np.tril(D)) if not inf else np.eye(128)).T  # stored column-major
            if j == g.NB // 128 - 1:
                self.log.append(("factor", p, bool(side)))
        if part in (0, 2) and c1 < g.NB:
            Lcol = pan[c1:, c0:c1]
            pan[c1:, c1:g.NB] -= Lcol @ Lcol[: g.NB - c1, :].T

    def update_trailing(self, packed, p, q0, q1, stride, side):
        g = self.geom
        Lp = self._panel(packed, p)
        for q in range(q0, min(q1, g.P), stride):
            r0 = (q - p) * g.NB
            self._panel(packed, q)[...] -= Lp[r0:, :] @ Lp[r0:r0 + g.NB, :].T
            self.log.append(("update", p, q, bool(side)))

    def update_range(self, packed, p0, p1, q0, q1, stride, side):
        for p in range(p0, p1):                       # the same products in the same order, panel by panel
            self.update_trailing(packed, p, q0, q1, stride, side)

    def aux_after_panel(self):
        self.log.append(("aux_after_panel",))

    def aux_after_main(self):
        pass

    def join_aux(self):
        self.log.append(("join_aux",))

    def trsv_step_aux(self, packed, winv, b, p, work=None):
        """Forward-substitution step of panel p: z_p = L_pp^-1 z_p, then z_below -= L[below, p] z_p."""
        g = self.geom
        pan = self._panel(packed, p)
        z = b.numpy()
        c0 = p * g.NB
        z[c0:c0 + g.NB] = sl.solve_triangular(np.tril(pan[:g.NB, :g.NB]), z[c0:c0 + g.NB], lower=True)
        z[c0 + g.NB:] -= pan[g.NB:, :] @ z[c0:c0 + g.NB]
        self.log.append(("fwd", p))

    def trsv(self, packed, winv, b, transpose, work):
        L = self.dense_L(packed)
        b.numpy()[...] = sl.solve_triangular(L.T if transpose else L, b.numpy(), lower=not transpose)

    def logp(self, packed, y, alpha, out):
        L = self.dense_L(packed)
        n = self.n
        out[0] = -0.5 * float(y.numpy()[:n] @ alpha.numpy()[:n]) - np.log(np.diag(L)[:n]).sum() - n / 2 * np.log(2 * np.pi)

    def read_info(self, info):
        return int(info[0])

    def predict(self, X, y, packed, winv, alpha, Xs, ns, mean, var):
        if ns == 0:
            return
        n = self.n
        L = self.dense_L(packed)[:n, :n]
        m, v = orc.gpr_predict(self.kernel_id, self.params, self._points(X), L, alpha.numpy()[:n], Xs.numpy().reshape(-1, self.d)[:ns].T)
        mean.numpy()[:ns] = m
        var.numpy()[:ns] = v
