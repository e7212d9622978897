"""One-process-per-GPU driver of the GP predict step (fit + predict) over torch.distributed.

Sharding (DESIGN.md "Multi-GPU"):
  * the factor lives in the packed block-column layout; panel p (512 columns, contiguous in memory) is
    OWNED by rank p mod G (1-D block-cyclic).  Every rank allocates the whole packed buffer: it fills
    and updates only its own panels and RECEIVES the others, so after the factorisation L is
    replicated (17 GB at n = 65536, of 288 GB) and the predict needs no communication;
  * the one exchange step of the path: the owner broadcasts the factored panel (+ the inverses of its
    four diagonal blocks) -- RCCL broadcast over xGMI on the GPU box, gloo in the CPU tests;
  * look-ahead: the owner of panel p+1 updates and factors that panel on a side stream while the
    rest of trailing update p is still running on the main stream, and every rank posts the receive
    of panel p+1 on its side stream at the same point, so the broadcast overlaps the update;
  * alpha is solved redundantly on every rank; the n* test points are split in contiguous slices.

The arithmetic is behind an `ops` object.  `HipOps` (below) is the product: native HIP kernels through
the C ABI on torch-owned device memory.  The CPU tests inject an oracle-backed ops object of the same
shape from tests/ -- nothing in this package falls back to the CPU.
"""
from __future__ import annotations

import ctypes as C
import os
from contextlib import contextmanager

import numpy as np

from . import _native as nat


class Geometry:
    """Packed block-column geometry of an n x n factor (pure index arithmetic from the C ABI)."""

    def __init__(self, n: int):
        L = nat.lib()
        self.n = int(n)
        self.NB = int(L.gprc_panel_width())
        self.n_pad = int(L.gprc_pad(n))
        self.P = int(L.gprc_panel_count(self.n_pad))
        self.offsets = [int(L.gprc_panel_offset(self.n_pad, p)) for p in range(self.P + 1)]
        self.packed_size = int(L.gprc_packed_size(self.n_pad))
        self.winv_size = int(L.gprc_winv_size(self.n_pad))
        self.winv_per_panel = self.winv_size // self.P
        self.trsv_work = int(L.gprc_trsv_work_size(self.n_pad))

    def panel_slice(self, p):
        return slice(self.offsets[p], self.offsets[p + 1])

    def winv_slice(self, p):
        return slice(p * self.winv_per_panel, (p + 1) * self.winv_per_panel)

    def panel_part_slice(self, p, j):
        """Columns [128 j, 128 (j+1)) of panel p: contiguous in the packed buffer (column-major, ld = n_pad - p NB)."""
        ld = self.n_pad - p * self.NB
        return slice(self.offsets[p] + j * 128 * ld, self.offsets[p] + (j + 1) * 128 * ld)


class SingleComm:
    """world_size 1: no exchange."""
    rank, world = 0, 1

    def broadcast(self, t, src):
        pass

    def min_positive(self, value: int) -> int:
        return value

    def gather_concat(self, t, sizes):
        return t

    def barrier(self):
        pass


class TorchComm:
    """torch.distributed (backend nccl == RCCL on ROCm; gloo on CPU).

    Panel broadcast strategy.  xGMI is point-to-point: a rooted broadcast that pipelines through a ring moves the
    panel at ONE link's rate, while the root has 7 links.  `scatter_allgather` has the root scatter 1/G of the panel
    to every peer (G-1 links in parallel) and then all-gathers the pieces (every link carries 1/G), the classic
    large-message broadcast.  Which one wins depends on what RCCL's own broadcast does on the node at hand, so
    `calibrate()` times both on a panel-sized buffer once and all ranks adopt the faster (GPRC_BCAST=broadcast |
    scatter_allgather pins it).  Results are identical either way (pure data movement)."""

    def __init__(self, strategy=None):
        import torch.distributed as dist
        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.strategy = strategy or os.environ.get("GPRC_BCAST", "auto")
        if self.strategy not in ("auto", "broadcast", "scatter_allgather"):
            raise ValueError("GPRC_BCAST must be auto, broadcast or scatter_allgather")
        self.choice = "broadcast" if self.strategy == "auto" else self.strategy   # until calibrate() decides
        self.calibration = None

    MIN_SPLIT = 1 << 17   # doubles (1 MiB): below this two collectives cost more latency than they save

    def broadcast(self, t, src):
        pinned = self.strategy == "scatter_allgather"   # pinned: also with one rank, to exercise the two collectives
        if (self.choice == "scatter_allgather" and (self.world > 1 or pinned) and t.numel() % self.world == 0
                and t.numel() >= self.MIN_SPLIT and t.is_contiguous()):
            self._scatter_allgather(t, src)
        else:
            self.dist.broadcast(t, src=src)

    def _scatter_allgather(self, t, src):
        c = t.numel() // self.world
        mine = t[self.rank * c:(self.rank + 1) * c]
        if self.rank == src:
            self.dist.scatter(mine, scatter_list=[t[r * c:(r + 1) * c] for r in range(self.world)], src=src)
        else:
            self.dist.scatter(mine, src=src)
        self.dist.all_gather_into_tensor(t, mine)   # in place: `mine` is this rank's slice of t

    def calibrate(self, make_buffer, sync, count=1 << 24, reps=3):
        """Times both strategies on a `count`-double buffer (make_buffer(count) -> tensor; sync() waits for the
        device), checks that they deliver the same data, and fixes `self.choice` identically on every rank."""
        import time
        import torch
        if self.world == 1 or self.strategy != "auto":
            return self.choice
        count -= count % self.world
        buf = make_buffer(count)
        ref = make_buffer(count)
        if self.dist.get_backend() != "nccl" and buf.is_cuda:
            # gloo moving DEVICE tensors (ranks sharing one GPU in the rehearsal runs) has no scatter /
            # all_gather_into_tensor: nothing to choose between.  Decided from the backend name and the buffer's device,
            # identically on every rank, before any collective.
            self.calibration = {"choice": self.choice, "skipped": "backend " + self.dist.get_backend() + " on device tensors"}
            return self.choice
        times, ok = {}, True
        # No try/except around the collectives: an error raised by ONE rank inside a collective must end that rank
        # (the launcher then tears the job down) -- swallowing it would leave the other ranks blocked in the collective.
        for mode in ("broadcast", "scatter_allgather"):
            self.choice = mode
            for it in range(reps + 1):
                if it == 1:
                    sync(); self.dist.barrier(); t0 = time.perf_counter()
                src = it % self.world
                if self.rank == src:
                    buf.copy_(torch.arange(count, dtype=buf.dtype, device=buf.device) * (it + 1))
                else:
                    buf.zero_()
                self.broadcast(buf, src)
            sync()
            times[mode] = (time.perf_counter() - t0) / reps
            if mode == "broadcast":
                ref.copy_(buf)
            else:
                ok = bool(torch.equal(ref, buf))
        stat = torch.tensor([times["broadcast"], times["scatter_allgather"], 0.0 if ok else 1.0], dtype=torch.float64, device=buf.device)
        self.dist.all_reduce(stat, op=self.dist.ReduceOp.MAX)      # slowest rank decides; any mismatch vetoes
        tb, ts, bad = (float(x) for x in stat.cpu())
        self.choice = "scatter_allgather" if (bad == 0.0 and ts < 0.9 * tb) else "broadcast"
        self.calibration = {"broadcast_ms": round(tb * 1e3, 3), "scatter_allgather_ms": round(ts * 1e3, 3),
                            "doubles": count, "agree": bad == 0.0, "choice": self.choice}
        return self.choice

    def min_positive(self, value: int) -> int:
        import torch
        big = 2 ** 62
        t = torch.tensor([value if value > 0 else big], dtype=torch.int64)
        if self.dist.get_backend() == "nccl":
            t = t.cuda()
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        v = int(t.item())
        return 0 if v == big else v

    def gather_concat(self, t, sizes):
        """All ranks contribute a 1-D tensor (rank r: sizes[r] valid entries); returns the concatenation."""
        import torch
        m = max(sizes)
        buf = torch.zeros(m, dtype=t.dtype, device=t.device)
        buf[: t.numel()] = t
        outs = [torch.empty_like(buf) for _ in range(self.world)]
        self.dist.all_gather(outs, buf)
        return torch.cat([o[:s] for o, s in zip(outs, sizes)])

    def barrier(self):
        self.dist.barrier()


class HipOps:
    """Native stage launcher on one GPU: torch owns memory and streams, the C ABI does the arithmetic."""

    def __init__(self, device: int, kernel_id: int, params, d: int, n: int, noise: float):
        import torch
        self.torch = torch
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        self.main_stream = torch.cuda.Stream(device=self.device)
        # high priority: the look-ahead panel work (a few small kernels) must be dispatched AHEAD of the tens of
        # thousands of queued workgroups of the trailing update that runs beside it
        prio = int(os.environ.get("GPRC_SIDE_PRIORITY", "-1"))
        self.side_stream = torch.cuda.Stream(device=self.device, priority=prio)
        self.comm_stream = torch.cuda.Stream(device=self.device, priority=prio)   # where the panel broadcasts are posted
        self.aux_stream = torch.cuda.Stream(device=self.device)                   # the forward solve, run beside the sweep
        self.ctx_aux = None
        self.ctx_main = nat.Context(device, self.main_stream.cuda_stream)
        self.ctx_side = nat.Context(device, self.side_stream.cuda_stream)
        self.ctx_aux = nat.Context(device, self.aux_stream.cuda_stream)
        self.kernel_id = int(kernel_id)
        self.params, self._pp, self._np = nat.params_array(params)
        self.d, self.n, self.noise = int(d), int(n), float(noise)
        self.geom = Geometry(n)
        self.L = nat.lib()
        # the explicit inverses of the panels' diagonal blocks the vector solves work with (gprc_dev_solve_prepare): computed
        # where the solve runs -- every rank for every panel, on the auxiliary stream beside the sweep -- never exchanged
        self.inv = None
        self._prepared = set()

    def begin_fit(self):
        self._prepared = set()
        self._solve_inv()

    def _solve_inv(self):
        # uninitialised on purpose: a fill kernel queued on the main stream could run AFTER the auxiliary stream has written the
        # first inverses; the solves read only what gprc_dev_solve_prepare has written
        if self.inv is None:
            with self.torch.cuda.stream(self.main_stream):
                self.inv = self.torch.empty(int(self.L.gprc_solve_inv_size(self.geom.n_pad)), dtype=self.torch.float64, device=self.device)
        return self.inv

    def _prepare(self, ctx, packed, winv, panels):
        """gprc_dev_solve_prepare for the listed panels that have not been prepared since begin_fit (runs of consecutive panels
        in one launch)."""
        todo = sorted(p for p in panels if p not in self._prepared)
        inv = self._solve_inv()
        i = 0
        while i < len(todo):
            j = i
            while j + 1 < len(todo) and todo[j + 1] == todo[j] + 1:
                j += 1
            nat.check(self.L.gprc_dev_solve_prepare(ctx, packed.data_ptr(), winv.data_ptr(), self.geom.n_pad, inv.data_ptr(), todo[i], todo[j] + 1))
            i = j + 1
        self._prepared.update(todo)

    # ---- memory ----
    def zeros(self, count, dtype=None):
        with self.torch.cuda.stream(self.main_stream):
            return self.torch.zeros(int(count), dtype=dtype or self.torch.float64, device=self.device)

    def from_host(self, a):
        with self.torch.cuda.stream(self.main_stream):
            return self.torch.from_numpy(np.ascontiguousarray(a)).to(self.device)

    def to_host(self, t):
        self.synchronize()
        return t.cpu().numpy()

    # ---- streams ----
    @contextmanager
    def on(self, side: bool):
        with self.torch.cuda.stream(self.side_stream if side else self.main_stream):
            yield

    @contextmanager
    def on_comm(self):
        with self.torch.cuda.stream(self.comm_stream):
            yield

    def fork_side(self):
        self.side_stream.wait_stream(self.main_stream)

    def comm_after_side(self):
        """The communication stream waits for everything queued on the side stream so far."""
        self.comm_stream.wait_stream(self.side_stream)

    def join_side(self):
        self.main_stream.wait_stream(self.side_stream)
        self.main_stream.wait_stream(self.comm_stream)

    def aux_after_panel(self):
        """The auxiliary stream waits for the panel just completed (factored on the side stream / received on comm)."""
        self.aux_stream.wait_stream(self.side_stream)
        self.aux_stream.wait_stream(self.comm_stream)

    def aux_after_main(self):
        self.aux_stream.wait_stream(self.main_stream)

    def join_aux(self):
        self.main_stream.wait_stream(self.aux_stream)

    def synchronize(self):
        self.main_stream.synchronize()
        self.side_stream.synchronize()
        self.comm_stream.synchronize()
        self.aux_stream.synchronize()

    def _ctx(self, side):
        return (self.ctx_side if side else self.ctx_main).handle

    # ---- stages ----
    def fill_panel(self, X, packed, p):
        g = self.geom
        nat.check(self.L.gprc_dev_fill_panel(self._ctx(False), self.kernel_id, self._pp, self._np, X.data_ptr(), self.d, self.n, g.n_pad,
                                             self.noise, packed.data_ptr(), p))

    def factor_panel(self, packed, p, winv, info, side):
        nat.check(self.L.gprc_dev_factor_panel(self._ctx(side), packed.data_ptr(), self.geom.n_pad, p, winv.data_ptr(), info.data_ptr()))

    def factor_subpanel(self, packed, p, j, part, winv, info, side):
        nat.check(self.L.gprc_dev_factor_subpanel(self._ctx(side), packed.data_ptr(), self.geom.n_pad, p, j, part, winv.data_ptr(),
                                                  info.data_ptr()))

    def factor_all(self, packed, winv, info):
        """Every panel on this one GPU (no exchange): the native grouped left-looking sweep, bit-identical to the
        factor_panel / update_trailing loop."""
        nat.check(self.L.gprc_dev_factor_all(self._ctx(False), packed.data_ptr(), self.geom.n_pad, winv.data_ptr(), info.data_ptr(),
                                             self._solve_inv().data_ptr()))
        self._prepared = set(range(self.geom.P))

    def update_trailing(self, packed, p, q0, q1, stride, side):
        if q0 < q1:
            nat.check(self.L.gprc_dev_update_trailing(self._ctx(side), packed.data_ptr(), self.geom.n_pad, p, q0, q1, stride))

    def update_range(self, packed, p0, p1, q0, q1, stride, side):
        """Source panels [p0, p1) applied to the targets q0, q0 + stride, ... < q1 in one pass (bit-identical to p1 - p0
        single-panel updates in order)."""
        if q0 < q1 and p0 < p1:
            nat.check(self.L.gprc_dev_update_range(self._ctx(side), packed.data_ptr(), self.geom.n_pad, p0, p1, q0, q1, stride))

    def trsv(self, packed, winv, b, transpose, work):
        self._prepare(self._ctx(False), packed, winv, range(self.geom.P))
        nat.check(self.L.gprc_dev_trsv(self._ctx(False), packed.data_ptr(), self.inv.data_ptr(), self.geom.n_pad, b.data_ptr(), int(transpose),
                                       work.data_ptr()))

    def trsv_step_aux(self, packed, winv, b, p, work):
        """Forward-solve step of panel p on the auxiliary stream (the panel's explicit diagonal inverse first)."""
        with self.torch.cuda.stream(self.aux_stream):
            self._prepare(self.ctx_aux.handle, packed, winv, [p])
            nat.check(self.L.gprc_dev_trsv_step(self.ctx_aux.handle, packed.data_ptr(), self.inv.data_ptr(), self.geom.n_pad, b.data_ptr(), 0, p,
                                                work.data_ptr()))

    def logp(self, packed, y, alpha, out):
        nat.check(self.L.gprc_dev_logp(self._ctx(False), packed.data_ptr(), self.geom.n_pad, self.n, y.data_ptr(), alpha.data_ptr(),
                                       out.data_ptr()))

    def read_info(self, info) -> int:
        self.synchronize()
        return int(info[0].item())

    def predict(self, X, y, packed, winv, alpha, Xs, ns, mean, var):
        """Pointwise predict of `ns` test points (device buffers) against the replicated factor."""
        if ns == 0:
            return
        model = C.c_void_p()
        nat.check(self.L.gprc_gpr_model_from_device(self._ctx(False), self.kernel_id, self._pp, self._np, X.data_ptr(), self.d, self.n,
                                                    y.data_ptr(), packed.data_ptr(), winv.data_ptr(), alpha.data_ptr(), self.noise, 0.0,
                                                    C.byref(model)))
        try:
            nat.check(self.L.gprc_gpr_predict(model, Xs.data_ptr(), ns, 1, mean.data_ptr(), var.data_ptr()))
        finally:
            self.L.gprc_model_free(model)

    closed = False

    def close(self):
        if self.closed:
            return
        self.synchronize()
        self.ctx_main.close()
        self.ctx_side.close()
        self.ctx_aux.close()
        self.closed = True


def owned_after(p: int, rank: int, world: int) -> int:
    """Smallest panel index q > p with q mod world == rank."""
    q = p + 1
    return q + (rank - q) % world


class DistributedGPR:
    """The predict step of BASELINE.json on G ranks: fit (F1-F3) + pointwise predict (P1-P3)."""

    def __init__(self, ops, comm, lookahead=None):
        self.ops, self.comm = ops, comm
        # Look-ahead pays when there is a broadcast to hide (world > 1).  On ONE GPU the side-stream panel chain
        # only competes with the trailing update for CUs (measured: potf2 0.17 -> 0.42 ms under contention, step
        # time +0..25 %), so it is off by default there.  GPRC_LOOKAHEAD=0/1 overrides.
        env = os.environ.get("GPRC_LOOKAHEAD")
        if lookahead is None:
            lookahead = (comm.world > 1) if env is None else (env == "1")
        self.lookahead = bool(lookahead)
        # pipelined panel exchange: the owner broadcasts each 128-column quarter of a panel as soon as it is final,
        # while it is still factoring the rest (GPRC_PIPE_BCAST=0: factor the whole panel, then one broadcast)
        self.pipeline = os.environ.get("GPRC_PIPE_BCAST", "1") != "0" and hasattr(ops, "factor_subpanel")
        # batched trailing update: a rank applies the panels it has received to its FAR panels (everything but the one
        # it has to factor next) only every `batch` steps, all of them in one pass with the C tiles held in the
        # accumulators -- the left-looking saving (one tile prologue and one C load/store per batch instead of per
        # panel); the panel a rank factors next is always brought up to date at once.  GPRC_UPDATE_BATCH=1: per panel.
        self.batch = max(1, int(os.environ.get("GPRC_UPDATE_BATCH", "4"))) if hasattr(ops, "update_range") else 1
        g = ops.geom
        self.geom = g
        self.packed = ops.zeros(g.packed_size)
        self.winv = ops.zeros(g.winv_size)
        self.alpha = ops.zeros(g.n_pad)
        self.work = ops.zeros(g.trsv_work)
        self.info = ops.zeros(4, dtype=_int32(ops))
        self.scal = ops.zeros(8)
        self.logp = None
        self.info_value = 0

    # -- F1 + F2 + F3 ------------------------------------------------------------------------------
    def fit(self, X, y_pad):
        """X: d x n device buffer (point-major); y_pad: n_pad doubles, zero padded.  Returns LAPACK info."""
        ops, comm, g = self.ops, self.comm, self.geom
        rank, G, P = comm.rank, comm.world, g.P
        if hasattr(ops, "begin_fit"):
            ops.begin_fit()
        with ops.on(False):
            self.info.zero_() if hasattr(self.info, "zero_") else self.info.fill(0)
            for p in range(rank, P, G):                       # F1: own panels only, no communication
                ops.fill_panel(X, self.packed, p)
        if G == 1 and not self.lookahead and isinstance(comm, SingleComm) and hasattr(ops, "factor_all"):
            with ops.on(False):                               # one rank, nothing to exchange: the whole sweep natively
                ops.factor_all(self.packed, self.winv, self.info)
            if hasattr(ops, "L") and ops.read_info(self.info) == nat.INFO_WAIT_TIMEOUT and ops.L.gprc_factor_service(-1) == 1:
                # the factor service's persistent launch and the sweep's kernels did not run concurrently (a tool that serialises
                # dispatches, e.g. rocprofv3 --pmc): once per process, switch it off, rebuild the matrix and factor again
                ops.L.gprc_factor_service(0)
                with ops.on(False):
                    self.info.zero_()
                    for p in range(P):
                        ops.fill_panel(X, self.packed, p)
                    ops.factor_all(self.packed, self.winv, self.info)
            return self._finish_fit(y_pad)
        # F3 starts inside F2: the forward solve L z = y needs only panels <= p at step p, so it runs on an auxiliary
        # stream beside the trailing updates (128 latency-bound steps that would otherwise follow the sweep)
        self._fwd_in_sweep = hasattr(ops, "trsv_step_aux")
        if self._fwd_in_sweep:
            with ops.on(False):
                _copy(self.alpha, y_pad)
            ops.aux_after_main()
        ops.fork_side()
        self._factor_and_share(0)
        far_from = 0                                          # panels [0, far_from) are applied to all my unfactored panels
        for p in range(P):                                    # F2: right-looking, one panel per step
            if self._fwd_in_sweep:
                ops.aux_after_panel()                         # panel p and its inverses are complete on side / comm
                ops.trsv_step_aux(self.packed, self.winv, self.alpha, p, self.work)
            ops.join_side()                                   # panel p factored (owner) / received (others)
            if p + 1 < P:
                nxt = (p + 1) % G
                mine = rank == nxt                            # I factor panel p + 1 next
                flush = (p + 1 - far_from >= self.batch) or p + 2 >= P
                if mine and not self.lookahead:
                    with ops.on(False):                       # no look-ahead: bring all my panels up to date, then factor
                        ops.update_range(self.packed, far_from, p + 1, p + 1, P, G, False)
                    far_from = p + 1
                    ops.fork_side()
                    self._factor_and_share(p + 1)
                    continue
                if mine:
                    ops.fork_side()                           # look-ahead: panel p+1 first, on the side stream
                    with ops.on(True):
                        ops.update_range(self.packed, far_from, p + 1, p + 1, p + 2, 1, True)
                    q0 = p + 1 + G
                else:
                    q0 = owned_after(p, rank, G)
                self._factor_and_share(p + 1)                 # owner: side stream; everybody: broadcasts on the comm stream
                if flush:
                    with ops.on(False):                       # runs beside the factorisation and the exchange
                        ops.update_range(self.packed, far_from, p + 1, q0, P, G, False)
                    far_from = p + 1
        ops.join_side()
        if self._fwd_in_sweep:
            ops.join_aux()
        return self._finish_fit(y_pad, forward_done=self._fwd_in_sweep)

    def _factor_and_share(self, p):
        """Panel p has received every update on its owner: factor it there (side stream) and replicate it (comm stream).
        Pipelined form: sub-step j makes the columns [128 j, 128 (j+1)) final; their broadcast is posted at once and
        runs while the owner updates and factors the rest of the panel."""
        ops, comm, g = self.ops, self.comm, self.geom
        src = p % comm.world
        exchange = comm.world > 1 or isinstance(comm, TorchComm)
        if not (self.pipeline and exchange):
            if comm.rank == src:
                with ops.on(True):
                    ops.factor_panel(self.packed, p, self.winv, self.info, True)
            self._bcast_panel(p)
            return
        for j in range(g.NB // 128):
            if comm.rank == src:
                with ops.on(True):
                    ops.factor_subpanel(self.packed, p, j, 1, self.winv, self.info, True)   # factor + solve: columns final
                ops.comm_after_side()
                with ops.on(True):
                    ops.factor_subpanel(self.packed, p, j, 2, self.winv, self.info, True)   # right-looking ops: update the rest of
                    # the panel (the native ops work left-looking inside the panel: part 1 did everything, part 2 is empty)
            with ops.on_comm():
                comm.broadcast(self.packed[g.panel_part_slice(p, j)], src)
        with ops.on_comm():
            comm.broadcast(self.winv[g.winv_slice(p)], src)   # the four inverses were complete before the last quarter left

    def _finish_fit(self, y_pad, forward_done=False):
        ops, comm = self.ops, self.comm
        local_info = ops.read_info(self.info)
        if local_info < 0:   # a fused kernel's device-side dependency wait ran out (never seen; see gprc_internal.h): not a factor
            raise nat.GprcError(nat.ERR_HIP, f"device-side dependency wait timed out on rank {comm.rank} (info = {local_info}); under a tool that "
                                "serialises kernel dispatches (e.g. rocprofv3 --pmc) set GPRC_SERVICE=0")
        self.info_value = comm.min_positive(local_info)
        if self.info_value != 0:
            return self.info_value
        with ops.on(False):                                   # F3: replicated, L is complete everywhere
            if not forward_done:
                _copy(self.alpha, y_pad)
                ops.trsv(self.packed, self.winv, self.alpha, False, self.work)
            ops.trsv(self.packed, self.winv, self.alpha, True, self.work)
            ops.logp(self.packed, y_pad, self.alpha, self.scal)
        return 0

    def _bcast_panel(self, p):
        if self.comm.world == 1 and not isinstance(self.comm, TorchComm):
            return
        g = self.geom
        src = p % self.comm.world
        if self.comm.rank == src:
            self.ops.comm_after_side()                        # the panel is complete on the side stream
        with self.ops.on_comm():
            self.comm.broadcast(self.packed[g.panel_slice(p)], src)
            self.comm.broadcast(self.winv[g.winv_slice(p)], src)

    # -- P1 + P2 + P3 ------------------------------------------------------------------------------
    @staticmethod
    def slice_bounds(ns: int, world: int):
        per = -(-ns // world)
        return [(min(r * per, ns), min((r + 1) * per, ns)) for r in range(world)]

    def predict_local(self, X, y_pad, Xs_local, ns_local, mean_local, var_local):
        with self.ops.on(False):
            self.ops.predict(X, y_pad, self.packed, self.winv, self.alpha, Xs_local, ns_local, mean_local, var_local)


def _int32(ops):
    t = getattr(ops, "torch", None)
    return t.int32 if t is not None else np.int32


def _copy(dst, src):
    if hasattr(dst, "copy_"):
        dst.copy_(src)
    else:
        dst[...] = src
