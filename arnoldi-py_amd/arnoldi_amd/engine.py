"""Device engine behind ``partial_schur`` / ``arnoldi_decomposition``.

``ArnoldiContext`` owns one row shard's basis V, the device copy of H and the
workspace, and exposes the two O(n) seams of the reference's driver:

  * ``expand(H, start, end, tol)``  == arnoldi_decomposition(A, V, H, tol,
    start_dim=start, max_dim=end)              (src/arnoldi/decomposition.py:13-68)
  * ``truncate(Qp, m, p)``          == V[:, :p] = V[:, :m] @ Qp ; V[:, p] = V[:, m]
                                                   (src/arnoldi/krylov_schur.py:78,81)

With a CSR operator the whole expansion is one C call (``aks_arnoldi_expand`` on the
operator's ``aks_shard``) that enqueues every kernel with no host round trip -- on one GPU,
and on every rank of a row-sharded solve over RCCL, where the same call also issues the
ghost exchange of each SpMV and the all-reduces of the (J+1)-vector between the
Gram-Schmidt stages on the library's own communicator.  Only opaque host operators and
non-RCCL process groups (gloo: CPU tests) chain the stage kernels from Python.
"""
from __future__ import annotations

import ctypes as C
import gc
import os

import numpy as np

from . import _hip, device as dev, mem
from .dist import row_offsets, split_local_rows

C128 = np.complex128


class CsrOperator:
    """CSR operator resident in HBM, row-sharded over ``comm`` (or whole on one GPU)."""

    def __init__(self, A=None, *, local_rows=None, offsets=None, comm=None, device=None, spmv_form=None,
                 real=False, exchange_plan=None):
        """``spmv_form``: None/"auto" (by the structure of the matrix -- ``DeviceCSR.autotune``: the tile-binned
        form for large scattered matrices, the sliced form for matrices with column locality and rows of similar
        length, the CSR-stream kernel otherwise; AKS_SPMV_TUNE=measure times the candidate instead), "csr",
        "binned" or "sliced".  The form fixes the order in which a row's products are summed, i.e. the last
        bits of a solve; it is reported in ``partial_schur(..., stats=)`` as ``spmv_form``.  Environment
        override: AKS_SPMV_FORM.  ``real``: the operator works on real vectors (real-packed Krylov basis):
        real-vector SpMV kernels, float64 ghost exchange.  ``exchange_plan``: ``(dist.GhostPlan, asked)`` made by the caller
        instead of ``split_local_rows`` + ``comm.exchange_requests`` -- rehearsals of the exchange on ONE rank, whose
        communicator then sends the "ghost" entries to itself (tests/capture_exchange_worker.py)."""
        form = os.environ.get("AKS_SPMV_FORM", spmv_form or "auto")
        force = None if form == "auto" else form
        self.comm = comm
        self.real = bool(real)
        world = comm.size if comm is not None else 1
        rank = comm.rank if comm is not None else 0
        if local_rows is not None:
            assert offsets is not None, "local_rows needs the global row offsets"
            rows = dev.canonical_csr(local_rows)
            self.offsets = np.asarray(offsets, dtype=np.int64)
        else:
            full = dev.canonical_csr(A)
            assert full.shape[0] == full.shape[1]
            self.offsets = (np.asarray(offsets, dtype=np.int64) if offsets is not None
                            else row_offsets(full.shape[0], world, full.indptr if world > 1 else None))
            rows = full[int(self.offsets[rank]): int(self.offsets[rank + 1])] if world > 1 else full
        self.n = int(self.offsets[-1])
        self.r0, self.r1 = int(self.offsets[rank]), int(self.offsets[rank + 1])
        self.n_local = self.r1 - self.r0
        self.dtype = rows.dtype
        if self.real and np.iscomplexobj(rows.data):
            raise ValueError("real arithmetic needs a real matrix")
        if world > 1:
            # Every rank must describe the same partition of the same matrix BEFORE the first data-path
            # collective: ranks that disagree would otherwise die inside the all-to-all of the ghost plan
            # ("collective mismatch") instead of getting an error message.  One small all-gather.
            mine = np.concatenate([self.offsets, [rows.shape[0], rows.shape[1]]]).astype(np.int64)
            views = comm.allgather_int64(mine)
            for r, theirs in enumerate(views):
                ok = (len(theirs) == len(mine) and np.array_equal(theirs[:-2], mine[:-2])
                      and theirs[-1] == mine[-1] and theirs[-2] == theirs[r + 1] - theirs[r])
                if not ok:
                    raise ValueError(
                        f"row shards disagree: rank {r} has offsets {list(map(int, theirs[:-2]))} and a "
                        f"{int(theirs[-2])} x {int(theirs[-1])} block, rank {rank} has offsets "
                        f"{list(map(int, mine[:-2]))} and a {rows.shape[0]} x {rows.shape[1]} block")
            if rows.shape[1] != self.n or len(self.offsets) != world + 1:
                raise ValueError(f"shard of shape {rows.shape} does not fit offsets {self.offsets.tolist()}")
        if world == 1 and exchange_plan is None:
            self.diag = dev.DeviceCSR(rows, device)
            self.spmv_form = self.diag.autotune(force=force, real=self.real)
            self.off = None
            self.n_ghost = self.n_send = 0
            self.any_exchange = False
            self._build_shard()
            return
        plan = exchange_plan[0] if exchange_plan is not None else split_local_rows(rows, self.offsets, rank)
        self.diag = dev.DeviceCSR(plan.diag, device)
        self.off = dev.DeviceCSR(plan.off, device) if plan.off is not None else None
        self.spmv_form = self.diag.autotune(force=force, real=self.real)
        if self.off is not None:
            self.off.autotune(force=force, real=self.real)
        self.n_ghost = plan.n_ghost
        self.recv_counts = [int(c) for c in plan.recv_counts]
        asked = exchange_plan[1] if exchange_plan is not None else comm.exchange_requests(plan.ghost_cols, plan.recv_counts)
        self.send_counts = [int(a.shape[0]) for a in asked]
        send_idx = (np.concatenate(asked) - self.r0).astype(np.int32) if sum(self.send_counts) else np.zeros(0, np.int32)
        assert send_idx.size == 0 or (send_idx.min() >= 0 and send_idx.max() < self.n_local)
        d = self.diag.device
        self.send_idx = mem.upload(send_idx, d)
        self.n_send = int(send_idx.size)
        vec_dtype = mem.f64 if self.real else mem.c128
        self.sendbuf = mem.zeros(max(self.n_send, 1), vec_dtype, d)
        self.ghostbuf = mem.zeros(max(self.n_ghost, 1), vec_dtype, d)
        self.any_exchange = bool(sum(comm.allgather_int64([self.n_send + self.n_ghost])[r][0]
                                     for r in range(world)))
        self._build_shard()

    def _build_shard(self):
        """``aks_shard`` for the C entry points (``aks_shard_apply``, ``aks_arnoldi_expand``): the two CSR blocks
        with the SpMV form in use and, over RCCL, the communicator and the exchange plan."""
        sh = _hip.Shard()
        self.diag.block(sh.diag)
        if self.off is not None:
            self.off.block(sh.off)
        handle = self.comm.native() if (self.comm is not None and self.comm.active) else None
        self.native_comm = handle is not None
        if handle is not None:
            world = self.comm.size
            sh.comm = handle
            sh.any_exchange = int(self.any_exchange)
            if self.any_exchange:
                self._counts = ((C.c_int64 * world)(*self.send_counts), (C.c_int64 * world)(*self.recv_counts))
                sh.send_counts, sh.recv_counts = self._counts
                sh.d_send_idx, sh.n_send = self.send_idx.data_ptr(), self.n_send
                sh.d_sendbuf, sh.d_ghostbuf, sh.n_ghost = self.sendbuf.data_ptr(), self.ghostbuf.data_ptr(), self.n_ghost
        self.shard = sh
        # in C whenever this rank's collectives can be issued from C: one GPU, or RCCL
        self.c_driven = self.comm is None or not self.comm.active or self.native_comm

    def check_comm(self):
        """The shard carries the raw handle of the Comm's RCCL communicator: once that Comm has been closed the
        handle dangles, so every C call that would use it is refused here."""
        if self.native_comm and self.comm._native is None:
            raise _hip.HipLibraryError("the communicator this operator was built on has been closed")

    @property
    def shape(self):
        return (self.n, self.n)

    @property
    def nnz(self):
        return self.diag.nnz + (self.off.nnz if self.off is not None else 0)

    def form_defers(self):
        """Whether the library honours AKS_EXPAND_DEFER_SCALE for this operator's diagonal block (its own rule,
        ``block_defers`` in csrc/aks_kernels.hip): the binned form, or the sliced form with a mean padded row length
        of at most 8.  (The control block's ``deferred`` word says what an expansion actually did.)"""
        if self.spmv_form == "binned":
            return True
        if self.spmv_form == "sliced":
            d = self.diag.sliced.desc
            return int(d.n_slices) > 0 and int(d.nnz_pad) <= 8 * 64 * int(d.n_slices)
        return False

    def algorithmic_bytes(self):
        return (self.diag.algorithmic_bytes(self.real)
                + (self.off.algorithmic_bytes(self.real) if self.off is not None else 0))

    def apply(self, x, y, ws=None):
        """y = A x for this shard's rows; x, y are columns of V (local rows; real-packed if ``real``)."""
        real = self.real
        if self.c_driven:
            self.check_comm()
            rc = _hip.load().aks_shard_apply(C.byref(self.shard), dev._ptr(x), dev._ptr(y),
                                             dev._ptr(ws.buf) if ws is not None else C.c_void_p(0), dev._stream(),
                                             _hip.EXPAND_REAL_PACKED if real else 0)
            _hip.check(rc, "aks_shard_apply")
            return
        if self.comm is None or self.comm.size == 1 or not self.any_exchange:
            self.diag.spmv(x, y, False, ws, real)
            return
        w = 1 if real else 2                     # float64 words per vector entry
        if self.n_send:
            (dev.gather_f64 if real else dev.gather_c128)(self.n_send, self.send_idx, x, self.sendbuf)
        handle = self.comm.alltoallv_start(
            self.sendbuf.view(mem.f64)[: w * self.n_send], self.send_counts,
            self.ghostbuf.view(mem.f64)[: w * self.n_ghost], self.recv_counts, words=w)
        self.diag.spmv(x, y, False, ws, real)    # overlaps the exchange
        self.comm.alltoallv_finish(handle)
        if self.off is not None:
            self.off.spmv(self.ghostbuf, y, True, ws, real)


class HostOperator:
    """Opaque operator (LinearOperator, anything with ``@``): the matvec runs wherever the
    caller's object runs it, through host memory.  Functional, not fast."""

    def __init__(self, A, device=None):
        self.A = A
        self.n = int(A.shape[0])
        self.n_local, self.r0, self.r1 = self.n, 0, self.n
        self.comm = None
        self.dtype = np.dtype(getattr(A, "dtype", C128))
        self.offsets = np.array([0, self.n], np.int64)
        self.shape = (self.n, self.n)

    def apply(self, x, y, ws=None):
        hx = x[: self.n].cpu().numpy()
        hy = np.asarray(self.A @ hx, dtype=C128).reshape(-1)
        y[: self.n].copy_(mem.host(np.ascontiguousarray(hy)))


class NullOperator:
    """Sizes only: for contexts that orthogonalise / combine columns but never apply A."""

    def __init__(self, n):
        self.n = self.n_local = int(n)
        self.r0, self.r1, self.comm = 0, int(n), None
        self.shape = (self.n, self.n)

    def apply(self, x, y, ws=None):
        raise TypeError("this context has no operator")


def as_operator(A, comm=None, device=None, real=False):
    if isinstance(A, CsrOperator):
        if A.real != bool(real):
            raise ValueError("the CsrOperator was built for %s vectors" % ("real" if A.real else "complex"))
        return A
    if isinstance(A, HostOperator):
        if real:
            raise TypeError("real arithmetic needs a CSR operator")
        return A
    if dev.canonical_csr(A) is not None:
        return CsrOperator(A, comm=comm, device=device, real=real)
    if real:
        raise TypeError("real arithmetic needs a sparse/dense matrix, not an opaque operator")
    if comm is not None and comm.size > 1:
        raise TypeError("row-sharded solves need a sparse/dense matrix, not an opaque operator")
    return HostOperator(A, device)


class ArnoldiContext:
    def __init__(self, op, max_dim, device=None):
        if max_dim > _hip.MAX_DIM:
            raise _hip.HipLibraryError(f"max_dim={max_dim} exceeds the kernels' limit {_hip.MAX_DIM}")
        self.op = op
        self.comm = op.comm
        self.max_dim = int(max_dim)
        # real-packed mode follows the operator: a real basis is stored two rows per complex slot and every
        # panel kernel runs on basis.n_rows = ceil(n_local / 2) rows (include/arnoldi_hip.h, "real-packed")
        self.real = bool(getattr(op, "real", False))
        self.basis = dev.KrylovBasis(op.n_local, max_dim, device, real=self.real)
        self.ws = dev.Workspace(self.basis.n_rows, max_dim, device, real=self.real)
        self.matvecs = 0
        self._hess_mask = np.arange(self.max_dim + 1)[:, None] <= np.arange(self.max_dim)[None, :] + 1   # rows i <= j + 1
        self.probe = None   # optional _hip.Probe (bench.py): device time of SpMV / ortho launches
        self.spmv_events = None  # optional list (bench.py, Python-chained path): torch event pairs
        self.force_chained = False  # run the Python-chained stage path even on one GPU (tests, bench)
        # hipGraph replay of re-expansions: AKS_GRAPH=1 always, =0 never; default "auto" = for shards of at most
        # 4M rows, whose kernels last 20-150 us, so that the restart rate follows the host's launch latency (same box,
        # 2-D Laplace n = 1M: 50 / 119 restarts/s eager in a cold / warm process, 125 replayed in both)
        mode = os.environ.get("AKS_GRAPH", "auto")
        self.use_graph = mode == "1" or (mode == "auto" and getattr(op, "n_local", 1 << 62) <= 4_000_000)
        self._graphs = {}
        self._graphs_on_comm = 0    # ... of which captured operations of the communicator (counted there too)
        # look-ahead operator application (see expand): off with AKS_LOOKAHEAD=0
        self.allow_lookahead = os.environ.get("AKS_LOOKAHEAD", "1") != "0"
        # multi-rank: leave the third all-reduce out until a step turns out to need a second DGKS pass
        self.lazy_third = os.environ.get("AKS_LAZY_THIRD", "1") != "0"
        self.lazy_redos = 0
        # deferred normalisation of new basis columns (see expand): AKS_DEFER_SCALE=0 switches it off
        self.allow_defer = os.environ.get("AKS_DEFER_SCALE", "1") != "0"
        # ... for expansions of at most this many steps: a raw column is divided again by every panel kernel that
        # reads it until the restart -- (2..3) x (steps left) times per entry against ONE 32-byte pass saved.  Measured
        # through whole restarts (profiles/r03_defer_sell_ab.txt, r03_defer_size_ab.txt): 10 steps per restart
        # (k = 5, m = 20): +2.8 % random CSR, +3 % Markov; 25 steps with a second pass in each (2-D Laplace, m = 40):
        # -0.5 ... -2.8 % at every size.
        self.defer_max_steps = int(os.environ.get("AKS_DEFER_MAX_STEPS", "12"))
        self._raw_from = None       # first raw column of the basis, if an expansion left any
        self._raw_scale = {}        # column -> its scale beta (the host's copy of the workspace's colscale)
        self.deferred_expansions = 0
        self.graph_capture_failures = 0          # captures that failed and fell back to eager launches
        self.discarded_second_passes = self.discarded_steps = self.discarded_applies = 0   # work of repeated expansions
        self.last_ctrl = None
        self._coef_stage = self._coef_dev = None      # pinned / device buffers of the restart coefficients
        self._coef_turn = 0
        self._look = None           # scratch columns; [_look_col] holds A V[:, end] of the last expansion
        self._look_col = 0
        self._look_valid = False
        self.lookahead_applies = 0  # operator applications issued ahead of time (the last one of a solve is unused)

    # -- seam 1 ------------------------------------------------------------------
    def expand(self, H, start, end, tol, eta=dev.ETA_DGKS, *, lookahead=False, consume_lookahead=False,
               defer_scale=False):
        """Run Arnoldi steps j = start..end-1 on the device, then mirror the new columns
        of H into the host array exactly as the reference's in-place writes would.
        Returns n_iter (== end unless a step broke down).

        ``lookahead``: after the last step, also queue ``A V[:, end]`` into a scratch column.  H and the
        control block are copied back *before* it in stream order, so the host gets them while that SpMV
        (and its ghost exchange) still runs: the Krylov-Schur driver's host Schur step then overlaps
        device work instead of leaving the GPU idle.  ``consume_lookahead``: the caller guarantees that
        ``V[:, start]`` now holds the vector the look-ahead was applied to (Krylov-Schur:
        ``V[:, p] = V[:, m]``), so the first step starts from the stored product.  Results are identical
        with and without (same kernels, same operands).

        ``defer_scale``: the caller's next operation on the basis is ``truncate`` / ``truncate_active`` (the
        Krylov-Schur drivers): the new columns may then stay RAW -- the device keeps their norms as scales and every
        kernel that reads them divides on the fly, instead of a 32 n byte normalisation pass per step
        (AKS_EXPAND_DEFER_SCALE; honoured by the C-driven path when the diagonal block is in the binned form, or in the
        sliced form with short rows: ``CsrOperator.form_defers``).
        H is bit for bit the same (the on-the-fly division is the one the normalisation pass performs); the
        truncation that follows multiplies raw columns by coefficients scaled on the host (``_fold_scales``), a
        rounding-level difference.  Until then the columns ``> start`` must not be read by anything else
        (``local_columns`` / ``gather_columns`` refuse)."""
        b, ws, op = self.basis, self.ws, self.op
        # The reference's arnoldi_decomposition keeps no state between calls: a breakdown in the
        # last step of one expansion (n_iter == max_dim, accepted by the driver) must not turn the
        # next expansion into a no-op.  Clear the control block's (broken, n_iter) words.
        w_ready = bool(consume_lookahead and self._look_valid and end > start)
        self._look_valid = False
        native = isinstance(op, CsrOperator) and op.c_driven and not self.force_chained
        defer = bool(defer_scale and native and self.allow_defer and op.form_defers()
                     and (end - start <= self.defer_max_steps or self._raw_from is not None))
        if self._raw_from is not None and (start > self._raw_from or (start == self._raw_from and not defer)):
            raise _hip.HipLibraryError("expansion from a raw column: the basis must be truncated first")
        multi = self.comm is not None and self.comm.active
        # Multi-rank: the norm after a second DGKS pass needs a third all-reduce.  While no step has needed a
        # second pass it is left out (two collectives per step); the control block's second_passes count --
        # decided from all-reduced numbers, so identical on every rank -- tells afterwards whether a step
        # did need it, and then the expansion is repeated with the third all-reduce (and keeps it from then
        # on).  Its inputs V[:, :start+1] (and the look-ahead product) are untouched by the failed attempt.
        lazy = multi and self.lazy_third
        passes_before = int(self.last_ctrl.second_passes) if self.last_ctrl is not None else 0
        steps_before = int(self.last_ctrl.steps_done) if self.last_ctrl is not None else 0
        want_look = lookahead and self.allow_lookahead and isinstance(op, CsrOperator) and end > start
        if want_look and self._look is None:
            # two scratch columns: the product consumed by this expansion stays intact (a repeated
            # expansion needs it again) while the next look-ahead product is written to the other one
            self._look = dev.DeviceColumns(b.n_rows, 2, b.device)
        while True:
            ws.buf[:8].zero_()
            if w_ready:
                b.V[start + 1].copy_(self._look.V[self._look_col])       # device-to-device, stream-ordered
            if native:
                self._expand_native(start, end, tol, eta, w_ready, lazy, defer)
            else:
                self._expand_chained(start, end, tol, eta, w_ready, lazy, multi)
            fetch = dev.fetch_H_and_ctrl(b, ws)             # queued before the look-ahead, waited for after it
            if want_look and defer:                          # column `end` is raw: its scale comes from the workspace
                op.check_comm()
                rc = _hip.load().aks_shard_apply_col(
                    C.byref(op.shard), dev._ptr(b.V), b.ldv, end, dev._ptr(self._look.col(1 - self._look_col)),
                    dev._ptr(ws.buf), ws.nbytes, ws.max_dim, dev._stream(), _hip.EXPAND_REAL_PACKED if self.real else 0)
                _hip.check(rc, "aks_shard_apply_col")
            elif want_look:
                op.apply(b.col(end), self._look.col(1 - self._look_col), ws)   # a device no-op after a breakdown
            Hd, ctrl = fetch()
            if native and op.native_comm:
                _hip.comm_status(op.comm._native)           # a one-shot reduction that timed out (NaN in H) is an error here
            if lazy and int(ctrl.second_passes) != passes_before:
                self.lazy_third = lazy = False              # a step needed the second pass: do it over, exactly
                self.lazy_redos += 1
                # the device counters are cumulative: what the discarded attempt added is kept apart, so that
                # ``second_passes`` / ``steps`` describe the expansions whose results were used
                self.discarded_second_passes += int(ctrl.second_passes) - passes_before
                self.discarded_steps += int(ctrl.steps_done) - steps_before
                self.discarded_applies += (end - start - (1 if w_ready else 0)) + (1 if want_look else 0)
                passes_before, steps_before = int(ctrl.second_passes), int(ctrl.steps_done)
                continue
            break
        if want_look:
            self.lookahead_applies += 1
            self._look_col = 1 - self._look_col
            self._look_valid = not ctrl.broken
        n_iter = int(ctrl.n_iter) if ctrl.broken else end
        if defer and n_iter > start and not int(ctrl.deferred):
            # the library did not honour the flag (its own check of the block's form, or a stand-in): the columns are
            # normalised -- which is only consistent if the start column was
            if self._raw_from is not None:
                raise _hip.HipLibraryError("the expansion normalised its columns although it started from a raw one")
            defer = False
        if defer and n_iter > start:
            # columns start+1 .. n_iter hold beta_c v_c until the next truncation (beta_c = H[c, c-1]); column `start`
            # itself may be the raw column the previous truncation carried over (its scale is already recorded)
            if self._raw_from is None or self._raw_from > start:
                self._raw_from = start + 1
            for c in range(start + 1, n_iter + (0 if ctrl.broken else 1)):
                self._raw_scale[c] = float(Hd[c, c - 1].real)
            self.deferred_expansions += 1
        self.matvecs += n_iter - start
        if not np.iscomplexobj(H):
            Hd = Hd.real                                  # real-packed mode: a real Hessenberg matrix
        # the reference's in-place writes: column j gets rows 0 .. j+1 (0 .. j when the step broke down:
        # decomposition.py:61-63 returns before H[j+1, j] is written) -- one masked copy instead of a Python loop
        # over the columns (the host's share of a restart matters at the 8-GPU shard sizes)
        if n_iter > start:
            rows = min(H.shape[0], self.max_dim + 1)
            mask = self._hess_mask[:rows, start:n_iter]
            if ctrl.broken:
                mask = mask.copy()
                mask[n_iter, n_iter - 1 - start] = False
            np.copyto(H[:rows, start:n_iter], Hd[:rows, start:n_iter], where=mask)
        self.last_ctrl = ctrl
        return n_iter

    def _expand_native(self, start, end, tol, eta, w_ready, lazy, defer=False):
        """One C call: every kernel (and, over RCCL, every collective) of the expansion is enqueued by
        ``aks_arnoldi_expand`` with no host round trip."""
        b, ws, op = self.basis, self.ws, self.op
        flags = ((_hip.EXPAND_FROM_W if w_ready else 0) | (_hip.EXPAND_REAL_PACKED if self.real else 0)
                 | (_hip.EXPAND_LAZY_THIRD if lazy else 0) | (_hip.EXPAND_DEFER_SCALE if defer else 0))

        op.check_comm()

        def enqueue():
            rc = _hip.load().aks_arnoldi_expand(
                C.byref(op.shard), dev._ptr(b.V), b.ldv, dev._ptr(b.H), self.max_dim, start, end, tol, eta,
                dev._ptr(ws.buf), ws.nbytes, ws.max_dim,
                self.probe.handle if self.probe is not None else C.c_void_p(0), dev._stream(), flags)
            _hip.check(rc, "aks_arnoldi_expand")

        # The re-expansion (start = p) is the same launch sequence with the same arguments at
        # every restart (DGKS decisions and breakdown are taken on the device), so it can be
        # captured once into a hipGraph and replayed: one host call per restart instead of
        # ~10 launches per Arnoldi step (AKS_GRAPH; pays off when the host is slow
        # relative to the kernels).  Not used while a probe records per-kernel events.  With a communicator the
        # sequence contains the library's collectives -- see ``_comm_capturable``.
        key = (start, end, float(tol), float(eta), w_ready, lazy, defer)
        graph_ok = not op.native_comm or self._comm_capturable()
        if self.use_graph and self.probe is None and start > 0 and graph_ok and b.V.is_cuda:
            g = self._graphs.get(key)
            if g is None:
                gc_was_on = gc.isenabled()
                gc.disable()     # a collection during capture could free device objects (illegal in capture)
                try:
                    g = mem.Graph(enqueue)       # torch's capture API, or hipStreamBeginCapture / EndCapture (mem.py)
                except Exception as e:           # noqa: BLE001  a capture that did not come about (another thread's device-wide
                    # call invalidated it, the runtime refused a node): nothing of the sequence has run -- launch it eagerly,
                    # now and from here on; a graph is an optimisation, never a reason for a solve to fail.  It is COUNTED
                    # (``graph_capture_failures``, in ``partial_schur(stats=)``): the tests that replay graphs assert 0, so
                    # the capture invalidation round 5 fixed (mem._capture_lock) cannot come back behind this fall-back
                    self.use_graph = False
                    self.graph_capture_failures += 1
                    import warnings

                    warnings.warn(f"hipGraph capture of the re-expansion failed ({type(e).__name__}: {str(e)[:120]}); "
                                  "this solve launches eagerly", RuntimeWarning, stacklevel=2)
                    enqueue()
                    return
                finally:
                    if gc_was_on:
                        gc.enable()
                self._graphs[key] = g
                if op.native_comm:               # the communicator must outlive this graph: counted there (dist: close())
                    op.comm.adopt_graph_owner(self)
                    self._graphs_on_comm += 1
            g.replay()
        else:
            enqueue()

    def _comm_capturable(self):
        """Whether this rank's launch sequence WITH its collectives may be captured (AKS_GRAPH_COMM; all ranks read the same
        environment, so all decide alike):

        ``0`` (default)  never: a sharded expansion is launched eagerly.  At the shard sizes of the BASELINE configs on 8
                         GPUs (1.25M - 2M rows) a kernel lasts 20-60 us against ~4 us to launch it: not host-bound.
        ``1``            sequences whose only collectives are the stage reductions (``ncclAllReduce`` or the one-shot
                         kernel, which keeps its call counter on the device for exactly this purpose).
        ``exchange``     also sequences with a ghost exchange -- ONLY on a HIP runtime >= 7.2: a grouped ncclSend / ncclRecv
                         captured on a stream that JOINED the capture through an event (the communicator's side stream)
                         sends the end-of-capture walk of HIP 7.0.51831 (what a torch wheel bundles) into an unbounded
                         recursion, 174 573 nested frames of hip::Stream::EndCapture() (profiles/r05_capture_crash.txt);
                         the system's ROCm 7.2 captures and replays the very sequence
                         (test_a_ghost_exchange_is_capturable_on_the_system_runtime).  On an older runtime the value is
                         ignored and the sequence stays eager (test_sequences_with_a_ghost_exchange_are_never_captured).

        A graph that captured communicator operations must be destroyed BEFORE the communicator (ncclCommDestroy never
        returns otherwise): the context registers itself with the Comm, whose ``close()`` drops the graphs first, and the
        library refuses to destroy a communicator with graphs still counted on it."""
        mode = os.environ.get("AKS_GRAPH_COMM", "0")
        if mode not in ("1", "exchange"):
            return False
        if not self.op.any_exchange:
            return True
        return mode == "exchange" and _hip.runtime_versions()["hip_runtime"] >= 70200000

    def drop_graphs(self):
        """Destroy every captured graph now (and give the communicator its counts back)."""
        graphs, self._graphs = self._graphs, {}
        for g in graphs.values():
            g.destroy()
        n, self._graphs_on_comm = self._graphs_on_comm, 0
        comm = self.comm
        if n and comm is not None and getattr(comm, "_native", None) is not None:
            lib = _hip.load()
            for _ in range(n):
                lib.aks_comm_graph_release(comm._native)

    def __del__(self):
        try:
            if getattr(self, "_graphs_on_comm", 0):
                self.drop_graphs()
        except Exception:              # noqa: BLE001  (interpreter shutdown)
            pass

    def _expand_chained(self, start, end, tol, eta, w_ready, lazy, multi):
        """The same stages chained from Python: opaque host operators, and row-sharded solves whose
        collectives go through torch.distributed (gloo: CPU tests, several test ranks on one GPU)."""
        b, ws, op = self.basis, self.ws, self.op
        hbase = b.H.data_ptr()
        raw = isinstance(op, CsrOperator)      # columns as raw addresses: no per-step tensor views
        vbase, stride = b.V.data_ptr(), 16 * b.ldv
        with dev.cached_stream():
            for j in range(start, end):
                J = j + 1
                x = vbase + stride * j if raw else b.col(j)
                w = vbase + stride * J if raw else b.col(J)
                if w_ready and j == start:
                    pass                           # w = A V[:, start] was applied ahead of time
                elif self.spmv_events is not None:   # bench.py: device time of the sharded SpMV
                    e0, e1 = mem.Event(enable_timing=True), mem.Event(enable_timing=True)
                    e0.record()
                    op.apply(x, w, ws)
                    e1.record()
                    self.spmv_events.append((e0, e1))
                else:
                    op.apply(x, w, ws)
                if not multi:
                    dev.dgks_gs_device(b, J, w, hbase + 16 * j, self.max_dim, tol, ws, eta)
                    continue
                dev.gs_project(b, J, w, ws)
                self.comm.allreduce_sum_(ws.red(1, J + 1))
                dev.gs_update_project(b, J, w, ws)
                self.comm.allreduce_sum_(ws.red(2, J + 1))
                dev.gs_update_norm(b, J, w, ws, eta)
                if not lazy:
                    self.comm.allreduce_sum_(ws.red(3, 1))
                dev.gs_finish(b, J, w, hbase + 16 * j, self.max_dim, tol, ws, eta)

    def collectives_per_step(self):
        """Data-path collectives one Arnoldi step issues on the multi-rank path (0 on one GPU): the ghost
        exchange of the SpMV plus the all-reduces between the Gram-Schmidt stages."""
        if self.comm is None or not self.comm.active:
            return 0
        exchange = 1 if getattr(self.op, "any_exchange", False) else 0
        return exchange + (2 if self.lazy_third else 3)

    # -- seam 2 ------------------------------------------------------------------
    def _fold_scales(self, Q, col0, m, p):
        """Rows of the restart coefficients that multiply RAW columns are divided by those columns' scales (the
        columns hold beta v), and the book-keeping moves on: column ``col0 + p`` becomes a bit copy of column
        ``col0 + m`` and inherits its scale."""
        Q = np.array(Q, dtype=C128, copy=True).reshape(m, p)
        if self._raw_from is not None:
            first = max(self._raw_from, col0)
            if first < col0 + m:
                scales = np.array([self._raw_scale[c] for c in range(first, col0 + m)])
                Q[first - col0:, :] /= scales[:, None]
            carried = self._raw_scale.get(col0 + m) if col0 + m >= self._raw_from else None
            self._raw_scale = {} if carried is None else {col0 + p: carried}
            self._raw_from = None if carried is None else col0 + p
        return Q

    def _upload_coefficients(self, Q):
        """The restart coefficients (m x p complex128, a few KB) to the device: through one of two alternating pinned
        staging buffers with an asynchronous copy, so that the truncation kernel is enqueued right behind it -- a
        pageable ``.to(device)`` is a synchronous staged copy and put ~30 us of idle device time in front of every
        restart compression (profiles/r03_small_trace.txt: __amd_rocclr_copyBuffer -> k_truncate_mfma).  A staging
        buffer is reused two restarts later; the H read-back of the expansion in between has been waited for by then,
        which orders the host behind the copy that used it."""
        b = self.basis
        Q = np.ascontiguousarray(Q, dtype=C128)
        if not b.V.is_cuda or os.environ.get("AKS_COEF_UPLOAD") == "sync":     # CPU tensors (tests/fake_hip.py); A/B
            return mem.upload(Q, b.device)
        if self._coef_stage is None or self._coef_stage[0].numel() < Q.size:
            cap = max(Q.size, self.max_dim * self.max_dim)
            self._coef_stage = [mem.pinned_empty(cap, mem.c128) for _ in range(2)]
            self._coef_dev = [mem.empty(cap, mem.c128, b.device) for _ in range(2)]
            self._coef_copied = [None, None]
        self._coef_turn ^= 1
        stage, out = self._coef_stage[self._coef_turn], self._coef_dev[self._coef_turn]
        if self._coef_copied[self._coef_turn] is not None:
            # the copy that last read this staging buffer: long finished in the drivers' restart loop (see above); a
            # caller that compresses several times without waiting for the device in between is held here instead of
            # overwriting bytes a copy engine is still reading
            self._coef_copied[self._coef_turn].synchronize()
        stage[: Q.size].numpy()[:] = Q.reshape(-1)
        out[: Q.size].copy_(stage[: Q.size], non_blocking=True)
        if self._coef_copied[self._coef_turn] is None:
            self._coef_copied[self._coef_turn] = mem.Event()
        self._coef_copied[self._coef_turn].record()
        return out[: Q.size]

    def truncate(self, Qp, m, p):
        Qd = self._upload_coefficients(self._fold_scales(Qp, 0, m, p))
        dev.truncate(self.basis, m, p, Qd, self.ws)         # (raw columns x scaled coefficients: normalised results)

    def truncate_active(self, Zp, l, m, p):
        """Restart compression behind ``l`` locked columns (krylov_schur_locking.py):
        ``V[:, l:p] = V[:, l:m] @ Zp`` and ``V[:, p] = V[:, m]`` -- ``aks_truncate`` on the sub-basis that
        starts at column ``l``; the locked columns are neither read nor written."""
        b = self.basis
        Zd = self._upload_coefficients(self._fold_scales(Zp, l, m - l, p - l))
        dev.truncate(b, m - l, p - l, Zd, self.ws, col0=l)

    # -- data movement --------------------------------------------------------------
    def set_start_vector(self, v_full):
        self.basis.set_col(0, v_full[self.op.r0: self.op.r1])

    def _need_normalised(self, j1, what):
        """Every entry point that READS basis columns below ``j1`` outside the expansion / truncation pair goes through
        here: columns left raw by an expansion with deferred normalisation hold ``beta v`` until the basis has been
        truncated, and only the library's own readers know their scales (one check, every reader)."""
        if self._raw_from is not None and j1 > self._raw_from:
            raise _hip.HipLibraryError(f"{what}: columns >= {self._raw_from} of an expansion with deferred normalisation "
                                       "are raw until the basis has been truncated")

    def local_columns(self, j0, j1):
        self._need_normalised(j1, "local_columns")
        return self.basis.get_cols(j0, j1)

    def gather_columns(self, j0, j1):
        """Full (n, j1-j0) host array on every rank."""
        loc = self.local_columns(j0, j1)
        if self.comm is None or self.comm.size == 1:
            return np.asfortranarray(loc)
        return np.asfortranarray(self.comm.allgather_rows(loc))

    # -- verification on the device -----------------------------------------
    def true_residuals(self, T):
        """``(vals, ||A v_k - l_k v_k||, the same / |l_k|)`` for the eigenpairs of the partial Schur
        form held in the first ``k = T.shape[0]`` basis columns -- what the reference's
        ``RitzDecomposition.compute_true_residuals`` (decomposition.py:134-146) and its scripts
        compute on the host -- without moving any n-vector off the GPU: ``aks_combine`` forms
        ``vecs = Q S``, ``residual_norms`` does the rest."""
        k = T.shape[0]
        vals, S = np.linalg.eig(T)
        self._need_normalised(k, "true_residuals")
        if self.real:
            res = self._residual_norms_real(k, S, vals)
            return vals, res, res / np.abs(vals)
        vecs = self.combine(0, k, S)                              # aks_combine: vecs = Q S, out of place
        res = self.residual_norms(vecs, vals)
        return vals, res, res / np.abs(vals)

    def residual_norms_real_pairs(self, k, S, vals):
        """``||A u_i - vals[i] u_i||`` for ``u_i = V[:, :k] @ S[:, i]`` on a real-packed basis (``S`` complex: each
        eigenvector is a pair of real combinations of the real basis columns)."""
        assert self.real
        return self._residual_norms_real(k, S, vals)

    def _residual_norms_real(self, k, S, vals):
        """Real-packed basis (``Q`` real, eigenvectors ``u = Q S`` complex): with ``u = ur + i ui``,
        ``ur = Q Re S`` and ``ui = Q Im S`` are two real combinations (``aks_combine`` with real coefficients),
        ``A u`` is two real operator applications, and
        ``Re r = A ur - (lr ur - li ui)``, ``Im r = A ui - (li ur + lr ui)`` are J = 2 fused updates of the
        real-packed stage kernels on the panel ``[ur, ui]`` whose norm outputs add up to ``||A u - l u||^2``.
        Nothing of length n leaves the device."""
        b, ws, lib = self.basis, self.ws, _hip.load()
        self._clear_ctrl()
        S = np.asarray(S, dtype=C128).reshape(k, -1)
        out = np.zeros(len(vals))
        y = self._scratch_col()
        args = (dev._ptr(ws.buf), ws.nbytes, ws.max_dim, dev._stream())
        red1, red2 = ws.red(1, 2), ws.red(2, 3)
        for i, lam in enumerate(np.asarray(vals, dtype=C128)):
            pair = self.combine(0, k, np.stack([S[:, i].real, S[:, i].imag], axis=1).astype(C128))   # [ur, ui]
            total = 0.0
            for col, coef in ((0, (lam.real, -lam.imag)), (1, (lam.imag, lam.real))):
                self.op.apply(pair.col(col), y, ws)
                red1.copy_(mem.host(np.array([coef[0], 0.0, coef[1], 0.0], dtype=np.float64)))
                _hip.check(lib.aks_gs_update_project(b.n_rows, 2, dev._ptr(pair.V), pair.ldv, dev._ptr(y), *args),
                           "aks_gs_update_project")
                if self._multi():
                    self.comm.allreduce_sum_(red2)
                total += float(red2[4].item())
            out[i] = np.sqrt(total)
        return out

    # -- building blocks of the explicit-restart solvers ---------------------
    # All of them work on columns of the resident basis; nothing of length n moves to the host.
    def _clear_ctrl(self):
        """Stage kernels are no-ops once the control block says ``broken`` (a happy breakdown in the
        last expansion): clear (broken, n_iter) before using them outside an expansion."""
        self.ws.buf[:8].zero_()

    def _multi(self):
        return self.comm is not None and self.comm.active

    def mgs(self, k, j, tol):
        """``mgs(V[:, :k], V[:, j], tol)`` of src/arnoldi/explicit_restarts.py:64-78: modified
        Gram-Schmidt of column ``j`` against columns ``0..k-1`` one at a time (each a J = 1 projection
        + fused update of the DGKS stage kernels), then normalisation.  Returns beta."""
        b, ws, lib = self.basis, self.ws, _hip.load()
        self._clear_ctrl()
        w = b.V.data_ptr() + 16 * b.ldv * j
        self._need_normalised(max(k, j + 1), "mgs")
        args = (dev._ptr(ws.buf), ws.nbytes, ws.max_dim, dev._stream())
        multi = self._multi()
        for i in range(max(k, 1)):
            vi = b.V.data_ptr() + 16 * b.ldv * i if k else w       # k == 0: only ||w||^2 is wanted
            _hip.check(lib.aks_gs_project(b.n_rows, 1, vi, b.ldv, w, *args), "aks_gs_project")
            if multi:
                self.comm.allreduce_sum_(ws.red(1, 2))
            if k:
                _hip.check(lib.aks_gs_update_project(b.n_rows, 1, vi, b.ldv, w, *args), "aks_gs_update_project")
        slot = ws.red(2, 2) if k else ws.red(1, 2)                   # [<v, w>, ||w||^2] after the last step
        if multi and k:
            self.comm.allreduce_sum_(slot)
        beta = float(np.sqrt(slot[2].item()))
        assert beta > tol, "MGS: Too small norm when orthornormalizing"      # explicit_restarts.py:74-75
        dev.scale(b.n_rows, w, 1.0 / beta)
        return beta

    def ritz_vector_into_first(self, k, m, s):
        """``V[:, k] = V[:, k:m] @ s`` in place (the restart vector of both explicit-restart solvers,
        explicit_restarts.py:59 and :140): ``aks_truncate`` with p = 1 on the sub-basis that starts at
        column k.  (Its second effect, ``V[:, k+1] = V[:, m]``, lands in a column the next expansion
        recomputes.)"""
        b = self.basis
        Sd = mem.upload(np.ascontiguousarray(np.asarray(s, dtype=C128).reshape(m - k, 1)), b.device)
        self._need_normalised(m, "ritz_vector_into_first")
        rc = _hip.load().aks_truncate(b.n_rows, m - k, 1, b.V.data_ptr() + 16 * b.ldv * k, b.ldv,
                                      dev._ptr(Sd), dev._stream())
        _hip.check(rc, "aks_truncate")

    def ritz_vectors_into_first(self, k, m, S):
        """``V[:, k:k+q] = V[:, k:m] @ S`` in place for a few (q >= 1) coefficient columns -- the two real vectors
        that span a conjugate pair's invariant subspace in the real-arithmetic explicit-restart solver."""
        b = self.basis
        S = np.ascontiguousarray(np.asarray(S, dtype=C128).reshape(m - k, -1))
        q = S.shape[1]
        assert 1 <= q <= m - k
        self._need_normalised(m, "ritz_vectors_into_first")
        Sd = mem.upload(S, b.device)
        rc = _hip.load().aks_truncate(b.n_rows, m - k, q, b.V.data_ptr() + 16 * b.ldv * k, b.ldv,
                                      dev._ptr(Sd), dev._stream())
        _hip.check(rc, "aks_truncate")

    def _scratch_col(self):
        if getattr(self, "_scratch", None) is None:
            self._scratch = dev.DeviceColumns(self.basis.n_rows, 1, self.basis.device)
        return self._scratch.col(0)

    def rayleigh_column(self, k, ncols=None):
        """``[vdot(V[:, i], A @ V[:, k]) for i < ncols]`` (default ``ncols = k + 1``: explicit_restarts.py:149-150):
        one operator application into a scratch column and one ``ncols``-column projection."""
        b, ws = self.basis, self.ws
        ncols = k + 1 if ncols is None else int(ncols)
        self._need_normalised(max(k + 1, ncols), "rayleigh_column")
        self._clear_ctrl()
        y = self._scratch_col()
        self.op.apply(b.col(k), y, ws)
        dev.gs_project(b, ncols, y, ws)
        red = ws.red(1, ncols + 1)
        if self._multi():
            self.comm.allreduce_sum_(red)
        return red[: 2 * ncols].cpu().numpy().view(C128).copy()

    def residual_norms(self, block, values, j0=0):
        """``||A u_i - values[i] u_i||`` for the device columns ``u_i = block.col(j0 + i)`` (``block`` is
        the basis or a ``DeviceColumns``) -- RitzDecomposition.compute_true_residuals,
        decomposition.py:134-146: operator application into a scratch column, then
        ``aks_gs_update_project`` with J = 1 and coefficient values[i], whose norm output is
        ``||A u_i - values[i] u_i||^2``."""
        values = np.atleast_1d(np.asarray(values, dtype=C128))
        if self.real and values.imag.any():
            # a real-packed column is a REAL vector: it can only be an eigenvector of a real eigenvalue
            raise ValueError("real-packed basis: the columns are real vectors, so the values must be real; the "
                             "residuals of complex eigenpairs (pairs of real-packed columns) come from "
                             "residual_norms_real_pairs() / true_residuals()")
        b, ws, lib = self.basis, self.ws, _hip.load()
        if block is self.basis:
            self._need_normalised(j0 + len(values), "residual_norms")
        self._clear_ctrl()
        lam = mem.upload(values, b.device).view(mem.f64)
        y = self._scratch_col()
        red1, red2 = ws.red(1, 1), ws.red(2, 2)
        out = mem.zeros(len(values), mem.f64, b.device)
        for i in range(len(values)):
            u = block.col(j0 + i)
            self.op.apply(u, y, ws)
            red1.copy_(lam[2 * i: 2 * i + 2])
            _hip.check(lib.aks_gs_update_project(b.n_rows, 1, dev._ptr(u), block.ldv, dev._ptr(y), dev._ptr(ws.buf),
                                                 ws.nbytes, ws.max_dim, dev._stream()), "aks_gs_update_project")
            if self._multi():
                self.comm.allreduce_sum_(red2)
            out[i: i + 1].copy_(red2[2: 3])
        return np.sqrt(out.cpu().numpy())

    def combine(self, j0, m, S, out=None):
        """``V[:, j0:j0+m] @ S`` into a new (or the given) ``DeviceColumns`` block, out of place."""
        self._need_normalised(j0 + m, "combine")
        return dev.combine_columns(self.basis, j0, m, S, out)

    def gather_block(self, block):
        """Host copy of a ``DeviceColumns`` block with all rows (every rank), Fortran order."""
        loc = block.get_cols()
        if self.comm is None or self.comm.size == 1:
            return np.asfortranarray(loc)
        return np.asfortranarray(self.comm.allgather_rows(loc))


def default_comm():
    """The communicator of a solve that was given none.  With ``AKS_COMM=host`` and WORLD_SIZE > 1 in the environment: the
    torch-free ``dist.HostComm`` (TCP rendezvous + the library's communicator) -- on either allocator backend.  On the torch
    backend (``AKS_HOST_ALLOC=torch``) also: a ``Comm`` over the default torch.distributed group when the caller has
    initialised one with more than one rank.  The default (HIP runtime) backend cannot ride on a torch process group --
    torch's collectives run on torch's streams and allocator -- and says so instead of solving unsharded on every rank."""
    import sys

    if os.environ.get("AKS_COMM") == "host" and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        from .dist import host_comm_from_env

        return host_comm_from_env()
    if "torch.distributed" not in sys.modules:          # nobody can have initialised what was never imported
        return None
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if mem.BACKEND != "torch":
            raise _hip.HipLibraryError(
                "torch.distributed is initialised with several ranks, but arnoldi_amd runs on its torch-free backend: start the "
                "process with AKS_HOST_ALLOC=torch to shard over the torch process group, or with AKS_COMM=host to shard over "
                "dist.HostComm, or pass comm= explicitly")
        from .dist import comm_for

        return comm_for()            # one Comm (and one RCCL communicator) per process group, not per solve
    return None
