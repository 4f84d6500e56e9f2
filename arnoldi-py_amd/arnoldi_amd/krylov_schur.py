"""Krylov-Schur partial Schur decomposition -- the drop-in entry point.

Host driver with the reference's control flow (src/arnoldi/krylov_schur.py:10-114):
argument defaults, assertions, the m x m Schur/reorder step (SciPy LAPACK on the
host), H bookkeeping, convergence test, ``History`` and exceptions are the
reference's; the two O(n) seams -- Arnoldi expansion and basis compression -- run on
the MI355X through ``engine.ArnoldiContext``.
"""
from __future__ import annotations

import numpy as np

from .engine import ArnoldiContext, as_operator, default_comm
from .history import History
from .utils import StartVector, arg_largest_magnitude, complex_schur, host_blas_threads, reorder_schur

WORK_DTYPE = np.complex128  # krylov_schur.py:38: complex128 whatever A.dtype is


class KrylovSchurSolver:
    """State machine of one solve: ``start()`` then alternate ``contract()`` /
    ``expand()``.  ``partial_schur`` below is the reference-shaped loop over it; the
    benchmark times the same two calls."""

    def __init__(self, A, nev, max_dim, p, tol, sort_function, *, v0=None, comm=None, device=None):
        n = A.shape[0]
        self.n, self.nev, self.max_dim, self.p = n, nev, max_dim, p
        self.tol, self.sort_function = tol, sort_function
        # every rank draws the full vector so that the shards agree bit for bit with the
        # single-GPU (and the reference's) start vector; the draw runs while the operator is set up (utils.StartVector)
        drawn = StartVector(n, WORK_DTYPE, v0)
        try:
            self.op = as_operator(A, comm=comm, device=device)
            self.ctx = ArnoldiContext(self.op, max_dim, device)
        finally:
            start = drawn.get()
        start = np.asarray(start, dtype=WORK_DTYPE)
        assert start.shape == (n,)
        self.ctx.set_start_vector(start)
        self.H = np.zeros((max_dim + 1, max_dim), dtype=WORK_DTYPE)
        self.history = History.from_k(nev)
        self.m = 0
        self.restarts_run = 0
        self.schur_memo = {}          # utils.complex_schur: remembers that the real route met 2 x 2 blocks

    def start(self):
        """Initial m-step expansion (krylov_schur.py:51-54)."""
        self.m = self.ctx.expand(self.H, 0, self.max_dim, self.tol, lookahead=True, defer_scale=True)
        return self.m

    def contract(self, restart):
        """Rotation, truncation and convergence test of restart number ``restart``
        (krylov_schur.py:63-101).  Returns True when the first nev estimates are < tol."""
        H, m, p, nev = self.H, self.m, self.p, self.nev
        booked = restart * (self.max_dim - nev) + (m - nev)          # krylov_schur.py:63

        # Ordered Schur form of the projected matrix (host, LAPACK).  The reference calls
        # zgees twice (krylov_schur.py:69 and utils.py:45); the second call sees an upper-
        # triangular matrix and returns (T, I) unchanged, so one call followed by the same
        # ?trexc sequence gives the same (T, Q).  (complex_schur: zgees, or dgees while H is exactly real
        # and its spectrum is -- a third of the time, utils.py.)
        T, Q = complex_schur(H[:m, :m], self.schur_memo)
        T, Q = reorder_schur(T, Q, self.sort_function(np.diag(T)))

        # truncation on the device: V[:, :p] <- V[:, :m] Q[:, :p];  V[:, p] <- V[:, m]
        Qp = Q[:, :p]
        self.ctx.truncate(Qp, m, p)

        coupling = H[m, :m].copy()
        last = H[m, m - 1]
        H[:p, :p] = T[:p, :p]                                        # krylov_schur.py:83
        H[p, :p] = coupling @ Qp                                     # krylov_schur.py:86-87
        H[p, p:] = 0                                                 # krylov_schur.py:88

        estimate = np.abs(last * Q[m - 1, :]) / np.abs(np.diag(T))   # krylov_schur.py:91-92
        under = estimate[:nev] <= self.tol
        self.history.matvecs[under] = booked
        self.history.restarts[under] = restart + 1
        self.restarts_run = restart + 1
        self.estimate = estimate[:nev]
        return bool(np.all(estimate[:nev] < self.tol))               # krylov_schur.py:99

    def contract_invariant(self, restart):
        """Happy breakdown (the expansion stopped at ``m < max_dim`` because ``A V_m = V_m H_m`` holds
        to the invariance tolerance): the eigenvalues of ``H[:m, :m]`` are eigenvalues of A.  Rotate the
        wanted ``nev`` Schur vectors to the front and stop -- the reference has this as a TODO
        (krylov_schur.py:57-59); only reached with ``on_breakdown="deflate"``."""
        H, m, nev = self.H, self.m, self.nev
        if m < nev:
            raise ValueError(f"Happy breakdown: invariant subspace of dimension {m} < nev = {nev}")
        T, Q = complex_schur(H[:m, :m], self.schur_memo)
        T, Q = reorder_schur(T, Q, self.sort_function(np.diag(T)))
        self.ctx.truncate(Q[:, :nev], m, nev)
        H[:nev, :nev] = T[:nev, :nev]
        H[nev:, :nev] = 0
        self.history.matvecs[:] = restart * (self.max_dim - nev) + (m - nev)
        self.history.restarts[:] = restart + 1
        self.restarts_run = restart + 1
        self.estimate = np.zeros(nev)
        return True

    def expand(self):
        """Re-expansion from p to max_dim (krylov_schur.py:103-106).  Its first product
        ``A V[:, p]`` (= ``A V[:, m]`` of the previous cycle, ``contract`` copies that column) was
        queued at the end of the previous expansion and ran while the host did the Schur step."""
        self.m = self.ctx.expand(self.H, self.p, self.max_dim, self.tol, lookahead=True, consume_lookahead=True,
                                 defer_scale=True)
        return self.m

    def true_residuals(self):
        """Eigenvalues of the converged partial Schur form with ``||A v - l v||`` and
        ``||A v - l v|| / |l|`` evaluated on the device (no n-vector leaves the GPU)."""
        return self.ctx.true_residuals(self.H[: self.nev, : self.nev])

    def result(self, gather=True):
        comm = self.ctx.comm
        if comm is not None and comm.size > 1 and not gather:
            Q = np.asfortranarray(self.ctx.local_columns(0, self.nev))
        else:
            Q = self.ctx.gather_columns(0, self.nev)
        return Q, self.H[: self.nev, : self.nev].copy(), self.history


def partial_schur(A, nev, *, max_dim=None, stopping_criterion=None, max_restarts=100,
                  sort_function=None, p=None, v0=None, comm=None, device=None, gather=True,
                  stats=None, on_breakdown="raise", arithmetic="complex", locking=False):
    """Compute ``nev`` Schur vectors ``Q`` and the ``nev x nev`` upper-triangular ``T``
    with ``A Q ~= Q T`` by the Krylov-Schur algorithm.

    Positional/keyword arguments up to ``p`` are the reference's.  Extra, optional,
    keyword-only:

    v0      start vector (length n); default ``rand_normalized_vector(n)`` (global RNG,
            identical stream to the reference).
    comm    ``dist.Comm`` for a row-sharded multi-GPU solve; default: the default
            torch.distributed group if it is initialised with more than one rank.
    device  this rank's GPU: an index, "cuda:i" / a torch device on the torch backend (default: the current device).
    gather  multi-GPU only: return the full ``Q`` on every rank (True) or just this
            rank's rows (False).
    on_breakdown  "raise" (default, the reference's behaviour) or "deflate": when the Arnoldi expansion
            stops early because the Krylov space is A-invariant, return the ``nev`` wanted Schur
            vectors of that space instead of raising.
    arithmetic  "complex" (default: the reference's complex128 iteration, identical restart history) or
            "real": for a real matrix and a real start vector, iterate in real arithmetic on a
            real-packed basis (krylov_schur_real.py) -- half the memory traffic; same ``(Q, T)`` contract,
            restart counts may differ from the reference's.  "auto" picks "real" whenever it applies.
    locking  False (default: the reference's iteration, identical restart history) or True: lock converged
            Schur vectors and let the restart size grow with them (krylov_schur_locking.py; the
            reference's TODO, README.md:116).  Same ``(Q, T)`` contract; restart counts differ.
    stats   optional dict that receives ``restarts``, ``matvecs`` (true operator
            applications), ``second_passes``, ``graphs_captured`` / ``graph_capture_failures`` and the solver object.

    Returns ``(Q, T, history)``; raises ``ValueError("Has not converged !")`` /
    ``ValueError("Happy breakdown not supported yet")`` like the reference.
    """
    if stopping_criterion is None:
        tol = np.sqrt(np.finfo(A.dtype).eps)        # krylov_schur.py:16-17
    else:
        tol = stopping_criterion
    if sort_function is None:
        sort_function = arg_largest_magnitude
    assert max_restarts > 0

    n = A.shape[0]
    assert A.shape[1] == n
    if max_dim is None:
        max_dim = min(max(2 * nev + 1, 20), n)      # krylov_schur.py:29-30
    if p is None:
        p = min(nev + 5, max_dim - 1)               # size of the basis kept at a restart
    assert nev <= p < max_dim
    assert on_breakdown in ("raise", "deflate")
    assert arithmetic in ("complex", "real", "auto")
    if arithmetic == "auto":      # real whenever it applies: a real CSR-able matrix and a real start vector
        from . import device as _dev
        from .engine import CsrOperator

        real_matrix = not np.issubdtype(np.dtype(A.dtype), np.complexfloating)
        real_start = v0 is None or not (np.iscomplexobj(v0) and np.asarray(v0).imag.any())
        csr_able = (A.real if isinstance(A, CsrOperator) else _dev.canonical_csr(A) is not None)
        arithmetic = "real" if (real_matrix and real_start and csr_able and max_dim >= nev + 2) else "complex"

    if comm is None:
        comm = default_comm()
    if arithmetic == "real":
        from .krylov_schur_real import RealKrylovSchurSolver, RealLockingKrylovSchurSolver

        if np.issubdtype(np.dtype(A.dtype), np.complexfloating):
            raise ValueError("arithmetic='real' needs a real matrix")
        if max_dim < nev + 2:      # room to keep the partner of a conjugate pair cut at nev
            raise ValueError("arithmetic='real' needs max_dim >= nev + 2")
        if locking and on_breakdown != "raise":
            raise ValueError("on_breakdown='deflate' is not implemented together with locking=True")
        cls = RealLockingKrylovSchurSolver if locking else RealKrylovSchurSolver
        solver = cls(A, nev, max_dim, p, tol, sort_function, v0=v0, comm=comm, device=device)
    elif locking:
        from .krylov_schur_locking import LockingKrylovSchurSolver

        if on_breakdown != "raise":
            raise ValueError("on_breakdown='deflate' is not implemented together with locking=True")
        solver = LockingKrylovSchurSolver(A, nev, max_dim, p, tol, sort_function, v0=v0, comm=comm, device=device)
    else:
        solver = KrylovSchurSolver(A, nev, max_dim, p, tol, sort_function, v0=v0, comm=comm, device=device)

    converged = False
    with host_blas_threads():          # the m x m host LAPACK between two device waits: one BLAS thread (utils.py)
        solver.start()
        for restart in range(max_restarts):
            if solver.m != max_dim:
                if on_breakdown != "deflate":
                    raise ValueError("Happy breakdown not supported yet")   # krylov_schur.py:57-59
                converged = solver.contract_invariant(restart)
                break
            converged = solver.contract(restart)
            if converged:
                break
            solver.expand()

    if stats is not None:
        ctx = solver.ctx
        stats.update(restarts=solver.restarts_run, matvecs=ctx.matvecs,
                     second_passes=int(ctx.last_ctrl.second_passes) - ctx.discarded_second_passes, solver=solver,
                     lazy_redos=ctx.lazy_redos, discarded_operator_applies=ctx.discarded_applies,
                     deferred_normalisations=ctx.deferred_expansions,     # expansions whose new columns stayed raw
                     lookahead_applies=ctx.lookahead_applies, arithmetic=arithmetic,
                     graphs_captured=len(ctx._graphs),                     # re-expansions replayed as hipGraphs, and
                     graph_capture_failures=ctx.graph_capture_failures,   # captures that fell back to eager launches (0!)
                     tol=float(tol), max_dim=int(max_dim), p=int(p),
                     spmv_form=getattr(solver.op, "spmv_form", None),      # which kernel applied A: decides the order
                     spmv_form_chosen_by=getattr(getattr(solver.op, "diag", None), "tune_mode", None),   # of a row's sum

                     locked=int(getattr(solver, "locked", 0)), truncation_bytes=list(getattr(solver, "trunc_bytes", [])))
    if not converged:
        raise ValueError("Has not converged !")                      # krylov_schur.py:108-109
    return solver.result(gather)
