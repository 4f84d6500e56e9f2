"""Krylov-Schur in real arithmetic for real matrices -- the reference's "real arithmetic (real Schur,
``dtrexc``)" TODO (README.md:112-119, utils.py:64-65), opt-in through
``partial_schur(..., arithmetic="real")``.

For a real matrix and a real start vector the Krylov basis is real.  The device keeps it *real-packed*
(two rows per complex128 slot, include/arnoldi_hip.h) and runs the very same panel kernels on half the
rows: half the HBM traffic of Gram-Schmidt and of the restart compression, 8-byte gathers and products
in the SpMV, half the ghost-exchange volume between GPUs.  The host works with the real Schur form
(``dgees`` + ``dtrexc``): 1x1 and 2x2 diagonal blocks; the restart size moves by one when it would cut
a conjugate pair.  The result is converted to the reference's contract at the end
(``scipy.linalg.rsf2csf``): unitary ``Q``, complex upper-triangular ``T`` with ``A Q = Q T``.

This is a different (mathematically equivalent) iteration from the complex one: where the complex
driver keeps exactly ``p`` Schur vectors -- possibly one member of a conjugate pair -- this one keeps the
pair.  Restart counts can therefore differ from the reference's; eigenpairs and residuals agree to the
stopping tolerance (tests/test_gpu_real.py).
"""
from __future__ import annotations

import numpy as np
import scipy.linalg
from scipy.linalg.lapack import dtrexc, strexc

from .engine import ArnoldiContext, as_operator
from .history import History
from .utils import StartVector


def real_blocks(T):
    """``[(start, size)]`` of the 1x1 / 2x2 diagonal blocks of a real quasi-triangular matrix."""
    m = T.shape[0]
    out, i = [], 0
    while i < m:
        size = 2 if (i + 1 < m and T[i + 1, i] != 0.0) else 1
        out.append((i, size))
        i += size
    return out


def block_eigenvalues(T, blocks=None):
    """Eigenvalues of a real quasi-triangular matrix, one per diagonal position (a 2x2 block gives
    ``a + ib`` at its first row and ``a - ib`` at its second)."""
    ev = np.zeros(T.shape[0], dtype=np.complex128)
    for s, size in (real_blocks(T) if blocks is None else blocks):
        if size == 1:
            ev[s] = T[s, s]
        else:
            lam = np.linalg.eigvals(T[s: s + 2, s: s + 2])
            ev[s], ev[s + 1] = lam[np.argsort(-lam.imag)]      # (two real values if the block is not a pair)
    return ev


def reorder_real_schur(T, Z, sort_function):
    """Reorder the diagonal blocks of the real Schur form ``(T, Z)`` with ``dtrexc`` so that they
    follow ``sort_function`` (a full permutation of the eigenvalues, best first, as in
    utils.arg_largest_*); a 2x2 block takes the better rank of its two members and moves as a unit."""
    m = T.shape[0]
    blocks = real_blocks(T)
    ev = block_eigenvalues(T, blocks)
    rank = np.empty(m, dtype=np.int64)
    rank[np.asarray(sort_function(ev))] = np.arange(m)
    keys = [int(rank[s: s + size].min()) for s, size in blocks]
    sizes = [size for _, size in blocks]
    wanted = sorted(range(len(blocks)), key=lambda b: keys[b])
    current = list(range(len(blocks)))
    T = np.asfortranarray(T)
    Z = np.asfortranarray(Z)
    swap = strexc if T.dtype == np.float32 else dtrexc
    for target, bid in enumerate(wanted):
        src = current.index(bid)
        if src == target:
            continue
        ifst = 1 + sum(sizes[b] for b in current[:src])
        ilst = 1 + sum(sizes[b] for b in current[:target])
        T, Z, info = swap(T, Z, ifst, ilst)
        if info < 0:
            raise np.linalg.LinAlgError(f"dtrexc: illegal argument {-info}")
        # info == 1: two blocks too close to swap (T is still a valid Schur form, partially reordered)
        current.pop(src)
        current.insert(target, bid)
    return T, Z


class RealKrylovSchurSolver:
    """Same ``start`` / ``contract`` / ``expand`` / ``result`` protocol as ``KrylovSchurSolver``."""

    def __init__(self, A, nev, max_dim, p, tol, sort_function, *, v0=None, comm=None, device=None):
        n = A.shape[0]
        self.n, self.nev, self.max_dim, self.p = n, nev, max_dim, p
        self.tol, self.sort_function = tol, sort_function
        drawn = StartVector(n, np.float64, v0)             # the reference's draw (its imaginary part is 0), made while
        try:                                               # the operator is set up
            self.op = as_operator(A, comm=comm, device=device, real=True)
            self.ctx = ArnoldiContext(self.op, max_dim, device)
        finally:
            start = drawn.get()
        if v0 is not None:
            start = np.asarray(v0)
            if np.iscomplexobj(start):
                if start.imag.any():
                    raise ValueError("real arithmetic needs a real start vector")
                start = start.real
        assert start.shape == (n,)
        self.ctx.set_start_vector(np.ascontiguousarray(start, dtype=np.float64))
        self.H = np.zeros((max_dim + 1, max_dim), dtype=np.float64)
        self.history = History.from_k(nev)
        self.m = 0
        self.p_now = p
        self.nev_now = nev
        self.restarts_run = 0

    def start(self):
        self.m = self.ctx.expand(self.H, 0, self.max_dim, self.tol, lookahead=True, defer_scale=True)
        return self.m

    def contract(self, restart):
        H, m, nev = self.H, self.m, self.nev
        T, Z = scipy.linalg.schur(H[:m, :m], output="real")
        T, Z = reorder_real_schur(T, Z, self.sort_function)

        def cut(k):                       # never cut a 2x2 block: move the boundary up (or down at the end)
            if 0 < k < m and T[k, k - 1] != 0.0:
                return k + 1 if k + 1 < m else k - 1
            return k

        nev_now = min(cut(nev), m - 1)
        p = cut(self.p)
        if p < nev_now:
            p = nev_now
        assert 1 <= p < m
        self.ctx.truncate(Z[:, :p], m, p)                 # V[:, :p] = V[:, :m] Z_p ; V[:, p] = V[:, m]
        last = H[m, m - 1]
        coupling = last * Z[m - 1, :p]
        H[:] = 0.0
        H[:p, :p] = T[:p, :p]
        H[p, :p] = coupling

        ev = block_eigenvalues(T)
        tail = np.abs(Z[m - 1, :])
        for s, size in real_blocks(T):
            if size == 2:                                 # both members of a pair share the block's residual
                tail[s] = tail[s + 1] = np.hypot(Z[m - 1, s], Z[m - 1, s + 1])
        with np.errstate(divide="ignore", invalid="ignore"):
            estimate = np.abs(last) * tail / np.abs(ev)
        under = estimate[:nev] <= self.tol
        self.history.matvecs[under] = self.ctx.matvecs
        self.history.restarts[under] = restart + 1
        self.restarts_run = restart + 1
        self.estimate = estimate[:nev_now]
        self.p_now, self.nev_now = p, nev_now
        return bool(np.all(estimate[:nev_now] < self.tol))

    def contract_invariant(self, restart):
        """Happy breakdown (``A V_m = V_m H_m`` to the invariance tolerance, ``m < max_dim``): the eigenvalues of
        ``H[:m, :m]`` are eigenvalues of A.  Rotate the wanted ``nev`` real Schur vectors (one more if that
        completes a conjugate pair) to the front and stop -- ``on_breakdown="deflate"``, as the complex driver."""
        H, m, nev = self.H, self.m, self.nev
        if m < nev:
            raise ValueError(f"Happy breakdown: invariant subspace of dimension {m} < nev = {nev}")
        T, Z = scipy.linalg.schur(H[:m, :m], output="real")
        T, Z = reorder_real_schur(T, Z, self.sort_function)
        k = nev
        if 0 < k < m and T[k, k - 1] != 0.0:              # never cut a 2x2 block
            k += 1
        self.ctx.truncate(Z[:, :k], m, k)
        H[:] = 0.0
        H[:k, :k] = T[:k, :k]
        self.history.matvecs[:] = self.ctx.matvecs
        self.history.restarts[:] = restart + 1
        self.restarts_run = restart + 1
        self.estimate = np.zeros(k)
        self.p_now = self.nev_now = k
        return True

    def expand(self):
        self.m = self.ctx.expand(self.H, self.p_now, self.max_dim, self.tol, lookahead=True,
                                 consume_lookahead=True, defer_scale=True)
        return self.m

    def true_residuals(self):
        """Eigenvalues of the converged real partial Schur form with ``||A v - l v||`` and ``||A v - l v|| / |l|``
        evaluated on the device (real-packed kernels; no n-vector leaves the GPU).  All ``nev_now`` values: the
        partner of a pair cut at ``nev`` is included."""
        k = self.nev_now
        return self.ctx.true_residuals(self.H[:k, :k])

    def result(self, gather=True):
        """``(Q, T, history)`` in the reference's form: the real partial Schur pair (``nev`` columns, one
        more if that completes a conjugate pair) is rotated to complex upper-triangular form on the host
        (an O(n nev^2) product) and cut back to ``nev`` columns."""
        k = self.nev_now
        comm = self.ctx.comm
        if comm is not None and comm.size > 1 and not gather:
            Qr = np.asfortranarray(self.ctx.local_columns(0, k))
        else:
            Qr = self.ctx.gather_columns(0, k)
        Tc, U = scipy.linalg.rsf2csf(self.H[:k, :k].copy(), np.eye(k))
        Q = Qr @ U
        return np.asfortranarray(Q[:, : self.nev]), np.array(Tc[: self.nev, : self.nev]), self.history


class RealLockingKrylovSchurSolver(RealKrylovSchurSolver):
    """Locking and a dynamic restart size (krylov_schur_locking.py) on the real-packed basis: only the active
    block ``S = H[l:m, l:m]`` is rotated (real Schur form, ``dtrexc`` by the caller's sort key); converged leading
    1x1 / 2x2 blocks are locked as a whole (a conjugate pair locks, and counts, as two columns), the restart
    compression multiplies the active columns only, and neither ``l`` nor ``p`` ever cuts a pair."""

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        self.p0 = self.p
        self.locked = 0
        self.trunc_bytes = []
        self.locked_history = []

    def contract(self, restart):
        H, m, nev, l, tol = self.H, self.m, self.nev, self.locked, self.tol
        T, Z = scipy.linalg.schur(H[l:m, l:m], output="real")
        T, Z = reorder_real_schur(T, Z, self.sort_function)
        ma = m - l
        blocks = real_blocks(T)
        ev = block_eigenvalues(T, blocks)
        beta = H[m, m - 1]
        tail = np.abs(Z[-1, :])
        for s, size in blocks:
            if size == 2:
                tail[s] = tail[s + 1] = np.hypot(Z[-1, s], Z[-1, s + 1])
        with np.errstate(divide="ignore", invalid="ignore"):
            estimate = np.abs(beta) * tail / np.abs(ev)

        newly = 0                                           # lock leading wanted blocks, whole blocks only
        for s, size in blocks:
            if l + newly >= nev or not np.all(estimate[s: s + size] < tol):
                break
            newly += size
        l_new = l + newly
        done = l_new >= nev

        def no_cut(k):                                      # a boundary inside the active block must not split a pair
            if 0 < k < ma and T[k, k - 1] != 0.0:
                return k + 1 if k + 1 < ma else k - 1
            return k

        p_new = min(self.p0 + min(l_new, (self.max_dim - self.p0) // 2), m - 1)
        p_new = max(p_new, l_new + 1) if not done else max(p_new, l_new)
        pa = no_cut(p_new - l)
        pa = max(pa, newly)
        p_new = l + pa
        assert l_new <= p_new <= m - 1 or (done and p_new <= m)

        Zp = Z[:, :pa]
        self.ctx.truncate_active(Zp, l, m, p_new)           # V[:, l:p_new] = V[:, l:m] Zp ; V[:, p_new] = V[:, m]
        self.trunc_bytes.append(16 * self.ctx.basis.n_rows * (ma + pa + 2))
        R = H[:l, l:m].copy()
        coupling = beta * Z[-1, :pa]
        H[:l, l:p_new] = R @ Zp
        H[l:p_new, l:p_new] = T[:pa, :pa]
        H[p_new:, :p_new] = 0.0
        H[p_new, l:p_new] = coupling
        H[p_new, l:l_new] = 0.0                              # deflation: |coupling| < tol |theta|
        H[:, p_new:] = 0.0

        under = np.zeros(nev, bool)
        under[: min(l_new, nev)] = True
        k_act = max(min(nev - l, len(estimate)), 0)
        under[l: l + k_act] |= estimate[:k_act] <= tol
        first = under & (self.history.restarts == 0)
        self.history.matvecs[first] = self.ctx.matvecs
        self.history.restarts[first] = restart + 1
        self.restarts_run = restart + 1
        est = np.zeros(max(nev, l_new))
        est[l: l + k_act] = estimate[:k_act]
        self.estimate = est
        self.locked = l_new
        self.p_now = p_new
        # the columns returned: nev, or one more when a locked / leading pair straddles position nev
        k_out = max(nev, l_new) if done else nev
        if not done and l <= nev - 1 < l + ma - 1 and T[nev - l, nev - l - 1] != 0.0:
            k_out = nev + 1
        self.nev_now = k_out
        self.locked_history.append(l_new)
        return done

    def contract_invariant(self, restart):
        raise ValueError("Happy breakdown with locking=True is not supported; use locking=False")
