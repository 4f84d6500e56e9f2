"""Explicitly restarted Arnoldi solvers -- drop-ins for ``naive_explicit_restarts`` and
``explicit_restarts_with_deflation`` (src/arnoldi/explicit_restarts.py:31-61, 81-168) on the same
device seam as ``partial_schur``.

The basis never leaves HBM.  Per restart the host sees ``H`` (small), one norm (the ``mgs``
assertion) and, on convergence, the k+1 Rayleigh coefficients; everything of length n -- the Arnoldi
expansion, the restart vector ``V_m s``, modified Gram-Schmidt against the locked vectors, ``A v`` for
the residual / Rayleigh column, the final ``V Y`` -- is a kernel launch (``engine.ArnoldiContext``).
Control flow, defaults, book-keeping, assertions and exceptions follow the reference line by line so
that the same seed gives the same ``History``.
"""
from __future__ import annotations

import numpy as np

from .decomposition import RitzDecomposition
from .engine import ArnoldiContext, as_operator, default_comm
from .history import History  # noqa: F401  (re-exported: ``from arnoldi.explicit_restarts import History``)
from .utils import arg_largest_magnitude, host_blas_threads, rand_normalized_vector

WORK_DTYPE = np.complex128


def mgs(basis, w, tol):
    """Modified Gram-Schmidt of ``w`` (n,) against the columns of ``basis`` (n, k), then normalisation;
    ``w`` is modified in place and returned (explicit_restarts.py:64-78).  Host arrays in, device
    kernels in between: one J = 1 projection + fused update per basis column, one scale."""
    from .engine import NullOperator

    n, k = basis.shape
    ctx = ArnoldiContext(NullOperator(n), max(k, 1))
    if k:
        ctx.basis.set_cols(0, basis)
    ctx.basis.set_col(k, w)
    ctx.mgs(k, k, tol)
    w[:] = ctx.basis.get_cols(k, k + 1)[:, 0]
    return w


def _tolerance(A, stopping_criterion):
    if stopping_criterion is None:
        return np.sqrt(np.finfo(A.dtype).eps)       # explicit_restarts.py:33-34 / 86-87
    return stopping_criterion


def _column_block(ctx, j):
    """Snapshot of basis column j as a one-column device block (the returned Ritz vector must not
    change when the caller keeps iterating on the context)."""
    from . import device as dev

    blk = dev.DeviceColumns(ctx.basis.n_rows, 1, ctx.basis.device)
    blk.V[0].copy_(ctx.basis.V[j])
    return blk


@host_blas_threads()       # host LAPACK on m x m matrices between device waits: one BLAS thread (utils.py)
def naive_explicit_restarts(A, m=None, *, stopping_criterion=None, max_restarts=10, comm=None, device=None):
    """One eigenpair by m-step Arnoldi restarted from the dominant Ritz vector
    (explicit_restarts.py:31-61).  Returns ``(ritz, has_converged, restarts_used)``.

    The work arrays are complex128 whatever ``A.dtype`` is (the reference promotes to
    ``promote_types(A.dtype, complex64)``; for float64 / complex input that is complex128 as well).
    """
    tol = _tolerance(A, stopping_criterion)
    n = A.shape[0]
    k = 1
    if m is None:
        m = min(max(2 * k + 1, 20), n)
    if comm is None:
        comm = default_comm()
    op = as_operator(A, comm=comm, device=device)
    ctx = ArnoldiContext(op, m, device)
    H = np.zeros((m + 1, m), dtype=WORK_DTYPE)
    inv_tol = float(np.sqrt(np.finfo(A.dtype).eps))           # arnoldi_decomposition's default
    ctx.set_start_vector(rand_normalized_vector(n).astype(WORK_DTYPE))
    ritz = None
    for i in range(max_restarts):
        n_iter = ctx.expand(H, 0, m, inv_tol)
        vals, S = np.linalg.eig(H[:n_iter, :n_iter])          # RitzDecomposition.from_v_and_h, k = 1
        pick = arg_largest_magnitude(vals)[:k]
        s = S[:, pick[0]]
        approx = np.abs(H[n_iter, n_iter - 1] * S[-1, pick])
        ctx.ritz_vector_into_first(0, n_iter, s)              # V[:, 0] = V_m s: the Ritz vector and the next v0
        ritz = RitzDecomposition(vals[pick], None, approx, block=None, ctx=ctx, source=A)
        if approx[0] < tol:
            res = ctx.residual_norms(ctx.basis, vals[pick])
            if res[0] / max(np.abs(vals[pick][0]), tol) < tol:
                ritz._block = _column_block(ctx, 0)
                return ritz, True, i
    ritz._block = _column_block(ctx, 0)
    return ritz, False, max_restarts


@host_blas_threads()
def explicit_restarts_with_deflation(A, nev, *, max_dim=None, stopping_criterion=None, max_restarts=100,
                                     sort_function=None, comm=None, device=None, gather=True, stats=None,
                                     arithmetic="complex"):
    """``nev`` eigenpairs one after the other; converged Schur vectors stay locked in the first
    columns of the basis (explicit_restarts.py:81-168).  Returns ``(eigenvalues, eigenvectors,
    history)``; raises ``ValueError("Could not converge for value k")`` like the reference.

    Extra keyword-only arguments as in ``partial_schur``: ``comm`` / ``device`` (row-sharded multi-GPU
    solve), ``gather`` (full eigenvectors on every rank, or this rank's rows), ``stats`` (dict that
    receives the true operator-application count and the context); ``arithmetic="real"``: for a real matrix, iterate
    on a real-packed basis (half the memory traffic) -- real Ritz values as in the reference, a complex Ritz value
    together with its conjugate as two real Schur vectors (see ``_deflation_real``).
    """
    tol = _tolerance(A, stopping_criterion)
    if sort_function is None:
        sort_function = arg_largest_magnitude
    assert max_restarts > 0
    n = A.shape[0]
    assert A.shape[1] == n
    if max_dim is None:
        max_dim = min(max(2 * nev + 1, 20), n)
    if comm is None:
        comm = default_comm()
    assert arithmetic in ("complex", "real")
    if arithmetic == "real":
        if np.issubdtype(np.dtype(A.dtype), np.complexfloating):
            raise ValueError("arithmetic='real' needs a real matrix")
        return _deflation_real(A, nev, max_dim, float(tol), max_restarts, sort_function, comm, device, gather, stats)
    op = as_operator(A, comm=comm, device=device)
    ctx = ArnoldiContext(op, max_dim, device)
    H = np.zeros((max_dim + 1, max_dim), dtype=WORK_DTYPE)
    history = History.from_k(nev)
    extra_applies = 0

    for k in range(nev):
        v0 = rand_normalized_vector(n, WORK_DTYPE)            # every rank draws the full vector
        ctx.basis.set_col(k, v0[op.r0: op.r1])
        ctx.mgs(k, k, tol)
        for restart in range(max_restarts):
            m = ctx.expand(H, k, max_dim, float(tol))
            assert m > k
            happy_breakdown = m != max_dim
            matvecs = restart * (max_dim - k) + (m - k)

            # RitzDecomposition.from_v_and_h(V[:, k:], H[k:, k:], m - k): only the first vector is used
            vals, S = np.linalg.eig(H[k:m, k:m])
            ind = sort_function(vals)[: m - k]
            approximate_residuals = np.abs(H[m, m - 1] * S[-1, ind])
            ctx.ritz_vector_into_first(k, m, S[:, ind[0]])     # V[:, k] = V[:, k:m] s
            ctx.mgs(k, k, tol)

            with np.errstate(divide="ignore", invalid="ignore"):
                approximate_convergence = approximate_residuals / np.abs(vals[ind])
            has_converged = happy_breakdown or (approximate_convergence[0] < tol)
            if has_converged:
                H[: k + 1, k] = ctx.rayleigh_column(k)          # explicit_restarts.py:149-151
                H[k + 1:-1, k] = 0
                extra_applies += 1
                history.matvecs[k] = matvecs
                history.restarts[k] = restart + 1
                break
        else:
            raise ValueError(f"Could not converge for value {k}")

    # eigenpairs of the final nev x nev block (explicit_restarts.py:160-167)
    eivals, Y = np.linalg.eig(H[:nev, :nev])
    block = ctx.combine(0, nev, Y)
    if stats is not None:
        stats.update(matvecs=ctx.matvecs + extra_applies, ctx=ctx, H=H, tol=float(tol), max_dim=int(max_dim),
                     eigenvectors_device=block)
    if comm is not None and comm.size > 1 and not gather:
        return eivals, np.asfortranarray(block.get_cols()), history
    return eivals, ctx.gather_block(block), history


def _deflation_real(A, nev, max_dim, tol, max_restarts, sort_function, comm, device, gather, stats):
    """Explicit restarts with deflation in real arithmetic (real-packed basis, real H).

    Same outer structure as the reference's solver: Arnoldi from the locked columns, pick the wanted Ritz value of
    the active block, restart from its Ritz vector, lock on convergence.  A REAL wanted Ritz value is handled
    exactly as in the reference.  A COMPLEX one comes with its conjugate: the real and imaginary parts of its Ritz
    vector span the pair's invariant subspace, so the restart continues from the (real) real part, and on
    convergence BOTH are orthonormalised and locked -- two Schur vectors, a 2x2 diagonal block of the real
    quasi-triangular ``H[:K, :K]``.  The pair counts as two of the ``nev`` wanted values; if that makes
    ``K = nev + 1`` the member with negative imaginary part is dropped from the returned eigenpairs."""
    n = A.shape[0]
    if max_dim < nev + 2:
        raise ValueError("arithmetic='real' needs max_dim >= nev + 2")
    op = as_operator(A, comm=comm, device=device, real=True)
    ctx = ArnoldiContext(op, max_dim, device)
    H = np.zeros((max_dim + 1, max_dim), dtype=np.float64)
    history = History.from_k(nev)
    extra_applies = 0
    k = 0
    while k < nev:
        v0 = rand_normalized_vector(n).real                   # the reference's draw is real-valued
        ctx.basis.set_col(k, np.ascontiguousarray(v0[op.r0: op.r1]))
        ctx.mgs(k, k, tol)
        for restart in range(max_restarts):
            m = ctx.expand(H, k, max_dim, tol)
            assert m > k
            happy_breakdown = m != max_dim
            matvecs = restart * (max_dim - k) + (m - k)
            vals, S = np.linalg.eig(H[k:m, k:m])
            i0 = sort_function(vals)[0]
            theta, s = vals[i0], S[:, i0]
            pair = theta.imag != 0.0 and m - k >= 2
            approx = np.abs(H[m, m - 1] * s[-1])
            with np.errstate(divide="ignore", invalid="ignore"):
                has_converged = happy_breakdown or (approx / np.abs(theta) < tol)
            if pair:
                ctx.ritz_vectors_into_first(k, m, np.stack([s.real, s.imag], axis=1))
                ctx.mgs(k, k, tol)
                if has_converged:
                    ctx.mgs(k + 1, k + 1, tol)
            else:
                ctx.ritz_vector_into_first(k, m, s.real)
                ctx.mgs(k, k, tol)
            if has_converged:
                width = 2 if pair else 1
                for c in range(width):
                    col = ctx.rayleigh_column(k + c, k + width).real       # projections on all locked columns
                    H[: k + width, k + c] = col
                    H[k + width:, k + c] = 0.0
                extra_applies += width
                history.matvecs[k: k + width] = matvecs
                history.restarts[k: k + width] = restart + 1
                k += width
                break
        else:
            raise ValueError(f"Could not converge for value {k}")

    K = k
    eivals, Y = np.linalg.eig(H[:K, :K])
    keep = np.arange(K)
    if K > nev:                                               # the last locked block is a pair: drop its lower member
        tail = np.linalg.eigvals(H[K - 2: K, K - 2: K])
        lower = tail[np.argmin(tail.imag)]
        keep = np.delete(keep, int(np.argmin(np.abs(eivals - lower))))
    eivals, Y = eivals[keep], Y[:, keep]
    block_re, block_im = ctx.combine(0, K, Y.real), ctx.combine(0, K, Y.imag)
    if stats is not None:
        stats.update(matvecs=ctx.matvecs + extra_applies, ctx=ctx, H=H, tol=float(tol), max_dim=int(max_dim),
                     locked=K, schur_coefficients=Y, eigenvectors_device_real_imag=(block_re, block_im))

    def host(block):                                          # real-packed columns -> float64 rows of this rank
        loc = np.ascontiguousarray(block.V[:, : ctx.basis.n_rows].cpu().numpy()).view(np.float64)[:, : ctx.basis.n_real].T
        if comm is not None and comm.size > 1 and gather:
            return np.asfortranarray(comm.allgather_rows(np.ascontiguousarray(loc)))
        return np.asfortranarray(loc)

    vecs = host(block_re) + 1j * host(block_im)
    return eivals, np.asfortranarray(vecs), history
