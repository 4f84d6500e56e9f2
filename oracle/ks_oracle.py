"""NumPy/SciPy restatement of the reference's Krylov-Schur path (CPU oracle).

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.

Every function names the reference lines it follows (paths are relative to
``/root/reference``).  The arithmetic is the reference's own: SciPy's
``csr_matvec`` for the operator, OpenBLAS ``zgemv``/``dznrm2`` for the
Gram-Schmidt passes, LAPACK ``zgees``/``ztrexc`` for the dense step, and the
global legacy NumPy RNG for the start vector, so that on the same seed the
oracle reproduces the reference's iterates to rounding.
"""
from __future__ import annotations

import dataclasses
import time

import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp
from scipy.linalg import blas as _blas
from scipy.linalg import lapack as _lapack

C128 = np.complex128
ETA_DGKS = np.sqrt(0.5)  # src/arnoldi/ortho.py:6


# --------------------------------------------------------------------------
# small helpers
# --------------------------------------------------------------------------
def arg_largest_magnitude(vals):
    """src/arnoldi/utils.py:16-17 -- full permutation, largest |x| first."""
    return np.argsort(-np.abs(vals))


def arg_largest_real(vals):
    """src/arnoldi/utils.py:20-21 -- full permutation, largest Re(x) first."""
    return np.argsort(-np.real(vals))


def random_unit_vector(n, dtype=np.float64):
    """src/arnoldi/utils.py:7-13 -- n draws of the GLOBAL legacy RNG, cast, /norm."""
    x = np.random.randn(n).astype(dtype)
    x /= np.linalg.norm(x)
    return x


@dataclasses.dataclass
class History:
    """src/arnoldi/explicit_restarts.py:13-28."""

    matvecs: np.ndarray
    restarts: np.ndarray

    @classmethod
    def from_k(cls, k):
        return cls(np.zeros(k, np.int32), np.zeros(k, np.int32))

    @property
    def k(self):
        return self.matvecs.shape[0]

    @property
    def total_matvecs(self):
        return self.matvecs.sum()


# --------------------------------------------------------------------------
# test matrices
# --------------------------------------------------------------------------
def mark_matrix(m):
    """src/arnoldi/matrices.py:5-73 -- Markov walk on a triangular grid.

    Same triplets as the reference's loop (boundary moves are emitted twice and
    summed by the COO->CSR conversion); loops only over the m grid rows.
    """
    n = m * (m + 1) // 2
    cst = 0.5 / (m - 1)
    rows, cols, vals = [], [], []
    base = 0
    for i in range(m):
        jmax = m - i
        j = np.arange(jmax)
        ix = base + j
        inner = j < jmax - 1
        pd = cst * (i + j + 1)
        pu = 0.5 - cst * (i + j - 1)
        # north (ix -> ix+1), doubled on the i == 0 edge
        reps = 2 if i == 0 else 1
        for _ in range(reps):
            rows.append(ix[inner]); cols.append(ix[inner] + 1); vals.append(pd[inner])
        # east (ix -> ix+jmax), doubled on the j == 0 edge
        rows.append(ix[inner]); cols.append(ix[inner] + jmax); vals.append(pd[inner])
        if jmax > 1:
            rows.append(ix[:1]); cols.append(ix[:1] + jmax); vals.append(pd[:1])
        # south (ix -> ix-1) for j > 0
        rows.append(ix[1:]); cols.append(ix[1:] - 1); vals.append(pu[1:])
        # west (ix -> ix-jmax-1) for i > 0
        if i > 0:
            rows.append(ix); cols.append(ix - jmax - 1); vals.append(pu)
        base += jmax
    r = np.concatenate(rows)
    c = np.concatenate(cols)
    v = np.concatenate(vals)
    return sp.coo_matrix((v, (r, c)), shape=(n, n)).tocsr()


def laplace_1d(n, dtype=None):
    """src/arnoldi/matrices.py:87-95 -- tridiagonal (1, -2, 1)."""
    off = np.ones(n - 1, dtype=dtype)
    return sp.diags_array([-2 * np.ones(n, dtype=dtype), off, off], offsets=[0, -1, 1])


def laplace_1d_eigen(n):
    """src/arnoldi/matrices.py:76-84."""
    return -2 + 2 * np.cos(np.arange(1, n + 1) * np.pi / (n + 1))


# --------------------------------------------------------------------------
# operator apply
# --------------------------------------------------------------------------
def csr_matvec(A, x):
    """src/arnoldi/decomposition.py:58 -- ``A @ x`` (SciPy sparsetools for CSR)."""
    return A @ x


# --------------------------------------------------------------------------
# orthogonalisation
# --------------------------------------------------------------------------
def dgks_gs(w, V, h, tol=1e-8, eta=ETA_DGKS):
    """src/arnoldi/ortho.py:56-107 -- classical Gram-Schmidt + DGKS second pass.

    In place on ``w`` (n,) and ``h`` (J,).  Returns ``(beta, breakdown)``.
    Also returns, as a third item, whether the second pass ran (extra to the
    reference; used by the stage-level parity tests).
    """
    norm_in = _blas.dznrm2(w)                        # ortho.py:92
    c = _blas.zgemv(1.0, V, w, trans=2)              # ortho.py:94  V^H w
    h[:] = c                                         # ortho.py:95
    w -= _blas.zgemv(1.0, V, c)                      # ortho.py:96
    beta = _blas.dznrm2(w)                           # ortho.py:98
    again = bool(beta < norm_in * eta)               # ortho.py:101
    if again:
        c = _blas.zgemv(1.0, V, w, trans=2)          # ortho.py:102
        h += c                                       # ortho.py:103
        w -= _blas.zgemv(1.0, V, c)                  # ortho.py:104
        beta = _blas.dznrm2(w)                       # ortho.py:105
    return beta, bool(beta < tol), again             # ortho.py:107


# --------------------------------------------------------------------------
# Arnoldi expansion
# --------------------------------------------------------------------------
def arnoldi_expand(A, V, H, tol=None, *, start_dim=0, max_dim=None):
    """src/arnoldi/decomposition.py:13-68 -- in place on V (n, m+1) and H (m+1, m).

    Returns ``(V_view, H_view, n_iter)`` exactly as the reference does,
    including the early return on breakdown (column not normalised,
    ``H[j+1, j]`` not written).
    """
    if tol is None:
        tol = np.sqrt(np.finfo(A.dtype).eps)          # decomposition.py:41-42
    n = A.shape[0]
    m = V.shape[1] - 1
    assert A.shape[1] == n
    assert V.shape == (n, m + 1)
    assert H.shape == (m + 1, m)
    if max_dim is None:
        max_dim = m
    assert max_dim <= m

    for j in range(start_dim, max_dim):
        w = V[:, j + 1]
        w[:] = csr_matvec(A, V[:, j])                 # decomposition.py:58
        beta, broke, _ = dgks_gs(w, V[:, : j + 1], H[: j + 1, j], tol)
        if broke:                                     # decomposition.py:61-63
            return V[:, : j + 2], H[: j + 2, : j + 1], j + 1
        H[j + 1, j] = beta                            # decomposition.py:65
        w /= beta                                     # decomposition.py:66
    return V[:, : max_dim + 1], H[: max_dim + 1, :max_dim], max_dim


# --------------------------------------------------------------------------
# host dense step
# --------------------------------------------------------------------------
_TREXC = {
    np.dtype("float32"): _lapack.strexc,
    np.dtype("float64"): _lapack.dtrexc,
    np.dtype("complex64"): _lapack.ctrexc,
    np.dtype("complex128"): _lapack.ztrexc,
}


def ordered_schur(a, output="real", *, sort_function=None):
    """src/arnoldi/utils.py:32-67 -- Schur form with the diagonal put in the
    order ``sort_function(diag(T))`` by one ``?trexc`` move per misplaced
    eigenvalue (LAPACK indices are 1-based)."""
    if output != "complex":
        # utils.py:64-65: the real path raises after computing the Schur form
        sla.schur(a, output=output)
        raise ValueError("output!='complex' not implemented yet")
    mover = _TREXC[np.result_type(a.dtype, 1j)]
    if sort_function is None:
        sort_function = arg_largest_magnitude

    T, Z = sla.schur(a, output="complex")             # utils.py:45
    wanted = sort_function(np.diag(T))                # utils.py:49-50
    where = list(range(T.shape[0]))                   # where[pos] = original index
    for dst, orig in enumerate(wanted):
        src = where.index(orig)
        if src != dst:
            T, Z, _info = mover(T, Z, src + 1, dst + 1)   # utils.py:59
            where.insert(dst, where.pop(src))
    return T, Z


# --------------------------------------------------------------------------
# Krylov-Schur driver
# --------------------------------------------------------------------------
def krylov_schur(A, nev, *, max_dim=None, stopping_criterion=None, max_restarts=100,
                 sort_function=None, p=None, v0=None, trace=None):
    """src/arnoldi/krylov_schur.py:10-114.

    ``v0`` (optional, extra to the reference) replaces the random start vector;
    ``trace`` (optional dict) receives per-restart scalars for the golden tests.
    Raises the reference's exceptions with the reference's messages.
    """
    tol = np.sqrt(np.finfo(A.dtype).eps) if stopping_criterion is None else stopping_criterion
    if sort_function is None:
        sort_function = arg_largest_magnitude
    assert max_restarts > 0
    n = A.shape[0]
    assert A.shape[1] == n
    if max_dim is None:
        max_dim = min(max(2 * nev + 1, 20), n)        # krylov_schur.py:29-30
    if p is None:
        p = min(nev + 5, max_dim - 1)                 # krylov_schur.py:33-34
    assert nev <= p < max_dim

    V = np.zeros((n, max_dim + 1), dtype=C128, order="F")   # krylov_schur.py:42
    H = np.zeros((max_dim + 1, max_dim), dtype=C128)        # krylov_schur.py:43
    V[:, 0] = random_unit_vector(n, C128) if v0 is None else v0

    hist = History.from_k(nev)
    done = False
    _, _, m = arnoldi_expand(A, V, H, tol, start_dim=0, max_dim=max_dim)
    n_restarts = 0
    for restart in range(max_restarts):
        if trace is not None:
            trace.setdefault("t_restart", []).append(time.perf_counter())
        if m != max_dim:                               # krylov_schur.py:57-59
            raise ValueError("Happy breakdown not supported yet")
        matvecs = restart * (max_dim - nev) + (m - nev)   # krylov_schur.py:63

        T1, Q1 = sla.schur(H[:m, :m], output="complex")   # krylov_schur.py:69
        T2, Q2 = ordered_schur(T1, output="complex", sort_function=sort_function)
        Q = Q1 @ Q2                                       # krylov_schur.py:72
        Qp = Q[:, :p]
        coupling = H[m, :m].copy()                        # krylov_schur.py:86
        last = H[m, m - 1]

        V[:, :p] = V[:, :m] @ Qp                          # krylov_schur.py:78
        V[:, p] = V[:, m]                                 # krylov_schur.py:81
        H[:p, :p] = T2[:p, :p]                            # krylov_schur.py:83
        H[p, :p] = coupling @ Qp                          # krylov_schur.py:87
        H[p, p:] = 0                                      # krylov_schur.py:88

        ratio = np.abs(last * Q[m - 1, :]) / np.abs(np.diag(T2))  # :91-92
        for k in range(nev):                              # krylov_schur.py:94-97
            if ratio[k] <= tol:
                hist.matvecs[k] = matvecs
                hist.restarts[k] = restart + 1
        n_restarts = restart + 1
        if trace is not None:
            trace.setdefault("ratio", []).append(ratio[:nev].copy())
        done = bool(np.all(ratio[:nev] < tol))            # krylov_schur.py:99
        if done:
            break
        _, _, m = arnoldi_expand(A, V, H, tol, start_dim=p, max_dim=max_dim)

    if trace is not None:
        trace["restarts"] = n_restarts
    if not done:
        raise ValueError("Has not converged !")           # krylov_schur.py:108-109
    return V[:, :nev], H[:nev, :nev], hist


def eig_residuals(A, Q, T):
    """README.md:47-48 and scripts/benchmark-partial-schur.py:42-43,97-98:
    eigenpairs from the partial Schur form and ``||A v - lambda v|| / |lambda|``."""
    vals, S = np.linalg.eig(T)
    vecs = Q @ S
    res = np.linalg.norm(A @ vecs - vecs * vals, axis=0)
    return vals, vecs, res / np.abs(vals)


# --------------------------------------------------------------------------
# explicit restarts: Ritz extraction, MGS, the two solvers
# --------------------------------------------------------------------------
@dataclasses.dataclass
class Ritz:
    """src/arnoldi/decomposition.py:71-79 -- ``RitzDecomposition``'s three fields."""

    values: np.ndarray
    vectors: np.ndarray
    approximate_residuals: np.ndarray

    def compute_true_residuals(self, A):
        """decomposition.py:134-146 -- column norms of ``A U - U diag(values)``."""
        return np.linalg.norm(A @ self.vectors - self.values * self.vectors, axis=0)


def ritz_from_v_and_h(V, H, n_ritz, *, max_dim=None, sort_function=None):
    """src/arnoldi/decomposition.py:81-132 -- ``RitzDecomposition.from_v_and_h``:
    eigenpairs of ``H[:m, :m]`` (LAPACK zgeev through numpy), the first ``n_ritz`` in the order
    of ``sort_function``, Ritz vectors ``V[:, :m] S`` and ``|H[m, m-1] * S[m-1, :]|``."""
    max_dim = max_dim or V.shape[1] - 1               # decomposition.py:108
    assert H.shape[0] > max_dim                       # decomposition.py:110-113
    assert H.shape[1] >= max_dim
    assert V.shape[1] > max_dim
    assert n_ritz <= max_dim
    if sort_function is None:
        sort_function = arg_largest_magnitude
    w, S = np.linalg.eig(H[:max_dim, :max_dim])       # decomposition.py:121
    pick = sort_function(w)[:n_ritz]                  # decomposition.py:122
    S = S[:, pick]
    return Ritz(w[pick], V[:, :max_dim] @ S, np.abs(H[max_dim, max_dim - 1] * S[-1]))


def mgs(basis, w, tol):
    """src/arnoldi/explicit_restarts.py:64-78 -- modified Gram-Schmidt of ``w`` against the columns
    of ``basis`` one at a time, then normalisation; in place; asserts on a norm <= tol."""
    for j in range(basis.shape[1]):
        w -= np.vdot(basis[:, j], w) * basis[:, j]
    beta = np.linalg.norm(w)
    assert beta > tol, "MGS: Too small norm when orthornormalizing"
    w /= beta
    return w


def naive_explicit_restarts(A, m=None, *, stopping_criterion=None, max_restarts=10):
    """src/arnoldi/explicit_restarts.py:31-61 -- one eigenpair: m Arnoldi steps, restart from the
    dominant Ritz vector.  Returns ``(ritz, converged, restarts_used)``."""
    tol = np.sqrt(np.finfo(A.dtype).eps) if stopping_criterion is None else stopping_criterion
    dtype = np.promote_types(A.dtype, np.complex64)   # explicit_restarts.py:37
    n = A.shape[0]
    k = 1
    if m is None:
        m = min(max(2 * k + 1, 20), n)
    V = np.zeros((n, m + 1), dtype)
    H = np.zeros((m + 1, m), dtype)
    v0 = random_unit_vector(n).astype(dtype)          # explicit_restarts.py:48
    ritz = None
    for i in range(max_restarts):
        V[:, 0] = v0
        Va, Ha, _ = arnoldi_expand(A, V, H)           # default invariant tolerance
        ritz = ritz_from_v_and_h(Va, Ha, k)
        if ritz.approximate_residuals[0] < tol:       # explicit_restarts.py:53-56
            res = ritz.compute_true_residuals(A)
            if res[0] / max(np.abs(ritz.values[0]), tol) < tol:
                return ritz, True, i
        v0 = ritz.vectors[:, 0]                       # explicit_restarts.py:58-59
    return ritz, False, max_restarts


def explicit_restarts_with_deflation(A, nev, *, max_dim=None, stopping_criterion=None,
                                     max_restarts=100, sort_function=None):
    """src/arnoldi/explicit_restarts.py:81-168 -- eigenpairs one at a time; converged Schur vectors
    stay locked in ``V[:, :k]`` (the expansion from ``start_dim = k`` orthogonalises against them)
    and ``H[:k+1, k]`` is rebuilt from them.  Returns ``(eigenvalues, eigenvectors, History)``."""
    tol = np.sqrt(np.finfo(A.dtype).eps) if stopping_criterion is None else stopping_criterion
    if sort_function is None:
        sort_function = arg_largest_magnitude
    assert max_restarts > 0
    n = A.shape[0]
    assert A.shape[1] == n
    if max_dim is None:
        max_dim = min(max(2 * nev + 1, 20), n)
    V = np.zeros((n, max_dim + 1), dtype=C128)
    H = np.zeros((max_dim + 1, max_dim), dtype=C128)
    hist = History.from_k(nev)
    for k in range(nev):
        v0 = random_unit_vector(n, C128)              # explicit_restarts.py:111-113
        mgs(V[:, :k], v0, tol)
        V[:, k] = v0
        for restart in range(max_restarts):
            Va, Ha, m = arnoldi_expand(A, V, H, tol, start_dim=k)
            assert m > k
            lucky = m != max_dim                      # explicit_restarts.py:123-126
            booked = restart * (max_dim - k) + (m - k)
            ritz = ritz_from_v_and_h(Va[:, k:], Ha[k:, k:], m - k, sort_function=sort_function)
            V[:, k] = ritz.vectors[:, 0]              # explicit_restarts.py:140-141
            mgs(V[:, :k], V[:, k], tol)
            if lucky or ritz.approximate_residuals[0] / np.abs(ritz.values[0]) < tol:
                Av = A @ V[:, k]                      # explicit_restarts.py:149-151
                for i in range(k + 1):
                    H[i, k] = np.vdot(V[:, i], Av)
                H[k + 1:-1, k] = 0
                hist.matvecs[k] = booked
                hist.restarts[k] = restart + 1
                break
        else:
            raise ValueError(f"Could not converge for value {k}")
    vals, Y = np.linalg.eig(H[:nev, :nev])            # explicit_restarts.py:166-167
    return vals, V[:, :nev] @ Y, hist
