"""``arnoldi_decomposition`` on caller-owned NumPy arrays -- the L1 seam of the reference
(src/arnoldi/decomposition.py:13-68) as a drop-in: same arguments, in-place semantics,
return triple and early return on breakdown, computed on the MI355X.

``partial_schur`` does not go through this wrapper (its basis stays resident in HBM,
``engine.ArnoldiContext``); this entry point uploads the known columns of ``V``, runs the
same device expansion, and copies the new columns and ``H`` entries back.
"""
from __future__ import annotations

import numpy as np

from .engine import ArnoldiContext, as_operator

C128 = np.complex128


def arnoldi_decomposition(A, V, H, invariant_tol=None, *, start_dim=0, max_dim=None):
    """Expand the Arnoldi relation ``A V_j = V_{j+1} H_j`` for j = start_dim .. max_dim-1.

    ``V`` is ``(n, m+1)`` complex128 (any memory order) whose columns ``0..start_dim`` are
    given; ``H`` is ``(m+1, m)`` complex128.  Both are updated in place.  Returns
    ``(V[:, :k+1], H[:k+1, :k], k)`` with ``k = max_dim`` or, on breakdown at step j
    (residual norm below ``invariant_tol``), ``k = j+1``.
    """
    if invariant_tol is None:
        invariant_tol = np.sqrt(np.finfo(A.dtype).eps)
    n = A.shape[0]
    m = V.shape[1] - 1
    assert A.shape[1] == n, "A is expected to be square matrix"
    assert V.shape == (n, m + 1), "V must have the same number of rows as A"
    assert H.shape == (m + 1, m), f"H must be {m+1, m}, is {H.shape}"
    assert V.dtype == C128 and H.dtype == C128, "work arrays must be complex128"
    if max_dim is None:
        max_dim = m
    assert max_dim <= m, "max_dim > m violated"

    ctx = ArnoldiContext(as_operator(A), m)
    ctx.basis.set_cols(0, V[:, : start_dim + 1])
    k = ctx.expand(H, start_dim, max_dim, float(invariant_tol))
    if k > start_dim:
        V[:, start_dim + 1: k + 1] = ctx.local_columns(start_dim + 1, k + 1)
    return V[:, : k + 1], H[: k + 1, :k], k


class RitzDecomposition:
    """Ritz values / vectors / residual estimates of an Arnoldi factorisation -- the reference's
    dataclass of the same name (src/arnoldi/decomposition.py:71-146) with the O(n) parts on the GPU:
    the Ritz vectors ``V_m @ S`` are formed by ``aks_combine`` (f64 MFMA) and stay in HBM; ``vectors``
    copies them to the host on first access; ``compute_true_residuals`` runs SpMV + norm kernels on the
    resident copy.
    """

    def __init__(self, values, vectors, approximate_residuals, *, block=None, ctx=None, source=None):
        self.values = values
        self.approximate_residuals = approximate_residuals
        self._vectors = vectors
        self._block, self._ctx, self._source = block, ctx, source

    def __repr__(self):
        return (f"RitzDecomposition(values={self.values!r}, "
                f"approximate_residuals={self.approximate_residuals!r})")

    @property
    def vectors(self):
        if self._vectors is None:
            self._vectors = (self._ctx.gather_block(self._block) if self._ctx is not None
                             else np.asfortranarray(self._block.get_cols()))
        return self._vectors

    @classmethod
    def from_v_and_h(cls, V, H, n_ritz, *, max_dim=None, sort_function=None):
        """Ritz pairs of ``A V[:, :m] = V[:, :m] H[:m, :m] + H[m, m-1] V[:, m] e_m^H``
        (decomposition.py:81-132).  ``V`` is a host array (n, >= m+1); the m x m eigenproblem is
        LAPACK on the host, ``V_m @ S`` runs on the device."""
        from . import device as dev
        from .utils import arg_largest_magnitude

        max_dim = max_dim or V.shape[1] - 1
        assert H.shape[0] > max_dim
        assert H.shape[1] >= max_dim
        assert V.shape[1] > max_dim
        assert n_ritz <= max_dim
        if sort_function is None:
            sort_function = arg_largest_magnitude
        eigvals, eigvecs = np.linalg.eig(H[:max_dim, :max_dim])
        ind = sort_function(eigvals)[:n_ritz]
        S = np.ascontiguousarray(eigvecs[:, ind], dtype=C128)
        approx = np.abs(H[max_dim, max_dim - 1] * S[-1])
        n = V.shape[0]
        Vd = dev.DeviceColumns(n, max_dim)
        Vd.set_cols(0, V[:, :max_dim])
        block = dev.combine_columns(Vd, 0, max_dim, S)
        return cls(eigvals[ind], None, approx, block=block)

    def compute_true_residuals(self, A):
        """``norm(A @ vectors - values * vectors, axis=0)`` (decomposition.py:134-146) on the device."""
        if self._block is None:                       # built from host vectors: upload them
            from . import device as dev

            vec = np.asarray(self._vectors, dtype=C128)
            self._block = dev.DeviceColumns(vec.shape[0], vec.shape[1])
            self._block.set_cols(0, vec)
        ctx = self._ctx if (self._ctx is not None and A is self._source) else None
        if ctx is None:
            op = as_operator(A)
            if op.n_local != self._block.n_rows:
                raise ValueError("A and the Ritz vectors have different sizes")
            ctx = ArnoldiContext(op, 1)
        return ctx.residual_norms(self._block, self.values)

