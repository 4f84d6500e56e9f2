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
