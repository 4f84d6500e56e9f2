"""``dgks_gs`` on caller-owned NumPy arrays (drop-in for src/arnoldi/ortho.py:56-107).

Classical Gram-Schmidt of ``w`` against the orthonormal columns of ``V`` with the
DGKS-triggered second pass, run by the device stage kernels (projection, fused
update + re-projection, conditional second update, finish).
"""
from __future__ import annotations

import numpy as np

from . import device as dev, mem

C128 = np.complex128
M_SQRT1_2 = dev.ETA_DGKS


def dgks_gs(w, V, h, tol=1e-8, eta=M_SQRT1_2, *, info=None):
    """Orthogonalise ``w`` (n,) against ``V`` (n, J) in place; coefficients go to ``h`` (J,).

    Returns ``(beta, breakdown)``: the norm of the orthogonalised ``w`` and whether it is
    below ``tol``.  ``info`` (optional dict) receives ``second_pass``.
    """
    n, J = V.shape
    assert w.shape == (n,) and h.shape[0] >= J
    basis = dev.KrylovBasis(n, J)  # columns 0..J-1 = V, column J = w
    ws = dev.Workspace(n, J)
    basis.set_cols(0, V)
    basis.set_col(J, w)
    hdev = mem.zeros(J + 1, mem.c128, basis.device)
    dev.dgks_gs_device(basis, J, basis.col(J), hdev.data_ptr(), 1, float(tol), ws, float(eta),
                       normalize=False)
    ctrl = ws.read_ctrl()
    w[:] = basis.get_cols(J, J + 1)[:, 0]
    h[:J] = hdev[:J].cpu().numpy()
    if info is not None:
        info["second_pass"] = bool(ctrl.second_passes)
    return float(ctrl.beta), bool(ctrl.broken)
