"""Krylov-Schur with locking and a dynamic restart size -- the reference's open TODO
("implement locking and dynamic p", /root/reference/README.md:116), SURVEY 8(f) rank 4.

Opt-in (``partial_schur(..., locking=True)``): it is a different -- mathematically equivalent --
iteration, so restart counts differ from the reference's (the default path reproduces them exactly).

Locking (Stewart's Krylov-Schur, sect. 4).  At a restart the ACTIVE part of the Krylov decomposition is

    A [Q_l  U] = [Q_l  U  u] [ T_l   R  ]
                             [  0    S  ]        Q_l: l locked Schur vectors, T_l upper triangular,
                             [  0   b^H ]        S: (m - l) x (m - l), u = V[:, m]

Only ``S`` is rotated: ``S = Z T Z^H`` ordered by the caller's sort key.  The leading wanted Ritz values
whose coupling ``|b^H z_i|`` is below ``tol |theta_i|`` are locked: their coupling is set to zero
(a perturbation of A of that size), they join ``Q_l`` and never take part in a rotation again.  On the
device this means

  * the restart compression multiplies the ACTIVE columns only:
    ``V[:, l:l+pa] = V[:, l:m] Z[:, :pa]`` -- ``aks_truncate`` on the sub-basis that starts at column l:
    ``16 n (m + p - 2 l + 2)`` bytes instead of ``16 n (m + p + 2)``;
  * the Arnoldi expansion is unchanged: new vectors are orthogonalised against locked and active
    columns alike (their projections onto ``Q_l`` are the rows ``R`` of H).

Dynamic p (ARPACK's rule, dnaup2): with ``l`` values locked the restart keeps
``p = p0 + min(l, (m - p0) // 2)`` vectors, so the active window does not shrink as values converge.
"""
from __future__ import annotations

import numpy as np

from .krylov_schur import KrylovSchurSolver
from .utils import complex_schur, reorder_schur


class LockingKrylovSchurSolver(KrylovSchurSolver):
    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        self.p0 = self.p                 # the caller's restart size
        self.locked = 0
        self.trunc_bytes = []            # bytes the restart compression moved, per restart
        self.locked_history = []

    def contract(self, restart):
        H, m, nev, l = self.H, self.m, self.nev, self.locked
        tol = self.tol
        booked = restart * (self.max_dim - nev) + (m - nev)          # as krylov_schur.py:63

        S = H[l:m, l:m]
        T, Z = complex_schur(S, self.schur_memo)
        T, Z = reorder_schur(T, Z, self.sort_function(np.diag(T)))
        beta = H[m, m - 1]                                           # the residual row of H is beta e_m^T
        coupling = beta * Z[-1, :]
        estimate = np.abs(coupling) / np.abs(np.diag(T))

        # lock the leading wanted values that have converged (in order: value i only after values < i)
        newly = 0
        while l + newly < nev and newly < len(estimate) and estimate[newly] < tol:
            newly += 1
        l_new = l + newly
        p_new = min(self.p0 + min(l_new, (self.max_dim - self.p0) // 2), m - 1)
        p_new = max(p_new, l_new + 1) if l_new < nev else max(p_new, l_new)
        pa = p_new - l                                               # active columns kept (newly locked in front)

        Zp = Z[:, :pa]
        self.ctx.truncate_active(Zp, l, m, p_new)                    # V[:, l:p_new] = V[:, l:m] Zp ; V[:, p_new] = V[:, m]
        n_panel = self.ctx.basis.n_rows
        self.trunc_bytes.append(16 * n_panel * ((m - l) + pa + 2))

        R = H[:l, l:m].copy()
        H[:l, l:p_new] = R @ Zp                                      # projections of the active block on Q_l
        H[l:p_new, l:p_new] = T[:pa, :pa]
        H[p_new:, :p_new] = 0                                        # (the spike row moves with p)
        H[p_new, l:p_new] = coupling[:pa]
        H[p_new, l:l_new] = 0                                        # deflation: |coupling| < tol |theta|
        H[:, p_new:] = 0                                             # rebuilt column by column by the expansion

        under = np.zeros(nev, bool)
        under[:l_new] = True
        k_act = min(nev - l, len(estimate))
        under[l:l + k_act] |= estimate[:k_act] <= tol
        self.history.matvecs[under & (self.history.restarts == 0)] = booked
        self.history.restarts[under & (self.history.restarts == 0)] = restart + 1
        self.restarts_run = restart + 1
        est = np.zeros(nev)
        est[l:l + k_act] = estimate[:k_act]
        self.estimate = est
        self.locked = l_new
        self.p = p_new
        self.locked_history.append(l_new)
        return l_new >= nev

    def contract_invariant(self, restart):
        raise ValueError("Happy breakdown with locking=True is not supported; use locking=False")
