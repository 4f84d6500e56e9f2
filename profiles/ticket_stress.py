#!/usr/bin/env python3
"""Stress of the in-kernel norm sum behind the second DGKS pass (the last-arriver ticket of k_update<true>, the one place
the shipped schedule uses it): ``aks_dgks_gs(normalize=0)`` on vectors that NEED the second pass, thousands of times, each
result (beta = sqrt(red3), booked by the kernel's last workgroup) checked against ||w|| recomputed from the vector.  A
stale partial row shows as a relative error of ~1/512.

    AKS_LIB_PATH=... python profiles/ticket_stress.py [n] [iterations]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "arnoldi-py_amd"))
import numpy as np, torch
from arnoldi_amd import _hip, device as dev

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
    m = 20
    torch.manual_seed(0)
    basis, ws = dev.KrylovBasis(n, m), dev.Workspace(n, m)
    # orthonormal-ish panel: random columns are orthogonal to ~1e-3 at this n, good enough for "w in span(V) + noise"
    basis.V[:, :n].copy_(torch.randn((m + 1, n), dtype=torch.complex128, device="cuda") / np.sqrt(2 * n))
    H = torch.zeros((m + 1, m), dtype=torch.complex128, device="cuda")
    worst, bad, second = 0.0, 0, 0
    for it in range(iters):
        J = 8 + it % 12
        coef = torch.randn(J, dtype=torch.complex128, device="cuda")
        w = basis.col(J)
        w.copy_((coef.unsqueeze(0) @ basis.V[:J]).squeeze(0) + 1e-3 * basis.V[m] )      # mostly inside span(V[:, :J])
        ws.buf[:8].zero_()
        before = ws.read_ctrl().second_passes if it % 97 == 0 else None
        rc = _hip.load().aks_dgks_gs(n, J, dev._ptr(basis.V), basis.ldv, dev._ptr(w), C_void(H.data_ptr() + 16 * (J - 1)), m, 0.0,
                                     dev.ETA_DGKS, 0, dev._ptr(ws.buf), ws.nbytes, m, dev._stream())
        _hip.check(rc, "aks_dgks_gs")
        ctrl = ws.read_ctrl()
        true = float(torch.linalg.norm(w[:n]))
        err = abs(ctrl.beta - true) / true
        worst = max(worst, err)
        bad += err > 1e-9
        if before is not None:
            second += ctrl.second_passes - before
    print(f"lib {os.path.basename(os.path.dirname(_hip.LIB_PATH))}: n={n}, {iters} steps (second passes in the sampled steps: {second}), "
          f"worst |beta - ||w||| / ||w|| = {worst:.3e}, steps off by more than 1e-9: {bad}")
    return 1 if bad else 0

import ctypes
C_void = ctypes.c_void_p
if __name__ == "__main__":
    sys.exit(main())
