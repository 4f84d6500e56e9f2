#!/usr/bin/env python3
"""CSR-stream SpMV: sweep lanes-per-row for matrices of different mean row length (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "arnoldi-py_amd"))
import numpy as np, torch
from arnoldi_amd import matrices
from arnoldi_amd.device import DeviceCSR, choose_lanes_per_row
for name, A in (("laplace2d 2M (5/row)", matrices.laplace2d(1414, 1415)), ("laplace3d 4M (7/row)", matrices.laplace3d(158, 159, 160)),
                ("banded 1.5M (35/row)", matrices.banded_csr(1_508_065, 35)), ("banded 1M (15/row)", matrices.banded_csr(1_000_000, 15)),
                ("banded 0.5M (101/row)", matrices.banded_csr(500_000, 101))):
    n = A.shape[0]
    x = torch.randn(n, dtype=torch.complex128, device="cuda"); y = torch.empty(n, dtype=torch.complex128, device="cuda")
    out = []
    for lpr in (1, 2, 4, 8, 16, 32, 64):
        d = DeviceCSR(A, lanes_per_row=lpr)
        for _ in range(3): d.spmv(x, y)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): d.spmv(x, y)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        out.append(f"lpr={lpr}: {d.algorithmic_bytes() / ms / 1e6:.0f}")
    print(f"{name:24s} heuristic lpr={choose_lanes_per_row(n, A.nnz):2d}  GB/s -> " + "  ".join(out), flush=True)
