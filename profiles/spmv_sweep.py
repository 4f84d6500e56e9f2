#!/usr/bin/env python3
"""SpMV micro-benchmark (GPU box): time aks_csr_spmv on BASELINE-shaped matrices of growing n.

Shows where the x gathers are served from: n <= 256K -> one XCD's 4 MiB L2; n <= ~10M -> the
256 MiB Infinity Cache (if the streamed CSR arrays do not evict it); beyond -> HBM.

    python profiles/spmv_sweep.py [random|laplace2d|laplace3d] [reps]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "arnoldi-py_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from arnoldi_amd import matrices  # noqa: E402
from arnoldi_amd.device import DeviceCSR  # noqa: E402


def time_spmv(A, reps, lanes=0):
    n = A.shape[0]
    dA = DeviceCSR(A, lanes_per_row=lanes)
    x = torch.randn(n, dtype=torch.complex128, device="cuda")
    y = torch.empty(n, dtype=torch.complex128, device="cuda")
    for _ in range(3):
        dA.spmv(x, y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        dA.spmv(x, y)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return ms, dA.algorithmic_bytes(), dA.nnz


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "random"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    sizes = [1 << k for k in range(16, 25)] + [10_000_000]
    print(f"{'n':>10} {'nnz':>11} {'ms':>8} {'alg GB/s':>9} {'ns/nnz':>7}")
    for n in sorted(sizes):
        if kind == "random":
            A = matrices.random_csr(n, 5, 1234)
        elif kind == "laplace2d":
            nx = int(round(n ** 0.5))
            A = matrices.laplace2d(nx, nx + 1)
        else:
            nx = int(round(n ** (1 / 3)))
            A = matrices.laplace3d(nx, nx + 1, nx + 2)
        ms, nbytes, nnz = time_spmv(A, reps)
        print(f"{A.shape[0]:>10} {nnz:>11} {ms:>8.4f} {nbytes / ms / 1e6:>9.1f} {ms * 1e6 / nnz:>7.3f}", flush=True)


if __name__ == "__main__":
    main()
