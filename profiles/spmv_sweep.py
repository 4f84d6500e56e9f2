#!/usr/bin/env python3
"""SpMV micro-benchmark (GPU box): both SpMV forms on BASELINE-shaped matrices of growing n.

Shows where the x gathers of the CSR-stream kernel are served from (n <= 256K: one XCD's 4 MiB
L2; beyond: fabric requests, one per gather) and what the slab-binned two-phase form buys.

    python profiles/spmv_sweep.py [random|laplace2d|laplace3d] [reps] [min_log2n] [max_log2n]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "arnoldi-py_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from arnoldi_amd import matrices  # noqa: E402
from arnoldi_amd.device import DeviceCSR  # noqa: E402


def time_form(dA, x, y, reps):
    for _ in range(3):
        dA.spmv(x, y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        dA.spmv(x, y)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "random"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    lo = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    hi = int(sys.argv[4]) if len(sys.argv) > 4 else 24
    sizes = [1 << k for k in range(lo, hi + 1)]
    if lo <= 23 <= hi:
        sizes.append(10_000_000)
    print(f"{'n':>10} {'nnz':>11} {'csr ms':>8} {'GB/s':>7} {'binned ms':>9} {'GB/s':>7}  (algorithmic bytes)")
    for n in sorted(sizes):
        if kind == "random":
            A = matrices.random_csr(n, 5, 1234)
        elif kind == "laplace2d":
            nx = int(round(n ** 0.5))
            A = matrices.laplace2d(nx, nx + 1)
        else:
            nx = int(round(n ** (1 / 3)))
            A = matrices.laplace3d(nx, nx + 1, nx + 2)
        dA = DeviceCSR(A)
        dA.build_binned()
        x = torch.randn(A.shape[1], dtype=torch.complex128, device="cuda")
        y = torch.empty(A.shape[0], dtype=torch.complex128, device="cuda")
        dA.use_binned = False
        t_csr = time_form(dA, x, y, reps)
        ref = y.clone()
        dA.use_binned = True
        t_pb = time_form(dA, x, y, reps)
        err = float(torch.linalg.norm(y - ref) / torch.linalg.norm(ref))
        b = dA.algorithmic_bytes()
        print(f"{A.shape[0]:>10} {dA.nnz:>11} {t_csr:>8.4f} {b / t_csr / 1e6:>7.0f} {t_pb:>9.4f} "
              f"{b / t_pb / 1e6:>7.0f}  rel.diff {err:.1e}", flush=True)
        del dA, x, y, ref
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
