#!/usr/bin/env python3
"""A/B of hipGraph replay for the re-expansion (GPU box): restarts/s with AKS_GRAPH=0 / 1.
    python profiles/graph_ab.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, time
sys.path.insert(0, os.path.join(%r, "arnoldi-py_amd"))
import numpy as np, torch
from arnoldi_amd import matrices
from arnoldi_amd.krylov_schur import KrylovSchurSolver
from arnoldi_amd.utils import arg_largest_magnitude
for name, A, k, m in (("laplace2d 1000x1001 k=10 m=40", matrices.laplace2d(1000, 1001), 10, 40),
                      ("random n=1.25M k=5 m=20", matrices.random_csr(1_250_000, 5, 1234), 5, 20),
                      ("random n=10M k=5 m=20", matrices.random_csr(10_000_000, 5, 1234), 5, 20)):
    np.random.seed(0)
    s = KrylovSchurSolver(A, k, m, min(k + 5, m - 1), 1e-8, arg_largest_magnitude)
    s.start()
    for i in range(3):
        s.contract(i); s.expand()
    torch.cuda.synchronize(); t0 = time.perf_counter(); R = 30
    for i in range(R):
        s.contract(3 + i); s.expand()
    torch.cuda.synchronize()
    print(f"AKS_GRAPH={os.environ.get('AKS_GRAPH')}  {name:32s} {(time.perf_counter()-t0)/R*1e3:8.3f} ms/restart", flush=True)
''' % ROOT
for g in ("0", "1"):
    subprocess.run([sys.executable, "-c", code], env=dict(os.environ, AKS_GRAPH=g), check=True)
