#!/usr/bin/env python3
"""A/B the slab-binned SpMV of several library builds (GPU box), interleaved rounds.
    python profiles/ab_spmv.py n rounds lib1.so lib2.so ..."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r'''
import os, sys
sys.path.insert(0, os.path.join(%r, "arnoldi-py_amd"))
import torch
from arnoldi_amd import matrices
from arnoldi_amd.device import DeviceCSR
n = int(sys.argv[1])
d = DeviceCSR(matrices.random_csr(n, 5, 1234)); d.autotune(force="binned")
x = torch.randn(n, dtype=torch.complex128, device="cuda"); y = torch.empty(n, dtype=torch.complex128, device="cuda")
for _ in range(3): d.spmv(x, y)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): d.spmv(x, y)
e1.record(); torch.cuda.synchronize()
print(e0.elapsed_time(e1) / 20)
''' % ROOT
n, rounds, libs = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
res = {l: [] for l in libs}
for _ in range(rounds):
    for l in libs:
        r = subprocess.run([sys.executable, "-c", WORKER, n], capture_output=True, text=True,
                           env=dict(os.environ, AKS_LIB_PATH=os.path.abspath(l)))
        if r.returncode: print(r.stderr[-1500:]); sys.exit(1)
        res[l].append(float(r.stdout.strip().splitlines()[-1]))
for l in libs:
    v = sorted(res[l]); print(f"{os.path.basename(l):20s} median {v[len(v)//2]:.4f} ms  min {v[0]:.4f}")
