#!/usr/bin/env python3
"""Wall time of explicit_restarts_with_deflation on the C5-shaped planted matrix (device) and of the
CPU oracle on a smaller sample of the same generator.  Run on the GPU box:

    python profiles/explicit_restart_timing.py [n] [cpu_n]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "arnoldi-py_amd")]

import torch  # noqa: E402

import oracle  # noqa: E402  (reported CPU baseline only)
from arnoldi_amd import matrices  # noqa: E402
from arnoldi_amd.engine import CsrOperator  # noqa: E402
from arnoldi_amd.explicit_restarts import explicit_restarts_with_deflation  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
cpu_n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
planted = (4.0, 3.7, 3.4, 3.1, 2.8, 2.5)
kw = dict(max_dim=20, stopping_criterion=1e-8, max_restarts=200)

A = matrices.random_csr(n, 5, seed=1234, planted=planted)
op = CsrOperator(A)
out = {"n": n, "nev": 3, **{k: v for k, v in kw.items()}}
for rep in range(2):                                   # second pass: warm
    np.random.seed(0)
    st = {}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    vals, vecs, hist = explicit_restarts_with_deflation(op, 3, stats=st, gather=False, **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
res = st["ctx"].residual_norms(st["eigenvectors_device"], vals)
out.update(gpu_s=dt, restarts=hist.restarts.tolist(), matvecs=hist.matvecs.tolist(), eigenvalues=vals.real.tolist(),
           max_rel_residual=float((res / np.abs(vals)).max()), applies=int(st["matvecs"]),
           gpu_ms_per_apply=1e3 * dt / st["matvecs"])

Ac = matrices.random_csr(cpu_n, 5, seed=1234, planted=planted)
np.random.seed(0)
t0 = time.perf_counter()
vo, xo, ho = oracle.explicit_restarts_with_deflation(Ac, 3, **kw)
dtc = time.perf_counter() - t0
out.update(cpu_n=cpu_n, cpu_s=dtc, cpu_restarts=ho.restarts.tolist(), cpu_matvecs=int(ho.matvecs.sum()),
           cpu_ms_per_matvec_scaled_to_n=1e3 * dtc / int(ho.matvecs.sum()) * n / cpu_n)
print(json.dumps(out))
