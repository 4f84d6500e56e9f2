#!/usr/bin/env python3
"""Is the sliced SpMV slower inside a solve than alone?  Same process, same operator (GPU box):
  (a) 20 launches back to back, HIP events around the whole batch;
  (b) the same with one event pair PER launch (what the probe of aks_arnoldi_expand does);
  (c) inside restarts, from the probe.
    python profiles/sell_context_probe.py [workload] [rows] [per_row] [nev] [max_dim]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "arnoldi-py_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from arnoldi_amd import _hip
from arnoldi_amd.dist import row_offsets
from arnoldi_amd.engine import CsrOperator
from arnoldi_amd.krylov_schur import KrylovSchurSolver
from arnoldi_amd.utils import arg_largest_magnitude

wl = sys.argv[1] if len(sys.argv) > 1 else "banded"
rows, per_row, nev, m = (int(a) for a in (sys.argv[2:6] + ["1500000", "35", "20", "41"][len(sys.argv[2:6]):]))
args = bench.parse_args(["--workload", wl, "--rows", str(rows), "--per-row", str(per_row), "--nev", str(nev), "--max-dim", str(m)])
n, dims = bench.problem_size(args)
off = row_offsets(n, 1)
op = CsrOperator(local_rows=bench.build_rows(args, 0, n, n, dims), offsets=off, comm=None)
print("form", op.spmv_form, "n", n, "algorithmic MB", op.algorithmic_bytes() / 1e6)
x = torch.randn(n, dtype=torch.complex128, device="cuda"); y = torch.empty_like(x)
d = op.diag
for _ in range(5): d.spmv(x, y)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): d.spmv(x, y)
e1.record(); torch.cuda.synchronize()
a = e0.elapsed_time(e1) / 20
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
for s, t in ev:
    s.record(); d.spmv(x, y); t.record()
torch.cuda.synchronize()
b = sum(s.elapsed_time(t) for s, t in ev) / 20
p = min(nev + 5, m - 1)
np.random.seed(0)
solver = KrylovSchurSolver(op, nev, m, p, 1e-8, arg_largest_magnitude, comm=None)
solver.start()
for i in range(2): solver.contract(i); solver.expand()
probe = _hip.Probe(capacity=2 * m * 5 + 8); solver.ctx.probe = probe
for i in range(5): solver.contract(2 + i); solver.expand()
torch.cuda.synchronize()
k, ms = probe.read(_hip.PROBE_SPMV)
alg = op.algorithmic_bytes()
for name, t in (("back to back", a), ("event pair per launch", b), ("inside restarts (probe)", ms / k)):
    print(f"{name:28s} {t:.4f} ms  {alg / t / 1e9:.3f} TB/s = {alg / t / 8e9:.3f} of 8 TB/s")
