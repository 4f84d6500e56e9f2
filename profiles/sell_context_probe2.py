#!/usr/bin/env python3
"""What in a solve's context slows the sliced SpMV down (banded n = 1.5M: 0.115 ms alone, 0.130-0.140 ms inside restarts)?
Event pairs around the SpMV only; in between, one of: nothing / a 1-GB streaming read (what Gram-Schmidt leaves in the
caches) / a rewrite of x (what k_finish does) / both."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "arnoldi-py_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from arnoldi_amd.dist import row_offsets
from arnoldi_amd.engine import CsrOperator

args = bench.parse_args(["--workload", "banded", "--rows", "1500000", "--per-row", "35", "--nev", "20", "--max-dim", "41"])
n, dims = bench.problem_size(args)
op = CsrOperator(local_rows=bench.build_rows(args, 0, n, n, dims), offsets=row_offsets(n, 1), comm=None)
d = op.diag
x = torch.randn(n, dtype=torch.complex128, device="cuda"); y = torch.empty_like(x)
big = torch.randn(64 * 1024 * 1024, dtype=torch.complex128, device="cuda")     # 1 GB
alg = op.algorithmic_bytes()

def run(between, reps=20):
    for _ in range(3):
        between(); d.spmv(x, y)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for s, t in ev:
        between(); s.record(); d.spmv(x, y); t.record()
    torch.cuda.synchronize()
    return sum(s.elapsed_time(t) for s, t in ev) / reps

cases = {
    "nothing in between": lambda: None,
    "1 GB streaming read in between": lambda: big.sum(),
    "x rewritten in between (x *= 1)": lambda: x.mul_(1.0),
    "both": lambda: (big.sum(), x.mul_(1.0)),
    "y rewritten in between": lambda: y.zero_(),
}
for name, f in cases.items():
    t = run(f)
    print(f"{name:36s} {t:.4f} ms  {alg / t / 8e9:.3f} of 8 TB/s", flush=True)
