#!/usr/bin/env python3
"""Times, per panel width J, the three Gram-Schmidt passes of one Arnoldi step at n rows (GPU box):
projection, fused update + re-projection, and the plain update + norm (the predicated second-pass
kernel with its predicate forced true).  Shows what a "light" first pass without the speculative
re-projection could gain.   python profiles/update_variants.py [n]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "arnoldi-py_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from arnoldi_amd import device as dev  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
m = 20
basis = dev.KrylovBasis(n, m)
ws = dev.Workspace(n, m)
basis.V.copy_(torch.randn(basis.V.shape, dtype=torch.complex128, device="cuda") * (1.0 / np.sqrt(n)))


def timeit(fn, reps=10):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


print(f"n={n}: ms (TB/s algorithmic)")
tot = [0.0, 0.0, 0.0]
for J in range(11, m + 1):
    w = basis.col(J)
    tp = timeit(lambda: dev.gs_project(basis, J, w, ws))
    tu = timeit(lambda: dev.gs_update_project(basis, J, w, ws))

    def forced():
        ws.red(1, J + 1)[2 * J] = 1e300       # ||w||^2 before: makes the DGKS test fire
        ws.red(2, J + 1)[2 * J] = 1.0
        dev.gs_update_norm(basis, J, w, ws)
    tl = timeit(forced)
    bp, bu = 16 * n * (J + 1), 16 * n * (J + 2)
    tot = [tot[0] + tp, tot[1] + tu, tot[2] + tl]
    print(f"J={J:2d}  project {tp:.3f} ({bp / tp / 1e9:.2f})   update+reproject {tu:.3f} ({bu / tu / 1e9:.2f})   "
          f"update+norm {tl:.3f} ({bu / tl / 1e9:.2f})")
print(f"sum over J=11..20: project {tot[0]:.3f}  update+reproject {tot[1]:.3f}  update+norm {tot[2]:.3f} ms")
