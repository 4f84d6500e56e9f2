"""Wall time of whole partial_schur calls on SMALL matrices (the reference's README example and a few sizes above it):
the drop-in against the CPU oracle on this host.  Fixed costs (launches, waits, host LAPACK) decide here, not bandwidth.
    python profiles/small_call_probe.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "arnoldi-py_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import oracle  # noqa: E402
from arnoldi_amd import matrices, partial_schur  # noqa: E402
from arnoldi_amd.utils import arg_largest_real  # noqa: E402

torch.zeros(1, device="cuda")
cases = [("mark(50) n=1275 LR", matrices.mark(50), dict(max_dim=20, stopping_criterion=1e-8, sort_function=arg_largest_real), dict(max_dim=20, stopping_criterion=1e-8, sort_function=oracle.arg_largest_real)),
         ("mark(200) n=20100 LR", matrices.mark(200), dict(max_dim=20, stopping_criterion=1e-8, sort_function=arg_largest_real), dict(max_dim=20, stopping_criterion=1e-8, sort_function=oracle.arg_largest_real)),
         ("random planted n=100k", matrices.random_csr(100_000, 5, 1234, planted=(4.0, 3.7, 3.4, 3.1, 2.8, 2.5)), dict(max_dim=20), dict(max_dim=20)),
         ("laplace2d 300x301", matrices.laplace2d(300, 301), dict(max_dim=40, max_restarts=2000), dict(max_dim=40, max_restarts=2000))]
for name, A, kw, kwo in cases:
    nev = 5
    times = []
    for rep in range(3):
        np.random.seed(0)
        st = {}
        t = time.perf_counter()
        partial_schur(A, nev, stats=st, **kw)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t)
    np.random.seed(0)
    t = time.perf_counter()
    oracle.krylov_schur(A, nev, **kwo)
    t_cpu = time.perf_counter() - t
    print(f"{name:24s} restarts {st['restarts']:4d}: drop-in {times[0] * 1e3:8.1f} / {min(times[1:]) * 1e3:8.1f} ms (first / later call), "
          f"{min(times[1:]) / st['restarts'] * 1e3:6.2f} ms per restart; CPU oracle {t_cpu * 1e3:8.1f} ms")
