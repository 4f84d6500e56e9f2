"""A whole PROCESS that solves the reference's README example once: interpreter start, imports, library load, device
initialisation, set-up, solve.  Run under both host backends:
    python profiles/cold_process_probe.py                         (the default: HIP runtime backend, no torch)
    AKS_HOST_ALLOC=torch python profiles/cold_process_probe.py    (torch interop backend)"""
import time

T0 = time.perf_counter()
import os  # noqa: E402
import sys  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "arnoldi-py_amd"))
import numpy as np  # noqa: E402

t_np = time.perf_counter()
import arnoldi_amd  # noqa: E402
from arnoldi_amd.matrices import mark  # noqa: E402
from arnoldi_amd.utils import arg_largest_real  # noqa: E402

t_imp = time.perf_counter()
A = mark(50)
np.random.seed(0)
Q, T, hist = arnoldi_amd.partial_schur(A, 5, max_dim=20, sort_function=arg_largest_real, stopping_criterion=1e-8)
t_first = time.perf_counter()
np.random.seed(0)
Q, T, hist = arnoldi_amd.partial_schur(A, 5, max_dim=20, sort_function=arg_largest_real, stopping_criterion=1e-8)
t_second = time.perf_counter()
print(f"backend {arnoldi_amd.mem.BACKEND:5s}: numpy import {1e3 * (t_np - T0):7.1f} ms, arnoldi_amd import {1e3 * (t_imp - t_np):7.1f} ms, "
      f"first solve {1e3 * (t_first - t_imp):7.1f} ms, second solve {1e3 * (t_second - t_first):6.1f} ms, torch loaded: {'torch' in sys.modules}")
