import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "arnoldi-py_amd"))
import numpy as np, torch
from arnoldi_amd import matrices
from arnoldi_amd.engine import CsrOperator
from arnoldi_amd.krylov_schur_real import RealKrylovSchurSolver
from arnoldi_amd.krylov_schur import KrylovSchurSolver
from arnoldi_amd.utils import arg_largest_magnitude
A = matrices.banded_csr(1_500_000, 35, 1234)
for real in (True, False):
    op = CsrOperator(A, real=real)
    np.random.seed(0)
    s = (RealKrylovSchurSolver if real else KrylovSchurSolver)(op, 20, 41, 25, 1e-8, arg_largest_magnitude)
    s.start()
    for i in range(2):
        s.contract(i); s.expand()
    torch.cuda.synchronize()
    pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable()
    for i in range(10):
        c = s.contract(2 + i); s.expand()
    torch.cuda.synchronize(); pr.disable()
    print("real" if real else "complex", f"{(time.perf_counter()-t0)/10*1e3:.3f} ms per restart; p_now", getattr(s, "p_now", s.p), "m", s.m, "second passes", s.ctx.last_ctrl.second_passes, "steps", s.ctx.last_ctrl.steps_done)
    pstats.Stats(pr).sort_stats("tottime").print_stats(8)
