#!/usr/bin/env python3
"""cProfile of the host side of the restart loop (GPU box): where the wall time between kernels goes.
    python profiles/host_profile.py [n] [restarts]"""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "arnoldi-py_amd"))
import numpy as np, torch
from arnoldi_amd import matrices
from arnoldi_amd.engine import CsrOperator
from arnoldi_amd.krylov_schur import KrylovSchurSolver
from arnoldi_amd.utils import arg_largest_magnitude

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
R = int(sys.argv[2]) if len(sys.argv) > 2 else 30
if os.environ.get("USE_STREAM"):
    torch.cuda.set_stream(torch.cuda.Stream())
comm = None
if os.environ.get("FORCE_COMM"):      # one-rank RCCL group: the multi-GPU host path with real collectives
    import torch.distributed as dist
    from arnoldi_amd.dist import Comm
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29547")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    comm = Comm(force=True)
op = CsrOperator(matrices.random_csr(n, 5, 1234), comm=comm)
np.random.seed(0)
s = KrylovSchurSolver(op, 5, 20, 10, 1e-8, arg_largest_magnitude, comm=comm)
if os.environ.get("CHAINED") or comm is not None:
    s.ctx.force_chained = True
s.start()
for i in range(3):
    s.contract(i); s.expand()
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for i in range(R):
    s.contract(3 + i); s.expand()
torch.cuda.synchronize()
pr.disable()
print(f"n={n}: {(time.perf_counter()-t0)/R*1e3:.3f} ms per restart")
pstats.Stats(pr).sort_stats(os.environ.get("SORT", "cumulative")).print_stats(int(os.environ.get("TOP", "6")))
