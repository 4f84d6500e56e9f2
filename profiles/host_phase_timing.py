#!/usr/bin/env python3
"""A/B inside ONE process (the rate at this size moves by 2x between processes and boxes): Krylov-Schur
restarts of the 8-GPU shard size (random CSR n = 1.25M) through
  native   aks_arnoldi_expand on one GPU, no communicator
  rccl-c   the same entry point with the library's RCCL communicator on a one-rank group: the
           all-reduces between the Gram-Schmidt stages are issued from C (two per step)
  rccl-py  the stages chained from Python with torch.distributed all-reduces (AKS_DIST_PATH=python)
alternating blocks of restarts.  Usage (GPU box): python profiles/host_phase_timing.py [n] [blocks]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "arnoldi-py_amd"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29539")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

torch.cuda.set_device(0)
from arnoldi_amd import matrices  # noqa: E402
from arnoldi_amd.dist import Comm  # noqa: E402
from arnoldi_amd.engine import CsrOperator  # noqa: E402
from arnoldi_amd.krylov_schur import KrylovSchurSolver  # noqa: E402
from arnoldi_amd.utils import arg_largest_magnitude  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
A = matrices.random_csr(n, 5, 1234)
solvers = {}
for name in ("native", "rccl-c", "rccl-py"):
    if name == "rccl-py":
        os.environ["AKS_DIST_PATH"] = "python"
    comm = None if name == "native" else Comm(force=True)
    op = CsrOperator(A, comm=comm)
    os.environ.pop("AKS_DIST_PATH", None)
    np.random.seed(0)
    s = KrylovSchurSolver(op, 5, 20, 10, 1e-8, arg_largest_magnitude, comm=comm)
    s.start()
    for i in range(3):
        s.contract(i)
        s.expand()
    solvers[name] = [s, 3, 0.0, 0]
    print(name, "c_driven", op.c_driven, "native_comm", op.native_comm, "collectives/step", s.ctx.collectives_per_step(), flush=True)
torch.cuda.synchronize()
for b in range(blocks):
    for name, st in solvers.items():
        s = st[0]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            s.contract(st[1])
            s.expand()
            st[1] += 1
        torch.cuda.synchronize()
        st[2] += time.perf_counter() - t0
        st[3] += 10
base = solvers["native"][2] / solvers["native"][3]
for name, st in solvers.items():
    per = st[2] / st[3]
    print(f"{name:8s} {per * 1e3:7.3f} ms per restart  ({per / base:5.3f} x native), {st[3]} restarts", flush=True)
dist.destroy_process_group()
