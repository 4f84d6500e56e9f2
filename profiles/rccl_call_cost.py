#!/usr/bin/env python3
"""What does one small all-reduce cost on a ONE-rank RCCL communicator (the only kind a one-GPU box can
build)?  Host time per call (enqueue) and stream time per call, for the library's communicator
(aks_comm_allreduce_sum -> ncclAllReduce) and for torch.distributed's, next to an empty kernel launch.
    python profiles/rccl_call_cost.py          (GPU box)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "arnoldi-py_amd")):
    sys.path.insert(0, p)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29537")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from arnoldi_amd import _hip, device as dev  # noqa: E402
from arnoldi_amd.dist import Comm  # noqa: E402

comm = Comm(force=True)
handle = comm.native()
lib = _hip.load()
buf = torch.zeros(64, dtype=torch.float64, device="cuda")
N = 2000


def timed(name, call):
    for _ in range(50):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(N):
        call()
    host = time.perf_counter() - t0
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:48s} host {host / N * 1e6:7.2f} us/call   stream {e0.elapsed_time(e1) / N * 1e3:7.2f} us/call", flush=True)


stream = dev._stream()
ptr = dev._ptr(buf)
timed("aks_comm_allreduce_sum (42 doubles, 1 rank)", lambda: lib.aks_comm_allreduce_sum(handle, ptr, 42, stream))
timed("torch.distributed.all_reduce (42 doubles, 1 rank)", lambda: dist.all_reduce(buf[:42]))
timed("aks_scale on 64 rows (one tiny kernel)", lambda: lib.aks_scale(64, ptr, 1.0, 0.0, stream))
comm.close()
dist.destroy_process_group()
