#!/usr/bin/env python3
"""Where does the host spend a restart at the 8-GPU shard size?  (Round 3: at n = 1.25M-2M the same build ran at
2.5 ms or at 5.7 ms per restart from one process to the next on one box, with identical kernel times.)

    python profiles/host_gap_probe.py [n] [restarts]

One process: Krylov-Schur restarts on random CSR, wall time per restart split into the host-side pieces (Schur step,
coefficient upload + truncate launch, the C call that enqueues the expansion, the wait for H), plus the CPU the
thread ran on and its context switches.  Prints one JSON line.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "arnoldi-py_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

torch.cuda.set_device(0)
from arnoldi_amd import device as dev  # noqa: E402
from arnoldi_amd import matrices  # noqa: E402
from arnoldi_amd.engine import CsrOperator  # noqa: E402
from arnoldi_amd.krylov_schur import KrylovSchurSolver  # noqa: E402
from arnoldi_amd.utils import arg_largest_magnitude  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
restarts = int(sys.argv[2]) if len(sys.argv) > 2 else 40
acc = {}


def timed(name, fn):
    def wrapper(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
    return wrapper


A = matrices.random_csr(n, 5, 1234)
op = CsrOperator(A)
np.random.seed(0)
s = KrylovSchurSolver(op, 5, 20, 10, 1e-8, arg_largest_magnitude)
ctx = s.ctx
ctx.use_graph = os.environ.get("PROBE_GRAPH", "0") == "1"
s.start()
for i in range(3):
    s.contract(i)
    s.expand()
ctx.truncate = timed("upload_Q+truncate_launch", ctx.truncate)
ctx._expand_native = timed("expand_C_call", ctx._expand_native)
_fetch = dev.fetch_H_and_ctrl


def fetch_wrapped(b, ws):
    t0 = time.perf_counter()
    w = _fetch(b, ws)
    acc["queue_H_copy"] = acc.get("queue_H_copy", 0.0) + time.perf_counter() - t0
    return timed("wait_for_H", w)


dev.fetch_H_and_ctrl = fetch_wrapped
cpus = set()
import ctypes  # noqa: E402
_libc = ctypes.CDLL(None)


def ctxsw():
    d = {}
    for line in open("/proc/self/status"):
        if "ctxt_switches" in line:
            k, v = line.split(":")
            d[k.strip()] = int(v)
    return d


sw0 = ctxsw()
torch.cuda.synchronize()
t_all = time.perf_counter()
t_contract = t_expand = 0.0
for i in range(restarts):
    t0 = time.perf_counter()
    s.contract(3 + i)
    t1 = time.perf_counter()
    s.expand()
    t2 = time.perf_counter()
    t_contract += t1 - t0
    t_expand += t2 - t1
    cpus.add(int(_libc.sched_getcpu()))
torch.cuda.synchronize()
t_all = time.perf_counter() - t_all
sw1 = ctxsw()
pr = torch.cuda.get_device_properties(0)
bus = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
try:
    gpu_node = int(open(f"/sys/bus/pci/devices/{bus}/numa_node").read())
except OSError:
    gpu_node = None
out = {"n": n, "gpu_pci": bus, "gpu_numa_node": gpu_node, "ms_per_restart": round(t_all / restarts * 1e3, 3),
       "contract_ms": round(t_contract / restarts * 1e3, 3), "expand_ms": round(t_expand / restarts * 1e3, 3),
       "pieces_ms": {k: round(v / restarts * 1e3, 3) for k, v in acc.items()},
       "cpus_seen": sorted(cpus), "affinity": len(os.sched_getaffinity(0)),
       "ctx_switches": {k: sw1[k] - sw0[k] for k in sw1}, "graph": ctx.use_graph,
       "env": {k: v for k, v in os.environ.items() if k.startswith(("HSA_", "HIP_", "GPU_", "AMD_", "OMP_", "OPENBLAS", "MKL_"))}}
print(json.dumps(out), flush=True)
