"""Where the host time of BinnedCSR.__init__ goes at n = 10M (the steps replicated with a timer around each).
    AKS_PLAN_TIMING=1 python profiles/binned_setup_steps.py"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "arnoldi-py_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from arnoldi_amd import _hip, device as dev, matrices  # noqa: E402

A = matrices.random_csr(10_000_000, 5, 1234)
M = dev.canonical_csr(A)
torch.zeros(1, device="cuda")
torch.cuda.synchronize()
lib = _hip.load()
device = torch.device("cuda")
for rep in range(2):
    T = [("start", time.perf_counter())]
    lap = lambda name: T.append((name, time.perf_counter()))  # noqa: E731
    indptr = np.ascontiguousarray(M.indptr, dtype=np.int32)
    indices = np.ascontiguousarray(M.indices, dtype=np.int32)
    values = np.ascontiguousarray(M.data)
    sz = _hip.PbSizes()
    plan = lib.aks_pb_plan_create(indptr.ctypes.data, indices.ctypes.data, values.ctypes.data, 0, M.shape[0], M.shape[1], C.byref(sz))
    lap("plan_create")
    val = np.empty(sz.nnz_pad, values.dtype)
    lcol = np.empty(sz.nnz_pad, np.uint16)
    slab_begin = np.empty(sz.n_slabs, np.int32)
    slab_end = np.empty(sz.n_slabs, np.int32)
    runs = np.empty((sz.n_runs, 4), np.uint32)
    rb_run_ptr = np.empty(sz.n_rowblocks + 1, np.int32)
    lrow = np.empty(sz.n_lrow, np.uint16)
    lap("np.empty x 7")
    lib.aks_pb_plan_export(plan, val.ctypes.data, lcol.ctypes.data, slab_begin.ctypes.data, slab_end.ctypes.data, runs.ctypes.data,
                           rb_run_ptr.ctypes.data, lrow.ctypes.data)
    lap("plan_export")
    lib.aks_pb_plan_destroy(plan)
    lap("plan_destroy")
    narrow = {np.dtype(np.uint16): np.int16, np.dtype(np.uint32): np.int32}
    ups = []
    for name, a in (("val", val), ("lcol", lcol), ("lrow", lrow), ("runs", runs), ("small", slab_begin)):
        ups.append(torch.from_numpy(a.view(narrow.get(a.dtype, a.dtype))).to(device))
        lap(f"upload {name} ({a.nbytes / 1e6:.0f} MB)")
    prod = torch.zeros(int(sz.nnz_pad), dtype=torch.complex128, device=device)
    lap("prod zeros")
    rpr = _hip.PB_RUNS_PER_ROUND
    info = runs[:-rpr, 3]
    lv = float(((info[::rpr] >> 21) & 15).mean())
    filled = np.count_nonzero((info >> 14) & 127)
    lap("stats")
    del val, lcol, lrow, runs
    lap("free numpy arrays")
    torch.cuda.synchronize()
    lap("synchronize")
    print(f"rep {rep}: " + ", ".join(f"{b[0]} {(b[1] - a[1]) * 1e3:.1f}" for a, b in zip(T, T[1:])) + f"; total {(T[-1][1] - T[0][1]) * 1e3:.1f} ms")
    del ups, prod
