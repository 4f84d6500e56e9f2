"""Host time of SlicedCSR.__init__ on the shell-structured config-3 matrix, step by step, with the planner's output in
ordinary numpy arrays and in an anonymous private mapping with MADV_HUGEPAGE (tried in round 4: no gain -- the 41 ms it
takes to release 0.6 GB after the uploads are the same either way; a SHARED anonymous map, Python's default, is 10 x slower
to fault in).
    python profiles/sliced_setup_steps.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "arnoldi-py_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from arnoldi_amd import _hip, device as dev, matrices  # noqa: E402


def huge_scratch(count, dtype):
    import mmap

    nbytes = int(count) * np.dtype(dtype).itemsize
    mm = mmap.mmap(-1, nbytes, flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS)
    mm.madvise(mmap.MADV_HUGEPAGE)
    return np.frombuffer(mm, dtype=dtype, count=int(count))


print(open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip())
M = dev.canonical_csr(matrices.shell_csr(549, 549, 5, 1234))
torch.zeros(1, device="cuda")
torch.cuda.synchronize()
lib = _hip.load()
device = torch.device("cuda")
indptr = np.ascontiguousarray(M.indptr, dtype=np.int32)
indices = np.ascontiguousarray(M.indices, dtype=np.int32)
values = np.ascontiguousarray(M.data)
n = M.shape[0]
for rep in range(2):
    for name, alloc in (("np.empty", lambda c, d: np.empty(c, d)), ("huge-page map", huge_scratch)):
        T = [("start", time.perf_counter())]
        lap = lambda what: T.append((what, time.perf_counter()))  # noqa: E731
        nnz_pad = int(lib.aks_sell_plan_size(indptr.ctypes.data, n))
        lap("plan_size")
        slice_ptr = np.empty((n + 63) // 64 + 1, np.int64)
        col = alloc(nnz_pad, np.int32)
        val = alloc(nnz_pad, values.dtype)
        lap("allocate")
        lib.aks_sell_plan_fill(indptr.ctypes.data, indices.ctypes.data, values.ctypes.data, 0, n, slice_ptr.ctypes.data, col.ctypes.data,
                               val.ctypes.data)
        lap("plan_fill")
        d_col = torch.from_numpy(col).to(device)
        lap(f"upload col ({col.nbytes / 1e6:.0f} MB)")
        d_val = torch.from_numpy(val).to(device)
        lap(f"upload val ({val.nbytes / 1e6:.0f} MB)")
        del col, val
        lap("free host arrays")
        torch.cuda.synchronize()
        lap("synchronize")
        print(f"rep {rep} {name}: " + ", ".join(f"{b[0]} {(b[1] - a[1]) * 1e3:.1f}" for a, b in zip(T, T[1:])) + f"; total {(T[-1][1] - T[0][1]) * 1e3:.1f} ms")
        del d_col, d_val
