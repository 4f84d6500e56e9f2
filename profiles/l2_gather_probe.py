import os, sys
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "arnoldi-py_amd"))
import numpy as np, scipy.sparse as sp, torch
from arnoldi_amd.device import DeviceCSR
n = 10_000_000
rng = np.random.default_rng(0)
for width_bits in (14, 16, 17, 18, 20, 24):
    W = 1 << width_bits
    cols = np.sort(rng.integers(0, W, (n, 5), dtype=np.int64), axis=1).astype(np.int32)
    A = sp.csr_matrix((rng.uniform(-1, 1, (n, 5)).ravel(), cols.ravel(), np.arange(0, 5*n+1, 5, dtype=np.int32)), shape=(n, W if W >= 1 else 1))
    dA = DeviceCSR.__new__(DeviceCSR)
    # bypass canonical (duplicates are fine for timing): build manually
    from arnoldi_amd import _hip
    import ctypes as C
    lib = _hip.load()
    indptr = A.indptr.astype(np.int32); tiles = np.empty(n + 2, np.int32)
    nt = lib.aks_csr_plan_tiles(indptr.ctypes.data, n, 256, tiles.ctypes.data, n + 2)
    dA.n_rows, dA.n_cols, dA.nnz, dA.values_complex = n, W, A.nnz, 0
    dA.n_tiles, dA.lanes_per_row = int(nt), 1
    dA.indptr = torch.from_numpy(indptr).cuda(); dA.indices = torch.from_numpy(A.indices.astype(np.int32)).cuda()
    dA.values = torch.from_numpy(A.data).cuda(); dA.tiles = torch.from_numpy(tiles[:nt+1].copy()).cuda()
    x = torch.randn(max(W, 1), dtype=torch.complex128, device="cuda"); y = torch.empty(n, dtype=torch.complex128, device="cuda")
    for _ in range(3): dA.spmv(x, y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): dA.spmv(x, y)
    e1.record(); torch.cuda.synchronize()
    print(f"x slab 2^{width_bits} entries ({W*16/2**20:.2f} MiB): {e0.elapsed_time(e1)/20:.4f} ms", flush=True)
