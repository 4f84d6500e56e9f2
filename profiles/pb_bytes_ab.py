#!/usr/bin/env python3
"""the two experiments that change the BYTES or the PIECE SIZE of the tile-binned SpMV on BASELINE
config 5 (random CSR n = 10M, 5 per row), measured instead of argued:

  (a) sub-slabs of 10 240 columns (the whole 160 KiB of LDS; build the library with -DAKS_PB_SLAB_COLS=10240): the
      (sub-slab, row block) tiles -- the pieces phase 2 gathers -- hold 1.25 x as many products; traffic unchanged;
  (b) column groups: phase 1 + phase 2 per GROUP of sub-slabs (2, 4 or 8 groups), y accumulated from group to group, all
      groups writing their products into ONE scratch buffer of nnz / groups entries, so that a group's products
      (200 MB at 4 groups) can stay in the 256 MB Infinity Cache between the two phases -- 1.6 GB of HBM traffic to
      save against 0.32 GB of y re-streamed per extra group (library with plain product stores: -DAKS_PB_NT_STORE=0,
      and as shipped with non-temporal ones).

    AKS_LIB_PATH=... python profiles/pb_bytes_ab.py [n] [groups,groups,...]

prints ms per SpMV (HIP events over 10 launches after 3 warm-ups) and the largest difference from the one-pass result."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "arnoldi-py_amd"))
import numpy as np, scipy.sparse as sp, torch
from arnoldi_amd import _hip, device as dev, matrices

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    groups = [int(g) for g in (sys.argv[2] if len(sys.argv) > 2 else "2,4,8").split(",") if g]
    A = matrices.random_csr(n, 5, 1234)
    slab_bits, _, _ = _hip.pb_params()
    x = torch.randn(n, dtype=torch.complex128, device="cuda")
    y = torch.empty(n, dtype=torch.complex128, device="cuda")

    def timed(fn):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 10

    one = dev.DeviceCSR(A); one.autotune(force="binned")
    ms = timed(lambda: one.spmv(x, y))
    ref = y.clone()
    alg = 12 * A.nnz + 36 * n + 4
    tiles = one.binned.n_slabs
    print(f"lib {os.path.basename(os.path.dirname(_hip.LIB_PATH))}: n={n} nnz={A.nnz}  sub-slabs {tiles}")
    print(f"  one pass (phase 1 + phase 2)            {ms:8.4f} ms  {alg / ms / 1e9:6.3f} TB/s algorithmic = {alg / ms / 8e9:.4f} of 8 TB/s;"
          f"  lanes per wave-load {one.binned.lanes_per_load:.1f}, levels per round {one.binned.levels_per_round:.2f}")
    del one
    col_of = A.indices
    row_of = np.repeat(np.arange(n, dtype=np.int64), np.diff(A.indptr))
    for G in groups:
        # contiguous column groups (multiples of 8192 columns: no sub-slab is shared by two groups for the shipped width)
        edges = (np.linspace(0, (n + 8191) // 8192, G + 1).astype(np.int64) * 8192).clip(0, n)
        parts = []
        for g in range(G):
            keep = (col_of >= edges[g]) & (col_of < edges[g + 1])
            indptr = np.concatenate([[0], np.cumsum(np.bincount(row_of[keep], minlength=n))]).astype(np.int32)
            M = sp.csr_matrix((A.data[keep], A.indices[keep], indptr), shape=A.shape)
            d = dev.DeviceCSR(M); d.autotune(force="binned")
            parts.append(d)
        shared = torch.zeros(max(int(p.binned.desc.nnz_pad) for p in parts), dtype=torch.complex128, device="cuda")
        for p in parts:
            p.binned.prod = shared                       # ONE product scratch for all groups: written, read, overwritten
            p.binned.desc.d_prod = shared.data_ptr()

        def run():
            for g, p in enumerate(parts):
                p.spmv(x, y, accumulate=g > 0)
        ms = timed(run)
        err = float((y - ref).abs().max() / ref.abs().max())
        print(f"  {G} column groups, shared {shared.numel() * 16 / 1e6:6.0f} MB scratch   {ms:8.4f} ms  {alg / ms / 1e9:6.3f} TB/s algorithmic = "
              f"{alg / ms / 8e9:.4f};  max rel diff from one pass {err:.1e}")
        del parts, shared
        torch.cuda.empty_cache()

if __name__ == "__main__":
    main()
