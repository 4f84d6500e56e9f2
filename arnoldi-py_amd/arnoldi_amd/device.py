"""Device-resident state of one row shard: CSR blocks, Krylov basis, workspace.

torch is used for exactly three things: HBM allocations, the current HIP stream,
and (in ``dist.py``) ``torch.distributed``.  All arithmetic goes through the C ABI
of ``libarnoldi_hip.so`` (``_hip.py``); nothing here computes on the CPU.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import scipy.sparse as sp
import torch

from . import _hip

C128 = np.complex128
ETA_DGKS = float(np.sqrt(0.5))  # reference: src/arnoldi/ortho.py:6


def _require_gpu(device):
    if not torch.cuda.is_available():
        raise _hip.HipLibraryError(
            "no HIP device visible: arnoldi_amd runs its hot path on an MI355X only "
            "(there is no CPU fallback)"
        )
    return torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def canonical_csr(A):
    """scipy sparse / dense ndarray -> canonical CSR (sorted, duplicates summed),
    int32 indices, float64 or complex128 values.  Returns None for opaque operators."""
    if sp.issparse(A):
        M = sp.csr_matrix(A)
    elif isinstance(A, np.ndarray) and A.ndim == 2:
        M = sp.csr_matrix(A)
    else:
        return None
    if not M.has_canonical_format:
        M = M.copy()
        M.sum_duplicates()
    if M.nnz >= 2**31 - 1 or max(M.shape) >= 2**31 - 1:
        raise _hip.HipLibraryError("matrices with >= 2^31 rows or non-zeros are not supported")
    dt = C128 if np.iscomplexobj(M.data) else np.float64
    return sp.csr_matrix(
        (np.ascontiguousarray(M.data, dtype=dt), M.indices.astype(np.int32, copy=False),
         M.indptr.astype(np.int32, copy=False)), shape=M.shape)


def choose_lanes_per_row(n_rows, nnz):
    """Lanes that share one row in the LDS row-sum phase of the SpMV kernel."""
    mean = nnz / max(n_rows, 1)
    lpr = 1
    while lpr < 64 and mean > 8 * lpr:
        lpr *= 2
    return lpr


class DeviceCSR:
    """One CSR block in HBM plus its wave-tile plan (``aks_csr_plan_tiles``).

    Replaces the operator side of ``A @ V[:, j]`` (src/arnoldi/decomposition.py:58).
    """

    def __init__(self, M, device=None, lanes_per_row=0):
        device = _require_gpu(device)
        M = canonical_csr(M)
        if M is None:
            raise TypeError("DeviceCSR needs a scipy sparse matrix or a dense 2-D array")
        lib = _hip.load()
        self.shape = M.shape
        self.n_rows, self.n_cols = M.shape
        self.nnz = int(M.nnz)
        self.values_complex = int(M.data.dtype == C128)
        self.device = device
        indptr = np.ascontiguousarray(M.indptr, dtype=np.int32)
        cap = self.n_rows + 2
        tiles = np.empty(cap, np.int32)
        nt = lib.aks_csr_plan_tiles(indptr.ctypes.data, self.n_rows, _hip.SPMV_TILE_NNZ,
                                    tiles.ctypes.data, cap)
        _hip.check(nt, "aks_csr_plan_tiles")
        self.n_tiles = int(nt)
        self.lanes_per_row = lanes_per_row or choose_lanes_per_row(self.n_rows, self.nnz)
        self.indptr = torch.from_numpy(indptr).to(device)
        self.indices = torch.from_numpy(np.ascontiguousarray(M.indices, dtype=np.int32)).to(device)
        self.values = torch.from_numpy(np.ascontiguousarray(M.data)).to(device)
        self.tiles = torch.from_numpy(tiles[: self.n_tiles + 1].copy()).to(device)

    def algorithmic_bytes(self):
        """12 nnz + 36 n + 4 (f64 values) or 20 nnz + 36 n + 4 (c128 values)."""
        per_nnz = 20 if self.values_complex else 12
        return per_nnz * self.nnz + 36 * self.n_rows + 4

    def spmv(self, x, y, accumulate=False, ws=None):
        """y (=|+=) A x on the current stream; x, y are complex128 device tensors."""
        assert x.dtype == torch.complex128 and y.dtype == torch.complex128
        assert x.numel() >= self.n_cols and y.numel() >= self.n_rows
        assert x.is_contiguous() and y.is_contiguous()
        rc = _hip.load().aks_csr_spmv(
            self.n_rows, _ptr(self.indptr), _ptr(self.indices), _ptr(self.values), self.values_complex,
            _ptr(self.tiles), self.n_tiles, self.lanes_per_row, _ptr(x), _ptr(y), int(accumulate),
            _ptr(ws.buf) if ws is not None else C.c_void_p(0), _stream())
        _hip.check(rc, "aks_csr_spmv")


class Workspace:
    """Device scratch for the reductions + the control block (``aks_ws_layout``)."""

    def __init__(self, n_rows, max_dim, device=None):
        device = _require_gpu(device)
        self.n_rows, self.max_dim = int(n_rows), int(max_dim)
        self.layout = _hip.workspace_layout(self.n_rows, self.max_dim)
        self.nbytes = int(self.layout.total_bytes)
        self._raw = torch.empty(self.nbytes + 256, dtype=torch.uint8, device=device)
        skew = (-self._raw.data_ptr()) % 256
        self.buf = self._raw[skew: skew + self.nbytes]   # 256-byte aligned view
        assert self.buf.data_ptr() % 256 == 0
        self.reset()

    def reset(self):
        rc = _hip.load().aks_workspace_init(_ptr(self.buf), self.nbytes, self.n_rows, self.max_dim, _stream())
        _hip.check(rc, "aks_workspace_init")

    def _slot(self, off, n_c128):
        return self.buf[off: off + 16 * n_c128].view(torch.float64)

    def red(self, which, n_c128):
        """float64 view (2 doubles per complex) of reduction slot 1, 2 or 3 -- what a
        multi-GPU host all-reduces between the Gram-Schmidt stages."""
        off = {1: self.layout.red1_off, 2: self.layout.red2_off, 3: self.layout.red3_off}[which]
        return self._slot(off, n_c128)

    def read_ctrl(self):
        """Synchronising read-back of the 64-byte control block."""
        raw = self.buf[:64].cpu().numpy().tobytes()
        return _hip.Ctrl.from_buffer_copy(raw)


class KrylovBasis:
    """V (n x (m+1), column-major, ld = ldv) and the device copy of H ((m+1) x m, row-major).

    Mirrors the work arrays of src/arnoldi/krylov_schur.py:42-43.  ``V[j]`` is column j.
    """

    def __init__(self, n_rows, max_dim, device=None):
        device = _require_gpu(device)
        self.n_rows, self.max_dim = int(n_rows), int(max_dim)
        self.ldv = (self.n_rows + 63) // 64 * 64
        self.V = torch.zeros((self.max_dim + 1, self.ldv), dtype=torch.complex128, device=device)
        self.H = torch.zeros((self.max_dim + 1, self.max_dim), dtype=torch.complex128, device=device)
        self.device = device

    def col(self, j):
        return self.V[j]

    def set_col(self, j, host_vec):
        v = torch.from_numpy(np.ascontiguousarray(host_vec, dtype=C128))
        self.V[j, : self.n_rows].copy_(v)

    def get_cols(self, j0, j1):
        """Host copy, shape (n, j1-j0), Fortran order (like the reference's V views)."""
        out = self.V[j0:j1, : self.n_rows].cpu().numpy()  # (cols, n) C-order == (n, cols) F-order
        return out.T

    def set_cols(self, j0, host_cols):
        a = np.ascontiguousarray(np.asarray(host_cols, dtype=C128).T)
        self.V[j0: j0 + a.shape[0], : self.n_rows].copy_(torch.from_numpy(a))

    def download_H(self):
        return self.H.cpu().numpy()


# --------------------------------------------------------------------------- stage wrappers
def gs_project(basis, J, w, ws):
    rc = _hip.load().aks_gs_project(basis.n_rows, J, _ptr(basis.V), basis.ldv, _ptr(w), _ptr(ws.buf),
                                    ws.nbytes, ws.max_dim, _stream())
    _hip.check(rc, "aks_gs_project")


def gs_update_project(basis, J, w, ws):
    rc = _hip.load().aks_gs_update_project(basis.n_rows, J, _ptr(basis.V), basis.ldv, _ptr(w), _ptr(ws.buf),
                                           ws.nbytes, ws.max_dim, _stream())
    _hip.check(rc, "aks_gs_update_project")


def gs_update_norm(basis, J, w, ws, eta=ETA_DGKS):
    rc = _hip.load().aks_gs_update_norm(basis.n_rows, J, _ptr(basis.V), basis.ldv, _ptr(w), eta, _ptr(ws.buf),
                                        ws.nbytes, ws.max_dim, _stream())
    _hip.check(rc, "aks_gs_update_norm")


def gs_finish(basis, J, w, hcol, ldh, tol, ws, eta=ETA_DGKS, normalize=True):
    rc = _hip.load().aks_gs_finish(basis.n_rows, J, _ptr(w), C.c_void_p(hcol), ldh, tol, eta, int(normalize),
                                   _ptr(ws.buf), ws.nbytes, ws.max_dim, _stream())
    _hip.check(rc, "aks_gs_finish")


def dgks_gs_device(basis, J, w, hcol, ldh, tol, ws, eta=ETA_DGKS, normalize=True):
    rc = _hip.load().aks_dgks_gs(basis.n_rows, J, _ptr(basis.V), basis.ldv, _ptr(w), C.c_void_p(hcol), ldh,
                                 tol, eta, int(normalize), _ptr(ws.buf), ws.nbytes, ws.max_dim, _stream())
    _hip.check(rc, "aks_dgks_gs")


def truncate(basis, m, p, Qp_dev):
    rc = _hip.load().aks_truncate(basis.n_rows, m, p, _ptr(basis.V), basis.ldv, _ptr(Qp_dev), _stream())
    _hip.check(rc, "aks_truncate")


def gather_c128(count, idx, src, dst):
    rc = _hip.load().aks_gather_c128(count, _ptr(idx), _ptr(src), _ptr(dst), _stream())
    _hip.check(rc, "aks_gather_c128")
