"""Device-resident state of one row shard: CSR blocks, Krylov basis, workspace.

Memory, streams and events come from ``mem.py`` -- torch tensors by default, or raw HIP allocations through ctypes
(``AKS_HOST_ALLOC=hip``: no torch in the process at all).  All arithmetic goes through the C ABI
of ``libarnoldi_hip.so`` (``_hip.py``); nothing here computes on the CPU.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np
import scipy.sparse as sp

from . import _hip, mem

C128 = np.complex128
ETA_DGKS = float(np.sqrt(0.5))  # reference: src/arnoldi/ortho.py:6


def _require_gpu(device):
    if not mem.gpu_available():
        raise _hip.HipLibraryError(
            "no HIP device visible: arnoldi_amd runs its hot path on an MI355X only "
            "(there is no CPU fallback)"
        )
    device = mem.as_device(device)
    current = mem.current_device()
    if device.index is not None and device.index != current:
        # Kernels are launched on the CURRENT device's stream (one process per GPU: SURVEY 8(b), 8(e)); buffers on another
        # device would be handed to kernels that run here -- a fault at best.  Refuse, and say what to do.
        raise _hip.HipLibraryError(
            f"device {device.index} was asked for but device {current} is current: make it current first "
            f"(arnoldi_amd.mem.set_device({device.index}); one process drives one GPU) -- a solve runs on the current device's stream")
    _hip.device_init(device.index if device.index is not None else current)
    return device


def _ptr(t):
    """Device pointer argument: a torch tensor, a raw address (int) or None."""
    if t is None:
        return C.c_void_p(0)
    return C.c_void_p(t if isinstance(t, int) else t.data_ptr())


_tls = threading.local()       # per host thread: independent solves may run concurrently, each on its own stream


def _stream():
    """Current HIP stream as a void*.  ``cached_stream()`` pins the lookup for a hot loop."""
    cached = getattr(_tls, "stream", None)
    if cached is not None:
        return cached
    return C.c_void_p(mem.stream_ptr())


class cached_stream:
    """Context manager: resolve torch's current stream once for a sequence of launches (the
    lookup costs several microseconds per call, comparable to a launch)."""

    def __enter__(self):
        self._prev = getattr(_tls, "stream", None)
        _tls.stream = None
        _tls.stream = _stream()
        return self

    def __exit__(self, *exc):
        _tls.stream = self._prev
        return False


def canonical_csr(A):
    """scipy sparse / dense ndarray -> canonical CSR (sorted, duplicates summed),
    int32 indices, float64 or complex128 values.  Returns None for opaque operators."""
    if sp.issparse(A):
        # (a CSR input is used as it is: a new csr_matrix object would forget what scipy already knows about its format)
        M = A if getattr(A, "format", None) == "csr" and isinstance(A, (sp.csr_matrix, sp.csr_array)) else sp.csr_matrix(A)
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
    out = sp.csr_matrix(
        (np.ascontiguousarray(M.data, dtype=dt), M.indices.astype(np.int32, copy=False),
         M.indptr.astype(np.int32, copy=False)), shape=M.shape)
    out.has_canonical_format = True      # just established: the next canonical_csr() of this object need not scan it again
    if M is A and M.data.dtype == dt and M.indices.dtype == np.int32 and M.indptr.dtype == np.int32:
        return A                         # already in the library's form: no new object either
    return out


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
        # The arrays of the CSR-stream kernel go to the device when that form is first needed (``_csr_arrays``): a
        # matrix that ends up in the binned or sliced form never uploads them (0.6 GB of HBM and of host-to-device
        # traffic at n = 10M).
        self._tiles_host = tiles[: self.n_tiles + 1].copy()
        self.indptr = self.indices = self.values = self.tiles = None

        self._host = M            # kept until the SpMV form has been chosen (autotune)
        self.binned = None        # BinnedCSR once built
        self.sliced = None        # SlicedCSR once built
        self.form = "csr"         # which form spmv() / aks_arnoldi_expand use: "csr" | "binned" | "sliced"

    @property
    def use_binned(self):
        return self.form == "binned"

    @use_binned.setter
    def use_binned(self, flag):
        self.form = "binned" if flag else "csr"

    def _csr_arrays(self):
        """Upload the CSR-stream kernel's arrays on first use (needs the host copy: before ``autotune`` releases it,
        or -- after it -- only if the CSR-stream form was the one chosen)."""
        if self.indptr is None:
            M = self._host
            if M is None:
                raise _hip.HipLibraryError("the CSR-stream form of this matrix was not kept: another form was chosen "
                                           "and the host copy released (build the DeviceCSR with force='csr')")
            self.indptr = mem.upload(np.ascontiguousarray(M.indptr, dtype=np.int32), self.device)
            self.indices = mem.upload(np.ascontiguousarray(M.indices, dtype=np.int32), self.device)
            self.values = mem.upload(np.ascontiguousarray(M.data), self.device)
            self.tiles = mem.upload(self._tiles_host, self.device)
        return self.indptr, self.indices, self.values, self.tiles

    def block(self, out=None):
        """Fill (and return) an ``aks_csr_block`` for this matrix with the SpMV form in use."""
        b = out if out is not None else _hip.CsrBlock()
        b.n_rows, b.n_cols = self.n_rows, self.n_cols
        if self.form == "csr":
            indptr, indices, values, tiles = self._csr_arrays()
            b.d_indptr, b.d_indices, b.d_values = indptr.data_ptr(), indices.data_ptr(), values.data_ptr()
            b.d_tiles, b.n_tiles = tiles.data_ptr(), self.n_tiles
        else:                                   # (apply_block dispatches on pb / sell first: these are never read)
            b.d_indptr = b.d_indices = b.d_values = b.d_tiles = None
            b.n_tiles = self.n_tiles
        b.values_complex, b.lanes_per_row = self.values_complex, self.lanes_per_row
        b.pb = C.pointer(self.binned.desc) if self.form == "binned" else None
        b.sell = C.pointer(self.sliced.desc) if self.form == "sliced" else None
        return b

    def algorithmic_bytes(self, real=False):
        """12 nnz + 36 n + 4 (f64 values) or 20 nnz + 36 n + 4 (c128 values); with real
        vectors x and y are 8 bytes per row: 12 nnz + 20 n + 4."""
        per_nnz = 20 if self.values_complex else 12
        return per_nnz * self.nnz + (20 if real else 36) * self.n_rows + 4

    # -- choice of SpMV form ------------------------------------------------------------------
    def scatter_ratio(self, sample_windows=64):
        """Distinct 128-byte lines of x touched per non-zero in windows of 64 consecutive rows
        (~0.1 for stencils, ~1 for random graphs): how little L2 reuse the CSR kernel can get."""
        M = self._host
        if M is None or self.nnz == 0:
            return 0.0
        starts = np.linspace(0, max(self.n_rows - 64, 0), sample_windows).astype(np.int64)
        lines = nnz = 0
        for r in starts:
            k0, k1 = M.indptr[r], M.indptr[min(r + 64, self.n_rows)]
            if k1 > k0:
                lines += np.unique(M.indices[k0:k1] >> 3).size
                nnz += k1 - k0
        return lines / max(nnz, 1)

    def build_binned(self):
        if self.binned is None:
            if self._host is None:
                raise _hip.HipLibraryError("host copy of the matrix already released")
            self.binned = BinnedCSR(self._host, self.device)
        return self.binned

    SLICED_MAX_PADDING = 1.25      # the sliced form is tried while its padded size stays below this x nnz

    def sliced_padding(self):
        """Padded size of the sliced form relative to nnz (1.0 = every row of a slice equally long)."""
        if self._host is None or self.nnz == 0:
            return float("inf")
        indptr = np.ascontiguousarray(self._host.indptr, dtype=np.int32)
        return int(_hip.check(_hip.load().aks_sell_plan_size(indptr.ctypes.data, self.n_rows), "aks_sell_plan_size")) / self.nnz

    def build_sliced(self):
        if self.sliced is None:
            if self._host is None:
                raise _hip.HipLibraryError("host copy of the matrix already released")
            self.sliced = SlicedCSR(self._host, self.device)
        return self.sliced

    def autotune(self, min_nnz=2_000_000, reps=3, force=None, real=False, measure=None):
        """Pick the SpMV form.  Candidates by STRUCTURE: the tile-binned form for matrices that are large and
        scattered enough to miss L2 (distinct x lines per non-zero > 0.5, >= 2M non-zeros, x larger than an
        XCD's L2); the sliced form for matrices with column locality whose rows are of similar length (padding
        <= 1.25 x nnz); the CSR-stream kernel otherwise.  By default a candidate is taken as it is -- every
        measurement so far has it ahead (binned 2x on random graphs, sliced 1.1-1.35x on stencils, bands and
        the Markov chain) -- so the choice, and with it the summation order of a row and the BITS of a solve,
        is a function of the matrix BLOCK alone: identical from run to run.  (In a row-sharded solve every rank
        tunes its own diagonal and off-diagonal block, so the forms -- and with them the summation order and whether
        normalisations are deferred -- follow the sharding; results of different shardings agree to rounding, not
        bit for bit.)
        ``measure=True`` (or AKS_SPMV_TUNE=measure) times the candidate against the CSR-stream kernel on this
        device instead and keeps it only if it is >= 10 % (binned) / 5 % (sliced) faster: the answer then
        depends on a timing (``tune_ms`` records it).
        ``force`` = "csr" | "binned" | "sliced" skips both.  Frees the host copy.
        ``real``: time the real-vector kernels (real-packed mode)."""
        if measure is None:
            measure = os.environ.get("AKS_SPMV_TUNE", "structure") == "measure"
        choice = force
        self.tune_ms = {}
        self.tune_mode = "forced" if force else ("measured" if measure else "structure")
        if choice is None:
            forms = ["csr"]
            scattered = self.nnz >= min_nnz // 4 and self.scatter_ratio() > 0.5      # gathers without locality
            if scattered and self.nnz >= min_nnz and self.n_cols * 16 > (4 << 20):
                try:
                    self.build_binned()
                    forms.append("binned")
                except _hip.HipLibraryError:      # too many tiles / non-zeros for the binned form
                    pass
            if not scattered and self.nnz >= min_nnz // 4 and self.sliced_padding() <= self.SLICED_MAX_PADDING:
                self.build_sliced()
                forms.append("sliced")
            if len(forms) == 1:
                choice = "csr"
            elif not measure:
                choice = forms[1]
        if choice is None:
            x = mem.zeros(self.n_cols, mem.c128, self.device)
            y = mem.empty(self.n_rows, mem.c128, self.device)
            times = {}
            for form in forms:                 # two warm-up launches, then the fastest of 2 * reps timed ones
                self.form = form
                self.spmv(x, y, real=real)
                self.spmv(x, y, real=real)
                marks = [mem.Event(enable_timing=True) for _ in range(2 * reps + 1)]
                marks[0].record()
                for e in marks[1:]:
                    self.spmv(x, y, real=real)
                    e.record()
                mem.synchronize()
                times[form] = min(a.elapsed_time(b) for a, b in zip(marks, marks[1:]))
            self.tune_ms = times
            choice = "csr"
            best = min((f for f in forms if f != "csr"), key=times.get)
            if times[best] < (0.9 if best == "binned" else 0.95) * times["csr"]:
                choice = best
        if choice == "binned":
            self.build_binned()
        elif choice == "sliced":
            self.build_sliced()
        if choice != "binned":
            self.binned = None
        if choice != "sliced":
            self.sliced = None
        self.form = choice
        if choice == "csr":
            self._csr_arrays()                  # (while the host copy is still there)
        else:
            self.indptr = self.indices = self.values = self.tiles = None     # (a measured tuning run uploaded them)
        self._host = None
        return choice

    def spmv(self, x, y, accumulate=False, ws=None, real=False):
        """y (=|+=) A x on the current stream; x, y are complex128 device tensors -- or, with ``real``,
        real vectors (float64 tensors, or real-packed complex128 columns: two rows per slot)."""
        for t, need in ((x, self.n_cols), (y, self.n_rows)):   # raw addresses (hot loop) skip the checks
            if not isinstance(t, int):
                have = t.numel() * (2 if (real and t.dtype == mem.c128) else 1)
                assert t.dtype == mem.c128 or (real and t.dtype == mem.f64)
                assert have >= need and t.is_contiguous()
        wsp = _ptr(ws.buf) if ws is not None else C.c_void_p(0)
        lib = _hip.load()
        if real:
            assert not self.values_complex, "real vectors need real matrix values"
        sfx = "_real" if real else ""
        if self.form == "binned":
            rc = getattr(lib, "aks_pb_spmv" + sfx)(C.byref(self.binned.desc), _ptr(x), _ptr(y), int(accumulate), wsp, _stream())
        elif self.form == "sliced":
            rc = getattr(lib, "aks_sell_spmv" + sfx)(C.byref(self.sliced.desc), _ptr(x), _ptr(y), int(accumulate), wsp, _stream())
        elif real:
            self._csr_arrays()
            rc = lib.aks_csr_spmv_real(self.n_rows, _ptr(self.indptr), _ptr(self.indices), _ptr(self.values),
                                       _ptr(self.tiles), self.n_tiles, self.lanes_per_row, _ptr(x), _ptr(y),
                                       int(accumulate), wsp, _stream())
        else:
            self._csr_arrays()
            rc = lib.aks_csr_spmv(self.n_rows, _ptr(self.indptr), _ptr(self.indices), _ptr(self.values),
                                  self.values_complex, _ptr(self.tiles), self.n_tiles, self.lanes_per_row, _ptr(x),
                                  _ptr(y), int(accumulate), wsp, _stream())
        _hip.check(rc, "aks_*_spmv" + sfx)


class SlicedCSR:
    """Sliced form of a CSR block (``aks_sell_matrix``): slices of 64 rows stored entry-major, planned on the
    host by ``aks_sell_plan_size`` / ``aks_sell_plan_fill`` and uploaded."""

    def __init__(self, M, device):
        lib = _hip.load()
        n_rows, n_cols = M.shape
        indptr = np.ascontiguousarray(M.indptr, dtype=np.int32)
        indices = np.ascontiguousarray(M.indices, dtype=np.int32)
        values = np.ascontiguousarray(M.data)
        cplx = int(values.dtype == C128)
        nnz_pad = int(_hip.check(lib.aks_sell_plan_size(indptr.ctypes.data, n_rows), "aks_sell_plan_size"))
        n_slices = (n_rows + 63) // 64
        slice_ptr = np.empty(n_slices + 1, np.int64)
        col = np.empty(max(nnz_pad, 1), np.int32)
        val = np.empty(max(nnz_pad, 1), values.dtype)
        _hip.check(lib.aks_sell_plan_fill(indptr.ctypes.data, indices.ctypes.data, values.ctypes.data, cplx, n_rows,
                                          slice_ptr.ctypes.data, col.ctypes.data, val.ctypes.data), "aks_sell_plan_fill")
        self.slice_ptr, self.col, self.val = (mem.upload(a, device) for a in (slice_ptr, col, val))
        self.padding = nnz_pad / max(int(M.nnz), 1)
        d = _hip.SellMatrix()
        d.n_rows, d.n_cols, d.nnz, d.nnz_pad, d.n_slices = n_rows, n_cols, int(M.nnz), nnz_pad, n_slices
        d.values_complex, d.pad_ = cplx, 0
        d.d_slice_ptr, d.d_col, d.d_val = self.slice_ptr.data_ptr(), self.col.data_ptr(), self.val.data_ptr()
        self.desc = d


class BinnedCSR:
    """Tile-binned two-phase form of a CSR block (``aks_pb_matrix``): arrays planned on the host by
    ``aks_pb_plan_create`` / ``aks_pb_plan_export``, uploaded, plus the product scratch."""

    def __init__(self, M, device):
        lib = _hip.load()
        n_rows, n_cols = M.shape
        nnz = int(M.nnz)
        indptr = np.ascontiguousarray(M.indptr, dtype=np.int32)
        indices = np.ascontiguousarray(M.indices, dtype=np.int32)
        cplx = int(M.data.dtype == C128)
        values = np.ascontiguousarray(M.data)
        sz = _hip.PbSizes()
        plan = lib.aks_pb_plan_create(indptr.ctypes.data, indices.ctypes.data, values.ctypes.data, cplx,
                                      n_rows, n_cols, C.byref(sz))
        if not plan:
            msg = lib.aks_last_error()
            raise _hip.HipLibraryError(f"aks_pb_plan_create failed: {msg.decode() if msg else '?'}")
        try:
            # the plan's own arrays, uploaded straight from where the planner wrote them (aks_pb_plan_view: no second
            # host copy of 0.6 GB at n = 10M); the views die with the plan
            arrs = _hip.PbPlanArrays()
            _hip.check(lib.aks_pb_plan_view(plan, C.byref(arrs)), "aks_pb_plan_view")

            def view(ptr, count, dtype):
                count = int(count)
                if count == 0:
                    return np.empty(0, dtype)
                buf = (C.c_char * (count * np.dtype(dtype).itemsize)).from_address(ptr)
                return np.frombuffer(buf, dtype=dtype, count=count)

            val = view(arrs.val, sz.nnz_pad, values.dtype)
            lcol = view(arrs.lcol, sz.nnz_pad, np.uint16)
            slab_begin = view(arrs.slab_begin, sz.n_slabs, np.int32)
            slab_end = view(arrs.slab_end, sz.n_slabs, np.int32)
            runs = view(arrs.runs, 4 * sz.n_runs, np.uint32).reshape(-1, 4)
            rb_run_ptr = view(arrs.rb_run_ptr, sz.n_rowblocks + 1, np.int32)
            lrow = view(arrs.lrow, sz.n_lrow, np.uint16)
            narrow = {np.dtype(np.uint16): np.int16, np.dtype(np.uint32): np.int32}     # torch has no unsigned 16/32
            aliasing = getattr(device, "type", "cuda") == "cpu"     # CPU "device" of tests/fake_hip.py: no copy is made
            up = lambda a: mem.upload((a.copy() if aliasing else a).view(narrow.get(a.dtype, a.dtype)), device)  # noqa: E731
            self.val, self.lcol, self.lrow, self.runs = up(val), up(lcol), up(lrow), up(runs)
            self.slab_begin, self.slab_end, self.rb_run_ptr = up(slab_begin), up(slab_end), up(rb_run_ptr)
            rpr = _hip.PB_RUNS_PER_ROUND
            info = runs[:-rpr, 3]                                   # (the last round is the planner's empty one)
            self.levels_per_round = float(((info[::rpr] >> 21) & 15).mean()) if len(info) else 0.0
            filled = np.count_nonzero((info >> 14) & 127)
            self.lanes_per_load = nnz / filled if filled else 0.0   # of 64
            # (mem.upload is synchronous for pageable host memory: the plan's arrays have been read when it returns)
            del val, lcol, slab_begin, slab_end, runs, rb_run_ptr, lrow, info
        finally:
            lib.aks_pb_plan_destroy(plan)
        self.prod = mem.zeros(int(sz.nnz_pad), mem.c128, device)
        self.n_slabs, self.n_rowblocks = int(sz.n_slabs), int(sz.n_rowblocks)
        d = _hip.PbMatrix()
        d.n_rows, d.n_cols, d.nnz, d.nnz_pad = n_rows, n_cols, nnz, int(sz.nnz_pad)
        d.n_runs, d.n_lrow = int(sz.n_runs), int(sz.n_lrow)
        d.n_slabs, d.n_rowblocks, d.values_complex, d.pad_ = int(sz.n_slabs), int(sz.n_rowblocks), cplx, 0
        d.d_val, d.d_lcol, d.d_lrow, d.d_runs = (t.data_ptr() for t in (self.val, self.lcol, self.lrow, self.runs))
        d.d_slab_begin, d.d_slab_end = self.slab_begin.data_ptr(), self.slab_end.data_ptr()
        d.d_rb_run_ptr, d.d_prod = self.rb_run_ptr.data_ptr(), self.prod.data_ptr()
        self.desc = d

    def moved_bytes(self, real=False):
        """Bytes the two phases stream per SpMV (excluding x and y): value + 2-byte column read and the
        product written in phase 1; product + the (level, row) words (64 per wave-load) + wave-load
        descriptors read in phase 2."""
        d = self.desc
        v, pr = (16 if d.values_complex else 8), (8 if real else 16)
        return (v + 2 + pr) * int(d.nnz_pad) + pr * int(d.nnz) + 2 * int(d.n_lrow) + 16 * int(d.n_runs)


class Workspace:
    """Device scratch for the reductions + the control block (``aks_ws_layout``)."""

    def __init__(self, n_rows, max_dim, device=None, real=False):
        device = _require_gpu(device)
        self.n_rows, self.max_dim = int(n_rows), int(max_dim)
        self.real = bool(real)      # real-packed panels: reductions drop the imaginary parts
        self.layout = _hip.workspace_layout(self.n_rows, self.max_dim)
        self.nbytes = int(self.layout.total_bytes)
        self._raw = mem.empty(self.nbytes + 256, mem.u8, device)
        skew = (-self._raw.data_ptr()) % 256
        self.buf = self._raw[skew: skew + self.nbytes]   # 256-byte aligned view
        assert self.buf.data_ptr() % 256 == 0
        self._red_views = {}
        self.reset()

    def reset(self):
        rc = _hip.load().aks_workspace_init(_ptr(self.buf), self.nbytes, self.n_rows, self.max_dim, _stream())
        _hip.check(rc, "aks_workspace_init")
        if self.real:
            _hip.check(_hip.load().aks_workspace_set_real(_ptr(self.buf), 1, _stream()), "aks_workspace_set_real")

    def _slot(self, off, n_c128):
        return self.buf[off: off + 16 * n_c128].view(mem.f64)

    def red(self, which, n_c128):
        """float64 view (2 doubles per complex) of reduction slot 1, 2 or 3 -- what a
        multi-GPU host all-reduces between the Gram-Schmidt stages."""
        key = (which, n_c128)
        view = self._red_views.get(key)
        if view is None:
            off = {1: self.layout.red1_off, 2: self.layout.red2_off, 3: self.layout.red3_off}[which]
            view = self._red_views[key] = self._slot(off, n_c128)
        return view

    def read_ctrl(self):
        """Synchronising read-back of the 64-byte control block."""
        raw = self.buf[:64].cpu().numpy().tobytes()
        return _hip.Ctrl.from_buffer_copy(raw)


class KrylovBasis:
    """V (n x (m+1), column-major, ld = ldv) and the device copy of H ((m+1) x m, row-major).

    Mirrors the work arrays of src/arnoldi/krylov_schur.py:42-43.  ``V[j]`` is column j.
    """

    def __init__(self, n_rows, max_dim, device=None, real=False):
        """``real``: real-packed basis -- ``n_rows`` real rows are stored two per complex slot, so the
        panel the kernels see has ``self.n_rows = ceil(n_rows / 2)`` rows (``self.n_real`` keeps the
        vector length).  Host vectors go in and come out as float64."""
        device = _require_gpu(device)
        self.real = bool(real)
        self.n_real = int(n_rows)
        self.n_rows, self.max_dim = ((int(n_rows) + 1) // 2 if real else int(n_rows)), int(max_dim)
        self.ldv = (self.n_rows + 63) // 64 * 64
        self.V = mem.zeros((self.max_dim + 1, self.ldv), mem.c128, device)
        self.H = mem.zeros((self.max_dim + 1, self.max_dim), mem.c128, device)
        self.device = device

    def col(self, j):
        return self.V[j]

    def _pack(self, host_cols_T):
        """(k, n) host rows -> (k, n_rows) complex128 (real-packed: pairs of reals, zero tail)."""
        if not self.real:
            return np.ascontiguousarray(host_cols_T, dtype=C128)
        a = np.asarray(host_cols_T)
        assert not np.iscomplexobj(a) or not a.imag.any(), "real-packed basis takes real vectors"
        buf = np.zeros((a.shape[0], 2 * self.n_rows), np.float64)
        buf[:, : self.n_real] = a.real
        return buf.view(C128)

    def set_col(self, j, host_vec):
        v = mem.host(self._pack(np.asarray(host_vec).reshape(1, -1))[0])
        self.V[j, : self.n_rows].copy_(v)

    def get_cols(self, j0, j1):
        """Host copy, shape (n, j1-j0), Fortran order (like the reference's V views); float64 for a
        real-packed basis."""
        out = self.V[j0:j1, : self.n_rows].cpu().numpy()  # (cols, n) C-order == (n, cols) F-order
        if self.real:
            out = np.ascontiguousarray(out).view(np.float64)[:, : self.n_real]
        return out.T

    def set_cols(self, j0, host_cols):
        a = self._pack(np.asarray(host_cols).T)
        self.V[j0: j0 + a.shape[0], : self.n_rows].copy_(mem.host(np.ascontiguousarray(a)))

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


def truncate(basis, m, p, Qp_dev, ws=None, col0=0):
    """``V[:, col0:col0+p] = V[:, col0:col0+m] @ Qp ; V[:, col0+p] = V[:, col0+m]`` in place.  With ``ws``: through
    ``aks_truncate_ws``, which only BOOKS the scales of raw columns (clears them; column ``col0+p`` inherits the scale
    of column ``col0+m``): the kernel itself is ``aks_truncate``'s and divides nothing -- for columns left raw by an
    expansion that deferred its normalisations the CALLER must have divided the matching rows of ``Qp`` by those
    columns' scales (``ArnoldiContext._fold_scales``; include/arnoldi_hip.h says the same)."""
    first = basis.V.data_ptr() + 16 * basis.ldv * col0
    if ws is None:
        rc = _hip.load().aks_truncate(basis.n_rows, m, p, first, basis.ldv, _ptr(Qp_dev), _stream())
        _hip.check(rc, "aks_truncate")
    else:
        rc = _hip.load().aks_truncate_ws(basis.n_rows, m, p, first, basis.ldv, _ptr(Qp_dev), col0, _ptr(ws.buf),
                                         ws.nbytes, ws.max_dim, _stream())
        _hip.check(rc, "aks_truncate_ws")


def gather_c128(count, idx, src, dst):
    rc = _hip.load().aks_gather_c128(count, _ptr(idx), _ptr(src), _ptr(dst), _stream())
    _hip.check(rc, "aks_gather_c128")


def gather_f64(count, idx, src, dst):
    rc = _hip.load().aks_gather_f64(count, _ptr(idx), _ptr(src), _ptr(dst), _stream())
    _hip.check(rc, "aks_gather_f64")


def combine(n_rows, m, V, ldv, S_dev, out, ldo):
    """out[:, :q] = V[:, :m] @ S on the device (S_dev: (m, q) complex128 tensor), out of place."""
    q = int(S_dev.shape[1])
    rc = _hip.load().aks_combine(n_rows, m, q, _ptr(V), ldv, _ptr(S_dev), _ptr(out), ldo, _stream())
    _hip.check(rc, "aks_combine")


def scale(n_rows, w, alpha):
    alpha = complex(alpha)
    rc = _hip.load().aks_scale(n_rows, _ptr(w), alpha.real, alpha.imag, _stream())
    _hip.check(rc, "aks_scale")


class DeviceColumns:
    """``n_cols`` complex128 vectors of ``n_rows`` entries in HBM, column-major with the basis'
    padded leading dimension (Ritz vectors, eigenvectors, scratch columns)."""

    def __init__(self, n_rows, n_cols, device=None):
        device = _require_gpu(device)
        self.n_rows, self.n_cols = int(n_rows), int(n_cols)
        self.ldv = (self.n_rows + 63) // 64 * 64
        self.V = mem.zeros((self.n_cols, self.ldv), mem.c128, device)
        self.device = device

    def col(self, j):
        return self.V[j]

    def set_cols(self, j0, host_cols):
        a = np.ascontiguousarray(np.asarray(host_cols, dtype=C128).T)
        self.V[j0: j0 + a.shape[0], : self.n_rows].copy_(mem.host(a))

    def get_cols(self, j0=0, j1=None):
        j1 = self.n_cols if j1 is None else j1
        return self.V[j0:j1, : self.n_rows].cpu().numpy().T


def combine_columns(cols, j0, m, S, out=None):
    """``cols[:, j0:j0+m] @ S`` on the device (``aks_combine``) -> DeviceColumns with S.shape[1]
    columns.  ``cols`` is a ``KrylovBasis`` or a ``DeviceColumns``; S is a host (m, q) array."""
    S = np.ascontiguousarray(np.asarray(S, dtype=C128).reshape(m, -1))
    q = S.shape[1]
    if out is None:
        out = DeviceColumns(cols.n_rows, q, cols.device)
    for c0 in range(0, q, 64):                        # column chunks keep S within the kernel's LDS budget
        Sd = mem.upload(np.ascontiguousarray(S[:, c0: c0 + 64]), cols.device)
        combine(cols.n_rows, m, cols.V.data_ptr() + 16 * cols.ldv * j0, cols.ldv, Sd,
                out.V.data_ptr() + 16 * out.ldv * c0, out.ldv)
    return out


def fetch_H_and_ctrl(basis, ws):
    """Queue the copy of the device H and the 64-byte control block to pinned host memory on the current
    stream and return a function that waits for exactly that copy (an event, not the whole stream) and
    gives ``(H_host, ctrl)``.  Kernels queued after this call keep running while the host works on H."""
    if not basis.V.is_cuda:                                   # CPU tensors (tests/fake_hip.py)
        return lambda: (basis.download_H(), ws.read_ctrl())
    if getattr(basis, "_H_pinned", None) is None:
        basis._H_pinned = mem.pinned_empty(tuple(basis.H.shape), basis.H.dtype)
        basis._ctrl_pinned = mem.pinned_empty(64, mem.u8)
    basis._H_pinned.copy_(basis.H, non_blocking=True)
    basis._ctrl_pinned.copy_(ws.buf[:64], non_blocking=True)
    ev = mem.Event()
    ev.record()

    def wait():
        ev.synchronize()
        return basis._H_pinned.numpy(), _hip.Ctrl.from_buffer_copy(basis._ctrl_pinned.numpy().tobytes())

    return wait
