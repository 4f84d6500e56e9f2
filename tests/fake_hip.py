"""TEST INFRASTRUCTURE: a NumPy stand-in for the *device* entry points of libarnoldi_hip.so.

The product has no CPU path.  To exercise the host logic above the C ABI on a machine
without a GPU (argument plumbing, the workspace protocol, the row-sharded driver with its
all-reduces and ghost exchange over ``gloo``), the CPU tests swap ``_hip.load()`` for this
object: it takes the same raw pointers the real library takes (here they point into CPU
torch tensors) and performs each entry point's documented effect with NumPy/SciPy.  Pure
host entry points (layout, tile planner, version) are forwarded to the real library.

Never imported by anything under ``arnoldi-py_amd/``.
"""
import ctypes as C

import numpy as np
import scipy.sparse as sp

C128 = np.complex128


def _addr(p):
    if isinstance(p, C.c_void_p):
        return p.value or 0
    return int(p or 0)


def _view(p, dtype, count):
    dtype = np.dtype(dtype)
    if count == 0:
        return np.zeros(0, dtype)
    buf = (C.c_char * (dtype.itemsize * count)).from_address(_addr(p))
    return np.frombuffer(buf, dtype=dtype, count=count)


class FakeHip:
    def __init__(self, real):
        self._real = real
        for name in ("aks_last_error", "aks_abi_version", "aks_workspace_layout", "aks_csr_plan_tiles",
                     "aks_pb_params", "aks_pb_plan_create", "aks_pb_plan_export", "aks_pb_plan_view", "aks_pb_plan_destroy",
                     "aks_sell_plan_size", "aks_sell_plan_fill"):
            setattr(self, name, getattr(real, name))
        self.calls = []

    def aks_device_init(self):
        return 0

    # ---- workspace ---------------------------------------------------------------
    def _ws(self, ws, n_rows, max_dim):
        from arnoldi_amd import _hip

        lay = _hip.WsLayout()
        assert self._real.aks_workspace_layout(n_rows, max_dim, C.byref(lay)) == 0
        base = _addr(ws)
        ctrl_i = _view(base, np.int32, 4)
        ctrl_d = _view(base + 16, np.float64, 2)
        red = lambda off, k: _view(base + off, C128, k)  # noqa: E731
        return lay, ctrl_i, ctrl_d, red(lay.red1_off, lay.red_len), red(lay.red2_off, lay.red_len), red(lay.red3_off, 2)

    def aks_workspace_init(self, ws, nbytes, n_rows, max_dim, stream):
        lay = self._ws(ws, n_rows, max_dim)[0]
        _view(ws, np.uint8, lay.partial_off)[:] = 0
        return 0

    def aks_workspace_set_real(self, ws, flag, stream):
        _view(_addr(ws) + 32, np.int32, 1)[0] = 1 if flag else 0
        return 0

    @staticmethod
    def _is_real_mode(ws):
        """aks_ctrl.real_mode: real-packed panels, the reductions drop the imaginary parts."""
        return bool(_view(_addr(ws) + 32, np.int32, 1)[0])

    def aks_csr_spmv_real(self, n_rows, indptr, indices, values, tiles, n_tiles, lpr, x, y, acc, ws, stream):
        self.calls.append("spmv_real")
        if _addr(ws) and _view(ws, np.int32, 1)[0]:
            return 0
        ip = _view(indptr, np.int32, n_rows + 1)
        nnz = int(ip[-1])
        ix = _view(indices, np.int32, nnz)
        vals = _view(values, np.float64, nnz)
        n_cols = int(ix.max()) + 1 if nnz else 1
        A = sp.csr_matrix((vals, ix, ip), shape=(n_rows, n_cols))
        r = A @ _view(x, np.float64, n_cols)
        yv = _view(y, np.float64, n_rows)
        yv[:] = yv + r if acc else r
        return 0

    def _pb_replay(self, A, x, y, acc, vec_dtype):
        """The two phases of the tile-binned form, replayed with NumPy on the planned arrays: phase 1 walks
        the sub-slab ranges, phase 2 the wave-load descriptors and their (level, row) words."""
        d = A._obj if hasattr(A, "_obj") else A.contents
        nnz, n_rows, n_cols, npad = int(d.nnz), int(d.n_rows), int(d.n_cols), int(d.nnz_pad)
        yv = _view(y, vec_dtype, n_rows)
        out = np.zeros(n_rows, vec_dtype)
        if nnz:
            val = _view(d.d_val, C128 if d.values_complex else np.float64, npad)
            lcol = _view(d.d_lcol, np.uint16, npad).astype(np.int64)
            sb = _view(d.d_slab_begin, np.int32, d.n_slabs).astype(np.int64)
            se = _view(d.d_slab_end, np.int32, d.n_slabs).astype(np.int64)
            assert np.all(sb % 8 == 0) and np.all(sb <= se) and np.all(se[:-1] <= sb[1:]) and se[-1] <= npad
            assert int((se - sb).sum()) == nnz
            xv = _view(x, vec_dtype, n_cols)
            prod = _view(d.d_prod, vec_dtype, npad)                                 # (real: first half of the scratch)
            slab_of = np.repeat(np.arange(d.n_slabs, dtype=np.int64), se - sb)
            k = np.concatenate([np.arange(b, e) for b, e in zip(sb, se)]) if d.n_slabs else np.zeros(0, np.int64)
            prod[k] = val[k] * xv[(slab_of << 13) + lcol[k]]                        # phase 1
            runs = _view(d.d_runs, np.uint32, 4 * int(d.n_runs)).reshape(-1, 4).astype(np.int64)
            rbp = _view(d.d_rb_run_ptr, np.int32, d.n_rowblocks + 1).astype(np.int64)
            lrow = _view(d.d_lrow, np.uint16, int(d.n_lrow)).astype(np.int64)
            assert rbp[0] == 0 and rbp[-1] == len(runs) - 32 and np.all(np.diff(rbp) % 32 == 0)   # + one empty round
            assert int(d.n_lrow) == 64 * len(runs)
            info = runs[:, 3]
            l0, l01, total, levels = info & 127, (info >> 7) & 127, (info >> 14) & 127, (info >> 21) & 15
            assert total.max() <= 64 and int(total.sum()) == nnz and np.all(l0 <= l01) and np.all(l01 <= total)
            assert np.all(levels[total > 0] >= 1) and np.all(total[rbp[-1]:] == 0)
            n_chunks = min(int(d.n_rowblocks), 256)                                  # AKS_PB_CHUNKS: chunk-interleaved order
            rb_at = np.concatenate([np.arange(c, d.n_rowblocks, n_chunks) for c in range(n_chunks)]).astype(np.int64)
            rb_of_run = np.repeat(rb_at, np.diff(rbp))
            rep = np.repeat(np.arange(len(runs)), total)                            # wave-load of every lane
            lane = np.arange(nnz) - np.repeat(np.cumsum(total) - total, total)
            base = np.where(lane < l0[rep], runs[rep, 0], np.where(lane < l01[rep], runs[rep, 1], runs[rep, 2]))
            entry = (base + lane) & 0xFFFFFFFF
            assert np.array_equal(np.sort(entry), k)                                # every product exactly once
            slot = rep % 32
            words = lrow[(rep // 32) * 2048 + (slot // 4) * 256 + lane * 4 + slot % 4]
            assert np.all((words >> 13) < levels[rep])                               # every level gets its barrier
            rows = (rb_of_run[rep] << 13) + (words & 8191)
            assert rows.max() < n_rows
            np.add.at(out, rows, prod[entry])                                       # phase 2
        yv[:] = yv + out if acc else out

    def aks_pb_spmv_real(self, A, x, y, acc, ws, stream):
        self.calls.append("pb_spmv_real")
        if _addr(ws) and _view(ws, np.int32, 1)[0]:
            return 0
        d = A._obj if hasattr(A, "_obj") else A.contents
        assert not d.values_complex
        self._pb_replay(A, x, y, acc, np.float64)
        return 0

    def aks_gather_f64(self, count, idx, src, dst, stream):
        if count == 0:
            return 0
        ix = _view(idx, np.int32, count)
        _view(dst, np.float64, count)[:] = _view(src, np.float64, int(ix.max()) + 1)[ix]
        return 0

    # ---- SpMV ----------------------------------------------------------------------
    def aks_csr_spmv(self, n_rows, indptr, indices, values, cplx, tiles, n_tiles, lpr, x, y, acc, ws, stream):
        self.calls.append("spmv")
        if _addr(ws) and _view(ws, np.int32, 1)[0]:
            return 0
        ip = _view(indptr, np.int32, n_rows + 1)
        nnz = int(ip[-1])
        ix = _view(indices, np.int32, nnz)
        vals = _view(values, C128 if cplx else np.float64, nnz)
        n_cols = int(ix.max()) + 1 if nnz else 1
        A = sp.csr_matrix((vals, ix, ip), shape=(n_rows, n_cols))
        xv = _view(x, C128, n_cols)
        yv = _view(y, C128, n_rows)
        # the tile plan must cover every row exactly once
        t = _view(tiles, np.int32, n_tiles + 1)
        assert t[0] == 0 and t[-1] == n_rows and np.all(np.diff(t) > 0)
        r = A @ xv
        yv[:] = yv + r if acc else r
        return 0

    def _sell_replay(self, A, x, y, acc, vec_dtype):
        """The sliced form replayed with NumPy on the planned arrays: a row is the sum, in slot order, of its
        non-padding slots slice_ptr[s] + k * 64 + lane."""
        d = self._deref(A)
        n_rows, n_cols, nnz, npad, ns = int(d.n_rows), int(d.n_cols), int(d.nnz), int(d.nnz_pad), int(d.n_slices)
        assert ns == -(-n_rows // 64) and npad % 64 == 0
        sp = _view(d.d_slice_ptr, np.int64, ns + 1)
        assert sp[0] == 0 and sp[-1] == npad and np.all(np.diff(sp) % 64 == 0) and np.all(np.diff(sp) >= 0)
        yv = _view(y, vec_dtype, n_rows)
        out = np.zeros(n_rows, vec_dtype)
        if npad:
            col = _view(d.d_col, np.int32, npad).astype(np.int64)
            val = _view(d.d_val, C128 if d.values_complex else np.float64, npad)
            assert int((col >= 0).sum()) == nnz and col.max() < n_cols and np.all(val[col < 0] == 0)
            slot = np.arange(npad)
            sl = np.searchsorted(sp, slot, side="right") - 1
            row = sl * 64 + (slot - sp[sl]) % 64
            live = col >= 0
            assert row[live].max() < n_rows
            xv = _view(x, vec_dtype, n_cols)
            np.add.at(out, row[live], val[live] * xv[col[live]])
        yv[:] = yv + out if acc else out

    def aks_sell_spmv(self, A, x, y, acc, ws, stream):
        self.calls.append("sell_spmv")
        if _addr(ws) and _view(ws, np.int32, 1)[0]:
            return 0
        self._sell_replay(A, x, y, acc, C128)
        return 0

    def aks_sell_spmv_real(self, A, x, y, acc, ws, stream):
        self.calls.append("sell_spmv_real")
        if _addr(ws) and _view(ws, np.int32, 1)[0]:
            return 0
        assert not self._deref(A).values_complex
        self._sell_replay(A, x, y, acc, np.float64)
        return 0

    def aks_pb_spmv(self, A, x, y, acc, ws, stream):
        self.calls.append("pb_spmv")
        if _addr(ws) and _view(ws, np.int32, 1)[0]:
            return 0
        self._pb_replay(A, x, y, acc, C128)
        return 0

    # ---- Gram-Schmidt stages ---------------------------------------------------------
    @staticmethod
    def _panel(V, ldv, J, n):
        return _view(V, C128, ldv * J).reshape(J, ldv)[:, :n]

    def aks_gs_project(self, n, J, V, ldv, w, ws, ws_bytes, max_dim, stream):
        lay, ci, cd, r1, r2, r3 = self._ws(ws, n, max_dim)
        if ci[0]:
            return 0
        P, wv = self._panel(V, ldv, J, n), _view(w, C128, n)
        r1[:J] = P.conj() @ wv
        if self._is_real_mode(ws):
            r1[:J] = r1[:J].real
        r1[J] = np.vdot(wv, wv).real
        r3[0] = 0
        return 0

    def aks_gs_update_project(self, n, J, V, ldv, w, ws, ws_bytes, max_dim, stream):
        lay, ci, cd, r1, r2, r3 = self._ws(ws, n, max_dim)
        if ci[0]:
            return 0
        P, wv = self._panel(V, ldv, J, n), _view(w, C128, n)
        wv -= r1[:J] @ P
        r2[:J] = P.conj() @ wv
        if self._is_real_mode(ws):
            r2[:J] = r2[:J].real
        r2[J] = np.vdot(wv, wv).real
        return 0

    @staticmethod
    def _twice(r1, r2, J, eta):
        return np.sqrt(r2[J].real) < np.sqrt(r1[J].real) * eta

    def aks_gs_update_norm(self, n, J, V, ldv, w, eta, ws, ws_bytes, max_dim, stream):
        lay, ci, cd, r1, r2, r3 = self._ws(ws, n, max_dim)
        if ci[0] or not self._twice(r1, r2, J, eta):
            return 0
        P, wv = self._panel(V, ldv, J, n), _view(w, C128, n)
        wv -= r2[:J] @ P
        r3[0] = np.vdot(wv, wv).real
        return 0

    def aks_gs_finish(self, n, J, w, Hcol, ldh, tol, eta, normalize, ws, ws_bytes, max_dim, stream):
        lay, ci, cd, r1, r2, r3 = self._ws(ws, n, max_dim)
        if ci[0]:
            return 0
        twice = self._twice(r1, r2, J, eta)
        beta = float(np.sqrt(r3[0].real if twice else r2[J].real))
        broke = beta < tol
        H = _view(Hcol, C128, ldh * J + 1)
        H[: ldh * J: ldh] = r1[:J] + (r2[:J] if twice else 0)
        if not broke and normalize:
            H[ldh * J] = beta
        cd[0], cd[1] = np.sqrt(r1[J].real), beta
        ci[2] += 1
        ci[3] += int(twice)
        if broke:
            ci[1], ci[0] = J, 1
        elif normalize:
            wv = _view(w, C128, n)
            wv /= beta
        return 0

    def aks_dgks_gs(self, n, J, V, ldv, w, Hcol, ldh, tol, eta, normalize, ws, ws_bytes, max_dim, stream):
        self.aks_gs_project(n, J, V, ldv, w, ws, ws_bytes, max_dim, stream)
        self.aks_gs_update_project(n, J, V, ldv, w, ws, ws_bytes, max_dim, stream)
        self.aks_gs_update_norm(n, J, V, ldv, w, eta, ws, ws_bytes, max_dim, stream)
        return self.aks_gs_finish(n, J, w, Hcol, ldh, tol, eta, normalize, ws, ws_bytes, max_dim, stream)

    @staticmethod
    def _deref(A):
        return A._obj if hasattr(A, "_obj") else A.contents

    def _apply_block(self, B, x, y, acc, ws, stream, real):
        if B.n_rows <= 0:
            return
        pb = B.pb if bool(B.pb) else None
        if bool(B.sell):
            (self.aks_sell_spmv_real if real else self.aks_sell_spmv)(B.sell, x, y, acc, ws, stream)
        elif real and pb is not None:
            self.aks_pb_spmv_real(pb, x, y, acc, ws, stream)
        elif real:
            self.aks_csr_spmv_real(B.n_rows, B.d_indptr, B.d_indices, B.d_values, B.d_tiles, B.n_tiles,
                                   B.lanes_per_row, x, y, acc, ws, stream)
        elif pb is not None:
            self.aks_pb_spmv(pb, x, y, acc, ws, stream)
        else:
            self.aks_csr_spmv(B.n_rows, B.d_indptr, B.d_indices, B.d_values, B.values_complex, B.d_tiles, B.n_tiles,
                              B.lanes_per_row, x, y, acc, ws, stream)

    def aks_shard_apply(self, A, x, y, ws, stream, flags):
        sh = self._deref(A)
        assert not sh.comm, "the CPU stand-in has no RCCL: multi-rank tests chain the stages over gloo"
        self._apply_block(sh.diag, x, y, 0, ws, stream, bool(flags & 2))
        return 0

    def aks_arnoldi_expand(self, A, V, ldv, H, ldh, start, end, tol, eta, ws, ws_bytes, max_dim, probe, stream, flags):
        sh = self._deref(A)
        first_w_ready, real = bool(flags & 1), bool(flags & 2)
        self.calls.append(("expand_from_w" if first_w_ready else "expand") + ("_real" if real else ""))
        n = int(sh.diag.n_rows)
        n_panel = (n + 1) // 2 if real else n
        assert not real or self._is_real_mode(ws)
        for j in range(start, end):
            x = _addr(V) + 16 * ldv * j
            w = _addr(V) + 16 * ldv * (j + 1)
            if not (first_w_ready and j == start):
                self.aks_shard_apply(A, x, w, ws, stream, flags)
            self.aks_dgks_gs(n_panel, j + 1, V, ldv, w, _addr(H) + 16 * j, ldh, tol, eta, 1, ws, ws_bytes, max_dim,
                             stream)
        return 0

    def aks_shard_apply_col(self, A, V, ldv, col, y, ws, ws_bytes, max_dim, stream, flags):
        return self.aks_shard_apply(A, _addr(V) + 16 * ldv * col, y, ws, stream, flags)     # (the stand-in never defers)

    def aks_truncate_ws(self, n, m, p, V, ldv, Qp, col0, ws, ws_bytes, max_dim, stream):
        return self.aks_truncate(n, m, p, V, ldv, Qp, stream)

    # ---- restart compression ---------------------------------------------------------------
    def aks_truncate(self, n, m, p, V, ldv, Qp, stream):
        self.calls.append("truncate")
        Vv = _view(V, C128, ldv * (m + 1)).reshape(m + 1, ldv)
        Q = _view(Qp, C128, m * p).reshape(m, p)
        new = Q.T @ Vv[:m, :n]
        last = Vv[m, :n].copy()
        Vv[:p, :n] = new
        Vv[p, :n] = last
        return 0

    def aks_combine(self, n, m, q, V, ldv, S, out, ldo, stream):
        self.calls.append("combine")
        Vv = _view(V, C128, ldv * (m - 1) + n)
        cols = np.stack([Vv[c * ldv: c * ldv + n] for c in range(m)])          # (m, n)
        Sm = _view(S, C128, m * q).reshape(m, q)
        Ov = _view(out, C128, ldo * (q - 1) + n)
        new = Sm.T @ cols
        for c in range(q):
            Ov[c * ldo: c * ldo + n] = new[c]
        return 0

    def aks_scale(self, n, w, a_re, a_im, stream):
        _view(w, C128, n)[:] *= complex(a_re, a_im)
        return 0

    def aks_gather_c128(self, count, idx, src, dst, stream):
        if count == 0:
            return 0
        ix = _view(idx, np.int32, count)
        _view(dst, C128, count)[:] = _view(src, C128, int(ix.max()) + 1)[ix]
        return 0

    # ---- probes: nothing to time on the CPU ---------------------------------------------------
    def aks_probe_create(self, cap, out):
        return 0

    def aks_probe_destroy(self, p):
        return 0

    def aks_probe_reset(self, p):
        return 0

    def aks_probe_read(self, p, tag, n_out, ms_out):
        return 0


def install(monkeypatch=None):
    """Route arnoldi_amd's device layer to CPU tensors + FakeHip.  Returns the fake."""
    import torch
    from arnoldi_amd import _hip, device as dev

    import os

    fake = FakeHip(_hip.load())
    if monkeypatch is not None:
        monkeypatch.setenv("AKS_GRAPH", "0")    # no hipGraph capture on CPU tensors
    else:
        os.environ["AKS_GRAPH"] = "0"
    patches = [
        (_hip, "load", lambda: fake),
        (dev, "_require_gpu", lambda device=None: torch.device("cpu")),
        (dev, "_stream", lambda: C.c_void_p(0)),
    ]
    for obj, name, val in patches:
        if monkeypatch is not None:
            monkeypatch.setattr(obj, name, val)
        else:
            setattr(obj, name, val)
    return fake
