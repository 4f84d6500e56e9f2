"""Row sharding across the GPUs of one node (one process per GPU, RCCL over xGMI).

The reference has no distributed code; this is the layout SURVEY 8(e) fixes:

  * rows of V, w and of the CSR matrix are split into contiguous blocks, one per
    rank; H, Qp and the whole host dense step are replicated;
  * the only data-path collectives are (1) a sum all-reduce of the (J+1) complex
    numbers [V^H w ; ||w||^2] after each Gram-Schmidt stage and (2) an exchange
    of the x entries other shards need before each SpMV (all-to-all of packed
    "ghost" entries; the local (diagonal-block) SpMV runs while it is in flight).

Everything in this file is host logic (numpy planning + set-up exchanges over torch.distributed);
it never touches matrix values on the CPU after the plan is built.  On the data path of an RCCL
group the collectives are issued by libarnoldi_hip.so itself (``Comm.native()`` hands it a
communicator, ``aks_shard_apply`` / ``aks_arnoldi_expand`` do the rest); the torch.distributed
data-path calls below serve gloo groups (CPU tests, several test ranks on one GPU) and
AKS_DIST_PATH=python.
"""
from __future__ import annotations

import atexit

import importlib

import numpy as np
import scipy.sparse as sp


class _Lazy:
    """torch / torch.distributed, imported on first use: the partition helpers below are numpy only, and a process
    that never makes a ``Comm`` (AKS_HOST_ALLOC=hip: one GPU, raw HIP allocations) never loads torch."""

    def __init__(self, name):
        self._name, self._mod = name, None

    def __getattr__(self, attr):
        if self._mod is None:
            self._mod = importlib.import_module(self._name)
        return getattr(self._mod, attr)


torch = _Lazy("torch")
dist = _Lazy("torch.distributed")


# --------------------------------------------------------------------------- partition
def row_offsets(n, world, indptr=None):
    """Contiguous row blocks: offsets[r]..offsets[r+1].  Balanced by non-zeros when the
    full ``indptr`` is known, else by rows."""
    if indptr is None:
        base, extra = divmod(n, world)
        sizes = np.full(world, base, np.int64)
        sizes[:extra] += 1
        return np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    indptr = np.asarray(indptr, dtype=np.int64)
    nnz = indptr[-1]
    # weight = nnz + rows (rows also cost: V panel traffic dominates)
    w = indptr + np.arange(n + 1, dtype=np.int64) * max(1, int(nnz // max(n, 1)) * 4)
    targets = w[-1] * np.arange(1, world, dtype=np.float64) / world
    cuts = np.searchsorted(w, targets, side="left")
    offs = np.concatenate([[0], cuts, [n]]).astype(np.int64)
    return np.maximum.accumulate(offs)


def slab_offsets(dims, world):
    """Row offsets of a grid operator (x fastest) cut along its LAST dimension into ``world`` slabs of whole planes
    (lines in 2-D), as even as the plane count allows -- SURVEY 8(e): "Laplace: z-slabs".  A rank then exchanges
    exactly one plane with each neighbour.  Falls back to even rows when there are fewer planes than ranks."""
    dims = tuple(int(d) for d in dims)
    plane = int(np.prod(dims[:-1])) if len(dims) > 1 else 1
    n = plane * dims[-1]
    if dims[-1] < world:
        return row_offsets(n, world)
    return row_offsets(dims[-1], world) * plane


class GhostPlan:
    """What one rank needs from the others for  y = A_local x.

    ``diag``   CSR block with local column ids (0 .. n_local-1)
    ``off``    CSR block whose column ids index the ghost buffer (0 .. n_ghost-1), or None
    ``ghost_cols``   global column id of each ghost entry, sorted (=> grouped by owner)
    ``recv_counts``  ghosts owned by each rank
    """

    def __init__(self, diag, off, ghost_cols, recv_counts):
        self.diag, self.off = diag, off
        self.ghost_cols, self.recv_counts = ghost_cols, recv_counts
        self.n_ghost = int(ghost_cols.shape[0])


def split_local_rows(A_rows, offsets, rank):
    """Split this rank's row block (global column ids) into diagonal / off-diagonal CSR."""
    A_rows = sp.csr_matrix(A_rows)
    r0, r1 = int(offsets[rank]), int(offsets[rank + 1])
    assert A_rows.shape[0] == r1 - r0
    cols = A_rows.indices.astype(np.int64)
    local = (cols >= r0) & (cols < r1)
    row_of = np.repeat(np.arange(A_rows.shape[0], dtype=np.int64), np.diff(A_rows.indptr))

    def build(mask, new_cols, ncols):
        counts = np.bincount(row_of[mask], minlength=A_rows.shape[0])
        indptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
        return sp.csr_matrix((A_rows.data[mask], new_cols.astype(np.int32), indptr),
                             shape=(A_rows.shape[0], max(int(ncols), 1)))

    diag = build(local, cols[local] - r0, r1 - r0)
    off_cols = cols[~local]
    n_cols = int(A_rows.shape[1])
    if off_cols.size and n_cols <= (1 << 28):
        # the distinct remote columns and every entry's position among them through a table over the columns: two passes
        # of O(n + nnz_off) with sequential or table-sized random access.  (Sorting 12.5M columns and binary-searching
        # them in 4.6M distinct ones -- np.unique + np.searchsorted -- took 3.2 s of a 4.2 s operator set-up at
        # 2 ranks x 5M rows; this takes 0.1 s.)
        seen = np.zeros(n_cols, dtype=bool)
        seen[off_cols] = True
        ghost_cols = np.flatnonzero(seen).astype(np.int64)              # ascending, as np.unique would give them
        place = np.cumsum(seen, dtype=np.int32)
        place -= 1
        off_new = place[off_cols]
        del seen, place
    else:
        ghost_cols = np.unique(off_cols)
        off_new = np.searchsorted(ghost_cols, off_cols)
    if ghost_cols.size == 0:
        return GhostPlan(diag, None, ghost_cols, np.zeros(len(offsets) - 1, np.int64))
    off = build(~local, off_new, ghost_cols.size)
    owner = np.searchsorted(offsets, ghost_cols, side="right") - 1
    recv_counts = np.bincount(owner, minlength=len(offsets) - 1).astype(np.int64)
    return GhostPlan(diag, off, ghost_cols, recv_counts)


# --------------------------------------------------------------------------- communicator
_live_comms = set()          # Comms that own an RCCL communicator (strong references: a communicator is never left to the
                             # garbage collector): destroyed by close() or at interpreter exit
_default_comms = {}          # process group -> Comm, so that repeated solves share one communicator


def _close_all():
    for c in list(_live_comms):
        try:
            c.close()
        except Exception:
            pass


atexit.register(_close_all)


def comm_for(group=None):
    """The Comm of a process group (the default group if None), created once per group: every
    ``partial_schur`` without an explicit ``comm`` used to build a new one -- and with it a new RCCL
    communicator (bootstrap, device buffers) that nothing destroyed.  A group that has been destroyed and
    initialised again is a different group: its Comm is made afresh (and the old one closed)."""
    pg = group if group is not None else dist.distributed_c10d._get_default_group()
    entry = _default_comms.get("default" if group is None else id(group))
    if entry is not None and entry[0] is pg:
        return entry[1]
    if entry is not None:
        entry[1].close()
    c = Comm(group)
    _default_comms["default" if group is None else id(group)] = (pg, c)
    return c


class Comm:
    """Thin wrapper over a torch.distributed process group.

    ``backend == "nccl"`` is RCCL on ROCm: device tensors go straight to the
    collectives.  With ``gloo`` (CPU tests, or several test ranks sharing one GPU)
    device tensors are staged through host memory.
    """

    def __init__(self, group=None, force=False):
        """``force``: issue the collectives even in a one-rank group (lets a single GPU exercise
        the RCCL calls of the multi-rank path)."""
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.force = bool(force)
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self._native = None

    @property
    def active(self):
        """True when collectives have to be issued (more than one rank, or forced)."""
        return self.size > 1 or self.force

    def native(self):
        """``aks_comm`` handle (RCCL communicator owned by libarnoldi_hip.so) for this group, or None when the
        group does not run over RCCL (gloo: CPU tests, several test ranks on one GPU) or AKS_DIST_PATH=python
        asks for the stage chaining in Python.  With it the whole expansion -- ghost exchange and the reductions
        between the Gram-Schmidt stages included -- is one C call per rank (``aks_arnoldi_expand``);
        torch.distributed is then only the out-of-band channel that carries the communicator's id.

        A communicator that cannot be created, or fails its self-test, is an ERROR on every rank
        (``HipLibraryError``), not a quiet change of path: which path ran must not depend on a warning nobody
        reads.  Every rank votes after each step (id drawn, communicator created, self-test passed) BEFORE
        the next collective is entered, so a rank that failed never leaves its peers waiting in one."""
        import os

        path = os.environ.get("AKS_DIST_PATH", "native")
        over_gloo = os.environ.get("AKS_COMM_OVER_GLOO") == "1"     # tests: the id travels over gloo, the ranks
        if path == "python" or (self.backend != "nccl" and not over_gloo):   # share a GPU (tests/mock_rccl)
            return None
        if self._native is not None:
            return self._native
        import ctypes as C

        from . import _hip

        lib = _hip.load()
        dev = torch.device("cuda", torch.cuda.current_device())

        def agree(ok, what, err=None):
            """All ranks learn whether every rank got through ``what``; the first failure raises everywhere."""
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self._wire_device())
            if self.size > 1:
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
            if int(flag.item()) == 0:
                raise _hip.HipLibraryError(
                    f"RCCL communicator of the row-sharded solve: {what} failed on "
                    + (f"this rank ({err})" if err is not None else "another rank")
                    + "; set AKS_DIST_PATH=python to chain the stages through torch.distributed instead")

        ident = (C.c_char * _hip.COMM_ID_BYTES)()
        box, err = [None], None
        if self.rank == 0:
            try:
                _hip.check(lib.aks_comm_unique_id(C.cast(ident, C.c_void_p)), "aks_comm_unique_id")
                box = [bytes(ident)]
            except _hip.HipLibraryError as e:
                err = e
        if self.size > 1:
            dist.broadcast_object_list(box, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0,
                                       group=self.group)
        agree(box[0] is not None, "aks_comm_unique_id", err)
        handle, err = C.c_void_p(), None
        ident = (C.c_char * _hip.COMM_ID_BYTES).from_buffer_copy(box[0])
        try:
            _hip.check(lib.aks_comm_create(C.cast(ident, C.c_void_p), self.rank, self.size, C.byref(handle)),
                       "aks_comm_create")
        except _hip.HipLibraryError as e:
            err = e
        try:
            agree(err is None, "aks_comm_create", err)
            # prove the communicator before relying on it: sum of (rank + 1) over the ranks
            probe = torch.full((2,), float(self.rank + 1), dtype=torch.float64, device=dev)
            try:
                _hip.check(lib.aks_comm_allreduce_sum(handle, C.c_void_p(probe.data_ptr()), 2,
                                                      C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                           "aks_comm_allreduce_sum")
                if abs(float(probe[0].item()) - self.size * (self.size + 1) / 2) > 1e-9:
                    raise _hip.HipLibraryError(f"all-reduce self-test gave {probe.tolist()}")
            except _hip.HipLibraryError as e:
                err = e
            agree(err is None, "the all-reduce self-test", err)
        except _hip.HipLibraryError:
            if handle:
                lib.aks_comm_destroy(handle)
            raise
        self._native = handle
        _live_comms.add(self)
        return self._native

    def close(self):
        if self._native is not None:
            from . import _hip

            _hip.load().aks_comm_destroy(self._native)
            self._native = None
        _live_comms.discard(self)

    def __del__(self):
        # NOT destroyed here: ncclCommDestroy is a collective among the ranks of a node, and the garbage
        # collector runs at a different point on every rank -- possibly in the middle of a hipGraph capture.  The
        # communicator goes with close(), with comm_for()'s replacement of a re-initialised group's Comm, or at exit.
        if getattr(self, "_native", None) is not None:
            import warnings

            warnings.warn("arnoldi_amd.dist.Comm collected with a live RCCL communicator: call close() "
                          "(it stays alive until the interpreter exits)", ResourceWarning, stacklevel=2)

    # -- small host-side exchanges used while building plans -----------------
    def allgather_int64(self, values):
        values = np.asarray(values, dtype=np.int64)
        objs = [None] * self.size
        dist.all_gather_object(objs, values, group=self.group)
        return objs

    def _wire_device(self):
        """Where tensors handed to the collectives must live (nccl: this rank's GPU)."""
        return torch.device("cuda", torch.cuda.current_device()) if self.backend == "nccl" else torch.device("cpu")

    def exchange_requests(self, ghost_cols, recv_counts):
        """Tell every owner which of its entries this rank needs (``ghost_cols`` is sorted, so
        it is already grouped by owner; ``recv_counts[r]`` of them belong to rank r).  Returns,
        per peer, the global ids that peer asked from us.  Two all-to-alls: counts, then ids."""
        dev = self._wire_device()
        want = torch.as_tensor(np.asarray(recv_counts, dtype=np.int64), device=dev)
        asked = torch.empty_like(want)
        dist.all_to_all_single(asked, want, group=self.group)
        send_counts = [int(c) for c in asked.cpu().tolist()]
        ids_out = torch.as_tensor(np.ascontiguousarray(ghost_cols, dtype=np.int64), device=dev)
        ids_in = torch.empty(sum(send_counts), dtype=torch.int64, device=dev)
        dist.all_to_all_single(ids_in, ids_out, send_counts, [int(c) for c in recv_counts], group=self.group)
        ids_in = ids_in.cpu().numpy()
        out, pos = [], 0
        for c in send_counts:
            out.append(ids_in[pos: pos + c])
            pos += c
        return out

    def allgather_rows(self, local):
        """Row blocks of a (n_local, k) host array from every rank, stacked in rank order, on every rank.
        Tensor all-gather of equally padded blocks (no pickling: the final Schur vectors are n x nev
        complex128 -- 2.5 GB at BASELINE config 4)."""
        local = np.ascontiguousarray(local)
        if self.size == 1:
            return local
        rows = [int(v[0]) for v in self.allgather_int64([local.shape[0]])]
        k = int(local.shape[1])
        dev = self._wire_device()
        pad = torch.zeros((max(rows), k), dtype=torch.from_numpy(local[:0]).dtype, device=dev)
        pad[: local.shape[0]].copy_(torch.from_numpy(local))
        parts = [torch.empty_like(pad) for _ in range(self.size)]
        dist.all_gather(parts, pad, group=self.group)
        return np.concatenate([parts[r][: rows[r]].cpu().numpy() for r in range(self.size)], axis=0)

    # -- data-path collectives ---------------------------------------------------
    def allreduce_sum_(self, t):
        """In-place sum all-reduce of a small float64 device tensor (stream ordered on nccl)."""
        if not self.active:
            return
        if self.backend == "nccl" or not t.is_cuda:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        else:
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(h)

    def alltoallv_start(self, send, send_counts, recv, recv_counts, words=2):
        """Start  recv <- all-to-all(send)  on float64 views (``words`` doubles per vector entry: 2 for
        complex128, 1 for real vectors).  Returns a handle for ``alltoallv_finish``."""
        ss = [words * int(c) for c in send_counts]
        rs = [words * int(c) for c in recv_counts]
        if self.backend == "nccl" or not send.is_cuda:
            return dist.all_to_all_single(recv, send, rs, ss, group=self.group, async_op=True)
        hs, hr = send.cpu(), torch.empty(recv.shape, dtype=recv.dtype)
        dist.all_to_all_single(hr, hs, rs, ss, group=self.group)
        recv.copy_(hr)
        return None

    @staticmethod
    def alltoallv_finish(handle):
        if handle is not None:
            handle.wait()

    def max_float(self, x):
        """Largest ``x`` over the ranks (bench.py: the slowest rank's wall time)."""
        if self.size == 1:
            return float(x)
        t = torch.tensor([float(x)], dtype=torch.float64, device=self._wire_device())
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return float(t.item())

    def barrier(self):
        if self.size > 1:
            dist.barrier(group=self.group)
