"""Row sharding across the GPUs of one node (one process per GPU, RCCL over xGMI).

The reference has no distributed code; this is the layout SURVEY 8(e) fixes:

  * rows of V, w and of the CSR matrix are split into contiguous blocks, one per
    rank; H, Qp and the whole host dense step are replicated;
  * the only data-path collectives are (1) a sum all-reduce of the (J+1) complex
    numbers [V^H w ; ||w||^2] after each Gram-Schmidt stage and (2) an exchange
    of the x entries other shards need before each SpMV (all-to-all of packed
    "ghost" entries; the local (diagonal-block) SpMV runs while it is in flight).

Everything in this file is host logic (numpy planning + set-up exchanges);
it never touches matrix values on the CPU after the plan is built.  On the data path of an RCCL
group the collectives are issued by libarnoldi_hip.so itself (``Comm.native()`` hands it a
communicator, ``aks_shard_apply`` / ``aks_arnoldi_expand`` do the rest).

Two interchangeable rank-to-rank layers carry the SET-UP (communicator id, partition sizes, ghost requests, the
final row gather):

``Comm``       over a torch.distributed process group (nccl = RCCL, or gloo for CPU tests / ranks sharing a GPU); its
               torch.distributed data-path calls serve gloo groups and AKS_DIST_PATH=python.
``HostComm``   torch-free (round 5): a TCP rendezvous among the ranks (MASTER_ADDR / MASTER_PORT / RANK / WORLD_SIZE, or
               ``AKS_RENDEZVOUS=host:port``) carries the 128-byte communicator id and the few-byte control messages;
               the bulk exchanges -- ghost requests, Schur-vector row blocks -- go through the library's own
               communicator (``aks_comm_alltoallv``: grouped ncclSend / ncclRecv).  With ``AKS_HOST_ALLOC=hip`` a
               multi-rank solve then needs numpy + scipy + the ROCm runtime, like the reference (pyproject.toml:9-13).
"""
from __future__ import annotations

import atexit
import importlib
import os
import socket
import struct
import time

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


class _GraphOwners:
    """Contexts whose hipGraphs captured operations of this communicator (engine.ArnoldiContext.adopt...): the graphs must
    be destroyed BEFORE the communicator -- ``ncclCommDestroy`` does not return while a graph holds a captured send / recv
    (profiles/r05_capture_crash.txt section 4).  ``close()`` drops them first; the library counts them as well
    (``aks_comm_graph_retain`` / ``_release``) and REFUSES to destroy a communicator with graphs still counted on it, so a
    graph this registry does not know of is an error message, not a hang."""

    def adopt_graph_owner(self, ctx):
        import weakref

        from . import _hip

        if getattr(self, "_graph_owners", None) is None:
            self._graph_owners = weakref.WeakSet()
        _hip.check(_hip.load().aks_comm_graph_retain(self._native), "aks_comm_graph_retain")
        self._graph_owners.add(ctx)

    def _destroy_native(self):
        """Graphs first, then the communicator; raises (communicator intact) if the library still counts a graph."""
        if self._native is None:
            return
        from . import _hip

        for ctx in list(getattr(self, "_graph_owners", None) or ()):
            ctx.drop_graphs()
        _hip.check(_hip.load().aks_comm_destroy(self._native), "aks_comm_destroy")
        self._native = None


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


class Comm(_GraphOwners):
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
        self._destroy_native()
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


# --------------------------------------------------------------------------- torch-free communicator
def _job_token():
    """32 bytes every rank of one job derives alike and a stranger does not know by accident: ``AKS_COMM_TOKEN`` (set it --
    ``bench.py --gpus N`` draws a random one for its ranks -- when the rendezvous port is reachable by others), else a
    digest of the launcher's coordinates (MASTER_ADDR / MASTER_PORT / WORLD_SIZE / TORCHELASTIC_RUN_ID), which keeps
    apart jobs that meet on one port by mistake."""
    import hashlib

    secret = os.environ.get("AKS_COMM_TOKEN")
    if not secret:
        secret = "|".join(os.environ.get(k, "") for k in ("MASTER_ADDR", "MASTER_PORT", "WORLD_SIZE", "TORCHELASTIC_RUN_ID"))
    return hashlib.sha256(("aks-rendezvous:" + secret).encode()).digest()


class _Hub:
    """Star of TCP connections: rank 0 listens, every other rank connects; ``gather`` / ``alltoall`` of byte strings in
    lock step (every rank makes the same sequence of calls -- they are collectives).  Control plane only: the messages
    are bytes to a few MB; a lost peer is a time-out (``AKS_COMM_TIMEOUT_S``, default 300), never a silent hang.  A peer
    must open with the job's token (``_job_token``) and a free rank; one message is at most ``AKS_COMM_MAX_MSG`` bytes
    (default 512 MiB) and is read in chunks, so an announced length never allocates what has not arrived."""

    CHUNK = 16 << 20

    def __init__(self, rank, size, host, port, timeout, span=1):
        """``span``: how many consecutive ports, from ``port`` on, may carry the rendezvous.  With an address derived from
        the launcher's (MASTER_PORT + 1, which nobody reserved) rank 0 listens on the first port of the span it can bind,
        and the others find it by the handshake: a listener that does not answer the hello with the job's token -- some
        other service that happens to own the port -- is not it, and the next port is tried.  An explicit
        ``AKS_RENDEZVOUS=host:port`` names one port (span 1)."""
        self.rank, self.size, self.timeout = rank, size, timeout
        self.max_msg = int(os.environ.get("AKS_COMM_MAX_MSG", str(512 << 20)))
        self.peers = {}                      # rank 0: peer rank -> socket; others: {0: socket}
        self.port = None
        if size == 1:
            return
        token = _job_token()
        bind_host = host if host not in ("", "localhost") else "127.0.0.1"
        if rank == 0:
            srv, err = None, None
            for candidate in range(port, port + max(int(span), 1)):
                srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                try:
                    srv.bind((bind_host, candidate))
                    self.port = candidate
                    break
                except OSError as e:
                    srv.close()
                    srv, err = None, e
            if srv is None:
                raise RuntimeError(f"rendezvous: rank 0 cannot listen on {host}:{port}" + (f"..{port + span - 1}" if span > 1 else "")
                                   + f" ({err}); choose another port with AKS_RENDEZVOUS=host:port on every rank") from None
            srv.listen(size)
            srv.settimeout(timeout)
            deadline = time.monotonic() + timeout
            try:
                while len(self.peers) < size - 1:
                    srv.settimeout(max(deadline - time.monotonic(), 0.01))
                    conn, _ = srv.accept()
                    try:
                        conn.settimeout(min(timeout, 10.0))         # the hello is 40 bytes: a silent stranger does not hold the job up
                        conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        hello = self._recv_exact(conn, 40)
                        (peer,) = struct.unpack("<q", hello[:8])
                        if hello[8:] != token:
                            raise RuntimeError("rendezvous: a peer presented another job's token")
                        if not 0 < peer < size or peer in self.peers:
                            raise RuntimeError(f"rendezvous: unexpected peer rank {peer}")
                        conn.sendall(b"AKS1" + token[:28])           # the answer a peer waits for: this IS its job's rendezvous
                        conn.settimeout(timeout)
                    except (RuntimeError, OSError) as e:
                        conn.close()                                 # a stranger (or a broken hello) is dropped; the ranks
                        sys_stderr(f"rendezvous: dropped a connection ({e})")   # of THIS job can still arrive
                        continue
                    self.peers[peer] = conn
            except socket.timeout:
                self.close()
                raise RuntimeError(f"rendezvous at {host}:{port}: only {len(self.peers) + 1} of {size} ranks arrived "
                                   f"within {timeout:.0f} s") from None
            except BaseException:
                self.close()
                raise
            finally:
                srv.close()
        else:
            deadline = time.monotonic() + timeout
            conn = None
            while conn is None:
                for candidate in range(port, port + max(int(span), 1)):
                    try:
                        c = socket.create_connection((host, candidate), timeout=min(5.0, timeout))
                    except OSError:
                        continue
                    try:
                        c.settimeout(min(timeout, 10.0))
                        c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        c.sendall(struct.pack("<q", rank) + token)
                        if self._recv_exact(c, 32) == b"AKS1" + token[:28]:
                            c.settimeout(timeout)
                            conn, self.port = c, candidate
                            break
                    except (RuntimeError, OSError):
                        pass                                        # not our listener (or it turned us away): next port
                    c.close()
                if conn is None:
                    if time.monotonic() > deadline:
                        raise RuntimeError(f"rendezvous: rank 0 not reachable at {host}:{port}"
                                           + (f"..{port + span - 1}" if span > 1 else "") + f" within {timeout:.0f} s") from None
                    time.sleep(0.05)
            self.peers[0] = conn

    @classmethod
    def _recv_exact(cls, conn, n):
        """``n`` bytes, received chunk by chunk: memory grows with what has ARRIVED, not with what was announced."""
        if n <= cls.CHUNK:
            buf = bytearray(n)
            view, got = memoryview(buf), 0
            while got < n:
                k = conn.recv_into(view[got:], n - got)
                if k == 0:
                    raise RuntimeError("rendezvous: a peer closed its connection")
                got += k
            return bytes(buf)
        parts, left = [], n
        while left:
            parts.append(cls._recv_exact(conn, min(left, cls.CHUNK)))
            left -= len(parts[-1])
        return b"".join(parts)

    def _send_blobs(self, conn, blobs):
        if sum(len(b) for b in blobs) > self.max_msg:
            raise RuntimeError(f"rendezvous: a message of {sum(len(b) for b in blobs)} bytes exceeds AKS_COMM_MAX_MSG = {self.max_msg} "
                               "(the bulk exchanges go through the library's communicator; raise the limit for AKS_DIST_PATH=python)")
        conn.sendall(struct.pack(f"<q{len(blobs)}q", len(blobs), *(len(b) for b in blobs)))
        for b in blobs:
            if len(b):
                conn.sendall(b)

    def _recv_blobs(self, conn):
        (k,) = struct.unpack("<q", self._recv_exact(conn, 8))
        if not 0 <= k <= max(self.size, 1):
            raise RuntimeError(f"rendezvous: malformed message ({k} parts announced)")
        sizes = struct.unpack(f"<{k}q", self._recv_exact(conn, 8 * k)) if k else ()
        if any(n < 0 for n in sizes):
            raise RuntimeError("rendezvous: malformed message (negative length)")
        if sum(sizes) > self.max_msg:
            raise RuntimeError(f"rendezvous: a peer announced {sum(sizes)} bytes, more than AKS_COMM_MAX_MSG = {self.max_msg}")
        return [self._recv_exact(conn, n) if n else b"" for n in sizes]

    def alltoall(self, blobs):
        """``blobs[r]`` goes to rank r; returns the list of what every rank sent here, in rank order."""
        assert len(blobs) == self.size
        if self.size == 1:
            return [bytes(blobs[0])]
        if self.rank != 0:
            self._send_blobs(self.peers[0], blobs)
            return self._recv_blobs(self.peers[0])
        table = [None] * self.size
        table[0] = [bytes(b) for b in blobs]
        for peer in range(1, self.size):
            table[peer] = self._recv_blobs(self.peers[peer])
            if len(table[peer]) != self.size:
                raise RuntimeError("rendezvous: ranks made different calls")
        for peer in range(1, self.size):
            self._send_blobs(self.peers[peer], [table[src][peer] for src in range(self.size)])
        return [table[src][0] for src in range(self.size)]

    def gather(self, blob):
        """Every rank's ``blob`` on every rank, in rank order (each rank sends its blob ONCE)."""
        if self.size == 1:
            return [bytes(blob)]
        if self.rank != 0:
            self._send_blobs(self.peers[0], [blob])
            return self._recv_blobs(self.peers[0])
        parts = [bytes(blob)]
        for peer in range(1, self.size):
            got = self._recv_blobs(self.peers[peer])
            if len(got) != 1:
                raise RuntimeError("rendezvous: ranks made different calls")
            parts.append(got[0])
        for peer in range(1, self.size):
            self._send_blobs(self.peers[peer], parts)
        return parts

    def set_timeout(self, seconds):
        """How long a collective of this hub waits for its slowest rank (``bench.py`` raises it around its child-process
        legs, whose own time-outs are longer than the default of a set-up exchange); returns the previous value."""
        before, self.timeout = self.timeout, float(seconds)
        for conn in self.peers.values():
            conn.settimeout(self.timeout)
        return before

    def close(self):
        for conn in self.peers.values():
            try:
                conn.close()
            except OSError:
                pass
        self.peers = {}


def sys_stderr(msg):
    import sys

    sys.stderr.write(msg + "\n")


RENDEZVOUS_SPAN = 16


def rendezvous_address():
    """(host, port, span) of the torch-free rendezvous: ``AKS_RENDEZVOUS=host:port`` (that port, span 1), else MASTER_ADDR
    and the ``RENDEZVOUS_SPAN`` ports from MASTER_PORT + 1 on (the launcher's own store listens on MASTER_PORT itself when
    the ranks were started by torch.distributed.run; nobody reserved the ports behind it, so rank 0 takes the first one
    it can bind and the handshake tells the others which)."""
    spec = os.environ.get("AKS_RENDEZVOUS")
    if spec:
        host, _, port = spec.rpartition(":")
        return host or "127.0.0.1", int(port), 1
    return os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ.get("MASTER_PORT", "29400")) + 1, RENDEZVOUS_SPAN


class HostComm(_GraphOwners):
    """The ``Comm`` interface without torch: set-up exchanges over a TCP rendezvous (control) and over the library's own
    communicator (bulk); see the module docstring.  Device staging buffers come from ``mem`` (either backend)."""

    backend = "aks"

    def __init__(self, rank=None, size=None, address=None, force=False):
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        self.size = int(os.environ.get("WORLD_SIZE", "1")) if size is None else int(size)
        if not 0 <= self.rank < self.size:
            raise ValueError(f"HostComm: rank {self.rank} of {self.size}")
        host, port, span = (tuple(address) + (1,))[:3] if address is not None else rendezvous_address()
        self.force, self.group, self._native = bool(force), None, None
        self._hub = _Hub(self.rank, self.size, host, port, float(os.environ.get("AKS_COMM_TIMEOUT_S", "300")), span)
        _live_comms.add(self)

    @property
    def active(self):
        return self.size > 1 or self.force

    # -- control plane -----------------------------------------------------------------------------------
    def allgather_int64(self, values):
        blob = np.ascontiguousarray(values, dtype=np.int64).tobytes()
        return [np.frombuffer(b, dtype=np.int64).copy() for b in self._hub.gather(blob)]

    def max_float(self, x):
        return float(max(struct.unpack("<d", b)[0] for b in self._hub.gather(struct.pack("<d", float(x)))))

    def barrier(self):
        self._hub.gather(b"")

    def _agree(self, ok, what, err=None):
        from . import _hip

        votes = [v == b"y" for v in self._hub.gather(b"y" if ok else b"n")]
        if not all(votes):
            where = f"this rank ({err})" if err is not None else f"rank {votes.index(False)}"
            raise _hip.HipLibraryError(f"RCCL communicator of the row-sharded solve: {what} failed on {where}")

    # -- the library's communicator ------------------------------------------------------------------------
    def native(self):
        """``aks_comm`` handle, created on first use: rank 0 draws the id, the rendezvous hands it round, every rank
        calls ``aks_comm_create`` and proves the communicator with one all-reduce; each step is voted on before the next
        collective is entered (as ``Comm.native``).  None with AKS_DIST_PATH=python."""
        if os.environ.get("AKS_DIST_PATH", "native") == "python":
            return None
        if self._native is not None:
            return self._native
        import ctypes as C

        from . import _hip, mem

        lib = _hip.load()
        ident, err = b"", None
        if self.rank == 0:
            buf = (C.c_char * _hip.COMM_ID_BYTES)()
            try:
                _hip.check(lib.aks_comm_unique_id(C.cast(buf, C.c_void_p)), "aks_comm_unique_id")
                ident = bytes(buf)
            except _hip.HipLibraryError as e:
                err = e
        ident = self._hub.gather(ident)[0]
        self._agree(len(ident) == _hip.COMM_ID_BYTES, "aks_comm_unique_id", err)
        handle, err = C.c_void_p(), None
        buf = (C.c_char * _hip.COMM_ID_BYTES).from_buffer_copy(ident)
        try:
            _hip.check(lib.aks_comm_create(C.cast(buf, C.c_void_p), self.rank, self.size, C.byref(handle)), "aks_comm_create")
        except _hip.HipLibraryError as e:
            err = e
        try:
            self._agree(err is None, "aks_comm_create", err)
            try:
                probe = mem.upload(np.full(2, float(self.rank + 1)), mem.as_device(None))
                _hip.check(lib.aks_comm_allreduce_sum(handle, C.c_void_p(probe.data_ptr()), 2, C.c_void_p(mem.stream_ptr())),
                           "aks_comm_allreduce_sum")
                got = np.asarray(probe.cpu().numpy())
                if abs(float(got[0]) - self.size * (self.size + 1) / 2) > 1e-9:
                    raise _hip.HipLibraryError(f"all-reduce self-test gave {got.tolist()}")
            except _hip.HipLibraryError as e:
                err = e
            self._agree(err is None, "the all-reduce self-test", err)
        except _hip.HipLibraryError:
            if handle:
                lib.aks_comm_destroy(handle)
            raise
        self._native = handle
        return handle

    def _device_alltoallv(self, send, send_off, send_n, recv_n):
        """Bytes of the host array ``send`` (uint8) through ``aks_comm_alltoallv``: rank r gets ``send[send_off[r] :
        send_off[r] + send_n[r]]``; returns the received bytes, peer after peer."""
        import ctypes as C

        from . import _hip, mem

        device = mem.as_device(None)
        recv_off = np.concatenate([[0], np.cumsum(recv_n)]).astype(np.int64)
        d_send = mem.upload(np.ascontiguousarray(send) if send.size else np.zeros(8, np.uint8), device)
        d_recv = mem.empty(max(int(recv_off[-1]), 8), mem.u8, device)
        arr = lambda a: np.ascontiguousarray(a, dtype=np.int64)            # noqa: E731
        so, sn, ro, rn = arr(send_off), arr(send_n), arr(recv_off[:-1]), arr(recv_n)
        p64 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int64))             # noqa: E731
        _hip.check(_hip.load().aks_comm_alltoallv(self._native, C.c_void_p(d_send.data_ptr()), p64(so), p64(sn),
                                                  C.c_void_p(d_recv.data_ptr()), p64(ro), p64(rn), C.c_void_p(mem.stream_ptr())),
                   "aks_comm_alltoallv")
        out = np.asarray(d_recv.cpu().numpy()).reshape(-1)[: int(recv_off[-1])]
        return out, recv_off

    def exchange_requests(self, ghost_cols, recv_counts):
        """As ``Comm.exchange_requests``: per peer, the global ids that peer needs from this rank.  Counts travel over the
        rendezvous; the ids (millions at BASELINE config 5) through the library's communicator, every message led by
        its own length so that no pair of ranks is ever without one (an empty group would leave a rank out of the
        collective)."""
        ghost_cols = np.ascontiguousarray(ghost_cols, dtype=np.int64)
        recv_counts = np.asarray(recv_counts, dtype=np.int64)
        asked = [int(np.frombuffer(b, np.int64)[0])
                 for b in self._hub.alltoall([struct.pack("<q", int(c)) for c in recv_counts])]
        lo = np.concatenate([[0], np.cumsum(recv_counts)]).astype(np.int64)
        if self.size == 1:
            return [ghost_cols[: int(recv_counts[0])].copy()]
        if self.native() is None:
            blobs = self._hub.alltoall([ghost_cols[lo[r]: lo[r + 1]].tobytes() for r in range(self.size)])
            return [np.frombuffer(b, np.int64).copy() for b in blobs]
        parts = []
        for r in range(self.size):                                         # [count, ids...] per peer
            parts.append(np.array([recv_counts[r]], np.int64))
            parts.append(ghost_cols[lo[r]: lo[r + 1]])
        send = np.concatenate(parts).view(np.uint8)
        send_n = 8 * (recv_counts + 1)
        send_off = np.concatenate([[0], np.cumsum(send_n)])[:-1]
        got, off = self._device_alltoallv(send, send_off, send_n, 8 * (np.asarray(asked, np.int64) + 1))
        words = got.view(np.int64)
        out = []
        for r in range(self.size):
            block = words[off[r] // 8: off[r + 1] // 8]
            if int(block[0]) != asked[r]:
                raise RuntimeError(f"exchange_requests: rank {r} announced {asked[r]} ids and sent {int(block[0])}")
            out.append(block[1:].copy())
        return out

    def allgather_rows(self, local):
        """Row blocks of a (n_local, k) host array from every rank, stacked in rank order, on every rank."""
        local = np.ascontiguousarray(local)
        if self.size == 1:
            return local
        k, item = int(local.shape[1]), local.dtype.itemsize
        rows = [int(v[0]) for v in self.allgather_int64([local.shape[0]])]
        if self.native() is None:
            blobs = self._hub.gather(local.tobytes())
            return np.concatenate([np.frombuffer(b, local.dtype).reshape(-1, k) for b in blobs], axis=0)
        mine = local.shape[0] * k * item
        got, _ = self._device_alltoallv(local.reshape(-1).view(np.uint8), np.zeros(self.size, np.int64),
                                        np.full(self.size, mine, np.int64), np.asarray(rows, np.int64) * k * item)
        return got.view(local.dtype).reshape(-1, k)

    # -- data-path collectives of the Python-chained path (AKS_DIST_PATH=python: functional, staged through the host) ----
    def allreduce_sum_(self, t):
        if not self.active:
            return
        from . import mem

        h = t.cpu()
        a = np.ascontiguousarray(h.numpy() if hasattr(h, "numpy") else h)
        parts = [np.frombuffer(b, a.dtype) for b in self._hub.gather(a.tobytes())]
        total = parts[0].copy()
        for p_ in parts[1:]:                                               # rank order: the same bits on every rank
            total += p_
        t.copy_(mem.host(total.reshape(a.shape)))

    def alltoallv_start(self, send, send_counts, recv, recv_counts, words=2):
        from . import mem

        h = send.cpu()
        a = np.ascontiguousarray(h.numpy() if hasattr(h, "numpy") else h).reshape(-1).view(np.float64)
        lo = np.concatenate([[0], np.cumsum([words * int(c) for c in send_counts])]).astype(np.int64)
        blobs = self._hub.alltoall([a[lo[r]: lo[r + 1]].tobytes() for r in range(self.size)])
        got = np.concatenate([np.frombuffer(b, np.float64) for b in blobs]) if blobs else np.zeros(0)
        if got.size != words * int(sum(int(c) for c in recv_counts)):
            raise RuntimeError("alltoallv: received another size than announced")
        if got.size:
            recv[: got.size].copy_(mem.host(got))          # (``recv`` is the float64 view of the ghost buffer, engine.py)
        return None

    @staticmethod
    def alltoallv_finish(handle):
        return None

    def close(self):
        self._destroy_native()          # (raises, with everything intact, while the library still counts a captured graph:
        self._hub.close()               #  destroy that graph, then close() again)
        _live_comms.discard(self)


_host_comm = None


def host_comm_from_env():
    """One ``HostComm`` per process for the ranks an environment describes (RANK / WORLD_SIZE / MASTER_*)."""
    global _host_comm
    if _host_comm is None:
        from . import mem

        if mem.gpu_available() and "LOCAL_RANK" in os.environ:      # one process per GPU: rank r of a node drives GPU LOCAL_RANK
            mem.set_device(int(os.environ["LOCAL_RANK"]) % max(mem.device_count(), 1))
        _host_comm = HostComm()
    return _host_comm
