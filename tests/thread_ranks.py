"""TEST INFRASTRUCTURE: N ranks of the row-sharded solve as N host THREADS of one process, all on GPU 0.

Why: a GPU box of the development pool has one GPU and admits at most six processes on it, so the eight ranks of
BASELINE configs 4 and 5 cannot be rehearsed as eight processes.  The library keeps no process-global state
(``aks_device_init`` aside), so eight threads -- each with its own HIP stream, its own ``ArnoldiContext`` and its
own communicator handle from ``tests/mock_rccl`` -- drive exactly the C entry points the eight processes of
``bench.py --gpus 8`` drive: ``aks_arnoldi_expand`` on an ``aks_shard`` with the grouped send / recv of the ghost
exchange and the stage all-reduces issued from C.  The stand-in's barriers run inside ctypes calls (GIL released), so
the threads really wait for one another the way ranks do, and its order / peer / size checks apply unchanged.

``ThreadComm`` is the ``arnoldi_amd.dist.Comm`` interface over in-process hand-offs (the set-up exchanges that
torch.distributed carries between processes); ``run_ranks`` starts the threads.  The process must have been started
with AKS_LIB_PATH pointing at tests/mock_rccl/libarnoldi_hip.so (see tests/thread_ranks_worker.py).
"""
import ctypes as C
import threading
import traceback

import numpy as np


class ThreadGroup:
    def __init__(self, size):
        self.size = int(size)
        self.barrier = threading.Barrier(self.size)
        self.slots = [None] * self.size
        self.box = {}
        self.lock = threading.Lock()
        self.rng_lock = threading.Lock()          # numpy's global RNG is per process: ranks draw v0 one at a time


def _graph_owners_base():
    from arnoldi_amd.dist import _GraphOwners          # (graphs before the communicator: the product's own rule)

    return _GraphOwners


class ThreadComm(_graph_owners_base()):
    """One rank's view of a ``ThreadGroup``: same methods as ``arnoldi_amd.dist.Comm``."""

    backend = "threads"

    def __init__(self, group, rank):
        self.g, self.rank, self.size = group, int(rank), group.size
        self.force = False
        self._native = None
        self.group = None

    @property
    def active(self):
        return self.size > 1

    # -- hand-offs -----------------------------------------------------------------------------------------
    def _gather(self, value):
        """Every rank deposits ``value``; returns the list of all deposits in rank order."""
        g = self.g
        g.slots[self.rank] = value
        g.barrier.wait()
        out = list(g.slots)
        g.barrier.wait()
        return out

    def allgather_int64(self, values):
        return self._gather(np.asarray(values, dtype=np.int64))

    def exchange_requests(self, ghost_cols, recv_counts):
        ghost_cols = np.ascontiguousarray(ghost_cols, dtype=np.int64)
        recv_counts = np.asarray(recv_counts, dtype=np.int64)
        everyone = self._gather((ghost_cols, recv_counts))
        out = []
        for cols, counts in everyone:              # what peer r asked of this rank: its ghosts owned by self.rank
            lo = int(counts[: self.rank].sum())
            out.append(cols[lo: lo + int(counts[self.rank])].copy())
        return out

    def allgather_rows(self, local):
        parts = self._gather(np.ascontiguousarray(local))
        return np.concatenate(parts, axis=0)

    def allreduce_sum_(self, t):
        if not self.active:
            return
        parts = self._gather(t.detach().cpu().clone())
        total = parts[0].clone()
        for p in parts[1:]:                        # rank order: the same bits on every rank
            total += p
        t.copy_(total)

    def max_float(self, x):
        return float(max(self._gather(float(x))))

    def barrier(self):
        self.g.barrier.wait()

    def alltoallv_start(self, *a, **k):            # the Python-chained data path is not what these tests are for
        raise NotImplementedError("ThreadComm serves the C-driven path only")

    alltoallv_finish = staticmethod(lambda handle: None)

    # -- the library's own communicator (tests/mock_rccl) ----------------------------------------------------
    def native(self):
        if self._native is not None:
            return self._native
        from arnoldi_amd import _hip

        lib = _hip.load()
        ident = (C.c_char * _hip.COMM_ID_BYTES)()
        if self.rank == 0:
            _hip.check(lib.aks_comm_unique_id(C.cast(ident, C.c_void_p)), "aks_comm_unique_id")
        ids = self._gather(bytes(ident) if self.rank == 0 else None)
        ident = (C.c_char * _hip.COMM_ID_BYTES).from_buffer_copy(ids[0])
        handle = C.c_void_p()
        _hip.check(lib.aks_comm_create(C.cast(ident, C.c_void_p), self.rank, self.size, C.byref(handle)), "aks_comm_create")
        self._native = handle
        return handle

    def close(self):
        self._destroy_native()


def run_ranks(size, fn, timeout=900):
    """``fn(comm, rank)`` on ``size`` threads, each inside its own torch HIP stream on GPU 0.  Returns the list of
    results in rank order; the first exception of any rank is re-raised (the others' barriers are broken so that
    they end too)."""
    import torch

    group = ThreadGroup(size)
    results, errors = [None] * size, [None] * size

    gpu = torch.cuda.is_available()            # (without one only the hand-offs themselves can be exercised)

    def body(rank):
        import contextlib

        try:
            if gpu:
                torch.cuda.set_device(0)
            with (torch.cuda.stream(torch.cuda.Stream()) if gpu else contextlib.nullcontext()):
                comm = ThreadComm(group, rank)
                try:
                    results[rank] = fn(comm, rank)
                    if gpu:
                        torch.cuda.synchronize()
                finally:
                    comm.close()
        except BaseException:                                  # noqa: BLE001
            errors[rank] = traceback.format_exc()
            group.barrier.abort()

    threads = [threading.Thread(target=body, args=(r,), name=f"rank{r}") for r in range(size)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout)
    alive = [t.name for t in threads if t.is_alive()]
    if alive:
        raise RuntimeError(f"thread ranks still running after {timeout} s: {alive}")
    real = [e for e in errors if e is not None and "BrokenBarrierError" not in e]
    if real or any(errors):
        raise RuntimeError("a thread rank failed:\n" + (real[0] if real else next(e for e in errors if e)))
    return results
