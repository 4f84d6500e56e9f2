"""Worker of the TORCH-FREE multi-rank tests: one plain process per rank, no launcher (RANK / WORLD_SIZE /
AKS_RENDEZVOUS in the environment), ranks talking through ``arnoldi_amd.dist.HostComm``.

``--case setup``   (anywhere, no GPU) the set-up exchanges themselves: all-gather, ghost requests against a brute-force
                   answer, row gather, all-reduce and all-to-all of the Python-chained path, max, barrier -- over the TCP
                   rendezvous alone (AKS_DIST_PATH=python: no communicator of the library's is created).
``--case solve_chained``  (GPU) the same cases with AKS_DIST_PATH=python: the stages chained from Python, all-reduces and the ghost
                   all-to-all staged through the host and carried by the rendezvous (functional path, no library communicator).
``--case solve``   (GPU) the cases of tests/dist_cases.py on the C-driven path with ``AKS_HOST_ALLOC=hip`` and the library
                   built against tests/mock_rccl (the ranks share GPU 0): communicator id over the rendezvous, ghost
                   requests and Schur-vector rows through ``aks_comm_alltoallv``.
Either way ``torch`` must not be in ``sys.modules`` when the rank is done.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "arnoldi-py_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402


class _HostArray:
    """The two methods HostComm's Python-chained collectives use of a device array, on a NumPy array (no GPU here)."""

    def __init__(self, a):
        self.a = a

    def cpu(self):
        return self.a

    def __getitem__(self, key):
        return _HostArray(self.a[key])

    def copy_(self, src):
        self.a[...] = np.asarray(src.numpy() if hasattr(src, "numpy") else src).reshape(self.a.shape)


def setup_case(comm):
    rank, size = comm.rank, comm.size
    n = 1000
    offs = np.linspace(0, n, size + 1).astype(np.int64)
    rng = np.random.default_rng(0)
    wanted = [np.unique(rng.integers(0, n, 200)) for _ in range(size)]
    wanted = [w[(w < offs[r]) | (w >= offs[r + 1])] for r, w in enumerate(wanted)]       # remote ids only, sorted
    counts = np.bincount(np.searchsorted(offs, wanted[rank], side="right") - 1, minlength=size)
    asked = comm.exchange_requests(wanted[rank], counts)
    for peer in range(size):                                 # peer asked this rank for its ids that this rank owns
        want = wanted[peer][(wanted[peer] >= offs[rank]) & (wanted[peer] < offs[rank + 1])]
        np.testing.assert_array_equal(asked[peer], want)
    ag = comm.allgather_int64([rank, rank * rank])
    assert [list(a) for a in ag] == [[r, r * r] for r in range(size)]
    rows = comm.allgather_rows(np.full((rank + 1, 2), float(rank)) + 0j)
    assert rows.shape == (sum(range(1, size + 1)), 2) and rows[-1, 0] == size - 1 and rows.dtype == np.complex128
    t = _HostArray(np.array([1.0 * rank, 2.0]))
    comm.allreduce_sum_(t)
    np.testing.assert_array_equal(t.a, [sum(range(size)), 2.0 * size])
    # all-to-all of the chained SpMV's ghost exchange: rank r sends (peer + 1) complex entries to every peer
    send_counts = [p_ + 1 for p_ in range(size)]
    recv_counts = [rank + 1] * size
    send = _HostArray(np.concatenate([np.full(2 * (p_ + 1), 100.0 * rank + p_) for p_ in range(size)]))
    recv = _HostArray(np.zeros(2 * sum(recv_counts)))
    comm.alltoallv_finish(comm.alltoallv_start(send, send_counts, recv, recv_counts, words=2))
    np.testing.assert_array_equal(recv.a, np.concatenate([np.full(2 * (rank + 1), 100.0 * p_ + rank) for p_ in range(size)]))
    assert comm.max_float(rank * 1.5) == (size - 1) * 1.5
    comm.barrier()
    assert comm.native() is None                                  # AKS_DIST_PATH=python
    return {"setup": "ok", "size": size}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", choices=["setup", "solve", "solve_chained"], required=True)
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    if args.case == "solve_chained":   # the same cases with the stages chained from Python: every collective through the rendezvous
        os.environ["AKS_DIST_PATH"] = "python"
        os.environ["AKS_HOST_ALLOC"] = "hip"
        os.environ["AKS_COMM"] = "host"
    elif args.case == "solve":         # before arnoldi_amd is imported: _hip / mem read these at import
        os.environ["AKS_LIB_PATH"] = os.path.join(ROOT, "tests", "mock_rccl", "libarnoldi_hip.so")
        os.environ["AKS_HOST_ALLOC"] = "hip"
        os.environ["AKS_COMM"] = "host"
        os.environ["AKS_GRAPH"] = "0"          # the stand-in synchronises streams: nothing to capture
    else:
        os.environ["AKS_DIST_PATH"] = "python"
        os.environ["AKS_HOST_ALLOC"] = "hip"   # (mem's torch backend imports torch; nothing here touches a device)
    from arnoldi_amd.dist import HostComm

    comm = HostComm()
    if args.case == "setup":
        verdict = setup_case(comm)
    else:
        from dist_cases import run_cases

        verdict = run_cases(comm, comm.rank, comm.size)
        if args.case == "solve":
            import ctypes

            from arnoldi_amd import _hip

            why = ctypes.create_string_buffer(256)
            verdict["allreduce_path"] = [int(_hip.load().aks_comm_allreduce_path(comm.native(), why, 256)), why.value.decode()]
    verdict["torch_imported"] = "torch" in sys.modules
    with open(os.path.join(args.out, f"rank{comm.rank}.json"), "w") as f:
        json.dump(verdict, f)
    comm.barrier()
    comm.close()


if __name__ == "__main__":
    main()
