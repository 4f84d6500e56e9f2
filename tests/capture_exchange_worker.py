"""(1) A sharded SpMV WITH its ghost exchange captured into a hipGraph and replayed, in a process WITHOUT torch -- i.e. on the
system's ROCm runtime (HIP 7.2 + RCCL 2.27 in this image), where the capture that crashes torch's bundled HIP 7.0 / RCCL 2.26
(profiles/r05_capture_crash.txt) goes through.  ``aks_shard_apply`` on a one-rank communicator: this rank "sends" k packed
entries to itself -- grouped ncclSend / ncclRecv on the communicator's SIDE stream, forked from and joined to the capturing
stream by events -- while the diagonal block runs; expected  y = D x + O x[send_idx].

(2) (round 6) The same through the ENGINE: whole Krylov-Schur solves whose re-expansions -- ``aks_arnoldi_expand`` with the ghost
exchange of every SpMV and the stage reductions in the sequence, ``ncclAllReduce`` and the one-shot kernel in turn -- are
captured once (AKS_GRAPH=1 AKS_GRAPH_COMM=exchange) and replayed: H after every expansion bit-identical to the eager
solve's.  The operator is built on a hand-made exchange plan (``CsrOperator(exchange_plan=...)``): the columns of the upper
half of the index range are "remote", packed, sent to this same rank and applied through the off-diagonal block.
(3) The order of destruction: ``comm.close()`` drops the contexts' graphs first; a graph the library still counts makes
``aks_comm_destroy`` REFUSE (an error, where ncclCommDestroy would hang).

    AKS_HOST_ALLOC=hip python tests/capture_exchange_worker.py OUT.json"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "arnoldi-py_amd")):
    sys.path.insert(0, p)
assert os.environ.get("AKS_HOST_ALLOC") == "hip"

import numpy as np  # noqa: E402
import scipy.sparse as sp  # noqa: E402


def main(out_path):
    from arnoldi_amd import _hip, device as dev, mem
    from arnoldi_amd.dist import HostComm

    comm = HostComm(rank=0, size=1, force=True)
    handle = comm.native()
    version = C.c_int(0)
    mem._rt().hipRuntimeGetVersion(C.byref(version))
    rng = np.random.default_rng(3)
    n, k = 5000, 700
    D = sp.random(n, n, density=2e-3, random_state=np.random.RandomState(1), format="csr")
    O = sp.random(n, k, density=5e-3, random_state=np.random.RandomState(2), format="csr")
    send_idx = np.sort(rng.choice(n, k, replace=False)).astype(np.int32)
    device = mem.as_device(None)
    dD, dO = dev.DeviceCSR(D), dev.DeviceCSR(O)
    sh = _hip.Shard()
    dD.block(sh.diag)
    dO.block(sh.off)
    sh.comm = handle
    sh.any_exchange = 1
    counts = (C.c_int64 * 1)(k)
    sh.send_counts, sh.recv_counts = counts, counts
    d_idx = mem.upload(send_idx, device)
    sendbuf, ghost = mem.zeros(k, mem.c128, device), mem.zeros(k, mem.c128, device)
    sh.d_send_idx, sh.n_send, sh.d_sendbuf = d_idx.data_ptr(), k, sendbuf.data_ptr()
    sh.d_ghostbuf, sh.n_ghost = ghost.data_ptr(), k
    res = {"hip_runtime_version": int(version.value), "cases": []}
    graphs = []
    for trial in range(2):                                   # two input vectors: a replay must read the CURRENT x
        xs = [rng.standard_normal(n) + 1j * rng.standard_normal(n) for _ in range(2)]
        x = mem.upload(np.ascontiguousarray(xs[0]), device)
        y = mem.zeros(n, mem.c128, device)

        def apply():
            rc = _hip.load().aks_shard_apply(C.byref(sh), dev._ptr(x), dev._ptr(y), C.c_void_p(0), dev._stream(), 0)
            _hip.check(rc, "aks_shard_apply")

        apply()
        mem.synchronize()
        eager = float(np.abs(np.asarray(y.cpu().numpy()) - (D @ xs[0] + O @ xs[0][send_idx])).max())
        g = mem.Graph(apply)                                  # hipStreamBeginCapture / EndCapture through ctypes
        graphs.append(g)
        errs = []
        for xv in (xs[1], xs[0], xs[1]):
            x.copy_(mem.host(np.ascontiguousarray(xv)))
            y.zero_()
            g.replay()
            mem.synchronize()
            errs.append(float(np.abs(np.asarray(y.cpu().numpy()) - (D @ xv + O @ xv[send_idx])).max()))
        res["cases"].append({"eager_err": eager, "replay_errs": errs})
    res["torch_imported"] = "torch" in sys.modules
    json.dump(res, open(out_path, "w"))
    print(res, flush=True)
    # ORDER MATTERS: the captured sequences first (hipGraphExecDestroy), then the communicator.  With a graph that holds a
    # captured send / recv group still alive, ncclCommDestroy never returns (RCCL 2.27.7: measured, 60 s time-out twice;
    # AKS_CAPTURE_WORKER_KEEP_GRAPHS=1 shows it).
    if os.environ.get("AKS_CAPTURE_WORKER_KEEP_GRAPHS") != "1":
        import gc

        del g
        graphs.clear()
        gc.collect()
        mem.synchronize()
    comm.close()
    res["communicator_destroyed"] = True
    res["versions"] = _hip.runtime_versions()
    for allreduce in ("nccl", "oneshot"):
        res["engine_" + allreduce] = engine_case(allreduce)
    res["torch_imported"] = "torch" in sys.modules
    json.dump(res, open(out_path, "w"))
    print({k: v for k, v in res.items() if k.startswith("engine_")}, flush=True)


def self_exchange_operator(A, comm):
    """``A`` (n x n) as ONE rank's shard whose columns >= n // 2 count as remote: diagonal block with the local columns,
    off-diagonal block over the ghost buffer, the ghost entries requested from -- this same rank."""
    from arnoldi_amd.dist import GhostPlan
    from arnoldi_amd.engine import CsrOperator

    A = sp.csr_matrix(A)
    n = A.shape[0]
    coo = A.tocoo()
    far = coo.col >= n // 2
    ghost_cols = np.unique(coo.col[far]).astype(np.int64)
    diag = sp.csr_matrix((coo.data[~far], (coo.row[~far], coo.col[~far])), shape=(n, n))
    off = sp.csr_matrix((coo.data[far], (coo.row[far], np.searchsorted(ghost_cols, coo.col[far]))), shape=(n, ghost_cols.size))
    plan = GhostPlan(diag, off, ghost_cols, np.array([ghost_cols.size], np.int64))
    return CsrOperator(local_rows=A, offsets=np.array([0, n], np.int64), comm=comm, exchange_plan=(plan, [ghost_cols.copy()]))


def engine_case(allreduce):
    from arnoldi_amd import _hip, matrices, mem
    from arnoldi_amd.dist import HostComm
    from arnoldi_amd.krylov_schur import KrylovSchurSolver
    from arnoldi_amd.utils import arg_largest_magnitude

    if allreduce == "oneshot":
        os.environ["AKS_ALLREDUCE"] = "oneshot"
    else:
        os.environ.pop("AKS_ALLREDUCE", None)
    A = matrices.random_csr(30_000, 5, 7, planted=(4.0, 3.7, 3.4, 3.1, 2.8, 2.5))
    v0 = np.random.default_rng(11).standard_normal(A.shape[0])
    v0 /= np.linalg.norm(v0)
    out = {}
    runs = {}
    try:
        for mode in ("eager", "replay"):
            os.environ["AKS_GRAPH"] = "1" if mode == "replay" else "0"
            os.environ["AKS_GRAPH_COMM"] = "exchange"
            comm = HostComm(rank=0, size=1, force=True)
            op = self_exchange_operator(A, comm)
            why = C.create_string_buffer(256)
            path = _hip.load().aks_comm_allreduce_path(comm.native(), why, 256)
            s = KrylovSchurSolver(op, 5, 20, 10, 1e-10, arg_largest_magnitude, comm=comm, v0=v0.copy())
            s.start()
            Hs = [s.H.copy()]
            for r in range(6):
                if s.contract(r):
                    break
                s.expand()
                Hs.append(s.H.copy())
            mem.synchronize()
            ctx = s.ctx
            runs[mode] = np.stack(Hs)
            out[mode] = {"native": bool(op.native_comm and op.c_driven), "any_exchange": bool(op.any_exchange), "n_ghost": int(op.n_ghost),
                         "allreduce_path": int(path), "why_not": why.value.decode(), "use_graph": bool(ctx.use_graph),
                         "graphs_captured": len(ctx._graphs), "graphs_on_comm": int(ctx._graphs_on_comm),
                         "graph_capture_failures": int(ctx.graph_capture_failures), "expansions": len(Hs),
                         "collectives_per_step": ctx.collectives_per_step()}
            if mode == "replay":
                # (3) a graph the registry has lost: the library's own count refuses the destruction -- an error, not a hang
                lib = _hip.load()
                lib.aks_comm_graph_retain(comm._native)
                try:
                    comm._destroy_native()
                    out["refused_with_a_counted_graph"] = False
                except _hip.HipLibraryError as e:
                    out["refused_with_a_counted_graph"] = "still alive" in str(e)
                lib.aks_comm_graph_release(comm._native)
                out["graphs_after_refusal"] = len(ctx._graphs)        # (the registry's drop ran before the refusal)
            comm.close()                                             # graphs first, then the communicator
            out[mode]["closed"] = comm._native is None
            del s, op, ctx
    finally:
        for k in ("AKS_GRAPH", "AKS_GRAPH_COMM", "AKS_ALLREDUCE"):
            os.environ.pop(k, None)
    out["bit_identical"] = bool(np.array_equal(runs["eager"], runs["replay"]))
    out["finite"] = bool(np.isfinite(runs["replay"]).all())
    # against the plain one-GPU solve of the same matrix (another summation order in the split SpMV: rounding level)
    os.environ["AKS_GRAPH"] = "0"
    try:
        s1 = KrylovSchurSolver(A, 5, 20, 10, 1e-10, arg_largest_magnitude, v0=v0.copy())
        s1.start()
        H1 = s1.H.copy()
    finally:
        os.environ.pop("AKS_GRAPH", None)
    out["first_expansion_vs_one_gpu"] = float(np.abs(runs["eager"][0] - H1).max() / np.abs(H1).max())
    return out


if __name__ == "__main__":
    main(sys.argv[1])
