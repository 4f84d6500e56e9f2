"""One-rank RCCL rehearsal (GPU box): the collectives of the multi-rank path are issued through
torch.distributed's nccl backend (= RCCL) on a one-rank group, with the real HIP kernels, and the
solve is compared with the CPU oracle.  Multi-rank *logic* is covered by the gloo tests; this
checks that the device tensors / views / split sizes we hand to RCCL are accepted and correct.

    python tests/nccl_single_worker.py OUT.json
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "arnoldi-py_amd")):
    sys.path.insert(0, p)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main(out_path):
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    import oracle
    from arnoldi_amd import matrices, partial_schur
    from arnoldi_amd.dist import Comm

    comm = Comm(force=True)
    res = {"backend": comm.backend}

    # all-reduce on a float64 view at an offset inside a byte buffer (the workspace slots)
    buf = torch.zeros(4096, dtype=torch.uint8, device="cuda")
    view = buf[256: 256 + 16 * 21].view(torch.float64)
    view.copy_(torch.arange(42, dtype=torch.float64, device="cuda"))
    comm.allreduce_sum_(view)
    res["allreduce_ok"] = bool(torch.equal(view.cpu(), torch.arange(42, dtype=torch.float64)))

    # uneven all-to-all on float64 views of complex buffers, including empty messages
    send = torch.randn(1000, dtype=torch.complex128, device="cuda")
    recv = torch.zeros(1000, dtype=torch.complex128, device="cuda")
    h = comm.alltoallv_start(send.view(torch.float64)[: 2 * 700], [700], recv.view(torch.float64)[: 2 * 700], [700])
    comm.alltoallv_finish(h)
    torch.cuda.synchronize()
    res["alltoall_ok"] = bool(torch.equal(recv[:700], send[:700]) and float(recv[700:].abs().sum()) == 0.0)
    h = comm.alltoallv_start(send.view(torch.float64)[:0], [0], recv.view(torch.float64)[:0], [0])
    comm.alltoallv_finish(h)
    asked = comm.exchange_requests(np.array([5, 9, 11], np.int64), [3])
    res["requests_ok"] = bool(len(asked) == 1 and asked[0].tolist() == [5, 9, 11])
    res["allgather_ok"] = comm.allgather_int64([7])[0].tolist() == [7]

    # the staged (multi-rank) expansion with an RCCL all-reduce after every Gram-Schmidt stage
    A = matrices.mark(50)
    kw = dict(max_dim=20, stopping_criterion=1e-8)
    np.random.seed(0)
    stats = {}
    Q, T, hist = partial_schur(A, 5, sort_function=oracle.arg_largest_real, comm=comm, stats=stats, **kw)
    np.random.seed(0)
    Qo, To, histo = oracle.krylov_schur(A, 5, sort_function=oracle.arg_largest_real, **kw)
    _, _, rel = oracle.eig_residuals(A, Q, T)
    _, _, rel_o = oracle.eig_residuals(A, Qo, To)
    res["solve"] = {
        "restarts_equal": bool(np.array_equal(hist.restarts, histo.restarts)),
        "eig_err": float(np.abs(np.diag(T) - np.diag(To)).max()),
        "rel": float(rel.max()), "rel_oracle": float(rel_o.max()),
    }
    comm.barrier()
    dist.destroy_process_group()
    json.dump(res, open(out_path, "w"))


if __name__ == "__main__":
    main(sys.argv[1])
