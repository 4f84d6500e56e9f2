"""A sharded SpMV WITH its ghost exchange captured into a hipGraph and replayed, in a process WITHOUT torch -- i.e. on the
system's ROCm runtime (HIP 7.2 + RCCL 2.27 in this image), where the capture that crashes torch's bundled HIP 7.0 / RCCL 2.26
(profiles/r05_capture_crash.txt) goes through.  ``aks_shard_apply`` on a one-rank communicator: this rank "sends" k packed
entries to itself -- grouped ncclSend / ncclRecv on the communicator's SIDE stream, forked from and joined to the capturing
stream by events -- while the diagonal block runs; expected  y = D x + O x[send_idx].

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
    json.dump(res, open(out_path, "w"))


if __name__ == "__main__":
    main(sys.argv[1])
