#!/usr/bin/env python3
"""Round 3's crash, once more, with the faulting frame on record: hipGraph capture of ONE aks_shard_apply whose
ghost exchange is a grouped ncclSend / ncclRecv TO SELF on a one-rank RCCL communicator, forked onto the communicator's
side stream.  Run it under the debugger so that a SIGSEGV leaves a native backtrace:

    rocgdb -batch -ex "set pagination off" -ex run -ex bt -ex "thread apply all bt 12" --args python3 profiles/r05_capture_crash_probe.py

Steps (each announced on stderr before it starts): eager apply; capture with torch's API (relaxed mode); replay; the same
capture in global mode.  faulthandler prints the Python frames of a fatal signal as well."""
import ctypes as C
import faulthandler
import os
import sys

faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "arnoldi-py_amd"))
import numpy as np  # noqa: E402
import scipy.sparse as sp  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def say(msg):
    sys.stderr.write(f"[probe] {msg}\n")
    sys.stderr.flush()


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29731")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from arnoldi_amd import _hip, device as dev
    from arnoldi_amd.dist import Comm

    comm = Comm(force=True)
    say(f"RCCL communicator of one rank: handle {comm.native()}")
    rng = np.random.default_rng(3)
    n, k = 5000, 700
    D = sp.random(n, n, density=2e-3, random_state=np.random.RandomState(1), format="csr")
    O = sp.random(n, k, density=5e-3, random_state=np.random.RandomState(2), format="csr")
    send_idx = np.sort(rng.choice(n, k, replace=False)).astype(np.int32)
    dD, dO = dev.DeviceCSR(D), dev.DeviceCSR(O)
    sh = _hip.Shard()
    dD.block(sh.diag)
    dO.block(sh.off)
    sh.comm = comm.native()
    sh.any_exchange = 1
    counts = (C.c_int64 * 1)(k)
    sh.send_counts, sh.recv_counts = counts, counts
    d_idx = torch.from_numpy(send_idx).cuda()
    sendbuf = torch.zeros(k, dtype=torch.complex128, device="cuda")
    ghost = torch.zeros(k, dtype=torch.complex128, device="cuda")
    sh.d_send_idx, sh.n_send, sh.d_sendbuf = d_idx.data_ptr(), k, sendbuf.data_ptr()
    sh.d_ghostbuf, sh.n_ghost = ghost.data_ptr(), k
    xh = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    x = torch.from_numpy(xh).cuda()
    y = torch.zeros(n, dtype=torch.complex128, device="cuda")
    want = D @ xh + O @ xh[send_idx]

    def apply():
        rc = _hip.load().aks_shard_apply(C.byref(sh), dev._ptr(x), dev._ptr(y), C.c_void_p(0), dev._stream(), 0)
        _hip.check(rc, "aks_shard_apply")

    say("eager apply x 3")
    for _ in range(3):
        apply()
    torch.cuda.synchronize()
    say(f"eager error {float(np.abs(y.cpu().numpy() - want).max()):.2e}")
    for mode in ("relaxed", "thread_local", "global"):
        say(f"capture of one aks_shard_apply, capture_error_mode={mode}: begin")
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g, capture_error_mode=mode):
                apply()
        except Exception as e:                                   # noqa: BLE001
            say(f"capture ({mode}) raised {type(e).__name__}: {str(e)[:300]}")
            continue
        say(f"capture ({mode}) ended; replay x 3")
        y.zero_()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        say(f"replay error {float(np.abs(y.cpu().numpy() - want).max()):.2e}")
    say("done without a fatal signal")
    comm.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
