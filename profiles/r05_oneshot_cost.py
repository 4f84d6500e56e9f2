#!/usr/bin/env python3
"""Device-side and host-side cost of ONE one-shot all-reduce (AKS_ALLREDUCE=oneshot) of 42 doubles between N rank PROCESSES
that share this box's GPU (mailboxes mapped with hipIpc*): a batch of back-to-back reductions between two events.

    python profiles/r05_oneshot_cost.py 2 4 6          # rank counts to try (at most 6 processes fit a GPU box)

What it can say: the price of 2 one-workgroup kernels + 1 hipStreamWaitValue64 per reduction when the peers are local
memory.  What it cannot: the xGMI hop of the posts and of the remote atomic adds -- that needs a multi-GPU node."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rank_main():
    sys.path.insert(0, os.path.join(ROOT, "arnoldi-py_amd"))
    import ctypes as C

    import numpy as np
    from arnoldi_amd import _hip, mem
    from arnoldi_amd.dist import HostComm

    comm = HostComm()
    handle = comm.native()
    lib = _hip.load()
    why = C.create_string_buffer(256)
    path = lib.aks_comm_allreduce_path(handle, why, 256)
    buf = mem.upload(np.full(42, float(comm.rank + 1)), mem.as_device(None))
    stream = C.c_void_p(mem.stream_ptr())
    res = {"rank": comm.rank, "path": int(path), "why": why.value.decode()}
    for batch in (1, 10, 200):
        for _ in range(3):                              # warm-up + two timed rounds: keep the last
            comm.barrier()
            ev0, ev1 = mem.Event(enable_timing=True), mem.Event(enable_timing=True)
            ev0.record()
            t0 = time.perf_counter()
            for _ in range(batch):
                _hip.check(lib.aks_comm_allreduce_sum(handle, C.c_void_p(buf.data_ptr()), 42, stream), "allreduce")
            host_us = (time.perf_counter() - t0) / batch * 1e6
            ev1.record()
            mem.synchronize()
            res[f"device_us_per_call_batch{batch}"] = round(ev0.elapsed_time(ev1) * 1e3 / batch, 2)
            res[f"host_enqueue_us_per_call_batch{batch}"] = round(host_us, 2)
    got = np.asarray(buf.cpu().numpy())
    res["finite"] = bool(np.isfinite(got).all())
    comm.barrier()
    print("RESULT " + json.dumps(res), flush=True)
    comm.close()


def main():
    if os.environ.get("AKS_ONESHOT_COST_RANK") == "1":
        return rank_main()
    import socket

    for n in [int(a) for a in sys.argv[1:]] or [2, 4]:
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        env = dict(os.environ, AKS_ONESHOT_COST_RANK="1", WORLD_SIZE=str(n), AKS_RENDEZVOUS=f"127.0.0.1:{port}", AKS_ALLREDUCE="oneshot",
                   AKS_HOST_ALLOC="hip", AKS_COMM="host", AKS_COMM_TIMEOUT_S="60",
                   AKS_LIB_PATH=os.path.join(ROOT, "tests", "mock_rccl", "libarnoldi_hip.so"))
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                                  stderr=subprocess.STDOUT, text=True) for r in range(n)]
        outs = []
        for p in procs:
            try:
                outs.append(p.communicate(timeout=120)[0])
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                print(f"{n} ranks: TIMED OUT")
                outs = None
                break
        if outs is None:
            continue
        for o in outs:
            line = [x for x in o.splitlines() if x.startswith("RESULT ")]
            print(f"{n} ranks:", line[0][7:] if line else o[-400:])


if __name__ == "__main__":
    main()
