#!/usr/bin/env python3
"""Which ingredient of round 3's capture crash is needed (after profiles/r05_capture_crash_probe.py showed WHERE it dies: 174 573
nested frames of hip::Stream::EndCapture() under hipStreamEndCapture -- unbounded recursion in the HIP runtime, a stack
overflow).  One variant per process (a crash ends it):

    python profiles/r05_capture_crash_variants.py forkjoin_kernel     # fork to a side stream, a plain kernel there, join: no RCCL
    python profiles/r05_capture_crash_variants.py p2p_same_stream     # grouped ncclSend/ncclRecv to self ON the capturing stream
    python profiles/r05_capture_crash_variants.py p2p_forked          # the same on a forked side stream, joined back (= aks_shard_apply)
    python profiles/r05_capture_crash_variants.py allreduce_forked    # ncclAllReduce on the forked side stream, joined back

RCCL is called through ctypes on the library the process already has (the communicator is torch.distributed's own)."""
import ctypes as C
import faulthandler
import os
import sys

faulthandler.enable()
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main(variant):
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29741")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    rccl = C.CDLL("librccl.so")
    class UniqueId(C.Structure):                      # ncclUniqueId: 128 bytes, passed BY VALUE to ncclCommInitRank
        _fields_ = [("internal", C.c_char * 128)]

    uid = UniqueId()
    comm = C.c_void_p()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    rc = rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0)
    assert rc == 0, rc
    for f in (rccl.ncclSend, rccl.ncclRecv):
        f.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    rccl.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    NCCL_DOUBLE, NCCL_SUM = 8, 0
    a = torch.arange(1400, dtype=torch.float64, device="cuda")
    b = torch.zeros(1400, dtype=torch.float64, device="cuda")
    side = torch.cuda.Stream()

    def p2p(stream):
        assert rccl.ncclGroupStart() == 0
        assert rccl.ncclSend(a.data_ptr(), 1400, NCCL_DOUBLE, 0, comm, stream.cuda_stream) == 0
        assert rccl.ncclRecv(b.data_ptr(), 1400, NCCL_DOUBLE, 0, comm, stream.cuda_stream) == 0
        assert rccl.ncclGroupEnd() == 0

    def body():
        cur = torch.cuda.current_stream()
        if variant == "p2p_same_stream":
            p2p(cur)
            return
        fork, join = torch.cuda.Event(), torch.cuda.Event()
        fork.record(cur)
        side.wait_event(fork)
        if variant == "forkjoin_kernel":
            with torch.cuda.stream(side):
                b.copy_(a)
        elif variant == "p2p_forked":
            p2p(side)
        elif variant == "allreduce_forked":
            assert rccl.ncclAllReduce(a.data_ptr(), b.data_ptr(), 1400, NCCL_DOUBLE, NCCL_SUM, comm, side.cuda_stream) == 0
        join.record(side)
        cur.wait_event(join)

    sys.stderr.write(f"[variant {variant}] eager\n")
    body()
    torch.cuda.synchronize()
    sys.stderr.write(f"[variant {variant}] capture: begin\n")
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="relaxed"):
        body()
    sys.stderr.write(f"[variant {variant}] capture ended; replay\n")
    b.zero_()
    g.replay()
    torch.cuda.synchronize()
    sys.stderr.write(f"[variant {variant}] OK: replayed, b == a: {bool(torch.equal(a, b))}\n")
    rccl.ncclCommDestroy(comm)
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
