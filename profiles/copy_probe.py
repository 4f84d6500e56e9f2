"""How fast is the look-ahead column's device-to-device copy (engine.ArnoldiContext.expand: V[:, start+1] <- scratch)?
torch's ``copy_`` on complex128 rows against ``hipMemcpyAsync`` through ctypes, n = 10M and 1.25M rows.
    python profiles/copy_probe.py"""
import ctypes as C

import torch

rt = C.CDLL("libamdhip64.so")
rt.hipMemcpyAsync.restype = C.c_int


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for n in (10_000_000, 1_250_000):
    ld = (n + 63) // 64 * 64
    V = torch.zeros((22, ld), dtype=torch.complex128, device="cuda")
    L = torch.zeros((2, ld), dtype=torch.complex128, device="cuda")
    big = torch.zeros(600_000_000 // 8, dtype=torch.float64, device="cuda")       # evicts the caches between copies
    s = torch.cuda.current_stream().cuda_stream
    nbytes = ld * 16

    def t_copy():
        V[11].copy_(L[0])

    def t_copy_f64():
        V[11].view(torch.float64).copy_(L[0].view(torch.float64))

    def t_memcpy():
        assert rt.hipMemcpyAsync(C.c_void_p(V[11].data_ptr()), C.c_void_p(L[0].data_ptr()), C.c_size_t(nbytes), 3, C.c_void_p(s)) == 0

    for name, fn in (("torch copy_ c128", t_copy), ("torch copy_ as f64", t_copy_f64), ("hipMemcpyAsync D2D", t_memcpy)):
        def with_sweep():
            big.add_(1.0)
            fn()
        base = timed(lambda: big.add_(1.0))
        us = timed(with_sweep) - base
        print(f"n = {n:>9}: {name:20s} {timed(fn):8.1f} us back to back, {us:8.1f} us behind a 600 MB sweep ({2 * nbytes / us / 1e6:.2f} TB/s)")
