// Micro-benchmark (gfx950): what do 50M scattered 16-byte accesses cost when their target window is
// L2-resident?  Decides the layout of phase 1 of the slab-binned SpMV (aks_pb_*):
//   gather   lane reads 16 B at a pseudo-random slot of a 1 MiB window (today's phase 1: x slab)
//   sorted   lanes read nearly consecutive slots, 5 lanes per slot     (entries sorted by column)
//   scatter  lane writes 16 B to a pseudo-random slot of a W-byte window; every slot of the window is
//            written exactly once by the blocks that own the window (so L2 can merge full lines)
// hipcc --offload-arch=gfx950 -O3 -o scatter_gather scatter_gather.hip && ./scatter_gather
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef double2 c128;
constexpr int BLOCK = 256, PER = 8, CHUNK = BLOCK * PER;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(BLOCK) void k_gather(const c128 *__restrict__ x, const double *__restrict__ val,
                                                 c128 *__restrict__ out, long chunks_per_xcd, int win_entries_log2,
                                                 long chunks_per_window, int sorted) {
    const long c = (long)(blockIdx.x & 7) * chunks_per_xcd + (blockIdx.x >> 3);
    const long window = c / chunks_per_window;
    const c128 *xs = x + (window << win_entries_log2);
    const unsigned mask = (1u << win_entries_log2) - 1;
    c128 xv[PER];
    double a[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const long k = c * CHUNK + q * BLOCK + threadIdx.x;
        a[q] = val[k];
        const unsigned e = (unsigned)(k % ((long)chunks_per_window * CHUNK));
        const unsigned slot = sorted ? (e / 5) & mask : (e * 40503u) & mask;
        xv[q] = xs[slot];
    }
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const long k = c * CHUNK + q * BLOCK + threadIdx.x;
        out[k] = make_double2(a[q] * xv[q].x, a[q] * xv[q].y);   // coalesced store
    }
}

__global__ __launch_bounds__(BLOCK) void k_scatter(const c128 *__restrict__ x, const double *__restrict__ val,
                                                  c128 *__restrict__ out, long chunks_per_xcd, int win_entries_log2,
                                                  long chunks_per_window) {
    const long c = (long)(blockIdx.x & 7) * chunks_per_xcd + (blockIdx.x >> 3);
    const long window = c / chunks_per_window;
    c128 *dst = out + (window << win_entries_log2);
    const unsigned mask = (1u << win_entries_log2) - 1;
    c128 xv[PER];
    double a[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const long k = c * CHUNK + q * BLOCK + threadIdx.x;
        a[q] = val[k];
        xv[q] = x[k / 5];                                          // nearly coalesced read
    }
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const long k = c * CHUNK + q * BLOCK + threadIdx.x;
        const unsigned e = (unsigned)(k & mask);                   // window holds exactly 2^log2 entries
        dst[(e * 40503u) & mask] = make_double2(a[q] * xv[q].x, a[q] * xv[q].y);   // bijective scatter
    }
}

int main(int argc, char **argv) {
    const long nnz = 50331648;  // 48 Mi entries (multiple of every window size used)
    c128 *x, *out;
    double *val;
    CK(hipMalloc(&x, nnz * sizeof(c128)));
    CK(hipMalloc(&out, nnz * sizeof(c128)));
    CK(hipMalloc(&val, nnz * sizeof(double)));
    CK(hipMemset(x, 0, nnz * sizeof(c128)));
    CK(hipMemset(val, 0, nnz * sizeof(double)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const long n_chunks = nnz / CHUNK, cpx = n_chunks / 8;
    const dim3 grid((unsigned)n_chunks);
    auto run = [&](const char *name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        CK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-46s %8.4f ms  (%6.1f G entries/s)\n", name, ms / 10, nnz / (ms / 10) / 1e6);
    };
    for (int wl = 14; wl <= 18; wl += 2) {   // window of 2^wl entries = 256 KiB .. 4 MiB
        const long cpw = ((1L << wl) + CHUNK - 1) / CHUNK;
        char name[96];
        snprintf(name, sizeof name, "gather  random, window %5ld KiB", (16L << wl) >> 10);
        run(name, [&] { hipLaunchKernelGGL(k_gather, grid, dim3(BLOCK), 0, 0, x, val, out, cpx, wl, cpw, 0); });
        snprintf(name, sizeof name, "gather  sorted (5 per slot), window %5ld KiB", (16L << wl) >> 10);
        run(name, [&] { hipLaunchKernelGGL(k_gather, grid, dim3(BLOCK), 0, 0, x, val, out, cpx, wl, cpw, 1); });
        snprintf(name, sizeof name, "scatter once-per-slot, window %5ld KiB", (16L << wl) >> 10);
        run(name, [&] { hipLaunchKernelGGL(k_scatter, grid, dim3(BLOCK), 0, 0, x, val, out, cpx, wl, cpw); });
    }
    return 0;
}
