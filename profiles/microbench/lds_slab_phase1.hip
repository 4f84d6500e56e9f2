// Micro-benchmark (gfx950): phase 1 of the slab-binned SpMV with the x slab staged in LDS.
// A workgroup copies its slab of C complex128 entries of x into LDS (coalesced), then streams its
// E entries (val f64, lcol u16, dest i32), takes x from LDS and writes val*x to out[dest].  dest is
// made of runs of R consecutive slots at pseudo-random places of the 0.8 GB product array (R is what
// the tile (slab x row block) holds on average: 33 for 65536 x 1024 at 5 nnz/row and n = 10M, 4 for
// 8192 x 1024).  Compares against the L2-gather form that is in the library today.
// hipcc --offload-arch=gfx950 -O3 -o lds_slab_phase1 lds_slab_phase1.hip && ./lds_slab_phase1
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef double2 c128;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void k_fill(long nnz, int R, long n_runs_mask, int shift, int C, int32_t *dest, uint16_t *lcol, double *val) {
    for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (long)gridDim.x * blockDim.x) {
        const long run = k / R;
        const long slot = ((run * 2654435761L) & n_runs_mask) * R + (k % R) + shift;
        dest[k] = (int32_t)slot;
        lcol[k] = (uint16_t)((k * 40503u) % (unsigned)C);
        val[k] = 1.0 + (double)(k & 7);
    }
}

template <int THREADS, int UNROLL>
__global__ __launch_bounds__(THREADS) void k_phase1_lds(int C, int E, const c128 *__restrict__ x,
                                                       const double *__restrict__ val, const uint16_t *__restrict__ lcol,
                                                       const int32_t *__restrict__ dest, c128 *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    c128 *xs = reinterpret_cast<c128 *>(smem);
    const long slab = blockIdx.x;
    const c128 *xg = x + slab * C;
    for (int i = threadIdx.x; i < C; i += THREADS) xs[i] = xg[i];
    __syncthreads();
    const long base = slab * (long)E;
    for (int i0 = 0; i0 < E; i0 += THREADS * UNROLL) {
        double a[UNROLL];
        int c[UNROLL], d[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const long k = base + i0 + u * THREADS + threadIdx.x;
            a[u] = val[k];
            c[u] = lcol[k];
            d[u] = dest[k];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const c128 xv = xs[c[u]];
            out[d[u]] = make_double2(a[u] * xv.x, a[u] * xv.y);
        }
    }
}

// today's form: gather from an L2-resident 1 MiB window, dest runs of R
template <int UNROLL>
__global__ __launch_bounds__(256) void k_phase1_l2(long chunks_per_xcd, long chunks_per_window, const c128 *__restrict__ x,
                                                  const double *__restrict__ val, const uint16_t *__restrict__ lcol,
                                                  const int32_t *__restrict__ dest, c128 *__restrict__ out) {
    const long c = (long)(blockIdx.x & 7) * chunks_per_xcd + (blockIdx.x >> 3);
    const c128 *xs = x + (c / chunks_per_window) * 65536;
    double a[UNROLL];
    int cc[UNROLL], d[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        const long k = c * (256 * UNROLL) + u * 256 + threadIdx.x;
        a[u] = val[k];
        cc[u] = (lcol[k] * 8 + (k & 7)) & 65535;     // spread over the 64K window
        d[u] = dest[k];
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        const c128 xv = xs[cc[u]];
        out[d[u]] = make_double2(a[u] * xv.x, a[u] * xv.y);
    }
}

int main() {
    const int E = 40960;
    const long n_slabs = 1228, nnz = n_slabs * E;            // 50.3M entries
    long n_runs_pow2 = 1;
    c128 *x, *out;
    double *val;
    uint16_t *lcol;
    int32_t *dest;
    CK(hipMalloc(&x, (n_slabs + 8) * 65536L * sizeof(c128) / 8 + (1 << 24)));
    CK(hipMalloc(&out, (nnz + (1 << 26)) * sizeof(c128)));
    CK(hipMalloc(&val, nnz * sizeof(double)));
    CK(hipMalloc(&lcol, nnz * sizeof(uint16_t)));
    CK(hipMalloc(&dest, nnz * sizeof(int32_t)));
    CK(hipMemset(x, 0, (n_slabs + 8) * 65536L * sizeof(c128) / 8 + (1 << 24)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto run = [&](const char *name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        CK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-64s %8.4f ms  (%6.1f G entries/s)\n", name, ms / 10, nnz / (ms / 10) / 1e6);
    };
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_phase1_lds<1024, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 16));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_phase1_lds<512, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 16));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_phase1_lds<1024, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 16));
    for (int R : {1, 2, 4, 8, 32}) {
        for (int shift : {0, 1}) {
            // runs: power-of-two count covering nnz/R (bijective multiplicative hash on that range)
            n_runs_pow2 = 1;
            while (n_runs_pow2 * R < nnz) n_runs_pow2 <<= 1;
            hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, nnz, R, n_runs_pow2 - 1, shift, 8192, dest, lcol, val);
            CK(hipDeviceSynchronize());
            char name[128];
            snprintf(name, sizeof name, "LDS slab 8192 (128 KiB), 1024 thr x8, dest runs of %2d (+%d)", R, shift);
            run(name, [&] { hipLaunchKernelGGL((k_phase1_lds<1024, 8>), dim3(n_slabs), dim3(1024), 8192 * 16, 0, 8192, E, x, val, lcol, dest, out); });
            if (shift == 0) {
                snprintf(name, sizeof name, "LDS slab 8192 (128 KiB),  512 thr x8, dest runs of %2d", R);
                run(name, [&] { hipLaunchKernelGGL((k_phase1_lds<512, 8>), dim3(n_slabs), dim3(512), 8192 * 16, 0, 8192, E, x, val, lcol, dest, out); });
                snprintf(name, sizeof name, "LDS slab 4096 ( 64 KiB), 1024 thr x4, dest runs of %2d", R);
                run(name, [&] { hipLaunchKernelGGL((k_phase1_lds<1024, 4>), dim3(n_slabs * 2), dim3(1024), 4096 * 16, 0, 4096, E / 2, x, val, lcol, dest, out); });
                const long n_chunks = nnz / 2048, cpx = n_chunks / 8;
                snprintf(name, sizeof name, "L2 window 1 MiB (today), 256 thr x8, dest runs of %2d", R);
                run(name, [&] { hipLaunchKernelGGL((k_phase1_l2<8>), dim3(cpx * 8), dim3(256), 0, 0, cpx, 160, x, val, lcol, dest, out); });
            }
        }
    }
    return 0;
}
