// Does the 256 MiB Infinity Cache serve the second pass over a Krylov basis that does not fit it, when the second pass
// walks the rows in the OPPOSITE order?  (8-GPU shard sizes: 1.25M rows x 21 columns x 16 B = 420 MB per pass.)
// Two streaming passes over one buffer, the way k_proj and k_update_proj read the panel: forward / forward against
// forward / backward, plain loads against non-temporal loads, for a range of buffer sizes.
//   hipcc -O3 --offload-arch=gfx950 -o mall_reuse mall_reuse.hip && ./mall_reuse
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int BLOCK = 256, UNROLL = 8;                 // 256 threads x 8 x 16 B = 32 KiB per workgroup

template <bool NT>
__global__ __launch_bounds__(BLOCK) void k_pass(const double2 *__restrict__ p, long n_tiles, double *__restrict__ out, int reverse) {
    const long b = reverse ? n_tiles - 1 - blockIdx.x : blockIdx.x;
    const double2 *q = p + b * (long)(BLOCK * UNROLL) + threadIdx.x;
    double2 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        if (NT) {
            v[u].x = __builtin_nontemporal_load(&q[u * BLOCK].x);
            v[u].y = __builtin_nontemporal_load(&q[u * BLOCK].y);
        } else
            v[u] = q[u * BLOCK];
    }
    double s = 0;
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) s += v[u].x + v[u].y;
    if (s == 12345.678) out[blockIdx.x] = s;          // (never: keeps the loads alive)
}

template <bool NT>
float run(const double2 *buf, long n_tiles, double *out, int reverse_second, int reps, hipStream_t st) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) {
        k_pass<NT><<<dim3((unsigned)n_tiles), BLOCK, 0, st>>>(buf, n_tiles, out, 0);
        k_pass<NT><<<dim3((unsigned)n_tiles), BLOCK, 0, st>>>(buf, n_tiles, out, reverse_second);
    }
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r) {
        k_pass<NT><<<dim3((unsigned)n_tiles), BLOCK, 0, st>>>(buf, n_tiles, out, 0);
        k_pass<NT><<<dim3((unsigned)n_tiles), BLOCK, 0, st>>>(buf, n_tiles, out, reverse_second);
    }
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;                                   // one forward pass + one second pass
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    const long sizes_mb[] = {64, 128, 200, 256, 320, 420, 640, 840, 1700, 3400};
    double *out; CK(hipMalloc(&out, 1 << 24));
    printf("# two passes over one buffer (GB/s over both passes; 'rev' = the second pass walks the tiles backwards)\n");
    printf("# %8s %12s %12s %12s %12s\n", "MB", "plain fwd", "plain rev", "nt fwd", "nt rev");
    for (long mb : sizes_mb) {
        const long bytes = mb * 1000000L, n_tiles = bytes / (BLOCK * UNROLL * 16);
        double2 *buf; CK(hipMalloc(&buf, n_tiles * BLOCK * UNROLL * 16));
        CK(hipMemsetAsync(buf, 0, n_tiles * BLOCK * UNROLL * 16, st));
        const double gb = 2.0 * n_tiles * BLOCK * UNROLL * 16 / 1e9;
        const int reps = mb > 1000 ? 10 : 30;
        float a = run<false>(buf, n_tiles, out, 0, reps, st), b = run<false>(buf, n_tiles, out, 1, reps, st);
        float c = run<true>(buf, n_tiles, out, 0, reps, st), d = run<true>(buf, n_tiles, out, 1, reps, st);
        printf("  %8ld %12.0f %12.0f %12.0f %12.0f\n", mb, gb / a * 1e3, gb / b * 1e3, gb / c * 1e3, gb / d * 1e3);
        fflush(stdout);
        CK(hipFree(buf));
    }
    return 0;
}
