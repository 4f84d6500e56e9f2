// Streaming ceiling of a READ + WRITE mix on this box (gfx950): a kernel that does nothing but read R column streams
// and write W column streams of 16 bytes per lane (non-temporal, one row per lane per trip, grid-stride) -- what
// the restart compression (R = m + 1, W = p + 1) and phase 1 of the binned SpMV could reach if everything but
// their memory traffic were free.
//   hipcc --offload-arch=gfx950 -O3 -o profiles/microbench/rw_mix_ceiling profiles/microbench/rw_mix_ceiling.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));

template <int R, int W, int U>
__global__ __launch_bounds__(256) void k_mix(int64_t n, const v2d *__restrict__ in, v2d *__restrict__ out, int64_t ld) {
    const int64_t stride = (int64_t)gridDim.x * 256 * U;
    for (int64_t i0 = (int64_t)blockIdx.x * 256 * U + threadIdx.x; i0 < n; i0 += stride) {
        v2d acc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] = (v2d){0.0, 0.0};
#pragma unroll
        for (int c = 0; c < R; ++c)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t i = i0 + u * 256;
                if (i < n) acc[u] += __builtin_nontemporal_load(&in[i + c * ld]);
            }
#pragma unroll
        for (int c = 0; c < W; ++c)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t i = i0 + u * 256;
                if (i < n) __builtin_nontemporal_store(acc[u] * (double)(c + 1), &out[i + c * ld]);
            }
    }
}

template <int R, int W, int U> void run(int64_t n, const v2d *in, v2d *out, int64_t ld, int grid) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k_mix<R, W, U>), dim3(grid), dim3(256), 0, 0, n, in, out, ld);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k_mix<R, W, U>), dim3(grid), dim3(256), 0, 0, n, in, out, ld);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    const double bytes = 16.0 * n * (R + W);
    printf("  read %2d + write %2d streams, %d rows per lane per trip, grid %5d: %.4f ms  %.2f TB/s (%.3f of 8 TB/s)\n", R, W, U, grid, ms,
           bytes / ms / 1e9, bytes / ms / 8e9);
}

int main() {
    const int64_t n = 10000000, ld = 10000064;
    v2d *in, *out;
    CK(hipMalloc(&in, 22 * ld * 16)); CK(hipMalloc(&out, 17 * ld * 16));
    CK(hipMemset(in, 0, 22 * ld * 16));
    printf("n = %lld rows of 16 bytes per stream\n", (long long)n);
    for (int grid : {1024, 4096}) {
        run<21, 0, 2>(n, in, out, ld, grid);
        run<21, 11, 1>(n, in, out, ld, grid);
        run<21, 11, 2>(n, in, out, ld, grid);
        run<21, 11, 4>(n, in, out, ld, grid);
        run<10, 16, 2>(n, in, out, ld, grid);
        run<1, 1, 4>(n, in, out, ld, grid);
        run<0, 11, 2>(n, in, out, ld, grid);
    }
    return 0;
}
