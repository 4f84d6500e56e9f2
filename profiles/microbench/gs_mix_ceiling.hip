// Streaming ceiling of the Gram-Schmidt update pattern on this box (gfx950): R panel streams read (non-temporal, 16 B
// per lane), one stream read AND rewritten in place (the vector w), everything else free -- what k_update_proj could
// reach if only its memory traffic counted; beside it the pure read (k_proj's pattern).
//   hipcc --offload-arch=gfx950 -O3 -o profiles/microbench/gs_mix_ceiling profiles/microbench/gs_mix_ceiling.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));

// MODE 0: read only (sum kept, written once per thread at the end)   1: w rewritten in place, plain store
//      2: w rewritten in place, non-temporal store                   3: w written to ANOTHER buffer (plain store)
//      4: w read with a plain load and rewritten in place with a plain store
template <int R, int MODE, int U>
__global__ __launch_bounds__(256) void k_mix(int64_t n, const v2d *__restrict__ V, v2d *w, v2d *other, int64_t ld, v2d *sink) {
    const int64_t stride = (int64_t)gridDim.x * 256 * U;
    v2d tot = (v2d){0.0, 0.0};
    for (int64_t i0 = (int64_t)blockIdx.x * 256 * U + threadIdx.x; i0 < n; i0 += stride) {
        v2d acc[U], wv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + u * 256;
            acc[u] = (v2d){0.0, 0.0};
            wv[u] = i < n ? (MODE == 4 ? w[i] : __builtin_nontemporal_load(&w[i])) : (v2d){0.0, 0.0};
        }
#pragma unroll
        for (int c = 0; c < R; ++c)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t i = i0 + u * 256;
                if (i < n) acc[u] += __builtin_nontemporal_load(&V[i + c * ld]);
            }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + u * 256;
            const v2d r = wv[u] - acc[u] * 1e-3;
            tot += r;
            if (i < n) {
                if (MODE == 1 || MODE == 4) w[i] = r;
                if (MODE == 2) __builtin_nontemporal_store(r, &w[i]);
                if (MODE == 3) other[i] = r;
            }
        }
    }
    if (tot.x == 12345.678) sink[0] = tot;
}

// MODE 5 of the question "why does one written stream in 22 cost 18 % of the time": the results of S consecutive
// trips are parked in LDS (thread-private slots, no barrier) and written back to back afterwards -- longer write bursts
// per wave, same bytes.  Rows are dealt to blocks in contiguous spans of S * 256.
template <int R, int S, bool ALL = false>
__global__ __launch_bounds__(256) void k_burst(int64_t n, const v2d *__restrict__ V, v2d *w, int64_t ld, v2d *sink) {
    extern __shared__ v2d park[];            // [S][256]
    const int64_t span = (int64_t)S * 256;
    const int64_t n_span = (n + span - 1) / span;
    v2d tot = (v2d){0.0, 0.0};
    for (int64_t sp = blockIdx.x; sp < n_span; sp += gridDim.x) {
        const int64_t base = sp * span + threadIdx.x;
#pragma unroll 1
        for (int t = 0; t < S; ++t) {
            const int64_t i = base + (int64_t)t * 256;
            v2d acc = (v2d){0.0, 0.0};
            v2d wv = (v2d){0.0, 0.0};
            if (i < n) {
                wv = w[i];
                if (ALL) {                      // every panel load of the trip in flight at once (as k_update_proj)
                    v2d v[R];
#pragma unroll
                    for (int c = 0; c < R; ++c) v[c] = __builtin_nontemporal_load(&V[i + c * ld]);
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int c = 0; c < R; ++c) acc += v[c];
                } else {
#pragma unroll
                    for (int c = 0; c < R; ++c) acc += __builtin_nontemporal_load(&V[i + c * ld]);
                }
            }
            const v2d r = wv - acc * 1e-3;
            tot += r;
            park[t * 256 + threadIdx.x] = r;
        }
#pragma unroll 4
        for (int t = 0; t < S; ++t) {
            const int64_t i = base + (int64_t)t * 256;
            if (i < n) w[i] = park[t * 256 + threadIdx.x];
        }
    }
    if (tot.x == 12345.678) sink[0] = tot;
}

template <int R, int S, bool ALL = false> void run_burst(int64_t n, const v2d *V, v2d *w, int64_t ld, v2d *sink, int grid) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t smem = (size_t)S * 256 * 16;
    CK(hipFuncSetAttribute((const void *)k_burst<R, S, ALL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k_burst<R, S, ALL>), dim3(grid), dim3(256), smem, 0, n, V, w, ld, sink);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k_burst<R, S, ALL>), dim3(grid), dim3(256), smem, 0, n, V, w, ld, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    const double bytes = 16.0 * n * (R + 2);
    printf("  %2d panel streams + w, writes parked in LDS for %3d trips (%3d KB per block)%s grid %5d: %.4f ms  %.2f TB/s\n", R, S,
           (int)(smem >> 10), ALL ? " all loads in flight" : "", grid, ms, bytes / ms / 1e9);
    fflush(stdout);
}

template <int R, int MODE, int U> void run(int64_t n, const v2d *V, v2d *w, v2d *other, int64_t ld, v2d *sink, int grid, const char *what) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k_mix<R, MODE, U>), dim3(grid), dim3(256), 0, 0, n, V, w, other, ld, sink);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k_mix<R, MODE, U>), dim3(grid), dim3(256), 0, 0, n, V, w, other, ld, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    const double bytes = 16.0 * n * (R + 1 + (MODE != 0));
    printf("  %2d panel streams + w, %-44s U=%d grid %5d: %.4f ms  %.2f TB/s\n", R, what, U, grid, ms, bytes / ms / 1e9);
    fflush(stdout);
}

template <int R> void sweep(int64_t n, const v2d *V, v2d *w, v2d *other, int64_t ld, v2d *sink) {
    for (int grid : {1024, 2048, 4096}) {
        run<R, 0, 1>(n, V, w, other, ld, sink, grid, "read only");
        run<R, 1, 1>(n, V, w, other, ld, sink, grid, "w rewritten in place (plain store)");
        run<R, 2, 1>(n, V, w, other, ld, sink, grid, "w rewritten in place (non-temporal store)");
        run<R, 3, 1>(n, V, w, other, ld, sink, grid, "result to another buffer (plain store)");
        run<R, 4, 1>(n, V, w, other, ld, sink, grid, "w plain load + plain store in place");
    }
    run<R, 0, 2>(n, V, w, other, ld, sink, 2048, "read only");
    run<R, 1, 2>(n, V, w, other, ld, sink, 2048, "w rewritten in place (plain store)");
    run<R, 2, 2>(n, V, w, other, ld, sink, 2048, "w rewritten in place (non-temporal store)");
    for (int grid : {256, 512, 1024, 2048}) {
        run_burst<R, 1>(n, V, w, ld, sink, grid);
        run_burst<R, 4>(n, V, w, ld, sink, grid);
        run_burst<R, 8>(n, V, w, ld, sink, grid);
        run_burst<R, 16>(n, V, w, ld, sink, grid);
        run_burst<R, 32>(n, V, w, ld, sink, grid);
        run_burst<R, 1, true>(n, V, w, ld, sink, grid);
        run_burst<R, 8, true>(n, V, w, ld, sink, grid);
        run_burst<R, 32, true>(n, V, w, ld, sink, grid);
    }
}

int main() {
    const int64_t n = 10000000, ld = 10000064;
    v2d *V, *w, *other, *sink;
    CK(hipMalloc(&V, 41 * ld * 16)); CK(hipMalloc(&w, ld * 16)); CK(hipMalloc(&other, ld * 16)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(V, 0, 41 * ld * 16)); CK(hipMemset(w, 0, ld * 16));
    printf("n = %lld rows of 16 bytes per stream\n", (long long)n);
    sweep<12>(n, V, w, other, ld, sink);
    sweep<20>(n, V, w, other, ld, sink);
    return 0;
}
