// Micro-benchmark (gfx950): LDS cost of the phase-2 accumulation of the tile-binned SpMV.
// Each wave adds to pseudo-random rows of a workgroup-shared accumulator array (8192 complex entries):
//   atomic   2 x ds_add_f64 per entry (re plane, im plane)              -- what k_pb_phase2 issues
//   rmw128   ds_read_b128 + 2 adds + ds_write_b128 on interleaved (re, im)  -- needs conflict-free scheduling
//   atomic32 2 x ds_add_f32 (for comparison of the 32- vs 64-bit atomic rate)
// Reported: cycles per wave-instruction group per CU at 2.4 GHz, for 64 and for 33 active lanes.
// hipcc --offload-arch=gfx950 -O3 -o lds_rmw_cost lds_rmw_cost.hip && ./lds_rmw_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(512) void k_lds(int iters, int active, double *out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *acc = reinterpret_cast<double *>(smem);
    float *accf = reinterpret_cast<float *>(smem);
    double2 *acc2 = reinterpret_cast<double2 *>(smem);
    for (int i = threadIdx.x; i < 16384; i += 512) acc[i] = 0.0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    const bool on = lane < active;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            h = h * 1664525u + 1013904223u;
            const int row = (h >> 12) & 8191;
            if (on) {
                if (MODE == 0) { unsafeAtomicAdd(&acc[row], 1.0); unsafeAtomicAdd(&acc[8192 + row], 2.0); }
                else if (MODE == 1) { double2 v = acc2[row]; v.x += 1.0; v.y += 2.0; acc2[row] = v; }
                else { unsafeAtomicAdd(&accf[row], 1.0f); unsafeAtomicAdd(&accf[8192 + row], 2.0f); }
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = acc[0] + acc[8192];
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    double *out;
    CK(hipMalloc(&out, 1 << 20));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_lds<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_lds<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_lds<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    const char *names[] = {"2 x ds_add_f64", "ds_read_b128 + ds_write_b128", "2 x ds_add_f32"};
    const int iters = 2000;
    for (int mode = 0; mode < 3; ++mode)
        for (int active : {64, 33, 8}) {
            auto launch = [&](int it) {
                if (mode == 0) hipLaunchKernelGGL(k_lds<0>, dim3(cus), dim3(512), 131072, 0, it, active, out);
                else if (mode == 1) hipLaunchKernelGGL(k_lds<1>, dim3(cus), dim3(512), 131072, 0, it, active, out);
                else hipLaunchKernelGGL(k_lds<2>, dim3(cus), dim3(512), 131072, 0, it, active, out);
            };
            launch(10);
            CK(hipEventRecord(e0));
            launch(iters);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipGetLastError());
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double groups_per_cu = 8.0 * iters * 8;      // 8 waves x iters x 8 entries-per-lane groups
            const double ns = ms * 1e6 / groups_per_cu;
            printf("%-30s %2d active lanes: %7.2f ns per wave-group per CU = %6.1f cycles at 2.4 GHz (%.2f cycles per entry)\n", names[mode], active, ns,
                   ns * 2.4, ns * 2.4 / active);
        }
    return 0;
}
