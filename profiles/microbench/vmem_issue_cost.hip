// Micro-benchmark (gfx950): what does ONE wave-level vector-memory instruction cost a CU when the data
// is cache resident?  Decides the lane mapping of phase 2 of the tile-binned SpMV: is it the number of
// VMEM instructions, the number of distinct 128-byte lines per instruction, or the bytes that count?
//
// Every wave issues LOADS independent loads per iteration with a given address pattern over a window
// that fits L1 (16 KiB per CU) or L2 (2 MiB), at full occupancy (8 waves per SIMD), so the figure is a
// throughput, not a latency.  Reported: cycles per wave-instruction per CU (at the measured clock).
// hipcc --offload-arch=gfx950 -O3 -o vmem_issue_cost vmem_issue_cost.hip && ./vmem_issue_cost
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef double2 c128;

enum Pattern { COALESCED = 0, SHIFTED = 1, GROUP8 = 2, GROUP4 = 3, DIVERGENT = 4, RUN33 = 5, RUN33_SHIFT = 6 };

// element index (16-byte units for T = c128, 2-byte units for T = uint16_t) of lane `lane`, load `j`
__device__ __forceinline__ unsigned addr_of(int pattern, int lane, unsigned j, unsigned wave_seed, unsigned mask) {
    const unsigned h = (j * 2654435761u + wave_seed * 40503u);
    switch (pattern) {
        case COALESCED: return (((h & mask) & ~63u) + lane) & mask;                       // 64 consecutive, aligned to 64 elements
        case SHIFTED: return (((h & mask) & ~63u) + 1 + lane) & mask;            // the same, start shifted by one element
        case GROUP8: return ((((h + (lane >> 3) * 7919u) & mask) & ~7u) + (lane & 7)) & mask;   // 8 runs of 8, aligned
        case GROUP4: return ((((h + (lane >> 2) * 7919u) & mask) & ~3u) + (lane & 3)) & mask;   // 16 runs of 4, aligned
        case DIVERGENT: return (h + lane * 7919u * 9u) & mask;                   // 64 scattered elements
        case RUN33: return (((h & mask) & ~7u) + lane) & mask;                            // lanes 0..32 active (see mask below)
        default: return (((h & mask) & ~7u) + 3 + lane) & mask;
    }
}

template <typename T, int LOADS>
__global__ __launch_bounds__(256) void k_issue(const T *__restrict__ buf, int pattern, unsigned mask, int iters,
                                              double *__restrict__ sink) {
    const int lane = threadIdx.x & 63;
    const unsigned seed = blockIdx.x * 4 + (threadIdx.x >> 6);
    const bool active = (pattern == RUN33 || pattern == RUN33_SHIFT) ? lane < 33 : true;
    double acc = 0.0;
    for (int it = 0; it < iters; ++it) {
        T v[LOADS];
#pragma unroll
        for (int j = 0; j < LOADS; ++j)
            if (active) v[j] = buf[addr_of(pattern, lane, (unsigned)(it * LOADS + j), seed, mask)];
#pragma unroll
        for (int j = 0; j < LOADS; ++j) {
            if (active) {
                if constexpr (sizeof(T) == 16) acc += v[j].x;
                else acc += (double)v[j];
            }
        }
    }
    if (acc == 1.2345e301) sink[blockIdx.x] = acc;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double ghz = prop.clockRate / 1e6;
    printf("%s: %d CUs, %.2f GHz (nominal)\n", prop.name, cus, ghz);
    const size_t bytes = 64u << 20;
    void *buf;
    double *sink;
    CK(hipMalloc(&buf, bytes + (1u << 20)));      // every index is masked to the window; slack on top
    CK(hipMemset(buf, 0, bytes + (1u << 20)));
    CK(hipMalloc(&sink, 1 << 20));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const char *names[] = {"64 consecutive, aligned", "64 consecutive, start +1 element", "8 runs of 8 elements", "16 runs of 4 elements",
                           "64 scattered elements", "33 consecutive (31 lanes off), aligned to 8", "33 consecutive, start +3 elements"};
    constexpr int LOADS = 8;
    const int iters = 400, blocks = cus * 8;      // 8 blocks of 4 waves per CU = 8 waves per SIMD
    struct Win { size_t bytes; const char *name; } wins[] = {{16u << 10, "16 KiB window (L1)"}, {2u << 20, "2 MiB window (L2)"}, {64u << 20, "64 MiB window (Infinity Cache)"}};
    for (auto &w : wins) {
        for (int tsel = 0; tsel < 2; ++tsel) {
            const int eb = tsel == 0 ? 16 : 2;
            const unsigned mask = (unsigned)(w.bytes / eb) - 1;
            printf("\n%s, %d-byte elements\n", w.name, eb);
            for (int p = 0; p < 7; ++p) {
                auto launch = [&](int it) {
                    if (tsel == 0) hipLaunchKernelGGL((k_issue<c128, LOADS>), dim3(blocks), dim3(256), 0, 0, (const c128 *)buf, p, mask, it, sink);
                    else hipLaunchKernelGGL((k_issue<uint16_t, LOADS>), dim3(blocks), dim3(256), 0, 0, (const uint16_t *)buf, p, mask, it, sink);
                };
                launch(20);
                CK(hipEventRecord(e0));
                launch(iters);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipGetLastError());
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                const double instr_per_cu = (double)blocks * 4 * iters * LOADS / cus;
                const double ns = ms * 1e6 / instr_per_cu;
                printf("  %-46s %7.2f ns per wave-instruction per CU = %6.1f cycles at 2.4 GHz\n", names[p], ns, ns * 2.4);
            }
        }
    }
    return 0;
}
