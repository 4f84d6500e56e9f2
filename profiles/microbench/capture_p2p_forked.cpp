// Round 3's capture crash against the SYSTEM's ROCm (no torch in the process): a grouped ncclSend / ncclRecv to self on a stream
// that joined the capture through an event -- the variant that sends torch's bundled HIP 7.0.51831 + RCCL 2.26.6 into an unbounded
// recursion of hip::Stream::EndCapture() (profiles/r05_capture_crash.txt).  Does HIP 7.2 + RCCL 2.27 (what /opt/rocm ships) do the same?
//   hipcc -O2 --offload-arch=gfx950 -o capture_p2p_forked capture_p2p_forked.cpp -lrccl && ./capture_p2p_forked [same|forked]
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <cstdio>
#include <cstring>
#include <unistd.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 2; } } while (0)
#define NK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { printf("%s -> %s (line %d)\n", #x, ncclGetErrorString(r_), __LINE__); return 3; } } while (0)

int main(int argc, char **argv) {
    const bool forked = argc < 2 || strcmp(argv[1], "same") != 0;
    ncclUniqueId id;
    ncclComm_t comm;
    NK(ncclGetUniqueId(&id));
    NK(ncclCommInitRank(&comm, 1, id, 0));
    const size_t n = 1400;
    std::vector<double> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = (double)i;
    double *a, *b;
    CK(hipMalloc(&a, n * 8));
    CK(hipMalloc(&b, n * 8));
    CK(hipMemcpy(a, h.data(), n * 8, hipMemcpyHostToDevice));
    hipStream_t origin, side;
    hipEvent_t fork, join;
    CK(hipStreamCreateWithFlags(&origin, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    auto body = [&]() -> int {
        hipStream_t s = origin;
        if (forked) { CK(hipEventRecord(fork, origin)); CK(hipStreamWaitEvent(side, fork, 0)); s = side; }
        NK(ncclGroupStart());
        NK(ncclSend(a, n, ncclDouble, 0, comm, s));
        NK(ncclRecv(b, n, ncclDouble, 0, comm, s));
        NK(ncclGroupEnd());
        if (forked) { CK(hipEventRecord(join, side)); CK(hipStreamWaitEvent(origin, join, 0)); }
        return 0;
    };
    printf("[%s] eager\n", forked ? "forked" : "same stream"); fflush(stdout);
    if (body() != 0) return 4;
    CK(hipStreamSynchronize(origin));
    CK(hipStreamSynchronize(side));
    printf("[%s] capture: begin\n", forked ? "forked" : "same stream"); fflush(stdout);
    hipGraph_t graph;
    hipGraphExec_t exec;
    CK(hipStreamBeginCapture(origin, hipStreamCaptureModeRelaxed));
    if (body() != 0) return 5;
    CK(hipStreamEndCapture(origin, &graph));
    printf("[%s] capture ended; instantiate + launch\n", forked ? "forked" : "same stream"); fflush(stdout);
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    CK(hipMemset(b, 0, n * 8));
    CK(hipGraphLaunch(exec, origin));
    CK(hipStreamSynchronize(origin));
    std::vector<double> out(n);
    CK(hipMemcpy(out.data(), b, n * 8, hipMemcpyDeviceToHost));
    printf("[%s] OK: replayed, b == a: %s\n", forked ? "forked" : "same stream", memcmp(out.data(), h.data(), n * 8) == 0 ? "true" : "FALSE"); fflush(stdout);
    _exit(0);          // (ncclCommDestroy after a captured p2p hung in the torch-bundled variant: not what this probe is about)
}
