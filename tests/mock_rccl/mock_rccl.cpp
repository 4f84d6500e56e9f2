// TEST INFRASTRUCTURE: a stand-in for the seven RCCL entry points libarnoldi_hip.so uses, for ranks that SHARE
// one GPU (RCCL itself refuses two ranks on one device, and the GPU box of the tests has one).  The collectives
// go through a POSIX shared-memory segment and host copies; every call blocks the host until all ranks have
// made the matching call (a barrier with a time-out), so a mismatch in call ORDER, peer, or message size
// between the ranks -- what an untested multi-rank C path would get wrong -- fails loudly instead of passing.
//
// tests/mock_rccl/Makefile compiles the product's kernel source once more with every nccl* name mapped to a
// mock_nccl* name (so that the real RCCL torch has loaded into the process cannot satisfy the references) and
// links it against this file: tests/mock_rccl/libarnoldi_hip.so.  Never shipped, never loaded by the product.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>          // types + (renamed) prototypes

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

constexpr int MAX_RANKS = 8;
constexpr size_t AR_CAP = 8192;                 // doubles per all-reduce
// bytes per (source, destination) mailbox: AKS_MOCK_BOX_MB, default 192 MiB -- the 2-rank exchange of BASELINE config 5
// (n = 10M random CSR) is 73 MB per message (the old fixed 16 MiB could not rehearse it).  The segment
// holds nranks^2 mailboxes and is SPARSE: /dev/shm pages exist only for bytes that were actually sent.
constexpr size_t BOX_CAP_DEFAULT_MB = 192;
constexpr double TIMEOUT_DEFAULT_S = 180.0;     // AKS_MOCK_TIMEOUT_S

size_t env_size(const char *name, size_t dflt) {
    const char *v = getenv(name);
    if (v == nullptr || *v == 0) return dflt;
    char *end = nullptr;
    const unsigned long long x = strtoull(v, &end, 10);
    return (end == v || x == 0) ? dflt : (size_t)x;
}
const double TIMEOUT_S = (double)env_size("AKS_MOCK_TIMEOUT_S", (size_t)TIMEOUT_DEFAULT_S);

struct Shared {
    std::atomic<int> ready, count, sense;
    int nranks;
    size_t box_cap;                              // bytes per mailbox (every rank must have computed the same)
    std::atomic<long long> calls[MAX_RANKS];     // collective calls made so far, per rank (diagnostics)
    size_t box_bytes[MAX_RANKS][MAX_RANKS];
    double ar[MAX_RANKS][AR_CAP];
    // followed by nranks * nranks mailboxes of box_cap bytes
};

struct MockComm {
    Shared *sh = nullptr;
    char *boxes = nullptr;
    size_t bytes = 0, box_cap = 0;
    int rank = 0, nranks = 1, local_sense = 0;
    std::string name;
};

struct Op { bool send; void *ptr; size_t bytes; int peer; MockComm *comm; hipStream_t stream; };
thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;

[[noreturn]] void die(MockComm *c, const char *what) {
    fprintf(stderr, "mock RCCL, rank %d of %d: %s\n", c ? c->rank : -1, c ? c->nranks : -1, what);
    if (c && c->sh)
        for (int r = 0; r < c->nranks; ++r) fprintf(stderr, "   rank %d has made %lld collective calls\n", r, (long long)c->sh->calls[r].load());
    _exit(86);
}

void barrier(MockComm *c) {
    Shared *s = c->sh;
    c->local_sense ^= 1;
    if (s->count.fetch_add(1) + 1 == c->nranks) {
        s->count.store(0);
        s->sense.store(c->local_sense);
        return;
    }
    const auto t0 = std::chrono::steady_clock::now();
    while (s->sense.load() != c->local_sense) {
        sched_yield();
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > TIMEOUT_S)
            die(c, "timed out in a barrier: the ranks did not make matching collective calls");
    }
}

char *box(MockComm *c, int src, int dst) { return c->boxes + ((size_t)src * c->nranks + dst) * c->box_cap; }

void run_group(std::vector<Op> &ops) {
    if (ops.empty()) return;
    MockComm *c = ops[0].comm;
    for (auto &o : ops) {
        if (o.comm != c) die(c, "one group spans two communicators");
        if (hipStreamSynchronize(o.stream) != hipSuccess) die(c, "hipStreamSynchronize failed");
    }
    // The library issues the group on its side stream BEHIND an event recorded on the compute stream (the pack kernel).
    // RCCL's kernels would be ordered behind that event by the GPU; these host copies are only as ordered as
    // hipStreamSynchronize(side stream) makes them, and at full size that was observed not to cover the pack kernel
    // every time (run-to-run different bits with 73 MB messages, tests/thread_ranks_worker.py --case repro).  A
    // stand-in has no business being subtle: wait for the whole device.
    if (hipDeviceSynchronize() != hipSuccess) die(c, "hipDeviceSynchronize failed");
    c->sh->calls[c->rank].fetch_add(1);
    for (auto &o : ops)
        if (o.send) {
            if (o.bytes > c->box_cap) die(c, "message larger than the mock's mailbox (raise AKS_MOCK_BOX_MB)");
            if (hipMemcpy(box(c, c->rank, o.peer), o.ptr, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) die(c, "D2H copy failed");
            c->sh->box_bytes[c->rank][o.peer] = o.bytes;
        }
    barrier(c);
    for (auto &o : ops)
        if (!o.send) {
            if (c->sh->box_bytes[o.peer][c->rank] != o.bytes) {
                char msg[160];
                snprintf(msg, sizeof msg, "recv of %zu bytes from rank %d, but that rank sent %zu", o.bytes, o.peer,
                         c->sh->box_bytes[o.peer][c->rank]);
                die(c, msg);
            }
            if (hipMemcpy(o.ptr, box(c, o.peer, c->rank), o.bytes, hipMemcpyHostToDevice) != hipSuccess) die(c, "H2D copy failed");
        }
    // hipMemcpy from PAGEABLE host memory may return once the data sits in the runtime's staging buffer, before the DMA
    // to the device has finished -- and the kernels that read the ghost buffer run on non-blocking streams, which the
    // null stream does not order.  Without this wait a 73 MB message (2-rank config 5) could land after its reader had
    // started: run-to-run different bits (round 4, tests/thread_ranks_worker.py --case repro).
    if (hipStreamSynchronize(nullptr) != hipSuccess) die(c, "hipStreamSynchronize(null stream) failed");
    barrier(c);
    for (auto &o : ops)
        if (o.send) c->sh->box_bytes[c->rank][o.peer] = (size_t)-1;     // consumed: a recv without a send now mismatches
    barrier(c);
    ops.clear();
}

}  // namespace

extern "C" {

const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "mock RCCL error"; }

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    memset(id, 0, sizeof *id);
    unsigned long long v = (unsigned long long)getpid() * 2654435761ull ^ (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count();
    snprintf(id->internal, sizeof id->internal, "/aks_mock_rccl_%llx", v);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int nranks, ncclUniqueId id, int rank) {
    if (nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    MockComm *c = new MockComm();
    c->rank = rank; c->nranks = nranks; c->name = id.internal;
    c->box_cap = env_size("AKS_MOCK_BOX_MB", BOX_CAP_DEFAULT_MB) << 20;
    c->bytes = sizeof(Shared) + (size_t)nranks * nranks * c->box_cap;
    int fd = -1;
    const auto t0 = std::chrono::steady_clock::now();
    if (rank == 0) {
        fd = shm_open(c->name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) { delete c; return ncclSystemError; }
    } else {
        while ((fd = shm_open(c->name.c_str(), O_RDWR, 0600)) < 0) {
            usleep(1000);
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > TIMEOUT_S) { delete c; return ncclSystemError; }
        }
    }
    void *p = nullptr;
    for (;;) {          // (a non-zero rank may have opened the segment before rank 0 sized it)
        p = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        if (p != MAP_FAILED && (rank == 0 || lseek(fd, 0, SEEK_END) >= (off_t)c->bytes)) break;
        if (p != MAP_FAILED) munmap(p, c->bytes);
        usleep(1000);
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > TIMEOUT_S) { close(fd); delete c; return ncclSystemError; }
    }
    close(fd);
    c->sh = static_cast<Shared *>(p);
    c->boxes = static_cast<char *>(p) + sizeof(Shared);
    if (rank == 0) {
        c->sh->nranks = nranks;
        c->sh->box_cap = c->box_cap;
        c->sh->count.store(0); c->sh->sense.store(0);
        for (int a = 0; a < MAX_RANKS; ++a) { c->sh->calls[a].store(0); for (int b = 0; b < MAX_RANKS; ++b) c->sh->box_bytes[a][b] = (size_t)-1; }
        c->sh->ready.store(1);
    } else {
        while (c->sh->ready.load() != 1) {
            usleep(1000);
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > TIMEOUT_S) die(c, "rank 0 never initialised the segment");
        }
        if (c->sh->nranks != nranks) die(c, "ranks disagree on the communicator's size");
        if (c->sh->box_cap != c->box_cap) die(c, "ranks disagree on AKS_MOCK_BOX_MB");
    }
    barrier(c);
    *out = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    MockComm *c = reinterpret_cast<MockComm *>(comm);
    if (c == nullptr) return ncclSuccess;
    if (c->rank == 0) shm_unlink(c->name.c_str());
    munmap(c->sh, c->bytes);
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclGroupStart() { ++g_depth; return ncclSuccess; }

ncclResult_t ncclGroupEnd() {
    if (g_depth <= 0) return ncclInvalidUsage;
    if (--g_depth == 0) run_group(g_ops);
    return ncclSuccess;
}

static size_t dt_width(ncclDataType_t dt) { return dt == ncclDouble ? 8 : (dt == ncclInt8 ? 1 : 0); }   // (what the library sends)

ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t stream) {
    MockComm *c = reinterpret_cast<MockComm *>(comm);
    if (dt_width(dt) == 0 || peer < 0 || peer >= c->nranks) return ncclInvalidArgument;
    g_ops.push_back(Op{true, const_cast<void *>(buf), count * dt_width(dt), peer, c, stream});
    if (g_depth == 0) run_group(g_ops);
    return ncclSuccess;
}

ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t stream) {
    MockComm *c = reinterpret_cast<MockComm *>(comm);
    if (dt_width(dt) == 0 || peer < 0 || peer >= c->nranks) return ncclInvalidArgument;
    g_ops.push_back(Op{false, buf, count * dt_width(dt), peer, c, stream});
    if (g_depth == 0) run_group(g_ops);
    return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void *sendbuf, void *recvbuf, size_t count, ncclDataType_t dt, ncclRedOp_t op,
                           ncclComm_t comm, hipStream_t stream) {
    MockComm *c = reinterpret_cast<MockComm *>(comm);
    if (dt != ncclDouble || op != ncclSum || count > AR_CAP) return ncclInvalidArgument;
    if (g_depth != 0) die(c, "all-reduce inside a send/recv group");
    if (hipStreamSynchronize(stream) != hipSuccess) die(c, "hipStreamSynchronize failed");
    c->sh->calls[c->rank].fetch_add(1);
    if (hipMemcpy(c->sh->ar[c->rank], sendbuf, count * 8, hipMemcpyDeviceToHost) != hipSuccess) die(c, "D2H copy failed");
    c->sh->box_bytes[c->rank][c->rank] = count;           // (the diagonal mailbox size doubles as the all-reduce length)
    barrier(c);
    std::vector<double> sum(count, 0.0);
    for (int r = 0; r < c->nranks; ++r) {                 // rank order: the same bits on every rank
        if (c->sh->box_bytes[r][r] != count) die(c, "ranks disagree on the length of an all-reduce");
        for (size_t i = 0; i < count; ++i) sum[i] += c->sh->ar[r][i];
    }
    if (hipMemcpy(recvbuf, sum.data(), count * 8, hipMemcpyHostToDevice) != hipSuccess) die(c, "H2D copy failed");
    if (hipStreamSynchronize(nullptr) != hipSuccess) die(c, "hipStreamSynchronize(null stream) failed");   // (see run_group)
    barrier(c);
    c->sh->box_bytes[c->rank][c->rank] = (size_t)-1;
    barrier(c);
    return ncclSuccess;
}

}  // extern "C"
