// Micro-benchmark (gfx950): a sliced-ELL SpMV (one lane per row, a slice of 64 rows stored entry-major so that
// the value / column loads of a wave are contiguous) on the structured matrices of BASELINE.json -- the question
// being whether such a form beats the library's CSR-stream kernel (k_spmv: 0.41 ms on the 3-D Laplacian
// n = 16.2M, 0.161 ms on the Markov matrix n = 10M, 0.157 ms on the banded stand-in) by enough to be worth a
// third SpMV form.  y = A x, A real (f64), x / y complex128.
//   hipcc --offload-arch=gfx950 -O3 -o profiles/microbench/sell_spmv profiles/microbench/sell_spmv.hip
//   ./profiles/microbench/sell_spmv
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <climits>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef double2 c128;

template <int U, bool XCD, bool NT = false>
__global__ __launch_bounds__(256) void k_sell(int64_t n_rows, const int64_t *__restrict__ slice_ptr,
                                             const int32_t *__restrict__ col, const double *__restrict__ val,
                                             const c128 *__restrict__ x, c128 *__restrict__ y) {
    const int lane = threadIdx.x & 63;
    // XCD: workgroups with the same blockIdx % 8 share an XCD (observed placement); give each XCD a contiguous
    // eighth of the rows so that the x entries its rows share are fetched into ONE L2, not into all eight
    const int64_t per = (gridDim.x + 7) / 8;
    const int64_t wg = XCD ? (int64_t)(blockIdx.x & 7) * per + (blockIdx.x >> 3) : (int64_t)blockIdx.x;
    const int64_t slice = wg * 4 + (threadIdx.x >> 6);
    const int64_t row = slice * 64 + lane;
    if (slice * 64 >= n_rows) return;
    const int64_t p0 = slice_ptr[slice], p1 = slice_ptr[slice + 1];
    const int W = (int)((p1 - p0) >> 6);
    double sr = 0.0, si = 0.0;
    const int32_t *c = col + p0 + lane;
    const double *v = val + p0 + lane;
    int k = 0;
    for (; k + U <= W; k += U) {
        int32_t cc[U];
        double vv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            cc[u] = NT ? __builtin_nontemporal_load(&c[(int64_t)(k + u) * 64]) : c[(int64_t)(k + u) * 64];
            vv[u] = NT ? __builtin_nontemporal_load(&v[(int64_t)(k + u) * 64]) : v[(int64_t)(k + u) * 64];
        }
        c128 xx[U];
#pragma unroll
        for (int u = 0; u < U; ++u) xx[u] = cc[u] >= 0 ? x[cc[u]] : make_double2(0.0, 0.0);
#pragma unroll
        for (int u = 0; u < U; ++u) { sr = fma(vv[u], xx[u].x, sr); si = fma(vv[u], xx[u].y, si); }
    }
    for (; k < W; ++k) {
        const int32_t cc = c[(int64_t)k * 64];
        const double vv = v[(int64_t)k * 64];
        const c128 xx = cc >= 0 ? x[cc] : make_double2(0.0, 0.0);
        sr = fma(vv, xx.x, sr);
        si = fma(vv, xx.y, si);
    }
    if (row < n_rows) y[row] = make_double2(sr, si);
}

// Round 4: fewer, wider vector-memory instructions.  A vector-memory instruction costs a CU >= 21 cycles whatever it
// loads (vmem_issue_cost.txt), and the entry-major layout spends one 4-byte and one 8-byte load per entry: 3 instructions
// per entry with the gather.  Here the entries of a lane are stored in GROUPS of G consecutive k ([group][lane][G]): one
// 16-byte load brings G = 4 columns, two bring 4 values (G = 2: one 8-byte + one 16-byte load) -- 7 instead of 12
// instructions per 4 entries.  The W % G last entries of a slice stay entry-major.  Same summation order (k ascending).
template <int G, int UG, bool NT>
__global__ __launch_bounds__(256) void k_sell_grp(int64_t n_rows, const int64_t *__restrict__ slice_ptr,
                                                 const int32_t *__restrict__ col, const double *__restrict__ val,
                                                 const c128 *__restrict__ x, c128 *__restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int64_t per = (gridDim.x + 7) / 8;
    const int64_t wg = (int64_t)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
    const int64_t slice = wg * 4 + (threadIdx.x >> 6);
    const int64_t row = slice * 64 + lane;
    if (slice * 64 >= n_rows) return;
    const int64_t p0 = slice_ptr[slice], p1 = slice_ptr[slice + 1];
    const int W = (int)((p1 - p0) >> 6);
    const int ng = W / G;
    double sr = 0.0, si = 0.0;
    typedef int32_t ivec __attribute__((ext_vector_type(G)));
    typedef double dvec __attribute__((ext_vector_type(G)));
    const ivec *cg = reinterpret_cast<const ivec *>(col + p0) + lane;
    const dvec *vg = reinterpret_cast<const dvec *>(val + p0) + lane;
    int g = 0;
    for (; g + UG <= ng; g += UG) {
        ivec cc[UG];
        dvec vv[UG];
#pragma unroll
        for (int u = 0; u < UG; ++u) {
            cc[u] = NT ? __builtin_nontemporal_load(&cg[(int64_t)(g + u) * 64]) : cg[(int64_t)(g + u) * 64];
            vv[u] = NT ? __builtin_nontemporal_load(&vg[(int64_t)(g + u) * 64]) : vg[(int64_t)(g + u) * 64];
        }
        c128 xx[UG][G];
#pragma unroll
        for (int u = 0; u < UG; ++u)
#pragma unroll
            for (int e = 0; e < G; ++e) xx[u][e] = x[cc[u][e] >= 0 ? cc[u][e] : 0];
#pragma unroll
        for (int u = 0; u < UG; ++u)
#pragma unroll
            for (int e = 0; e < G; ++e)
                if (cc[u][e] >= 0) { sr = fma(vv[u][e], xx[u][e].x, sr); si = fma(vv[u][e], xx[u][e].y, si); }
    }
    for (; g < ng; ++g) {
        const ivec cc = cg[(int64_t)g * 64];
        const dvec vv = vg[(int64_t)g * 64];
#pragma unroll
        for (int e = 0; e < G; ++e) {
            const c128 xx = x[cc[e] >= 0 ? cc[e] : 0];
            if (cc[e] >= 0) { sr = fma(vv[e], xx.x, sr); si = fma(vv[e], xx.y, si); }
        }
    }
    const int32_t *c = col + p0 + (int64_t)ng * G * 64 + lane;           // entry-major tail
    const double *v = val + p0 + (int64_t)ng * G * 64 + lane;
    for (int k = 0; k < W - ng * G; ++k) {
        const int32_t cc = c[(int64_t)k * 64];
        const double vv = v[(int64_t)k * 64];
        const c128 xx = x[cc >= 0 ? cc : 0];
        if (cc >= 0) { sr = fma(vv, xx.x, sr); si = fma(vv, xx.y, si); }
    }
    if (row < n_rows) y[row] = make_double2(sr, si);
}

// Round 3: the same walk as a two-stage software pipeline.  Two register sets (A, B) of U entries alternate: the
// gathers of one set are issued, THEN the (column, value) loads of the set after next go out, then the gathered
// set is summed -- vmcnt counts in issue order, so waiting for the gathers leaves the 2 U younger stream loads in
// flight.  All loads are unconditional (slots past the slice's end re-read its last slot and are masked at use).
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_sell_ab(int64_t n_rows, const int64_t *__restrict__ slice_ptr,
                                                const int32_t *__restrict__ col, const double *__restrict__ val,
                                                const c128 *__restrict__ x, c128 *__restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int64_t per = (gridDim.x + 7) / 8;
    const int64_t wg = (int64_t)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
    const int64_t slice = wg * 4 + (threadIdx.x >> 6);
    const int64_t row = slice * 64 + lane;
    if (slice * 64 >= n_rows) return;
    const int64_t p0 = slice_ptr[slice], p1 = slice_ptr[slice + 1];
    const int W = (int)((p1 - p0) >> 6);
    double sr = 0.0, si = 0.0;
    if (W > 0) {
        const int32_t *c = col + p0 + lane;
        const double *v = val + p0 + lane;
        int32_t cA[U], cB[U];
        double vA[U], vB[U];
        auto fetch = [&](int32_t (&cc)[U], double (&vv)[U], int k0) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = min(k0 + u, W - 1);
                cc[u] = NT ? __builtin_nontemporal_load(&c[(int64_t)k * 64]) : c[(int64_t)k * 64];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = min(k0 + u, W - 1);
                vv[u] = NT ? __builtin_nontemporal_load(&v[(int64_t)k * 64]) : v[(int64_t)k * 64];
            }
        };
        // three register sets rotate (no register copies: a copy of a loaded value makes the compiler wait for it):
        // trip t gathers set t % 3, then prefetches block t + 2 into the set trip t - 1 consumed, then sums
        int32_t cC[U];
        double vC[U];
        fetch(cA, vA, 0);
        fetch(cB, vB, U);
        auto trip = [&](int32_t (&cc)[U], double (&vv)[U], int32_t (&cn)[U], double (&vn)[U], int k0) {
            c128 xx[U];
#pragma unroll
            for (int u = 0; u < U; ++u) xx[u] = x[max(cc[u], 0)];
            __builtin_amdgcn_sched_barrier(0);      // the machine scheduler otherwise sinks each gather to its use
            fetch(cn, vn, k0 + 2 * U);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (k0 + u < W && cc[u] >= 0) { sr = fma(vv[u], xx[u].x, sr); si = fma(vv[u], xx[u].y, si); }
            __builtin_amdgcn_sched_barrier(0);
        };
        for (int k0 = 0; k0 < W; k0 += 3 * U) {
            trip(cA, vA, cC, vC, k0);
            trip(cB, vB, cA, vA, k0 + U);
            trip(cC, vC, cB, vB, k0 + 2 * U);
        }
    }
    if (row < n_rows) y[row] = make_double2(sr, si);
}

// Round 4, second experiment: 16-bit column deltas.  In a sliced layout slot k of the 64 rows of a slice holds nearly the
// same diagonal of a structured matrix, so a column is  base[slice][k] + delta  with a 16-bit delta (0xFFFF = padding);
// two deltas share one 32-bit word ([pair][lane]): 2 instead of 4 bytes per entry AND one column load per two entries.
// Matrix bytes 12 -> 10 per entry (+ 4 bytes per 64 entries of bases).  Same values, same summation order.
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_sell16(int64_t n_rows, const int64_t *__restrict__ slice_ptr, const int64_t *__restrict__ pair_ptr,
                                               const uint32_t *__restrict__ cpair, const int32_t *__restrict__ base,
                                               const double *__restrict__ val, const c128 *__restrict__ x, c128 *__restrict__ y) {
    static_assert(U % 2 == 0, "pairs");
    const int lane = threadIdx.x & 63;
    const int64_t per = (gridDim.x + 7) / 8;
    const int64_t wg = (int64_t)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
    const int64_t slice = wg * 4 + (threadIdx.x >> 6);
    const int64_t row = slice * 64 + lane;
    if (slice * 64 >= n_rows) return;
    const int64_t p0 = slice_ptr[slice], p1 = slice_ptr[slice + 1];
    const int W = (int)((p1 - p0) >> 6);
    double sr = 0.0, si = 0.0;
    const uint32_t *c = cpair + pair_ptr[slice] + lane;
    const double *v = val + p0 + lane;
    const int32_t *b = base + (p0 >> 6);
    int k = 0;
    for (; k + U <= W; k += U) {
        uint32_t pp[U / 2];
        double vv[U];
        int32_t cc[U];
#pragma unroll
        for (int h = 0; h < U / 2; ++h) pp[h] = NT ? __builtin_nontemporal_load(&c[(int64_t)(k / 2 + h) * 64]) : c[(int64_t)(k / 2 + h) * 64];
#pragma unroll
        for (int u = 0; u < U; ++u) vv[u] = NT ? __builtin_nontemporal_load(&v[(int64_t)(k + u) * 64]) : v[(int64_t)(k + u) * 64];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t d = (u & 1) ? pp[u / 2] >> 16 : pp[u / 2] & 0xFFFFu;
            cc[u] = d == 0xFFFFu ? -1 : b[k + u] + (int32_t)d;
        }
        c128 xx[U];
#pragma unroll
        for (int u = 0; u < U; ++u) xx[u] = cc[u] >= 0 ? x[cc[u]] : make_double2(0.0, 0.0);
#pragma unroll
        for (int u = 0; u < U; ++u) { sr = fma(vv[u], xx[u].x, sr); si = fma(vv[u], xx[u].y, si); }
    }
    for (; k < W; ++k) {
        const uint32_t pw = c[(int64_t)(k / 2) * 64];
        const uint32_t d = (k & 1) ? pw >> 16 : pw & 0xFFFFu;
        const int32_t cc = d == 0xFFFFu ? -1 : b[k] + (int32_t)d;
        const double vv = v[(int64_t)k * 64];
        const c128 xx = cc >= 0 ? x[cc] : make_double2(0.0, 0.0);
        sr = fma(vv, xx.x, sr);
        si = fma(vv, xx.y, si);
    }
    if (row < n_rows) y[row] = make_double2(sr, si);
}

struct Csr { int64_t n; std::vector<int64_t> ptr; std::vector<int32_t> idx; std::vector<double> val; };

static Csr laplace3d(int nx, int ny, int nz) {
    Csr A; A.n = (int64_t)nx * ny * nz; A.ptr.assign(A.n + 1, 0);
    A.idx.reserve(7 * A.n); A.val.reserve(7 * A.n);
    for (int z = 0; z < nz; ++z) for (int yy = 0; yy < ny; ++yy) for (int xx = 0; xx < nx; ++xx) {
        const int64_t r = ((int64_t)z * ny + yy) * nx + xx;
        auto add = [&](int64_t c, double v) { A.idx.push_back((int32_t)c); A.val.push_back(v); };
        if (z > 0) add(r - (int64_t)nx * ny, -1.0);
        if (yy > 0) add(r - nx, -1.0);
        if (xx > 0) add(r - 1, -1.0);
        add(r, 6.0);
        if (xx + 1 < nx) add(r + 1, -1.0);
        if (yy + 1 < ny) add(r + nx, -1.0);
        if (z + 1 < nz) add(r + (int64_t)nx * ny, -1.0);
        A.ptr[r + 1] = (int64_t)A.idx.size();
    }
    return A;
}

static Csr banded(int64_t n, int per_row) {          // like bench.py --workload banded: per_row entries around the diagonal
    Csr A; A.n = n; A.ptr.assign(n + 1, 0);
    for (int64_t r = 0; r < n; ++r) {
        for (int j = 0; j < per_row; ++j) {
            const int64_t c = r + (int64_t)(j - per_row / 2);      // a dense band, as matrices.banded_csr
            if (c >= 0 && c < n) { A.idx.push_back((int32_t)c); A.val.push_back(1.0 + 0.01 * j); }
        }
        A.ptr[r + 1] = (int64_t)A.idx.size();
    }
    return A;
}

static Csr markov_like(int64_t n) {                  // 2-4 entries per row at +-1 and +-~sqrt(2n) (mark()'s pattern)
    Csr A; A.n = n; A.ptr.assign(n + 1, 0);
    const int64_t w = (int64_t)std::sqrt(2.0 * (double)n);
    for (int64_t r = 0; r < n; ++r) {
        const int64_t cs[4] = {r - w, r - 1, r + 1, r + w};
        for (int j = 0; j < 4; ++j)
            if (cs[j] >= 0 && cs[j] < n && ((r + j) % 7 != 0)) { A.idx.push_back((int32_t)cs[j]); A.val.push_back(0.25); }
        A.ptr[r + 1] = (int64_t)A.idx.size();
    }
    return A;
}

template <typename T> static T *upload(const std::vector<T> &v) {
    T *d; CK(hipMalloc(&d, std::max<size_t>(v.size(), 1) * sizeof(T)));
    CK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return d;
}

static void run(const char *name, const Csr &A) {
    const int64_t n = A.n, nnz = A.ptr[n], ns = (n + 63) / 64;
    std::vector<int64_t> sp(ns + 1, 0);
    for (int64_t s = 0; s < ns; ++s) {
        int64_t w = 0;
        for (int64_t r = s * 64; r < std::min(n, (s + 1) * 64); ++r) w = std::max(w, A.ptr[r + 1] - A.ptr[r]);
        sp[s + 1] = sp[s] + w * 64;
    }
    std::vector<int32_t> col(sp[ns], -1);
    std::vector<double> val(sp[ns], 0.0);
    for (int64_t r = 0; r < n; ++r)
        for (int64_t k = A.ptr[r]; k < A.ptr[r + 1]; ++k) {
            const int64_t q = sp[r / 64] + (k - A.ptr[r]) * 64 + (r % 64);
            col[q] = A.idx[k]; val[q] = A.val[k];
        }
    std::vector<double> hx(2 * n);
    for (int64_t i = 0; i < n; ++i) { hx[2 * i] = std::sin(0.001 * (double)(i % 100003)) + 0.5; hx[2 * i + 1] = std::cos(0.003 * (double)(i % 70001)); }
    int64_t *d_sp = upload(sp); int32_t *d_col = upload(col); double *d_val = upload(val);
    c128 *x = (c128 *)upload(hx), *y; CK(hipMalloc(&y, n * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double alg = 12.0 * nnz + 36.0 * n + 4;
    printf("%s: n = %lld, nnz = %lld, padded %lld (x %.3f)\n", name, (long long)n, (long long)nnz, (long long)sp[ns], (double)sp[ns] / nnz);
    auto time_it = [&](auto launch, const char *what) {
        for (int i = 0; i < 3; ++i) launch();
        CK(hipEventRecord(e0));
        for (int i = 0; i < 20; ++i) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
        printf("   %-10s %.4f ms  %.3f TB/s algorithmic (%.3f of 8 TB/s)\n", what, ms, alg / ms / 1e9, alg / ms / 8e9);
    };
    const unsigned grid = (unsigned)((ns + 3) / 4);
    const unsigned grid8 = (grid + 7) / 8 * 8;
    time_it([&] { hipLaunchKernelGGL((k_sell<2, false>), dim3(grid), dim3(256), 0, 0, n, d_sp, d_col, d_val, x, y); }, "unroll 2");
    time_it([&] { hipLaunchKernelGGL((k_sell<4, false>), dim3(grid), dim3(256), 0, 0, n, d_sp, d_col, d_val, x, y); }, "unroll 4");
    time_it([&] { hipLaunchKernelGGL((k_sell<2, true>), dim3(grid8), dim3(256), 0, 0, n, d_sp, d_col, d_val, x, y); }, "u2 + XCD");
    time_it([&] { hipLaunchKernelGGL((k_sell<4, true>), dim3(grid8), dim3(256), 0, 0, n, d_sp, d_col, d_val, x, y); }, "u4 + XCD");
    time_it([&] { hipLaunchKernelGGL((k_sell<4, true, true>), dim3(grid8), dim3(256), 0, 0, n, d_sp, d_col, d_val, x, y); }, "u4 XCD nt");
    time_it([&] { hipLaunchKernelGGL((k_sell<7, true, true>), dim3(grid8), dim3(256), 0, 0, n, d_sp, d_col, d_val, x, y); }, "u7 XCD nt");
    time_it([&] { hipLaunchKernelGGL((k_sell<8, true, true>), dim3(grid8), dim3(256), 0, 0, n, d_sp, d_col, d_val, x, y); }, "u8 XCD nt");
    time_it([&] { hipLaunchKernelGGL((k_sell_ab<2, true>), dim3(grid8), dim3(256), 0, 0, n, d_sp, d_col, d_val, x, y); }, "ab2 XCD nt");
    time_it([&] { hipLaunchKernelGGL((k_sell_ab<4, true>), dim3(grid8), dim3(256), 0, 0, n, d_sp, d_col, d_val, x, y); }, "ab4 XCD nt");
    time_it([&] { hipLaunchKernelGGL((k_sell_ab<6, true>), dim3(grid8), dim3(256), 0, 0, n, d_sp, d_col, d_val, x, y); }, "ab6 XCD nt");
    time_it([&] { hipLaunchKernelGGL((k_sell_ab<8, true>), dim3(grid8), dim3(256), 0, 0, n, d_sp, d_col, d_val, x, y); }, "ab8 XCD nt");
    // grouped layouts (G = 4, 2): [group][lane][G] for the W / G full groups of a slice, entry-major tail
    for (int G : {4, 2}) {
        std::vector<int32_t> colg(sp[ns], -1);
        std::vector<double> valg(sp[ns], 0.0);
        for (int64_t r = 0; r < n; ++r) {
            const int64_t s0 = sp[r / 64], W = (sp[r / 64 + 1] - s0) / 64, ng = W / G, lane = r % 64;
            for (int64_t k = A.ptr[r]; k < A.ptr[r + 1]; ++k) {
                const int64_t kk = k - A.ptr[r];
                const int64_t q = kk < ng * G ? s0 + ((kk / G) * 64 + lane) * G + kk % G : s0 + ng * G * 64 + (kk - ng * G) * 64 + lane;
                colg[q] = A.idx[k]; valg[q] = A.val[k];
            }
        }
        int32_t *d_cg = upload(colg); double *d_vg = upload(valg);
        if (G == 4) {
            time_it([&] { hipLaunchKernelGGL((k_sell_grp<4, 1, true>), dim3(grid8), dim3(256), 0, 0, n, d_sp, d_cg, d_vg, x, y); }, "g4 u1 nt");
            time_it([&] { hipLaunchKernelGGL((k_sell_grp<4, 2, true>), dim3(grid8), dim3(256), 0, 0, n, d_sp, d_cg, d_vg, x, y); }, "g4 u2 nt");
            time_it([&] { hipLaunchKernelGGL((k_sell_grp<4, 1, false>), dim3(grid8), dim3(256), 0, 0, n, d_sp, d_cg, d_vg, x, y); }, "g4 u1");
        } else {
            time_it([&] { hipLaunchKernelGGL((k_sell_grp<2, 2, true>), dim3(grid8), dim3(256), 0, 0, n, d_sp, d_cg, d_vg, x, y); }, "g2 u2 nt");
            time_it([&] { hipLaunchKernelGGL((k_sell_grp<2, 4, true>), dim3(grid8), dim3(256), 0, 0, n, d_sp, d_cg, d_vg, x, y); }, "g2 u4 nt");
        }
        CK(hipFree(d_cg)); CK(hipFree(d_vg));
    }
    {   // 16-bit deltas against a base per (slice, slot); two deltas per 32-bit word
        std::vector<int64_t> pp(ns + 1, 0);
        for (int64_t sl = 0; sl < ns; ++sl) pp[sl + 1] = pp[sl] + (((sp[sl + 1] - sp[sl]) / 64 + 1) / 2) * 64;
        std::vector<uint32_t> cpair(pp[ns], 0xFFFFFFFFu);
        std::vector<int32_t> base(sp[ns] / 64, 0);
        bool fits = true;
        for (int64_t sl = 0; sl < ns; ++sl) {
            const int64_t W = (sp[sl + 1] - sp[sl]) / 64;
            for (int64_t k = 0; k < W; ++k) {
                int64_t lo = INT64_MAX, hi = -1;
                for (int l = 0; l < 64; ++l) { const int32_t cq = col[sp[sl] + k * 64 + l]; if (cq >= 0) { lo = std::min<int64_t>(lo, cq); hi = std::max<int64_t>(hi, cq); } }
                if (hi < 0) lo = 0;
                if (hi - lo > 65534) fits = false;
                base[sp[sl] / 64 + k] = (int32_t)lo;
                for (int l = 0; l < 64; ++l) {
                    const int32_t cq = col[sp[sl] + k * 64 + l];
                    const uint32_t d = cq >= 0 ? (uint32_t)(cq - lo) & 0xFFFFu : 0xFFFFu;
                    uint32_t &w = cpair[pp[sl] + (k / 2) * 64 + l];
                    w = (k & 1) ? (w & 0x0000FFFFu) | (d << 16) : (w & 0xFFFF0000u) | d;
                }
            }
        }
        if (!fits) printf("   16-bit deltas: a (slice, slot) spans more than 65534 columns -- not representable\n");
        else {
            int64_t *d_pp = upload(pp); uint32_t *d_cp = upload(cpair); int32_t *d_b = upload(base);
            printf("   (16-bit deltas: matrix bytes %.1f MB instead of %.1f MB)\n", (sp[ns] * 8.0 + pp[ns] * 4.0 + base.size() * 4.0) / 1e6, sp[ns] * 12.0 / 1e6);
            time_it([&] { hipLaunchKernelGGL((k_sell16<4, true>), dim3(grid8), dim3(256), 0, 0, n, d_sp, d_pp, d_cp, d_b, d_val, x, y); }, "d16 u4 nt");
            time_it([&] { hipLaunchKernelGGL((k_sell16<8, true>), dim3(grid8), dim3(256), 0, 0, n, d_sp, d_pp, d_cp, d_b, d_val, x, y); }, "d16 u8 nt");
            time_it([&] { hipLaunchKernelGGL((k_sell16<4, false>), dim3(grid8), dim3(256), 0, 0, n, d_sp, d_pp, d_cp, d_b, d_val, x, y); }, "d16 u4");
            time_it([&] { hipLaunchKernelGGL((k_sell<4, true, true>), dim3(grid8), dim3(256), 0, 0, n, d_sp, d_col, d_val, x, y); }, "u4 XCD nt");
            time_it([&] { hipLaunchKernelGGL((k_sell16<4, true>), dim3(grid8), dim3(256), 0, 0, n, d_sp, d_pp, d_cp, d_b, d_val, x, y); }, "d16 u4 nt");
            CK(hipFree(d_pp)); CK(hipFree(d_cp)); CK(hipFree(d_b));
        }
    }
    // check against the host CSR product on a sample of rows
    std::vector<double> hy(2 * n);
    CK(hipMemcpy(hy.data(), y, n * 16, hipMemcpyDeviceToHost));
    double err = 0;
    for (int64_t r = 0; r < n; r += 9973) {
        double sr = 0, si = 0;
        for (int64_t k = A.ptr[r]; k < A.ptr[r + 1]; ++k) { sr = std::fma(A.val[k], hx[2 * A.idx[k]], sr); si = std::fma(A.val[k], hx[2 * A.idx[k] + 1], si); }
        err = std::max(err, std::max(std::fabs(sr - hy[2 * r]), std::fabs(si - hy[2 * r + 1])));
    }
    printf("   max error on sampled rows %.3e\n", err);
    CK(hipFree(d_sp)); CK(hipFree(d_col)); CK(hipFree(d_val)); CK(hipFree(x)); CK(hipFree(y));
}

int main() {
    run("3-D Laplace 253x254x255", laplace3d(253, 254, 255));
    run("banded 35 per row, n = 1.5M", banded(1500000, 35));
    run("Markov-like, n = 10M", markov_like(10000000));
    run("2-D-like 5-point, n = 1M (as a 1000x1x1001 3-D grid)", laplace3d(1000, 1, 1001));
    return 0;
}
