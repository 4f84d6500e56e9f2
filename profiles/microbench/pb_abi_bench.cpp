// Times the library's two SpMV forms through the C ABI (no Python): random CSR like BASELINE config 5,
// aks_pb_plan_create/export -> aks_pb_spmv vs aks_csr_spmv, checks both against each other and twice
// against itself (bitwise).  Build + run (repo root):
//   hipcc --offload-arch=gfx950 -O3 -Iinclude -o profiles/microbench/pb_abi_bench profiles/microbench/pb_abi_bench.cpp \
//       -Larnoldi-py_amd/arnoldi_amd/lib -larnoldi_hip -Wl,-rpath,'$ORIGIN/../../arnoldi-py_amd/arnoldi_amd/lib'
//   ./profiles/microbench/pb_abi_bench [n] [per_row] [complex_values 0|1] [real_vectors 0|1]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "arnoldi_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define AK(x) do { int rc_ = (x); if (rc_ < 0) { printf("aks error %d (%s) at line %d\n", rc_, aks_last_error(), __LINE__); exit(1); } } while (0)

template <typename T> static T *upload(const std::vector<T> &v) {
    T *d;
    CK(hipMalloc(&d, std::max<size_t>(v.size(), 1) * sizeof(T)));
    CK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return d;
}

int main(int argc, char **argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 10000000;
    const int per_row = argc > 2 ? atoi(argv[2]) : 5;
    const int cplx = argc > 3 ? atoi(argv[3]) : 0;
    const int realv = argc > 4 ? atoi(argv[4]) : 0;
    const int64_t nnz = n * per_row;
    printf("random CSR n=%lld nnz=%lld values %s, vectors %s\n", (long long)n, (long long)nnz, cplx ? "complex128" : "float64",
           realv ? "float64" : "complex128");
    std::vector<int32_t> indptr(n + 1), indices(nnz);
    std::vector<double> values(nnz * (cplx ? 2 : 1));
    uint64_t st = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    for (int64_t r = 0; r <= n; ++r) indptr[r] = (int32_t)(r * per_row);
    for (int64_t k = 0; k < nnz; ++k) indices[k] = (int32_t)(rnd() % (uint64_t)n);
    for (auto &v : values) v = (double)(rnd() >> 11) * (2.0 / 9007199254740992.0) - 1.0;
    for (int64_t r = 0; r < n; ++r) std::sort(indices.begin() + r * per_row, indices.begin() + (r + 1) * per_row);

    AK(aks_device_init());
    // ---- plans
    std::vector<int32_t> tiles(n + 2);
    const int64_t n_tiles = aks_csr_plan_tiles(indptr.data(), n, AKS_SPMV_TILE_NNZ, tiles.data(), n + 2);
    AK((int)std::min<int64_t>(n_tiles, 0));
    auto t0 = std::chrono::steady_clock::now();
    aks_pb_sizes sz;
    void *plan = aks_pb_plan_create(indptr.data(), indices.data(), values.data(), cplx, n, n, &sz);
    if (!plan) { printf("plan failed: %s\n", aks_last_error()); return 1; }
    std::vector<double> val(sz.nnz_pad * (cplx ? 2 : 1));
    std::vector<uint16_t> lcol(sz.nnz_pad), lrow(sz.n_lrow);
    std::vector<int32_t> sb(sz.n_slabs), se(sz.n_slabs), rbp(sz.n_rowblocks + 1);
    std::vector<aks_pb_run> runs(sz.n_runs);
    AK(aks_pb_plan_export(plan, val.data(), lcol.data(), sb.data(), se.data(), runs.data(), rbp.data(), lrow.data()));
    aks_pb_plan_destroy(plan);
    printf("plan: %.2f s, nnz_pad %lld, %lld wave-loads, %lld (level,row) words, %d sub-slabs x %d row blocks\n",
           std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), (long long)sz.nnz_pad,
           (long long)sz.n_runs, (long long)sz.n_lrow, sz.n_slabs, sz.n_rowblocks);
    int32_t slab_bits, rb_bits, rpr;
    AK(aks_pb_params(&slab_bits, &rb_bits, &rpr));
    double mean_levels = 0, filled = 0;
    for (int64_t r = 0; r + rpr < sz.n_runs; r += rpr) mean_levels += (runs[r].info >> 21) & 15;
    for (int64_t r = 0; r < sz.n_runs; ++r) filled += ((runs[r].info >> 14) & 127) != 0;
    printf("%d wave-loads per round, mean levels per round %.2f, mean lanes per wave-load %.1f\n", rpr,
           mean_levels / (sz.n_runs / rpr - 1), nnz / filled);

    aks_pb_matrix A;
    memset(&A, 0, sizeof A);
    A.n_rows = A.n_cols = n; A.nnz = nnz; A.nnz_pad = sz.nnz_pad; A.n_runs = sz.n_runs; A.n_lrow = sz.n_lrow;
    A.n_slabs = sz.n_slabs; A.n_rowblocks = sz.n_rowblocks; A.values_complex = cplx;
    A.d_val = upload(val); A.d_lcol = upload(lcol); A.d_slab_begin = upload(sb); A.d_slab_end = upload(se);
    A.d_runs = upload(runs); A.d_rb_run_ptr = upload(rbp); A.d_lrow = upload(lrow);
    CK(hipMalloc((void **)&A.d_prod, sz.nnz_pad * 16));
    CK(hipMemset(A.d_prod, 0, sz.nnz_pad * 16));
    int32_t *d_indptr = upload(indptr), *d_indices = upload(indices), *d_tiles = upload(tiles);
    double *d_values = upload(values);

    std::vector<double> hx(2 * n);
    for (int64_t i = 0; i < n; ++i) { hx[2 * i] = std::sin(0.001 * (double)(i % 100003)) + 0.5; hx[2 * i + 1] = realv ? hx[2 * i] * 0.5 : std::cos(0.003 * (double)(i % 70001)); }
    double *x = upload(hx), *y1, *y2;
    CK(hipMalloc(&y1, n * 16));
    CK(hipMalloc(&y2, n * 16));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto time_ms = [&](auto f, int reps) {
        for (int i = 0; i < 3; ++i) f();
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) f();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        return (double)ms / reps;
    };
    auto csr = [&](double *y, int acc) {
        if (realv) AK(aks_csr_spmv_real(n, d_indptr, d_indices, d_values, d_tiles, n_tiles, 0, x, y, acc, nullptr, nullptr));
        else AK(aks_csr_spmv(n, d_indptr, d_indices, d_values, cplx, d_tiles, n_tiles, 0, (const aks_c128 *)x, (aks_c128 *)y, acc, nullptr, nullptr));
    };
    auto pb = [&](double *y, int acc) {
        if (realv) AK(aks_pb_spmv_real(&A, x, y, acc, nullptr, nullptr));
        else AK(aks_pb_spmv(&A, (const aks_c128 *)x, (aks_c128 *)y, acc, nullptr, nullptr));
    };
    const double alg = (cplx ? 20.0 : 12.0) * nnz + (realv ? 20.0 : 36.0) * n + 4;
    const double t_csr = time_ms([&] { csr(y1, 0); }, 10);
    const double t_pb = time_ms([&] { pb(y2, 0); }, 20);
    printf("CSR-stream  %8.4f ms  %6.3f TB/s algorithmic (%.3f of 8 TB/s)\n", t_csr, alg / t_csr / 1e9, alg / t_csr / 8e9);
    printf("tile-binned %8.4f ms  %6.3f TB/s algorithmic (%.3f of 8 TB/s)\n", t_pb, alg / t_pb / 1e9, alg / t_pb / 8e9);
    const size_t w = realv ? 1 : 2;
    std::vector<double> h1(w * n), h2(w * n), h3(w * n);
    CK(hipMemcpy(h1.data(), y1, w * n * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h2.data(), y2, w * n * 8, hipMemcpyDeviceToHost));
    pb(y2, 0);
    CK(hipMemcpy(h3.data(), y2, w * n * 8, hipMemcpyDeviceToHost));
    double err = 0, ref = 0;
    for (size_t i = 0; i < w * n; ++i) { err = std::max(err, std::fabs(h1[i] - h2[i])); ref = std::max(ref, std::fabs(h1[i])); }
    printf("max |binned - csr| = %.3e (max |y| %.3f); second binned run bitwise equal: %s\n", err, ref,
           memcmp(h2.data(), h3.data(), w * n * 8) == 0 ? "yes" : "NO");
    if (getenv("PB_TICKS")) {   // library built with -DAKS_PB_TICKS=1: per-workgroup s_memtime sums sit in the scratch
        const int n_wg = std::min(256, sz.n_rowblocks);
        std::vector<double> dbg(8 * (size_t)n_wg);
        CK(hipMemcpy(dbg.data(), A.d_prod, dbg.size() * 8, hipMemcpyDeviceToHost));
        double t[8] = {0};
        for (int b = 0; b < n_wg; ++b) for (int i = 0; i < 8; ++i) t[i] += dbg[8 * (size_t)b + i];
        printf("phase 2, wave 0, s_memtime ticks per round: issue loads + first-level adds %.0f | their barrier (adds of all waves done) %.0f | "
               "other levels + write-out %.0f | total %.0f (%.1f rounds per workgroup)\n",
               t[0] / t[7], t[1] / t[7], t[2] / t[7], t[6] / t[7], t[7] / n_wg);
    }
    // accumulate form
    pb(y2, 1);
    CK(hipMemcpy(h3.data(), y2, w * n * 8, hipMemcpyDeviceToHost));
    double err2 = 0;
    for (size_t i = 0; i < w * n; ++i) err2 = std::max(err2, std::fabs(h3[i] - 2.0 * h2[i]));
    printf("accumulate: max |(y + A x) - 2 y| = %.3e\n", err2);
    return (err > 1e-12 * std::max(ref, 1.0) || err2 > 1e-12 * std::max(ref, 1.0)) ? 2 : 0;
}
