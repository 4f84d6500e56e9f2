// Independent comparator for the operator apply (decomposition.py:58) on BASELINE config 5: rocSPARSE's CSR SpMV
// (float64 values, complex128 vectors -- the same mixed types the library's kernels use) on the SAME random
// matrix, box and vector, every CSR algorithm rocSPARSE offers, analysis (preprocess) stage excluded from the
// timing.  A cross-check only: nothing here is linked into the product; the library is linked in to put its own
// two forms beside rocSPARSE's numbers in one log and to compare the results.
//   hipcc --offload-arch=gfx950 -O3 -Iinclude -o profiles/microbench/rocsparse_crosscheck \
//       profiles/microbench/rocsparse_crosscheck.cpp -Larnoldi-py_amd/arnoldi_amd/lib -larnoldi_hip -lrocsparse \
//       -Wl,-rpath,'$ORIGIN/../../arnoldi-py_amd/arnoldi_amd/lib'
//   ./profiles/microbench/rocsparse_crosscheck n per_row alg     alg: 0..4 ONE of the algorithms listed below, 9 none (the
//                                                                library's forms only)
// ONE rocSPARSE algorithm per process, by construction: analysing one matrix descriptor for a second algorithm faulted
// inside rocSPARSE 4.2 ("Memory access fault by GPU ... address (nil)", round 3), so the program no longer has a mode
// that does that -- loop over the algorithms from the shell (profiles/r03_rocsparse_crosscheck.txt was made that way).
#include <hip/hip_runtime.h>
#include <rocsparse/rocsparse.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "arnoldi_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define AK(x) do { int rc_ = (x); if (rc_ < 0) { printf("aks error %d (%s) at line %d\n", rc_, aks_last_error(), __LINE__); exit(1); } } while (0)
#define RS(x) do { rocsparse_status s_ = (x); if (s_ != rocsparse_status_success) { printf("rocsparse status %d at line %d\n", (int)s_, __LINE__); exit(1); } } while (0)

#pragma clang diagnostic ignored "-Wdeprecated-declarations"

template <typename T> static T *upload(const std::vector<T> &v) {
    T *d;
    CK(hipMalloc(&d, std::max<size_t>(v.size(), 1) * sizeof(T)));
    CK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return d;
}

int main(int argc, char **argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 10000000;
    const int per_row = argc > 2 ? atoi(argv[2]) : 5;
    const int which = argc > 3 ? atoi(argv[3]) : -1;
    if (which < 0 || (which > 4 && which != 9)) {
        fprintf(stderr, "usage: rocsparse_crosscheck n per_row alg   (alg: 0..4 one rocSPARSE algorithm, 9 none; one per process)\n");
        return 2;
    }
    const int64_t nnz = n * per_row;
    setvbuf(stdout, nullptr, _IOLBF, 0);
    printf("random CSR n=%lld nnz=%lld, values float64, vectors complex128 (rocSPARSE %d)\n", (long long)n, (long long)nnz,
           ROCSPARSE_VERSION_MAJOR * 10000 + ROCSPARSE_VERSION_MINOR * 100 + ROCSPARSE_VERSION_PATCH);
    std::vector<int32_t> indptr(n + 1), indices(nnz);
    std::vector<double> values(nnz);
    uint64_t st = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    for (int64_t r = 0; r <= n; ++r) indptr[r] = (int32_t)(r * per_row);
    for (int64_t k = 0; k < nnz; ++k) indices[k] = (int32_t)(rnd() % (uint64_t)n);
    for (auto &v : values) v = (double)(rnd() >> 11) * (2.0 / 9007199254740992.0) - 1.0;
    for (int64_t r = 0; r < n; ++r) std::sort(indices.begin() + r * per_row, indices.begin() + (r + 1) * per_row);
    std::vector<double> hx(2 * n);
    for (int64_t i = 0; i < n; ++i) { hx[2 * i] = std::sin(0.001 * (double)(i % 100003)) + 0.5; hx[2 * i + 1] = std::cos(0.003 * (double)(i % 70001)); }
    int32_t *d_indptr = upload(indptr), *d_indices = upload(indices);
    double *d_values = upload(values), *x = upload(hx), *y_ref, *y;
    CK(hipMalloc(&y_ref, n * 16));
    CK(hipMalloc(&y, n * 16));
    const double alg_bytes = 12.0 * nnz + 36.0 * n + 4;

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
    auto report = [&](const char *name, double ms) {
        printf("%-44s %8.4f ms  %6.3f TB/s algorithmic (%.3f of 8 TB/s)\n", name, ms, alg_bytes / ms / 1e9, alg_bytes / ms / 8e9);
    };

    // ---- the library's two forms (reference result = the CSR-stream kernel)
    AK(aks_device_init());
    std::vector<int32_t> tiles(n + 2);
    const int64_t n_tiles = aks_csr_plan_tiles(indptr.data(), n, AKS_SPMV_TILE_NNZ, tiles.data(), n + 2);
    AK((int)std::min<int64_t>(n_tiles, 0));
    int32_t *d_tiles = upload(tiles);
    report("libarnoldi_hip  CSR-stream (k_spmv)",
           time_ms([&] { AK(aks_csr_spmv(n, d_indptr, d_indices, d_values, 0, d_tiles, n_tiles, 0, (const aks_c128 *)x, (aks_c128 *)y_ref, 0, nullptr, nullptr)); }, 10));
    {
        aks_pb_sizes sz;
        void *plan = aks_pb_plan_create(indptr.data(), indices.data(), values.data(), 0, n, n, &sz);
        if (!plan) { printf("plan failed: %s\n", aks_last_error()); return 1; }
        std::vector<double> val(sz.nnz_pad);
        std::vector<uint16_t> lcol(sz.nnz_pad), lrow(sz.n_lrow);
        std::vector<int32_t> sb(sz.n_slabs), se(sz.n_slabs), rbp(sz.n_rowblocks + 1);
        std::vector<aks_pb_run> runs(sz.n_runs);
        AK(aks_pb_plan_export(plan, val.data(), lcol.data(), sb.data(), se.data(), runs.data(), rbp.data(), lrow.data()));
        aks_pb_plan_destroy(plan);
        aks_pb_matrix A;
        memset(&A, 0, sizeof A);
        A.n_rows = A.n_cols = n; A.nnz = nnz; A.nnz_pad = sz.nnz_pad; A.n_runs = sz.n_runs; A.n_lrow = sz.n_lrow;
        A.n_slabs = sz.n_slabs; A.n_rowblocks = sz.n_rowblocks; A.values_complex = 0;
        A.d_val = upload(val); A.d_lcol = upload(lcol); A.d_slab_begin = upload(sb); A.d_slab_end = upload(se);
        A.d_runs = upload(runs); A.d_rb_run_ptr = upload(rbp); A.d_lrow = upload(lrow);
        CK(hipMalloc((void **)&A.d_prod, sz.nnz_pad * 16));
        report("libarnoldi_hip  tile-binned (k_pb_phase1+2)",
               time_ms([&] { AK(aks_pb_spmv(&A, (const aks_c128 *)x, (aks_c128 *)y, 0, nullptr, nullptr)); }, 20));
        CK(hipFree(A.d_prod)); CK(hipFree((void *)A.d_val)); CK(hipFree((void *)A.d_lcol)); CK(hipFree((void *)A.d_lrow)); CK(hipFree((void *)A.d_runs));
    }
    std::vector<double> h_ref(2 * n), h(2 * n);
    CK(hipMemcpy(h_ref.data(), y_ref, n * 16, hipMemcpyDeviceToHost));
    double ymax = 0;
    for (double v : h_ref) ymax = std::max(ymax, std::fabs(v));

    // ---- rocSPARSE, generic API: the one CSR algorithm asked for
    rocsparse_handle handle;
    RS(rocsparse_create_handle(&handle));
    rocsparse_spmat_descr matA;
    rocsparse_dnvec_descr vecX, vecY;
    RS(rocsparse_create_csr_descr(&matA, n, n, nnz, d_indptr, d_indices, d_values, rocsparse_indextype_i32, rocsparse_indextype_i32,
                                  rocsparse_index_base_zero, rocsparse_datatype_f64_r));
    RS(rocsparse_create_dnvec_descr(&vecX, n, x, rocsparse_datatype_f64_c));
    RS(rocsparse_create_dnvec_descr(&vecY, n, y, rocsparse_datatype_f64_c));
    const rocsparse_double_complex alpha = {1.0, 0.0}, beta = {0.0, 0.0};
    struct { rocsparse_spmv_alg alg; const char *name; } algs[] = {
        {rocsparse_spmv_alg_default, "rocSPARSE spmv  alg_default"},
        {rocsparse_spmv_alg_csr_adaptive, "rocSPARSE spmv  csr_adaptive"},
        {rocsparse_spmv_alg_csr_rowsplit, "rocSPARSE spmv  csr_rowsplit (stream)"},
        {rocsparse_spmv_alg_csr_lrb, "rocSPARSE spmv  csr_lrb"},
        {rocsparse_spmv_alg_csr_nnzsplit, "rocSPARSE spmv  csr_nnzsplit"},
    };
    int index = -1;
    for (auto &a : algs) {
        ++index;
        if (which != index) continue;              // exactly one algorithm ever touches the descriptor
        printf("%-44s ...\n", a.name);
        size_t bytes = 0;
        rocsparse_status s = rocsparse_spmv(handle, rocsparse_operation_none, &alpha, matA, vecX, &beta, vecY, rocsparse_datatype_f64_c,
                                            a.alg, rocsparse_spmv_stage_buffer_size, &bytes, nullptr);
        if (s != rocsparse_status_success) { printf("%-44s not available (status %d)\n", a.name, (int)s); continue; }
        void *buf = nullptr;
        CK(hipMalloc(&buf, std::max<size_t>(bytes, 16)));
        s = rocsparse_spmv(handle, rocsparse_operation_none, &alpha, matA, vecX, &beta, vecY, rocsparse_datatype_f64_c, a.alg,
                           rocsparse_spmv_stage_preprocess, &bytes, buf);
        if (s != rocsparse_status_success) { printf("%-44s preprocess failed (status %d)\n", a.name, (int)s); CK(hipFree(buf)); continue; }
        CK(hipDeviceSynchronize());
        CK(hipMemset(y, 0, n * 16));
        const double ms = time_ms([&] {
            RS(rocsparse_spmv(handle, rocsparse_operation_none, &alpha, matA, vecX, &beta, vecY, rocsparse_datatype_f64_c, a.alg,
                              rocsparse_spmv_stage_compute, &bytes, buf));
        }, 10);
        report(a.name, ms);
        CK(hipMemcpy(h.data(), y, n * 16, hipMemcpyDeviceToHost));
        double err = 0;
        for (size_t i = 0; i < h.size(); ++i) err = std::max(err, std::fabs(h[i] - h_ref[i]));
        printf("%-44s max |y - y_lib| = %.2e (max |y| %.3f)\n", "", err, ymax);
        CK(hipFree(buf));
    }
    RS(rocsparse_destroy_spmat_descr(matA));
    RS(rocsparse_destroy_dnvec_descr(vecX));
    RS(rocsparse_destroy_dnvec_descr(vecY));
    RS(rocsparse_destroy_handle(handle));
    return 0;
}
