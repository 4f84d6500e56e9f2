/* Plain-C host for libarnoldi_hip.so: no Python, no torch -- the C ABI of include/arnoldi_hip.h is
 * the whole interface.  Builds a 1-D Laplacian (tridiagonal, n = 5000) in CSR, runs an m = 12 step
 * Arnoldi expansion on the device (aks_arnoldi_expand = the reference's arnoldi_decomposition,
 * src/arnoldi/decomposition.py:13-68), copies V and H back and checks the Arnoldi invariants
 *     V^H V = I      and      A V_m = V_{m+1} H
 * on the host, then compresses the basis with aks_truncate and re-checks orthonormality, and repeats the
 * factorisation with the operator in the sliced form (aks_sell_plan_size / _fill, aks_csr_block.sell) and in
 * real-packed mode (aks_workspace_set_real + aks_arnoldi_expand with AKS_EXPAND_REAL_PACKED), and ends with the communicator entry
 * points on a one-rank communicator (both all-reduce paths, the byte exchange).
 *
 *   hipcc -x c abi_smoke.c -I../../include -L<dir of the .so> -larnoldi_hip -o abi_smoke
 * Exit code 0 = pass.  Used by tests/test_gpu_parity.py::test_c_abi_from_plain_c. */
#include <complex.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include "arnoldi_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)
#define CHECK_AKS(x) do { int rc_ = (int)(x); if (rc_ < 0) { \
    fprintf(stderr, "aks error %d (%s) at line %d\n", rc_, aks_last_error(), __LINE__); return 3; } } while (0)

int main(void) {
    const int64_t n = 5000;
    const int32_t m = 12, p = 5;
    const int64_t nnz = 3 * n - 2, ldv = (n + 63) / 64 * 64;

    /* host CSR of tridiag(1, -2, 1) */
    int32_t *indptr = malloc((n + 1) * sizeof *indptr), *indices = malloc(nnz * sizeof *indices);
    double *values = malloc(nnz * sizeof *values);
    int64_t k = 0;
    for (int64_t i = 0; i < n; ++i) {
        indptr[i] = (int32_t)k;
        if (i > 0) { indices[k] = (int32_t)(i - 1); values[k++] = 1.0; }
        indices[k] = (int32_t)i; values[k++] = -2.0;
        if (i < n - 1) { indices[k] = (int32_t)(i + 1); values[k++] = 1.0; }
    }
    indptr[n] = (int32_t)k;
    int32_t *tiles = malloc((n + 2) * sizeof *tiles);
    const int64_t n_tiles = aks_csr_plan_tiles(indptr, n, AKS_SPMV_TILE_NNZ, tiles, n + 2);
    CHECK_AKS(n_tiles);

    CHECK_AKS(aks_device_init());                    /* once per device: dynamic-LDS limits of the large-LDS kernels */
    aks_ws_layout lay;
    CHECK_AKS(aks_workspace_layout(n, m, &lay));

    int32_t *d_indptr, *d_indices, *d_tiles;
    double *d_values;
    aks_c128 *d_V, *d_H, *d_Q;
    void *d_ws;
    CHECK_HIP(hipMalloc((void **)&d_indptr, (n + 1) * 4));
    CHECK_HIP(hipMalloc((void **)&d_indices, nnz * 4));
    CHECK_HIP(hipMalloc((void **)&d_values, nnz * 8));
    CHECK_HIP(hipMalloc((void **)&d_tiles, (n_tiles + 1) * 4));
    CHECK_HIP(hipMalloc((void **)&d_V, (size_t)(m + 1) * ldv * 16));
    CHECK_HIP(hipMalloc((void **)&d_H, (size_t)(m + 1) * m * 16));
    CHECK_HIP(hipMalloc((void **)&d_Q, (size_t)m * p * 16));
    CHECK_HIP(hipMalloc(&d_ws, lay.total_bytes));
    CHECK_HIP(hipMemcpy(d_indptr, indptr, (n + 1) * 4, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_indices, indices, nnz * 4, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_values, values, nnz * 8, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_tiles, tiles, (n_tiles + 1) * 4, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemset(d_V, 0, (size_t)(m + 1) * ldv * 16));
    CHECK_HIP(hipMemset(d_H, 0, (size_t)(m + 1) * m * 16));

    /* start vector: deterministic, complex, unit norm */
    double complex *V = calloc((size_t)(m + 1) * ldv, sizeof *V);
    double nrm = 0.0;
    for (int64_t i = 0; i < n; ++i) { V[i] = sin(0.37 * (double)i + 1.0) + 0.5 * I * cos(0.11 * (double)i); nrm += creal(V[i] * conj(V[i])); }
    for (int64_t i = 0; i < n; ++i) V[i] /= sqrt(nrm);
    CHECK_HIP(hipMemcpy(d_V, V, (size_t)n * 16, hipMemcpyHostToDevice));

    CHECK_AKS(aks_workspace_init(d_ws, lay.total_bytes, n, m, NULL));
    aks_shard A;                                    /* one GPU: the whole matrix is the diagonal block */
    memset(&A, 0, sizeof A);
    A.diag.n_rows = A.diag.n_cols = n;
    A.diag.d_indptr = d_indptr; A.diag.d_indices = d_indices; A.diag.d_values = d_values;
    A.diag.d_tiles = d_tiles; A.diag.n_tiles = n_tiles;
    CHECK_AKS(aks_arnoldi_expand(&A, d_V, ldv, d_H, m, 0, m, 1e-8, sqrt(0.5), d_ws, lay.total_bytes, m, NULL, NULL, 0));
    CHECK_HIP(hipDeviceSynchronize());

    double complex *H = malloc((size_t)(m + 1) * m * sizeof *H);
    aks_ctrl ctrl;
    CHECK_HIP(hipMemcpy(V, d_V, (size_t)(m + 1) * ldv * 16, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(H, d_H, (size_t)(m + 1) * m * 16, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(&ctrl, d_ws, sizeof ctrl, hipMemcpyDeviceToHost));
    if (ctrl.broken || ctrl.steps_done != m) { fprintf(stderr, "unexpected control block\n"); return 4; }

    double worst_orth = 0.0, worst_rel = 0.0;
    for (int a = 0; a <= m; ++a)
        for (int b = 0; b <= m; ++b) {
            double complex s = 0.0;
            for (int64_t i = 0; i < n; ++i) s += conj(V[a * ldv + i]) * V[b * ldv + i];
            const double d = cabs(s - (a == b ? 1.0 : 0.0));
            if (d > worst_orth) worst_orth = d;
        }
    for (int j = 0; j < m; ++j)                       /* A v_j - sum_i V_i H[i][j] */
        for (int64_t i = 0; i < n; ++i) {
            double complex r = -2.0 * V[j * ldv + i];
            if (i > 0) r += V[j * ldv + i - 1];
            if (i < n - 1) r += V[j * ldv + i + 1];
            for (int c = 0; c <= j + 1; ++c) r -= V[c * ldv + i] * H[c * m + j];
            if (cabs(r) > worst_rel) worst_rel = cabs(r);
        }

    /* restart compression with a unitary Qp: first p columns of a Householder reflector */
    double complex *Q = calloc((size_t)m * p, sizeof *Q);
    for (int r = 0; r < m; ++r)
        for (int c = 0; c < p; ++c) Q[r * p + c] = (r == c ? 1.0 : 0.0) - 2.0 / m;   /* I - 2 u u^T, u = 1/sqrt(m) */
    CHECK_HIP(hipMemcpy(d_Q, Q, (size_t)m * p * 16, hipMemcpyHostToDevice));
    CHECK_AKS(aks_truncate(n, m, p, d_V, ldv, d_Q, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    CHECK_HIP(hipMemcpy(V, d_V, (size_t)(p + 1) * ldv * 16, hipMemcpyDeviceToHost));
    double worst_orth2 = 0.0;
    for (int a = 0; a <= p; ++a)
        for (int b = 0; b <= p; ++b) {
            double complex s = 0.0;
            for (int64_t i = 0; i < n; ++i) s += conj(V[a * ldv + i]) * V[b * ldv + i];
            const double d = cabs(s - (a == b ? 1.0 : 0.0));
            if (d > worst_orth2) worst_orth2 = d;
        }
    printf("abi %d: |V^H V - I| = %.2e, |A V - V H| = %.2e, after truncate |V^H V - I| = %.2e, second passes %d\n",
           aks_abi_version(), worst_orth, worst_rel, worst_orth2, ctrl.second_passes);
    if (!(worst_orth < 1e-12 && worst_rel < 1e-12 && worst_orth2 < 1e-12)) return 1;

    /* ---- the same factorisation with the operator in the sliced form (aks_sell_*): planned on the host by the
     * library, uploaded here, handed over through aks_csr_block.sell; H must come out the same ---- */
    {
        const int64_t n_slices = (n + 63) / 64, nnz_pad = aks_sell_plan_size(indptr, n);
        CHECK_AKS((int)(nnz_pad < 0 ? nnz_pad : 0));
        int64_t *slice_ptr = malloc((n_slices + 1) * sizeof *slice_ptr);
        int32_t *scol = malloc(nnz_pad * sizeof *scol);
        double *sval = malloc(nnz_pad * sizeof *sval);
        CHECK_AKS(aks_sell_plan_fill(indptr, indices, values, 0, n, slice_ptr, scol, sval));
        aks_sell_matrix S;
        memset(&S, 0, sizeof S);
        S.n_rows = S.n_cols = n; S.nnz = nnz; S.nnz_pad = nnz_pad; S.n_slices = n_slices;
        void *d_sp, *d_sc, *d_sv;
        CHECK_HIP(hipMalloc(&d_sp, (n_slices + 1) * 8));
        CHECK_HIP(hipMalloc(&d_sc, nnz_pad * 4));
        CHECK_HIP(hipMalloc(&d_sv, nnz_pad * 8));
        CHECK_HIP(hipMemcpy(d_sp, slice_ptr, (n_slices + 1) * 8, hipMemcpyHostToDevice));
        CHECK_HIP(hipMemcpy(d_sc, scol, nnz_pad * 4, hipMemcpyHostToDevice));
        CHECK_HIP(hipMemcpy(d_sv, sval, nnz_pad * 8, hipMemcpyHostToDevice));
        S.d_slice_ptr = d_sp; S.d_col = d_sc; S.d_val = d_sv;
        aks_c128 *d_V2, *d_H2;
        CHECK_HIP(hipMalloc((void **)&d_V2, (size_t)(m + 1) * ldv * 16));
        CHECK_HIP(hipMalloc((void **)&d_H2, (size_t)(m + 1) * m * 16));
        CHECK_HIP(hipMemset(d_V2, 0, (size_t)(m + 1) * ldv * 16));
        CHECK_HIP(hipMemset(d_H2, 0, (size_t)(m + 1) * m * 16));
        double complex *v0 = malloc((size_t)n * sizeof *v0);
        for (int64_t i = 0; i < n; ++i) v0[i] = sin(0.37 * (double)i + 1.0) + 0.5 * I * cos(0.11 * (double)i);
        for (int64_t i = 0; i < n; ++i) v0[i] /= sqrt(nrm);
        CHECK_HIP(hipMemcpy(d_V2, v0, (size_t)n * 16, hipMemcpyHostToDevice));
        CHECK_AKS(aks_workspace_init(d_ws, lay.total_bytes, n, m, NULL));
        aks_shard A2 = A;
        A2.diag.sell = &S;
        CHECK_AKS(aks_arnoldi_expand(&A2, d_V2, ldv, d_H2, m, 0, m, 1e-8, sqrt(0.5), d_ws, lay.total_bytes, m, NULL, NULL, 0));
        CHECK_HIP(hipDeviceSynchronize());
        double complex *H2 = malloc((size_t)(m + 1) * m * sizeof *H2);
        CHECK_HIP(hipMemcpy(H2, d_H2, (size_t)(m + 1) * m * 16, hipMemcpyDeviceToHost));
        double worst_h = 0.0;
        for (int i = 0; i < (m + 1) * m; ++i)
            if (cabs(H2[i] - H[i]) > worst_h) worst_h = cabs(H2[i] - H[i]);
        printf("sliced form (padding x %.3f): max |H_sliced - H_csr| = %.2e\n", (double)nnz_pad / (double)nnz, worst_h);
        if (!(worst_h < 1e-12)) return 1;
    }

    /* ---- the same factorisation in real-packed mode: a column is n float64 = ceil(n/2) complex slots ---- */
    const int64_t n_panel = (n + 1) / 2, ldp = (n_panel + 63) / 64 * 64;
    aks_ws_layout layr;
    CHECK_AKS(aks_workspace_layout(n_panel, m, &layr));
    aks_c128 *d_Vr, *d_Hr;
    void *d_wsr;
    CHECK_HIP(hipMalloc((void **)&d_Vr, (size_t)(m + 1) * ldp * 16));
    CHECK_HIP(hipMalloc((void **)&d_Hr, (size_t)(m + 1) * m * 16));
    CHECK_HIP(hipMalloc(&d_wsr, layr.total_bytes));
    CHECK_HIP(hipMemset(d_Vr, 0, (size_t)(m + 1) * ldp * 16));
    CHECK_HIP(hipMemset(d_Hr, 0, (size_t)(m + 1) * m * 16));
    double *Vr = calloc((size_t)(m + 1) * ldp * 2, sizeof *Vr);      /* column c starts at Vr + c * 2 * ldp */
    nrm = 0.0;
    for (int64_t i = 0; i < n; ++i) { Vr[i] = sin(0.37 * (double)i + 1.0); nrm += Vr[i] * Vr[i]; }
    for (int64_t i = 0; i < n; ++i) Vr[i] /= sqrt(nrm);
    CHECK_HIP(hipMemcpy(d_Vr, Vr, (size_t)n * 8, hipMemcpyHostToDevice));
    CHECK_AKS(aks_workspace_init(d_wsr, layr.total_bytes, n_panel, m, NULL));
    CHECK_AKS(aks_workspace_set_real(d_wsr, 1, NULL));
    CHECK_AKS(aks_arnoldi_expand(&A, d_Vr, ldp, d_Hr, m, 0, m, 1e-8, sqrt(0.5), d_wsr, layr.total_bytes, m, NULL, NULL,
                                 AKS_EXPAND_REAL_PACKED));
    CHECK_HIP(hipDeviceSynchronize());
    CHECK_HIP(hipMemcpy(Vr, d_Vr, (size_t)(m + 1) * ldp * 16, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(H, d_Hr, (size_t)(m + 1) * m * 16, hipMemcpyDeviceToHost));
    double worst_r = 0.0, worst_o = 0.0, worst_im = 0.0;
    for (int a = 0; a <= m; ++a)
        for (int b = 0; b <= m; ++b) {
            double sdot = 0.0;
            for (int64_t i = 0; i < n; ++i) sdot += Vr[(size_t)a * 2 * ldp + i] * Vr[(size_t)b * 2 * ldp + i];
            const double d = fabs(sdot - (a == b ? 1.0 : 0.0));
            if (d > worst_o) worst_o = d;
        }
    for (int j = 0; j < m; ++j) {
        for (int c = 0; c <= j + 1; ++c)
            if (fabs(cimag(H[c * m + j])) > worst_im) worst_im = fabs(cimag(H[c * m + j]));
        for (int64_t i = 0; i < n; ++i) {
            const double *vj = Vr + (size_t)j * 2 * ldp;
            double r = -2.0 * vj[i] + (i > 0 ? vj[i - 1] : 0.0) + (i < n - 1 ? vj[i + 1] : 0.0);
            for (int c = 0; c <= j + 1; ++c) r -= Vr[(size_t)c * 2 * ldp + i] * creal(H[c * m + j]);
            if (fabs(r) > worst_r) worst_r = fabs(r);
        }
    }
    printf("real-packed: |V^T V - I| = %.2e, |A V - V H| = %.2e, max |Im H| = %.1e\n", worst_o, worst_r, worst_im);
    if (!(worst_o < 1e-12 && worst_r < 1e-12 && worst_im == 0.0)) return 1;

    /* ---- the communicator entry points from plain C (ABI 5 and 6), on a ONE-rank communicator -- all a one-GPU box can make; RCCL is
     * the system's (dlopen("librccl.so")), no Python in the process: id, create, the small all-reduce through ncclAllReduce and
     * -- a second communicator made with AKS_ALLREDUCE=oneshot -- through the one-shot mailbox exchange, and the byte exchange the
     * torch-free host layer builds its set-up on (aks_comm_alltoallv: this rank's own slice is a device copy). ---- */
    if (getenv("AKS_ABI_SMOKE_NO_COMM") == NULL) {
        for (int pass = 0; pass < 2; ++pass) {
            char ident[AKS_COMM_ID_BYTES], why[128];
            void *comm = NULL;
            if (pass == 1) setenv("AKS_ALLREDUCE", "oneshot", 1);
            CHECK_AKS(aks_comm_unique_id(ident));
            CHECK_AKS(aks_comm_create(ident, 0, 1, &comm));
            const int path = aks_comm_allreduce_path(comm, why, (int64_t)sizeof why);
            CHECK_AKS(path);
            double h_buf[6] = {1.5, -2.0, 3.25, 0.0, 7.0, -0.5}, h_out[6], *d_buf, *d_out;
            CHECK_HIP(hipMalloc((void **)&d_buf, sizeof h_buf));
            CHECK_HIP(hipMalloc((void **)&d_out, sizeof h_buf));
            CHECK_HIP(hipMemcpy(d_buf, h_buf, sizeof h_buf, hipMemcpyHostToDevice));
            for (int rep = 0; rep < 3; ++rep) CHECK_AKS(aks_comm_allreduce_sum(comm, d_buf, 6, NULL));    /* sum over one rank = itself */
            const int64_t zero = 0, bytes = (int64_t)sizeof h_buf;
            CHECK_AKS(aks_comm_alltoallv(comm, d_buf, &zero, &bytes, d_out, &zero, &bytes, NULL));
            CHECK_HIP(hipDeviceSynchronize());
            CHECK_HIP(hipMemcpy(h_out, d_out, sizeof h_buf, hipMemcpyDeviceToHost));
            int same = 1;
            for (int i = 0; i < 6; ++i) same = same && h_out[i] == h_buf[i];
            printf("communicator from plain C, pass %d: all-reduce path %d%s%s, values %s\n", pass, path, why[0] ? " -- " : "", why,
                   same ? "intact" : "WRONG");
            /* ABI 6: no reduction timed out; a communicator that still counts a captured graph refuses to be destroyed (and is
             * intact afterwards), then goes once the count is back at zero */
            if (aks_comm_status(comm, why, (int64_t)sizeof why) != 0) { printf("aks_comm_status: %s\n", why); return 1; }
            if (aks_comm_graph_retain(comm) != 1) return 1;
            if (aks_comm_destroy(comm) >= 0) { printf("aks_comm_destroy accepted a communicator with a counted graph\n"); return 1; }
            CHECK_AKS(aks_comm_allreduce_sum(comm, d_buf, 6, NULL));                 /* still usable after the refusal */
            CHECK_HIP(hipDeviceSynchronize());
            if (aks_comm_graph_release(comm) != 0) return 1;
            CHECK_AKS(aks_comm_destroy(comm));
            CHECK_HIP(hipFree(d_buf));
            CHECK_HIP(hipFree(d_out));
            if (!same || path != pass) return 1;
        }
    }
    /* ---- ABI 6 measurement aids: the streaming copy (bit-exact) and the versions of what this process runs on ---- */
    {
        const int64_t bytes = 1 << 20;
        unsigned char *h = (unsigned char *)malloc((size_t)bytes), *d_a, *d_b;
        for (int64_t i = 0; i < bytes; ++i) h[i] = (unsigned char)(i * 131 + 7);
        CHECK_HIP(hipMalloc((void **)&d_a, (size_t)bytes));
        CHECK_HIP(hipMalloc((void **)&d_b, (size_t)bytes));
        CHECK_HIP(hipMemcpy(d_a, h, (size_t)bytes, hipMemcpyHostToDevice));
        CHECK_AKS(aks_stream_copy(d_b, d_a, bytes, NULL));
        if (aks_stream_copy(d_b, d_a, bytes - 8, NULL) >= 0) { printf("aks_stream_copy accepted a size that is no multiple of 16\n"); return 1; }
        CHECK_HIP(hipDeviceSynchronize());
        unsigned char *back = (unsigned char *)malloc((size_t)bytes);
        CHECK_HIP(hipMemcpy(back, d_b, (size_t)bytes, hipMemcpyDeviceToHost));
        const int same = memcmp(back, h, (size_t)bytes) == 0;
        int32_t v_rt = 0, v_drv = 0, v_rccl = 0;
        CHECK_AKS(aks_runtime_versions(&v_rt, &v_drv, &v_rccl));
        printf("aks_stream_copy: %s; hip runtime %d, driver %d, rccl %d\n", same ? "bit-exact" : "WRONG", v_rt, v_drv, v_rccl);
        free(h); free(back);
        CHECK_HIP(hipFree(d_a));
        CHECK_HIP(hipFree(d_b));
        if (!same || v_rt <= 0) return 1;
    }
    return 0;
}
