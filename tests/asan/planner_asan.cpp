// TEST INFRASTRUCTURE (CPU only): drives the HOST-SIDE planner entry points of libarnoldi_hip -- index-heavy C++
// that nothing but functional tests had looked at -- in a build with -fsanitize=address,undefined on the host code
// (the device code is compiled as usual and never launched here: no GPU is needed).
//   aks_workspace_layout, aks_csr_plan_tiles, aks_pb_plan_create / _export / _view / _destroy, aks_sell_plan_size / _fill
// on the shapes of tests/test_gpu_parity.py::test_spmv_forms_agree_on_random_shapes plus degenerate ones
// (no entries, one row, rows of every length incl. empty ones and a hub row, non-square blocks, complex values,
// more row blocks than AKS_PB_CHUNKS, a tile count the binned form must refuse).  Every planned array is also
// checked against the CSR matrix it came from, so an out-of-bounds WRITE that stays inside a heap block shows up too.
//   make -C tests/asan && tests/asan/planner_asan
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

#include "arnoldi_hip.h"

#define REQUIRE(c) do { if (!(c)) { fprintf(stderr, "planner_asan: line %d: %s failed (%s)\n", __LINE__, #c, aks_last_error()); exit(1); } } while (0)

struct Csr { int64_t n_rows, n_cols; std::vector<int32_t> ptr, idx; std::vector<double> val; bool cplx; };

static uint64_t st = 0x243F6A8885A308D3ull;
static uint64_t rnd() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; }

static Csr make(int64_t n_rows, int64_t n_cols, double mean, bool cplx, int hub_row = -1, int64_t hub_len = 0) {
    Csr A{n_rows, n_cols, std::vector<int32_t>(n_rows + 1, 0), {}, {}, cplx};
    for (int64_t r = 0; r < n_rows; ++r) {
        int64_t len = mean <= 0 ? 0 : (int64_t)(rnd() % (uint64_t)(2 * mean + 1));
        if (rnd() % 7 == 0) len = 0;                       // empty rows
        if (r == hub_row) len = hub_len;
        len = std::min(len, n_cols);
        std::vector<int32_t> cols;
        for (int64_t k = 0; k < len; ++k) cols.push_back((int32_t)(rnd() % (uint64_t)n_cols));
        std::sort(cols.begin(), cols.end());
        cols.erase(std::unique(cols.begin(), cols.end()), cols.end());
        for (int32_t c : cols) {
            A.idx.push_back(c);
            A.val.push_back((double)(rnd() % 2001) / 1000.0 - 1.0);
            if (cplx) A.val.push_back((double)(rnd() % 2001) / 1000.0 - 1.0);
        }
        A.ptr[r + 1] = (int32_t)A.idx.size();
    }
    return A;
}

static void check(const Csr &A, const char *what) {
    const int64_t nnz = A.ptr[A.n_rows];
    const int vw = A.cplx ? 2 : 1;
    std::vector<int32_t> dummy_idx(1, 0);
    std::vector<double> dummy_val(2, 0.0);
    const int32_t *idx = nnz ? A.idx.data() : dummy_idx.data();     // (non-null pointers for a matrix without entries)
    const double *val = nnz ? A.val.data() : dummy_val.data();
    // ---- tiles
    {
        std::vector<int32_t> tiles(A.n_rows + 2);
        const int64_t nt = aks_csr_plan_tiles(A.ptr.data(), A.n_rows, AKS_SPMV_TILE_NNZ, tiles.data(), A.n_rows + 2);
        REQUIRE(nt >= 1 && tiles[0] == 0 && tiles[nt] == A.n_rows);
        for (int64_t t = 0; t < nt; ++t) {
            REQUIRE(tiles[t] < tiles[t + 1]);
            const int64_t k = A.ptr[tiles[t + 1]] - A.ptr[tiles[t]];
            REQUIRE(k <= AKS_SPMV_TILE_NNZ || tiles[t + 1] == tiles[t] + 1);
        }
        REQUIRE(aks_csr_plan_tiles(A.ptr.data(), A.n_rows, AKS_SPMV_TILE_NNZ, tiles.data(), 1) < 0);   // capacity too small
    }
    // ---- sliced form
    {
        const int64_t pad = aks_sell_plan_size(A.ptr.data(), A.n_rows);
        REQUIRE(pad >= nnz && pad % 64 == 0);
        const int64_t ns = (A.n_rows + 63) / 64;
        std::vector<int64_t> sp(ns + 1);
        std::vector<int32_t> col(std::max<int64_t>(pad, 1));
        std::vector<double> v(std::max<int64_t>(pad, 1) * vw);
        REQUIRE(aks_sell_plan_fill(A.ptr.data(), idx, val, A.cplx, A.n_rows, sp.data(), col.data(), v.data()) == AKS_OK);
        REQUIRE(sp[0] == 0 && sp[ns] == pad);
        int64_t live = 0;
        for (int64_t r = 0; r < A.n_rows; ++r) {
            const int64_t s = r / 64, w = (sp[s + 1] - sp[s]) / 64;
            REQUIRE(A.ptr[r + 1] - A.ptr[r] <= w);
            for (int64_t k = 0; k < w; ++k) {
                const int64_t q = sp[s] + k * 64 + (r - 64 * s);
                if (k < A.ptr[r + 1] - A.ptr[r]) {
                    REQUIRE(col[q] == idx[A.ptr[r] + k] && v[q * vw] == val[(A.ptr[r] + k) * vw]);
                    ++live;
                } else {
                    REQUIRE(col[q] == -1 && v[q * vw] == 0.0);
                }
            }
        }
        REQUIRE(live == nnz);
    }
    // ---- binned form
    {
        aks_pb_sizes sz;
        void *plan = aks_pb_plan_create(A.ptr.data(), idx, val, A.cplx, A.n_rows, A.n_cols, &sz);
        REQUIRE(plan != nullptr);
        std::vector<double> v(sz.nnz_pad * vw);
        std::vector<uint16_t> lcol(sz.nnz_pad), lrow(sz.n_lrow);
        std::vector<int32_t> sb(sz.n_slabs), se(sz.n_slabs), rbp(sz.n_rowblocks + 1);
        std::vector<aks_pb_run> runs(sz.n_runs);
        REQUIRE(aks_pb_plan_export(plan, v.data(), lcol.data(), sb.data(), se.data(), runs.data(), rbp.data(), lrow.data()) == AKS_OK);
        {   // the zero-copy view names the same bytes
            aks_pb_plan_arrays pa;
            REQUIRE(aks_pb_plan_view(plan, &pa) == AKS_OK && aks_pb_plan_view(nullptr, &pa) != AKS_OK);
            REQUIRE(aks_pb_plan_view(plan, &pa) == AKS_OK);
            REQUIRE(memcmp(pa.val, v.data(), v.size() * sizeof(v[0])) == 0 && memcmp(pa.lcol, lcol.data(), lcol.size() * 2) == 0);
            REQUIRE(memcmp(pa.slab_begin, sb.data(), sb.size() * 4) == 0 && memcmp(pa.slab_end, se.data(), se.size() * 4) == 0);
            REQUIRE(memcmp(pa.runs, runs.data(), runs.size() * sizeof(aks_pb_run)) == 0 && memcmp(pa.rb_run_ptr, rbp.data(), rbp.size() * 4) == 0);
            REQUIRE(memcmp(pa.lrow, lrow.data(), lrow.size() * 2) == 0);
        }
        aks_pb_plan_destroy(plan);
        int32_t slab_bits, rb_bits, rpr;
        REQUIRE(aks_pb_params(&slab_bits, &rb_bits, &rpr) == AKS_OK);
        REQUIRE(sz.n_lrow == sz.n_runs * AKS_PB_RUN_MAX && sz.n_runs % rpr == 0 && rbp[0] == 0 && rbp[sz.n_rowblocks] == sz.n_runs - rpr);
        // every entry appears exactly once, in a slot of its sub-slab, with its local column, row word and value
        std::map<std::pair<int64_t, int32_t>, double> want;       // (row, column) -> value (re)
        for (int64_t r = 0; r < A.n_rows; ++r) for (int32_t k = A.ptr[r]; k < A.ptr[r + 1]; ++k) want[{r, idx[k]}] = val[(int64_t)k * vw];
        const int64_t n_chunks = std::min<int64_t>(sz.n_rowblocks, AKS_PB_CHUNKS);
        std::vector<int64_t> rb_at;
        for (int64_t c = 0; c < n_chunks; ++c) for (int64_t rb = c; rb < sz.n_rowblocks; rb += n_chunks) rb_at.push_back(rb);
        std::vector<int32_t> slab_of(sz.nnz_pad, -1);
        for (int32_t s = 0; s < sz.n_slabs; ++s) {
            REQUIRE(sb[s] % 8 == 0 && sb[s] <= se[s] && se[s] <= sz.nnz_pad);
            for (int32_t q = sb[s]; q < se[s]; ++q) slab_of[q] = s;
        }
        int64_t seen = 0;
        const int64_t per_wave = rpr / AKS_PB_WAVES;
        for (int64_t pos = 0; pos < sz.n_rowblocks; ++pos)
            for (int64_t run = rbp[pos]; run < rbp[pos + 1]; ++run) {
                const uint32_t info = runs[run].info, l0 = info & 127, l01 = (info >> 7) & 127, total = (info >> 14) & 127;
                REQUIRE(l0 <= l01 && l01 <= total && total <= AKS_PB_RUN_MAX);
                for (uint32_t l = 0; l < total; ++l) {
                    const uint32_t q = (l < l0 ? runs[run].start0 : l < l01 ? runs[run].start1 : runs[run].start2) + l;
                    REQUIRE((int64_t)q < sz.nnz_pad && slab_of[q] >= 0);
                    const int64_t round = run / rpr, slot = run % rpr;
                    const uint16_t word = lrow[round * AKS_PB_ROUND_WORDS + (slot / per_wave) * (per_wave * AKS_PB_RUN_MAX) + l * per_wave + slot % per_wave];
                    const int64_t row = (rb_at[pos] << rb_bits) + (word & ((1 << rb_bits) - 1));
                    const int32_t c = (slab_of[q] << slab_bits) + lcol[q];
                    auto it = want.find({row, c});
                    REQUIRE(it != want.end() && it->second == v[(int64_t)q * vw]);
                    want.erase(it);
                    ++seen;
                }
            }
        REQUIRE(seen == nnz && want.empty());
    }
    // ---- workspace layout
    aks_ws_layout lay;
    REQUIRE(aks_workspace_layout(std::max<int64_t>(A.n_rows, 1), 20, &lay) == AKS_OK && lay.total_bytes > 0);
    printf("  ok  %-34s %8lld x %-8lld nnz %lld%s\n", what, (long long)A.n_rows, (long long)A.n_cols, (long long)nnz, A.cplx ? " (complex)" : "");
}

int main() {
    const int64_t shapes[][2] = {{1, 1}, {1, 70000}, {63, 64}, {64, 63}, {65, 9000}, {257, 257}, {5000, 5000}, {3000, 200000},
                                 {70000, 900}, {20000, 20000}, {8193, 8191}, {16385, 3}};
    for (auto &s : shapes)
        for (int cplx = 0; cplx < 2; ++cplx) {
            check(make(s[0], s[1], 6, cplx != 0), "random rows");
            check(make(s[0], s[1], 0, cplx != 0), "no entries at all");
        }
    check(make(20000, 20000, 5, false, 1, 17000), "hub row (every level)");
    check(make(300, 100000, 40, false, 7, 60000), "long rows, wide block");
    check(make(2130000, 30000, 1, false), "261 row blocks (> AKS_PB_CHUNKS)");
    {   // refused: more than 2^26 (sub-slab, row block) tiles; bad arguments
        std::vector<int32_t> ptr(70000001, 0);
        aks_pb_sizes sz;
        REQUIRE(aks_pb_plan_create(ptr.data(), ptr.data(), ptr.data(), 0, 70000000, 70000000, &sz) == nullptr);
        REQUIRE(aks_pb_plan_create(nullptr, ptr.data(), ptr.data(), 0, 10, 10, &sz) == nullptr);
        int32_t bad_ptr[3] = {0, 5, 3}, cols[5] = {0, 1, 2, 3, 4};
        double vals[5] = {1, 1, 1, 1, 1};
        REQUIRE(aks_pb_plan_create(bad_ptr, cols, vals, 0, 2, 10, &sz) == nullptr);                 // indptr not monotone
        int32_t ok_ptr[2] = {0, 2}, bad_cols[2] = {1, 99};
        REQUIRE(aks_pb_plan_create(ok_ptr, bad_cols, vals, 0, 1, 10, &sz) == nullptr);              // column out of range
        REQUIRE(aks_sell_plan_size(bad_ptr, 2) < 0);
        aks_ws_layout lay;
        REQUIRE(aks_workspace_layout(0, 20, &lay) < 0 && aks_workspace_layout(10, AKS_MAX_DIM + 1, &lay) < 0);
    }
    printf("planner_asan: all planner checks passed\n");
    return 0;
}
