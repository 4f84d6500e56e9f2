// libarnoldi_hip -- gfx950 (MI355X, CDNA4) kernels behind include/arnoldi_hip.h.
//
// Everything here is HBM-bound complex128 streaming work (0.17 - 5 flop/B):
// the design goal is "every byte of V / CSR crosses HBM once per stage, in 16-byte
// per-lane, 1-KiB per-wave coalesced loads, with enough of them in flight".
//
//   k_spmv<VT,XT>   CSR-stream SpMV: one 64-lane wave per tile of <= 256 non-zeros;
//                   indices/values are read coalesced, x is gathered (16 B/lane; 8 B for real
//                   vectors), products are staged in LDS and rows are summed from LDS by
//                   1..64 lanes per row.                (reference: decomposition.py:58)
//   k_pb_phase1/2   the same operator for matrices without column locality: products are formed
//                   sub-slab by sub-slab (8192 x entries staged in LDS) and written sequentially,
//                   then summed per 8192-row block in LDS (levels + barriers keep the order fixed).
//   k_sell<VT,XT>   the same operator for matrices with column locality and rows of similar length
//                   (stencils, bands): a lane per row, 64-row slices stored entry-major, rows summed in
//                   registers, XCD-contiguous slice order.
//   k_proj<NC>      tall-skinny  V[:, c0:c0+NC]^H w  with one lane per row and NC
//                   complex accumulators per lane (exact NC: no masked loads), plus
//                   ||w||^2.                            (ortho.py:92-94, 102)
//   k_update_proj<NC>  w -= V h fused with the re-projection V^H w and ||w||^2: the
//                   row's NC panel entries stay in registers between the two uses,
//                   so the panel is read once instead of twice.   (ortho.py:96-98,102)
//   k_update_proj_split<NQ>  the same for wide panels: the J columns are dealt to the 4
//                   waves of a block, partial updates meet in LDS.
//   k_update<true>  second DGKS pass  w -= V h + ||w||^2, predicated on the device.  When no n-sized normalisation
//                   follows (deferred normalisation) it also BOOKS the step -- H column, beta, breakdown, counters:
//                   workgroup 0 when the pass is not needed, the last workgroup to finish (ticket) when it is -- so
//                   neither k_reduce<true> nor k_finish is launched (round 4).   (ortho.py:101, 104-105)
//   k_reduce        deterministic second stage of the block partial sums.
//   k_finish        H column, beta, breakdown test, w /= beta  (the n-sized normalisation pass, and the book-keeping
//                   when the second-pass kernel did not do it).    (ortho.py:95,103,107; decomposition.py:61-66)
//   k_truncate_mfma<MT,NS>  V[:, :p] = V[:, :m] Qp in place on v_mfma_f64_16x16x4_f64 (one wave owns
//                   16 NS rows), V[:, p] = V[:, m].     (krylov_schur.py:78,81)
//   host side       aks_shard_apply / aks_arnoldi_expand chain these per Arnoldi step; with an RCCL
//                   communicator (aks_comm_*) they also issue the ghost exchange of the SpMV and the
//                   all-reduces between the Gram-Schmidt stages: one C call per expansion on every rank.
//
// Reductions are two-stage (per-block partials, then one fixed-order sum); the only atomics are LDS
// adds whose order is fixed by construction (one wave in program order, or level by level behind
// workgroup barriers in k_pb_phase2): results are bitwise reproducible.  Panel streams that are
// read once (the Krylov basis in the Gram-Schmidt and compression kernels) use non-temporal loads.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include <rccl/rccl.h>
#include <dlfcn.h>
#include <sys/mman.h>
#include <unistd.h>

#include "arnoldi_hip.h"

typedef double2 c128;

namespace {

thread_local std::string g_err;

int fail(int code, const char *what) {
    g_err = what;
    return code;
}

int hip_fail(hipError_t e, const char *where) {
    g_err = std::string(where) + ": " + hipGetErrorString(e);
    return AKS_ERR_HIP;
}

#define AKS_CHECK_LAUNCH(where)                                  \
    do {                                                         \
        hipError_t e_ = hipGetLastError();                       \
        if (e_ != hipSuccess) return hip_fail(e_, where);        \
    } while (0)

// A kernel launch that carries a start and / or a stop event (bench.py's probe): the events take the kernel's own
// begin / end time stamps -- no marker packets in front of and behind the kernel, which cost a 20-us launch 10 % and
// sit inside the timed region.
struct EvPair { hipEvent_t start = nullptr, stop = nullptr; };
template <typename K, typename... Args>
inline void launch_timed(K kernel, dim3 grid, dim3 block, size_t smem, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1, Args... args) {
    if (ev0 == nullptr && ev1 == nullptr) hipLaunchKernelGGL(kernel, grid, block, smem, s, args...);
    else hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)smem, s, ev0, ev1, 0u, args...);
}

constexpr int BLOCK = 256;
constexpr int WAVES = BLOCK / 64;
constexpr int MAX_ROW_BLOCKS = 1024;
#ifndef AKS_NC_MAX
#define AKS_NC_MAX 40             // (one launch instead of two groups for J = 33..40: -3 %, profiles/r03_proj40_ab.txt)
#endif
constexpr int NC_MAX = AKS_NC_MAX;        // widest exact-width projection kernel (accumulators only)

inline int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

// ------------------------------------------------------------------ device helpers
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Sum `v` over the block's 4 waves; result valid in thread 0.  `scratch` holds >= WAVES doubles.
__device__ __forceinline__ double block_sum(double v, double *scratch) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < WAVES; ++k) r += scratch[k];
    }
    return r;
}

// Panel loads.  The Krylov basis (GBs) streams through every Gram-Schmidt kernel once and is far larger than any
// cache: these loads are non-temporal.  Measured at n = 10M (profiles/ab_kernels.py, r02_b_ab_nt_panel.txt):
// projection 0.571 -> 0.507 ms at J = 20 (6.6 TB/s), fused update + re-projection 0.647 -> 0.610 ms; 11-14 % at
// J = 12..16.  AKS_NT_PANEL=0 / AKS_NT_TRUNC=0 / AKS_NT_TRUNC_STORE=0 rebuild the plain variants for A/B runs.
#ifndef AKS_NT_PANEL
#define AKS_NT_PANEL 1
#endif
#ifndef AKS_NT_TRUNC             // restart compression: panel loads and the stores of the compressed columns, both
#define AKS_NT_TRUNC 1            // streamed once: 1.005 -> 0.956 ms at m = 20, p = 10, n = 10M with 128 rows per wave
#endif                           // (profiles/r02_c_ab_truncate.txt; with 64 rows per wave the load hint alone was neutral)
#ifndef AKS_NT_TRUNC_STORE
#define AKS_NT_TRUNC_STORE 1
#endif
typedef double v2d_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ c128 ld_panel(const c128 *p) {
#if AKS_NT_PANEL
    const v2d_t v = __builtin_nontemporal_load(reinterpret_cast<const v2d_t *>(p));
    return make_double2(v.x, v.y);
#else
    return *p;
#endif
}

// ---- many-value wave reduction ------------------------------------------------------------
// Summing K per-lane values over the 64 lanes with K independent butterflies costs 6 K dependent
// cross-lane moves (2 ds_bpermute each for a double).  Here every step HALVES the values a lane
// still carries: lanes with the mask bit set keep the upper half and send the lower one to
// their partner (and vice versa), so K values need K - 1 exchanges in 6 rounds whose moves are
// all independent.  After the last round lane l holds the total of value  l >> (6 - log2 KP).
constexpr int next_pow2(int x) { int p = 1; while (p < x) p <<= 1; return p; }

template <int KP, int S>
__device__ __forceinline__ void transpose_reduce_step(double (&v)[KP], int lane) {
    if constexpr (S < 6) {
        constexpr int mask = 32 >> S;
        constexpr int half = KP >> (S + 1);
        if constexpr (half >= 1) {
            const bool upper = (lane & mask) != 0;
#pragma unroll
            for (int i = 0; i < half; ++i) {
                const double send = upper ? v[i] : v[i + half];
                const double keep = upper ? v[i + half] : v[i];
                v[i] = keep + __shfl_xor(send, mask, 64);
            }
        } else {
            v[0] += __shfl_xor(v[0], mask, 64);
        }
        transpose_reduce_step<KP, S + 1>(v, lane);
    }
}

// ---- last-arriver hand-off: the second stage of a reduction inside the producing kernel ---------------------------------
// The panel kernels sum their per-workgroup partial rows with a kernel of their own (k_reduce).  The second-pass kernel
// (k_update<true>) instead lets the LAST workgroup to finish do it when it also books the step (FIN_* below), which
// removes two launches (k_reduce<true> and k_finish) from every step that takes the second pass: every workgroup stores
// its partial write-through (sc1), drains (s_waitcnt vmcnt(0)), and one lane draws a ticket with an agent-scope atomic
// add behind the workgroup barrier; the workgroup that draws gridDim.x - 1 performs an agent-scope acquire, reads all
// partials back with sc1 loads and sums them in EXACTLY k_reduce's order -- thread t takes rows t, t + 256, ... in
// increasing order, then block_sum -- so H and every other number are bit for bit what the separate kernels produced.
// The ticket words live in the control block; the last arriver zeroes its word again, aks_workspace_init zeroes them all.
// (The same mechanism inside the two wide panel kernels was measured a LOSS -- 256 arrivals queue on one counter behind a
// streaming kernel whose workgroups all finish within a microsecond -- and is not in this file: profiles/HISTORY.md,
// profiles/r04_tail_ab.txt, profiles/r05_removed_variants.patch.)
#ifndef AKS_FOLD_FINISH
#define AKS_FOLD_FINISH 1
#endif
__device__ __forceinline__ void st_partial(c128 *p, double re, double im) {
    double *d = reinterpret_cast<double *>(p);
    __hip_atomic_store(d, re, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);          // global_store_dwordx2 ... sc1
    __hip_atomic_store(d + 1, im, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ c128 ld_partial(const c128 *p) {
    double *d = reinterpret_cast<double *>(const_cast<c128 *>(p));
    const double re = __hip_atomic_load(d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // global_load_dwordx2 ... sc1
    const double im = __hip_atomic_load(d + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_double2(re, im);
}
// All threads of the workgroup call this after their partial-row stores.  True in every thread of the workgroup that
// arrived last (all rows of `partial` written by this launch are then complete and readable with ld_partial).
// The hand-off in the documented release / acquire form (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement &
// inter-workgroup visibility"):
//   producer  every storing wave drains its stores (s_waitcnt vmcnt(0)), the workgroup meets at a barrier, then ONE lane
//             issues an agent-scope RELEASE (buffer_wbl2 sc1 + wait) and draws its ticket with an agent-scope add;
//   consumer  the lane whose add returned gridDim.x - 1 issues an agent-scope ACQUIRE (buffer_inv sc1), waits for it,
//             and only then releases the workgroup's other waves through the second barrier.
// The partials are in addition stored and loaded write-through / L1-bypassing (st_partial / ld_partial: sc1), which the
// guide measures as valid on its own only with one workgroup per CU -- the second-pass kernel runs two, and round 4's
// tree, which had neither fence, once in a while summed one STALE partial at full size (beta off by 1/512: residual
// 2.2e-3 instead of 8.7e-9).  tests/test_gpu_parity.py::test_ticket_hand_off_books_the_step pins the hand-off.  The
// release costs nothing measurable: restarts of Laplacians (every step takes this path) 7.725 / 15.18 / 107.8 ms with it,
// 7.765 / 15.17 / 109.8 ms without, builds interleaved on one box (profiles/r05_ticket_release_ab.txt).
__device__ __forceinline__ bool last_block_arrives(unsigned *ticket) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // EVERY storing wave: its stores have left the CU
    __shared__ int s_last;
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const unsigned old = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = old + 1u == gridDim.x;
        if (last) {
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);          // for the next launch
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // holds the barrier below until the invalidate is complete
        }
        s_last = last;
    }
    __syncthreads();
    return s_last != 0;
}

// Block-wide sums of the 2 NC + 1 accumulators of the panel kernels:
//   out[c] = (sum ar[c], sum ai[c]) for c < NC, and *nrm_out = sum nrm (if nrm_out != nullptr).
// Fixed evaluation order => bitwise reproducible.  Must be called by all 256 threads.
// `out` is this workgroup's partial row, which k_reduce sums.
template <int NC>
__device__ __forceinline__ void block_reduce_panel(const double (&ar)[NC], const double (&ai)[NC], double nrm,
                                                   c128 *__restrict__ out, c128 *__restrict__ nrm_out) {
    constexpr int K = 2 * NC + 1;
    constexpr int ROUNDS = (K + 63) / 64;
    constexpr int KP = ROUNDS == 1 ? next_pow2(K) : 64;   // values per round
    __shared__ double red[WAVES][ROUNDS * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        double v[KP];
#pragma unroll
        for (int i = 0; i < KP; ++i) {
            const int idx = r * 64 + i;   // constant after unrolling: static register selection
            v[i] = idx < 2 * NC ? ((idx & 1) ? ai[idx >> 1 < NC ? idx >> 1 : 0] : ar[idx >> 1 < NC ? idx >> 1 : 0])
                                : (idx == 2 * NC ? nrm : 0.0);
        }
        transpose_reduce_step<KP, 0>(v, lane);
        constexpr int G = 64 / KP;       // lanes that end up with the same value
        if ((lane & (G - 1)) == 0) red[wave][r * 64 + lane / G] = v[0];
    }
    __syncthreads();
    const int t = threadIdx.x;
    if (t < NC) {
        double sr = 0.0, si = 0.0;
#pragma unroll
        for (int k = 0; k < WAVES; ++k) { sr += red[k][2 * t]; si += red[k][2 * t + 1]; }
        out[t] = make_double2(sr, si);
    }
    if (nrm_out != nullptr && t == 64) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < WAVES; ++k) s += red[k][2 * NC];
        *nrm_out = make_double2(s, 0.0);
    }
}

// Deferred normalisation (AKS_EXPAND_DEFER_SCALE).  The reference normalises a new basis vector at once
// (decomposition.py:66, w /= beta): a pass of 32 n bytes per Arnoldi step that does nothing but scale.  Here a column
// may stay RAW instead -- it holds beta v, and its scale beta sits in the workspace (colscale[column]; 0 = the column
// is normalised) -- and every kernel that reads basis columns divides a raw column's entries by its scale as it loads
// them: the same IEEE division k_finish performs, so every value that enters an FMA is bit for bit the one the
// normalised column would have held.  Raw columns only exist between an expansion and the restart compression
// that follows it (aks_truncate_ws writes normalised columns and clears the scales).
__device__ __forceinline__ bool is_raw(double s) { return __builtin_bit_cast(long long, s) != 0ll; }
__device__ __forceinline__ c128 unscale(c128 v, double s) { return make_double2(v.x / s, v.y / s); }
__device__ __forceinline__ double unscale(double v, double s) { return v / s; }

__device__ __forceinline__ bool second_pass_needed(const c128 *red1, const c128 *red2, int J, double eta) {
    // ortho.py:101   beta < beta_before * eta
    return sqrt(red2[J].x) < sqrt(red1[J].x) * eta;
}

// ------------------------------------------------------------------ projection
// partial[bx*ldp + c0 + c] = sum over this block's rows of conj(V[i, c0+c]) * w[i]
// partial[bx*ldp + nrm_slot] = sum |w[i]|^2                    (only if nrm_slot >= 0)
template <int NC>
__global__ __launch_bounds__(BLOCK) void k_proj(int64_t n, int c0, const c128 *__restrict__ V, int64_t ldv,
                                               const c128 *__restrict__ w, c128 *partial,
                                               int ldp, int nrm_slot, const aks_ctrl *__restrict__ ctrl,
                                               const double *__restrict__ cs, int raw0) {
    if (ctrl->broken) return;
    __shared__ double ssc[NC];                            // scales of this group's columns (raw columns: >= raw0)
    if (threadIdx.x < NC) ssc[threadIdx.x] = cs[c0 + threadIdx.x];
    __syncthreads();
    double ar[NC], ai[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) ar[c] = ai[c] = 0.0;
    double nrm = 0.0;
    const c128 *Vc = V + (int64_t)c0 * ldv;
    const int64_t stride = (int64_t)gridDim.x * BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
        const c128 wv = w[i];
        c128 v[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) v[c] = ld_panel(&Vc[i + (int64_t)c * ldv]);
#pragma unroll
        for (int c = 0; c < NC; ++c)
            if (c0 + c >= raw0 && is_raw(ssc[c])) v[c] = unscale(v[c], ssc[c]);      // (wave-uniform)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            ar[c] = fma(v[c].x, wv.x, fma(v[c].y, wv.y, ar[c]));
            ai[c] = fma(v[c].x, wv.y, fma(-v[c].y, wv.x, ai[c]));
        }
        nrm = fma(wv.x, wv.x, fma(wv.y, wv.y, nrm));
    }
    c128 *row = partial + (int64_t)blockIdx.x * ldp;
    block_reduce_panel<NC>(ar, ai, nrm, row + c0, nrm_slot >= 0 ? row + nrm_slot : nullptr);
}

// ------------------------------------------------------------------ fused update + re-projection
// w[i] -= sum_c V[i,c] h[c];  partial[.., c] = sum conj(V[i,c]) w'[i];  partial[.., NC] = sum |w'[i]|^2
#ifndef AKS_UPD_HS_LDS_FROM
#define AKS_UPD_HS_LDS_FROM 18   // widths above this re-read the coefficients from LDS per tile (see the loop)
#endif
template <int NC>
__global__ __launch_bounds__(BLOCK) void k_update_proj(int64_t n, const c128 *__restrict__ V, int64_t ldv,
                                                      c128 *__restrict__ w, const c128 *__restrict__ h,
                                                      c128 *partial, int ldp,
                                                      const aks_ctrl *__restrict__ ctrl,
                                                      const double *__restrict__ cs, int raw0) {
    if (ctrl->broken) return;
    __shared__ c128 hs[NC];
    __shared__ double ssc[NC];
    if (threadIdx.x < NC) { hs[threadIdx.x] = h[threadIdx.x]; ssc[threadIdx.x] = cs[threadIdx.x]; }
    __syncthreads();
    double ar[NC], ai[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) ar[c] = ai[c] = 0.0;
    double nrm = 0.0;
    const int64_t stride = (int64_t)gridDim.x * BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
        // Keeps the NC coefficients in LDS (re-read per tile as broadcast ds_reads) instead of letting
        // the compiler hoist them into 4 NC VGPRs for the whole loop: that is the difference between
        // one and two waves per SIMD for NC = 20..28 (A/B at n = 1M, J = 28: 0.147 -> 0.102 ms).
        if constexpr (NC > AKS_UPD_HS_LDS_FROM) asm volatile("" ::: "memory");
        c128 wv = w[i];
        c128 v[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) v[c] = ld_panel(&V[i + (int64_t)c * ldv]);
#pragma unroll
        for (int c = 0; c < NC; ++c)
            if (c >= raw0 && is_raw(ssc[c])) v[c] = unscale(v[c], ssc[c]);           // (wave-uniform)
        double sr = 0.0, si = 0.0;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const c128 hc = hs[c];
            sr = fma(v[c].x, hc.x, fma(-v[c].y, hc.y, sr));
            si = fma(v[c].x, hc.y, fma(v[c].y, hc.x, si));
        }
        wv.x -= sr;
        wv.y -= si;
        w[i] = wv;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            ar[c] = fma(v[c].x, wv.x, fma(v[c].y, wv.y, ar[c]));
            ai[c] = fma(v[c].x, wv.y, fma(-v[c].y, wv.x, ai[c]));
        }
        nrm = fma(wv.x, wv.x, fma(wv.y, wv.y, nrm));
    }
    c128 *row = partial + (int64_t)blockIdx.x * ldp;
    block_reduce_panel<NC>(ar, ai, nrm, row, row + NC);
}

// ------------------------------------------------------------------ fused update + re-projection, wide panels
// Same operation as k_update_proj for J > 20: the J columns are dealt to the 4 waves of the block
// (NQ = ceil(J / 4) per wave, all four on the SAME 64 rows), so a wave holds 8 NQ instead of 8 J
// VGPRs and several waves per SIMD overlap loads with arithmetic.  The row's update needs all
// columns: each wave puts its partial  sum_c V[i,c] h[c]  in LDS, one barrier per 64-row tile
// (double-buffered), and every wave adds the four partials in the same order.  The re-projection
// accumulators of a wave belong to its own columns, so no cross-wave sum is needed at the end.
template <int NQ>
__global__ __launch_bounds__(BLOCK) void k_update_proj_split(int64_t n, int J, const c128 *__restrict__ V,
                                                            int64_t ldv, c128 *__restrict__ w,
                                                            const c128 *__restrict__ h, c128 *__restrict__ partial,
                                                            int ldp, const aks_ctrl *__restrict__ ctrl,
                                                            const double *__restrict__ cs, int raw0) {
    if (ctrl->broken) return;
    __shared__ c128 hs[AKS_MAX_DIM + 4];
    __shared__ double ssc[AKS_MAX_DIM + 4];
    __shared__ double ps_re[2][WAVES][64], ps_im[2][WAVES][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int base = J / WAVES, extra = J % WAVES;
    const int c_begin = wave * base + min(wave, extra);
    const int cnt = base + (wave < extra ? 1 : 0);      // <= NQ
    for (int c = threadIdx.x; c < J; c += BLOCK) { hs[c] = h[c]; ssc[c] = cs[c]; }
    __syncthreads();
    if (threadIdx.x < 4) hs[J + threadIdx.x] = make_double2(0.0, 0.0);   // coefficient of padding slots
    __syncthreads();
    const c128 *hq = hs + c_begin;                     // this wave's coefficients, re-read per tile
    const int pad = J - c_begin;                       // hq[pad] == 0
    double ar[NQ], ai[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) ar[q] = ai[q] = 0.0;
    double nrm = 0.0;
    const c128 *Vw = V + (int64_t)c_begin * ldv;
    const int64_t n_tiles = (n + 63) / 64;
    int buf = 0;
    for (int64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {   // same trip count for the 4 waves
        const int64_t row = t * 64 + lane;
        const bool valid = row < n;
        const int64_t i = valid ? row : n - 1;
        asm volatile("" ::: "memory");                 // keep the coefficients in LDS, not in 4 NQ VGPRs
        c128 wv = w[i];
        c128 v[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) v[q] = ld_panel(&Vw[i + (int64_t)min(q, cnt - 1) * ldv]);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int cq = c_begin + min(q, cnt - 1);                                // (wave-uniform)
            if (cq >= raw0 && is_raw(ssc[cq])) v[q] = unscale(v[q], ssc[cq]);
        }
        double sr = 0.0, si = 0.0;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const c128 hc = hq[q < cnt ? q : pad];
            sr = fma(v[q].x, hc.x, fma(-v[q].y, hc.y, sr));
            si = fma(v[q].x, hc.y, fma(v[q].y, hc.x, si));
        }
        ps_re[buf][wave][lane] = sr;
        ps_im[buf][wave][lane] = si;
        __syncthreads();
        double tr = 0.0, ti = 0.0;
#pragma unroll
        for (int k = 0; k < WAVES; ++k) { tr += ps_re[buf][k][lane]; ti += ps_im[buf][k][lane]; }
        wv.x -= tr;
        wv.y -= ti;
        if (wave == 0 && valid) w[row] = wv;
        if (!valid) wv = make_double2(0.0, 0.0);       // rows past the end contribute nothing
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            ar[q] = fma(v[q].x, wv.x, fma(v[q].y, wv.y, ar[q]));
            ai[q] = fma(v[q].x, wv.y, fma(-v[q].y, wv.x, ai[q]));
        }
        nrm = fma(wv.x, wv.x, fma(wv.y, wv.y, nrm));
        buf ^= 1;
    }
    // per-wave totals of this wave's columns (+ ||w'||^2, reported by wave 0)
    constexpr int K = 2 * NQ + 1;
    constexpr int ROUNDS = (K + 63) / 64;
    constexpr int KP = ROUNDS == 1 ? next_pow2(K) : 64;
    c128 *prow = partial + (int64_t)blockIdx.x * ldp;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        double x[KP];
#pragma unroll
        for (int e = 0; e < KP; ++e) {
            const int idx = r * 64 + e;
            x[e] = idx < 2 * NQ ? ((idx & 1) ? ai[idx >> 1 < NQ ? idx >> 1 : 0] : ar[idx >> 1 < NQ ? idx >> 1 : 0])
                                : (idx == 2 * NQ ? nrm : 0.0);
        }
        transpose_reduce_step<KP, 0>(x, lane);
        constexpr int G = 64 / KP;
        const int idx = r * 64 + lane / G;              // value this lane ended up with
        if ((lane & (G - 1)) == 0) {
            double *out = reinterpret_cast<double *>(prow);
            if (idx < 2 * NQ) {
                const int q = idx >> 1;
                if (q < cnt) out[2 * (c_begin + q) + (idx & 1)] = x[0];
            } else if (idx == 2 * NQ && wave == 0) {
                prow[J] = make_double2(x[0], 0.0);
            }
        }
    }
}

// ------------------------------------------------------------------ the step's book-keeping
// h -> H[0:J, j], beta, breakdown test, counters (ortho.py:95,103,107; decomposition.py:61-65).  Called by ALL threads
// of ONE workgroup; no barrier inside.  normalize as in k_finish (the n-sized division of mode 1 is k_finish's own).
// `twice` / `nrm3`: whether the step ran the second pass, and ||w''||^2 if it did.
__device__ __forceinline__ bool finish_step(int J, c128 *__restrict__ Hcol, int64_t ldh, double tol, int normalize,
                                            const c128 *red1, const c128 *red2, bool twice, double nrm3,
                                            aks_ctrl *ctrl, double *__restrict__ cs) {
    const double beta = sqrt(twice ? nrm3 : red2[J].x);
    const bool broke = beta < tol;  // ortho.py:107
    for (int c = threadIdx.x; c < J; c += BLOCK) {
        c128 hv = red1[c];  // ortho.py:95
        if (twice) { hv.x += red2[c].x; hv.y += red2[c].y; }  // ortho.py:103
        Hcol[(int64_t)c * ldh] = hv;
    }
    if (threadIdx.x == 0) {
        if (!broke && normalize) Hcol[(int64_t)J * ldh] = make_double2(beta, 0.0);  // decomposition.py:65
        if (!broke && normalize == 2) cs[J] = beta;
        // a breakdown ends the expansion: no column from J on is a raw basis column, whatever an earlier, discarded
        // attempt at this expansion (lazy third all-reduce) may have booked there
        if (broke && normalize == 2)
            for (int c = J; c < AKS_MAX_DIM + 2; ++c) cs[c] = 0.0;
        ctrl->deferred = normalize == 2 ? 1 : 0;
        ctrl->beta_in = sqrt(red1[J].x);
        ctrl->beta = beta;
        ctrl->steps_done += 1;
        ctrl->second_passes += twice ? 1 : 0;
        if (broke) { ctrl->n_iter = J; ctrl->broken = 1; }  // decomposition.py:61-63 (n_iter = j+1 = J)
    }
    return broke;
}

// What the second-pass kernels do about the step's book-keeping (it needs the LAST norm of the step, which only they
// -- or nobody, when the DGKS test does not fire -- produce):
//   FIN_NONE      nothing: k_finish follows (the public stage entry points; normalize == 1, whose n-sized division
//                 is a kernel of its own anyway)
//   FIN_ALWAYS    the step ends here: workgroup 0 books it when no second pass is needed, the last workgroup of the
//                 second pass otherwise (one GPU, or several with the lazy third all-reduce)
//   FIN_IF_ONCE   only when no second pass is needed; after a second pass the norm has to be summed over the ranks
//                 first, so k_finish follows (and does nothing if the step was booked here: its `only_if_twice`)
enum { FIN_NONE = 0, FIN_ALWAYS = 1, FIN_IF_ONCE = 2 };
struct FinArgs {
    c128 *Hcol;
    int64_t ldh;
    double tol;
    double *cs;
    c128 *red3;              // where the norm after the second pass goes (always written when the pass runs)
    unsigned *ticket;        // nullptr: the block partials are summed by k_reduce<true> (no in-kernel second stage)
    int normalize, mode;
};

// Common ending of the second-pass kernels: this workgroup's norm partial -> [last workgroup: sum, red3, book-keeping].
__device__ __forceinline__ void second_pass_tail(double block_nrm, c128 *partial, int ldp, int nrm_slot, int J,
                                                 const c128 *red1, const c128 *red2, aks_ctrl *ctrl, const FinArgs &fin,
                                                 double *scratch) {
    if (threadIdx.x == 0) st_partial(&partial[(int64_t)blockIdx.x * ldp + nrm_slot], block_nrm, 0.0);
    if (fin.ticket == nullptr) return;
    if (!last_block_arrives(fin.ticket)) return;
    double sr = 0.0;                                     // k_reduce<true>'s sum: thread t takes rows t, t + 256, ...
    for (int b = threadIdx.x; b < (int)gridDim.x; b += BLOCK) sr += ld_partial(&partial[(int64_t)b * ldp + nrm_slot]).x;
    sr = block_sum(sr, scratch);                         // (valid in thread 0)
    __shared__ double s_nrm3;
    if (threadIdx.x == 0) { s_nrm3 = sr; fin.red3[0] = make_double2(sr, 0.0); }
    __syncthreads();
    if (fin.mode == FIN_ALWAYS) finish_step(J, fin.Hcol, fin.ldh, fin.tol, fin.normalize, red1, red2, true, s_nrm3, ctrl, fin.cs);
}

// ------------------------------------------------------------------ update (any width)
// w[i] -= sum_{c<J} V[i,c] h[c];  partial[bx*ldp + nrm_slot] = sum |w'[i]|^2.
// PRED: run only if the DGKS test asks for the second pass (ortho.py:101).
template <bool PRED>
__global__ __launch_bounds__(BLOCK) void k_update(int64_t n, int J, const c128 *__restrict__ V, int64_t ldv,
                                                 c128 *__restrict__ w, const c128 *__restrict__ h,
                                                 c128 *partial, int ldp, int nrm_slot,
                                                 const c128 *red1, const c128 *red2,
                                                 double eta, aks_ctrl *ctrl,
                                                 const double *__restrict__ cs, int raw0, FinArgs fin) {
    if (ctrl->broken) return;
    if (PRED && !second_pass_needed(red1, red2, J, eta)) {
        if (fin.mode != FIN_NONE && blockIdx.x == 0)
            finish_step(J, fin.Hcol, fin.ldh, fin.tol, fin.normalize, red1, red2, false, 0.0, ctrl, fin.cs);
        return;
    }
    __shared__ c128 hs[AKS_MAX_DIM + 8];
    __shared__ double ssc[AKS_MAX_DIM + 8];
    __shared__ double red_n[WAVES];
    const int Jpad = (J + 3) & ~3;
    for (int c = threadIdx.x; c < Jpad; c += BLOCK) {
        hs[c] = c < J ? h[c] : make_double2(0.0, 0.0);
        ssc[c] = c < J ? cs[c] : 0.0;
    }
    __syncthreads();
    double nrm = 0.0;
    const int64_t stride = (int64_t)gridDim.x * BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
        c128 wv = w[i];
        double sr = 0.0, si = 0.0;
        for (int c = 0; c < Jpad; c += 4) {
            c128 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int cc = min(c + u, J - 1);  // padded columns re-read a valid one; their h is 0
                v[u] = ld_panel(&V[i + (int64_t)cc * ldv]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int cc = min(c + u, J - 1);
                if (cc >= raw0 && is_raw(ssc[cc])) v[u] = unscale(v[u], ssc[cc]);    // (wave-uniform)
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const c128 hc = hs[c + u];
                sr = fma(v[u].x, hc.x, fma(-v[u].y, hc.y, sr));
                si = fma(v[u].x, hc.y, fma(v[u].y, hc.x, si));
            }
        }
        wv.x -= sr;
        wv.y -= si;
        w[i] = wv;
        nrm = fma(wv.x, wv.x, fma(wv.y, wv.y, nrm));
    }
    const double s = block_sum(nrm, red_n);
    second_pass_tail(s, partial, ldp, nrm_slot, J, red1, red2, ctrl, fin, red_n);
}

// (An exact-width variant of this kernel -- the panel width a template parameter, all J column loads of a row in flight
// at once -- was measured 0 - 2 % BEHIND this generic one at every width and is not in this file: the second pass is a
// read stream with one rewritten column and sits at that pattern's rate with four loads in flight per lane as well as
// with forty.  profiles/r04_update_nc_ab.txt, profiles/r05_removed_variants.patch.)

// ------------------------------------------------------------------ second-stage reduction
// out[c] = sum_b partial[b*ldp + first + c], c = blockIdx.x < count; fixed order => reproducible.
// PRED as in k_update (skips when no second pass ran, leaving `out` untouched).
template <bool PRED>
__global__ __launch_bounds__(BLOCK) void k_reduce(const c128 *__restrict__ partial, int n_blocks, int ldp,
                                                 int first, c128 *__restrict__ out, int J,
                                                 const c128 *__restrict__ red1, const c128 *__restrict__ red2,
                                                 double eta, c128 *__restrict__ zero_slot,
                                                 const aks_ctrl *__restrict__ ctrl) {
    if (ctrl->broken) return;
    if (PRED && !second_pass_needed(red1, red2, J, eta)) return;
    // keeps the (unconditionally all-reduced) red3 slot bounded when no second pass overwrites it
    if (zero_slot != nullptr && blockIdx.x == 0 && threadIdx.x == 0) zero_slot[0] = make_double2(0.0, 0.0);
    __shared__ double sc[WAVES];
    const int c = blockIdx.x;
    double sr = 0.0, si = 0.0;
    for (int b = threadIdx.x; b < n_blocks; b += BLOCK) {
        const c128 p = partial[(int64_t)b * ldp + first + c];
        sr += p.x;
        si += p.y;
    }
    sr = block_sum(sr, sc);
    si = block_sum(si, sc);
    // real-packed mode (aks_workspace_set_real): the panel holds REAL vectors, two rows per complex slot;
    // Re(V^H w) is the real dot product and the coefficients used downstream must be real
    if (threadIdx.x == 0) out[c] = make_double2(sr, ctrl->real_mode ? 0.0 : si);
}

// ------------------------------------------------------------------ finish
__global__ __launch_bounds__(BLOCK) void k_finish(int64_t n, int J, c128 *__restrict__ w, c128 *__restrict__ Hcol,
                                                 int64_t ldh, double tol, double eta, int normalize,
                                                 const c128 *__restrict__ red1, const c128 *__restrict__ red2,
                                                 const c128 *__restrict__ red3, aks_ctrl *ctrl,
                                                 double *__restrict__ cs, int only_if_twice) {
    // normalize: 0 = leave w as it is; 1 = H[J, j] = beta and w /= beta (decomposition.py:65-66); 2 = H[J, j] = beta
    // and column J stays raw with colscale[J] = beta (deferred normalisation: its readers divide)
    // only_if_twice: a step that needed no second pass has been booked by the second-pass kernel (FIN_IF_ONCE)
    if (ctrl->broken) return;
    const bool twice = second_pass_needed(red1, red2, J, eta);
    if (only_if_twice && !twice) return;
    const double beta = sqrt(twice ? red3[0].x : red2[J].x);
    const bool broke = beta < tol;  // ortho.py:107
    if (blockIdx.x == 0) finish_step(J, Hcol, ldh, tol, normalize, red1, red2, twice, red3[0].x, ctrl, cs);
    if (broke || normalize != 1) return;
    const int64_t stride = (int64_t)gridDim.x * BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
        c128 wv = w[i];
        wv.x /= beta;  // decomposition.py:66 (true division, as numpy's complex / real)
        wv.y /= beta;
        w[i] = wv;
    }
}

// ------------------------------------------------------------------ CSR-stream SpMV
constexpr int TILE = AKS_SPMV_TILE_NNZ;  // non-zeros per wave tile
constexpr int NPT = TILE / 64;           // per lane

__device__ __forceinline__ c128 cmul(double a, c128 x) { return make_double2(a * x.x, a * x.y); }
__device__ __forceinline__ double cmul(double a, double x) { return a * x; }
__device__ __forceinline__ c128 cmul(c128 a, c128 x) {
    return make_double2(fma(a.x, x.x, -a.y * x.y), fma(a.x, x.y, a.y * x.x));
}

// Index / value streams of the CSR kernel are read once; x must stay cached.  AKS_NT_CSR=1: non-temporal stream loads.
#ifndef AKS_NT_CSR
#define AKS_NT_CSR 0
#endif
__device__ __forceinline__ int ld_stream(const int32_t *p) { return AKS_NT_CSR ? __builtin_nontemporal_load(p) : *p; }
__device__ __forceinline__ double ld_stream(const double *p) { return AKS_NT_CSR ? __builtin_nontemporal_load(p) : *p; }
__device__ __forceinline__ c128 ld_stream(const c128 *p) {
#if AKS_NT_CSR
    const v2d_t v = __builtin_nontemporal_load(reinterpret_cast<const v2d_t *>(p));
    return make_double2(v.x, v.y);
#else
    return *p;
#endif
}

// XT: vector entry type -- c128, or double for real vectors (real-packed mode: 8-byte gathers, one LDS plane).
__device__ __forceinline__ double xt_re(c128 v) { return v.x; }
__device__ __forceinline__ double xt_im(c128 v) { return v.y; }
__device__ __forceinline__ double xt_re(double v) { return v; }
__device__ __forceinline__ double xt_im(double) { return 0.0; }
__device__ __forceinline__ void xt_make(double re, double im, c128 *out) { *out = make_double2(re, im); }
__device__ __forceinline__ void xt_make(double re, double, double *out) { *out = re; }
__device__ __forceinline__ c128 xt_add(c128 a, c128 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double xt_add(double a, double b) { return a + b; }

template <typename VT, typename XT, bool ACC>
__global__ __launch_bounds__(BLOCK) void k_spmv(int64_t n_tiles, const int32_t *__restrict__ indptr,
                                               const int32_t *__restrict__ indices, const VT *__restrict__ vals,
                                               const int32_t *__restrict__ tiles, int lpr,
                                               const XT *__restrict__ x, XT *__restrict__ y,
                                               const aks_ctrl *__restrict__ ctrl) {
    if (ctrl != nullptr && ctrl->broken) return;
    constexpr bool CPLX = sizeof(XT) == sizeof(c128);
    __shared__ double pr[WAVES][TILE], pi[CPLX ? WAVES : 1][CPLX ? TILE : 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t t = (int64_t)blockIdx.x * WAVES + wave;
    if (t >= n_tiles) return;  // waves are independent: no block-level barrier below
    const int r0 = tiles[t], r1 = tiles[t + 1];
    const int k0 = indptr[r0], k1 = indptr[r1];
    const int nnz = k1 - k0;
    if (nnz <= TILE) {
        if (nnz > 0) {
            int col[NPT];
            VT a[NPT];
            XT xv[NPT];
#pragma unroll
            for (int q = 0; q < NPT; ++q) {
                const int k = min(k0 + q * 64 + lane, k1 - 1);
                col[q] = ld_stream(&indices[k]);
                a[q] = ld_stream(&vals[k]);
            }
#pragma unroll
            for (int q = 0; q < NPT; ++q) xv[q] = x[col[q]];
#pragma unroll
            for (int q = 0; q < NPT; ++q) {
                const int s = q * 64 + lane;
                if (s < nnz) {
                    const XT p = cmul(a[q], xv[q]);
                    pr[wave][s] = xt_re(p);
                    if (CPLX) pi[wave][s] = xt_im(p);
                }
            }
        }
        // LDS hand-off inside one wave: order the ds_writes before the ds_reads.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int rows_per_pass = 64 / lpr;
        const int sub = lane % lpr;
        for (int rb = r0; rb < r1; rb += rows_per_pass) {
            const int r = rb + lane / lpr;
            double sr = 0.0, si = 0.0;
            if (r < r1) {
                const int a0 = indptr[r] - k0, a1 = indptr[r + 1] - k0;
                for (int s = a0 + sub; s < a1; s += lpr) {
                    sr += pr[wave][s];
                    if (CPLX) si += pi[wave][s];
                }
            }
            for (int off = lpr >> 1; off > 0; off >>= 1) {
                sr += __shfl_xor(sr, off, 64);
                if (CPLX) si += __shfl_xor(si, off, 64);
            }
            if (r < r1 && sub == 0) {
                if (ACC) {
                    const XT old = y[r];
                    sr += xt_re(old);
                    si += xt_im(old);
                }
                xt_make(sr, si, &y[r]);
            }
        }
    } else {
        // one long row per tile: the wave strides over it
        double sr = 0.0, si = 0.0;
        for (int k = k0 + lane; k < k1; k += 64) {
            const XT p = cmul(vals[k], x[indices[k]]);
            sr += xt_re(p);
            si += xt_im(p);
        }
        sr = wave_sum(sr);
        if (CPLX) si = wave_sum(si);
        if (lane == 0) {
            if (ACC) {
                const XT old = y[r0];
                sr += xt_re(old);
                si += xt_im(old);
            }
            xt_make(sr, si, &y[r0]);
        }
    }
}

// ------------------------------------------------------------------ sliced SpMV (regular matrices)
// One lane per row; a SLICE of 64 consecutive rows is stored entry-major (entry k of all 64 rows, then
// entry k + 1, ...), padded to the slice's longest row with column -1: a wave's value and column loads are
// contiguous, its x gathers are as local as the matrix is (neighbouring rows of a stencil or a band read
// neighbouring x), and a row is summed in column order in registers -- no LDS, no second pass.
// Workgroups with the same blockIdx % 8 share an XCD (observed placement; speed only): each XCD gets a
// contiguous eighth of the slices, so the x entries that rows a grid line or plane apart share are fetched
// into ONE L2 rather than into all eight (profiles/microbench/sell_spmv.txt: 3-D Laplace 0.400 -> 0.347 ms).
#ifndef AKS_SELL_U
#define AKS_SELL_U 4            // entries of a row in flight per lane (A/B: profiles/r05_sell_in_solve.txt)
#endif
constexpr int SELL_U = AKS_SELL_U;
#ifndef AKS_NT_SELL
#define AKS_NT_SELL 1            // non-temporal loads of the value / column streams (read once): 1-8 % (sell_spmv.txt)
#endif
__device__ __forceinline__ int ld_sell(const int32_t *p) { return AKS_NT_SELL ? __builtin_nontemporal_load(p) : *p; }
__device__ __forceinline__ double ld_sell(const double *p) { return AKS_NT_SELL ? __builtin_nontemporal_load(p) : *p; }
__device__ __forceinline__ c128 ld_sell(const c128 *p) {
#if AKS_NT_SELL
    typedef double nt_v2d __attribute__((ext_vector_type(2)));
    const nt_v2d v = __builtin_nontemporal_load(reinterpret_cast<const nt_v2d *>(p));
    return make_double2(v.x, v.y);
#else
    return *p;
#endif
}
template <typename VT, typename XT, bool ACC>
__global__ __launch_bounds__(BLOCK) void k_sell(int64_t n_rows, int64_t n_slices, const int64_t *__restrict__ slice_ptr,
                                               const int32_t *__restrict__ col, const VT *__restrict__ val,
                                               const XT *__restrict__ x, XT *__restrict__ y,
                                               const aks_ctrl *__restrict__ ctrl, const double *__restrict__ x_div) {
    if (ctrl != nullptr && ctrl->broken) return;
    // x may be a RAW basis column (deferred normalisation, only offered for short rows: sell_defers): every gathered
    // entry is divided by the column's scale -- the division k_finish would have done once per entry, here once per
    // non-zero, which a kernel that waits for memory has the cycles for as long as rows are short
    const double xd = x_div != nullptr ? *x_div : 0.0;
    const bool raw = is_raw(xd);
    const int lane = threadIdx.x & 63;
    const int64_t per_xcd = (gridDim.x + 7) >> 3;
    const int64_t wg = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    const int64_t slice = wg * WAVES + (threadIdx.x >> 6);
    if (slice >= n_slices) return;
    const int64_t p0 = slice_ptr[slice];
    const int W = (int)((slice_ptr[slice + 1] - p0) >> 6);
    const int32_t *c = col + p0 + lane;
    const VT *v = val + p0 + lane;
    XT acc;
    xt_make(0.0, 0.0, &acc);
    int k = 0;
    for (; k + SELL_U <= W; k += SELL_U) {
        int cc[SELL_U];
        VT vv[SELL_U];
        XT xx[SELL_U];
#pragma unroll
        for (int u = 0; u < SELL_U; ++u) { cc[u] = ld_sell(&c[(int64_t)(k + u) * 64]); vv[u] = ld_sell(&v[(int64_t)(k + u) * 64]); }
#pragma unroll
        for (int u = 0; u < SELL_U; ++u) xx[u] = x[max(cc[u], 0)];
        if (raw) {
#pragma unroll
            for (int u = 0; u < SELL_U; ++u) xx[u] = unscale(xx[u], xd);
        }
#pragma unroll
        for (int u = 0; u < SELL_U; ++u)
            if (cc[u] >= 0) acc = xt_add(acc, cmul(vv[u], xx[u]));
    }
    for (; k < W; ++k) {
        const int cc = ld_sell(&c[(int64_t)k * 64]);
        const VT vv = ld_sell(&v[(int64_t)k * 64]);
        XT xx = x[max(cc, 0)];
        if (raw) xx = unscale(xx, xd);
        if (cc >= 0) acc = xt_add(acc, cmul(vv, xx));
    }
    const int64_t row = slice * 64 + lane;
    if (row < n_rows) {
        if (ACC) acc = xt_add(acc, y[row]);
        y[row] = acc;
    }
}

// ------------------------------------------------------------------ restart compression
// V[:, :p] = V[:, :m] Qp in place and V[:, p] = V[:, m] (krylov_schur.py:78,81) on the matrix cores
// (v_mfma_f64_16x16x4_f64): the one place on this path where a real K-dimension -- the m basis
// columns -- is summed.  Complex arithmetic as a real GEMM
//     [Cr Ci] = [Vr Vi] . [[Qr Qi], [-Qi Qr]]        (n x 2m) . (2m x 2p)
// computed transposed, D = A.B with  M = 16 real OUTPUT columns (8 complex), N = 16 rows of V,
// K = 4 complex input columns per load: lane (g = l>>4, r = l&15) loads the 16-byte element
// V[row r, column c0+g] once and feeds two MFMAs -- its real part against row 2(c0+g) of the real
// Q block, its imaginary part against row 2(c0+g)+1 -- so no cross-lane shuffles are needed
// (the K order inside an MFMA is free as long as A and B agree).  The D layout of the f64 MFMA
// (row = (l>>4) + 4 reg, col = l&15) puts 16 consecutive rows of V on consecutive lanes, so the
// stores are 256-byte contiguous runs like the loads.  One wave owns 64 rows (4 N-tiles) and all MT
// M-tiles; it reads all m columns of its rows before it overwrites any, hence in place.
typedef double v4d __attribute__((ext_vector_type(4)));

// The same kernel also forms Ritz vectors out of place (aks_combine: O != V, no column copy).
template <int MT, int NS>   // NS row sub-tiles of 16 per wave (4, or 2 when MT is large)
__global__ __launch_bounds__(BLOCK) void k_truncate_mfma(int64_t n, int m, int p, c128 *V, int64_t ldv,
                                                        const c128 *__restrict__ Qp, c128 *O, int64_t ldo,
                                                        int copy_last) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    c128 *qs = reinterpret_cast<c128 *>(smem_raw);      // [mpad][PP] complex, zero padded
    constexpr int PP = MT * 8;
    const int mpad = (m + 3) & ~3;
    for (int e = threadIdx.x; e < mpad * PP; e += BLOCK) {
        const int k = e / PP, c = e - k * PP;
        qs[e] = (k < m && c < p) ? Qp[(int64_t)k * p + c] : make_double2(0.0, 0.0);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, r16 = lane & 15;
    const int comp = r16 & 1;                            // this lane's real output column is Re (0) / Im (1)
    constexpr int ROWS = 16 * NS;
    const int64_t n_tiles = (n + ROWS - 1) / ROWS;
    for (int64_t t = (int64_t)blockIdx.x * WAVES + wave; t < n_tiles; t += (int64_t)gridDim.x * WAVES) {
        const int64_t row0 = t * ROWS;
        v4d acc[NS][MT];
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[s][mt] = (v4d){0.0, 0.0, 0.0, 0.0};
        int64_t rows[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) rows[s] = min(row0 + s * 16 + r16, n - 1);
        for (int c0 = 0; c0 < mpad; c0 += 4) {
            const int c = min(c0 + g, m - 1);            // padded K slots re-read a valid column; their Q rows are 0
            c128 v[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
#if AKS_NT_TRUNC
                v[s] = ld_panel(&V[rows[s] + (int64_t)c * ldv]);
#else
                v[s] = V[rows[s] + (int64_t)c * ldv];
#endif
            }
            const c128 *qrow = qs + (c0 + g) * PP;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const c128 q = qrow[mt * 8 + (r16 >> 1)];
                const double a_re = comp ? q.y : q.x;    // row 2c   of [[Qr Qi], [-Qi Qr]]
                const double a_im = comp ? q.x : -q.y;   // row 2c+1
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    acc[s][mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_re, v[s].x, acc[s][mt], 0, 0, 0);
                    acc[s][mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_im, v[s].y, acc[s][mt], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int64_t row = row0 + s * 16 + r16;
            if (row < n) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int jj = mt * 16 + g + 4 * r;      // real output column held in register r
                        const int j = jj >> 1;
                        if (j < p) {
                            double *dst = reinterpret_cast<double *>(O + row + (int64_t)j * ldo) + (jj & 1);
#if AKS_NT_TRUNC_STORE
                            __builtin_nontemporal_store(acc[s][mt][r], dst);
#else
                            *dst = acc[s][mt][r];
#endif
                        }
                    }
                }
            }
        }
        // V[:, p] = V[:, m] (krylov_schur.py:81).  Column m is never a destination of the stores
        // above (they end at column p - 1 <= m - 1), so it can be read here.
        if (copy_last && g == 0) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int64_t row = row0 + s * 16 + r16;
                if (row < n) O[row + (int64_t)p * ldo] = V[row + (int64_t)m * ldv];
            }
        }
    }
}

// after a restart compression of columns [0, m] into [0, p]: the p new columns are normalised; column p is a bit
// copy of column m and inherits its scale; columns behind it are dead
// (Round 4: a real bug of round 3's deferred normalisation lived here.  The one-thread form  sm = cs[m]; clear cs[0..m];
// cs[p] = sm  compiled to an s_load (scalar cache path) of cs[m] followed by vector stores that clear cs[m] WITHOUT waiting
// for the scalar load -- the two paths are not ordered by the hardware, so when the load missed the scalar cache the
// stores overtook it, the carried scale read as 0 and column p -- a raw column -- was taken for a normalised one by the
// next expansion: errors of 1e-3 .. 1e-1 in H, a few times in a hundred restarts, timing-dependent (found at full size:
// tests/thread_ranks_worker.py --case repro; the 2.2e-3 residual of a one-GPU config-5 solve was the same bug).  Now:
// every lane LOADS with a vector load, the workgroup's barrier waits for all of them, then lane c stores column c's
// scale.)
__global__ __launch_bounds__(BLOCK) void k_colscale_after_truncate(double *cs, int m, int p) {
    const double sm = __hip_atomic_load(&cs[m], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // global_load (not s_load)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    const int c = threadIdx.x;
    if (blockIdx.x == 0 && c <= m) cs[c] = c == p ? sm : 0.0;
}

// w *= alpha (the normalisation at the end of the explicit-restart solvers' mgs, explicit_restarts.py:77)
__global__ __launch_bounds__(BLOCK) void k_scale(int64_t n, c128 *__restrict__ w, c128 alpha) {
    const int64_t stride = (int64_t)gridDim.x * BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) w[i] = cmul(alpha, w[i]);
}

// dst[i] = src[i] over 16-byte items, non-temporal both ways: the plain read + write stream every panel kernel is a variant
// of.  bench.py times 50 launches of it next to the headline ("calibration"): restarts/s divided by this box's streaming
// rate is comparable between boxes whose clocks / power states differ (a +-4 % move must be attributable).
__global__ __launch_bounds__(BLOCK) void k_stream_copy(int64_t items, const c128 *__restrict__ src, c128 *__restrict__ dst) {
    const int64_t stride = (int64_t)gridDim.x * BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < items; i += stride) {
        const c128 v = ld_panel(src + i);
        v2d_t t; t.x = v.x; t.y = v.y;
        __builtin_nontemporal_store(t, reinterpret_cast<v2d_t *>(dst + i));
    }
}

template <typename T>
__global__ __launch_bounds__(BLOCK) void k_gather(int64_t count, const int32_t *__restrict__ idx,
                                                 const T *__restrict__ src, T *__restrict__ dst,
                                                 const double *__restrict__ div) {
    const double d = div != nullptr ? *div : 0.0;          // the source column is raw: pack normalised entries
    const int64_t stride = (int64_t)gridDim.x * BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < count; i += stride) {
        T v = src[idx[i]];
        if (is_raw(d)) v = unscale(v, d);
        dst[i] = v;
    }
}

// ------------------------------------------------------------------ tile-binned two-phase SpMV
// (include/arnoldi_hip.h, "tile-binned two-phase SpMV"; measurements behind the shape of both kernels:
// profiles/microbench/tile_binned_spmv.hip, vmem_issue_cost.hip)
// columns per sub-slab: 2^AKS_PB_SLAB_BITS, or -DAKS_PB_SLAB_COLS=<any count <= 10240> for experiments (10240 complex
// entries fill the 160 KiB of LDS: 1.25 x longer pieces for phase 2, profiles/r04_pb_bytes_ab.txt)
#ifndef AKS_PB_SLAB_COLS
#define AKS_PB_SLAB_COLS (1 << AKS_PB_SLAB_BITS)
#endif
constexpr int PB_CW_BITS = AKS_PB_SLAB_BITS, PB_CW = AKS_PB_SLAB_COLS;
constexpr int PB_RB_BITS = AKS_PB_ROWBLOCK_BITS, PB_RB = 1 << PB_RB_BITS;   // rows per row block
constexpr int PB_W = AKS_PB_WAVES, PB_K = AKS_PB_RUNS_PER_WAVE, PB_RPR = PB_W * PB_K;
#ifndef AKS_PB_DEPTH
#define AKS_PB_DEPTH 4
#endif
constexpr int PB_D = AKS_PB_DEPTH;                      // pipeline stages per wave: a round is loaded PB_D - 1 steps before its adds
constexpr int PB_B = 8;                      // rounds whose descriptors one vector load fetches (PB_B * PB_K lanes)
constexpr int PB_P1_THREADS = 1024, PB_P1_U = 4;
constexpr int PB_RW = AKS_PB_ROUND_WORDS;    // (level, row) words of one round: [wave][lane][k]
constexpr int PB_MAX_LEVELS = 8;             // 3-bit level field next to the 13-bit row
#ifndef AKS_PB_TICKS
#define AKS_PB_TICKS 0           // diagnostic build: wave 0 of every phase-2 workgroup leaves s_memtime sums per pipeline
#endif                           // section in the first doubles of the product scratch (pb_abi_bench prints them)
static_assert(PB_CW <= 65536 && PB_RB_BITS <= 13, "lcol is a 16-bit, lrow a 13-bit field");
static_assert(PB_B % PB_D == 0 && PB_B * PB_K <= 64, "descriptor block: a multiple of the depth, one lane per slot");
static_assert((PB_K == 4 || PB_K == 8) && AKS_PB_RUN_MAX == 64, "a lane's words of a round are one 8- or 16-byte load");
struct alignas(PB_K * 2) PbWords { unsigned v[PB_K / 2]; };     // a lane's PB_K (level, row) words of one round
static_assert(PB_W <= PB_MAX_LEVELS, "a level counts waves");

typedef double v2d __attribute__((ext_vector_type(2)));
#ifndef AKS_PB_NT_STORE
#define AKS_PB_NT_STORE 1        // 0: plain product stores (experiment: products that should stay in the Infinity Cache)
#endif
__device__ __forceinline__ void store_stream(c128 v, c128 *p) {      // written once, read by another kernel
#if AKS_PB_NT_STORE
    v2d t; t.x = v.x; t.y = v.y;
    __builtin_nontemporal_store(t, reinterpret_cast<v2d *>(p));
#else
    *p = v;
#endif
}
__device__ __forceinline__ void store_stream(double v, double *p) {
#if AKS_PB_NT_STORE
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
#ifndef AKS_NT_PB
#define AKS_NT_PB 0              // A/B knob: non-temporal loads of the streams the binned kernels read once
#endif
__device__ __forceinline__ double ld_once(const double *p) { return AKS_NT_PB ? __builtin_nontemporal_load(p) : *p; }
__device__ __forceinline__ c128 ld_once(const c128 *p) {
#if AKS_NT_PB
    const v2d v = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(p));
    return make_double2(v.x, v.y);
#else
    return *p;
#endif
}
__device__ __forceinline__ int ld_once(const uint16_t *p) { return AKS_NT_PB ? (int)__builtin_nontemporal_load(p) : (int)*p; }

// Phase 1: prod[k] = val[k] * x[slab * 8192 + lcol[k]] for the sub-slab's entries; x slice in LDS.
// VT: value type (double | c128), XT: vector entry type (c128 | double for real-packed vectors).
template <typename VT, typename XT>
__global__ __launch_bounds__(PB_P1_THREADS) void k_pb_phase1(int64_t n_cols, const int32_t *__restrict__ slab_begin,
                                                            const int32_t *__restrict__ slab_end,
                                                            const VT *__restrict__ val, const uint16_t *__restrict__ lcol,
                                                            const XT *__restrict__ x, XT *__restrict__ prod,
                                                            const aks_ctrl *__restrict__ ctrl,
                                                            const double *__restrict__ x_div) {
    if (ctrl != nullptr && ctrl->broken) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char pb_smem[];
    XT *xs = reinterpret_cast<XT *>(pb_smem);
    const int s = blockIdx.x;
    if (slab_begin[s] >= slab_end[s]) return;            // a sub-slab without entries (off-diagonal blocks; column groups)
    const int64_t c0 = (int64_t)s * PB_CW;
    const int cw = (int)min((int64_t)PB_CW, n_cols - c0);
    // x may be a RAW basis column (deferred normalisation): its entries are divided by the column's scale as they are
    // staged -- once per entry, where k_finish would have divided them in a pass of its own
    const double xd = x_div != nullptr ? *x_div : 0.0;
    if (is_raw(xd)) {
        for (int i = threadIdx.x; i < cw; i += PB_P1_THREADS) xs[i] = unscale(x[c0 + i], xd);
    } else {
        for (int i = threadIdx.x; i < cw; i += PB_P1_THREADS) xs[i] = x[c0 + i];
    }
    __syncthreads();
    const int k0 = slab_begin[s], k1 = (slab_end[s] + 7) & ~7;       // pad slots hold val = 0, lcol = 0
    for (int base = k0; base < k1; base += PB_P1_THREADS * PB_P1_U) {
        VT a[PB_P1_U];
        int c[PB_P1_U];
#pragma unroll
        for (int u = 0; u < PB_P1_U; ++u) {
            const int k = min(base + u * PB_P1_THREADS + (int)threadIdx.x, k1 - 1);
            a[u] = ld_once(&val[k]);
            c[u] = ld_once(&lcol[k]);
        }
#pragma unroll
        for (int u = 0; u < PB_P1_U; ++u) {
            const int k = base + u * PB_P1_THREADS + (int)threadIdx.x;
            if (k < k1) store_stream(cmul(a[u], xs[c[u]]), &prod[k]);
        }
    }
}

__device__ __forceinline__ void pb_lds_barrier() {
    // LDS-only release/acquire + barrier: loads already in flight (the product prefetch) stay in flight
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
__device__ __forceinline__ void pb_acc_add(double *acc, int row, c128 p) {
    unsafeAtomicAdd(&acc[row], p.x);
    unsafeAtomicAdd(&acc[PB_RB + row], p.y);
}
__device__ __forceinline__ void pb_acc_add(double *acc, int row, double p) { unsafeAtomicAdd(&acc[row], p); }
__device__ __forceinline__ c128 pb_acc_get(const double *acc, int i, c128) { return make_double2(acc[i], acc[PB_RB + i]); }
__device__ __forceinline__ double pb_acc_get(const double *acc, int i, double) { return acc[i]; }
__device__ __forceinline__ void pb_acc_clear(double *acc, int i, c128) { acc[i] = 0.0; acc[PB_RB + i] = 0.0; }
__device__ __forceinline__ void pb_acc_clear(double *acc, int i, double) { acc[i] = 0.0; }
__device__ __forceinline__ c128 pb_sum(c128 a, c128 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double pb_sum(double a, double b) { return a + b; }
template <typename XT> __device__ __forceinline__ XT pb_zero();
template <> __device__ __forceinline__ c128 pb_zero<c128>() { return make_double2(0.0, 0.0); }
template <> __device__ __forceinline__ double pb_zero<double>() { return 0.0; }

// Phase 2: one workgroup per row block.  Software pipeline per wave, one step per round:
//   descriptors (a block of PB_B rounds per vector load, one block ahead, broadcast with v_readlane)
//   -> products (a lane picks its piece of the wave-load) + the lane's four (level, row) words, PB_D - 1
//      steps ahead, unconditional loads so that the compiler's s_waitcnt counts stay exact
//   -> LDS adds level by level, workgroup barrier after each level.
// The kernel is bound by the LDS atomics (profiles/microbench/pb_abi_bench.txt: cycle sums per round), which
// is why a wave-load is filled to 64 lanes from up to three pieces instead of taking one tile.
template <typename XT, bool ACC>
__global__ __launch_bounds__(PB_W * 64) void k_pb_phase2(int64_t n_rows, int n_rb, int n_chunks, int chunks_per_xcd,
                                                        const int32_t *__restrict__ rb_run_ptr,
                                                        const uint4 *__restrict__ runs,
                                                        const uint16_t *__restrict__ lrow,
                                                        const XT *__restrict__ prod, XT *__restrict__ y,
                                                        const aks_ctrl *__restrict__ ctrl) {
    if (ctrl != nullptr && ctrl->broken) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char pb_smem[];
    constexpr int T = PB_W * 64, NACC = PB_RB * (int)(sizeof(XT) / sizeof(double));
    double *acc = reinterpret_cast<double *>(pb_smem);                      // re plane [, im plane]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // One workgroup per CHUNK of row blocks (one chunk per CU of an MI355X): the pipeline below runs across
    // the row-block boundaries, so only the first round of a chunk waits for memory with nothing else in
    // flight.  Chunks are interleaved -- chunk c takes row blocks c, c + n_chunks, ... -- and consecutive
    // chunks sit on one XCD (blockIdx % 8, observed placement; speed only): neighbouring row blocks are
    // then worked on at about the same time by neighbouring CUs, and the 128-byte lines in which the
    // products of two of them meet are fetched into that XCD's L2 once.
    if ((int)(blockIdx.x >> 3) >= chunks_per_xcd) return;
    const int chunk = (blockIdx.x & 7) * chunks_per_xcd + (blockIdx.x >> 3);
    if (chunk >= n_chunks) return;
    // chunk c owns the row blocks c, c + n_chunks, c + 2 n_chunks, ...; the planner stores the row blocks in
    // that order (aks_pb_matrix.d_rb_run_ptr is indexed by position), so a chunk's rounds are contiguous
    const int q = n_rb / n_chunks, r = n_rb % n_chunks;
    const int pos_lo = chunk * q + min(chunk, r), pos_hi = pos_lo + q + (chunk < r ? 1 : 0);
    if (pos_lo >= pos_hi) return;
    for (int i = threadIdx.x; i < NACC; i += T) acc[i] = 0.0;
    const int R0 = rb_run_ptr[pos_lo];
    const int n_rounds = (rb_run_ptr[pos_hi] - R0) / PB_RPR;
    int rb = chunk;                              // row block of the round whose adds are next
    const uint4 *my_runs = runs + R0 + wave * PB_K;
    const PbWords *my_words = reinterpret_cast<const PbWords *>(lrow + (size_t)(R0 / PB_RPR) * PB_RW) + wave * 64 + lane;
    auto load_block = [&](int first_round) {
        const int l = lane & (PB_B * PB_K - 1);
        const int round = first_round + l / PB_K;
        uint4 v = my_runs[(size_t)max(min(round, n_rounds - 1), 0) * PB_RPR + l % PB_K];
        if (round < 0 || round >= n_rounds) v = make_uint4(0u, 0u, 0u, 0u);
        return v;
    };
    XT p[PB_D][PB_K];
    unsigned info[PB_D][PB_K];
    PbWords words[PB_D];
#pragma unroll
    for (int d = 0; d < PB_D; ++d) {
        words[d] = PbWords{};
#pragma unroll
        for (int k = 0; k < PB_K; ++k) { info[d][k] = 0u; p[d][k] = pb_zero<XT>(); }
    }
    // Step t issues the loads of round t - 1 into the stage that step t - 1 freed and adds round t - PB_D
    // (stage t % PB_D).  (Measured: loads before, between or after the first-level adds -- no difference; with
    // 64-lane wave-loads and shared boundary lines the kernel runs at the rate of its memory traffic.)
    // The descriptor block of steps i0 .. i0 + PB_B - 1 therefore holds rounds i0 - 1 .. i0 + PB_B - 2.
    unsigned long long tsum[6] = {0, 0, 0, 0, 0, 0}, tk = 0, tk0 = 0;
#define PB_TICK(i) do { if (AKS_PB_TICKS) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tsum[i] += now_ - tk; tk = now_; } } while (0)
    uint4 dv, dvn = load_block(-1);
    pb_lds_barrier();
    if (AKS_PB_TICKS) tk0 = tk = __builtin_amdgcn_s_memtime();
    // A descriptor block is made to land inside straight-line code (here and at step PB_B - 2 of a block),
    // where the compiler counts the loads issued since exactly; met across the loop's back edge, its wait
    // would cover (nearly) everything in flight.
    asm volatile("" : "+v"(dvn.x), "+v"(dvn.y), "+v"(dvn.z), "+v"(dvn.w));
    // Whole blocks of PB_B steps, with NO branch around any load: steps outside the rounds work on
    // all-zero descriptors (no adds; their loads re-read entry 0 / the chunk's first words).  A guard here
    // would make the compiler's s_waitcnt placement fall back to vmcnt(0) and serialise the prefetch.
    const int n_steps = n_rounds + PB_D;
    for (int i0 = 0; i0 < n_steps; i0 += PB_B) {
        dv = dvn;
        dvn = load_block(i0 + PB_B - 1);
#pragma unroll
        for (int j = 0; j < PB_B; ++j) {
            constexpr int D = PB_D;
            const int d = j % D, dp = (j + D - 1) % D;
            // Round i0 + j - 1 is loaded into stage dp while round i0 + j - PB_D (stage d) is added.
            const int rnd = min(max(i0 + j - 1, 0), max(n_rounds - 1, 0));
            auto issue_load = [&](int k) {
                const unsigned s0 = __builtin_amdgcn_readlane(dv.x, j * PB_K + k);
                const unsigned s1 = __builtin_amdgcn_readlane(dv.y, j * PB_K + k);
                const unsigned s2 = __builtin_amdgcn_readlane(dv.z, j * PB_K + k);
                const unsigned inf = __builtin_amdgcn_readlane(dv.w, j * PB_K + k);
                info[dp][k] = inf;
                const unsigned lc = min((unsigned)lane, max((inf >> 14) & 127u, 1u) - 1u);
                const unsigned base = lc < (inf & 127u) ? s0 : (lc < ((inf >> 7) & 127u) ? s1 : s2);
                p[dp][k] = ld_once(&prod[base + lc]);
            };
            unsigned m[PB_K];                   // this lane's word per wave-load; inactive lanes match no level
#pragma unroll
            for (int k = 0; k < PB_K; ++k) {
                const unsigned w = words[d].v[k / 2];
                m[k] = lane < (int)((info[d][k] >> 14) & 127u) ? ((k & 1) ? w >> 16 : w & 0xffffu) : 0xffffffffu;
            }
            const int nph = max((int)((info[d][0] >> 21) & 15u), 1);
            const bool last_of_rb = ((info[d][0] >> 25) & 1u) != 0u;
            auto add_slot = [&](int k) {
                if ((int)(m[k] >> 13) == 0) pb_acc_add(acc, (int)(m[k] & (PB_RB - 1)), p[d][k]);
            };
            words[dp] = my_words[(size_t)rnd * (PB_W * 64)];
#pragma unroll
            for (int k = 0; k < PB_K; ++k) issue_load(k);
#pragma unroll
            for (int k = 0; k < PB_K; ++k) add_slot(k);
            if (j == PB_B - 2) asm volatile("" : "+v"(dvn.x), "+v"(dvn.y), "+v"(dvn.z), "+v"(dvn.w));
            PB_TICK(0);
            pb_lds_barrier();
            PB_TICK(1);
            for (int ph = 1; ph < nph; ++ph) {
#pragma unroll
                for (int k = 0; k < PB_K; ++k)
                    if ((int)(m[k] >> 13) == ph) pb_acc_add(acc, (int)(m[k] & (PB_RB - 1)), p[d][k]);
                pb_lds_barrier();
            }
            if (last_of_rb) {                    // the row block is complete: write it out, clear the accumulators
                const int64_t row0 = (int64_t)rb << PB_RB_BITS;
                for (int i = threadIdx.x; i < PB_RB; i += T) {
                    XT v = pb_acc_get(acc, i, XT());
                    pb_acc_clear(acc, i, XT());
                    if (row0 + i < n_rows) {
                        if (ACC) v = pb_sum(v, y[row0 + i]);
                        y[row0 + i] = v;
                    }
                }
                rb += n_chunks;
                pb_lds_barrier();
            }
            PB_TICK(2);
        }
    }
    if (AKS_PB_TICKS && threadIdx.x == 0) {
        double *dbg = reinterpret_cast<double *>(const_cast<XT *>(prod)) + (size_t)chunk * 8;
        for (int i = 0; i < 6; ++i) dbg[i] = (double)tsum[i];
        dbg[6] = (double)(tk - tk0);
        dbg[7] = (double)n_rounds;
    }
#undef PB_TICK
}

// ------------------------------------------------------------------ host-side plumbing
struct Ws {
    aks_ws_layout lay;
    aks_ctrl *ctrl;
    c128 *red1, *red2, *red3, *partial;
    double *colscale;          // per basis column: 0 = normalised, else the column is raw and this is its divisor
    unsigned *ticket(int i) const { return reinterpret_cast<unsigned *>(ctrl->ticket) + i; }
};
enum { TICKET_UPDATE = 2 };      // (words 0 and 1 of aks_ctrl.ticket: unused since the panel kernels lost their tails)

int bind_ws(void *d_ws, int64_t ws_bytes, int64_t n_rows, int32_t max_dim, Ws *out) {
    if (d_ws == nullptr) return fail(AKS_ERR_ARG, "workspace pointer is null");
    if ((reinterpret_cast<uintptr_t>(d_ws) & 255) != 0) return fail(AKS_ERR_ARG, "workspace must be 256-byte aligned");
    int rc = aks_workspace_layout(n_rows, max_dim, &out->lay);
    if (rc != AKS_OK) return rc;
    if (ws_bytes < out->lay.total_bytes) return fail(AKS_ERR_ARG, "workspace too small (see aks_workspace_layout)");
    char *base = static_cast<char *>(d_ws);
    out->ctrl = reinterpret_cast<aks_ctrl *>(base + out->lay.ctrl_off);
    out->red1 = reinterpret_cast<c128 *>(base + out->lay.red1_off);
    out->red2 = reinterpret_cast<c128 *>(base + out->lay.red2_off);
    out->red3 = reinterpret_cast<c128 *>(base + out->lay.red3_off);
    out->partial = reinterpret_cast<c128 *>(base + out->lay.partial_off);
    out->colscale = reinterpret_cast<double *>(base + out->lay.colscale_off);
    return AKS_OK;
}

int check_panel(int64_t n_rows, int32_t J, const void *V, int64_t ldv, const void *w, int32_t max_dim) {
    if (n_rows <= 0) return fail(AKS_ERR_ARG, "n_rows must be positive");
    if (J < 1 || J > max_dim) return fail(AKS_ERR_ARG, "J must satisfy 1 <= J <= max_dim");
    if (max_dim > AKS_MAX_DIM) return fail(AKS_ERR_UNSUPPORTED, "max_dim exceeds AKS_MAX_DIM");
    if (V == nullptr || w == nullptr) return fail(AKS_ERR_ARG, "null panel / vector pointer");
    if (ldv < n_rows) return fail(AKS_ERR_ARG, "ldv < n_rows");
    if ((reinterpret_cast<uintptr_t>(V) & 15) || (reinterpret_cast<uintptr_t>(w) & 15))
        return fail(AKS_ERR_ARG, "complex128 arrays must be 16-byte aligned");
    return AKS_OK;
}

template <int NC>
void launch_proj_nc(dim3 grid, hipStream_t s, int64_t n, int c0, const c128 *V, int64_t ldv, const c128 *w,
                    c128 *partial, int ldp, int nrm_slot, const aks_ctrl *ctrl, const double *cs, int raw0, hipEvent_t ev0) {
    launch_timed(k_proj<NC>, grid, dim3(BLOCK), 0, s, ev0, (hipEvent_t) nullptr, n, c0, V, ldv, w, partial, ldp, nrm_slot, ctrl, cs, raw0);
}
template <int NC>
void launch_update_proj_nc(dim3 grid, hipStream_t s, int64_t n, const c128 *V, int64_t ldv, c128 *w,
                           const c128 *h, c128 *partial, int ldp, const aks_ctrl *ctrl, const double *cs, int raw0) {
    hipLaunchKernelGGL(k_update_proj<NC>, grid, dim3(BLOCK), 0, s, n, V, ldv, w, h, partial, ldp, ctrl, cs, raw0);
}

#define AKS_NC_CASES(M) \
    M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15) M(16) \
    M(17) M(18) M(19) M(20) M(21) M(22) M(23) M(24) M(25) M(26) M(27) M(28) M(29) M(30) M(31) M(32)

void dispatch_proj(int nc, dim3 grid, hipStream_t s, int64_t n, int c0, const c128 *V, int64_t ldv,
                   const c128 *w, c128 *partial, int ldp, int nrm_slot, const aks_ctrl *ctrl, const double *cs, int raw0,
                   hipEvent_t ev0) {
    switch (nc) {
#define M(N) case N: launch_proj_nc<N>(grid, s, n, c0, V, ldv, w, partial, ldp, nrm_slot, ctrl, cs, raw0, ev0); break;
        AKS_NC_CASES(M)
#if AKS_NC_MAX > 32
        M(33) M(34) M(35) M(36) M(37) M(38) M(39) M(40)
#endif
#undef M
        default: break;
    }
}

// exact-width fused kernel up to FUSED_EXACT_MAX columns, column-split fused kernel beyond
#ifndef AKS_FUSED_EXACT_MAX
#define AKS_FUSED_EXACT_MAX 40
#endif
constexpr int FUSED_EXACT_MAX = AKS_FUSED_EXACT_MAX;

#define AKS_NC_CASES_WIDE(M) M(33) M(34) M(35) M(36) M(37) M(38) M(39) M(40)
void dispatch_update_proj(int nc, dim3 grid, hipStream_t s, int64_t n, const c128 *V, int64_t ldv, c128 *w,
                          const c128 *h, c128 *partial, int ldp, const aks_ctrl *ctrl, const double *cs, int raw0) {
    switch (nc) {
#define M(N) case N: launch_update_proj_nc<N>(grid, s, n, V, ldv, w, h, partial, ldp, ctrl, cs, raw0); break;
        AKS_NC_CASES(M)
        AKS_NC_CASES_WIDE(M)
#undef M
        default: break;
    }
}

template <int NQ>
void launch_update_proj_split(dim3 grid, hipStream_t s, int64_t n, int J, const c128 *V, int64_t ldv, c128 *w,
                              const c128 *h, c128 *partial, int ldp, const aks_ctrl *ctrl, const double *cs, int raw0) {
    hipLaunchKernelGGL(k_update_proj_split<NQ>, grid, dim3(BLOCK), 0, s, n, J, V, ldv, w, h, partial, ldp, ctrl, cs, raw0);
}

#define AKS_NQ_CASES(M) \
    M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15) M(16) M(17) M(18) M(19) \
    M(20) M(21) M(22) M(23) M(24) M(25) M(26) M(27) M(28) M(29) M(30) M(31) M(32)

void dispatch_update_proj_split(int J, dim3 grid, hipStream_t s, int64_t n, const c128 *V, int64_t ldv, c128 *w,
                                const c128 *h, c128 *partial, int ldp, const aks_ctrl *ctrl, const double *cs, int raw0) {
    switch ((J + WAVES - 1) / WAVES) {
#define M(N) case N: launch_update_proj_split<N>(grid, s, n, J, V, ldv, w, h, partial, ldp, ctrl, cs, raw0); break;
        AKS_NQ_CASES(M)
#undef M
        default: break;
    }
}

// Workgroups of the exact-width panel kernels (k_proj<NC>, k_update_proj<NC>).  These hold a whole row of the panel in
// registers (two waves per SIMD fit at most) and stream best with ONE workgroup per CU: measured against the 1024
// workgroups the other kernels use, -2...-5 % per launch for every width from 4 to 40 at n = 10M and -3...-10 % at
// n = 1.25M (profiles/r03_rowblocks_ab2.txt; 1.5 workgroups per CU: slower; narrower than 4 columns a single wave per
// SIMD has too few bytes in flight).  The column-split kernel wants the many workgroups (+60 % with one per CU).
#ifndef AKS_PANEL_ONE_PER_CU
#define AKS_PANEL_ONE_PER_CU 1
#endif
static int panel_blocks(const Ws &ws, int width, int per_cu = 1) {
    int dev = 0, cus = 0;
    if (!AKS_PANEL_ONE_PER_CU || width < 4 || hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
        return ws.lay.n_blocks;
    return per_cu * cus < ws.lay.n_blocks ? per_cu * cus : ws.lay.n_blocks;
}

// projection of all J columns in groups of <= NC_MAX columns of (nearly) equal width
void enqueue_projection(hipStream_t s, const Ws &ws, int64_t n, int J, const c128 *V, int64_t ldv,
                        const c128 *w, c128 *red_out, c128 *zero_slot, int raw0, hipEvent_t ev0 = nullptr) {
    const int groups = (J + NC_MAX - 1) / NC_MAX;
    const int base = J / groups, extra = J % groups;
    const int n_blocks = panel_blocks(ws, base);          // (every group writes the same rows of `partial`)
    const dim3 grid(n_blocks);
    int c0 = 0;
    for (int g = 0; g < groups; ++g) {
        const int nc = base + (g < extra ? 1 : 0);
        dispatch_proj(nc, grid, s, n, c0, V, ldv, w, ws.partial, ws.lay.ld_partial, g == 0 ? J : -1, ws.ctrl, ws.colscale, raw0,
                      g == 0 ? ev0 : nullptr);
        c0 += nc;
    }
    hipLaunchKernelGGL(k_reduce<false>, dim3(J + 1), dim3(BLOCK), 0, s, ws.partial, n_blocks,
                       ws.lay.ld_partial, 0, red_out, J, nullptr, nullptr, 0.0, zero_slot, ws.ctrl);
}

// Dynamic LDS above the default limit has to be allowed per kernel AND per device: aks_device_init does that for
// every instantiation below on the current device; the library itself keeps no state (no "done" flags).
constexpr size_t AKS_LDS_BYTES = 160 * 1024;
template <typename K> int raise_lds(K kernel, size_t bytes, const char *where) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)bytes);
    return e == hipSuccess ? AKS_OK : hip_fail(e, where);
}

template <int MT>
int launch_truncate_mfma(hipStream_t s, int64_t n, int m, int p, c128 *V, int64_t ldv, const c128 *Qp, c128 *O,
                         int64_t ldo, int copy_last, bool init_only = false) {
    // rows per wave = 16 NS: 128 for p <= 16 (more loads in flight per K-step: 1.041 -> 1.011 ms at m = 20, p = 10,
    // n = 10M; the read/write mix of this kernel streams at ~5.4 TB/s = 0.95 ms), fewer as the accumulators grow
    constexpr int NS = MT <= 2 ? 8 : (MT <= 9 ? 4 : 2);      // keeps NS * MT * 8 accumulator registers below the spill point
    const size_t smem = (size_t)((m + 3) & ~3) * MT * 8 * sizeof(c128);
    if (smem > AKS_LDS_BYTES) return fail(AKS_ERR_UNSUPPORTED, "Qp does not fit the 160 KiB LDS");
    if (init_only) return raise_lds(k_truncate_mfma<MT, NS>, AKS_LDS_BYTES, "hipFuncSetAttribute(k_truncate_mfma)");
    const int64_t want = ((n + 16 * NS - 1) / (16 * NS) + WAVES - 1) / WAVES;
#ifndef AKS_TRUNC_BLOCKS
#define AKS_TRUNC_BLOCKS 2048     // against 4096: -1 % (m = 20, p = 10) ... -3 % (m = 41, p = 25) at n = 10M, -3 ... -5 % at
#endif                            // n = 1.25M; 256: +20 % (profiles/r03_trunc_grid_ab.txt)
    const dim3 grid((unsigned)(want < AKS_TRUNC_BLOCKS ? want : AKS_TRUNC_BLOCKS));
    hipLaunchKernelGGL((k_truncate_mfma<MT, NS>), grid, dim3(BLOCK), smem, s, n, m, p, V, ldv, Qp, O, ldo, copy_last);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        hip_fail(e, "k_truncate_mfma");
        if (smem > 48 * 1024) g_err += " (dynamic LDS above the default limit: call aks_device_init on this device first)";
        return AKS_ERR_HIP;
    }
    return AKS_OK;
}

struct Probe {
    std::vector<hipEvent_t> start, stop;
    std::vector<int32_t> tag;
    int32_t used = 0;
    hipEvent_t begin(int32_t t, hipStream_t s) {
        if (used >= (int32_t)start.size()) return nullptr;
        tag[used] = t;
        (void)hipEventRecord(start[used], s);
        return stop[used++];
    }
    EvPair reserve(int32_t t) {          // a pair handed to launch_timed instead of being recorded around the launch
        EvPair e;
        if (used >= (int32_t)start.size()) return e;
        tag[used] = t;
        e.start = start[used];
        e.stop = stop[used++];
        return e;
    }
};

// ---- one-shot small all-reduce (opt-in: AKS_ALLREDUCE=oneshot) -------------------------------------------------------
// The reductions between the Gram-Schmidt stages are <= 2 (max_dim + 1) doubles: latency, not bandwidth (SURVEY 5, 8(e):
// "one-shot over all 7 xGMI links, not a ring").  Every rank owns a MAILBOX in fine-grained device memory that all its
// peers have mapped (hipIpc* across processes, the plain pointer inside one process): two parities x `size` rows of
// ONESHOT_CAP doubles and one arrival counter per parity.  An all-reduce is ONE kernel of one workgroup per rank
// (k_oneshot_allreduce), launched in stream order:
//   number   thread 0 advances the rank's call counter q -- it lives on the DEVICE (a word of the mailbox allocation only
//            this rank's kernels touch, in stream order), so the launch carries no per-call argument and the kernel can sit
//            in a hipGraph that is replayed; parity = q & 1;
//   post     this rank's values into ITS row of EVERY peer's mailbox (its own included) with system-scope stores,
//            __threadfence_system(), workgroup barrier, then one system-scope RELEASE add per peer on that peer's
//            arrival counter of the parity;
//   wait     thread 0 polls the LOCAL counter (system-scope ACQUIRE loads, s_sleep between polls) until it has reached
//            size * (calls made on this parity) -- with a DEADLINE on the constant-rate wall clock: every wave has an
//            exit it reaches whatever its peers do.  Past the deadline the kernel stores the failing call number into
//            the communicator's status word (pinned host memory: the host reads it without any device call), writes
//            NaN into the caller's buffer and returns; every later call of the communicator still posts (its peers
//            need not suffer) but does not wait again: a lost peer costs ONE deadline, not one per reduction;
//   sum      the `size` rows IN RANK ORDER starting from +0.0 (system-scope loads) into the caller's buffer -- the
//            same bits on every rank, and the bits tests/mock_rccl's all-reduce produces.
// Two parities suffice: a peer can post call q + 1 while this rank still sums call q (other parity), but call q + 2 only
// after its own wait of q + 1 has seen THIS rank's post of q + 1, which is behind this rank's sum of q in stream order.
// Nothing here blocks a hardware queue: rounds 5's form waited with hipStreamWaitValue64, which stalls the QUEUE its
// stream is mapped to -- a post queued behind it in the same queue never ran (thread ranks of one process, from the
// second solve on: profiles/r05_small_trace.txt), and the deadline's rescue write could sit behind the same wait
// (ADVICE r05).  A kernel that polls with a deadline cannot hang; the worst it does is time out.
// aks_comm_create sets the exchange up when asked to, proves it with one reduction per parity (short deadline), lets
// the ranks vote (over ncclAllReduce), and falls back to ncclAllReduce on every rank if any rank could not
// (aks_comm_allreduce_path tells which one runs).  Ranks that share a PROCESS are voted down unless
// AKS_ONESHOT_SAME_PROCESS=1: the runtime multiplexes a process's streams onto few hardware queues (GPU_MAX_HW_QUEUES,
// default 4), where one rank's polling kernel can sit in front of the kernel whose post it polls for -- that reduction
// would time out.  Rank processes (one per GPU: the product's mode) have queues of their own.
// What only multi-GPU hardware can confirm: system-scope visibility of the posts over xGMI and the cost of the remote
// adds; on one GPU the peers' mailboxes are local memory.
constexpr int ONESHOT_CAP = 2 * (AKS_MAX_DIM + 2);      // doubles per row: the widest stage reduction, [h ; ||w||^2]
constexpr int ONESHOT_MAX_RANKS = 16;
constexpr int ONESHOT_FLAG_STRIDE = 16;                 // counters 128 bytes apart
constexpr int ONESHOT_CALLS_WORD = 2 * ONESHOT_FLAG_STRIDE;   // the rank's own call counter, behind the two arrival counters
struct OneShotPeers {
    double *box[ONESHOT_MAX_RANKS];
    unsigned long long *flag[ONESHOT_MAX_RANKS];
};
struct OneShot {
    bool active = false;
    void *local = nullptr;                              // this rank's mailbox allocation (counters, then rows)
    void *opened[ONESHOT_MAX_RANKS] = {};               // hipIpcOpenMemHandle results to close again
    OneShotPeers peers = {};
    unsigned long long *status = nullptr;               // pinned host word: 0, or (call number << 8 | 1) of the reduction that timed out
    unsigned long long deadline_ticks = 0;              // of the device's constant-rate wall clock
    double ticks_per_ms = 1e5;
    bool mute = false;                                  // fault injection (AKS_ONESHOT_FAULT_RANK=<rank>): this rank's posts are lost
    std::string why_not;                                // set when the set-up was asked for and did not succeed
};
constexpr size_t ONESHOT_FLAG_BYTES = 3 * ONESHOT_FLAG_STRIDE * sizeof(unsigned long long);
inline size_t oneshot_bytes(int size) { return ONESHOT_FLAG_BYTES + (size_t)2 * size * ONESHOT_CAP * sizeof(double); }

__global__ __launch_bounds__(BLOCK) void k_oneshot_allreduce(double *__restrict__ buf, int count, OneShotPeers P, int size,
                                                            int rank, unsigned long long *mine, const double *box,
                                                            unsigned long long *status,
                                                            unsigned long long deadline_ticks, int mute) {
    // (mine = P.flag[rank], box = P.box[rank]: arguments of their own, so that the kernarg table is never indexed with a
    // run-time value through the scalar unit -- the ISA lint follows constant kernarg offsets only)
    __shared__ unsigned long long sh_q;
    __shared__ int sh_ok;
    // (every access to mailbox memory is an explicit atomic: VECTOR memory instructions -- a plain load of a uniform address
    // would become a scalar load, and the scalar cache is not coherent with the vector stores this kernel and its peers make
    // to the same lines: the hazard class csrc/check_scalar_hazards.py exists for)
    if (threadIdx.x == 0) {
        const unsigned long long q = __hip_atomic_load(&mine[ONESHOT_CALLS_WORD], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1ull;
        __hip_atomic_store(&mine[ONESHOT_CALLS_WORD], q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (only this rank's kernels
        sh_q = q;                                                                                       //  touch the word, in stream order)
    }
    __syncthreads();
    const unsigned long long q = sh_q;
    const int parity = (int)(q & 1ull);
    if (!mute) {
        for (int idx = threadIdx.x; idx < count * size; idx += BLOCK) {
            const int peer = idx / count, e = idx - peer * count;
            __hip_atomic_store(&P.box[peer][((size_t)parity * size + rank) * ONESHOT_CAP + e], buf[e], __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    __threadfence_system();                              // every storing thread: its rows are performed at system scope
    __syncthreads();
    if (!mute && (int)threadIdx.x < size)
        __hip_atomic_fetch_add(&P.flag[threadIdx.x][parity * ONESHOT_FLAG_STRIDE], 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (threadIdx.x == 0) {
        // (device-side copy of the status) an earlier call failed: do not wait again
        int ok = __hip_atomic_load(&mine[ONESHOT_CALLS_WORD + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0ull;
        if (ok) {
            const unsigned long long want = (unsigned long long)size * ((q + (unsigned long long)parity) / 2ull);   // calls on this parity so far
            const unsigned long long t0 = wall_clock64();
            while (__hip_atomic_load(&mine[parity * ONESHOT_FLAG_STRIDE], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < want) {
                if (wall_clock64() - t0 > deadline_ticks) { ok = 0; break; }
                __builtin_amdgcn_s_sleep(16);
            }
            if (!ok) {
                __hip_atomic_store(&mine[ONESHOT_CALLS_WORD + 1], q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(status, (q << 8) | 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        sh_ok = ok;
    }
    __syncthreads();
    __threadfence_system();                              // the rows are read after thread 0's acquire, by every thread
    const bool ok = sh_ok != 0;
    for (int e = threadIdx.x; e < count; e += BLOCK) {
        double s = 0.0;
        for (int src = 0; src < size; ++src)             // rank order: the same bits on every rank
            s += __hip_atomic_load(const_cast<double *>(&box[((size_t)parity * size + src) * ONESHOT_CAP + e]), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_SYSTEM);
        buf[e] = ok ? s : __builtin_nan("");
    }
}

struct Comm {
    ncclComm_t nccl = nullptr;
    hipStream_t side = nullptr;          // carries the ghost exchange next to the diagonal-block SpMV
    hipEvent_t packed = nullptr, arrived = nullptr;
    int rank = 0, size = 1;
    OneShot one;
    // hipGraphs that captured operations of this communicator (aks_comm_graph_retain / _release): they must be destroyed
    // BEFORE the communicator -- ncclCommDestroy never returns while a graph holds a captured send / recv
    // (profiles/r05_capture_crash.txt section 4) -- so aks_comm_destroy refuses while any is alive: an error, not a hang
    std::atomic<int> graphs{0};
};

// RCCL is loaded when the first communicator is asked for, not with the library: librccl.so is hundreds of megabytes of
// code objects whose registration costs a process 1.2 s at load time (5 s from a cold page cache) -- a single-GPU solve,
// which never calls it, should not pay that (a torch process has it loaded already; the torch-free backend has not:
// profiles/cold_process_probe.py).  Types and constants come from <rccl/rccl.h>; the nine entry points through this table.
#ifdef AKS_RCCL_DIRECT             // tests/mock_rccl: the stand-in is linked in, its symbols renamed on the command line
#define NCCL_CALL(name) nccl##name
static int rccl_load() { return AKS_OK; }
#else
struct RcclApi {
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;       // (optional: aks_runtime_versions)
};
static RcclApi g_rccl;                  // filled once (std::call_once), read-only afterwards
static std::once_flag g_rccl_once;
static std::string g_rccl_error;        // why loading failed (written once, inside call_once)
#define NCCL_CALL(name) g_rccl.name
static int rccl_load() {
    std::call_once(g_rccl_once, [] {
        void *h = nullptr;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (h != nullptr) break;
        }
        if (h == nullptr) {
            const char *why = dlerror();
            g_rccl_error = std::string("librccl.so could not be loaded: ") + (why ? why : "?");
            return;
        }
        bool ok = true;
        auto sym = [&](const char *name) { void *p = dlsym(h, name); if (p == nullptr) { ok = false; g_rccl_error = std::string("librccl.so lacks ") + name; } return p; };
        RcclApi api;
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
        api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
        api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
        api.Send = reinterpret_cast<decltype(api.Send)>(sym("ncclSend"));
        api.Recv = reinterpret_cast<decltype(api.Recv)>(sym("ncclRecv"));
        api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
        api.GetVersion = reinterpret_cast<decltype(api.GetVersion)>(dlsym(h, "ncclGetVersion"));
        if (ok) g_rccl = api;
    });
    if (g_rccl.AllReduce == nullptr) return fail(AKS_ERR_HIP, g_rccl_error.empty() ? "librccl.so could not be loaded" : g_rccl_error.c_str());
    return AKS_OK;
}
#endif

int nccl_fail(ncclResult_t r, const char *where) {
    g_err = std::string(where) + ": " + NCCL_CALL(GetErrorString)(r);
    return AKS_ERR_HIP;
}

// one CSR block applied with whichever form its plan selects; vectors complex128 or (real) float64
int launch_pb_any(const aks_pb_matrix *A, const void *x, void *y, int accumulate, const void *d_ws, void *stream, bool real,
                  const double *x_div, EvPair ev);

// (x_div: scale of a raw input column -- deferred normalisation.  The binned form applies it once per x entry, the
// sliced form once per non-zero and therefore only for short rows; aks_arnoldi_expand defers only when the diagonal
// block is in one of these two shapes: block_defers)
int sell_spmv_any(const aks_sell_matrix *A, const void *x, void *y, int accumulate, const void *d_ws, void *stream, bool real,
                  const double *x_div, EvPair ev);
#ifndef AKS_SELL_DEFER_WIDTH
#define AKS_SELL_DEFER_WIDTH 8    // mean padded row length up to which the sliced form divides raw x entries on the fly
#endif
static bool block_defers(const aks_csr_block &B) {
    if (B.pb != nullptr) return true;
    return B.sell != nullptr && B.sell->n_slices > 0 &&
           B.sell->nnz_pad <= (int64_t)AKS_SELL_DEFER_WIDTH * 64 * B.sell->n_slices;
}
int csr_spmv_any(const aks_csr_block &B, const void *x, void *y, int accumulate, const void *d_ws, void *stream, bool real,
                 EvPair ev);

int apply_block(const aks_csr_block &B, const void *x, void *y, int accumulate, const void *d_ws, void *stream, bool real,
                const double *x_div = nullptr, EvPair ev = EvPair()) {
    if (B.n_rows <= 0) return AKS_OK;
    if (x_div != nullptr && !block_defers(B))
        return fail(AKS_ERR_ARG, "a raw input column needs the binned form or the sliced form with short rows");
    if (real && B.values_complex) return fail(AKS_ERR_ARG, "real vectors need real matrix values");
    if (B.sell != nullptr) return sell_spmv_any(B.sell, x, y, accumulate, d_ws, stream, real, x_div, ev);
    if (B.pb != nullptr) return launch_pb_any(B.pb, x, y, accumulate, d_ws, stream, real, x_div, ev);
    return csr_spmv_any(B, x, y, accumulate, d_ws, stream, real, ev);
}

// ---- host threads of the planners.  A plan is a pure function of the matrix: every pass below is split into contiguous
// ranges of row blocks whose outputs are disjoint (or are concatenated in range order), so the plan is the same bytes
// for any number of threads.  AKS_PLAN_THREADS overrides the default min(hardware threads, 16).
static int plan_threads(int64_t work_items, int64_t nnz) {
    if (nnz < (int64_t)1 << 20) return 1;          // a millisecond of work: not worth starting threads for
    int nt = 0;
    if (const char *e = getenv("AKS_PLAN_THREADS")) nt = atoi(e);
    if (nt <= 0) {
        nt = (int)std::thread::hardware_concurrency();
        if (nt > 16) nt = 16;
    }
    if (nt > 64) nt = 64;
    if ((int64_t)nt > work_items) nt = (int)work_items;
    return nt < 1 ? 1 : nt;
}

struct PlanClock {      // AKS_PLAN_TIMING=1: the planner's phases on stderr
    bool on = getenv("AKS_PLAN_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void lap(const char *what) {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[aks plan] %-22s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};

// f(t, begin, end) on nt threads over [0, n) in contiguous ranges; the first exception is rethrown in the caller
template <typename F>
static void plan_parallel(int64_t n, int nt, F f) {
    if (nt <= 1 || n <= 1) { f(0, (int64_t)0, n); return; }
    std::vector<std::thread> pool;
    std::vector<std::exception_ptr> errs(nt);
    for (int t = 0; t < nt; ++t)
        pool.emplace_back([&, t] {
            try { f(t, n * t / nt, n * (t + 1) / nt); } catch (...) { errs[t] = std::current_exception(); }
        });
    for (auto &th : pool) th.join();
    for (auto &e : errs) if (e) std::rethrow_exception(e);
}

// ---- tile-binned SpMV: host-side plan and launcher
// the big arrays of a plan: allocated WITHOUT being written (a std::vector would zero-fill -- and first-touch -- hundreds of
// megabytes on one thread); every element is written exactly once by the pass that owns it
// Large ones sit on 2 MiB boundaries with MADV_HUGEPAGE: with transparent huge pages in "madvise" mode (these hosts) a
// 400 MB array is 200 pages instead of 100 000 -- its first touch and, above all, its release (75-90 ms for the plan
// of the 10M-row matrix with 4 KiB pages) stop being visible.
template <typename T>
struct RawArray {
    static_assert(std::is_trivially_default_constructible<T>::value && std::is_trivially_destructible<T>::value, "raw storage");
    struct Free { void operator()(T *q) const { free(q); } };
    std::unique_ptr<T, Free> p;
    size_t n = 0;
    void alloc(size_t count) {
        const size_t bytes = std::max<size_t>(count * sizeof(T), 1), huge = (size_t)2 << 20;
        void *q = nullptr;
        if (bytes >= 2 * huge) {
            const size_t len = (bytes + huge - 1) / huge * huge;
            if (posix_memalign(&q, huge, len) != 0) q = nullptr;
            if (q != nullptr) (void)madvise(q, len, MADV_HUGEPAGE);
        } else {
            q = malloc(bytes);
        }
        if (q == nullptr) throw std::bad_alloc();
        p.reset(static_cast<T *>(q));
        n = count;
    }
    T *data() { return p.get(); }
    const T *data() const { return p.get(); }
    size_t size() const { return n; }
    T &operator[](size_t i) { return p.get()[i]; }
    const T &operator[](size_t i) const { return p.get()[i]; }
};

struct PbPlan {
    aks_pb_sizes sz;
    int32_t values_complex;
    RawArray<double> val;                // nnz_pad (x2 if complex)
    RawArray<uint16_t> lcol, lrow;
    std::vector<int32_t> slab_begin, slab_end, rb_run_ptr;
    RawArray<aks_pb_run> runs;
};

static void plan_copy(void *dst, const void *src, size_t bytes) {        // memcpy on the planner's threads
    const int nt = plan_threads((int64_t)(bytes >> 22), (int64_t)bytes);
    plan_parallel((int64_t)bytes, nt, [&](int, int64_t b0, int64_t b1) {
        memcpy(static_cast<char *>(dst) + b0, static_cast<const char *>(src) + b0, (size_t)(b1 - b0));
    });
}

int check_pb(const aks_pb_matrix *A, const void *x, const void *y) {
    if (A == nullptr || x == nullptr || y == nullptr) return fail(AKS_ERR_ARG, "null pointer");
    if (A->n_rows <= 0 || A->n_cols <= 0 || A->nnz < 0 || A->nnz_pad < 8 || A->n_runs < PB_RPR ||
        A->n_runs % PB_RPR != 0 || A->n_lrow != A->n_runs * AKS_PB_RUN_MAX)
        return fail(AKS_ERR_ARG, "bad sizes (n_runs: whole rounds, the last one empty; n_lrow = 64 n_runs)");
    if (A->n_slabs != (int32_t)((A->n_cols + PB_CW - 1) / PB_CW) ||
        A->n_rowblocks != (int32_t)((A->n_rows + PB_RB - 1) >> PB_RB_BITS))
        return fail(AKS_ERR_ARG, "n_slabs / n_rowblocks do not match the shape");
    if (!A->d_val || !A->d_lcol || !A->d_slab_begin || !A->d_slab_end || !A->d_runs || !A->d_rb_run_ptr ||
        !A->d_lrow || !A->d_prod)
        return fail(AKS_ERR_ARG, "null array in aks_pb_matrix");
    if ((reinterpret_cast<uintptr_t>(A->d_lrow) & 7) || (reinterpret_cast<uintptr_t>(A->d_runs) & 15) ||
        (reinterpret_cast<uintptr_t>(A->d_prod) & 15))
        return fail(AKS_ERR_ARG, "d_lrow / d_runs / d_prod are not 8 / 16 / 16-byte aligned");
    if (x == y) return fail(AKS_ERR_ARG, "x and y must not alias");
    return AKS_OK;
}

template <typename VT, typename XT>
int launch_pb(const aks_pb_matrix *A, const XT *x, XT *y, int accumulate, const aks_ctrl *ctrl, hipStream_t s,
              bool init_only = false, const double *x_div = nullptr, EvPair ev = EvPair()) {
    const size_t lds1 = (size_t)PB_CW * sizeof(XT);
    const size_t lds2 = (size_t)PB_RB * sizeof(XT);
    if (init_only) {
        int rc = raise_lds(k_pb_phase1<VT, XT>, lds1, "hipFuncSetAttribute(k_pb_phase1)");
        if (rc == AKS_OK) rc = raise_lds(k_pb_phase2<XT, true>, lds2, "hipFuncSetAttribute(k_pb_phase2)");
        if (rc == AKS_OK) rc = raise_lds(k_pb_phase2<XT, false>, lds2, "hipFuncSetAttribute(k_pb_phase2)");
        return rc;
    }
    XT *prod = reinterpret_cast<XT *>(A->d_prod);       // real vectors use the first 8 nnz_pad bytes
    if (A->nnz > 0)
        launch_timed(k_pb_phase1<VT, XT>, dim3((unsigned)A->n_slabs), dim3(PB_P1_THREADS), lds1, s, ev.start, (hipEvent_t) nullptr,
                     A->n_cols, A->d_slab_begin, A->d_slab_end, static_cast<const VT *>(A->d_val), A->d_lcol, x, prod, ctrl, x_div);
    const hipEvent_t ev2 = A->nnz > 0 ? nullptr : ev.start;       // (phase 2 alone carries both events if phase 1 is skipped)
    const int n_chunks = (int)std::min<int64_t>(A->n_rowblocks, AKS_PB_CHUNKS);   // as the planner ordered them
    const int cpx = (n_chunks + 7) / 8;
    const uint4 *runs = reinterpret_cast<const uint4 *>(A->d_runs);
    if (accumulate)
        launch_timed(k_pb_phase2<XT, true>, dim3((unsigned)(cpx * 8)), dim3(PB_W * 64), lds2, s, ev2, ev.stop, A->n_rows,
                     A->n_rowblocks, n_chunks, cpx, A->d_rb_run_ptr, runs, A->d_lrow, prod, y, ctrl);
    else
        launch_timed(k_pb_phase2<XT, false>, dim3((unsigned)(cpx * 8)), dim3(PB_W * 64), lds2, s, ev2, ev.stop, A->n_rows,
                     A->n_rowblocks, n_chunks, cpx, A->d_rb_run_ptr, runs, A->d_lrow, prod, y, ctrl);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        hip_fail(e, "aks_pb_spmv");
        g_err += " (the binned kernels use 128 KiB of dynamic LDS: call aks_device_init on this device first)";
        return AKS_ERR_HIP;
    }
    return AKS_OK;
}

int launch_pb_any(const aks_pb_matrix *A, const void *x, void *y, int accumulate, const void *d_ws, void *stream, bool real,
                  const double *x_div, EvPair ev) {
    int rc = check_pb(A, x, y);
    if (rc != AKS_OK) return rc;
    const aks_ctrl *ctrl = static_cast<const aks_ctrl *>(d_ws);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (real) {
        if (A->values_complex) return fail(AKS_ERR_ARG, "real vectors need real matrix values");
        return launch_pb<double, double>(A, static_cast<const double *>(x), static_cast<double *>(y), accumulate, ctrl, s, false, x_div, ev);
    }
    const c128 *xc = static_cast<const c128 *>(x);
    c128 *yc = static_cast<c128 *>(y);
    return A->values_complex ? launch_pb<c128, c128>(A, xc, yc, accumulate, ctrl, s, false, x_div, ev)
                             : launch_pb<double, c128>(A, xc, yc, accumulate, ctrl, s, false, x_div, ev);
}

int check_sell(const aks_sell_matrix *A, const void *x, const void *y) {
    if (A == nullptr || x == nullptr || y == nullptr) return fail(AKS_ERR_ARG, "null pointer");
    if (A->n_rows <= 0 || A->n_cols <= 0 || A->nnz < 0 || A->nnz_pad < 0 || (A->nnz_pad & 63) != 0 ||
        A->n_slices != (A->n_rows + 63) / 64)
        return fail(AKS_ERR_ARG, "bad sizes in aks_sell_matrix");
    if (!A->d_slice_ptr || (A->nnz_pad > 0 && (!A->d_col || !A->d_val))) return fail(AKS_ERR_ARG, "null array in aks_sell_matrix");
    if (x == y) return fail(AKS_ERR_ARG, "x and y must not alias");
    return AKS_OK;
}

template <typename VT, typename XT>
int launch_sell(const aks_sell_matrix *A, const XT *x, XT *y, int accumulate, const aks_ctrl *ctrl, hipStream_t s,
                const double *x_div = nullptr, EvPair ev = EvPair()) {
    const dim3 grid((unsigned)(((A->n_slices + WAVES - 1) / WAVES + 7) / 8 * 8));    // whole groups of 8: the XCD order
    const VT *val = static_cast<const VT *>(A->d_val);
    if (accumulate)
        launch_timed(k_sell<VT, XT, true>, grid, dim3(BLOCK), 0, s, ev.start, ev.stop, A->n_rows, A->n_slices, A->d_slice_ptr, A->d_col, val, x, y, ctrl, x_div);
    else
        launch_timed(k_sell<VT, XT, false>, grid, dim3(BLOCK), 0, s, ev.start, ev.stop, A->n_rows, A->n_slices, A->d_slice_ptr, A->d_col, val, x, y, ctrl, x_div);
    AKS_CHECK_LAUNCH("aks_sell_spmv");
    return AKS_OK;
}

int sell_spmv_any(const aks_sell_matrix *A, const void *x, void *y, int accumulate, const void *d_ws, void *stream, bool real,
                  const double *x_div, EvPair ev) {
    int rc = check_sell(A, x, y);
    if (rc != AKS_OK) return rc;
    const aks_ctrl *ctrl = static_cast<const aks_ctrl *>(d_ws);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (real) {
        if (A->values_complex) return fail(AKS_ERR_ARG, "real vectors need real matrix values");
        return launch_sell<double, double>(A, static_cast<const double *>(x), static_cast<double *>(y), accumulate, ctrl, s, x_div, ev);
    }
    const c128 *xc = static_cast<const c128 *>(x);
    c128 *yc = static_cast<c128 *>(y);
    return A->values_complex ? launch_sell<c128, c128>(A, xc, yc, accumulate, ctrl, s, x_div, ev)
                             : launch_sell<double, c128>(A, xc, yc, accumulate, ctrl, s, x_div, ev);
}

int csr_spmv_any(const aks_csr_block &B, const void *x, void *y, int accumulate, const void *d_ws, void *stream, bool real,
                 EvPair ev) {
    if (B.n_rows <= 0 || B.n_tiles <= 0) return fail(AKS_ERR_ARG, "empty matrix");
    if (!B.d_indptr || !B.d_indices || !B.d_values || !B.d_tiles || !x || !y) return fail(AKS_ERR_ARG, "null pointer");
    if (x == y) return fail(AKS_ERR_ARG, "x and y must not alias");
    int lpr = B.lanes_per_row;
    if (lpr <= 0) lpr = 1;
    if (lpr > 64 || (lpr & (lpr - 1)) != 0) return fail(AKS_ERR_ARG, "lanes_per_row must be a power of two <= 64");
    const aks_ctrl *ctrl = static_cast<const aks_ctrl *>(d_ws);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)((B.n_tiles + WAVES - 1) / WAVES));
    if (real) {
        const double *v = static_cast<const double *>(B.d_values), *xr = static_cast<const double *>(x);
        double *yr = static_cast<double *>(y);
        if (accumulate)
            launch_timed(k_spmv<double, double, true>, grid, dim3(BLOCK), 0, s, ev.start, ev.stop, B.n_tiles, B.d_indptr, B.d_indices, v, B.d_tiles, lpr, xr, yr, ctrl);
        else
            launch_timed(k_spmv<double, double, false>, grid, dim3(BLOCK), 0, s, ev.start, ev.stop, B.n_tiles, B.d_indptr, B.d_indices, v, B.d_tiles, lpr, xr, yr, ctrl);
    } else {
        const c128 *xc = static_cast<const c128 *>(x);
        c128 *yc = static_cast<c128 *>(y);
        if (B.values_complex) {
            const c128 *v = static_cast<const c128 *>(B.d_values);
            if (accumulate)
                launch_timed(k_spmv<c128, c128, true>, grid, dim3(BLOCK), 0, s, ev.start, ev.stop, B.n_tiles, B.d_indptr, B.d_indices, v, B.d_tiles, lpr, xc, yc, ctrl);
            else
                launch_timed(k_spmv<c128, c128, false>, grid, dim3(BLOCK), 0, s, ev.start, ev.stop, B.n_tiles, B.d_indptr, B.d_indices, v, B.d_tiles, lpr, xc, yc, ctrl);
        } else {
            const double *v = static_cast<const double *>(B.d_values);
            if (accumulate)
                launch_timed(k_spmv<double, c128, true>, grid, dim3(BLOCK), 0, s, ev.start, ev.stop, B.n_tiles, B.d_indptr, B.d_indices, v, B.d_tiles, lpr, xc, yc, ctrl);
            else
                launch_timed(k_spmv<double, c128, false>, grid, dim3(BLOCK), 0, s, ev.start, ev.stop, B.n_tiles, B.d_indptr, B.d_indices, v, B.d_tiles, lpr, xc, yc, ctrl);
        }
    }
    AKS_CHECK_LAUNCH("k_spmv");
    return AKS_OK;
}

}  // namespace

// =================================================================== C ABI
extern "C" {

const char *aks_last_error(void) { return g_err.c_str(); }
int32_t aks_abi_version(void) { return AKS_ABI_VERSION; }

int aks_device_init(void) {
    int rc = launch_pb<double, c128>(nullptr, nullptr, nullptr, 0, nullptr, nullptr, true);
    if (rc == AKS_OK) rc = launch_pb<c128, c128>(nullptr, nullptr, nullptr, 0, nullptr, nullptr, true);
    if (rc == AKS_OK) rc = launch_pb<double, double>(nullptr, nullptr, nullptr, 0, nullptr, nullptr, true);
#define M(N) if (rc == AKS_OK) rc = launch_truncate_mfma<N>(nullptr, 0, 0, 0, nullptr, 0, nullptr, nullptr, 0, 0, true);
    M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12)
#undef M
    return rc;
}

int aks_workspace_layout(int64_t n_rows, int32_t max_dim, aks_ws_layout *out) {
    if (out == nullptr) return fail(AKS_ERR_ARG, "layout pointer is null");
    if (n_rows <= 0) return fail(AKS_ERR_ARG, "n_rows must be positive");
    if (max_dim < 1 || max_dim > AKS_MAX_DIM) return fail(AKS_ERR_UNSUPPORTED, "max_dim outside [1, AKS_MAX_DIM]");
    int64_t nb = (n_rows + BLOCK - 1) / BLOCK;
    if (nb > MAX_ROW_BLOCKS) nb = MAX_ROW_BLOCKS;
    const int32_t ld = max_dim + 2;
    int64_t off = 0;
    out->ctrl_off = off;                 off = align_up(off + (int64_t)sizeof(aks_ctrl), 256);
    out->red1_off = off;                 off = align_up(off + (int64_t)ld * 16, 256);
    out->red2_off = off;                 off = align_up(off + (int64_t)ld * 16, 256);
    out->red3_off = off;                 off = align_up(off + 2 * 16, 256);
    out->colscale_off = off;             off = align_up(off + (int64_t)(AKS_MAX_DIM + 2) * 8, 256);
    out->partial_off = off;              off = align_up(off + nb * ld * 16, 256);
    out->total_bytes = off;
    out->n_blocks = (int32_t)nb;
    out->ld_partial = ld;
    out->red_len = ld;
    out->pad_ = 0;
    return AKS_OK;
}

int aks_workspace_init(void *d_ws, int64_t ws_bytes, int64_t n_rows, int32_t max_dim, void *stream) {
    Ws ws;
    int rc = bind_ws(d_ws, ws_bytes, n_rows, max_dim, &ws);
    if (rc != AKS_OK) return rc;
    hipError_t e = hipMemsetAsync(d_ws, 0, (size_t)ws.lay.partial_off, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(workspace)");
    return AKS_OK;
}

int64_t aks_csr_plan_tiles(const int32_t *indptr, int64_t n_rows, int32_t tile_nnz, int32_t *tiles_out,
                           int64_t cap) {
    if (indptr == nullptr || tiles_out == nullptr) return fail(AKS_ERR_ARG, "null pointer");
    if (n_rows <= 0 || n_rows >= INT32_MAX) return fail(AKS_ERR_ARG, "n_rows out of range");
    if (tile_nnz != AKS_SPMV_TILE_NNZ) return fail(AKS_ERR_ARG, "tile_nnz must be AKS_SPMV_TILE_NNZ");
    const int64_t max_rows = 4 * tile_nnz;  // bounds the work of one wave on runs of empty rows
    int64_t count = 0;
    int64_t r = 0;
    while (r < n_rows) {
        if (count + 1 >= cap) return fail(AKS_ERR_ARG, "tiles_out too small");
        tiles_out[count++] = (int32_t)r;
        const int64_t k0 = indptr[r];
        int64_t e = r + 1;  // a tile always holds at least one row (possibly longer than tile_nnz)
        while (e < n_rows && e - r < max_rows && (int64_t)indptr[e + 1] - k0 <= tile_nnz) ++e;
        r = e;
    }
    tiles_out[count] = (int32_t)n_rows;
    return count;
}

int aks_csr_spmv(int64_t n_rows, const int32_t *d_indptr, const int32_t *d_indices, const void *d_values,
                 int32_t values_complex, const int32_t *d_tiles, int64_t n_tiles, int32_t lanes_per_row,
                 const aks_c128 *d_x, aks_c128 *d_y, int32_t accumulate, const void *d_ws, void *stream) {
    aks_csr_block B;
    memset(&B, 0, sizeof B);
    B.n_rows = n_rows; B.d_indptr = d_indptr; B.d_indices = d_indices; B.d_values = d_values; B.d_tiles = d_tiles;
    B.n_tiles = n_tiles; B.values_complex = values_complex; B.lanes_per_row = lanes_per_row;
    return csr_spmv_any(B, d_x, d_y, accumulate, d_ws, stream, false, EvPair());
}

int aks_csr_spmv_real(int64_t n_rows, const int32_t *d_indptr, const int32_t *d_indices, const double *d_values,
                      const int32_t *d_tiles, int64_t n_tiles, int32_t lanes_per_row, const double *d_x,
                      double *d_y, int32_t accumulate, const void *d_ws, void *stream) {
    aks_csr_block B;
    memset(&B, 0, sizeof B);
    B.n_rows = n_rows; B.d_indptr = d_indptr; B.d_indices = d_indices; B.d_values = d_values; B.d_tiles = d_tiles;
    B.n_tiles = n_tiles; B.values_complex = 0; B.lanes_per_row = lanes_per_row;
    return csr_spmv_any(B, d_x, d_y, accumulate, d_ws, stream, true, EvPair());
}

// The stage entry points.  `raw0`: first basis column that may be raw (deferred normalisation); the public entry
// points pass J -- they expect normalised columns; only aks_arnoldi_expand creates and reads raw ones.
static int gs_project_(int64_t n_rows, int32_t J, const aks_c128 *d_V, int64_t ldv, const aks_c128 *d_w,
                       void *d_ws, int64_t ws_bytes, int32_t max_dim, void *stream, int raw0, hipEvent_t ev0 = nullptr) {
    int rc = check_panel(n_rows, J, d_V, ldv, d_w, max_dim);
    if (rc != AKS_OK) return rc;
    Ws ws;
    rc = bind_ws(d_ws, ws_bytes, n_rows, max_dim, &ws);
    if (rc != AKS_OK) return rc;
    enqueue_projection(static_cast<hipStream_t>(stream), ws, n_rows, J, reinterpret_cast<const c128 *>(d_V), ldv,
                       reinterpret_cast<const c128 *>(d_w), ws.red1, ws.red3, raw0, ev0);
    AKS_CHECK_LAUNCH("aks_gs_project");
    return AKS_OK;
}

static int gs_update_project_(int64_t n_rows, int32_t J, const aks_c128 *d_V, int64_t ldv, aks_c128 *d_w,
                              void *d_ws, int64_t ws_bytes, int32_t max_dim, void *stream, int raw0) {
    int rc = check_panel(n_rows, J, d_V, ldv, d_w, max_dim);
    if (rc != AKS_OK) return rc;
    Ws ws;
    rc = bind_ws(d_ws, ws_bytes, n_rows, max_dim, &ws);
    if (rc != AKS_OK) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const c128 *V = reinterpret_cast<const c128 *>(d_V);
    c128 *w = reinterpret_cast<c128 *>(d_w);
    // With one workgroup per CU (panel_blocks) the exact-width kernel wins at every width it exists for (J <= 40):
    // against the column-split kernel -4...-7 % for J = 21..32 and -13...0 % for J = 5..12 at n = 10M, -1...-3 % at
    // n = 1.25M (profiles/r03_exact_split_ab.txt).  Under the 1024-workgroup geometry of rounds 1-2 the split kernel
    // had won for 5 <= J <= 12 and J > 20 (one wave per SIMD was then a loss, now it is the point).  The split kernel
    // serves J > 40.  (-DAKS_FUSED_EXACT_MAX=.. / -DAKS_FUSED_SPLIT_LOW=1 restore the old switch points for A/B builds.)
    constexpr int exact_max = FUSED_EXACT_MAX;
#ifndef AKS_FUSED_SPLIT_LOW
#define AKS_FUSED_SPLIT_LOW 0
#endif
    const bool exact = J <= exact_max && J <= 40 && !(AKS_FUSED_SPLIT_LOW && J >= 5 && J <= 12);
    const int n_blocks = exact ? panel_blocks(ws, J) : ws.lay.n_blocks;
    if (exact)
        dispatch_update_proj(J, dim3(n_blocks), s, n_rows, V, ldv, w, ws.red1, ws.partial,
                             ws.lay.ld_partial, ws.ctrl, ws.colscale, raw0);
    else
        dispatch_update_proj_split(J, dim3(n_blocks), s, n_rows, V, ldv, w, ws.red1, ws.partial,
                                   ws.lay.ld_partial, ws.ctrl, ws.colscale, raw0);
    hipLaunchKernelGGL(k_reduce<false>, dim3(J + 1), dim3(BLOCK), 0, s, ws.partial, n_blocks,
                       ws.lay.ld_partial, 0, ws.red2, J, nullptr, nullptr, 0.0, nullptr, ws.ctrl);
    AKS_CHECK_LAUNCH("aks_gs_update_project");
    return AKS_OK;
}

// Second pass + the norm behind it.  `fin_mode` / Hcol ... normalize: what the kernel does about the step's
// book-keeping (FIN_* above); ev1: the probe's stop event when this is the step's last launch.
static int gs_update_norm_(int64_t n_rows, int32_t J, const aks_c128 *d_V, int64_t ldv, aks_c128 *d_w, double eta,
                           void *d_ws, int64_t ws_bytes, int32_t max_dim, void *stream, int raw0,
                           int fin_mode = FIN_NONE, aks_c128 *d_Hcol = nullptr, int64_t ldh = 0, double tol = 0.0,
                           int32_t normalize = 0, hipEvent_t ev1 = nullptr) {
    int rc = check_panel(n_rows, J, d_V, ldv, d_w, max_dim);
    if (rc != AKS_OK) return rc;
    Ws ws;
    rc = bind_ws(d_ws, ws_bytes, n_rows, max_dim, &ws);
    if (rc != AKS_OK) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // two workgroups per CU: -0.9 % / -1.7 % per Gram-Schmidt step on the 3-D / 2-D Laplacian, whose every step runs this
    // kernel, against 1024 workgroups; one per CU: +6 % (profiles/r03_update_grid_ab.txt)
#ifdef AKS_UPDATE_BLOCKS
    const int n_blocks = AKS_UPDATE_BLOCKS < ws.lay.n_blocks ? AKS_UPDATE_BLOCKS : ws.lay.n_blocks;   // (A/B builds)
#else
    const int n_blocks = panel_blocks(ws, 4, 2);
#endif
    FinArgs fin;
    fin.Hcol = reinterpret_cast<c128 *>(d_Hcol);
    fin.ldh = ldh;
    fin.tol = tol;
    fin.cs = ws.colscale;
    fin.red3 = ws.red3;
    // the kernel sums its own norm partials (last-arriver ticket) exactly when it also books the step: only then does
    // that save launches (k_reduce<true> AND k_finish); a step that takes no second pass never reaches the ticket
    fin.ticket = fin_mode != FIN_NONE ? ws.ticket(TICKET_UPDATE) : nullptr;
    fin.normalize = normalize;
    fin.mode = fin_mode;
    const c128 *V = reinterpret_cast<const c128 *>(d_V);
    c128 *w = reinterpret_cast<c128 *>(d_w);
    launch_timed(k_update<true>, dim3(n_blocks), dim3(BLOCK), 0, s, (hipEvent_t) nullptr, ev1, n_rows, (int)J, V, ldv, w, ws.red2,
                 ws.partial, ws.lay.ld_partial, 0, ws.red1, ws.red2, eta, ws.ctrl, ws.colscale, raw0, fin);
    if (fin.ticket == nullptr)
        hipLaunchKernelGGL(k_reduce<true>, dim3(1), dim3(BLOCK), 0, s, ws.partial, n_blocks,
                           ws.lay.ld_partial, 0, ws.red3, J, ws.red1, ws.red2, eta, nullptr, ws.ctrl);
    AKS_CHECK_LAUNCH("aks_gs_update_norm");
    return AKS_OK;
}

int aks_gs_project(int64_t n_rows, int32_t J, const aks_c128 *d_V, int64_t ldv, const aks_c128 *d_w,
                   void *d_ws, int64_t ws_bytes, int32_t max_dim, void *stream) {
    return gs_project_(n_rows, J, d_V, ldv, d_w, d_ws, ws_bytes, max_dim, stream, J);
}

int aks_gs_update_project(int64_t n_rows, int32_t J, const aks_c128 *d_V, int64_t ldv, aks_c128 *d_w,
                          void *d_ws, int64_t ws_bytes, int32_t max_dim, void *stream) {
    return gs_update_project_(n_rows, J, d_V, ldv, d_w, d_ws, ws_bytes, max_dim, stream, J);
}

int aks_gs_update_norm(int64_t n_rows, int32_t J, const aks_c128 *d_V, int64_t ldv, aks_c128 *d_w, double eta,
                       void *d_ws, int64_t ws_bytes, int32_t max_dim, void *stream) {
    return gs_update_norm_(n_rows, J, d_V, ldv, d_w, eta, d_ws, ws_bytes, max_dim, stream, J);
}

static int gs_finish_(int64_t n_rows, int32_t J, aks_c128 *d_w, aks_c128 *d_Hcol, int64_t ldh, double tol,
                      double eta, int32_t normalize, void *d_ws, int64_t ws_bytes, int32_t max_dim, void *stream,
                      hipEvent_t ev1, int only_if_twice = 0) {
    if (d_w == nullptr || d_Hcol == nullptr) return fail(AKS_ERR_ARG, "null pointer");
    if (J < 1 || J > max_dim) return fail(AKS_ERR_ARG, "J must satisfy 1 <= J <= max_dim");
    if (ldh < 1) return fail(AKS_ERR_ARG, "ldh must be positive");
    if (normalize < 0 || normalize > 2) return fail(AKS_ERR_ARG, "normalize must be 0, 1 or 2");
    Ws ws;
    int rc = bind_ws(d_ws, ws_bytes, n_rows, max_dim, &ws);
    if (rc != AKS_OK) return rc;
    // (deferred normalisation: nothing of length n to do -- one block books H, beta and the column's scale)
    launch_timed(k_finish, dim3(normalize == 1 ? ws.lay.n_blocks : 1), dim3(BLOCK), 0, static_cast<hipStream_t>(stream),
                 (hipEvent_t) nullptr, ev1, n_rows, J, reinterpret_cast<c128 *>(d_w), reinterpret_cast<c128 *>(d_Hcol), ldh, tol, eta,
                 (int)normalize, ws.red1, ws.red2, ws.red3, ws.ctrl, ws.colscale, only_if_twice);
    AKS_CHECK_LAUNCH("k_finish");
    return AKS_OK;
}

int aks_gs_finish(int64_t n_rows, int32_t J, aks_c128 *d_w, aks_c128 *d_Hcol, int64_t ldh, double tol,
                  double eta, int32_t normalize, void *d_ws, int64_t ws_bytes, int32_t max_dim, void *stream) {
    return gs_finish_(n_rows, J, d_w, d_Hcol, ldh, tol, eta, normalize, d_ws, ws_bytes, max_dim, stream, nullptr);
}

static int dgks_gs_(int64_t n_rows, int32_t J, const aks_c128 *d_V, int64_t ldv, aks_c128 *d_w, aks_c128 *d_Hcol,
                    int64_t ldh, double tol, double eta, int32_t normalize, void *d_ws, int64_t ws_bytes,
                    int32_t max_dim, void *stream, int raw0, EvPair ev = EvPair()) {
    int rc = gs_project_(n_rows, J, d_V, ldv, d_w, d_ws, ws_bytes, max_dim, stream, raw0, ev.start);
    if (rc != AKS_OK) return rc;
    rc = gs_update_project_(n_rows, J, d_V, ldv, d_w, d_ws, ws_bytes, max_dim, stream, raw0);
    if (rc != AKS_OK) return rc;
    // normalize != 1: nothing of length n is left to do after the second pass, so the kernel that produces (or skips)
    // it books the step -- H column, beta, breakdown, counters -- and k_finish is not launched at all
    const bool fold = AKS_FOLD_FINISH && normalize != 1;
    rc = gs_update_norm_(n_rows, J, d_V, ldv, d_w, eta, d_ws, ws_bytes, max_dim, stream, raw0,
                         fold ? FIN_ALWAYS : FIN_NONE, d_Hcol, ldh, tol, normalize, fold ? ev.stop : nullptr);
    if (rc != AKS_OK || fold) return rc;
    return gs_finish_(n_rows, J, d_w, d_Hcol, ldh, tol, eta, normalize, d_ws, ws_bytes, max_dim, stream, ev.stop);
}

int aks_dgks_gs(int64_t n_rows, int32_t J, const aks_c128 *d_V, int64_t ldv, aks_c128 *d_w, aks_c128 *d_Hcol,
                int64_t ldh, double tol, double eta, int32_t normalize, void *d_ws, int64_t ws_bytes,
                int32_t max_dim, void *stream) {
    if (normalize == 2) return fail(AKS_ERR_ARG, "normalize = 2 (deferred) is reserved for aks_arnoldi_expand");
    return dgks_gs_(n_rows, J, d_V, ldv, d_w, d_Hcol, ldh, tol, eta, normalize, d_ws, ws_bytes, max_dim, stream, J);
}

// ---- tile-binned form: host planner --------------------------------------------------------
int aks_pb_params(int32_t *slab_bits, int32_t *rowblock_bits, int32_t *runs_per_round) {
    if (!slab_bits || !rowblock_bits || !runs_per_round) return fail(AKS_ERR_ARG, "null pointer");
    *slab_bits = PB_CW_BITS;
    *rowblock_bits = PB_RB_BITS;
    *runs_per_round = PB_RPR;
    return AKS_OK;
}

void *aks_pb_plan_create(const int32_t *indptr, const int32_t *indices, const void *values, int32_t values_complex,
                         int64_t n_rows, int64_t n_cols, aks_pb_sizes *sizes) try {
    if (!indptr || !indices || !values || !sizes) { fail(AKS_ERR_ARG, "null pointer"); return nullptr; }
    if (n_rows <= 0 || n_cols <= 0 || n_rows >= INT32_MAX || n_cols >= INT32_MAX) {
        fail(AKS_ERR_ARG, "matrix shape out of range");
        return nullptr;
    }
    const int64_t n_ss = (n_cols + PB_CW - 1) / PB_CW, n_rb = (n_rows + PB_RB - 1) >> PB_RB_BITS;
    if (n_ss * n_rb > ((int64_t)1 << 26)) {        // (checked before any array is read)
        fail(AKS_ERR_UNSUPPORTED, "too many (sub-slab, row block) tiles for the binned form");
        return nullptr;
    }
    const int64_t nnz = indptr[n_rows];
    if (nnz < 0 || nnz + 8 * n_ss > (int64_t)INT32_MAX - 64) {       // 32-bit entry positions in the kernels
        fail(AKS_ERR_UNSUPPORTED, "too many non-zeros for the binned form");
        return nullptr;
    }
    PbPlan *P = new PbPlan();
    std::unique_ptr<PbPlan> guard(P);
    P->values_complex = values_complex ? 1 : 0;
    // tile sizes
    PlanClock clock;
    std::vector<int32_t> cnt(n_ss * n_rb, 0), start(n_ss * n_rb);
    const int nt = plan_threads(n_rb, nnz);
    std::atomic<int> bad{0};                       // 1: indptr not monotone, 2: column out of range
    plan_parallel(n_rb, nt, [&](int, int64_t rb0, int64_t rb1) {        // a thread owns whole row blocks: disjoint counters
        const int64_t r_end = std::min<int64_t>(n_rows, rb1 << PB_RB_BITS);
        for (int64_t r = rb0 << PB_RB_BITS; r < r_end; ++r) {
            const int64_t rb = r >> PB_RB_BITS;
            if (indptr[r + 1] < indptr[r] || indptr[r] < 0 || (int64_t)indptr[r + 1] > nnz) { bad = 1; return; }
            for (int32_t k = indptr[r]; k < indptr[r + 1]; ++k) {
                const int32_t c = indices[k];
                if (c < 0 || c >= n_cols) { bad = 2; return; }
                ++cnt[(int64_t)(c / PB_CW) * n_rb + rb];
            }
        }
    });
    if (bad == 1) { fail(AKS_ERR_ARG, "indptr is not monotone"); return nullptr; }
    if (bad == 2) { fail(AKS_ERR_ARG, "column index out of range"); return nullptr; }
    if (clock.on) fprintf(stderr, "[aks plan] %d thread(s), %lld row blocks x %lld sub-slabs\n", nt, (long long)n_rb, (long long)n_ss);
    clock.lap("count");
    // phase-1 order: (sub-slab, row block, row, column); a sub-slab's slots start on a multiple of 8
    P->slab_begin.resize(n_ss);
    P->slab_end.resize(n_ss);
    int64_t pos = 0;
    for (int64_t s = 0; s < n_ss; ++s) {
        pos = (pos + 7) & ~(int64_t)7;
        P->slab_begin[s] = (int32_t)pos;
        for (int64_t rb = 0; rb < n_rb; ++rb) { start[s * n_rb + rb] = (int32_t)pos; pos += cnt[s * n_rb + rb]; }
        P->slab_end[s] = (int32_t)pos;
    }
    const int64_t nnz_pad = std::max<int64_t>((pos + 7) & ~(int64_t)7, 8);
    const int vw = values_complex ? 2 : 1;
    P->val.alloc((size_t)nnz_pad * vw);
    P->lcol.alloc(nnz_pad);
    RawArray<uint16_t> row13;                      // row inside its row block, phase-1 order
    row13.alloc(nnz_pad);
    for (int64_t s = 0; s < n_ss; ++s) {           // the padding slots behind a sub-slab (< 8 each): zero value, column 0
        const int64_t stop = s + 1 < n_ss ? P->slab_begin[s + 1] : nnz_pad;
        for (int64_t q = P->slab_end[s]; q < stop; ++q) {
            for (int j = 0; j < vw; ++j) P->val[(size_t)q * vw + j] = 0.0;
            P->lcol[q] = 0;
            row13[q] = 0;
        }
    }
    clock.lap("allocate");
    {
        std::vector<int32_t> cur(start);
        const double *vr = static_cast<const double *>(values);
        plan_parallel(n_rb, nt, [&](int, int64_t rb0, int64_t rb1) {    // tiles of a row block are filled by its owner only
            const int64_t r_end = std::min<int64_t>(n_rows, rb1 << PB_RB_BITS);
            for (int64_t r = rb0 << PB_RB_BITS; r < r_end; ++r) {
                const int64_t rb = r >> PB_RB_BITS;
                for (int32_t k = indptr[r]; k < indptr[r + 1]; ++k) {
                    const int32_t c = indices[k];
                    const int32_t q = cur[(int64_t)(c / PB_CW) * n_rb + rb]++;
                    P->lcol[q] = (uint16_t)(c % PB_CW);
                    row13[q] = (uint16_t)(r & (PB_RB - 1));
                    if (values_complex) {
                        P->val[2 * (size_t)q] = vr[2 * (size_t)k];
                        P->val[2 * (size_t)q + 1] = vr[2 * (size_t)k + 1];
                    } else {
                        P->val[q] = vr[k];
                    }
                }
            }
        });
    }
    clock.lap("place");
    // phase-2 schedule: wave-loads (up to 64 entries from up to AKS_PB_PIECES contiguous pieces), rounds, levels
    struct Piece { uint32_t start, len; };
    struct Load { Piece pc[AKS_PB_PIECES]; int n_pc; uint32_t total; };
    auto entry_of = [](const Load &L, uint32_t lane) {       // phase-1 position of a lane's entry
        for (int i = 0; i < L.n_pc; ++i) {
            if (lane < L.pc[i].len) return L.pc[i].start + lane;
            lane -= L.pc[i].len;
        }
        return 0u;
    };
    P->rb_run_ptr.resize(n_rb + 1);
    // row blocks are stored chunk by chunk, chunk c = row blocks c, c + n_chunks, ... (see k_pb_phase2)
    const int64_t n_chunks = std::min<int64_t>(n_rb, AKS_PB_CHUNKS);
    std::vector<int64_t> rb_at;
    for (int64_t c = 0; c < n_chunks; ++c)
        for (int64_t rb = c; rb < n_rb; rb += n_chunks) rb_at.push_back(rb);
    // every thread schedules a contiguous range of stored positions into arrays of its own; they are joined in order
    struct Part { std::vector<aks_pb_run> runs; std::vector<uint16_t> lrow; std::vector<int64_t> runs_before; };
    std::vector<Part> parts(nt);
    plan_parallel(n_rb, nt, [&](int t, int64_t pos0, int64_t pos1) {
    Part &part = parts[t];
    std::vector<aks_pb_run> &p_runs = part.runs;
    std::vector<uint16_t> &p_lrow = part.lrow;
    // per row of the block: the last entry of the current round that adds to it
    std::vector<int64_t> stamp(PB_RB, -1);
    std::vector<uint8_t> last_wave(PB_RB, 0), last_level(PB_RB, 0);
    std::vector<Load> loads;
    int64_t round_id = 0;
    for (int64_t pos = pos0; pos < pos1; ++pos) {
        const int64_t rb = rb_at[pos];
        part.runs_before.push_back((int64_t)p_runs.size());
        loads.clear();
        Load cur{};
        auto close = [&] { if (cur.total > 0) loads.push_back(cur); cur = Load{}; };
        for (int64_t s = 0; s < n_ss; ++s) {
            int32_t c = cnt[s * n_rb + rb], q = start[s * n_rb + rb];
            while (c > 0) {
                if (cur.n_pc == AKS_PB_PIECES || cur.total == AKS_PB_RUN_MAX) close();
                const int32_t take = std::min<int32_t>(c, AKS_PB_RUN_MAX - (int32_t)cur.total);
                cur.pc[cur.n_pc++] = Piece{(uint32_t)q, (uint32_t)take};
                cur.total += (uint32_t)take;
                q += take;
                c -= take;
            }
        }
        close();
        size_t next = 0;
        while (next < loads.size()) {
            // one round: consecutive wave-loads into slots 0 .. PB_RPR - 1 (slot j belongs to wave j / PB_K, its
            // k-th load).  Level of an entry = number of lower-index waves that add to its row in this round:
            // entries of one row then run in wave order, one barrier-separated level per wave; inside a wave they
            // run in program order (a wave's LDS operations complete in order; lanes of one ds_add that hit the
            // same address are serialised by the LDS in a fixed order).  With PB_W = 8 waves a level fits 3 bits.
            const size_t rr = p_runs.size(), wr = p_lrow.size();
            ++round_id;
            p_runs.resize(rr + PB_RPR, aks_pb_run{0u, 0u, 0u, 0u});
            p_lrow.resize(wr + PB_RW, (uint16_t)0);
            uint32_t levels = 1;
            for (int slot = 0; slot < PB_RPR && next < loads.size(); ++slot, ++next) {
                const Load &L = loads[next];
                const int w = slot / PB_K;
                aks_pb_run &run = p_runs[rr + slot];
                const uint32_t l0 = L.pc[0].len, l01 = l0 + (L.n_pc > 1 ? L.pc[1].len : 0u);
                run.start0 = L.pc[0].start;
                run.start1 = L.n_pc > 1 ? L.pc[1].start - l0 : 0u;
                run.start2 = L.n_pc > 2 ? L.pc[2].start - l01 : 0u;
                run.info = l0 | (l01 << 7) | (L.total << 14);
                for (uint32_t l = 0; l < L.total; ++l) {
                    const int row = row13[entry_of(L, l)];
                    uint32_t lv = 0;
                    if (stamp[row] == round_id) lv = last_wave[row] == w ? last_level[row] : last_level[row] + 1u;
                    stamp[row] = round_id;
                    last_wave[row] = (uint8_t)w;
                    last_level[row] = (uint8_t)lv;
                    p_lrow[wr + (size_t)w * (PB_K * AKS_PB_RUN_MAX) + (size_t)l * PB_K + (slot % PB_K)] = (uint16_t)(row | (lv << 13));
                    levels = std::max(levels, lv + 1u);
                }
            }
            for (int j = 0; j < PB_RPR; ++j) p_runs[rr + j].info |= levels << 21;
        }
        if ((size_t)part.runs_before.back() == p_runs.size()) {     // a row block without entries still owns one round
            p_runs.resize(p_runs.size() + PB_RPR, aks_pb_run{0u, 0u, 0u, 1u << 21});
            p_lrow.resize(p_lrow.size() + PB_RW, (uint16_t)0);
        }
        for (int j = 0; j < PB_RPR; ++j) p_runs[p_runs.size() - PB_RPR + j].info |= 1u << 25;   // last round of the block
    }
    });
    clock.lap("schedule");
    {   // join the parts: positions keep their order, run indices become global
        size_t n_runs_all = 0, n_lrow_all = 0;
        for (const Part &pt : parts) { n_runs_all += pt.runs.size(); n_lrow_all += pt.lrow.size(); }
        if (n_runs_all + PB_RPR >= (size_t)INT32_MAX / 2 || n_lrow_all + PB_RW >= (size_t)UINT32_MAX - 8) {
            fail(AKS_ERR_UNSUPPORTED, "binned form: schedule too large");
            return nullptr;
        }
        // one all-empty round behind the last row block: what the kernels' clamped prefetches of a row block
        // without rounds read
        P->runs.alloc(n_runs_all + PB_RPR);
        P->lrow.alloc(n_lrow_all + PB_RW);
        std::vector<size_t> run_at(nt + 1, 0), lrow_at(nt + 1, 0);
        int64_t pos = 0;
        for (int t = 0; t < nt; ++t) {
            for (int64_t before : parts[t].runs_before) P->rb_run_ptr[pos++] = (int32_t)((int64_t)run_at[t] + before);
            run_at[t + 1] = run_at[t] + parts[t].runs.size();
            lrow_at[t + 1] = lrow_at[t] + parts[t].lrow.size();
        }
        plan_parallel(nt, nt, [&](int, int64_t t0, int64_t t1) {
            for (int64_t t = t0; t < t1; ++t) {
                if (!parts[t].runs.empty()) memcpy(P->runs.data() + run_at[t], parts[t].runs.data(), parts[t].runs.size() * sizeof(aks_pb_run));
                if (!parts[t].lrow.empty()) memcpy(P->lrow.data() + lrow_at[t], parts[t].lrow.data(), parts[t].lrow.size() * sizeof(uint16_t));
                std::vector<aks_pb_run>().swap(parts[t].runs);
                std::vector<uint16_t>().swap(parts[t].lrow);
            }
        });
        for (int j = 0; j < PB_RPR; ++j) P->runs[n_runs_all + j] = aks_pb_run{0u, 0u, 0u, 0u};
        for (int j = 0; j < PB_RW; ++j) P->lrow[n_lrow_all + j] = (uint16_t)0;
        P->rb_run_ptr[n_rb] = (int32_t)n_runs_all;
    }
    clock.lap("join");
    P->sz.nnz_pad = nnz_pad;
    P->sz.n_runs = (int64_t)P->runs.size();
    P->sz.n_lrow = (int64_t)P->lrow.size();
    P->sz.n_slabs = (int32_t)n_ss;
    P->sz.n_rowblocks = (int32_t)n_rb;
    *sizes = P->sz;
    return guard.release();
} catch (const std::exception &e) {
    fail(AKS_ERR_ARG, e.what());      // e.g. std::bad_alloc: nothing may cross the C boundary
    return nullptr;
} catch (...) {
    fail(AKS_ERR_ARG, "unexpected C++ exception");
    return nullptr;
}

int aks_pb_plan_export(const void *plan, void *val_out, uint16_t *lcol_out, int32_t *slab_begin_out,
                       int32_t *slab_end_out, aks_pb_run *runs_out, int32_t *rb_run_ptr_out, uint16_t *lrow_out) {
    const PbPlan *P = static_cast<const PbPlan *>(plan);
    if (!P || !val_out || !lcol_out || !slab_begin_out || !slab_end_out || !runs_out || !rb_run_ptr_out || !lrow_out)
        return fail(AKS_ERR_ARG, "null pointer");
    PlanClock clock;
    plan_copy(val_out, P->val.data(), P->val.size() * sizeof(double));
    plan_copy(lcol_out, P->lcol.data(), P->lcol.size() * sizeof(uint16_t));
    memcpy(slab_begin_out, P->slab_begin.data(), P->slab_begin.size() * sizeof(int32_t));
    memcpy(slab_end_out, P->slab_end.data(), P->slab_end.size() * sizeof(int32_t));
    plan_copy(runs_out, P->runs.data(), P->runs.size() * sizeof(aks_pb_run));
    memcpy(rb_run_ptr_out, P->rb_run_ptr.data(), P->rb_run_ptr.size() * sizeof(int32_t));
    plan_copy(lrow_out, P->lrow.data(), P->lrow.size() * sizeof(uint16_t));
    clock.lap("export (copies)");
    return AKS_OK;
}

int aks_pb_plan_view(const void *plan, aks_pb_plan_arrays *out) {
    const PbPlan *P = static_cast<const PbPlan *>(plan);
    if (!P || !out) return fail(AKS_ERR_ARG, "null pointer");
    out->val = P->val.data();
    out->lcol = P->lcol.data();
    out->slab_begin = P->slab_begin.data();
    out->slab_end = P->slab_end.data();
    out->runs = P->runs.data();
    out->rb_run_ptr = P->rb_run_ptr.data();
    out->lrow = P->lrow.data();
    return AKS_OK;
}

void aks_pb_plan_destroy(void *plan) { delete static_cast<PbPlan *>(plan); }

int aks_pb_spmv(const aks_pb_matrix *A, const aks_c128 *d_x, aks_c128 *d_y, int32_t accumulate, const void *d_ws,
                void *stream) {
    return launch_pb_any(A, d_x, d_y, accumulate, d_ws, stream, false, nullptr, EvPair());
}

int aks_pb_spmv_real(const aks_pb_matrix *A, const double *d_x, double *d_y, int32_t accumulate, const void *d_ws,
                     void *stream) {
    return launch_pb_any(A, d_x, d_y, accumulate, d_ws, stream, true, nullptr, EvPair());
}

int aks_workspace_set_real(void *d_ws, int32_t real_packed, void *stream) {
    if (d_ws == nullptr) return fail(AKS_ERR_ARG, "workspace pointer is null");
    aks_ctrl *ctrl = static_cast<aks_ctrl *>(d_ws);
    hipError_t e = hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(&ctrl->real_mode), real_packed ? 1 : 0, 1,
                                     static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail(e, "hipMemsetD32Async(real_mode)");
    return AKS_OK;
}

// ---- sliced form: host helpers and entry points ---------------------------------------------
int64_t aks_sell_plan_size(const int32_t *indptr, int64_t n_rows) {
    if (indptr == nullptr || n_rows <= 0) return fail(AKS_ERR_ARG, "bad argument");
    int64_t total = 0;
    for (int64_t r0 = 0; r0 < n_rows; r0 += 64) {
        int64_t w = 0;
        for (int64_t r = r0; r < std::min(n_rows, r0 + 64); ++r) {
            if (indptr[r + 1] < indptr[r]) return fail(AKS_ERR_ARG, "indptr is not monotone");
            w = std::max<int64_t>(w, indptr[r + 1] - indptr[r]);
        }
        total += 64 * w;
    }
    return total;
}

int aks_sell_plan_fill(const int32_t *indptr, const int32_t *indices, const void *values, int32_t values_complex,
                       int64_t n_rows, int64_t *slice_ptr_out, int32_t *col_out, void *val_out) {
    if (!indptr || !indices || !values || !slice_ptr_out || n_rows <= 0) return fail(AKS_ERR_ARG, "bad argument");
    const int64_t n_slices = (n_rows + 63) / 64;
    const int64_t total = aks_sell_plan_size(indptr, n_rows);
    if (total < 0) return (int)total;
    if (total > 0 && (!col_out || !val_out)) return fail(AKS_ERR_ARG, "null output array");
    const size_t vw = values_complex ? 2 : 1;
    const double *vin = static_cast<const double *>(values);
    double *vout = static_cast<double *>(val_out);
    int64_t p = 0;
    for (int64_t s = 0; s < n_slices; ++s) {              // slice widths -> offsets (indptr only: a few milliseconds)
        slice_ptr_out[s] = p;
        int64_t w = 0;
        for (int64_t r = s * 64; r < std::min(n_rows, s * 64 + 64); ++r) w = std::max<int64_t>(w, indptr[r + 1] - indptr[r]);
        p += 64 * w;
    }
    slice_ptr_out[n_slices] = p;
    // every slice is written by one thread, entries and padding alike (no separate fill pass over the outputs)
    try {
        plan_parallel(n_slices, plan_threads(n_slices >> 4, indptr[n_rows]), [&](int, int64_t s0, int64_t s1) {
            for (int64_t s = s0; s < s1; ++s) {
                const int64_t p0 = slice_ptr_out[s], w = (slice_ptr_out[s + 1] - p0) >> 6;
                for (int64_t lane = 0; lane < 64; ++lane) {
                    const int64_t r = s * 64 + lane;
                    const int64_t len = r < n_rows ? (int64_t)indptr[r + 1] - indptr[r] : 0;
                    for (int64_t k = 0; k < len; ++k) {
                        const int64_t q = p0 + k * 64 + lane, src = indptr[r] + k;
                        col_out[q] = indices[src];
                        for (size_t c = 0; c < vw; ++c) vout[q * vw + c] = vin[src * vw + c];
                    }
                    for (int64_t k = len; k < w; ++k) {
                        const int64_t q = p0 + k * 64 + lane;
                        col_out[q] = -1;
                        for (size_t c = 0; c < vw; ++c) vout[q * vw + c] = 0.0;
                    }
                }
            }
        });
    } catch (const std::exception &e) {
        return fail(AKS_ERR_ARG, e.what());
    }
    return AKS_OK;
}

int aks_sell_spmv(const aks_sell_matrix *A, const aks_c128 *d_x, aks_c128 *d_y, int32_t accumulate, const void *d_ws,
                  void *stream) {
    return sell_spmv_any(A, d_x, d_y, accumulate, d_ws, stream, false, nullptr, EvPair());
}

int aks_sell_spmv_real(const aks_sell_matrix *A, const double *d_x, double *d_y, int32_t accumulate, const void *d_ws,
                       void *stream) {
    return sell_spmv_any(A, d_x, d_y, accumulate, d_ws, stream, true, nullptr, EvPair());
}

// ---- one-shot all-reduce: set-up, vote, tear-down ----------------------------------------------------------------------
struct OneShotCard {                     // what a rank tells its peers about its mailbox
    hipIpcMemHandle_t handle;
    unsigned long long address;          // the pointer itself: valid for peers inside the same process
    long long pid;
    int device, ok;
};

static void oneshot_release(Comm *c) {
    for (int p = 0; p < ONESHOT_MAX_RANKS; ++p)
        if (c->one.opened[p] != nullptr) { (void)hipIpcCloseMemHandle(c->one.opened[p]); c->one.opened[p] = nullptr; }
    if (c->one.local != nullptr) { (void)hipFree(c->one.local); c->one.local = nullptr; }
    if (c->one.status != nullptr) { (void)hipHostFree(c->one.status); c->one.status = nullptr; }
    c->one.active = false;
}

// min over the ranks of `mine` (0 / 1) through the library communicator: the ranks decide TOGETHER
static int oneshot_vote(Comm *c, double *d_flag, int mine, int *all) {
    const double v = mine ? 1.0 : 0.0;
    double got = 0.0;
    hipError_t e = hipMemcpyAsync(d_flag, &v, sizeof v, hipMemcpyHostToDevice, c->side);
    if (e == hipSuccess) e = hipStreamSynchronize(c->side);
    if (e != hipSuccess) return hip_fail(e, "oneshot vote");
    // sum of the yes-votes == size  <=>  everybody said yes
    ncclResult_t r = NCCL_CALL(AllReduce)(d_flag, d_flag, 1, ncclDouble, ncclSum, c->nccl, c->side);
    if (r != ncclSuccess) return nccl_fail(r, "ncclAllReduce(oneshot vote)");
    e = hipStreamSynchronize(c->side);
    if (e == hipSuccess) e = hipMemcpy(&got, d_flag, sizeof got, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return hip_fail(e, "oneshot vote");
    *all = got > c->size - 0.5;
    return AKS_OK;
}

static int oneshot_reduce(Comm *c, double *d_buf, int count, hipStream_t s);

static double env_ms(const char *name, double fallback) {
    const char *v = getenv(name);
    if (v == nullptr || *v == 0) return fallback;
    const double ms = atof(v);
    return ms > 0.0 ? ms : fallback;
}

static int oneshot_setup(Comm *c) {
    OneShot &o = c->one;
    int ok = 1, dev = 0, khz = 0;
    if (c->size > ONESHOT_MAX_RANKS) { ok = 0; o.why_not = "more ranks than ONESHOT_MAX_RANKS"; }
    if (ok && hipGetDevice(&dev) != hipSuccess) { ok = 0; o.why_not = "hipGetDevice failed"; (void)hipGetLastError(); }
    // the deadline of a reduction's wait, in ticks of the constant-rate clock wall_clock64() reads (100 MHz on gfx9)
    if (ok && hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) == hipSuccess && khz > 0) o.ticks_per_ms = (double)khz;
    (void)hipGetLastError();
    if (ok) {
        hipError_t e = hipExtMallocWithFlags(&o.local, oneshot_bytes(c->size), hipDeviceMallocFinegrained);
        if (e == hipSuccess) e = hipMemset(o.local, 0, oneshot_bytes(c->size));
        if (e != hipSuccess) { ok = 0; o.why_not = std::string("fine-grained mailbox: ") + hipGetErrorString(e); (void)hipGetLastError(); }
    }
    if (ok) {       // the status word: host memory the kernel writes at system scope and the host reads without a device call
        void *h = nullptr;
        hipError_t e = hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocCoherent);
        if (e != hipSuccess) { ok = 0; o.why_not = std::string("status word: ") + hipGetErrorString(e); (void)hipGetLastError(); }
        else { memset(h, 0, 64); o.status = static_cast<unsigned long long *>(h); }
    }
    // every rank's card to every rank (grouped send / recv on the side stream: the one collective form the library uses)
    OneShotCard mine = {};
    std::vector<OneShotCard> cards(c->size);
    mine.ok = ok;
    mine.pid = (long long)getpid();
    mine.device = dev;
    mine.address = (unsigned long long)reinterpret_cast<uintptr_t>(o.local);
    if (ok && hipIpcGetMemHandle(&mine.handle, o.local) != hipSuccess) {
        (void)hipGetLastError();                          // (only peers in OTHER processes need the handle: decided per peer below)
        memset(&mine.handle, 0, sizeof mine.handle);
        mine.ok = 2;                                      // usable inside this process only
    }
    static_assert(sizeof(OneShotCard) % 8 == 0, "cards travel as doubles");
    constexpr size_t CARD_D = sizeof(OneShotCard) / 8;
    double *d_cards = nullptr;
    hipError_t e = hipMalloc(&d_cards, (size_t)(c->size + 1) * sizeof(OneShotCard) + 8);
    if (e != hipSuccess) return hip_fail(e, "oneshot set-up");
    double *d_mine = d_cards + (size_t)c->size * CARD_D, *d_flag = d_mine + CARD_D;
    int rc = AKS_OK;
    e = hipMemcpy(d_mine, &mine, sizeof mine, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_cards + (size_t)c->rank * CARD_D, &mine, sizeof mine, hipMemcpyHostToDevice);
    if (e != hipSuccess) rc = hip_fail(e, "oneshot set-up");
    if (rc == AKS_OK && c->size > 1) {
        ncclResult_t r = NCCL_CALL(GroupStart)();
        for (int peer = 0; peer < c->size && r == ncclSuccess; ++peer) {
            if (peer == c->rank) continue;
            r = NCCL_CALL(Send)(d_mine, CARD_D, ncclDouble, peer, c->nccl, c->side);
            if (r == ncclSuccess) r = NCCL_CALL(Recv)(d_cards + (size_t)peer * CARD_D, CARD_D, ncclDouble, peer, c->nccl, c->side);
        }
        const ncclResult_t r2 = NCCL_CALL(GroupEnd)();
        if (r != ncclSuccess) rc = nccl_fail(r, "ncclSend/ncclRecv(oneshot cards)");
        else if (r2 != ncclSuccess) rc = nccl_fail(r2, "ncclGroupEnd(oneshot cards)");
    }
    if (rc == AKS_OK) {
        e = hipStreamSynchronize(c->side);
        if (e == hipSuccess) e = hipMemcpy(cards.data(), d_cards, (size_t)c->size * sizeof(OneShotCard), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = hip_fail(e, "oneshot set-up");
    }
    if (rc != AKS_OK) { (void)hipFree(d_cards); oneshot_release(c); return rc; }
    const char *same = getenv("AKS_ONESHOT_SAME_PROCESS");
    const bool allow_same_process = same != nullptr && strcmp(same, "1") == 0;
    for (int peer = 0; peer < c->size && ok; ++peer) {
        const OneShotCard &k = cards[peer];
        void *base = nullptr;
        if (!k.ok) { ok = 0; o.why_not = "rank " + std::to_string(peer) + " has no mailbox"; break; }
        if (peer == c->rank) base = o.local;
        else if (k.pid == mine.pid) {                     // thread ranks of one process: the pointer is the mapping
            if (!allow_same_process) {
                ok = 0;
                o.why_not = "ranks share a process: their streams can share a hardware queue, where a polling reduction sits in "
                            "front of the post it polls for and times out (AKS_ONESHOT_SAME_PROCESS=1 overrides)";
                break;
            }
            base = reinterpret_cast<void *>((uintptr_t)k.address);
            if (k.device != dev) (void)hipDeviceEnablePeerAccess(k.device, 0), (void)hipGetLastError();
        } else if (k.ok == 2 || mine.ok == 2) { ok = 0; o.why_not = "hipIpcGetMemHandle failed on a mailbox"; }
        else if (hipIpcOpenMemHandle(&base, k.handle, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
            ok = 0;
            o.why_not = std::string("hipIpcOpenMemHandle: ") + hipGetErrorString(hipGetLastError());
        } else o.opened[peer] = base;
        if (ok) {
            o.peers.flag[peer] = static_cast<unsigned long long *>(base);
            o.peers.box[peer] = reinterpret_cast<double *>(static_cast<char *>(base) + ONESHOT_FLAG_BYTES);
        }
    }
    int all = 0;
    rc = oneshot_vote(c, d_flag, ok, &all);
    if (rc == AKS_OK && all) {                            // prove it: sum of (rank + 1), twice (both parities)
        o.active = true;
        const char *fault = getenv("AKS_ONESHOT_FAULT_RANK");       // tests: what the peers of a rank whose posts never arrive do
        o.mute = fault != nullptr && *fault != 0 && atoi(fault) == c->rank;
        const double selftest_ms = env_ms("AKS_ONESHOT_SELFTEST_MS", 2000.0);
        o.deadline_ticks = (unsigned long long)(selftest_ms * o.ticks_per_ms);
        double got[2] = {0.0, 0.0};
        for (int rep = 0; rep < 2; ++rep) {           // (BOTH, whatever the first gave: every rank must post the same number of times)
            const double v = c->rank + 1.0;
            e = hipMemcpy(d_flag, &v, sizeof v, hipMemcpyHostToDevice);
            if (e == hipSuccess && oneshot_reduce(c, d_flag, 1, c->side) != AKS_OK) e = hipErrorUnknown;
            if (e == hipSuccess) e = hipStreamSynchronize(c->side);      // (bounded: the kernel's own deadline)
            if (e == hipSuccess && *static_cast<volatile unsigned long long *>(o.status) != 0ull) {
                ok = 0;                                   // the posts of some peer never arrived: a failed proof, not a hang
                if (o.why_not.empty())
                    o.why_not = "self-test: the arrival counter was not reached within " + std::to_string((long long)selftest_ms) +
                                " ms (posts of a peer not visible to this rank's reduction)";
                continue;
            }
            if (e == hipSuccess) e = hipMemcpy(&got[rep], d_flag, sizeof(double), hipMemcpyDeviceToHost);
            if (e != hipSuccess || got[rep] != c->size * (c->size + 1) / 2.0) {
                ok = 0;
                if (o.why_not.empty())                       // (the first reason stands)
                    o.why_not = e != hipSuccess ? std::string("self-test: ") + g_err : "self-test gave " + std::to_string(got[rep]);
                (void)hipGetLastError();
            }
        }
        rc = oneshot_vote(c, d_flag, ok, &all);
        o.deadline_ticks = (unsigned long long)(env_ms("AKS_ONESHOT_TIMEOUT_MS", 30000.0) * o.ticks_per_ms);
    }
    (void)hipFree(d_cards);
    if (rc != AKS_OK || !all) {
        if (o.why_not.empty()) o.why_not = "another rank could not set the one-shot all-reduce up";
        oneshot_release(c);
    }
    return rc;
}

static int oneshot_reduce(Comm *c, double *d_buf, int count, hipStream_t s) {
    OneShot &o = c->one;
    hipLaunchKernelGGL(k_oneshot_allreduce, dim3(1), dim3(BLOCK), 0, s, d_buf, count, o.peers, c->size, c->rank,
                       o.peers.flag[c->rank], static_cast<const double *>(o.peers.box[c->rank]), o.status, o.deadline_ticks,
                       o.mute ? 1 : 0);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "k_oneshot_allreduce");
    return AKS_OK;
}

// ---- communicator: RCCL + a side stream for the ghost exchange ------------------------------
int aks_comm_unique_id(void *id_out) {
    static_assert(sizeof(ncclUniqueId) <= AKS_COMM_ID_BYTES, "AKS_COMM_ID_BYTES too small");
    if (id_out == nullptr) return fail(AKS_ERR_ARG, "null pointer");
    if (rccl_load() != AKS_OK) return AKS_ERR_HIP;
    ncclUniqueId id;
    ncclResult_t r = NCCL_CALL(GetUniqueId)(&id);
    if (r != ncclSuccess) return nccl_fail(r, "ncclGetUniqueId");
    memset(id_out, 0, AKS_COMM_ID_BYTES);
    memcpy(id_out, &id, sizeof id);
    return AKS_OK;
}

int aks_comm_create(const void *id, int32_t rank, int32_t size, void **comm_out) {
    if (id == nullptr || comm_out == nullptr) return fail(AKS_ERR_ARG, "null pointer");
    if (size < 1 || rank < 0 || rank >= size) return fail(AKS_ERR_ARG, "need 0 <= rank < size");
    if (rccl_load() != AKS_OK) return AKS_ERR_HIP;
    Comm *c = new (std::nothrow) Comm();
    if (c == nullptr) return fail(AKS_ERR_ARG, "out of host memory");
    c->rank = rank;
    c->size = size;
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    ncclResult_t r = NCCL_CALL(CommInitRank)(&c->nccl, size, uid, rank);
    if (r != ncclSuccess) { delete c; return nccl_fail(r, "ncclCommInitRank"); }
    hipError_t e = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->packed, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->arrived, hipEventDisableTiming);
    if (e != hipSuccess) { (void)aks_comm_destroy(c); return hip_fail(e, "aks_comm_create"); }
    const char *want = getenv("AKS_ALLREDUCE");
    if (want != nullptr && strcmp(want, "oneshot") == 0) {
        const int rc = oneshot_setup(c);                 // AKS_OK also when the ranks agreed to stay with ncclAllReduce
        if (rc != AKS_OK) { (void)aks_comm_destroy(c); return rc; }
    }
    *comm_out = c;
    return AKS_OK;
}

int aks_comm_destroy(void *comm) {
    Comm *c = static_cast<Comm *>(comm);
    if (c == nullptr) return AKS_OK;
    if (const int alive = c->graphs.load()) {
        g_err = std::to_string(alive) + " hipGraph(s) that captured operations of this communicator are still alive: destroy them, "
                "call aks_comm_graph_release for each, then destroy the communicator (ncclCommDestroy does not return while a "
                "graph holds a captured send / recv)";
        return AKS_ERR_ARG;
    }
    oneshot_release(c);
    if (c->packed) (void)hipEventDestroy(c->packed);
    if (c->arrived) (void)hipEventDestroy(c->arrived);
    if (c->side) (void)hipStreamDestroy(c->side);
    if (c->nccl) (void)NCCL_CALL(CommDestroy)(c->nccl);
    delete c;
    return AKS_OK;
}

int aks_comm_allreduce_sum(void *comm, double *d_buf, int64_t count, void *stream) {
    Comm *c = static_cast<Comm *>(comm);
    if (c == nullptr || d_buf == nullptr || count < 1) return fail(AKS_ERR_ARG, "bad argument");
    if (c->one.active && count <= ONESHOT_CAP) return oneshot_reduce(c, d_buf, (int)count, static_cast<hipStream_t>(stream));
    ncclResult_t r = NCCL_CALL(AllReduce)(d_buf, d_buf, (size_t)count, ncclDouble, ncclSum, c->nccl, static_cast<hipStream_t>(stream));
    if (r != ncclSuccess) return nccl_fail(r, "ncclAllReduce");
    return AKS_OK;
}

int aks_comm_graph_retain(void *comm) {
    Comm *c = static_cast<Comm *>(comm);
    if (c == nullptr) return fail(AKS_ERR_ARG, "null communicator");
    return c->graphs.fetch_add(1) + 1;
}

int aks_comm_graph_release(void *comm) {
    Comm *c = static_cast<Comm *>(comm);
    if (c == nullptr) return fail(AKS_ERR_ARG, "null communicator");
    const int before = c->graphs.fetch_sub(1);
    if (before <= 0) { c->graphs.fetch_add(1); return fail(AKS_ERR_ARG, "aks_comm_graph_release without a matching retain"); }
    return before - 1;
}

int aks_comm_status(void *comm, char *why, int64_t why_bytes) {
    Comm *c = static_cast<Comm *>(comm);
    if (c == nullptr) return fail(AKS_ERR_ARG, "null communicator");
    if (why != nullptr && why_bytes > 0) why[0] = 0;
    if (!c->one.active || c->one.status == nullptr) return 0;
    const unsigned long long st = *static_cast<volatile unsigned long long *>(c->one.status);
    if (st == 0ull) return 0;
    if (why != nullptr && why_bytes > 0)
        snprintf(why, (size_t)why_bytes, "one-shot all-reduce number %llu of rank %d timed out after %.0f ms waiting for its peers' "
                 "posts (its result and every later one is NaN)", st >> 8, c->rank, (double)c->one.deadline_ticks / c->one.ticks_per_ms);
    return 1;
}

int aks_comm_allreduce_path(void *comm, char *why_not, int64_t why_bytes) {
    Comm *c = static_cast<Comm *>(comm);
    if (c == nullptr) return fail(AKS_ERR_ARG, "null communicator");
    if (why_not != nullptr && why_bytes > 0) snprintf(why_not, (size_t)why_bytes, "%s", c->one.why_not.c_str());
    return c->one.active ? 1 : 0;
}

int aks_comm_alltoallv(void *comm, const void *d_send, const int64_t *send_offsets, const int64_t *send_bytes,
                       void *d_recv, const int64_t *recv_offsets, const int64_t *recv_bytes, void *stream) {
    Comm *c = static_cast<Comm *>(comm);
    if (c == nullptr || !send_offsets || !send_bytes || !recv_offsets || !recv_bytes) return fail(AKS_ERR_ARG, "null pointer");
    int64_t total_s = 0, total_r = 0;
    for (int peer = 0; peer < c->size; ++peer) {     // validate BEFORE anything is enqueued (as shard_apply does)
        if (send_offsets[peer] < 0 || send_bytes[peer] < 0 || recv_offsets[peer] < 0 || recv_bytes[peer] < 0)
            return fail(AKS_ERR_ARG, "negative offset or size");
        total_s += send_bytes[peer];
        total_r += recv_bytes[peer];
    }
    if ((total_s > 0 && d_send == nullptr) || (total_r > 0 && d_recv == nullptr)) return fail(AKS_ERR_ARG, "null buffer");
    if (send_bytes[c->rank] != recv_bytes[c->rank]) return fail(AKS_ERR_ARG, "this rank's own slice: sizes differ");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const char *sb = static_cast<const char *>(d_send);
    char *rb = static_cast<char *>(d_recv);
    if (send_bytes[c->rank] > 0) {
        hipError_t e = hipMemcpyAsync(rb + recv_offsets[c->rank], sb + send_offsets[c->rank], (size_t)send_bytes[c->rank],
                                      hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) return hip_fail(e, "aks_comm_alltoallv(own slice)");
    }
    if (c->size == 1) return AKS_OK;
    ncclResult_t r = NCCL_CALL(GroupStart)();
    for (int peer = 0; peer < c->size && r == ncclSuccess; ++peer) {
        if (peer == c->rank) continue;
        if (send_bytes[peer] > 0) r = NCCL_CALL(Send)(sb + send_offsets[peer], (size_t)send_bytes[peer], ncclInt8, peer, c->nccl, s);
        if (recv_bytes[peer] > 0 && r == ncclSuccess)
            r = NCCL_CALL(Recv)(rb + recv_offsets[peer], (size_t)recv_bytes[peer], ncclInt8, peer, c->nccl, s);
    }
    const ncclResult_t r2 = NCCL_CALL(GroupEnd)();
    if (r != ncclSuccess) return nccl_fail(r, "ncclSend/ncclRecv");
    if (r2 != ncclSuccess) return nccl_fail(r2, "ncclGroupEnd");
    return AKS_OK;
}

// y = A x for one rank's rows.  With a communicator: pack -> [side stream: grouped send/recv] || diagonal block ->
// off-diagonal block.  ORDER OF THE COMMUNICATOR'S OPERATIONS: the all-reduces of the Gram-Schmidt stages (compute
// stream) and the exchanges (side stream) share ONE ncclComm_t, which RCCL allows as long as the operations
// are serialised -- and they are, by stream dependencies, not by luck: an exchange is issued behind the event
// `packed`, recorded on the compute stream after everything that precedes it there (the all-reduces of the step
// before included); the compute stream waits for `arrived`, recorded behind the exchange, before the off-diagonal
// block and hence before the next all-reduce.  Every rank issues the same sequence, so the n-th operation of the
// communicator is the same collective on every rank.  (tests/mock_rccl runs 2-4 ranks through exactly this code
// with a stand-in that aborts on any mismatch of order, peer or size.)
static int shard_apply(const aks_shard *A, const void *d_x, void *d_y, const void *d_ws, void *stream, int32_t flags,
                       Probe *pr, const double *x_div = nullptr) {
    if (A == nullptr || d_x == nullptr || d_y == nullptr) return fail(AKS_ERR_ARG, "null pointer");
    const bool real = (flags & AKS_EXPAND_REAL_PACKED) != 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    Comm *c = static_cast<Comm *>(A->comm);
    const bool exchange = c != nullptr && A->any_exchange != 0;
    if (!exchange)       // one block: the probe's event pair rides on its kernel launch(es)
        return apply_block(A->diag, d_x, d_y, 0, d_ws, stream, real, x_div, pr ? pr->reserve(AKS_PROBE_SPMV) : EvPair());
    hipEvent_t whole = pr ? pr->begin(AKS_PROBE_SPMV, s) : nullptr;     // sharded: pack .. off-diagonal block
    if (A->send_counts == nullptr || A->recv_counts == nullptr) return fail(AKS_ERR_ARG, "null exchange counts");
    if ((A->n_send > 0 && (!A->d_send_idx || !A->d_sendbuf)) || (A->n_ghost > 0 && !A->d_ghostbuf))
        return fail(AKS_ERR_ARG, "null exchange buffer");
    {   // validate the plan BEFORE anything is enqueued: a bad count must not leave a half-issued group behind
        int64_t so = 0, ro = 0;
        for (int peer = 0; peer < c->size; ++peer) {
            const int64_t ns = A->send_counts[peer], nr = A->recv_counts[peer];
            if (ns < 0 || nr < 0) return fail(AKS_ERR_ARG, "negative exchange count");
            so += ns;
            ro += nr;
        }
        if (so != A->n_send || ro != A->n_ghost) return fail(AKS_ERR_ARG, "exchange counts do not add up to n_send / n_ghost");
    }
    int rc = AKS_OK;
    hipEvent_t done = pr ? pr->begin(AKS_PROBE_PACK, s) : nullptr;
    if (A->n_send > 0) {                 // (entries of a raw column leave normalised: the receivers know no scale)
        const int64_t want = (A->n_send + BLOCK - 1) / BLOCK;
        const dim3 grid((unsigned)(want < 4096 ? want : 4096));
        if (real)
            hipLaunchKernelGGL(k_gather<double>, grid, dim3(BLOCK), 0, s, A->n_send, A->d_send_idx,
                               static_cast<const double *>(d_x), static_cast<double *>(A->d_sendbuf), x_div);
        else
            hipLaunchKernelGGL(k_gather<c128>, grid, dim3(BLOCK), 0, s, A->n_send, A->d_send_idx,
                               static_cast<const c128 *>(d_x), static_cast<c128 *>(A->d_sendbuf), x_div);
        hipError_t ge = hipGetLastError();
        if (ge != hipSuccess) rc = hip_fail(ge, "k_gather");
    }
    if (done) (void)hipEventRecord(done, s);
    if (rc != AKS_OK) return rc;
    hipError_t e = hipEventRecord(c->packed, s);
    if (e == hipSuccess) e = hipStreamWaitEvent(c->side, c->packed, 0);
    if (e != hipSuccess) return hip_fail(e, "aks_shard_apply(events)");
    done = pr ? pr->begin(AKS_PROBE_EXCHANGE, c->side) : nullptr;
    {   // all-to-all of the packed entries: every pair of ranks exchanges its slice, one group
        const size_t words = real ? 1 : 2;
        const double *sb = static_cast<const double *>(A->d_sendbuf);
        double *gb = static_cast<double *>(A->d_ghostbuf);
        ncclResult_t r = NCCL_CALL(GroupStart)();
        int64_t so = 0, ro = 0;
        for (int peer = 0; peer < c->size && r == ncclSuccess; ++peer) {
            const int64_t ns = A->send_counts[peer], nr = A->recv_counts[peer];
            if (ns > 0) r = NCCL_CALL(Send)(sb + so * words, (size_t)ns * words, ncclDouble, peer, c->nccl, c->side);
            if (nr > 0 && r == ncclSuccess) r = NCCL_CALL(Recv)(gb + ro * words, (size_t)nr * words, ncclDouble, peer, c->nccl, c->side);
            so += ns;
            ro += nr;
        }
        const ncclResult_t r2 = NCCL_CALL(GroupEnd)();
        if (r != ncclSuccess) return nccl_fail(r, "ncclSend/ncclRecv");
        if (r2 != ncclSuccess) return nccl_fail(r2, "ncclGroupEnd");
    }
    if (done) (void)hipEventRecord(done, c->side);
    e = hipEventRecord(c->arrived, c->side);
    if (e != hipSuccess) return hip_fail(e, "hipEventRecord");
    done = pr ? pr->begin(AKS_PROBE_DIAG, s) : nullptr;
    rc = apply_block(A->diag, d_x, d_y, 0, d_ws, stream, real, x_div);   // overlaps the exchange
    if (done) (void)hipEventRecord(done, s);
    if (rc != AKS_OK) return rc;
    done = pr ? pr->begin(AKS_PROBE_OFFDIAG, s) : nullptr;               // = wait for the ghosts + off-diagonal block
    e = hipStreamWaitEvent(s, c->arrived, 0);
    if (e != hipSuccess) return hip_fail(e, "hipStreamWaitEvent");
    if (A->off.n_rows > 0) rc = apply_block(A->off, A->d_ghostbuf, d_y, 1, d_ws, stream, real);
    if (done) (void)hipEventRecord(done, s);
    if (whole) (void)hipEventRecord(whole, s);
    return rc;
}

int aks_shard_apply(const aks_shard *A, const void *d_x, void *d_y, const void *d_ws, void *stream, int32_t flags) {
    return shard_apply(A, d_x, d_y, d_ws, stream, flags, nullptr);
}

int aks_arnoldi_expand(const aks_shard *A, aks_c128 *d_V, int64_t ldv, aks_c128 *d_H, int64_t ldh,
                       int32_t start_dim, int32_t end_dim, double tol, double eta, void *d_ws, int64_t ws_bytes,
                       int32_t max_dim, void *probe, void *stream, int32_t flags) {
    const bool first_w_ready = (flags & AKS_EXPAND_FROM_W) != 0, real = (flags & AKS_EXPAND_REAL_PACKED) != 0;
    const bool lazy_third = (flags & AKS_EXPAND_LAZY_THIRD) != 0;
    if (A == nullptr) return fail(AKS_ERR_ARG, "null operator");
    if (start_dim < 0 || end_dim > max_dim || start_dim > end_dim)
        return fail(AKS_ERR_ARG, "need 0 <= start_dim <= end_dim <= max_dim");
    if (d_V == nullptr || d_H == nullptr) return fail(AKS_ERR_ARG, "null pointer");
    if (ldh < max_dim) return fail(AKS_ERR_ARG, "ldh < max_dim");
    const int64_t n_rows = A->diag.n_rows;
    if (A->diag.n_cols != n_rows) return fail(AKS_ERR_ARG, "the diagonal block must be square");
    if (real && (A->diag.values_complex || (A->off.n_rows > 0 && A->off.values_complex)))
        return fail(AKS_ERR_ARG, "real-packed mode needs real matrix values");
    // real-packed: a column holds n_rows float64 = ceil(n_rows / 2) complex slots (an odd tail slot keeps Im = 0)
    const int64_t n_panel = real ? (n_rows + 1) / 2 : n_rows;
    if (ldv < n_panel) return fail(AKS_ERR_ARG, "ldv too small");
    Probe *pr = static_cast<Probe *>(probe);
    Comm *c = static_cast<Comm *>(A->comm);
    Ws ws;
    int rc = bind_ws(d_ws, ws_bytes, n_panel, max_dim, &ws);
    if (rc != AKS_OK) return rc;
    // Deferred normalisation: the new columns stay raw (k_finish books their scales instead of dividing), their
    // readers divide.  Only when this rank's diagonal block is in the binned form, whose phase 1 stages every x entry
    // exactly once, or in the sliced form with short rows (mean padded length <= AKS_SELL_DEFER_WIDTH): the sliced
    // and CSR-stream kernels gather an entry once per non-zero and divide that often.
    const bool defer = (flags & AKS_EXPAND_DEFER_SCALE) != 0 && block_defers(A->diag);
    const int norm_mode = defer ? 2 : 1;
    for (int32_t j = start_dim; j < end_dim; ++j) {
        const int32_t J = j + 1;
        aks_c128 *x = d_V + (int64_t)j * ldv;
        aks_c128 *w = d_V + (int64_t)J * ldv;
        const int raw0 = defer ? start_dim : J;         // columns < start_dim are normalised (precondition)
        if (!(first_w_ready && j == start_dim)) {
            rc = shard_apply(A, x, w, d_ws, stream, flags, pr, defer ? ws.colscale + j : nullptr);   // (books AKS_PROBE_SPMV)
            if (rc != AKS_OK) return rc;
        }
        // (the probe's pair for the orthogonalisation rides on its first and its last kernel)
        const EvPair eo = pr ? pr->reserve(AKS_PROBE_ORTHO) : EvPair();
        if (c == nullptr) {
            rc = dgks_gs_(n_panel, J, d_V, ldv, w, d_H + j, ldh, tol, eta, norm_mode, d_ws, ws_bytes, max_dim, stream, raw0, eo);
        } else {
            // the stage kernels with the reductions summed over the ranks in between
            // (with a probe: every reduction over the ranks between its own pair of recorded events, AKS_PROBE_ALLREDUCE)
            auto reduce_ranks = [&](c128 *slot, int64_t count) {
                hipEvent_t done = pr ? pr->begin(AKS_PROBE_ALLREDUCE, static_cast<hipStream_t>(stream)) : nullptr;
                const int r = aks_comm_allreduce_sum(c, reinterpret_cast<double *>(slot), count, stream);
                if (done) (void)hipEventRecord(done, static_cast<hipStream_t>(stream));
                return r;
            };
            rc = gs_project_(n_panel, J, d_V, ldv, w, d_ws, ws_bytes, max_dim, stream, raw0, eo.start);
            if (rc == AKS_OK) rc = reduce_ranks(ws.red1, 2 * (J + 1));
            if (rc == AKS_OK) rc = gs_update_project_(n_panel, J, d_V, ldv, w, d_ws, ws_bytes, max_dim, stream, raw0);
            if (rc == AKS_OK) rc = reduce_ranks(ws.red2, 2 * (J + 1));
            // the step's book-keeping rides on the second-pass kernel when no n-sized normalisation follows: always with
            // the lazy third all-reduce (a step that does take the second pass makes the caller repeat the expansion
            // anyway), otherwise only for the steps that need no second pass -- after one, the norm is summed over the
            // ranks first and k_finish books the step
            const bool fold = AKS_FOLD_FINISH && norm_mode != 1;
            const int fin_mode = !fold ? FIN_NONE : (lazy_third ? FIN_ALWAYS : FIN_IF_ONCE);
            const bool last = fin_mode == FIN_ALWAYS;
            if (rc == AKS_OK) rc = gs_update_norm_(n_panel, J, d_V, ldv, w, eta, d_ws, ws_bytes, max_dim, stream, raw0, fin_mode,
                                                   d_H + j, ldh, tol, norm_mode, last ? eo.stop : nullptr);
            if (rc == AKS_OK && !lazy_third) rc = reduce_ranks(ws.red3, 2);
            if (rc == AKS_OK && !last)
                rc = gs_finish_(n_panel, J, w, d_H + j, ldh, tol, eta, norm_mode, d_ws, ws_bytes, max_dim, stream, eo.stop,
                                fin_mode == FIN_IF_ONCE ? 1 : 0);
        }
        if (rc != AKS_OK) return rc;
    }
    return AKS_OK;
}

int aks_shard_apply_col(const aks_shard *A, const aks_c128 *d_V, int64_t ldv, int32_t col, void *d_y, void *d_ws,
                        int64_t ws_bytes, int32_t max_dim, void *stream, int32_t flags) {
    if (A == nullptr || d_V == nullptr) return fail(AKS_ERR_ARG, "null pointer");
    if (col < 0 || col > max_dim) return fail(AKS_ERR_ARG, "need 0 <= col <= max_dim");
    const bool real = (flags & AKS_EXPAND_REAL_PACKED) != 0;
    const int64_t n_panel = real ? (A->diag.n_rows + 1) / 2 : A->diag.n_rows;
    Ws ws;
    int rc = bind_ws(d_ws, ws_bytes, n_panel, max_dim, &ws);
    if (rc != AKS_OK) return rc;
    // (the scale is passed on only where it can be applied; a form that cannot never sees a raw column, see
    // aks_arnoldi_expand)
    return shard_apply(A, d_V + (int64_t)col * ldv, d_y, d_ws, stream, flags, nullptr,
                       block_defers(A->diag) ? ws.colscale + col : nullptr);
}

int aks_truncate_ws(int64_t n_rows, int32_t m, int32_t p, aks_c128 *d_V, int64_t ldv, const aks_c128 *d_Qp,
                    int32_t col0, void *d_ws, int64_t ws_bytes, int32_t max_dim, void *stream) {
    if (n_rows <= 0 || d_V == nullptr || d_Qp == nullptr) return fail(AKS_ERR_ARG, "bad argument");
    if (m < 1 || m > AKS_MAX_DIM) return fail(AKS_ERR_UNSUPPORTED, "m outside [1, AKS_MAX_DIM]");
    if (p < 1 || p >= m + 1) return fail(AKS_ERR_ARG, "need 1 <= p <= m");
    if (p > AKS_MAX_TRUNC) return fail(AKS_ERR_UNSUPPORTED, "p exceeds AKS_MAX_TRUNC");
    if (ldv < n_rows) return fail(AKS_ERR_ARG, "ldv < n_rows");
    if (col0 < 0 || col0 + m > max_dim) return fail(AKS_ERR_ARG, "need 0 <= col0 and col0 + m <= max_dim");
    Ws ws;
    int rc = bind_ws(d_ws, ws_bytes, n_rows, max_dim, &ws);
    if (rc != AKS_OK) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    c128 *V = reinterpret_cast<c128 *>(d_V);
    const c128 *Q = reinterpret_cast<const c128 *>(d_Qp);
    rc = AKS_ERR_UNSUPPORTED;
    switch ((p + 7) / 8) {
#define M(N) case N: rc = launch_truncate_mfma<N>(s, n_rows, m, p, V, ldv, Q, V, ldv, 1); break;
        M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12)
#undef M
        default: return fail(AKS_ERR_UNSUPPORTED, "p exceeds AKS_MAX_TRUNC");
    }
    if (rc != AKS_OK) return rc;
    static_assert(AKS_MAX_DIM + 1 <= BLOCK, "one lane per basis column");
    hipLaunchKernelGGL(k_colscale_after_truncate, dim3(1), dim3(BLOCK), 0, s, ws.colscale + col0, (int)m, (int)p);
    AKS_CHECK_LAUNCH("k_colscale_after_truncate");
    return AKS_OK;
}

int aks_truncate(int64_t n_rows, int32_t m, int32_t p, aks_c128 *d_V, int64_t ldv, const aks_c128 *d_Qp,
                 void *stream) {
    if (n_rows <= 0 || d_V == nullptr || d_Qp == nullptr) return fail(AKS_ERR_ARG, "bad argument");
    if (m < 1 || m > AKS_MAX_DIM) return fail(AKS_ERR_UNSUPPORTED, "m outside [1, AKS_MAX_DIM]");
    if (p < 1 || p >= m + 1) return fail(AKS_ERR_ARG, "need 1 <= p <= m");
    if (p > AKS_MAX_TRUNC) return fail(AKS_ERR_UNSUPPORTED, "p exceeds AKS_MAX_TRUNC");
    if (ldv < n_rows) return fail(AKS_ERR_ARG, "ldv < n_rows");
    hipStream_t s = static_cast<hipStream_t>(stream);
    c128 *V = reinterpret_cast<c128 *>(d_V);
    const c128 *Q = reinterpret_cast<const c128 *>(d_Qp);
    switch ((p + 7) / 8) {   // M-tiles of 16 real = 8 complex output columns
#define M(N) case N: return launch_truncate_mfma<N>(s, n_rows, m, p, V, ldv, Q, V, ldv, 1);
        M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12)
#undef M
        default: break;
    }
    return fail(AKS_ERR_UNSUPPORTED, "p exceeds AKS_MAX_TRUNC");
}

int aks_combine(int64_t n_rows, int32_t m, int32_t q, const aks_c128 *d_V, int64_t ldv, const aks_c128 *d_S,
                aks_c128 *d_out, int64_t ldo, void *stream) {
    if (n_rows <= 0 || d_V == nullptr || d_S == nullptr || d_out == nullptr) return fail(AKS_ERR_ARG, "bad argument");
    if (m < 1 || m > AKS_MAX_DIM) return fail(AKS_ERR_UNSUPPORTED, "m outside [1, AKS_MAX_DIM]");
    if (q < 1) return fail(AKS_ERR_ARG, "need q >= 1");
    if (q > AKS_MAX_TRUNC) return fail(AKS_ERR_UNSUPPORTED, "q exceeds AKS_MAX_TRUNC");
    if (ldv < n_rows || ldo < n_rows) return fail(AKS_ERR_ARG, "leading dimension < n_rows");
    {   // out of place only (a wave may overwrite rows another wave has not read yet unless O == V exactly)
        const char *v0 = reinterpret_cast<const char *>(d_V), *v1 = v0 + ((int64_t)(m - 1) * ldv + n_rows) * 16;
        const char *o0 = reinterpret_cast<const char *>(d_out), *o1 = o0 + ((int64_t)(q - 1) * ldo + n_rows) * 16;
        if (o0 < v1 && v0 < o1) return fail(AKS_ERR_ARG, "out overlaps V (use aks_truncate for the in-place form)");
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    c128 *V = const_cast<c128 *>(reinterpret_cast<const c128 *>(d_V));   // only read when copy_last == 0
    const c128 *S = reinterpret_cast<const c128 *>(d_S);
    c128 *O = reinterpret_cast<c128 *>(d_out);
    switch ((q + 7) / 8) {
#define M(N) case N: return launch_truncate_mfma<N>(s, n_rows, m, q, V, ldv, S, O, ldo, 0);
        M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12)
#undef M
        default: break;
    }
    return fail(AKS_ERR_UNSUPPORTED, "q exceeds AKS_MAX_TRUNC");
}

int aks_scale(int64_t n_rows, aks_c128 *d_w, double alpha_re, double alpha_im, void *stream) {
    if (n_rows <= 0 || d_w == nullptr) return fail(AKS_ERR_ARG, "bad argument");
    const int64_t want = (n_rows + BLOCK - 1) / BLOCK;
    const dim3 grid((unsigned)(want < 4096 ? want : 4096));
    hipLaunchKernelGGL(k_scale, grid, dim3(BLOCK), 0, static_cast<hipStream_t>(stream), n_rows,
                       reinterpret_cast<c128 *>(d_w), make_double2(alpha_re, alpha_im));
    AKS_CHECK_LAUNCH("k_scale");
    return AKS_OK;
}

int aks_stream_copy(void *d_dst, const void *d_src, int64_t bytes, void *stream) {
    if (d_dst == nullptr || d_src == nullptr || bytes < 16 || (bytes & 15) != 0) return fail(AKS_ERR_ARG, "need two buffers and a multiple of 16 bytes");
    if ((reinterpret_cast<uintptr_t>(d_dst) | reinterpret_cast<uintptr_t>(d_src)) & 15) return fail(AKS_ERR_ARG, "buffers must be 16-byte aligned");
    const int64_t items = bytes / 16, want = (items + BLOCK - 1) / BLOCK;
    const dim3 grid((unsigned)(want < 8192 ? want : 8192));
    hipLaunchKernelGGL(k_stream_copy, grid, dim3(BLOCK), 0, static_cast<hipStream_t>(stream), items,
                       static_cast<const c128 *>(d_src), static_cast<c128 *>(d_dst));
    AKS_CHECK_LAUNCH("k_stream_copy");
    return AKS_OK;
}

int aks_runtime_versions(int32_t *hip_runtime, int32_t *hip_driver, int32_t *rccl) {
    int v = 0;
    if (hip_runtime != nullptr) { *hip_runtime = hipRuntimeGetVersion(&v) == hipSuccess ? v : -1; }
    if (hip_driver != nullptr) { *hip_driver = hipDriverGetVersion(&v) == hipSuccess ? v : -1; }
    (void)hipGetLastError();
    if (rccl != nullptr) {
        *rccl = -1;                                      // not loaded (a one-GPU process never loads librccl)
#ifndef AKS_RCCL_DIRECT
        if (g_rccl.AllReduce != nullptr && g_rccl.GetVersion != nullptr && g_rccl.GetVersion(&v) == ncclSuccess) *rccl = v;
#else
        *rccl = 0;                                       // tests/mock_rccl: the stand-in
#endif
    }
    return AKS_OK;
}

int aks_probe_create(int32_t capacity, void **probe_out) try {
    if (capacity < 1 || probe_out == nullptr) return fail(AKS_ERR_ARG, "bad probe capacity / pointer");
    Probe *p = new (std::nothrow) Probe();
    if (p == nullptr) return fail(AKS_ERR_ARG, "out of host memory");
    p->start.resize(capacity);
    p->stop.resize(capacity);
    p->tag.assign(capacity, -1);
    for (int32_t i = 0; i < capacity; ++i) {
        hipError_t e = hipEventCreate(&p->start[i]);
        if (e == hipSuccess) e = hipEventCreate(&p->stop[i]);
        if (e != hipSuccess) {
            delete p;
            return hip_fail(e, "hipEventCreate");
        }
    }
    *probe_out = p;
    return AKS_OK;
} catch (const std::exception &e) {
    return fail(AKS_ERR_ARG, e.what());      // e.g. std::bad_alloc: nothing may cross the C boundary
} catch (...) {
    return fail(AKS_ERR_ARG, "unexpected C++ exception");
}

int aks_probe_destroy(void *probe) {
    Probe *p = static_cast<Probe *>(probe);
    if (p == nullptr) return AKS_OK;
    for (size_t i = 0; i < p->start.size(); ++i) {
        (void)hipEventDestroy(p->start[i]);
        (void)hipEventDestroy(p->stop[i]);
    }
    delete p;
    return AKS_OK;
}

int aks_probe_reset(void *probe) {
    if (probe == nullptr) return fail(AKS_ERR_ARG, "null probe");
    static_cast<Probe *>(probe)->used = 0;
    return AKS_OK;
}

int aks_probe_read(void *probe, int32_t tag, int32_t *count_out, double *total_ms_out) {
    Probe *p = static_cast<Probe *>(probe);
    if (p == nullptr || count_out == nullptr || total_ms_out == nullptr) return fail(AKS_ERR_ARG, "null pointer");
    int32_t count = 0;
    double total = 0.0;
    if (p->used > 0) {
        hipError_t e = hipEventSynchronize(p->stop[p->used - 1]);
        if (e != hipSuccess) return hip_fail(e, "hipEventSynchronize");
    }
    for (int32_t i = 0; i < p->used; ++i) {
        if (p->tag[i] != tag) continue;
        float ms = 0.f;
        hipError_t e = hipEventElapsedTime(&ms, p->start[i], p->stop[i]);
        if (e != hipSuccess) return hip_fail(e, "hipEventElapsedTime");
        total += ms;
        ++count;
    }
    *count_out = count;
    *total_ms_out = total;
    return AKS_OK;
}

int aks_gather_c128(int64_t count, const int32_t *d_idx, const aks_c128 *d_src, aks_c128 *d_dst, void *stream) {
    if (count == 0) return AKS_OK;
    if (count < 0 || !d_idx || !d_src || !d_dst) return fail(AKS_ERR_ARG, "bad argument");
    const int64_t want = (count + BLOCK - 1) / BLOCK;
    const dim3 grid((unsigned)(want < 4096 ? want : 4096));
    hipLaunchKernelGGL(k_gather<c128>, grid, dim3(BLOCK), 0, static_cast<hipStream_t>(stream), count, d_idx,
                       reinterpret_cast<const c128 *>(d_src), reinterpret_cast<c128 *>(d_dst), nullptr);
    AKS_CHECK_LAUNCH("k_gather");
    return AKS_OK;
}

int aks_gather_f64(int64_t count, const int32_t *d_idx, const double *d_src, double *d_dst, void *stream) {
    if (count == 0) return AKS_OK;
    if (count < 0 || !d_idx || !d_src || !d_dst) return fail(AKS_ERR_ARG, "bad argument");
    const int64_t want = (count + BLOCK - 1) / BLOCK;
    const dim3 grid((unsigned)(want < 4096 ? want : 4096));
    hipLaunchKernelGGL(k_gather<double>, grid, dim3(BLOCK), 0, static_cast<hipStream_t>(stream), count, d_idx, d_src,
                       d_dst, nullptr);
    AKS_CHECK_LAUNCH("k_gather_f64");
    return AKS_OK;
}

}  // extern "C"
