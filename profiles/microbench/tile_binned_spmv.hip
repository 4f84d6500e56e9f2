// Micro-benchmark (gfx950): the tile-binned two-phase SpMV (round 2 design) on the real random-CSR
// structure of BASELINE config 5 (n = 10M, 5 non-zeros per row), next to the streaming ceilings of
// its two phases and to the direct-gather pattern it replaces.
//
//   phase 1  one workgroup per sub-slab of 8192 columns: the sub-slab's x entries (128 KiB) are staged
//            in LDS, the sub-slab's entries (ordered by (row block, row)) are streamed -- val f64,
//            lcol u16 -- and val * x[lcol] is written SEQUENTIALLY (no per-entry destination);
//   phase 2  one workgroup (W waves) per block of 8192 rows; wave w owns rows [w*8192/W, (w+1)*8192/W)
//            and LDS float64 accumulators for them (wave-private => deterministic).  The block's
//            products are W-piece runs, one run per sub-slab; a lane takes one run (segment) at a time:
//            segment record = first product index + the W+1 piece offsets.
//
// hipcc --offload-arch=gfx950 -O3 -o tile_binned_spmv tile_binned_spmv.hip && ./tile_binned_spmv [n] [per_row]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef double2 c128;
typedef double v2d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ c128 nt_load(const c128 *p) { const v2d v = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(p)); return make_double2(v.x, v.y); }
__device__ __forceinline__ void nt_store(c128 v, c128 *p) { v2d t; t.x = v.x; t.y = v.y; __builtin_nontemporal_store(t, reinterpret_cast<v2d *>(p)); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int CW_BITS = 13, CW = 1 << CW_BITS;   // columns per sub-slab (x slice in LDS: 128 KiB)
constexpr int RB_BITS = 13, RB = 1 << RB_BITS;   // rows per phase-2 workgroup (accumulators: 128 KiB)
constexpr int SEG_MAX = 64;                       // entries per segment record

struct Plan {
    int64_t n_rows, n_cols, nnz, nnz_pad, n_seg;
    int n_ss, n_rb, W, rec_bytes;
    std::vector<double> val;
    std::vector<uint16_t> lcol, lrow;
    std::vector<int32_t> ss_begin, ss_end, rb_seg_ptr;
    std::vector<uint8_t> segs;
};

static void make_plan(int64_t n_rows, int64_t n_cols, const int32_t *indptr, const int32_t *indices,
                      const double *values, int W, Plan &P) {
    P.n_rows = n_rows; P.n_cols = n_cols; P.nnz = indptr[n_rows]; P.W = W;
    P.rec_bytes = W <= 8 ? 16 : 32;
    P.n_ss = (int)((n_cols + CW - 1) >> CW_BITS);
    P.n_rb = (int)((n_rows + RB - 1) >> RB_BITS);
    const int64_t n_tiles = (int64_t)P.n_ss * P.n_rb;
    std::vector<int32_t> cnt(n_tiles, 0), start(n_tiles);
    for (int64_t r = 0; r < n_rows; ++r) {
        const int64_t rb = r >> RB_BITS;
        for (int32_t k = indptr[r]; k < indptr[r + 1]; ++k) ++cnt[(int64_t)(indices[k] >> CW_BITS) * P.n_rb + rb];
    }
    P.ss_begin.resize(P.n_ss); P.ss_end.resize(P.n_ss);
    int64_t pos = 0;
    for (int s = 0; s < P.n_ss; ++s) {
        pos = (pos + 7) & ~(int64_t)7;                 // a sub-slab's products start on a 128-byte line
        P.ss_begin[s] = (int32_t)pos;
        for (int rb = 0; rb < P.n_rb; ++rb) { start[(int64_t)s * P.n_rb + rb] = (int32_t)pos; pos += cnt[(int64_t)s * P.n_rb + rb]; }
        P.ss_end[s] = (int32_t)pos;
    }
    P.nnz_pad = (pos + 7) & ~(int64_t)7;
    P.val.assign(P.nnz_pad, 0.0); P.lcol.assign(P.nnz_pad, 0); P.lrow.assign(P.nnz_pad, 0);
    {
        std::vector<int32_t> cur(start);
        for (int64_t r = 0; r < n_rows; ++r) {
            const int64_t rb = r >> RB_BITS;
            for (int32_t k = indptr[r]; k < indptr[r + 1]; ++k) {
                const int32_t c = indices[k];
                const int32_t q = cur[(int64_t)(c >> CW_BITS) * P.n_rb + rb]++;
                P.val[q] = values[k]; P.lcol[q] = (uint16_t)(c & (CW - 1)); P.lrow[q] = (uint16_t)(r & (RB - 1));
            }
        }
    }
    const int sb_bits = RB_BITS - (W == 8 ? 3 : 4);
    P.rb_seg_ptr.resize(P.n_rb + 1);
    P.segs.clear();
    int64_t n_seg = 0;
    for (int rb = 0; rb < P.n_rb; ++rb) {
        P.rb_seg_ptr[rb] = (int32_t)n_seg;
        for (int s = 0; s < P.n_ss; ++s) {
            const int64_t t = (int64_t)s * P.n_rb + rb;
            const int32_t c = cnt[t], q0 = start[t];
            for (int32_t o = 0; o < c; o += SEG_MAX) {
                const int len = std::min(SEG_MAX, c - o);
                const size_t at = P.segs.size();
                P.segs.resize(at + P.rec_bytes, 0);
                const uint32_t first = (uint32_t)(q0 + o);
                memcpy(&P.segs[at], &first, 4);
                int i = 0;
                for (int w = 0; w <= W; ++w) {       // off[w] = entries of the segment in sub-blocks < w
                    while (i < len && (P.lrow[q0 + o + i] >> sb_bits) < w) ++i;
                    P.segs[at + 4 + w] = (uint8_t)i;
                }
                ++n_seg;
            }
        }
    }
    P.rb_seg_ptr[P.n_rb] = (int32_t)n_seg;
    P.n_seg = n_seg;
}


// ---- phase 2, shared accumulators ("levels"): all W waves of the row block's workgroup add into ONE set of
// 8192 complex accumulators.  A wave-load takes one run (<= 64 entries of one tile, contiguous); a round is
// W*K consecutive runs of the row block, K per wave.  Two entries of one round that hit the same row from
// different waves get different LEVELS (3 bits on top of lrow): the round's adds go level by level with a
// workgroup barrier after each, so every row sees its adds in one fixed order => bitwise reproducible.
struct PlanL {
    int RPR;                                  // runs per round
    std::vector<uint32_t> run_start;          // per run: first entry (phase-1 order)
    std::vector<uint8_t> run_len;             // per run: entries (0 = padding run)
    std::vector<int32_t> rb_run_ptr;          // n_rb + 1: first run of each row block (multiple of RPR runs each)
    std::vector<uint8_t> round_phases;        // per round (run index / RPR): levels used
    std::vector<uint16_t> lrow_lv;            // nnz_pad: level << 13 | row in block
    std::vector<uint2> desc;                  // per run: x = first entry, y = len | phases of its round << 8
    double mean_phases;
};

static void make_plan_L(const Plan &P, int W, int K, PlanL &L, bool strict = false) {
    const int RPR = W * K;
    L.RPR = RPR;
    L.lrow_lv.assign(P.lrow.begin(), P.lrow.end());
    L.rb_run_ptr.resize(P.n_rb + 1);
    std::vector<uint8_t> seen(RB), seen_round_lo(RB);
    std::vector<int32_t> stamp(RB, -1);
    const int RS = P.rec_bytes;
    int64_t rounds = 0, phases = 0;
    for (int rb = 0; rb < P.n_rb; ++rb) {
        L.rb_run_ptr[rb] = (int32_t)L.run_start.size();
        for (int g = P.rb_seg_ptr[rb]; g < P.rb_seg_ptr[rb + 1]; ++g) {
            uint32_t q0;
            memcpy(&q0, &P.segs[(size_t)g * RS], 4);
            L.run_start.push_back(q0);
            L.run_len.push_back(P.segs[(size_t)g * RS + 4 + P.W]);
        }
        while ((L.run_start.size() - L.rb_run_ptr[rb]) % RPR) { L.run_start.push_back(0); L.run_len.push_back(0); }
        const int r0 = L.rb_run_ptr[rb], r1 = (int)L.run_start.size();
        for (int rr = r0; rr < r1; rr += RPR) {
            const int round_id = rr / RPR;
            // pass A: which waves touch each row in this round
            for (int j = 0; j < RPR; ++j) {
                const int w = j / K;
                for (int i = 0; i < L.run_len[rr + j]; ++i) {
                    const int row = P.lrow[L.run_start[rr + j] + i];
                    if (stamp[row] != round_id) { stamp[row] = round_id; seen[row] = 0; }
                    seen[row] |= (uint8_t)(1u << w);
                }
            }
            int maxlv = -1;
            if (strict)
                for (int j = 0; j < RPR; ++j)
                    for (int i = 0; i < L.run_len[rr + j]; ++i) seen[P.lrow[L.run_start[rr + j] + i]] = 0;
            for (int j = 0; j < RPR; ++j) {
                const int w = j / K;
                for (int i = 0; i < L.run_len[rr + j]; ++i) {
                    const uint32_t q = L.run_start[rr + j] + i;
                    const int row = P.lrow[q];
                    int lv;
                    if (strict) { lv = seen[row]++; if (lv > 7) { printf("level overflow\n"); exit(1); } }
                    else lv = __builtin_popcount(seen[row] & ((1u << w) - 1));
                    L.lrow_lv[q] = (uint16_t)(row | (lv << 13));
                    maxlv = std::max(maxlv, lv);
                }
            }
            L.round_phases.push_back((uint8_t)(maxlv + 1));
            ++rounds; phases += maxlv + 1;
        }
    }
    L.rb_run_ptr[P.n_rb] = (int32_t)L.run_start.size();
    L.desc.resize(L.run_start.size());
    for (size_t r = 0; r < L.run_start.size(); ++r)
        L.desc[r] = make_uint2(L.run_start[r], (unsigned)L.run_len[r] | ((unsigned)L.round_phases[r / RPR] << 8));
    L.mean_phases = (double)phases / std::max<int64_t>(rounds, 1);
}

// ------------------------------------------------------------------ kernels
template <int THREADS, int U, bool NT = false>
__global__ __launch_bounds__(THREADS) void k_phase1(int64_t n_cols, int split, const int32_t *__restrict__ ss_begin,
                                                   const int32_t *__restrict__ ss_end, const double *__restrict__ val,
                                                   const uint16_t *__restrict__ lcol, const c128 *__restrict__ x,
                                                   c128 *__restrict__ prod) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    c128 *xs = reinterpret_cast<c128 *>(smem);
    const int s = blockIdx.x / split, part = blockIdx.x % split;
    const int64_t c0 = (int64_t)s << CW_BITS;
    const int cw = (int)min((int64_t)CW, n_cols - c0);
    for (int i = threadIdx.x; i < cw; i += THREADS) xs[i] = x[c0 + i];
    __syncthreads();
    int k0 = ss_begin[s], k1 = (ss_end[s] + 7) & ~7;      // pad slots carry val = 0
    if (split > 1) {
        const int chunk = (((k1 - k0 + split - 1) / split) + 7) & ~7;
        k0 = k0 + part * chunk;
        k1 = min(k1, k0 + chunk);
    }
    for (int base = k0; base < k1; base += THREADS * U) {
        double a[U];
        int c[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = min(base + u * THREADS + (int)threadIdx.x, k1 - 1);
            a[u] = val[k];
            c[u] = lcol[k];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = base + u * THREADS + (int)threadIdx.x;
            if (k < k1) {
                const c128 xv = xs[c[u]];
                const c128 pv = make_double2(a[u] * xv.x, a[u] * xv.y);
                if (NT) nt_store(pv, &prod[k]); else prod[k] = pv;
            }
        }
    }
}

__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
    return v;
}

template <int W, int U, bool XCDMAP>
__global__ __launch_bounds__(W * 64) void k_phase2(int64_t n_rows, int n_rb, int rb_per_xcd,
                                                  const int32_t *__restrict__ rb_seg_ptr,
                                                  const uint8_t *__restrict__ segs, const uint16_t *__restrict__ lrow,
                                                  const c128 *__restrict__ prod, c128 *__restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int SB = RB / W, RS = W <= 8 ? 16 : 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int rb = blockIdx.x;
    if (XCDMAP) {
        if ((int)(blockIdx.x >> 3) >= rb_per_xcd) return;
        rb = (blockIdx.x & 7) * rb_per_xcd + (blockIdx.x >> 3);
    }
    if (rb >= n_rb) return;
    double *are = reinterpret_cast<double *>(smem) + (size_t)wave * 2 * SB, *aim = are + SB;
#pragma unroll
    for (int q = 0; q < SB / 64; ++q) { are[q * 64 + lane] = 0.0; aim[q * 64 + lane] = 0.0; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int g0 = rb_seg_ptr[rb], g1 = rb_seg_ptr[rb + 1];
    for (int gb = g0; gb < g1; gb += 64) {
        const int g = gb + lane;
        int b0 = 0, len = 0;
        if (g < g1) {
            const uint8_t *rec = segs + (size_t)g * RS;
            const uint32_t q0 = *reinterpret_cast<const uint32_t *>(rec);
            const int a = rec[4 + wave], b = rec[5 + wave];
            b0 = (int)q0 + a;
            len = b - a;
        }
        const int maxlen = wave_max_i(len);
        for (int i0 = 0; i0 < maxlen; i0 += U) {
            c128 p[U];
            int r[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (i0 + u < len) {
                    p[u] = prod[b0 + i0 + u];
                    r[u] = lrow[b0 + i0 + u] & (SB - 1);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (i0 + u < len) {
                    unsafeAtomicAdd(&are[r[u]], p[u].x);
                    unsafeAtomicAdd(&aim[r[u]], p[u].y);
                }
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int64_t row0 = ((int64_t)rb << RB_BITS) + (int64_t)wave * SB;
#pragma unroll 4
    for (int q = 0; q < SB / 64; ++q) {
        const int i = q * 64 + lane;
        if (row0 + i < n_rows) y[row0 + i] = make_double2(are[i], aim[i]);
    }
}


// Phase 2, grouped form: GS consecutive lanes share one segment and read its piece for this wave
// contiguously (a piece holds 8192/W * 33.5/8192 entries on average), U segments per group in flight.
template <int W, int GS, int U, bool XCDMAP, bool NT, int MODE = 0>
__global__ __launch_bounds__(W * 64) void k_phase2g(int64_t n_rows, int n_rb, int rb_per_xcd,
                                                   const int32_t *__restrict__ rb_seg_ptr,
                                                   const uint8_t *__restrict__ segs, const uint16_t *__restrict__ lrow,
                                                   const c128 *__restrict__ prod, c128 *__restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int SB = RB / W, RS = W <= 8 ? 16 : 32, NG = 64 / GS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane / GS, gl = lane % GS;
    int rb = blockIdx.x;
    if (XCDMAP) {
        if ((int)(blockIdx.x >> 3) >= rb_per_xcd) return;
        rb = (blockIdx.x & 7) * rb_per_xcd + (blockIdx.x >> 3);
    }
    if (rb >= n_rb) return;
    double *are = reinterpret_cast<double *>(smem) + (size_t)wave * 2 * SB, *aim = are + SB;
#pragma unroll
    for (int q = 0; q < SB / 64; ++q) { are[q * 64 + lane] = 0.0; aim[q * 64 + lane] = 0.0; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int g0 = rb_seg_ptr[rb], g1 = rb_seg_ptr[rb + 1];
    double dbg = 0.0;
    for (int gb = g0; gb < g1; gb += NG * U) {
        int b0[U], len[U];
        int longest = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int g = gb + u * NG + grp;
            b0[u] = 0; len[u] = 0;
            if (g < g1) {
                const uint8_t *rec = segs + (size_t)g * RS;
                const uint32_t q0 = *reinterpret_cast<const uint32_t *>(rec);
                const int a = rec[4 + wave], b = rec[5 + wave];
                b0[u] = (int)q0 + a;
                len[u] = b - a;
            }
            longest = max(longest, len[u]);
        }
        c128 p[U];
        int r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (gl < len[u]) {
                if (MODE == 2) { p[u] = make_double2(1.0, 2.0); r[u] = (b0[u] + gl) & (SB - 1); }
                else {
                    p[u] = NT ? nt_load(&prod[b0[u] + gl]) : prod[b0[u] + gl];
                    r[u] = lrow[b0[u] + gl] & (SB - 1);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (gl < len[u]) {
                if (MODE == 1) { dbg += p[u].x + p[u].y * r[u]; }
                else {
                    unsafeAtomicAdd(&are[r[u]], p[u].x);
                    unsafeAtomicAdd(&aim[r[u]], p[u].y);
                }
            }
        }
        if (__any(longest > GS)) {               // rare: pieces longer than the lane group
#pragma unroll 1
            for (int u = 0; u < U; ++u) {
                for (int i = GS + gl; __any(i < len[u]); i += GS) {
                    if (i < len[u]) {
                        const c128 pp = prod[b0[u] + i];
                        const int rr = lrow[b0[u] + i] & (SB - 1);
                        unsafeAtomicAdd(&are[rr], pp.x);
                        unsafeAtomicAdd(&aim[rr], pp.y);
                    }
                }
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int64_t row0 = ((int64_t)rb << RB_BITS) + (int64_t)wave * SB;
#pragma unroll 4
    for (int q = 0; q < SB / 64; ++q) {
        const int i = q * 64 + lane;
        if (row0 + i < n_rows) y[row0 + i] = make_double2(are[i] + (MODE == 1 ? dbg : 0.0), aim[i]);
    }
}


template <int W, int K, int D>
__global__ __launch_bounds__(W * 64) void k_phase2L(int64_t n_rows, int n_rb, int rb_per_xcd,
                                                   const int32_t *__restrict__ rb_run_ptr,
                                                   const uint2 *__restrict__ runs,      // x = first entry, y = len | phases << 8
                                                   const uint16_t *__restrict__ lrow_lv,
                                                   const c128 *__restrict__ prod, c128 *__restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int RPR = W * K, T = W * 64;
    double *are = reinterpret_cast<double *>(smem), *aim = are + RB;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if ((int)(blockIdx.x >> 3) >= rb_per_xcd) return;
    const int rb = (blockIdx.x & 7) * rb_per_xcd + (blockIdx.x >> 3);
    if (rb >= n_rb) return;
    for (int i = threadIdx.x; i < RB; i += T) { are[i] = 0.0; aim[i] = 0.0; }
    const int R0 = rb_run_ptr[rb];
    const int n_rounds = (rb_run_ptr[rb + 1] - R0) / RPR;
    const uint2 *my_runs = runs + R0 + wave * K;
    // three-stage software pipeline, D rounds per stage: descriptors (2D ahead) -> products (D ahead) -> adds
    c128 p[D][K];
    unsigned m[D][K];
    uint2 ds[D][K];       // descriptors of the round whose products are issued next from this slot
    unsigned info[D][K];  // len | phases << 8 of the round whose products sit in p[d]
    // every load below is unconditional (inactive lanes and rounds past the end re-read a valid
    // address): the waitcnt pass then counts loads exactly instead of falling back to vmcnt(0)
    auto load_desc = [&](int round, int d) {
        const int rc = min(round, n_rounds - 1);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            uint2 v = my_runs[(size_t)rc * RPR + k];
            if (round >= n_rounds) v = make_uint2(0u, 0u);
            ds[d][k] = v;
        }
    };
    auto issue = [&](int d) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const unsigned start = ds[d][k].x, len = ds[d][k].y & 255u;
            info[d][k] = ds[d][k].y;
            const unsigned idx = start + min((unsigned)lane, max(len, 1u) - 1u);
            p[d][k] = prod[idx];
            m[d][k] = lrow_lv[idx];
        }
    };
    if (n_rounds > 0) {
#pragma unroll
        for (int d = 0; d < D; ++d) load_desc(d, d);
#pragma unroll
        for (int d = 0; d < D; ++d) { issue(d); load_desc(D + d, d); }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (int r = 0; r < n_rounds; r += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int nph = (int)(info[d][0] >> 8);          // 0 for rounds past the end
            for (int ph = 0; ph < nph; ++ph) {
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    if (lane < (int)(info[d][k] & 255u) && (int)(m[d][k] >> 13) == ph) {
                        const int row = m[d][k] & (RB - 1);
                        unsafeAtomicAdd(&are[row], p[d][k].x);
                        unsafeAtomicAdd(&aim[row], p[d][k].y);
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
            issue(d);                                        // products of round r + d + D
            load_desc(r + d + 2 * D, d);
        }
    }
    const int64_t row0 = (int64_t)rb << RB_BITS;
    for (int i = threadIdx.x; i < RB; i += T)
        if (row0 + i < n_rows) y[row0 + i] = make_double2(are[i], aim[i]);
}

template <int W, int K, int D>
__global__ __launch_bounds__(W * 64) void k_phase2R(int64_t n_rows, int n_rb, int rb_per_xcd,
                                                   const int32_t *__restrict__ rb_run_ptr,
                                                   const uint2 *__restrict__ runs,      // x = first entry, y = len | phases << 8
                                                   const uint16_t *__restrict__ lrow_lv,
                                                   const c128 *__restrict__ prod, c128 *__restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int RPR = W * K, T = W * 64;
    c128 *acc = reinterpret_cast<c128 *>(smem);     // interleaved (re, im): one 16-byte read-modify-write per entry
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if ((int)(blockIdx.x >> 3) >= rb_per_xcd) return;
    const int rb = (blockIdx.x & 7) * rb_per_xcd + (blockIdx.x >> 3);
    if (rb >= n_rb) return;
    for (int i = threadIdx.x; i < RB; i += T) acc[i] = make_double2(0.0, 0.0);
    const int R0 = rb_run_ptr[rb];
    const int n_rounds = (rb_run_ptr[rb + 1] - R0) / RPR;
    const uint2 *my_runs = runs + R0 + wave * K;
    // three-stage software pipeline, D rounds per stage: descriptors (2D ahead) -> products (D ahead) -> adds
    c128 p[D][K];
    unsigned m[D][K];
    uint2 ds[D][K];       // descriptors of the round whose products are issued next from this slot
    unsigned info[D][K];  // len | phases << 8 of the round whose products sit in p[d]
    // every load below is unconditional (inactive lanes and rounds past the end re-read a valid
    // address): the waitcnt pass then counts loads exactly instead of falling back to vmcnt(0)
    auto load_desc = [&](int round, int d) {
        const int rc = min(round, n_rounds - 1);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            uint2 v = my_runs[(size_t)rc * RPR + k];
            if (round >= n_rounds) v = make_uint2(0u, 0u);
            ds[d][k] = v;
        }
    };
    auto issue = [&](int d) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const unsigned start = ds[d][k].x, len = ds[d][k].y & 255u;
            info[d][k] = ds[d][k].y;
            const unsigned idx = start + min((unsigned)lane, max(len, 1u) - 1u);
            p[d][k] = prod[idx];
            m[d][k] = lrow_lv[idx];
        }
    };
    if (n_rounds > 0) {
#pragma unroll
        for (int d = 0; d < D; ++d) load_desc(d, d);
#pragma unroll
        for (int d = 0; d < D; ++d) { issue(d); load_desc(D + d, d); }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (int r = 0; r < n_rounds; r += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int nph = (int)(info[d][0] >> 8);          // 0 for rounds past the end
            for (int ph = 0; ph < nph; ++ph) {
                // within one phase every row is touched by at most one lane of the whole workgroup
                c128 cur[K];
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const bool on = lane < (int)(info[d][k] & 255u) && (int)(m[d][k] >> 13) == ph;
                    if (on) cur[k] = acc[m[d][k] & (RB - 1)];
                }
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const bool on = lane < (int)(info[d][k] & 255u) && (int)(m[d][k] >> 13) == ph;
                    if (on) acc[m[d][k] & (RB - 1)] = make_double2(cur[k].x + p[d][k].x, cur[k].y + p[d][k].y);
                }
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
            issue(d);                                        // products of round r + d + D
            load_desc(r + d + 2 * D, d);
        }
    }
    const int64_t row0 = (int64_t)rb << RB_BITS;
    for (int i = threadIdx.x; i < RB; i += T)
        if (row0 + i < n_rows) y[row0 + i] = acc[i];
}


// Same schedule with the run descriptors in VECTOR registers: a block of 64/K rounds' descriptors is
// fetched by one vector load a block ahead (lane = round-in-block * K + k) and broadcast with
// v_readlane at issue time.  No scalar load is outstanding at a barrier (s_waitcnt lgkmcnt(0) would
// wait for it: SMEM and LDS share that counter).
template <int W, int K, int D, bool RMW>
__global__ __launch_bounds__(W * 64) void k_phase2V(int64_t n_rows, int n_rb, int rb_per_xcd,
                                                   const int32_t *__restrict__ rb_run_ptr,
                                                   const uint2 *__restrict__ runs,
                                                   const uint16_t *__restrict__ lrow_lv,
                                                   const c128 *__restrict__ prod, c128 *__restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int RPR = W * K, T = W * 64, B = 64 / K;
    static_assert(B % D == 0, "block of rounds must be a multiple of the pipeline depth");
    double *are = reinterpret_cast<double *>(smem), *aim = are + RB;
    c128 *acc = reinterpret_cast<c128 *>(smem);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if ((int)(blockIdx.x >> 3) >= rb_per_xcd) return;
    const int rb = (blockIdx.x & 7) * rb_per_xcd + (blockIdx.x >> 3);
    if (rb >= n_rb) return;
    for (int i = threadIdx.x; i < RB; i += T) { are[i] = 0.0; aim[i] = 0.0; }
    const int R0 = rb_run_ptr[rb];
    const int n_rounds = (rb_run_ptr[rb + 1] - R0) / RPR;
    const uint2 *my_runs = runs + R0 + wave * K;
    auto load_block = [&](int first_round) {     // descriptors of rounds first_round .. first_round + B - 1
        const int round = first_round + lane / K;
        uint2 v = my_runs[(size_t)min(round, max(n_rounds - 1, 0)) * RPR + lane % K];
        if (round >= n_rounds) v = make_uint2(0u, 0u);
        return v;
    };
    c128 p[D][K];
    unsigned m[D][K], info[D][K];
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int k = 0; k < K; ++k) { info[d][k] = 0u; m[d][k] = 0u; p[d][k] = make_double2(0.0, 0.0); }
    uint2 dv, dvn = load_block(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (int i0 = 0; i0 < n_rounds + D; i0 += B) {
        dv = dvn;
        dvn = load_block(i0 + B);
#pragma unroll
        for (int j = 0; j < B; ++j) {
            constexpr int dummy = 0; (void)dummy;
            const int d = j % D;
            // ---- consume round i0 + j - D (held in stage d)
            const int nph = (int)(info[d][0] >> 8);
            for (int ph = 0; ph < nph; ++ph) {
                if (RMW) {
                    c128 cur[K];
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const bool on = lane < (int)(info[d][k] & 255u) && (int)(m[d][k] >> 13) == ph;
                        if (on) cur[k] = acc[m[d][k] & (RB - 1)];
                    }
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const bool on = lane < (int)(info[d][k] & 255u) && (int)(m[d][k] >> 13) == ph;
                        if (on) acc[m[d][k] & (RB - 1)] = make_double2(cur[k].x + p[d][k].x, cur[k].y + p[d][k].y);
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        if (lane < (int)(info[d][k] & 255u) && (int)(m[d][k] >> 13) == ph) {
                            const int row = m[d][k] & (RB - 1);
                            unsafeAtomicAdd(&are[row], p[d][k].x);
                            unsafeAtomicAdd(&aim[row], p[d][k].y);
                        }
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
            // ---- issue round i0 + j into stage d
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const unsigned start = __builtin_amdgcn_readlane(dv.x, j * K + k);
                const unsigned yv = __builtin_amdgcn_readlane(dv.y, j * K + k);
                info[d][k] = yv;
                const unsigned idx = start + min((unsigned)lane, max(yv & 255u, 1u) - 1u);
                p[d][k] = prod[idx];
                m[d][k] = lrow_lv[idx];
            }
        }
    }
    const int64_t row0 = (int64_t)rb << RB_BITS;
    for (int i = threadIdx.x; i < RB; i += T) {
        if (row0 + i < n_rows) y[row0 + i] = RMW ? acc[i] : make_double2(are[i], aim[i]);
    }
}

// reference / "irreducible pattern": plain CSR, one lane per row, 16-byte gathers straight from x
__global__ __launch_bounds__(256) void k_csr_ref(int64_t n_rows, const int32_t *__restrict__ indptr,
                                                const int32_t *__restrict__ indices, const double *__restrict__ val,
                                                const c128 *__restrict__ x, c128 *__restrict__ y) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_rows) return;
    double sr = 0.0, si = 0.0;
    for (int k = indptr[r]; k < indptr[r + 1]; ++k) {
        const c128 xv = x[indices[k]];
        sr += val[k] * xv.x;
        si += val[k] * xv.y;
    }
    y[r] = make_double2(sr, si);
}

// streaming ceilings: the bytes of each phase with every access sequential and no LDS work
__global__ __launch_bounds__(256) void k_stream_p1(int64_t nnz, const double *__restrict__ val,
                                                  const uint16_t *__restrict__ lcol, c128 *__restrict__ prod) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < nnz; k += stride) {
        const double a = val[k];
        prod[k] = make_double2(a, a * (double)lcol[k]);
    }
}
__global__ __launch_bounds__(256) void k_stream_p2(int64_t nnz, const c128 *__restrict__ prod,
                                                  const uint16_t *__restrict__ lrow, double *__restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    double s = 0.0;
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < nnz; k += stride) {
        const c128 p = prod[k];
        s += p.x + p.y * (double)lrow[k];
    }
    if (s == 1.2345e300) out[blockIdx.x] = s;     // never true: keeps the loads
}
__global__ void k_permute(int64_t n, const uint32_t *perm, const c128 *in, c128 *out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = in[perm[i]];
}
__global__ void k_fill_x(int64_t n, c128 *x) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        x[i] = make_double2(std::sin(0.001 * (double)(i % 100003)) + 0.5, std::cos(0.003 * (double)(i % 70001)));
}

template <typename T> static T *upload(const std::vector<T> &v) {
    T *d;
    CK(hipMalloc(&d, std::max<size_t>(v.size(), 1) * sizeof(T)));
    CK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return d;
}

static hipEvent_t e0, e1;
template <typename F> static double time_ms(F launch, int reps = 20) {
    for (int i = 0; i < 3; ++i) launch();
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main(int argc, char **argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 10000000;
    const int per_row = argc > 2 ? atoi(argv[2]) : 5;
    const bool prof = argc > 3 && !strcmp(argv[3], "prof");
    const int64_t nnz = n * per_row;
    printf("random CSR n=%lld nnz=%lld\n", (long long)n, (long long)nnz);
    std::vector<int32_t> indptr(n + 1), indices(nnz);
    std::vector<double> values(nnz);
    uint64_t st = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    for (int64_t r = 0; r <= n; ++r) indptr[r] = (int32_t)(r * per_row);
    for (int64_t k = 0; k < nnz; ++k) {
        indices[k] = (int32_t)(rnd() % (uint64_t)n);
        values[k] = (double)(rnd() >> 11) * (2.0 / 9007199254740992.0) - 1.0;
    }
    for (int64_t r = 0; r < n; ++r) std::sort(indices.begin() + r * per_row, indices.begin() + (r + 1) * per_row);

    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    c128 *x, *y, *yref, *prod;
    CK(hipMalloc(&x, n * sizeof(c128)));
    CK(hipMalloc(&y, n * sizeof(c128)));
    CK(hipMalloc(&yref, n * sizeof(c128)));
    hipLaunchKernelGGL(k_fill_x, dim3(2048), dim3(256), 0, 0, n, x);
    int32_t *d_indptr = upload(indptr), *d_indices = upload(indices);
    double *d_values = upload(values);
    const double alg_bytes = 12.0 * nnz + 36.0 * n + 4;
    {
        const double ms = time_ms([&] { hipLaunchKernelGGL(k_csr_ref, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, n, d_indptr, d_indices, d_values, x, yref); });
        printf("%-58s %8.4f ms  %6.2f TB/s algorithmic\n", "direct gather: plain CSR, lane per row", ms, alg_bytes / ms / 1e9);
    }
    std::vector<c128> h_ref(n), h_y(n);
    CK(hipMemcpy(h_ref.data(), yref, n * sizeof(c128), hipMemcpyDeviceToHost));

    for (int W : {8, 16}) {
        if (prof && W != 8) break;
        Plan P;
        auto t0 = std::chrono::steady_clock::now();
        make_plan(n, n, indptr.data(), indices.data(), values.data(), W, P);
        const double plan_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("\nW=%d waves per row block: %d sub-slabs x %d row blocks, %lld segments (%.1f entries each), nnz_pad %lld, plan %.1f s\n",
               W, P.n_ss, P.n_rb, (long long)P.n_seg, (double)P.nnz / P.n_seg, (long long)P.nnz_pad, plan_s);
        double *d_val = upload(P.val);
        uint16_t *d_lcol = upload(P.lcol), *d_lrow = upload(P.lrow);
        int32_t *d_ssb = upload(P.ss_begin), *d_sse = upload(P.ss_end), *d_rbseg = upload(P.rb_seg_ptr);
        uint8_t *d_segs = upload(P.segs);
        CK(hipMalloc(&prod, P.nnz_pad * sizeof(c128)));
        const double p1_bytes = 26.0 * P.nnz_pad + 16.0 * n, p2_bytes = 18.0 * P.nnz + 16.0 * n + (double)P.segs.size();
        printf("  streamed bytes: phase 1 %.3f GB, phase 2 %.3f GB, sum %.3f GB = %.2f x algorithmic\n", p1_bytes / 1e9,
               p2_bytes / 1e9, (p1_bytes + p2_bytes) / 1e9, (p1_bytes + p2_bytes) / alg_bytes);
        if (W == 8 && !prof) {
            const double s1 = time_ms([&] { hipLaunchKernelGGL(k_stream_p1, dim3(4096), dim3(256), 0, 0, P.nnz_pad, d_val, d_lcol, prod); });
            const double s2 = time_ms([&] { hipLaunchKernelGGL(k_stream_p2, dim3(4096), dim3(256), 0, 0, P.nnz_pad, prod, d_lrow, reinterpret_cast<double *>(y)); });
            printf("  %-56s %8.4f ms  %6.2f TB/s\n", "streaming ceiling of phase 1 (26 B/nnz sequential)", s1, 26.0 * P.nnz_pad / s1 / 1e9);
            printf("  %-56s %8.4f ms  %6.2f TB/s\n", "streaming ceiling of phase 2 (18 B/nnz sequential)", s2, 18.0 * P.nnz_pad / s2 / 1e9);
        }
        const size_t lds1 = (size_t)CW * sizeof(c128), lds2 = (size_t)RB * sizeof(c128);
        auto set_lds = [&](const void *f, size_t b) { CK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b)); };
        set_lds(reinterpret_cast<const void *>(k_phase1<1024, 4>), lds1);
        set_lds(reinterpret_cast<const void *>(k_phase1<1024, 8>), lds1);
        set_lds(reinterpret_cast<const void *>(k_phase1<512, 8>), lds1);
        double best1 = 1e9, best2 = 1e9;
        if (W == 8 && !prof) {
            for (int split : {1}) {
                double ms = time_ms([&] { hipLaunchKernelGGL((k_phase1<1024, 4>), dim3(P.n_ss * split), dim3(1024), lds1, 0, n, split, d_ssb, d_sse, d_val, d_lcol, x, prod); });
                printf("  phase 1  1024 thr x4, split %d %33s %8.4f ms  %6.2f TB/s\n", split, "", ms, p1_bytes / ms / 1e9);
                best1 = std::min(best1, ms);
                ms = time_ms([&] { hipLaunchKernelGGL((k_phase1<1024, 8>), dim3(P.n_ss * split), dim3(1024), lds1, 0, n, split, d_ssb, d_sse, d_val, d_lcol, x, prod); });
                printf("  phase 1  1024 thr x8, split %d %33s %8.4f ms  %6.2f TB/s\n", split, "", ms, p1_bytes / ms / 1e9);
                best1 = std::min(best1, ms);
                ms = time_ms([&] { hipLaunchKernelGGL((k_phase1<512, 8>), dim3(P.n_ss * split), dim3(512), lds1, 0, n, split, d_ssb, d_sse, d_val, d_lcol, x, prod); });
                printf("  phase 1   512 thr x8, split %d %33s %8.4f ms  %6.2f TB/s\n", split, "", ms, p1_bytes / ms / 1e9);
                best1 = std::min(best1, ms);
            }
        }
        if (W == 8 && !prof) {
            set_lds(reinterpret_cast<const void *>(k_phase1<1024, 4, true>), lds1);
            const double ms = time_ms([&] { hipLaunchKernelGGL((k_phase1<1024, 4, true>), dim3(P.n_ss), dim3(1024), lds1, 0, n, 1, d_ssb, d_sse, d_val, d_lcol, x, prod); });
            printf("  phase 1  1024 thr x4, split 1, nt stores %21s %8.4f ms  %6.2f TB/s\n", "", ms, p1_bytes / ms / 1e9);
            best1 = std::min(best1, ms);
        }
        hipLaunchKernelGGL((k_phase1<1024, 4>), dim3(P.n_ss), dim3(1024), lds1, 0, n, 1, d_ssb, d_sse, d_val, d_lcol, x, prod);
        const int rbx = (P.n_rb + 7) / 8;
        auto run2 = [&](auto kern, const char *name, bool xcd) {
            set_lds(reinterpret_cast<const void *>(kern), lds2);
            CK(hipMemset(y, 0xff, n * sizeof(c128)));
            const double ms = time_ms([&] { hipLaunchKernelGGL(kern, dim3(xcd ? rbx * 8 : P.n_rb), dim3(W * 64), lds2, 0, n, P.n_rb, rbx, d_rbseg, d_segs, d_lrow, prod, y); });
            CK(hipMemcpy(h_y.data(), y, n * sizeof(c128), hipMemcpyDeviceToHost));
            double err = 0.0, ref = 0.0;
            for (int64_t i = 0; i < n; ++i) {
                err = std::max(err, std::max(std::fabs(h_y[i].x - h_ref[i].x), std::fabs(h_y[i].y - h_ref[i].y)));
                ref = std::max(ref, std::fabs(h_ref[i].x));
            }
            printf("  phase 2  %-46s %8.4f ms  %6.2f TB/s   max err %.2e (|y| max %.2f)\n", name, ms, p2_bytes / ms / 1e9, err, ref);
            best2 = std::min(best2, ms);
        };
        if (prof) {
        } else if (W == 8) {
            run2(k_phase2<8, 2, true>, "lane/segment, 8 waves, U=2, XCD-contiguous", true);
            run2(k_phase2g<8, 4, 8, true, false>, "group 4, 8 waves, U=8, XCD-contiguous", true);
        } else {
            run2(k_phase2g<16, 4, 4, true, false>, "group 4, 16 waves, U=4, XCD-contiguous", true);
        }

        if (W == 8) {
            auto runL = [&](auto kern, int Wl, int Kl, const char *name, bool strict = false, bool contig = false) {
                PlanL L;
                auto t1 = std::chrono::steady_clock::now();
                make_plan_L(P, Wl, Kl, L, strict);
                const double ps = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
                const c128 *prod_in = prod;
                c128 *prod2 = nullptr;
                if (contig) {
                    // HYPOTHETICAL layout for the experiment: the runs of a row block back to back (what phase 2 would
                    // read if phase 1 scattered its runs into row-block-major order): is phase 2 bound by the 528-byte
                    // granularity of its reads?
                    std::vector<uint32_t> perm(P.nnz_pad, 0);
                    std::vector<uint16_t> lv2(P.nnz_pad, 0);
                    uint32_t pos = 0;
                    for (size_t r = 0; r < L.desc.size(); ++r) {
                        const uint32_t st = L.desc[r].x, len = L.desc[r].y & 255u;
                        for (uint32_t i = 0; i < len; ++i) { perm[pos + i] = st + i; lv2[pos + i] = L.lrow_lv[st + i]; }
                        L.desc[r].x = pos;
                        pos += len;
                    }
                    L.lrow_lv = lv2;
                    uint32_t *d_perm = upload(perm);
                    CK(hipMalloc(&prod2, P.nnz_pad * sizeof(c128)));
                    hipLaunchKernelGGL(k_permute, dim3(4096), dim3(256), 0, 0, P.nnz_pad, d_perm, prod, prod2);
                    CK(hipDeviceSynchronize());
                    CK(hipFree(d_perm));
                    prod_in = prod2;
                }
                uint2 *d_desc = upload(L.desc);
                int32_t *d_rp = upload(L.rb_run_ptr);
                uint16_t *d_lv = upload(L.lrow_lv);
                set_lds(reinterpret_cast<const void *>(kern), lds2);
                CK(hipMemset(y, 0xff, n * sizeof(c128)));
                auto launch = [&] { hipLaunchKernelGGL(kern, dim3(rbx * 8), dim3(Wl * 64), lds2, 0, n, P.n_rb, rbx, d_rp, d_desc, d_lv, prod_in, y); };
                const double ms = time_ms(launch);
                CK(hipMemcpy(h_y.data(), y, n * sizeof(c128), hipMemcpyDeviceToHost));
                std::vector<c128> first(h_y);
                launch();
                CK(hipMemcpy(h_y.data(), y, n * sizeof(c128), hipMemcpyDeviceToHost));
                const bool same = memcmp(first.data(), h_y.data(), n * sizeof(c128)) == 0;
                double err = 0.0;
                for (int64_t i = 0; i < n; ++i) err = std::max(err, std::max(std::fabs(h_y[i].x - h_ref[i].x), std::fabs(h_y[i].y - h_ref[i].y)));
                const double bytes = 18.0 * P.nnz + 16.0 * n + 8.0 * L.run_start.size();
                printf("  phase 2L %-46s %8.4f ms  %6.2f TB/s   max err %.2e  bitwise repeat %s  (%.2f phases/round, plan %.1f s)\n", name, ms,
                       bytes / ms / 1e9, err, same ? "yes" : "NO", L.mean_phases, ps);
                best2 = std::min(best2, ms);
                CK(hipFree(d_desc)); CK(hipFree(d_rp)); CK(hipFree(d_lv));
                if (prod2) CK(hipFree(prod2));
            };
            runL(k_phase2V<8, 4, 2, false>, 8, 4, "atomics, vector desc, 8 waves x 4 runs, depth 2");
            if (prof) return 0;
            runL(k_phase2V<8, 4, 2, false>, 8, 4, "same kernel, runs of a row block CONTIGUOUS", false, true);
            runL(k_phase2V<8, 4, 4, false>, 8, 4, "depth 4, runs of a row block CONTIGUOUS", false, true);
            runL(k_phase2L<8, 4, 3>, 8, 4, "atomics, s_load desc, 8 waves x 4 runs, depth 3");
            runL(k_phase2V<8, 4, 4, false>, 8, 4, "atomics, vector desc, 8 waves x 4 runs, depth 4");
            runL(k_phase2V<8, 2, 4, false>, 8, 2, "atomics, vector desc, 8 waves x 2 runs, depth 4");
            runL(k_phase2V<8, 8, 2, false>, 8, 8, "atomics, vector desc, 8 waves x 8 runs, depth 2");
            runL(k_phase2V<8, 4, 4, true>, 8, 4, "RMW, vector desc, 8 waves x 4 runs, depth 4", true);
            runL(k_phase2V<16, 2, 4, true>, 16, 2, "RMW, vector desc, 16 waves x 2 runs, depth 4", true);
            runL(k_phase2V<16, 4, 2, true>, 16, 4, "RMW, vector desc, 16 waves x 4 runs, depth 2", true);
            runL(k_phase2V<16, 4, 4, true>, 16, 4, "RMW, vector desc, 16 waves x 4 runs, depth 4", true);
        }
        if (W == 8) {
            const double both = time_ms([&] {
                hipLaunchKernelGGL((k_phase1<1024, 4>), dim3(P.n_ss), dim3(1024), lds1, 0, n, 1, d_ssb, d_sse, d_val, d_lcol, x, prod);
                hipLaunchKernelGGL((k_phase2g<8, 8, 8, true, false>), dim3(rbx * 8), dim3(512), lds2, 0, n, P.n_rb, rbx, d_rbseg, d_segs, d_lrow, prod, y);
            });
            printf("  %-56s %8.4f ms  %6.2f TB/s algorithmic = %.3f of 8 TB/s\n", "both phases back to back (1024x4 split 1; group 8 U=8)", both,
                   alg_bytes / both / 1e9, alg_bytes / both / 1e9 / 8.0);
            printf("  best phase 1 + best phase 2 = %.4f ms -> %.3f of 8 TB/s\n", best1 + best2, alg_bytes / (best1 + best2) / 1e9 / 8.0);
        }
        CK(hipFree(d_val)); CK(hipFree(d_lcol)); CK(hipFree(d_lrow)); CK(hipFree(d_ssb)); CK(hipFree(d_sse));
        CK(hipFree(d_rbseg)); CK(hipFree(d_segs)); CK(hipFree(prod));
    }
    return 0;
}
