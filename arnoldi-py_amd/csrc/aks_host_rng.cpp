// Host helper of the drop-in (no device code): NumPy's LEGACY standard-normal stream, bit for bit, several times faster.
//
// The reference draws its start vector with np.random.randn(n) on the global legacy generator (src/arnoldi/utils.py:7-13);
// the drop-in must produce the same bits for the same seed.  At n = 10M that single-threaded call is
// 0.17-0.28 s -- longer than the device-side set-up and three times the solve.  What NumPy computes
// (numpy/random/src/legacy/legacy-distributions.c: legacy_gauss; numpy/random/src/mt19937/mt19937.{h,c}):
//
//     double():  a = next32() >> 5, b = next32() >> 6;  (a * 67108864.0 + b) / 9007199254740992.0
//     gauss():   if a value is cached: return it and clear the cache; otherwise
//                do { x1 = 2 double() - 1; x2 = 2 double() - 1; r2 = x1 x1 + x2 x2; } while (r2 >= 1 || r2 == 0);
//                f = sqrt(-2 log(r2) / r2);  cache f x1;  return f x2
//
// Only the Mersenne Twister is inherently sequential (about 1.5 ns per 32-bit word).  Here the raw words of a block of
// polar-method iterations are generated first, then the iterations are evaluated on host threads (same expressions, the
// same libm log / sqrt, no contraction), their acceptances counted, and the accepted pairs written in stream order.  A block
// never holds more iterations than pairs are still needed, so the stream is never read past the point where NumPy stops;
// the caller passes NumPy's state in (np.random.get_state()) and writes the returned one back (set_state), so every later
// draw of the process continues exactly as if np.random.randn(n) had been called.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <thread>
#include <vector>

#include "arnoldi_hostrng.h"

#ifdef __clang__
#pragma STDC FP_CONTRACT OFF      // (g++: -ffp-contract=off on the command line, arnoldi-py_amd/Makefile)
#endif

namespace {

constexpr int MT_N = 624, MT_M = 397;
constexpr uint32_t MATRIX_A = 0x9908b0dfu, UPPER = 0x80000000u, LOWER = 0x7fffffffu;

struct Mt {
    uint32_t *key;
    int pos;
    void regenerate() {          // mt19937_gen
        int kk = 0;
        uint32_t y;
        for (; kk < MT_N - MT_M; ++kk) {
            y = (key[kk] & UPPER) | (key[kk + 1] & LOWER);
            key[kk] = key[kk + MT_M] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
        }
        for (; kk < MT_N - 1; ++kk) {
            y = (key[kk] & UPPER) | (key[kk + 1] & LOWER);
            key[kk] = key[kk + (MT_M - MT_N)] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
        }
        y = (key[MT_N - 1] & UPPER) | (key[0] & LOWER);
        key[MT_N - 1] = key[MT_M - 1] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
        pos = 0;
    }
    static inline uint32_t temper(uint32_t y) {
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }
    inline uint32_t next() {
        if (pos == MT_N) regenerate();
        return temper(key[pos++]);
    }
    void fill(uint32_t *out, int64_t count) {      // the next `count` outputs, in order
        int64_t done = 0;
        while (done < count) {
            if (pos == MT_N) regenerate();
            const int64_t take = std::min<int64_t>(MT_N - pos, count - done);
            for (int64_t i = 0; i < take; ++i) out[done + i] = temper(key[pos + i]);
            pos += (int)take;
            done += take;
        }
    }
};

inline double to_double(uint32_t wa, uint32_t wb) {
    const int32_t a = (int32_t)(wa >> 5), b = (int32_t)(wb >> 6);
    return (a * 67108864.0 + b) / 9007199254740992.0;
}

// one polar-method iteration from four raw words; true if accepted (then g2 = f x2 comes first in the stream, g1 = f x1 second)
inline bool polar(const uint32_t *w, double &g2, double &g1) {
    const double x1 = 2.0 * to_double(w[0], w[1]) - 1.0;
    const double x2 = 2.0 * to_double(w[2], w[3]) - 1.0;
    const double r2 = x1 * x1 + x2 * x2;
    if (r2 >= 1.0 || r2 == 0.0) return false;
    const double f = std::sqrt(-2.0 * std::log(r2) / r2);
    g1 = f * x1;
    g2 = f * x2;
    return true;
}

int host_threads() {
    int nt = 0;
    if (const char *e = getenv("AKS_PLAN_THREADS")) nt = atoi(e);
    if (nt <= 0) nt = std::min<int>((int)std::thread::hardware_concurrency(), 16);
    return std::max(1, std::min(nt, 64));
}

template <typename F>
void run_threads(int nt, F f) {
    if (nt <= 1) { f(0); return; }
    std::vector<std::thread> pool;
    for (int t = 0; t < nt; ++t) pool.emplace_back([&, t] { f(t); });
    for (auto &th : pool) th.join();
}

}  // namespace

extern "C" int aks_legacy_randn(uint32_t *key, int32_t *pos, int32_t *has_gauss, double *gauss, double *out, int64_t n) try {
    if (!key || !pos || !has_gauss || !gauss || (n > 0 && !out) || n < 0 || *pos < 0 || *pos > MT_N) return 1;
    Mt mt{key, *pos};
    int64_t i = 0;
    if (*has_gauss && n > 0) {           // a value cached by an earlier odd draw comes first
        out[i++] = *gauss;
        *has_gauss = 0;
        *gauss = 0.0;
    }
    // whole pairs still to write; an odd remainder takes the first value of one more pair and caches the second
    int64_t pairs = (n - i) / 2;
    const bool odd = ((n - i) & 1) != 0;
    const int nt_max = host_threads();
    std::vector<uint32_t> raw;
    std::vector<int64_t> counts;
    while (pairs >= 4096) {
        const int64_t block = std::min<int64_t>(pairs, (int64_t)1 << 23);      // iterations: never more than pairs needed
        raw.resize((size_t)block * 4);
        mt.fill(raw.data(), block * 4);
        const int nt = (int)std::min<int64_t>(nt_max, block / 2048);
        counts.assign(nt + 1, 0);
        run_threads(nt, [&](int t) {           // pass 1: how many iterations of each range are accepted
            const int64_t b0 = block * t / nt, b1 = block * (t + 1) / nt;
            int64_t c = 0;
            for (int64_t k = b0; k < b1; ++k) {
                const uint32_t *w = &raw[(size_t)k * 4];
                const double x1 = 2.0 * to_double(w[0], w[1]) - 1.0, x2 = 2.0 * to_double(w[2], w[3]) - 1.0;
                const double r2 = x1 * x1 + x2 * x2;
                c += !(r2 >= 1.0 || r2 == 0.0);
            }
            counts[t + 1] = c;
        });
        for (int t = 0; t < nt; ++t) counts[t + 1] += counts[t];
        run_threads(nt, [&](int t) {           // pass 2: the accepted pairs, written where the stream puts them
            const int64_t b0 = block * t / nt, b1 = block * (t + 1) / nt;
            double *dst = out + i + 2 * counts[t];
            for (int64_t k = b0; k < b1; ++k) {
                double g2, g1;
                if (polar(&raw[(size_t)k * 4], g2, g1)) { dst[0] = g2; dst[1] = g1; dst += 2; }
            }
        });
        i += 2 * counts[nt];
        pairs -= counts[nt];
    }
    while (pairs > 0) {                     // the tail, one iteration at a time as NumPy does
        uint32_t w[4] = {0, 0, 0, 0};
        w[0] = mt.next(); w[1] = mt.next(); w[2] = mt.next(); w[3] = mt.next();
        double g2, g1;
        if (polar(w, g2, g1)) { out[i++] = g2; out[i++] = g1; --pairs; }
    }
    if (odd) {
        for (;;) {
            uint32_t w[4];
            w[0] = mt.next(); w[1] = mt.next(); w[2] = mt.next(); w[3] = mt.next();
            double g2, g1;
            if (polar(w, g2, g1)) { out[i++] = g2; *gauss = g1; *has_gauss = 1; break; }
        }
    }
    *pos = mt.pos;
    return i == n ? 0 : 2;
} catch (...) {
    return 3;
}
