// fk_exact_scan.hpp -- a PARALLEL prefix sum that reproduces numpy.cumsum's strictly
// sequential fp64 add chain bit-for-bit (needed because filterpy's resamplers compare
// positions against np.cumsum(weights): resampling.py:106,142,174 -- any re-associated
// scan flips indices at N ~ 1e7).
//
// Idea.  While the running sum c stays inside one binade, every representable value is an
// integer multiple of u = ulp(c):  c = C*u with 2^52 <= C < 2^53 (or C < 2^52 for
// subnormal c).  Adding a weight w = q*u + r (q = floor(w/u), 0 <= r < u, both exact because u is
// a power of two) gives the exact sum (C+q)*u + r, and IEEE round-to-nearest-even at
// granularity u is the INTEGER map
//        C -> C + q + [r > u/2]                      (no tie)
//        C -> C + q + ((C + q) & 1)                  (tie r == u/2: round half to even)
// which depends on C only through its parity.  Such maps form a monoid
// (ae, ao) = (increment when C is even, increment when C is odd), closed under composition,
// so an ordinary associative scan over int64 pairs yields every prefix exactly.  The
// segment ends at the first element whose result leaves the binade (C >= 2^53) -- that one
// element is added with a real fp64 add and the scan restarts in the new binade (~ once per
// doubling of the running sum: a few dozen times per weight vector).
//
// __host__ __device__: the same code is exercised on the host by tests/hostcheck.
#pragma once

#include <stdint.h>
#include <string.h>

#include "fk_math.hpp"

namespace fk {

struct Mono {
    long long ae, ao;   // increment applied when C is even / odd
};

constexpr long long MONO_LIMIT = 1LL << 53;   // C >= 2^53: left the binade
constexpr long long MONO_BIG = 1LL << 54;     // element that always ends the segment
constexpr long long MONO_SAT = 1LL << 60;

FK_HD long long mono_sat(long long v) { return v > MONO_SAT ? MONO_SAT : v; }

FK_HD Mono mono_identity() { return Mono{0, 0}; }

// apply f first, then g
FK_HD Mono mono_compose(const Mono &f, const Mono &g)
{
    Mono h;
    h.ae = mono_sat(f.ae + ((f.ae & 1) ? g.ao : g.ae));
    h.ao = mono_sat(f.ao + (((f.ao + 1) & 1) ? g.ao : g.ae));
    return h;
}

FK_HD double bits_to_double(uint64_t b)
{
    double d;
    memcpy(&d, &b, sizeof(d));
    return d;
}
FK_HD uint64_t double_to_bits(double d)
{
    uint64_t b;
    memcpy(&b, &d, sizeof(b));
    return b;
}

// ulp of a finite c > 0 (spacing of doubles in c's binade; 2^-1074 for subnormals)
FK_HD double ulp_of(double c)
{
    const uint64_t e = (double_to_bits(c) >> 52) & 0x7ffu;
    if (e <= 53) return bits_to_double(e <= 1 ? 1ull : (1ull << (e - 1)));
    return bits_to_double((e - 52) << 52);
}

// log2 of ulp_of(c): the ulp is 2^ulp_exp(c), -1074 <= ulp_exp <= 971
FK_HD int ulp_exp(double c)
{
    const int e = (int)((double_to_bits(c) >> 52) & 0x7ffu);
    return (e <= 1 ? 1 : e) - 1075;
}

// x * 2^k, exact when the result is representable (one v_ldexp_f64 on the GPU)
FK_HD double scale2(double x, int k)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_ldexp(x, k);
#else
    return ldexp(x, k);
#endif
}

// The map "add w" while the running sum has ulp u = 2^eu.
FK_HD Mono mono_elem(double w, double u, int eu)
{
    // negative / NaN / huge weights end the segment and are added with a real fp add
    if (!(w >= 0.0) || !(w < 0x1p1000)) return Mono{MONO_BIG, MONO_BIG};
    long long q = 0;
    double r = w;
    if (w >= u) {
        const double t = scale2(w, -eu);   // w / u, exact: u is a power of two and t >= 1
        if (!(t < 0x1p53)) return Mono{MONO_BIG, MONO_BIG};
        q = (long long)t;                  // floor (t >= 0)
        r = w - scale2((double)q, eu);     // exact
    }
    const double half = u * 0.5;           // 0 when u is the smallest subnormal (then r == 0 always)
    if (r == half && r != 0.0) return Mono{q + (q & 1), q + ((q + 1) & 1)};
    const long long inc = q + (r > half ? 1 : 0);
    return Mono{inc, inc};
}

// The same map in the common case, as ONE number: when the remainder is not an exact half ulp the
// map is "C -> C + inc" whatever C's parity, and inc (an integer < 2^53) is held exactly by a double.
// Composition is then a plain fp64 add -- exact while the sum stays below 2^53, and a sum that
// reaches 2^53 stays >= 2^53 after rounding, which is all the crossing test needs.  A tile without
// any tie (ties need the bits of w below u to be exactly 100...0: ~2^-23 per element for random
// weights) is scanned with doubles; a tile with one falls back to the Mono scan.
//   returns inc (2^54 for an element that must end the segment); tie = exact half-ulp remainder
FK_HD double fast_inc(double w, int eu, bool &tie)
{
    tie = false;
    if (!(w >= 0.0) || !(w < 0x1p1000)) return 0x1p54;
    const double t = scale2(w, -eu);       // w / u; an underflow only happens far below 1/2
    if (!(t < 0x1p53)) return 0x1p54;
    const double q = floor(t);
    const double r = t - q;                // exact, in [0, 1)
    tie = (r == 0.5);
    return q + (r > 0.5 ? 1.0 : 0.0);
}

// idx = #{ j < len : cs[j] <= p }  (upper bound of p in the tile's cumulative sums; the two-pointer merge
// of resampling.py:103-112 / 143-149 visits exactly this index).  The output positions are evenly spaced
// and the sums rise from c_lo (carry into the tile) to c_lo + len / inv_span, so an interpolated guess
// usually lands within a few elements: two probes confirm a 16-element window, five select-only
// power-of-two steps (8, 4, 2, 1, 1) count the elements <= p inside it.  A window the probes reject (skewed
// weights) falls back to the binary search of the side they left.  Every route returns the same index
// for non-decreasing cs; inv_span may be NaN / Inf for a degenerate tile (the guess is clamped).
//   The tile is GUARDED so that no read needs a bounds test: cs[-1] = -inf and cs[len .. len+TILE_GUARD) = +inf.
constexpr int TILE_GUARD = 16;
FK_HD int tile_upper_bound(const double *cs, int len, double p, double c_lo, double inv_span)
{
    const double gf = (p - c_lo) * inv_span;
    const int g = gf > 0.0 ? (gf < (double)len ? (int)gf : len - 1) : 0;
    const int a = g > 8 ? g - 8 : 0;
    const double *w = cs + a;
    const bool ok_lo = w[-1] <= p, ok_hi = !(w[16] <= p);
    if (!(ok_lo & ok_hi)) {
        int lo = 0, hi = len;
        if (!ok_lo) hi = a - 1;                          // cs[a-1] > p  (a >= 1: the low guard never rejects)
        else lo = a + 17 < len ? a + 17 : len;           // cs[a+16] <= p
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cs[mid] <= p) lo = mid + 1;
            else hi = mid;
        }
        return lo;
    }
    // count of elements <= p in [a, a+16): everything from a+16 on is > p, so no step can overshoot
    int r = 0;
    r += (w[r + 7] <= p) ? 8 : 0;
    r += (w[r + 3] <= p) ? 4 : 0;
    r += (w[r + 1] <= p) ? 2 : 0;
    r += (w[r] <= p) ? 1 : 0;
    r += (w[r] <= p) ? 1 : 0;
    return a + r;
}

// C after applying composite F to start value C0
FK_HD long long mono_apply(long long C0, const Mono &F) { return C0 + ((C0 & 1) ? F.ao : F.ae); }

}  // namespace fk
