// fk_resample_whole.hpp -- the exact cumulative sums of a WHOLE short weight vector in one round (host + device).
//
// systematic_resample / stratified_resample (filterpy/monte_carlo/resampling.py:117-150, :80-114) compare positions
// against numpy.cumsum(weights), a strictly sequential fp64 add chain.  resample_whole_kernel (resample_whole.hip) gives
// one workgroup the whole vector (Np <= 8 * threads) and reproduces that chain bit for bit without walking it:
//
//   1. plain fp64 prefix sums (any order, all terms >= 0) bound the exact running sum before and after every add to a
//      factor (1 +- WH_DELTA); an element whose two bounds lie in one binade e is CLEAN -- its add is the integer map
//      C -> C + inc_e(w) on the running sum in units of 2^e (fk_exact_scan.hpp) --, a zero weight is clean in any
//      binade; everything else -- the binade crossings themselves, the first non-zero weight (running sum 0), half-ulp
//      ties, ambiguous bounds -- is DIRTY and is added with a real IEEE add;
//   2. one wrapping 64-bit prefix sum of the increments and one of the dirty flags: the dirty elements (a few dozen for
//      random weights; more than WH_DMAX and the kernel runs the reference's loop literally) cut the vector into
//      SEGMENTS of clean elements that share one binade;
//   3. the segments are walked in order, two dependent adds per segment: the exact running sum entering a segment must
//      show the claimed binade and, with the segment's whole increment sum added, still show it -- then every prefix
//      inside did; a dirty element is one real add.  Nothing rests on the prediction: a segment that fails the check
//      fails the whole round (-> literal loop);
//   4. every element reads its segment's start and adds its scaled increment prefix: cs_j = c_start + (PS_j - PS0) 2^e.
//
// This is resample_onepass.hip's segmented_cumsum (round 2) re-cut so that (a) the weights stay in registers -- no LDS
// tile --, (b) the per-thread pieces are plain functions of the thread's eight elements, shared by the kernel and by the
// host emulation tests/hostcheck drives against numpy.cumsum and the merge loop before any GPU time is spent.
#pragma once

#include <stdint.h>

#include "fk_exact_scan.hpp"

namespace fk {

constexpr int WH_ITEMS = 8;               // consecutive weights per thread
constexpr int WH_DMAX = 256;              // dirty elements a vector may hold before the round declines
constexpr int WH_NONE = 0x7fffffff;       // a segment without a non-zero clean element claims no binade
// relative distance between the plain prefix sums and the sequential fp64 sums: both are within (n + 1) roundings of
// the real sum of n <= 8192 non-negative terms, 2 * 8193 * 2^-53 < 2^-38; 2^-35 leaves a factor of eight
constexpr double WH_DELTA = 0x1p-35;
constexpr double WH_SANE_LO = 0x1p-900, WH_SANE_HI = 0x1p900;

typedef unsigned long long wh_u64;

struct WhThread {
    unsigned dirty, claims;               // bit q: element q is dirty / claims a binade (clean and non-zero)
    int eq[WH_ITEMS];                     // ulp exponent of the binade the element's add happens in (if it claims)
    wh_u64 psum;                          // sum of the claiming elements' increments (wrapping)
    int ndirty;
};

// increment of a claiming element in its claimed binade: floor(w / 2^e + 1/2) (no tie: checked by wh_classify)
FK_HD wh_u64 wh_inc(double w, int e)
{
    return (wh_u64)floor(scale2(w, -e) + 0.5);
}

// step 1 for one thread: `before` = plain sum of every weight before the thread's first element, j0 = index of that
// element, len = length of the vector (elements >= len are padding: weight +0.0)
FK_HD void wh_classify(const double (&w)[WH_ITEMS], double before, int j0, int len, WhThread &t)
{
    t.dirty = t.claims = 0;
    t.psum = 0;
    double prev = before, arun = 0.0;
    FK_UNROLL for (int q = 0; q < WH_ITEMS; ++q) {
        arun += w[q];
        const double cur = before + arun;
        const double lo = prev * (1.0 - WH_DELTA), hi = cur * (1.0 + WH_DELTA);
        const bool known = lo > WH_SANE_LO && hi < WH_SANE_HI && ulp_exp(lo) == ulp_exp(hi);
        const int e = ulp_exp(lo);
        const double x = scale2(w[q], -e) + 0.5;
        const double i = floor(x);
        const bool zero = w[q] == 0.0;
        const bool ok = known && i != x && i < 0x1p53;         // no half-ulp tie; (i < 2^53 always holds: w <= hi)
        const bool in = j0 + q < len;
        if (in && !zero && ok) t.claims |= 1u << q;
        if (in && !zero && !ok) t.dirty |= 1u << q;
        t.eq[q] = e;
        t.psum += (in && !zero && ok) ? (wh_u64)i : (wh_u64)0;
        prev = cur;
    }
    t.ndirty = 0;
    FK_UNROLL for (int q = 0; q < WH_ITEMS; ++q) t.ndirty += (t.dirty >> q) & 1u;
}

// step 2 for one thread: dbase / pbase = dirty elements / increment sum before the thread.  Writes the thread's dirty
// elements into the lists and its claims into the segments (all claimants of a segment write the same value: checked
// by wh_claims_bad after a barrier).
FK_HD void wh_lists(const double (&w)[WH_ITEMS], const WhThread &t, int j0, int dbase, wh_u64 pbase, int *seg_e, int *d_pos,
                    double *d_w, wh_u64 *d_ps)
{
    int r = dbase;
    wh_u64 ps = pbase;
    FK_UNROLL for (int q = 0; q < WH_ITEMS; ++q) {
        if (t.claims & (1u << q)) {
            seg_e[r] = t.eq[q];
            ps += wh_inc(w[q], t.eq[q]);
        }
        if (t.dirty & (1u << q)) {
            d_pos[r] = j0 + q;
            d_w[r] = w[q];
            d_ps[r] = ps;
            ++r;
        }
    }
}

FK_HD bool wh_claims_bad(const WhThread &t, int dbase, const int *seg_e)
{
    int r = dbase;
    bool bad = false;
    FK_UNROLL for (int q = 0; q < WH_ITEMS; ++q) {
        if ((t.claims & (1u << q)) && seg_e[r] != t.eq[q]) bad = true;    // two binades claimed inside one segment
        if (t.dirty & (1u << q)) ++r;
    }
    return bad;
}

// what segment r (the clean elements before dirty element r; r = D: those after the last one) contributes to the
// chain, prepared independently of the running sum: its whole increment sum as a double in units of ONE (I 2^e: exact)
// and the biased exponent its running sum must show (-1: a segment of zeros, any)
struct WhSeg {
    double add;
    int xf;
    bool bad;                              // increment sum out of range / increments without a claim
    wh_u64 ps0;                            // increment prefix at its start
};
FK_HD WhSeg wh_segment(int r, int D, wh_u64 ptotal, const int *seg_e, const wh_u64 *d_ps)
{
    const wh_u64 end = r < D ? d_ps[r] : ptotal;
    const wh_u64 start = r >= 1 ? d_ps[r - 1] : 0;
    const int e = seg_e[r];
    const wh_u64 I = end - start;                                          // (wrapping; < 2^53 for a segment that passes)
    const bool claim = e != WH_NONE;
    WhSeg s;
    s.bad = (claim && !(I < (1ull << 53))) || (!claim && I != 0);
    s.add = claim ? scale2((double)I, e) : 0.0;
    s.xf = claim ? e + 1075 : -1;
    s.ps0 = start;
    return s;
}

// one step of the chain: c = running sum entering segment r.  Returns the running sum after the segment; `fail` is
// raised when the segment does not start and end in the claimed binade.
//   c + I 2^e is exact when C0 + I < 2^53 (the sum is representable); when it is not, the rounded result is >= 2^(e+53),
//   the next binade (rounding is monotone, the bound is representable): the exponent field of the result tells.
FK_HD double wh_chain_segment(double c, double add, int xf, bool &fail)
{
    const int x0 = (int)((double_to_bits(c) >> 52) & 0x7ffu);
    c = c + add;
    const int x1 = (int)((double_to_bits(c) >> 52) & 0x7ffu);
    fail = fail || (xf >= 0 && (x0 != xf || x1 != xf || xf <= 1075 - 900 || xf >= 1075 + 900 - 52));
    return c;
}

// step 3, serial form (host emulation; the kernel runs the same steps with the segments spread over the lanes of one
// wave): fills seg_c / seg_ps0 / d_cs, returns false when the round must decline
FK_HD bool wh_chain_serial(int D, wh_u64 ptotal, const int *seg_e, const wh_u64 *d_ps, const double *d_w, double *seg_c,
                           wh_u64 *seg_ps0, double *d_cs, double *carry_out)
{
    double c = 0.0;
    bool fail = false;
    for (int r = 0; r <= D; ++r) {
        const WhSeg s = wh_segment(r, D, ptotal, seg_e, d_ps);
        fail = fail || s.bad;
        seg_c[r] = c;
        seg_ps0[r] = s.ps0;
        c = wh_chain_segment(c, s.add, s.xf, fail);
        if (r < D) {
            c = c + d_w[r];                                                // the real IEEE add of the dirty element
            d_cs[r] = c;
        }
    }
    *carry_out = c;
    return !fail;
}

// step 4 for one thread: the cumulative sums of its elements (elements >= len: the carry-out, i.e. no slots)
FK_HD void wh_cumsums(const double (&w)[WH_ITEMS], const WhThread &t, int j0, int len, int dbase, wh_u64 pbase, const int *seg_e,
                      const double *seg_c, const wh_u64 *seg_ps0, const double *d_cs, double carry_out, double (&cs)[WH_ITEMS])
{
    int r = dbase;
    wh_u64 ps = pbase;
    FK_UNROLL for (int q = 0; q < WH_ITEMS; ++q) {
        if (t.claims & (1u << q)) ps += wh_inc(w[q], t.eq[q]);
        double c;
        if (t.dirty & (1u << q)) {
            c = d_cs[r];
            ++r;
        } else {
            // (C0 + dPS) 2^e = c_start + dPS 2^e: exact for the same reason as in the chain
            const int e = seg_e[r];
            c = e == WH_NONE ? seg_c[r] : seg_c[r] + scale2((double)(ps - seg_ps0[r]), e);
        }
        cs[q] = j0 + q < len ? c : carry_out;
    }
}

}  // namespace fk
