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
//      ties, ambiguous bounds, a weight of half the running sum or more -- is DIRTY and is added with a real IEEE add;
//   2. one wrapping 64-bit prefix sum of the increments and one of the dirty flags: the dirty elements (a few dozen for
//      random weights; more than WH_DMAX and the kernel runs the reference's loop literally) cut the vector into
//      SEGMENTS of clean elements that share one binade;
//   3. the segments are walked in order, two dependent adds per segment: the exact running sum entering a segment must
//      show the claimed binade and, with the segment's whole increment sum added, still show it -- then every prefix
//      inside did; a dirty element is one real add.  Nothing rests on the prediction: a segment that fails the check
//      fails the whole round (-> literal loop);
//   4. every element reads its segment's start and adds its scaled increment prefix: cs_j = c_start + (PS_j - PS0) 2^e;
//      its slot boundary n(cs_j) (fk_resample_math.hpp) follows from ONE fma on the increment prefix where the thread's
//      elements all lie in one segment that claims a binade, from cs_j itself elsewhere.
//
// This is resample_onepass.hip's segmented_cumsum (round 2) re-cut so that (a) the weights stay in registers -- no LDS
// tile --, (b) a thread whose eight elements share a binade (nearly all do: a vector of 8000 weights crosses ~13) never
// runs the per-element machinery -- the first cut did, and measured 41k clocks per filter, all VALU issue --, (c) the
// per-thread pieces are plain functions shared by the kernel and by the host emulation that tests/hostcheck drives
// against numpy.cumsum and the merge loop before any GPU time is spent.
#pragma once

#include <stdint.h>

#include "fk_exact_scan.hpp"
#include "fk_resample_math.hpp"

namespace fk {

constexpr int WH_ITEMS = 8;               // consecutive weights per thread
constexpr int WH_DMAX = 256;              // dirty elements a vector may hold before the round declines
constexpr int WH_NONE = 0x7fffffff;       // a segment without a non-zero clean element claims no binade
// relative distance between the plain prefix sums and the sequential fp64 sums: both are within (n + 1) roundings of
// the real sum of n <= 8192 non-negative terms, 2 * 8193 * 2^-53 < 2^-38; 2^-35 leaves a factor of eight
constexpr double WH_DELTA = 0x1p-35;
constexpr double WH_SANE_LO = 0x1p-900, WH_SANE_HI = 0x1p900;

typedef unsigned long long wh_u64;

// integer-valued double in [0, 2^64) <-> wh_u64, exactly, without the library's range handling: two 32-bit halves
FK_HD wh_u64 wh_to_u64(double v)
{
    const double hi = floor(v * 0x1p-32);
    const double lo = v - hi * 0x1p32;                 // exact: [0, 2^32)
    return ((wh_u64)(unsigned)hi << 32) | (wh_u64)(unsigned)lo;
}
FK_HD double wh_to_f64(wh_u64 v)                       // exact for v < 2^53
{
    return (double)(unsigned)(v >> 32) * 0x1p32 + (double)(unsigned)v;
}

struct WhThread {
    unsigned dirty, claims;               // bit q: element q is dirty / claims a binade (clean and non-zero)
    int eq[WH_ITEMS];                     // ulp exponent of the binade the element's add happens in (where it claims)
    double inc[WH_ITEMS];                 // its increment there, floor(w / 2^e + 1/2) < 2^52 (0 where it does not claim)
    wh_u64 psum;                          // sum of the thread's increments (wrapping)
    int ndirty;
    bool uniform;                         // all claims in ONE binade eq[0] and no dirty element: the one-fma boundary path
};

// one element against the bounds [lo, hi] on the running sum before / after its add
FK_HD void wh_classify_element(double w, double lo, double hi, bool in, int q, WhThread &t)
{
    const bool known = lo > WH_SANE_LO && hi < WH_SANE_HI && ulp_exp(lo) == ulp_exp(hi);
    const int e = ulp_exp(lo);
    const double x = scale2(w, -e) + 0.5;
    const double i = floor(x);
    const bool zero = w == 0.0;
    // no half-ulp tie, and an increment below 2^52 (a valid one always is: C >= 2^52 and C + inc < 2^53)
    const bool ok = known && i != x && i < 0x1p52;
    const bool claim = in && !zero && ok;
    if (claim) t.claims |= 1u << q;
    if (in && !zero && !ok) t.dirty |= 1u << q;
    t.eq[q] = e;
    t.inc[q] = claim ? i : 0.0;
}

// step 1 for one thread: `before` = plain sum of every weight before the thread's first element, `tsum` = plain sum of
// its own, j0 = index of its first element, len = length of the vector (elements >= len are padding: weight +0.0)
FK_HD void wh_classify(const double (&w)[WH_ITEMS], double before, double tsum, int j0, int len, WhThread &t)
{
    t.dirty = t.claims = 0;
    const double tlo = before * (1.0 - WH_DELTA), thi = (before + tsum) * (1.0 + WH_DELTA);
    const bool one = tlo > WH_SANE_LO && thi < WH_SANE_HI && ulp_exp(tlo) == ulp_exp(thi);
    if (one) {
        // the running sum stays in ONE binade across all eight adds: the bounds of every element lie inside [tlo, thi]
        const int e = ulp_exp(tlo);
        FK_UNROLL for (int q = 0; q < WH_ITEMS; ++q) {
            const double x = scale2(w[q], -e) + 0.5;
            const double i = floor(x);
            const bool zero = w[q] == 0.0;
            const bool ok = i != x && i < 0x1p52;
            const bool in = j0 + q < len;
            const bool claim = in && !zero && ok;
            if (claim) t.claims |= 1u << q;
            if (in && !zero && !ok) t.dirty |= 1u << q;
            t.eq[q] = e;
            t.inc[q] = claim ? i : 0.0;
        }
    } else {
        double prev = before, arun = 0.0;
        FK_UNROLL for (int q = 0; q < WH_ITEMS; ++q) {
            arun += w[q];
            const double cur = before + arun;
            wh_classify_element(w[q], prev * (1.0 - WH_DELTA), cur * (1.0 + WH_DELTA), j0 + q < len, q, t);
            prev = cur;
        }
    }
    t.ndirty = 0;
    t.psum = 0;
    FK_UNROLL for (int q = 0; q < WH_ITEMS; ++q) {
        t.ndirty += (t.dirty >> q) & 1u;
        t.psum += wh_to_u64(t.inc[q]);
    }
    t.uniform = one && t.dirty == 0 && j0 + WH_ITEMS <= len;          // (a thread with padding takes the general form)
}

// step 2 for one thread: dbase / pbase = dirty elements / increment sum before the thread.  Writes the thread's dirty
// elements into the lists and its claims into the segments (all claimants of a segment write the same value: checked
// by wh_claims_bad after a barrier).
FK_HD void wh_lists(const double (&w)[WH_ITEMS], const WhThread &t, int dbase, wh_u64 pbase, int *seg_e, double *d_w, wh_u64 *d_ps)
{
    if (t.uniform) {
        if (t.claims) seg_e[dbase] = t.eq[0];
        return;
    }
    int r = dbase;
    wh_u64 ps = pbase;
    FK_UNROLL for (int q = 0; q < WH_ITEMS; ++q) {
        if (t.claims & (1u << q)) {
            seg_e[r] = t.eq[q];
            ps += wh_to_u64(t.inc[q]);
        }
        if (t.dirty & (1u << q)) {
            d_w[r] = w[q];
            d_ps[r] = ps;
            ++r;
        }
    }
}

FK_HD bool wh_claims_bad(const WhThread &t, int dbase, const int *seg_e)
{
    if (t.uniform) return t.claims != 0 && seg_e[dbase] != t.eq[0];
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
    const wh_u64 I = end - start;                                          // (wrapping; < 2^52 for a segment that passes)
    const bool claim = e != WH_NONE;
    WhSeg s;
    s.bad = (claim && !(I < (1ull << 53))) || (!claim && I != 0);
    s.add = claim ? scale2(wh_to_f64(I), e) : 0.0;
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

// geometry of the positions, shared by every boundary of one vector
template <bool STRATIFIED>
struct WhPos {
    int Np;
    double Nd, halfNd, u_sys;
    const double *u_str;
};

// The rare paths below run as ONE rolled copy instead of eight inlined ones.  Registers cannot be indexed -- and a chain
// of selects on the loop counter is turned into an indexed scratch access by the optimiser, which moves the whole array
// to scratch memory for the hot path too -- so the arrays are ROTATED: every trip works on element 0 and shifts.
template <class T>
FK_HD T wh_rotate(T (&a)[WH_ITEMS], T last)
{
    const T first = a[0];
    FK_UNROLL for (int r = 0; r + 1 < WH_ITEMS; ++r) a[r] = a[r + 1];
    a[WH_ITEMS - 1] = last;
    return first;
}

// step 4 for one thread: the slot boundaries n(cs_j) of its elements (elements >= len: the carry-out, i.e. no slots)
template <bool STRATIFIED>
FK_HD void wh_boundaries(const WhThread &t, int j0, int len, int dbase, wh_u64 pbase, const int *seg_e, const double *seg_c,
                         const wh_u64 *seg_ps0, const double *d_cs, double carry_out, const WhPos<STRATIFIED> &px,
                         int (&nb)[WH_ITEMS])
{
    const int e0 = seg_e[dbase];
    if (t.uniform && e0 != WH_NONE) {
        // all eight elements lie in segment `dbase`, which claims binade e0: cs_j = (C0 + Et_j) 2^e0 exactly with
        // C0 = c_start / 2^e0 and Et_j the increment prefix since the segment's start, so N cs_j - u = fma(Et_j, N 2^e0,
        // C0 N 2^e0 - u): ONE fma per weight gives the estimate whose ceiling is n_j whenever it is not within eps of
        // an integer (n_boundary_fast's argument: two roundings of 2^-22 slots each, the same budget); the rare rest
        // takes the exact tests on cs_j.  (Round 2's quick_boundaries, per thread instead of per chunk.)
        const double C0 = scale2(seg_c[dbase], -e0), Nu = scale2(px.Nd, e0);
        const double K = fma(C0, Nu, STRATIFIED ? 0.0 : -px.u_sys);
        const double E0 = wh_to_f64(pbase - seg_ps0[dbase]);
        double Et = E0;
        unsigned unsure = 0;
        FK_UNROLL for (int q = 0; q < WH_ITEMS; ++q) {
            Et += t.inc[q];                                                // exact: C0 + Et < 2^53 (the chain checked it)
            const double est = fma(Et, Nu, K);
            const double fl = floor(est), fr = est - fl;
            bool sure = fr > N_BOUNDARY_EPS && fr < 1.0 - N_BOUNDARY_EPS && est < px.Nd;
            int n = (int)fl + 1;
            if (STRATIFIED) {
                const double uf = px.u_str[est < px.Nd ? (int)fl : 0];     // est >= 0 here
                const double gap = uf - fr;
                sure = sure && (gap > N_BOUNDARY_EPS || gap < -N_BOUNDARY_EPS);
                n = (int)fl + (gap > 0.0 ? 0 : 1);
            }
            unsure |= sure ? 0u : (1u << q);
            nb[q] = n;
        }
        if (unsure) {                                                      // about one weight in 10^5: the exact tests
            double incr[WH_ITEMS];
            FK_UNROLL for (int q = 0; q < WH_ITEMS; ++q) incr[q] = t.inc[q];
            double Eq = E0;
            _Pragma("nounroll") for (int q = 0; q < WH_ITEMS; ++q) {
                Eq += wh_rotate(incr, 0.0);
                int n = nb[0];
                if (unsure & 1u) n = n_boundary<STRATIFIED>(scale2(C0 + Eq, e0), px.Np, px.Nd, px.halfNd, px.u_sys, px.u_str);
                unsure >>= 1;
                wh_rotate(nb, n);
            }
        }
        return;
    }
    // the general form, one rolled copy: a thread that holds a dirty element or padding, or whose segment claims nothing
    int r = dbase;
    wh_u64 ps = pbase;
    double incr[WH_ITEMS];
    FK_UNROLL for (int q = 0; q < WH_ITEMS; ++q) {
        incr[q] = t.inc[q];
        nb[q] = 0;
    }
    unsigned claims = t.claims, dirty = t.dirty;
    _Pragma("nounroll") for (int q = 0; q < WH_ITEMS; ++q) {
        const double inc0 = wh_rotate(incr, 0.0);
        if (claims & 1u) ps += wh_to_u64(inc0);
        double c;
        if (dirty & 1u) {
            c = d_cs[r];
            ++r;
        } else {
            // (C0 + dPS) 2^e = c_start + dPS 2^e: exact for the same reason as in the chain
            const int e = seg_e[r];
            c = e == WH_NONE ? seg_c[r] : seg_c[r] + scale2(wh_to_f64(ps - seg_ps0[r]), e);
        }
        claims >>= 1;
        dirty >>= 1;
        if (j0 + q >= len) c = carry_out;                                  // padding: the vector's last boundary, no slots
        wh_rotate(nb, n_boundary_fast<STRATIFIED>(c, px.Np, px.Nd, px.halfNd, px.u_sys, px.u_str));
    }
}

// the cumulative sums themselves (tests only: the kernel goes straight to the boundaries)
FK_HD void wh_cumsums(const WhThread &t, int j0, int len, int dbase, wh_u64 pbase, const int *seg_e, const double *seg_c,
                      const wh_u64 *seg_ps0, const double *d_cs, double carry_out, double (&cs)[WH_ITEMS])
{
    int r = dbase;
    wh_u64 ps = pbase;
    for (int q = 0; q < WH_ITEMS; ++q) {
        if (t.claims & (1u << q)) ps += wh_to_u64(t.inc[q]);
        double c;
        if (t.dirty & (1u << q)) {
            c = d_cs[r];
            ++r;
        } else {
            const int e = seg_e[r];
            c = e == WH_NONE ? seg_c[r] : seg_c[r] + scale2(wh_to_f64(ps - seg_ps0[r]), e);
        }
        cs[q] = j0 + q < len ? c : carry_out;
    }
}

}  // namespace fk
