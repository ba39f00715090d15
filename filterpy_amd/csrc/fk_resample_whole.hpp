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

// The arithmetic below avoids v_ldexp_f64 / v_floor_f64 / v_cvt_* sequences: scalings are multiplications by a power of two
// assembled from its exponent field, roundings and integer conversions are additions of 2^52 (every integer involved is below
// 2^52: see wh_classify / wh_segment) -- one instruction where floor + convert are two.  (The first cut built on them spent 9k
// of its 32k clocks per filter classifying; they were taken for quarter-rate instructions then.  Measured since:
// tools/experiments/valu_latency.hip -- on gfx950 they issue at the full rate; only v_rcp / v_rsq_f64 are quarter rate.)
constexpr double WH_M52 = 0x1p52;
FK_HD double wh_pow2(int k)                            // 2^k, -1022 <= k <= 1023
{
    return bits_to_double((wh_u64)(unsigned)(k + 1023) << 52);
}
FK_HD wh_u64 wh_to_u64(double v)                       // integer-valued v in [0, 2^52)
{
    return double_to_bits(v + WH_M52) & 0x000fffffffffffffull;
}
FK_HD double wh_to_f64(wh_u64 v)                       // v < 2^52
{
    return bits_to_double(v | 0x4330000000000000ull) - WH_M52;
}
// the increment of weight w where the running sum has ulp 2^e (s = 2^-e): r = w / ulp rounded to the nearest integer
// (= floor(w / ulp + 1/2) of fk_exact_scan.hpp's fast_inc away from ties).  ok = no exact half-ulp tie and r < 2^52 (a
// valid increment always is: C >= 2^52 and C + inc < 2^53; above 2^52 the rounding trick returns garbage >= 2^52).
FK_HD bool wh_increment(double w, double s, double &r)
{
    const double t = w * s;                            // exact (or a gradual underflow far below 1/2)
    r = (t + WH_M52) - WH_M52;                         // nearest integer (ties to even), exact for t < 2^52
    const double d = t - r;                            // exact: |d| <= 1/2
    return fabs(d) != 0.5 && r < WH_M52;
}

struct WhThread {
    unsigned dirty, claims;               // bit q: element q is dirty / claims a binade (clean and non-zero)
    int eq[WH_ITEMS];                     // ulp exponent of the binade the element's add happens in (where it claims)
    double inc[WH_ITEMS];                 // its increment there, floor(w / 2^e + 1/2) < 2^52 (0 where it does not claim)
    wh_u64 psum;                          // sum of the thread's increments (wrapping)
    int ndirty;
    bool uniform;                         // all claims in ONE binade eq[0] and no dirty element
    bool overflow;                        // increment sum not below 2^53: the round declines
};

// one element against the bounds [lo, hi] on the running sum before / after its add
FK_HD void wh_classify_element(double w, double lo, double hi, bool in, int q, WhThread &t)
{
    const bool known = lo > WH_SANE_LO && hi < WH_SANE_HI && ulp_exp(lo) == ulp_exp(hi);
    const int e = known ? ulp_exp(lo) : 0;
    double i;
    const bool ok = wh_increment(w, wh_pow2(-e), i) && known;
    const bool zero = w == 0.0;
    const bool claim = in && !zero && ok;
    if (claim) t.claims |= 1u << q;
    if (in && !zero && !ok) t.dirty |= 1u << q;
    t.eq[q] = e;
    t.inc[q] = claim ? i : 0.0;
}

// elements [Q0, Q0 + 4) of a thread: `before` = plain sum of every weight before element Q0, `hsum` = plain sum of the four
template <int Q0>
FK_HD bool wh_classify_half(const double (&w)[WH_ITEMS], double before, double hsum, int j0, int len, WhThread &t)
{
    const double tlo = before * (1.0 - WH_DELTA), thi = (before + hsum) * (1.0 + WH_DELTA);
    const bool one = tlo > WH_SANE_LO && thi < WH_SANE_HI && ulp_exp(tlo) == ulp_exp(thi);
    if (one) {
        // the running sum stays in ONE binade across the four adds: the bounds of every element lie inside [tlo, thi]
        const int e = ulp_exp(tlo);
        const double sc = wh_pow2(-e);
        FK_UNROLL for (int q = Q0; q < Q0 + 4; ++q) {
            double i;
            const bool ok = wh_increment(w[q], sc, i);
            const bool zero = w[q] == 0.0;
            const bool in = j0 + q < len;
            const bool claim = in && !zero && ok;
            if (claim) t.claims |= 1u << q;
            if (in && !zero && !ok) t.dirty |= 1u << q;
            t.eq[q] = e;
            t.inc[q] = claim ? i : 0.0;
        }
    } else {
        double prev = before, arun = 0.0;
        FK_UNROLL for (int q = Q0; q < Q0 + 4; ++q) {
            arun += w[q];
            const double cur = before + arun;
            wh_classify_element(w[q], prev * (1.0 - WH_DELTA), cur * (1.0 + WH_DELTA), j0 + q < len, q, t);
            prev = cur;
        }
    }
    return one;
}

// step 1 for one thread: `before` = plain sum of every weight before the thread's first element, j0 = index of that
// element, len = length of the vector (elements >= len are padding: weight +0.0).  The per-element bounds are only
// formed in the HALF of the thread that holds a crossing (or the vector's start): the waves that hold one set the pace
// of the phase (it ends in a barrier).
FK_HD void wh_classify(const double (&w)[WH_ITEMS], double before, int j0, int len, WhThread &t)
{
    t.dirty = t.claims = 0;
    const double h0 = (w[0] + w[1]) + (w[2] + w[3]), h1 = (w[4] + w[5]) + (w[6] + w[7]);
    const bool one0 = wh_classify_half<0>(w, before, h0, j0, len, t);
    const bool one1 = wh_classify_half<4>(w, before + h0, h1, j0, len, t);
    t.uniform = one0 && one1 && t.eq[0] == t.eq[4] && t.dirty == 0;
    // the thread's increment sum (wrapping 64-bit), segment by segment: a valid segment starts at C >= 2^52 and ends
    // below 2^53, so its increments add up to less than 2^52 -- exactly, as doubles; a sum that reaches 2^52 stays there
    // after rounding (monotone): `overflow` fails the round.  Segments end at the thread's dirty elements; the
    // increments of different segments (other binades) do not add up to anything bounded and meet as integers.
    t.overflow = false;
    t.psum = 0;
    double acc = 0.0;
    if (t.dirty == 0) {
        FK_UNROLL for (int q = 0; q < WH_ITEMS; ++q) acc += t.inc[q];
    } else {
        FK_UNROLL for (int q = 0; q < WH_ITEMS; ++q) {
            if (t.dirty & (1u << q)) {
                t.overflow = t.overflow || !(acc < WH_M52);
                t.psum += wh_to_u64(acc < WH_M52 ? acc : 0.0);
                acc = 0.0;
            }
            acc += t.inc[q];
        }
    }
    t.overflow = t.overflow || !(acc < WH_M52);
    t.psum += wh_to_u64(acc < WH_M52 ? acc : 0.0);
#if defined(__HIP_DEVICE_COMPILE__)
    t.ndirty = __builtin_popcount(t.dirty);
#else
    t.ndirty = 0;
    for (int q = 0; q < WH_ITEMS; ++q) t.ndirty += (t.dirty >> q) & 1u;
#endif
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
    double acc = 0.0;                                                      // (as in wh_classify: one conversion per segment)
    FK_UNROLL for (int q = 0; q < WH_ITEMS; ++q) {
        if (t.claims & (1u << q)) {
            seg_e[r] = t.eq[q];
            acc += t.inc[q];
        }
        if (t.dirty & (1u << q)) {
            ps += wh_to_u64(acc < WH_M52 ? acc : 0.0);
            acc = 0.0;
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
    const wh_u64 I = end - start;                                          // (wrapping; < 2^52 for a valid segment)
    const bool claim = e != WH_NONE;
    WhSeg s;
    s.bad = (claim && !(I < (1ull << 52))) || (!claim && I != 0);
    s.add = (claim && !s.bad) ? wh_to_f64(I) * wh_pow2(e) : 0.0;
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

// what the one-fma boundary estimate needs from the segment an element lies in
template <bool STRATIFIED>
struct WhSegPos {
    double C0, Nu, K;                     // cs = (C0 + Et) 2^e;  N cs - u = fma(Et, Nu, K),  Nu = N 2^e,  K = C0 Nu - u
    int e;
};
template <bool STRATIFIED>
FK_HD WhSegPos<STRATIFIED> wh_seg_pos(int r, const int *seg_e, const double *seg_c, const WhPos<STRATIFIED> &px)
{
    WhSegPos<STRATIFIED> s;
    s.e = seg_e[r];
    const double c = seg_c[r];
    const bool claim = s.e != WH_NONE;
    // a segment of zeros claims no binade: its elements all have cs = c, the estimate is the constant N c - u
    s.C0 = claim ? c * wh_pow2(-s.e) : c;
    s.Nu = claim ? px.Nd * wh_pow2(s.e) : 0.0;
    s.K = fma(s.C0, claim ? s.Nu : px.Nd, STRATIFIED ? 0.0 : -px.u_sys);
    return s;
}

// step 4 for one thread: the slot boundaries n(cs_j) of its elements.  ONE loop for every thread:
//   * a clean element of a segment that claims binade e has cs_j = (C0 + Et_j) 2^e exactly, C0 = c_start / 2^e and
//     Et_j the increment prefix since the segment's start, so N cs_j - u = fma(Et_j, N 2^e, C0 N 2^e - u): one fma gives
//     the estimate whose ceiling is n_j whenever it is not within eps of an integer (n_boundary_fast's argument: two
//     roundings of 2^-22 slots each, the same budget; round 2's quick_boundaries); the rest -- about one weight in
//     10^5 -- takes the exact tests on cs_j;
//   * a dirty element (a few per vector) takes its cumulative sum from the chain and starts the next segment.
// The two rare cases are branches inside the loop: a wave runs them only at the positions where one of its lanes needs
// them.  (The first cut ran a general per-element form for every wave that held one dirty element: the slowest wave
// sets the pace of a phase that ends in a barrier, 9.8k of 40k clocks per filter.)  Padding elements (>= len) are zero
// weights: clean, no slots -- they repeat the boundary of the element before them.
template <bool STRATIFIED>
FK_HD void wh_boundaries(const WhThread &t, int dbase, wh_u64 pbase, const int *seg_e, const double *seg_c,
                         const wh_u64 *seg_ps0, const double *d_cs, const WhPos<STRATIFIED> &px, int (&nb)[WH_ITEMS])
{
    int r = dbase;
    WhSegPos<STRATIFIED> sp = wh_seg_pos<STRATIFIED>(r, seg_e, seg_c, px);
    double Et = wh_to_f64(pbase - seg_ps0[r]);
    FK_UNROLL for (int q = 0; q < WH_ITEMS; ++q) {
        int n;
        if (t.dirty & (1u << q)) {                                         // rare
            n = n_boundary_fast<STRATIFIED>(d_cs[r], px.Np, px.Nd, px.halfNd, px.u_sys, px.u_str);
            ++r;
            sp = wh_seg_pos<STRATIFIED>(r, seg_e, seg_c, px);
            Et = 0.0;                                                      // the next segment starts behind this element
        } else {
            Et += t.inc[q];                                                // exact: C0 + Et < 2^53 (the chain checked it)
            const double est = fma(Et, sp.Nu, sp.K);
            // floor(est) + 1 without v_floor / v_cvt: a = est + 1.5 2^52 holds the nearest integer of est in its low
            // mantissa bits (two's complement: est > -1 here), d = est - nearest is exact
            const double a = est + 0x1.8p52;
            const double d = est - (a - 0x1.8p52);
            const int ri = (int)(unsigned)double_to_bits(a);
            bool sure = fabs(d) > N_BOUNDARY_EPS && est < px.Nd;           // est within eps of an integer: the exact tests
            n = ri + (d < 0.0 ? 0 : 1);
            if (STRATIFIED) {
                const int fl = ri - (d < 0.0 ? 1 : 0);
                const double fr = d < 0.0 ? d + 1.0 : d;                   // est - floor(est), exact
                const double uf = px.u_str[(est < px.Nd && est >= 0.0) ? fl : 0];
                const double gap = uf - fr;
                sure = sure && est >= 0.0 && (gap > N_BOUNDARY_EPS || gap < -N_BOUNDARY_EPS);
                n = fl + (gap > 0.0 ? 0 : 1);
            }
            if (!sure) {                                                   // rare
                const double cs = sp.e != WH_NONE ? scale2(sp.C0 + Et, sp.e) : sp.C0;
                n = n_boundary<STRATIFIED>(cs, px.Np, px.Nd, px.halfNd, px.u_sys, px.u_str);
            }
        }
        nb[q] = n;
    }
}

// ---- step 0: the boundaries from the PLAIN prefix sums alone -------------------------------------------------------------
// n(c) = #{ i : pos_i < c } is a step function of c, and the slot boundary of weight j only asks on which side of the
// positions cs_j lies.  numpy's sequential cs_j is within j 2^-53 <= 2^-40 (n <= 8192) of the real sum s_j of the same
// non-negative terms; the plain prefix sum a_j formed here is within D 2^-53 of it, D = the DEPTH of its adds -- 8 inside a
// thread, 6 in the wave scan, 1 + 15 over the waves, 8 to the element: D <= 38 on the GPU, <= 96 in the host emulation (its
// wave scan is serial) --, i.e. within 2^-46.  So |N a_j - N cs_j| <= (2^-40 + 2^-46) N a_j; the estimate's own two roundings
// add 2^-52 N a_j, the positions' (fl(fl(u + i) / N), N <= 8192) 2^-39 slots.  Unless N a_j - u sits within
//     band = 1.5 * 2^-40 N a_j + 2^-36
// of an integer -- 1.45 times the relative terms, 8 times the absolute one -- its ceiling IS n(cs_j): no exact cumulative sum
// is needed.  (Rounds 3 / 4 used 2^-36 N a_j + 2^-30, pricing a_j at n roundings like cs_j: one vector of 8000 normalised
// weights in ~500 had an element inside that band; with this one it is one in ~5000 -- which is what lets the quick kernel of
// round 5 finish nearly every CALL by itself.)  Such a vector (and only such a vector) runs the exact round above.  Returns
// the mask of elements inside the band.
//   Stratified (pos_i = fl(fl(u_i + i) / N)): with f = floor(N c), slot f - 1 is below c, slot f + 1 is not, slot f is
// iff u_f < frac(N c); an estimate within the band of an integer k leaves f = k - 1 or k open, but n = k either way
// unless u_{k-1} or u_k is itself within the band of 1 or 0 -- a normalised vector ends exactly there (N c ~ N).
template <bool STRATIFIED>
FK_HD unsigned wh_approx_boundaries(const double (&w)[WH_ITEMS], double before, const WhPos<STRATIFIED> &px, int (&nb)[WH_ITEMS])
{
    unsigned unsure = 0;
    double a = before;
    FK_UNROLL for (int q = 0; q < WH_ITEMS; ++q) {
        a += w[q];                                                         // plain inclusive prefix
        const double pe = px.Nd * a;                                       // N a_j >= 0
        const double est = STRATIFIED ? pe : pe - px.u_sys;
        const double band = fma(pe, 0x1.8p-40, 0x1p-36);
        const bool past = !(est < px.Nd);                                  // every position is below a_j: n = Np
        // nearest integer k of est without v_floor / v_cvt (-1 < est < Nd < 2^31 where it is used)
        const double m = est + 0x1.8p52;
        const double d = est - (m - 0x1.8p52);                             // exact, |d| <= 1/2
        const int k = (int)(unsigned)double_to_bits(m);
        const bool clear = fabs(d) > band;                                 // est is not within the band of an integer
        bool sure = clear;
        int n = k + (d < 0.0 ? 0 : 1);                                     // systematic: the ceiling of est
        if (STRATIFIED) {
            const int fl = k - (d < 0.0 ? 1 : 0);
            const double fr = d < 0.0 ? d + 1.0 : d;
            const double uf = px.u_str[past ? 0 : fl];
            const double gap = uf - fr;
            sure = clear && (gap > band || gap < -band);
            n = fl + (gap > 0.0 ? 0 : 1);
            if (!clear && !past) {                                         // rare: est ~ k
                const double u_lo = k >= 1 ? px.u_str[k - 1] : 0.0, u_hi = k < px.Np ? px.u_str[k] : 1.0;
                sure = u_lo < 1.0 - 2.0 * band && u_hi > 2.0 * band;
                n = k;
            }
        }
        if (past) {
            n = px.Np;
            sure = true;
        }
        unsure |= sure ? 0u : (1u << q);
        nb[q] = n;
    }
    return unsure;
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
