// fk_resample_math.hpp -- division-free exact slot boundaries for the systematic / stratified resamplers.
//
// filterpy/monte_carlo/resampling.py:103,139 build the positions pos_i = fl(fl(u_i + i) / N) and the
// two-pointer merge (:106-112, :142-149) gives output slot i the index  #{ j : cs_j <= pos_i }.  Seen from the
// weights' side: weight j owns the contiguous slots [n_{j-1}, n_j), where
//
//        n(c) = #{ i : pos_i < c } = min{ i : pos_i >= c }            (positions are non-decreasing in i)
//
// evaluated at the cumulative sums c = cs_j.  n(c) is one multiply away from its estimate N*c - u; what makes
// it EXACT is the comparison pos_i >= c, and that comparison does not need the division:
//
//   RN(a / N) >= c   <=>   a / N  >=  the lower end of c's rounding interval, i.e. the midpoint between
//                          pred(c) and c (which rounds to c iff c's last mantissa bit is 0: ties-to-even)
//                    <=>   N*c - a  <=  N*h,   h = (c - pred(c)) / 2       (strict when c is odd)
//
// and d = fma(N, c, -a) delivers N*c - a EXACTLY whenever it is anywhere near N*h (it is then an integer
// below 2^31 in units of ulp(c)); farther away only its order relative to N*h matters and rounding is
// monotone.  N*h is a power of two times N: exact.  One FMA and one compare per test instead of the
// 14-instruction IEEE division -- and no rounding-mode or Markstein subtleties: the IEEE quotient is never formed.
//
// __host__ __device__: tests/hostcheck_rs holds n_boundary against the brute-force count with real divisions.
#pragma once

#include <stdint.h>

#include "fk_exact_scan.hpp"

namespace fk {

// is  fl(a / N) >= c ?   (a = fl(u + i) >= 0;  c > 0 finite, far from the subnormal range;  Nd = (double)N,
// halfNd = 0.5 * N)
FK_HD bool pos_ge(double a, double c, double Nd, double halfNd)
{
    const uint64_t cb = double_to_bits(c);
    const double gap = c - bits_to_double(cb - 1);      // c - pred(c): exact (also when c is a power of two)
    const double Nh = gap * halfNd;                     // N * h: exact
#if defined(__HIP_DEVICE_COMPILE__)
    const double d = __builtin_fma(Nd, c, -a);
#else
    const double d = fma(Nd, c, -a);
#endif
    return d < Nh || (d == Nh && (cb & 1) == 0);
}

// n(c) = min{ i in [0, Np] : pos_i >= c }  with pos_i = fl(fl(u_i + i) / Np);  u_i = u_sys (systematic) or
// u_str[i] (stratified).  c <= 0 -> 0.  The estimate brackets the answer within three candidates; the two
// rare fix-up loops make the result independent of that error analysis.
template <bool STRATIFIED>
FK_HD int n_boundary(double c, int Np, double Nd, double halfNd, double u_sys, const double *u_str)
{
    if (!(c > 0.0)) return 0;
    // systematic: N c - u  -> n in { floor, floor + 1, floor + 2 };  stratified: N c - 1 (n >= floor(N c) - 1)
    const double e = c * Nd - (STRATIFIED ? 1.0 : u_sys);
    if (e >= (double)Np) return Np;                     // every position is below c (margin ~1 slot >> rounding)
    const int b = e > 0.0 ? (int)e : 0;                 // in [0, Np - 1]
    auto test = [&](int i) -> bool {                    // pos_i >= c ?
        const double ui = STRATIFIED ? u_str[i] : u_sys;
        return pos_ge(ui + (double)i, c, Nd, halfNd);
    };
    const bool t0 = test(b);
    const bool t1 = (b + 1 >= Np) ? true : test(b + 1);
    int n = t0 ? b : (t1 ? b + 1 : b + 2);
    if (!t0 && !t1) {                                   // never taken if the estimate is within a slot
        while (n < Np && !test(n)) ++n;
    }
    if (t0) {                                           // rare (the estimate sits on an integer)
        while (n > 0 && test(n - 1)) --n;
    }
    return n;
}

// The same n(c), decided by the estimate alone whenever that is safe.  With e = N c - u (systematic) the answer is
// ceil(e) unless e sits within eps of an integer; every rounding that separates the computed e from the exact
// comparison pos_i >= c is bounded in slot units for Np < 2^31: the product and the difference 2^-22 each, the two
// roundings inside pos_i together N * 2^-52 <= 2^-21 -- 2^-20 in total, and eps = 2^-18.  Stratified: with
// f = floor(N c), slot f - 1 is below c, slot f + 1 is not, and slot f is decided by u_f against frac(N c) when
// they are farther apart than eps.  About one weight in 10^5 takes the exact tests above.
constexpr double N_BOUNDARY_EPS = 0x1p-18;

template <bool STRATIFIED>
FK_HD int n_boundary_fast(double c, int Np, double Nd, double halfNd, double u_sys, const double *u_str)
{
    if (!(c > 0.0)) return 0;
    const double e = c * Nd - (STRATIFIED ? 0.0 : u_sys);
    // every position is below c, with a margin of one slot (stratified: fl(u + Np - 1) may round up to Np, so
    // N c in [Np, Np + 1) still goes through the exact tests)
    if (e >= (double)Np + (STRATIFIED ? 1.0 : 0.0)) return Np;
    const double fl = floor(e), fr = e - fl;            // fr exact
    bool sure = fr > N_BOUNDARY_EPS && fr < 1.0 - N_BOUNDARY_EPS && fl < (double)Np;
    int n = (int)fl + 1;                                // systematic: ceil(e)  (e > -1: n >= 0)
    if (STRATIFIED) {
        const double uf = u_str[fl < (double)Np ? (int)fl : 0];   // e > 0 here (c > 0, u = 0)
        const double gap = uf - fr;
        sure = sure && (gap > N_BOUNDARY_EPS || gap < -N_BOUNDARY_EPS);
        n = (int)fl + (gap > 0.0 ? 0 : 1);
    }
    if (!sure) n = n_boundary<STRATIFIED>(c, Np, Nd, halfNd, u_sys, u_str);
    return n;
}

}  // namespace fk
