// fk_ukf.hpp -- one predict + update step of the fused linear-model UKF and one backward step of its smoother (the arithmetic
// of ukf_kernels.hip, see there).  __host__ __device__: the kernel runs it per lane, tests/hostcheck runs the very
// same code on the host against the oracle.
//
//   fresh() returns a view {sm, Wm, Wc} of the shared model (rowF / rowQ / rowH / rowR policy) and the padded
//   sigma-point weights; the kernel hands out a new, optimiser-opaque view on every call so that the broadcast
//   LDS reads of the 2n+1 unrolled points are not hoisted and held, the host hands out the same plain arrays.
//   fresh(v) additionally orders the view behind the arithmetic that produced v (the opaque point takes v as an input):
//   a chain of views that depend on nothing but each other is emitted -- reads and all -- in front of the arithmetic.
#pragma once

#include "fk_math.hpp"
#include "fk_math_sym.hpp"

#if defined(__HIP_DEVICE_COMPILE__)
#define FK_OPAQUE(v) asm volatile("" : "+v"(v))
#else
#define FK_OPAQUE(v) ((void)0)
#endif

namespace fk {

// sqrt(d) and 1 / sqrt(d) together; 1 / d.  On the device: the v_rsq_f64 / v_rcp_f64 seeds (2^-24 relative, measured:
// tools/experiments/rsq_seed_accuracy.hip, profiles/r03/rsq_seed_accuracy.jsonl) refined by ONE Goldschmidt / Newton step:
// 4e-15 / 2e-15 relative, 7 / 3 instructions (the compiler's correctly-rounded sqrt followed by a correctly-rounded division
// is ~28, with range scaling the pivots of a covariance factor do not need; a second step reaches 1e-16 and costs four more
// instructions on the critical path of every column).  The factor feeds covariance sums without cancellation and a mean in
// which the +l_k and -l_k terms cancel to first order, so its error arrives unamplified: far inside the 1e-10 bar.  A
// non-positive or non-finite pivot yields NaN / inf like sqrt would; the caller's pivot test reports it (ST_NOT_PD).  On the
// host (tests/hostcheck): the plain operations.
FK_HD void sqrt_rsqrt(double d, double &s, double &inv)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const double y = __builtin_amdgcn_rsq(d);
    double g = d * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    s = fma(g, r, g);
    h = fma(h, r, h);
    inv = h + h;
#else
    s = sqrt(d);
    inv = 1.0 / s;
#endif
}

FK_HD double rcp_refined(double d)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const double r = __builtin_amdgcn_rcp(d);
    const double e = fma(-d, r, 1.0);
    return fma(r, e, r);
#else
    return 1.0 / d;
#endif
}

// chol_packed (fk_math_sym.hpp) with the pivot's root and reciprocal from sqrt_rsqrt
template <int NX>
FK_HD bool chol_packed_rs(const double (&P)[NX * (NX + 1) / 2], double scale, double (&L)[NX * (NX + 1) / 2])
{
    bool pd = true;
    FK_UNROLL for (int j = 0; j < NX; ++j) {
        double d = scale * P[sym_idx<NX>(j, j)];
        FK_UNROLL for (int k = 0; k < NX; ++k)
            if (k < j) d = fma(-L[sym_idx<NX>(j, k)], L[sym_idx<NX>(j, k)], d);
        pd = pd && (d > 0.0);
        double ljj, inv;
        sqrt_rsqrt(d, ljj, inv);
        L[sym_idx<NX>(j, j)] = ljj;
        FK_UNROLL for (int i = 0; i < NX; ++i)
            if (i > j) {
                double t = scale * P[sym_idx<NX>(j, i)];
                FK_UNROLL for (int k = 0; k < NX; ++k)
                    if (k < j) t = fma(-L[sym_idx<NX>(i, k)], L[sym_idx<NX>(j, k)], t);
                L[sym_idx<NX>(i, j)] = t * inv;
            }
    }
    return pd;
}

// ldlt_packed (fk_math_sym.hpp) with the pivots' reciprocals from rcp_refined
template <int N>
FK_HD bool ldlt_packed_rs(double (&A)[N * (N + 1) / 2], double (&d)[N], double (&dinv)[N])
{
    bool pd = true;
    FK_UNROLL for (int j = 0; j < N; ++j) {
        double dj = A[sym_idx<N>(j, j)];
        FK_UNROLL for (int k = 0; k < N; ++k)
            if (k < j) {
                const double l = A[sym_idx<N>(j, k)];
                dj = fma(-l * l, d[k], dj);
            }
        pd = pd && (dj > 0.0);
        d[j] = dj;
        const double di = rcp_refined(dj);
        dinv[j] = di;
        FK_UNROLL for (int i = 0; i < N; ++i)
            if (i > j) {
                double t = A[sym_idx<N>(i, j)];
                FK_UNROLL for (int k = 0; k < N; ++k)
                    if (k < j) t = fma(-(A[sym_idx<N>(i, k)] * d[k]), A[sym_idx<N>(j, k)], t);
                A[sym_idx<N>(i, j)] = t * di;
            }
    }
    return pd;
}

// ldlt2 (fk_math.hpp) with the pivots' reciprocals from rcp_refined
template <int M>
FK_HD bool ldlt2_rs(double (&A)[M * M], double (&d)[M], double (&dinv)[M])
{
    bool pd = true;
    FK_UNROLL for (int j = 0; j < M; ++j) {
        double dj = A[j * M + j];
        FK_UNROLL for (int k = 0; k < j; ++k) {
            const double l = A[j * M + k];
            dj = fma(-l * l, d[k], dj);
        }
        pd = pd && (dj > 0.0);
        d[j] = dj;
        const double di = rcp_refined(dj);
        dinv[j] = di;
        FK_UNROLL for (int i = j + 1; i < M; ++i) {
            double s = A[i * M + j];
            FK_UNROLL for (int k = 0; k < j; ++k)
                s = fma(-(A[i * M + k] * d[k]), A[j * M + k], s);
            A[i * M + j] = s * di;
        }
    }
    return pd;
}

// One predict + update step (UKF.py:400-411, 462-481).  "V3" (round 3; round 2's V2 pushed every sigma point through F and
// H one by one, twice): the same sums over the same 2n+1 points in the same index order, but the images of the points are
// formed from the image of the FACTOR instead of point by point.  With fx(x) = F x the sigma-point matrix
// [x, x + l_k, x - l_k] (sigma_points.py:167-175) maps to [F x, F x + F l_k, F x - F l_k]: one pass over the rows of F
// produces F x and F L (column k of L has n - k non-zeros: n (n+1)/2 * n FMAs instead of (2n+1) n^2 per sweep, and V2
// made two sweeps), and every point's image is then ONE add per component in each of the two unscented-transform
// sweeps (mean, covariance -- unscented_transform.py:105-124 sums over all 2n+1 images, and so does this).  The same for
// hx(x) = H x, and the cross variance (UKF.py:483-497) takes the points' offsets sigma_i - x = +-l_k from the factor
// directly: the reference's (x + l) - x differs from l by one rounding of x, and the entries of l_k above the
// diagonal are exact zeros in both, so those terms are skipped, not approximated.
// Numerically each image differs from fl(F fl(x + l_k)) by a rounding of |F x| -- the same size as the reference's own
// rounding of that dot product, amplified by the same Merwe weights -- so the parity bar is the package's 1e-10 (held
// on the host against the oracle by tests/test_hostcheck_ukf.py, on the GPU against the live-reference goldens).
// (6,3): ~2000 VALU instructions per step against V2's 3780.
// load_z(z) delivers the step's measurement; it is called at the head of the update half -- the kernel issues the loads
// there (behind the predict half: no registers held across it, and the update's own arithmetic hides the latency).
template <int NX, int NZ, class LoadZ, class Fresh>
FK_HD int ukf_linear_step_v3(double (&x)[NX], double (&P)[NX * (NX + 1) / 2], LoadZ &&load_z, bool has_z,
                             double scale, Fresh &&fresh)
{
    constexpr int PL = NX * (NX + 1) / 2;
    int st = 0;
    // ---------------- predict (UKF.py:400-411)
    {
        double Fx[NX], FL[NX][NX];
        {
            double L[PL];
            if (!chol_packed_rs<NX>(P, scale, L)) st |= ST_NOT_PD;
            // one pass over the rows of F: F x and F L (FL[r][k] = sum_{c >= k} F[r][c] L[c][k])
            const auto mv = fresh();
            FK_UNROLL for (int r = 0; r < NX; ++r) {
                double f[NX];
                mv.sm.rowF(r, f);
                Fx[r] = dot<NX>(f, x);
                FK_UNROLL for (int k = 0; k < NX; ++k) {
                    double acc = f[k] * L[sym_idx<NX>(k, k)];
                    FK_UNROLL for (int c = 0; c < NX; ++c)
                        if (c > k) acc = fma(f[c], L[sym_idx<NX>(c, k)], acc);
                    FL[r][k] = acc;
                }
                FK_STAGE();
            }
        }
        // sweep 1: x- = sum_i Wm_i sf_i, points in index order 0, +k (k = 0..n-1), -k
        {
            const auto mv = fresh();
            const double *sWm = mv.Wm;
            FK_UNROLL for (int r = 0; r < NX; ++r) x[r] = sWm[0] * Fx[r];
            FK_UNROLL for (int k = 0; k < NX; ++k)
                FK_UNROLL for (int r = 0; r < NX; ++r) x[r] = fma(sWm[1 + k], Fx[r] + FL[r][k], x[r]);
            FK_UNROLL for (int k = 0; k < NX; ++k)
                FK_UNROLL for (int r = 0; r < NX; ++r) x[r] = fma(sWm[1 + NX + k], Fx[r] - FL[r][k], x[r]);
            FK_STAGE();
        }
        // sweep 2: P- = sum_i Wc_i y_i y_i' + Q, y_i = sf_i - x-   (upper triangle); the images are re-formed (one add)
        // from copies the optimiser cannot relate to sweep 1's, or it would hold all (2n+1) n of them
        // y_i = (F x - x-) +- F l_k: the offset of the centre point is formed once and the offsets of the others are one add
        // each (the reference rounds F(x + l_k) to an ulp of |F x| before it subtracts the mean; this keeps the digits of F l_k)
        FK_UNROLL for (int r = 0; r < NX; ++r) Fx[r] -= x[r];
        FK_UNROLL for (int r = 0; r < NX; ++r) FK_OPAQUE(Fx[r]);
        {
            FK_UNROLL for (int i = 0; i < 2 * NX + 1; ++i) {
                // a fresh view per point, ordered behind the previous point's arithmetic: read here, not hoisted and held
                const double wc = fresh(i ? P[PL - 1] : x[NX - 1]).Wc[i];
                double y[NX], wy[NX];
                FK_UNROLL for (int r = 0; r < NX; ++r)
                    y[r] = (i == 0) ? Fx[r] : (i <= NX) ? Fx[r] + FL[r][(i - 1) % NX] : Fx[r] - FL[r][(i - 1) % NX];
                FK_UNROLL for (int r = 0; r < NX; ++r) wy[r] = wc * y[r];
                FK_UNROLL for (int a2 = 0; a2 < NX; ++a2)
                    FK_UNROLL for (int b = 0; b < NX; ++b)
                        if (b >= a2)
                            P[sym_idx<NX>(a2, b)] = (i == 0) ? y[a2] * wy[b] : fma(y[a2], wy[b], P[sym_idx<NX>(a2, b)]);
                FK_STAGE();
            }
            FK_UNROLL for (int e = 0; e < PL; ++e) FK_OPAQUE(P[e]);     // every sum complete before the rows of Q are read
            const auto mv = fresh(P[PL - 1]);
            FK_UNROLL for (int r = 0; r < NX; ++r) {
                double q[NX];
                mv.sm.rowQ(r, q);
                FK_UNROLL for (int b = 0; b < NX; ++b)
                    if (b >= r) P[sym_idx<NX>(r, b)] += q[b];
            }
        }
    }
    // ---------------- update (UKF.py:462-481), sigma points regenerated from the prior (:407)
    if (has_z) {
        double z[NZ];
        load_z(z);
        double L[PL];
        if (!chol_packed_rs<NX>(P, scale, L)) st |= ST_NOT_PD;
        double Hx[NZ], HL[NZ][NX];
        {
            const auto mv = fresh();
            FK_UNROLL for (int r = 0; r < NZ; ++r) {
                double h[NX];
                mv.sm.rowH(r, h);
                Hx[r] = dot<NX>(h, x);
                FK_UNROLL for (int k = 0; k < NX; ++k) {
                    double acc = h[k] * L[sym_idx<NX>(k, k)];
                    FK_UNROLL for (int c = 0; c < NX; ++c)
                        if (c > k) acc = fma(h[c], L[sym_idx<NX>(c, k)], acc);
                    HL[r][k] = acc;
                }
            }
            FK_STAGE();
        }
        // sweep 1: zp = sum_i Wm_i sh_i
        double zp[NZ];
        {
            const auto mv = fresh();
            const double *sWm = mv.Wm;
            FK_UNROLL for (int r = 0; r < NZ; ++r) zp[r] = sWm[0] * Hx[r];
            FK_UNROLL for (int k = 0; k < NX; ++k)
                FK_UNROLL for (int r = 0; r < NZ; ++r) zp[r] = fma(sWm[1 + k], Hx[r] + HL[r][k], zp[r]);
            FK_UNROLL for (int k = 0; k < NX; ++k)
                FK_UNROLL for (int r = 0; r < NZ; ++r) zp[r] = fma(sWm[1 + NX + k], Hx[r] - HL[r][k], zp[r]);
        }
        // sweep 2: S = sum Wc_i d_i d_i' + R,  Pxz = sum Wc_i (sf_i - x) d_i',  d_i = sh_i - zp ; sf_0 - x = 0, sf_i - x = +-l_k
        double S[NZ * NZ], K[NX * NZ];
        FK_UNROLL for (int r = 0; r < NZ; ++r) Hx[r] -= zp[r];                 // d_i = (H x - zp) +- H l_k, as above
        FK_UNROLL for (int r = 0; r < NZ; ++r) FK_OPAQUE(Hx[r]);
        {
            FK_UNROLL for (int i = 0; i < 2 * NX + 1; ++i) {
                const double wc = fresh(i ? S[NZ * NZ - 1] : zp[NZ - 1]).Wc[i];
                double d[NZ], wd[NZ];
                FK_UNROLL for (int r = 0; r < NZ; ++r)
                    d[r] = (i == 0) ? Hx[r] : (i <= NX) ? Hx[r] + HL[r][(i - 1) % NX] : Hx[r] - HL[r][(i - 1) % NX];
                FK_UNROLL for (int r = 0; r < NZ; ++r) wd[r] = wc * d[r];
                // upper triangle; the lower one is its mirror image (the reference's w * outer(d, d) is symmetric bit for bit)
                FK_UNROLL for (int r = 0; r < NZ; ++r)
                    FK_UNROLL for (int c = 0; c < NZ; ++c)
                        if (c >= r) S[r * NZ + c] = (i == 0) ? d[r] * wd[c] : fma(d[r], wd[c], S[r * NZ + c]);
                if (i >= 1) {
                    const int k = (i - 1) % NX;
                    FK_UNROLL for (int r = 0; r < NX; ++r) {
                        if (r < k) continue;                                   // l_k is zero above the diagonal
                        const double lv = (i <= NX) ? L[sym_idx<NX>(r, k)] : -L[sym_idx<NX>(r, k)];
                        FK_UNROLL for (int c = 0; c < NZ; ++c)
                            K[r * NZ + c] = (i == 1) ? lv * wd[c] : fma(lv, wd[c], K[r * NZ + c]);
                    }
                }
                FK_STAGE();
            }
            FK_UNROLL for (int r = 0; r < NZ; ++r)
                FK_UNROLL for (int c = 0; c < NZ; ++c)
                    if (c < r) S[r * NZ + c] = S[c * NZ + r];
            FK_UNROLL for (int e = 0; e < NZ * NZ; ++e) FK_OPAQUE(S[e]);
            FK_UNROLL for (int e = 0; e < NX * NZ; ++e) FK_OPAQUE(K[e]);
            const auto mv = fresh(S[NZ * NZ - 1]);
            FK_UNROLL for (int r = 0; r < NZ; ++r) {
                double rr[NZ];
                mv.sm.rowR(r, rr);
                FK_UNROLL for (int c = 0; c < NZ; ++c) S[r * NZ + c] += rr[c];
            }
        }
        // K = Pxz S^-1
        double Lf[NZ * NZ], dd[NZ], dinv[NZ];
        FK_UNROLL for (int e = 0; e < NZ * NZ; ++e) Lf[e] = S[e];
        if (!ldlt2_rs<NZ>(Lf, dd, dinv)) st |= ST_NOT_PD;
        solve_rows_ldlt<NX, NZ>(Lf, dinv, K);
        // x += K (z - zp)
        FK_UNROLL for (int r = 0; r < NX; ++r) {
            double acc = K[r * NZ] * (z[0] - zp[0]);
            FK_UNROLL for (int c = 1; c < NZ; ++c) acc = fma(K[r * NZ + c], z[c] - zp[c], acc);
            x[r] += acc;
        }
        // P -= K (S K'), upper triangle
        FK_UNROLL for (int c2 = 0; c2 < NX; ++c2) {
            double sk[NZ];                 // column c2 of S K'
            FK_UNROLL for (int q = 0; q < NZ; ++q) {
                double acc = S[q * NZ] * K[c2 * NZ];
                FK_UNROLL for (int w = 1; w < NZ; ++w) acc = fma(S[q * NZ + w], K[c2 * NZ + w], acc);
                sk[q] = acc;
            }
            FK_UNROLL for (int r = 0; r < NX; ++r)
                if (r <= c2) {
                    double acc = K[r * NZ] * sk[0];
                    FK_UNROLL for (int q = 1; q < NZ; ++q) acc = fma(K[r * NZ + q], sk[q], acc);
                    P[sym_idx<NX>(r, c2)] -= acc;
                }
            FK_STAGE();
        }
    }
    return st;
}

// The pair table of the "V4" steps below: for sigma-point weights that are equal within every +- pair
// (Wm[1+k] == Wm[1+n+k], Wc[1+k] == Wc[1+n+k] -- true of MerweScaledSigmaPoints and JulierSigmaPoints,
// sigma_points.py:180-192, :358-372) the sums of unscented_transform.py:104-126 and UKF.py:483-497 over the 2n+1 points
// regroup over the n pairs:  Wp = [ sum_i Wm_i , sum_i Wc_i , Wc[1+k] + Wc[1+n+k] (k = 0..NX-1) ]  (sums in index order;
// padded pairs carry 0).  make_pair_table fills it from the PADDED weight arrays (index 0, 1..NX, NX+1..2NX).
template <int NX>
FK_HD void make_pair_table(const double *Wm, const double *Wc, double *Wp)
{
    double sm = Wm[0], sc = Wc[0];
    for (int i = 1; i < 2 * NX + 1; ++i) {
        sm += Wm[i];
        sc += Wc[i];
    }
    Wp[0] = sm;
    Wp[1] = sc;
    for (int k = 0; k < NX; ++k) Wp[2 + k] = Wc[1 + k] + Wc[1 + NX + k];
}

template <int NX>
FK_HD bool pair_weights_symmetric(const double *Wm, const double *Wc)
{
    bool ok = true;
    for (int k = 0; k < NX; ++k) ok = ok && (Wm[1 + k] == Wm[1 + NX + k]) && (Wc[1 + k] == Wc[1 + NX + k]);
    return ok;
}

// One predict + update step, "V4" (round 4): ukf_linear_step_v3 with the sums regrouped over the +- PAIRS of sigma points.
// The factor of scale * P, its image F L, the image of the centre point and every weighted sum of
// unscented_transform.py:104-126 / UKF.py:483-497 are still formed -- what changes is the ASSOCIATION of the sums.  With the
// images  sf_0 = F x,  sf_{+-k} = F x +- f_k  (f_k = F l_k, column k of F L)  and weights equal within a pair:
//     mean        sum_i Wm_i sf_i            = (sum_i Wm_i) F x                        [ + sum_k (Wm+ - Wm-) f_k = 0 ]
//     offsets     y_0 = F x - mean,  y_{+-k} = y_0 +- f_k
//     covariance  sum_i Wc_i y_i y_i'        = (sum_i Wc_i) y_0 y_0' + sum_k (Wc+ + Wc-) f_k f_k'
//                                                                              [ + sum_k (Wc+ - Wc-) (y_0 f_k' + f_k y_0') = 0 ]
//     cross       sum_i Wc_i (s_i - x) d_i'  = sum_k (Wc+ + Wc-) l_k h_k'   (s_{+-k} - x = +-l_k, d_{+-k} = d_0 +- h_k, h_k = H l_k)
// i.e. n + 1 rank-one terms per covariance instead of 2n + 1 and a mean without its 2n cancelling additions.  Against the
// reference's index-order sums this is a re-association (the reference's own result moves by as much when ITS sum is
// re-ordered: tests/golden/make_conditioning.py measures that spread); it is NOT the linear filter's F P F' + Q -- the step
// still factors scale * P twice, regenerates the points of the prior for the update (UKF.py:407) and applies the weights.
// (6,3): ~1200 VALU instructions per step against V3's 1801.  Callers assert the symmetry (FK_UKF_FLAG_PAIR_WEIGHTS); the
// index-order step remains for every other weight set and as the A/B switch (FK_UKF_PAIRED=0).
// fresh() views carry Wp (make_pair_table) next to Wm / Wc.
// Operand traffic (round 4): the shared model and the pair table sit in LDS, and a wave that asks for a row, waits, uses it
// and asks for the next pays an LDS round trip per row -- 79 waits per step, about as long as the step's arithmetic on a SIMD
// that holds one or two waves.  Here every operand of a half-step is REQUESTED AT ITS HEAD, in front of the factorisation
// (which needs none of them): rows of F and the pair table before the predict's Cholesky, the rows of Q once the rows of F
// are spent, rows of H / R and z before the update's Cholesky.  The round trips then run under ~130 instructions of
// arithmetic each instead of under nothing; the price is registers (F: 2 n^2) that the factorisation leaves free anyway.
template <int NX, int NZ, class LoadZ, class Fresh>
FK_HD int ukf_linear_step_v4(double (&x)[NX], double (&P)[NX * (NX + 1) / 2], LoadZ &&load_z, bool has_z,
                             double scale, Fresh &&fresh)
{
    constexpr int PL = NX * (NX + 1) / 2;
    int st = 0;
    // ---------------- predict (UKF.py:400-411)
    {
        double Fm[NX][NX], wt[2 + NX];
        {
            const auto mv = fresh();
            FK_UNROLL for (int r = 0; r < NX; ++r) mv.sm.rowF(r, Fm[r]);
            FK_UNROLL for (int k = 0; k < 2 + NX; ++k) wt[k] = mv.Wp[k];
        }
        FK_STAGE();
        double Fx[NX], FL[NX][NX];
        {
            double L[PL];
            if (!chol_packed_rs<NX>(P, scale, L)) st |= ST_NOT_PD;
            FK_STAGE();
            // one pass over the rows of F: F x and F L (FL[r][k] = sum_{c >= k} F[r][c] L[c][k])
            FK_UNROLL for (int r = 0; r < NX; ++r) {
                Fx[r] = dot<NX>(Fm[r], x);
                FK_UNROLL for (int k = 0; k < NX; ++k) {
                    double acc = Fm[r][k] * L[sym_idx<NX>(k, k)];
                    FK_UNROLL for (int c = 0; c < NX; ++c)
                        if (c > k) acc = fma(Fm[r][c], L[sym_idx<NX>(c, k)], acc);
                    FL[r][k] = acc;
                }
                FK_STAGE();
            }
        }
        // the rows of Q (upper triangle), requested now -- the registers of F are free -- and used after the last pair
        double Qm[PL];
        {
            const auto mv = fresh(FL[NX - 1][NX - 1]);
            FK_UNROLL for (int r = 0; r < NX; ++r) {
                double q[NX];
                mv.sm.rowQ(r, q);
                FK_UNROLL for (int b = 0; b < NX; ++b)
                    if (b >= r) Qm[sym_idx<NX>(r, b)] = q[b];
            }
        }
        FK_STAGE();
        // mean and the centre point's offset
        {
            const double wms = wt[0], wcs = wt[1];
            double wy[NX];
            FK_UNROLL for (int r = 0; r < NX; ++r) {
                x[r] = wms * Fx[r];
                Fx[r] -= x[r];                                             // y_0
            }
            FK_UNROLL for (int r = 0; r < NX; ++r) wy[r] = wcs * Fx[r];
            FK_UNROLL for (int a2 = 0; a2 < NX; ++a2)
                FK_UNROLL for (int b = 0; b < NX; ++b)
                    if (b >= a2) P[sym_idx<NX>(a2, b)] = Fx[a2] * wy[b];
            FK_STAGE();
        }
        // the pairs
        FK_UNROLL for (int k = 0; k < NX; ++k) {
            const double wp = wt[2 + k];
            double wf[NX];
            FK_UNROLL for (int r = 0; r < NX; ++r) wf[r] = wp * FL[r][k];
            FK_UNROLL for (int a2 = 0; a2 < NX; ++a2)
                FK_UNROLL for (int b = 0; b < NX; ++b)
                    if (b >= a2) P[sym_idx<NX>(a2, b)] = fma(FL[a2][k], wf[b], P[sym_idx<NX>(a2, b)]);
            FK_STAGE();
        }
        FK_UNROLL for (int e = 0; e < PL; ++e) P[e] += Qm[e];               // + Q last, like the reference
    }
    // ---------------- update (UKF.py:462-481), sigma points regenerated from the prior (:407)
    if (has_z) {
        double z[NZ];
        load_z(z);
        double Hm[NZ][NX], Rm[NZ * NZ], wt[2 + NX];
        {
            const auto mv = fresh(P[PL - 1]);
            FK_UNROLL for (int r = 0; r < NZ; ++r) mv.sm.rowH(r, Hm[r]);
            FK_UNROLL for (int r = 0; r < NZ; ++r) {
                double rr[NZ];
                mv.sm.rowR(r, rr);
                FK_UNROLL for (int c = 0; c < NZ; ++c) Rm[r * NZ + c] = rr[c];
            }
            FK_UNROLL for (int k = 0; k < 2 + NX; ++k) wt[k] = mv.Wp[k];
        }
        FK_STAGE();
        double L[PL];
        if (!chol_packed_rs<NX>(P, scale, L)) st |= ST_NOT_PD;
        FK_STAGE();
        double Hx[NZ], HL[NZ][NX];
        FK_UNROLL for (int r = 0; r < NZ; ++r) {
            Hx[r] = dot<NX>(Hm[r], x);
            FK_UNROLL for (int k = 0; k < NX; ++k) {
                double acc = Hm[r][k] * L[sym_idx<NX>(k, k)];
                FK_UNROLL for (int c = 0; c < NX; ++c)
                    if (c > k) acc = fma(Hm[r][c], L[sym_idx<NX>(c, k)], acc);
                HL[r][k] = acc;
            }
        }
        FK_STAGE();
        double zp[NZ], S[NZ * NZ], K[NX * NZ];
        {
            const double wms = wt[0], wcs = wt[1];
            double wd[NZ];
            FK_UNROLL for (int r = 0; r < NZ; ++r) {
                zp[r] = wms * Hx[r];
                Hx[r] -= zp[r];                                            // d_0
            }
            FK_UNROLL for (int r = 0; r < NZ; ++r) wd[r] = wcs * Hx[r];
            FK_UNROLL for (int r = 0; r < NZ; ++r)
                FK_UNROLL for (int c = 0; c < NZ; ++c)
                    if (c >= r) S[r * NZ + c] = Hx[r] * wd[c];
        }
        FK_UNROLL for (int k = 0; k < NX; ++k) {
            const double wp = wt[2 + k];
            double wh[NZ];
            FK_UNROLL for (int r = 0; r < NZ; ++r) wh[r] = wp * HL[r][k];
            FK_UNROLL for (int r = 0; r < NZ; ++r)
                FK_UNROLL for (int c = 0; c < NZ; ++c)
                    if (c >= r) S[r * NZ + c] = fma(HL[r][k], wh[c], S[r * NZ + c]);
            FK_UNROLL for (int r = 0; r < NX; ++r) {
                if (r < k) continue;                                       // l_k is zero above the diagonal
                FK_UNROLL for (int c = 0; c < NZ; ++c)
                    K[r * NZ + c] = (k == 0) ? L[sym_idx<NX>(r, k)] * wh[c] : fma(L[sym_idx<NX>(r, k)], wh[c], K[r * NZ + c]);
            }
            FK_STAGE();
        }
        FK_UNROLL for (int r = 0; r < NZ; ++r)
            FK_UNROLL for (int c = 0; c < NZ; ++c)
                if (c < r) S[r * NZ + c] = S[c * NZ + r];
        FK_UNROLL for (int e = 0; e < NZ * NZ; ++e) S[e] += Rm[e];          // + R last
        // K = Pxz S^-1
        double Lf[NZ * NZ], dd[NZ], dinv[NZ];
        FK_UNROLL for (int e = 0; e < NZ * NZ; ++e) Lf[e] = S[e];
        if (!ldlt2_rs<NZ>(Lf, dd, dinv)) st |= ST_NOT_PD;
        solve_rows_ldlt<NX, NZ>(Lf, dinv, K);
        // x += K (z - zp)
        FK_UNROLL for (int c = 0; c < NZ; ++c) z[c] -= zp[c];
        FK_UNROLL for (int r = 0; r < NX; ++r) {
            double acc = K[r * NZ] * z[0];
            FK_UNROLL for (int c = 1; c < NZ; ++c) acc = fma(K[r * NZ + c], z[c], acc);
            x[r] += acc;
        }
        // P -= K (S K'), upper triangle
        FK_UNROLL for (int c2 = 0; c2 < NX; ++c2) {
            double sk[NZ];                 // column c2 of S K'
            FK_UNROLL for (int q = 0; q < NZ; ++q) {
                double acc = S[q * NZ] * K[c2 * NZ];
                FK_UNROLL for (int w = 1; w < NZ; ++w) acc = fma(S[q * NZ + w], K[c2 * NZ + w], acc);
                sk[q] = acc;
            }
            FK_UNROLL for (int r = 0; r < NX; ++r)
                if (r <= c2) {
                    double acc = K[r * NZ] * sk[0];
                    FK_UNROLL for (int q = 1; q < NZ; ++q) acc = fma(K[r * NZ + q], sk[q], acc);
                    P[sym_idx<NX>(r, c2)] -= acc;
                }
            FK_STAGE();
        }
    }
    return st;
}

// One backward step of UnscentedKalmanFilter.rts_smoother with fx(x, dt) = F x (UKF.py:714-737), fused:
//   sigmas = sigma_points(xs[k], ps[k]) ; sigmas_f = F sigmas ; (xb, Pb) = UT(sigmas_f, Wm, Wc, Q)
//   Pxb = sum_i Wc_i (sigmas_i - Xs[k]) (sigmas_f_i - xb)' ; K = Pxb inv(Pb)
//   xs[k] += K (xs[k+1] - xb) ; ps[k] += K (ps[k+1] - Pb) K'
// x / P (packed upper triangle): the filter output of step k in (xs[k] = Xs[k] until this step touches it), the
// smoothed step k out; xn / Pn: the smoothed step k+1.  K: full n x n gain out.  The sums run over the sigma points
// in the reference's index order.  inv(Pb) is applied by an L D L' solve (like every other gain here).
// Two halves, so that the caller may keep xn / Pn out of the registers during the first:
//   ukf_linear_rts_gain_v3 (x, P) -> xb, Pb (packed), K            (the two sweeps and the solve)
//   ukf_linear_rts_correct x, P updated in place from xn, Pn, xb, Pb, K   (Pb is destroyed)
// The gain in the factor-image organisation (see ukf_linear_step_v3): one pass over the rows of F gives F x and
// F L, every point's image is one add per component in each sweep, and the offsets sigma_i - x = +-l_k of the cross
// variance come from the factor (zero above the diagonal: those terms are skipped).
template <int NX, class Fresh>
FK_HD int ukf_linear_rts_gain_v3(double (&x)[NX], const double (&P)[NX * (NX + 1) / 2], double scale, double (&xb)[NX],
                                 double (&Pb)[NX * (NX + 1) / 2], double (&K)[NX * NX], Fresh &&fresh)
{
    constexpr int PL = NX * (NX + 1) / 2;
    int st = 0;
    double L[PL];
    if (!chol_packed_rs<NX>(P, scale, L)) st |= ST_NOT_PD;
    double Fx[NX], FL[NX][NX];
    {
        const auto mv = fresh();
        FK_UNROLL for (int r = 0; r < NX; ++r) {
            double f[NX];
            mv.sm.rowF(r, f);
            Fx[r] = dot<NX>(f, x);
            FK_UNROLL for (int k = 0; k < NX; ++k) {
                double acc = f[k] * L[sym_idx<NX>(k, k)];
                FK_UNROLL for (int c = 0; c < NX; ++c)
                    if (c > k) acc = fma(f[c], L[sym_idx<NX>(c, k)], acc);
                FL[r][k] = acc;
            }
            FK_STAGE();
        }
    }
    // sweep 1: xb = sum_i Wm_i sf_i
    {
        const auto mv = fresh();
        const double *sWm = mv.Wm;
        FK_UNROLL for (int r = 0; r < NX; ++r) xb[r] = sWm[0] * Fx[r];
        FK_UNROLL for (int k = 0; k < NX; ++k)
            FK_UNROLL for (int r = 0; r < NX; ++r) xb[r] = fma(sWm[1 + k], Fx[r] + FL[r][k], xb[r]);
        FK_UNROLL for (int k = 0; k < NX; ++k)
            FK_UNROLL for (int r = 0; r < NX; ++r) xb[r] = fma(sWm[1 + NX + k], Fx[r] - FL[r][k], xb[r]);
        FK_STAGE();
    }
    // sweep 2: Pb = sum Wc_i y_i y_i' (+ Q), Pxb = sum Wc_i (sigma_i - x) y_i', y_i = sf_i - xb
    FK_UNROLL for (int r = 0; r < NX; ++r) Fx[r] -= xb[r];                     // y_i = (F x - xb) +- F l_k
    FK_UNROLL for (int r = 0; r < NX; ++r) FK_OPAQUE(Fx[r]);
    {
        FK_UNROLL for (int i = 0; i < 2 * NX + 1; ++i) {
            const double wc = fresh(i ? Pb[PL - 1] : xb[NX - 1]).Wc[i];   // a fresh view per point, behind the previous point
            double y[NX], wy[NX];
            FK_UNROLL for (int r = 0; r < NX; ++r)
                y[r] = (i == 0) ? Fx[r] : (i <= NX) ? Fx[r] + FL[r][(i - 1) % NX] : Fx[r] - FL[r][(i - 1) % NX];
            FK_UNROLL for (int r = 0; r < NX; ++r) wy[r] = wc * y[r];
            FK_UNROLL for (int a2 = 0; a2 < NX; ++a2)
                FK_UNROLL for (int b = 0; b < NX; ++b)
                    if (b >= a2)
                        Pb[sym_idx<NX>(a2, b)] = (i == 0) ? y[a2] * wy[b] : fma(y[a2], wy[b], Pb[sym_idx<NX>(a2, b)]);
            if (i >= 1) {
                const int k = (i - 1) % NX;
                FK_UNROLL for (int r = 0; r < NX; ++r) {
                    if (r < k) continue;                                       // l_k is zero above the diagonal
                    const double lv = (i <= NX) ? L[sym_idx<NX>(r, k)] : -L[sym_idx<NX>(r, k)];
                    // row r receives its first term from point 1 (k = 0): every row r >= 0 is below or on column 0's diagonal
                    FK_UNROLL for (int c = 0; c < NX; ++c)
                        K[r * NX + c] = (i == 1) ? lv * wy[c] : fma(lv, wy[c], K[r * NX + c]);
                }
            }
            FK_STAGE();
        }
        FK_UNROLL for (int e = 0; e < PL; ++e) FK_OPAQUE(Pb[e]);
        const auto mv = fresh(Pb[PL - 1]);
        FK_UNROLL for (int r = 0; r < NX; ++r) {
            double q[NX];
            mv.sm.rowQ(r, q);
            FK_UNROLL for (int b = 0; b < NX; ++b)
                if (b >= r) Pb[sym_idx<NX>(r, b)] += q[b];
        }
    }
    // K = Pxb inv(Pb)
    {
        double Lp[PL], d[NX], dinv[NX];
        FK_UNROLL for (int e = 0; e < PL; ++e) Lp[e] = Pb[e];
        if (!ldlt_packed_rs<NX>(Lp, d, dinv)) st |= ST_NOT_PD;
        FK_UNROLL for (int r = 0; r < NX; ++r) {
            double row[NX];
            FK_UNROLL for (int c = 0; c < NX; ++c) row[c] = K[r * NX + c];
            solve_row_packed<NX>(Lp, dinv, row);
            FK_UNROLL for (int c = 0; c < NX; ++c) K[r * NX + c] = row[c];
        }
    }
    FK_STAGE();
    return st;
}

// ukf_linear_rts_gain_v3 with the sums regrouped over the +- pairs (see ukf_linear_step_v4; weights equal within a pair):
//   xb = (sum Wm) F x ;  Pb = (sum Wc) y_0 y_0' + sum_k (Wc+ + Wc-) f_k f_k' + Q ;  Pxb = sum_k (Wc+ + Wc-) l_k f_k'
template <int NX, class Fresh>
FK_HD int ukf_linear_rts_gain_v4(double (&x)[NX], const double (&P)[NX * (NX + 1) / 2], double scale, double (&xb)[NX],
                                 double (&Pb)[NX * (NX + 1) / 2], double (&K)[NX * NX], Fresh &&fresh)
{
    constexpr int PL = NX * (NX + 1) / 2;
    int st = 0;
    // operands requested at the head, in front of the factorisation (see ukf_linear_step_v4)
    double Fm[NX][NX], wt[2 + NX];
    {
        const auto mv = fresh();
        FK_UNROLL for (int r = 0; r < NX; ++r) mv.sm.rowF(r, Fm[r]);
        FK_UNROLL for (int k = 0; k < 2 + NX; ++k) wt[k] = mv.Wp[k];
    }
    FK_STAGE();
    double L[PL];
    if (!chol_packed_rs<NX>(P, scale, L)) st |= ST_NOT_PD;
    FK_STAGE();
    double Fx[NX], FL[NX][NX];
    FK_UNROLL for (int r = 0; r < NX; ++r) {
        Fx[r] = dot<NX>(Fm[r], x);
        FK_UNROLL for (int k = 0; k < NX; ++k) {
            double acc = Fm[r][k] * L[sym_idx<NX>(k, k)];
            FK_UNROLL for (int c = 0; c < NX; ++c)
                if (c > k) acc = fma(Fm[r][c], L[sym_idx<NX>(c, k)], acc);
            FL[r][k] = acc;
        }
        FK_STAGE();
    }
    double Qm[PL];
    {
        const auto mv = fresh(FL[NX - 1][NX - 1]);
        FK_UNROLL for (int r = 0; r < NX; ++r) {
            double q[NX];
            mv.sm.rowQ(r, q);
            FK_UNROLL for (int b = 0; b < NX; ++b)
                if (b >= r) Qm[sym_idx<NX>(r, b)] = q[b];
        }
    }
    FK_STAGE();
    {
        const double wms = wt[0], wcs = wt[1];
        double wy[NX];
        FK_UNROLL for (int r = 0; r < NX; ++r) {
            xb[r] = wms * Fx[r];
            Fx[r] -= xb[r];                                                // y_0
        }
        FK_UNROLL for (int r = 0; r < NX; ++r) wy[r] = wcs * Fx[r];
        FK_UNROLL for (int a2 = 0; a2 < NX; ++a2)
            FK_UNROLL for (int b = 0; b < NX; ++b)
                if (b >= a2) Pb[sym_idx<NX>(a2, b)] = Fx[a2] * wy[b];
        FK_STAGE();
    }
    FK_UNROLL for (int k = 0; k < NX; ++k) {
        const double wp = wt[2 + k];
        double wf[NX];
        FK_UNROLL for (int r = 0; r < NX; ++r) wf[r] = wp * FL[r][k];
        FK_UNROLL for (int a2 = 0; a2 < NX; ++a2)
            FK_UNROLL for (int b = 0; b < NX; ++b)
                if (b >= a2) Pb[sym_idx<NX>(a2, b)] = fma(FL[a2][k], wf[b], Pb[sym_idx<NX>(a2, b)]);
        FK_UNROLL for (int r = 0; r < NX; ++r) {
            if (r < k) continue;                                           // l_k is zero above the diagonal
            FK_UNROLL for (int c = 0; c < NX; ++c)
                K[r * NX + c] = (k == 0) ? L[sym_idx<NX>(r, k)] * wf[c] : fma(L[sym_idx<NX>(r, k)], wf[c], K[r * NX + c]);
        }
        FK_STAGE();
    }
    FK_UNROLL for (int e = 0; e < PL; ++e) Pb[e] += Qm[e];                  // + Q last
    // K = Pxb inv(Pb)
    {
        double Lp[PL], d[NX], dinv[NX];
        FK_UNROLL for (int e = 0; e < PL; ++e) Lp[e] = Pb[e];
        if (!ldlt_packed_rs<NX>(Lp, d, dinv)) st |= ST_NOT_PD;
        FK_UNROLL for (int r = 0; r < NX; ++r) {
            double row[NX];
            FK_UNROLL for (int c = 0; c < NX; ++c) row[c] = K[r * NX + c];
            solve_row_packed<NX>(Lp, dinv, row);
            FK_UNROLL for (int c = 0; c < NX; ++c) K[r * NX + c] = row[c];
        }
    }
    FK_STAGE();
    return st;
}

template <int NX>
FK_HD void ukf_linear_rts_correct(double (&x)[NX], double (&P)[NX * (NX + 1) / 2], const double (&xn)[NX],
                                  const double (&Pn)[NX * (NX + 1) / 2], const double (&xb)[NX],
                                  double (&Pb)[NX * (NX + 1) / 2], const double (&K)[NX * NX])
{
    constexpr int PL = NX * (NX + 1) / 2;
    // x += K (xn - xb)
    {
        double dx[NX];
        FK_UNROLL for (int c = 0; c < NX; ++c) dx[c] = xn[c] - xb[c];
        FK_UNROLL for (int r = 0; r < NX; ++r) {
            double acc = K[r * NX] * dx[0];
            FK_UNROLL for (int c = 1; c < NX; ++c) acc = fma(K[r * NX + c], dx[c], acc);
            x[r] += acc;
        }
    }
    // P += (K (Pn - Pb)) K', upper triangle, one row of K (Pn - Pb) at a time
    FK_UNROLL for (int e = 0; e < PL; ++e) Pb[e] = Pn[e] - Pb[e];            // D
    FK_UNROLL for (int i = 0; i < NX; ++i) {
        double t1[NX];
        FK_UNROLL for (int c = 0; c < NX; ++c) {
            double acc = K[i * NX] * Pb[sym_idx<NX>(0, c)];
            FK_UNROLL for (int k = 1; k < NX; ++k) acc = fma(K[i * NX + k], Pb[sym_idx<NX>(k, c)], acc);
            t1[c] = acc;
        }
        FK_UNROLL for (int j = 0; j < NX; ++j)
            if (j >= i) {
                double acc = t1[0] * K[j * NX];
                FK_UNROLL for (int k = 1; k < NX; ++k) acc = fma(t1[k], K[j * NX + k], acc);
                P[sym_idx<NX>(i, j)] += acc;
            }
        FK_STAGE();
    }
}

}  // namespace fk
