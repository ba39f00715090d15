// ut_kernels.hip -- sigma-point / unscented-transform kernels for gfx950 (MI355X).
//
// One track per lane.  Stand-alone kernels mirror the reference's building blocks so
// that a UKF with arbitrary (host-side, vectorised) fx/hx can be assembled from them:
//   fk_ut_sigma_points_f64   <- MerweScaledSigmaPoints.sigma_points (sigma_points.py:124-177)
//                               JulierSigmaPoints.sigma_points      (sigma_points.py:289-357)
//   fk_ut_transform_f64      <- unscented_transform (unscented_transform.py:99-128)
//   fk_ut_cross_variance_f64 <- UnscentedKalmanFilter.cross_variance (UKF.py:493-504)
// and the fused linear-model filter keeps the whole predict/update of UKF.py:364-481 in
// registers over the time loop:
//   fk_ukf_linear_batch_f64  <- UnscentedKalmanFilter.batch_filter (UKF.py:524-632), fx = F x, hx = H x.
// Algorithmic bytes: sigma points 8(n + n^2 + (2n+1)n), UT 8((2n+1)n + n + n^2) per track;
// fused step 8(m + n + n^2) per track-step.
#include "../../include/filterhip.h"
#include "fk_device.hpp"
#include "fk_kernel_args.hpp"

namespace fk {

// ------------------------------------------------------------ sigma points --
template <int NX, int LAYOUT>
__global__ void __launch_bounds__(BLOCK)
sigma_kernel(int n, long N, double scale, const double *__restrict__ px, const double *__restrict__ pP,
             double *sig, int32_t *status)
{
    const long blk0 = (long)blockIdx.x * BLOCK;
    const Lane ln{blk0, threadIdx.x, N};
    if (blk0 + ln.tid >= N) return;
    double x[NX], P[NX * NX], L[NX * NX];
    load_rec<NX, 1, LAYOUT, false>(x, px, ln, n, 1, 0.0);
    load_rec<NX, NX, LAYOUT, false>(P, pP, ln, n, n, 1.0 / scale);   // padded block -> L = I
    const bool pd = chol_lower<NX>(P, scale, L);
    const RecView<LAYOUT> out(sig, ln, (2 * n + 1) * n);
    // sigma_0 = x ; sigma_{k+1} = x + U[k] ; sigma_{n+k+1} = x - U[k]   (U[k][c] = L[c][k])
    FK_UNROLL for (int c = 0; c < NX; ++c)
        if (c < n) out.store(c, x[c]);
    FK_UNROLL for (int k = 0; k < NX; ++k) {
        if (k < n) {
            FK_UNROLL for (int c = 0; c < NX; ++c) {
                if (c < n) {
                    // np.subtract(x, -U[k]) and np.subtract(x, U[k])  (sigma_points.py:174-175)
                    out.store((k + 1) * n + c, x[c] - (-L[c * NX + k]));
                    out.store((n + k + 1) * n + c, x[c] - L[c * NX + k]);
                }
            }
        }
    }
    if (status) status[blk0 + ln.tid] = pd ? 0 : ST_NOT_PD;
}

// Register-resident variant for the standard point sets (k == 2n+1, n == NX <= 6): every sigma
// point is read from HBM exactly once (the generic kernel below re-reads them for the second pass).
template <int NX, int LAYOUT>
__global__ void __launch_bounds__(BLOCK, (NX <= 4 ? 4 : 1))
ut_reg_kernel(long N, const double *__restrict__ sig, const double *__restrict__ Wm,
              const double *__restrict__ Wc, const double *__restrict__ noise, double *xo, double *Po)
{
    constexpr int K = 2 * NX + 1;
    const long blk0 = (long)blockIdx.x * BLOCK;
    const Lane ln{blk0, threadIdx.x, N};
    if (blk0 + ln.tid >= N) return;
    double s[K * NX];
    load_rec<K, NX, LAYOUT, true>(s, sig, ln, K, NX, 0.0);
    double x[NX];
    FK_UNROLL for (int c = 0; c < NX; ++c) {
        double acc = Wm[0] * s[c];
        FK_UNROLL for (int i = 1; i < K; ++i) acc = fma(Wm[i], s[i * NX + c], acc);
        x[c] = acc;
    }
    FK_UNROLL for (int i = 0; i < K; ++i)
        FK_UNROLL for (int c = 0; c < NX; ++c) s[i * NX + c] -= x[c];
    store_rec<NX, 1, LAYOUT, true>(x, xo, ln, NX, 1);
    // P = sum_i y_i (Wc_i y_i)': accumulate the upper triangle point by point (P is symmetric by
    // construction: the same two factors commute), mirror on store
    double U[NX * (NX + 1) / 2];
    FK_UNROLL for (int i = 0; i < K; ++i) {
        double wy[NX];
        FK_UNROLL for (int c = 0; c < NX; ++c) wy[c] = Wc[i] * s[i * NX + c];
        int t = 0;
        FK_UNROLL for (int a = 0; a < NX; ++a)
            FK_UNROLL for (int b2 = a; b2 < NX; ++b2, ++t)
                U[t] = (i == 0) ? s[a] * wy[b2] : fma(s[i * NX + a], wy[b2], U[t]);
        FK_STAGE();
    }
    const RecView<LAYOUT> pv(Po, ln, NX * NX);
    int t = 0;
    FK_UNROLL for (int a = 0; a < NX; ++a)
        FK_UNROLL for (int b2 = a; b2 < NX; ++b2, ++t) {
            const double v = noise ? U[t] + noise[a * NX + b2] : U[t];
            pv.store(a * NX + b2, v);
            if (b2 != a) pv.store(b2 * NX + a, noise ? U[t] + noise[b2 * NX + a] : U[t]);
        }
}

// ------------------------------------------------------ unscented transform --
template <int NX, int LAYOUT>
__global__ void __launch_bounds__(BLOCK)
ut_kernel(int n, int k, long N, const double *__restrict__ sig, const double *__restrict__ Wm,
          const double *__restrict__ Wc, const double *__restrict__ noise, double *xo, double *Po)
{
    const long blk0 = (long)blockIdx.x * BLOCK;
    const Lane ln{blk0, threadIdx.x, N};
    if (blk0 + ln.tid >= N) return;
    const RecView<LAYOUT> sv(sig, ln, k * n);
    // x = Wm . sigmas   (unscented_transform.py:104)
    double x[NX];
    FK_UNROLL for (int c = 0; c < NX; ++c) x[c] = 0.0;
    for (int i = 0; i < k; ++i) {
        const double w = Wm[i];
        FK_UNROLL for (int c = 0; c < NX; ++c)
            if (c < n) x[c] = (i == 0) ? w * sv.load(c) : fma(w, sv.load(i * n + c), x[c]);
    }
    // P = y' (diag(Wc) y)   (unscented_transform.py:117-118)
    double P[NX * NX];
    FK_UNROLL for (int e = 0; e < NX * NX; ++e) P[e] = 0.0;
    for (int i = 0; i < k; ++i) {
        const double w = Wc[i];
        double y[NX];
        FK_UNROLL for (int c = 0; c < NX; ++c) y[c] = (c < n) ? sv.load(i * n + c) - x[c] : 0.0;
        FK_UNROLL for (int a = 0; a < NX; ++a)
            FK_UNROLL for (int b = 0; b < NX; ++b) P[a * NX + b] = fma(y[a], w * y[b], P[a * NX + b]);
    }
    if (noise) {
        FK_UNROLL for (int a = 0; a < NX; ++a)
            FK_UNROLL for (int b = 0; b < NX; ++b)
                if (a < n && b < n) P[a * NX + b] += noise[a * n + b];
    }
    store_rec<NX, 1, LAYOUT, false>(x, xo, ln, n, 1);
    store_rec<NX, NX, LAYOUT, false>(P, Po, ln, n, n);
}

// ---------------------------------------------------------- cross variance --
template <int NX, int LAYOUT>
__global__ void __launch_bounds__(BLOCK)
cross_kernel(int n, int m, int k, long N, const double *__restrict__ px, const double *__restrict__ pz,
             const double *__restrict__ sf, const double *__restrict__ sh, const double *__restrict__ Wc,
             double *Pxz)
{
    const long blk0 = (long)blockIdx.x * BLOCK;
    const Lane ln{blk0, threadIdx.x, N};
    if (blk0 + ln.tid >= N) return;
    const RecView<LAYOUT> fv(sf, ln, k * n), hv(sh, ln, k * m), zv(pz, ln, m), ov(Pxz, ln, n * m);
    double x[NX];
    load_rec<NX, 1, LAYOUT, false>(x, px, ln, n, 1, 0.0);
    for (int c = 0; c < m; ++c) {
        const double zc = zv.load(c);
        double acc[NX];
        FK_UNROLL for (int a = 0; a < NX; ++a) acc[a] = 0.0;
        for (int i = 0; i < k; ++i) {
            // Pxz += Wc[i] * outer(dx, dz)   (UKF.py:500-503)
            const double dz = hv.load(i * m + c) - zc;
            const double w = Wc[i];
            FK_UNROLL for (int a = 0; a < NX; ++a)
                if (a < n) acc[a] += w * ((fv.load(i * n + a) - x[a]) * dz);
        }
        FK_UNROLL for (int a = 0; a < NX; ++a)
            if (a < n) ov.store(a * m + c, acc[a]);
    }
}

// ------------------------------------------------------------- UKF correct --
// The tail of UnscentedKalmanFilter.update (UKF.py:470-481) for arbitrary hx:
//   K = Pxz S^-1 ; x += K (z - zp) ; P -= K (S K')
template <int NX, int NZ, int LAYOUT>
__global__ void __launch_bounds__(BLOCK)
ukf_correct_kernel(int n, int m, long N, const double *__restrict__ pPxz, const double *__restrict__ pzp,
                   const double *__restrict__ pS, const double *__restrict__ pz, double *px, double *pP,
                   double *pK, int32_t *status)
{
    const long blk0 = (long)blockIdx.x * BLOCK;
    const Lane ln{blk0, threadIdx.x, N};
    if (blk0 + ln.tid >= N) return;
    double x[NX], P[NX * NX], K[NX * NZ], S[NZ * NZ], z[NZ], zp[NZ];
    load_rec<NX, 1, LAYOUT, false>(x, px, ln, n, 1, 0.0);
    load_rec<NX, NX, LAYOUT, false>(P, pP, ln, n, n, 0.0);
    load_rec<NX, NZ, LAYOUT, false>(K, pPxz, ln, n, m, 0.0);
    load_rec<NZ, NZ, LAYOUT, false>(S, pS, ln, m, m, 1.0);
    load_rec<NZ, 1, LAYOUT, false>(z, pz, ln, m, 1, 0.0);
    load_rec<NZ, 1, LAYOUT, false>(zp, pzp, ln, m, 1, 0.0);
    // the padded diagonal of K (load_rec pads a==b with diag_pad=0) is zero: nothing to undo
    double Lf[NZ * NZ], d[NZ], dinv[NZ];
    FK_UNROLL for (int e = 0; e < NZ * NZ; ++e) Lf[e] = S[e];
    int st = 0;
    if (!ldlt2<NZ>(Lf, d, dinv)) st |= ST_NOT_PD;
    solve_rows_ldlt<NX, NZ>(Lf, dinv, K);
    FK_UNROLL for (int r = 0; r < NX; ++r) {
        double acc = K[r * NZ] * (z[0] - zp[0]);
        FK_UNROLL for (int c = 1; c < NZ; ++c) acc = fma(K[r * NZ + c], z[c] - zp[c], acc);
        x[r] += acc;
    }
    double SK[NZ * NX];
    FK_UNROLL for (int c = 0; c < NZ; ++c)
        FK_UNROLL for (int r = 0; r < NX; ++r) {
            double acc = S[c * NZ] * K[r * NZ];
            FK_UNROLL for (int q = 1; q < NZ; ++q) acc = fma(S[c * NZ + q], K[r * NZ + q], acc);
            SK[c * NX + r] = acc;
        }
    FK_UNROLL for (int r = 0; r < NX; ++r)
        FK_UNROLL for (int c = 0; c < NX; ++c) {
            double acc = K[r * NZ] * SK[c];
            FK_UNROLL for (int q = 1; q < NZ; ++q) acc = fma(K[r * NZ + q], SK[q * NX + c], acc);
            P[r * NX + c] -= acc;
        }
    store_rec<NX, 1, LAYOUT, false>(x, px, ln, n, 1);
    store_rec<NX, NX, LAYOUT, false>(P, pP, ln, n, n);
    if (pK) store_rec<NX, NZ, LAYOUT, false>(K, pK, ln, n, m);
    if (status) {
        if (!all_finite<NX>(x) || !all_finite<NX * NX>(P)) st |= ST_NONFINITE;
        status[blk0 + ln.tid] = st;
    }
}

// --------------------------------------------------- fused linear-model UKF --
// Per step (UKF.py:400-411, 462-481) with fx(x) = F x, hx(x) = H x:
//   L  = chol(scale P);  sigma_i = x, x +- L[:,k]          (sigma_points.py:167-175)
//   sf_i = F sigma_i ;  (x,P) = UT(sf, Wm, Wc, Q)
//   L  = chol(scale P);  sf_i = sigma_i(x,P)  (regenerated, UKF.py:407)
//   sh_i = H sf_i ; (zp,S) = UT(sh, Wm, Wc, R) ; Pxz = sum Wc_i (sf_i-x)(sh_i-zp)'
//   K = Pxz S^-1 ; x += K (z-zp) ; P -= K (S K')
template <int NX, int NZ, int LAYOUT>
__global__ void __launch_bounds__(BLOCK, (NX <= 4 ? 2 : 1))
ukf_linear_kernel(const UkfArgs a, const double *__restrict__ pF, const double *__restrict__ pH,
                  const double *__restrict__ pQ, const double *__restrict__ pR,
                  const double *__restrict__ pWm, const double *__restrict__ pWc,
                  const double *__restrict__ pz, const uint8_t *__restrict__ pmask)
{
    constexpr int KS = 2 * NX + 1;
    using SharedModel = LdsModel<NX, NZ>;
    __shared__ double s_model[SharedModel::SIZE + 2 * KS];
    const long N = a.N;
    const long blk0 = (long)blockIdx.x * BLOCK;
    const Lane ln{blk0, threadIdx.x, N};
    const bool live = blk0 + ln.tid < N;
    const Lane lr{blk0, live ? ln.tid : 0u, N};
    const int n = a.n, m = a.m;
    const int ks = 2 * n + 1;

    lds_fill<NX, NX>(s_model + SharedModel::OFF_F, pF, n, n, 1.0, ln.tid);
    lds_fill<NX, NX>(s_model + SharedModel::OFF_Q, pQ, n, n, 1.0, ln.tid);   // padded block of P stays I
    lds_fill<NZ, NX>(s_model + SharedModel::OFF_H, pH, m, n, 0.0, ln.tid);
    lds_fill<NZ, NZ>(s_model + SharedModel::OFF_R, pR, m, m, 1.0, ln.tid);
    // weights, re-indexed from the runtime point set (0, 1..n, n+1..2n) to the padded one
    // (0, 1..NX, NX+1..2NX); padded points get weight 0
    for (unsigned q = ln.tid; q < (unsigned)(2 * KS); q += BLOCK) {
        const int which = q / KS, i = q % KS;
        int src = -1;
        if (i == 0) src = 0;
        else if (i <= NX) { if (i <= n) src = i; }
        else { if (i - NX <= n) src = n + (i - NX); }
        const double *W = which ? pWc : pWm;
        s_model[SharedModel::SIZE + q] = (src >= 0 && src < ks) ? W[src] : 0.0;
    }
    __syncthreads();
    const SharedModel sm{s_model};
    const double *sWm = s_model + SharedModel::SIZE, *sWc = sWm + KS;

    double x[NX], P[NX * NX];
    load_rec<NX, 1, LAYOUT, false>(x, a.x, lr, n, 1, 0.0);
    load_rec<NX, NX, LAYOUT, false>(P, a.P, lr, n, n, 1.0);
    int st = 0;

    for (long t = 0; t < a.T; ++t) {
        double z[NZ];
        bool has_z = true;
        if (pmask) has_z = pmask[t * N + lr.blk0 + lr.tid] != 0;
        load_rec<NZ, 1, LAYOUT, false>(z, pz + t * N * m, lr, m, 1, 0.0);

        // ---- predict
        double L[NX * NX];
        if (!chol_lower<NX>(P, a.scale, L)) st |= ST_NOT_PD;
        // sf_i = F sigma_i: F x and F L[:,k]
        double Fx[NX], FL[NX * NX];   // FL[r][k] = sum_c F[r][c] L[c][k]
        FK_UNROLL for (int r = 0; r < NX; ++r) {
            double f[NX];
            sm.rowF(r, f);
            Fx[r] = dot<NX>(f, x);
            FK_UNROLL for (int k = 0; k < NX; ++k) {
                // propagate the actual points (x + L[:,k]) like the reference does: F (x + l) is
                // not bitwise F x + F l, so form the point first
                double accp = 0.0, accm = 0.0;
                FK_UNROLL for (int c = 0; c < NX; ++c) {
                    const double sp = x[c] - (-L[c * NX + k]), smn = x[c] - L[c * NX + k];
                    accp = (c == 0) ? f[0] * sp : fma(f[c], sp, accp);
                    accm = (c == 0) ? f[0] * smn : fma(f[c], smn, accm);
                }
                FL[r * NX + k] = accp;       // component r of F sigma_{k+1}
                P[r * NX + k] = accm;        // reuse P as scratch: component r of F sigma_{NX+k+1}
            }
            FK_STAGE();
        }
        // mean: x = sum Wm_i sf_i (index order 0, 1..NX, NX+1..2NX)
        double xm[NX];
        FK_UNROLL for (int r = 0; r < NX; ++r) {
            double acc = sWm[0] * Fx[r];
            FK_UNROLL for (int k = 0; k < NX; ++k) acc = fma(sWm[1 + k], FL[r * NX + k], acc);
            FK_UNROLL for (int k = 0; k < NX; ++k) acc = fma(sWm[1 + NX + k], P[r * NX + k], acc);
            xm[r] = acc;
        }
        // covariance: Pn = sum Wc_i y_i y_i' + Q
        double Pn[NX * NX];
        {
            double y0[NX];
            FK_UNROLL for (int r = 0; r < NX; ++r) y0[r] = Fx[r] - xm[r];
            FK_UNROLL for (int r = 0; r < NX; ++r) {
                FK_UNROLL for (int k = 0; k < NX; ++k) {
                    FL[r * NX + k] -= xm[r];
                    P[r * NX + k] -= xm[r];
                }
            }
            FK_UNROLL for (int a2 = 0; a2 < NX; ++a2) {
                double q[NX];
                sm.rowQ(a2, q);
                FK_UNROLL for (int b = 0; b < NX; ++b) {
                    double acc = y0[a2] * (sWc[0] * y0[b]);
                    FK_UNROLL for (int k = 0; k < NX; ++k) acc = fma(FL[a2 * NX + k], sWc[1 + k] * FL[b * NX + k], acc);
                    FK_UNROLL for (int k = 0; k < NX; ++k) acc = fma(P[a2 * NX + k], sWc[1 + NX + k] * P[b * NX + k], acc);
                    Pn[a2 * NX + b] = acc + q[b];
                }
                FK_STAGE();
            }
        }
        FK_UNROLL for (int r = 0; r < NX; ++r) x[r] = xm[r];
        FK_UNROLL for (int e = 0; e < NX * NX; ++e) P[e] = Pn[e];

        // ---- update
        if (has_z) {
            if (!chol_lower<NX>(P, a.scale, L)) st |= ST_NOT_PD;
            // sh_i = H sf_i with sf_i regenerated from the prior
            double h0[NZ], hp[NZ * NX], hm[NZ * NX];
            FK_UNROLL for (int r = 0; r < NZ; ++r) {
                double h[NX];
                sm.rowH(r, h);
                h0[r] = dot<NX>(h, x);
                FK_UNROLL for (int k = 0; k < NX; ++k) {
                    double accp = 0.0, accm = 0.0;
                    FK_UNROLL for (int c = 0; c < NX; ++c) {
                        const double sp = x[c] - (-L[c * NX + k]), smn = x[c] - L[c * NX + k];
                        accp = (c == 0) ? h[0] * sp : fma(h[c], sp, accp);
                        accm = (c == 0) ? h[0] * smn : fma(h[c], smn, accm);
                    }
                    hp[r * NX + k] = accp;
                    hm[r * NX + k] = accm;
                }
                FK_STAGE();
            }
            double zp[NZ];
            FK_UNROLL for (int r = 0; r < NZ; ++r) {
                double acc = sWm[0] * h0[r];
                FK_UNROLL for (int k = 0; k < NX; ++k) acc = fma(sWm[1 + k], hp[r * NX + k], acc);
                FK_UNROLL for (int k = 0; k < NX; ++k) acc = fma(sWm[1 + NX + k], hm[r * NX + k], acc);
                zp[r] = acc;
            }
            FK_UNROLL for (int r = 0; r < NZ; ++r) {
                h0[r] -= zp[r];
                FK_UNROLL for (int k = 0; k < NX; ++k) {
                    hp[r * NX + k] -= zp[r];
                    hm[r * NX + k] -= zp[r];
                }
            }
            double S[NZ * NZ];
            FK_UNROLL for (int r = 0; r < NZ; ++r) {
                double rr[NZ];
                sm.rowR(r, rr);
                FK_UNROLL for (int c = 0; c < NZ; ++c) {
                    double acc = h0[r] * (sWc[0] * h0[c]);
                    FK_UNROLL for (int k = 0; k < NX; ++k) acc = fma(hp[r * NX + k], sWc[1 + k] * hp[c * NX + k], acc);
                    FK_UNROLL for (int k = 0; k < NX; ++k) acc = fma(hm[r * NX + k], sWc[1 + NX + k] * hm[c * NX + k], acc);
                    S[r * NZ + c] = acc + rr[c];
                }
            }
            // Pxz = sum Wc_i (sf_i - x)(sh_i - zp)' ; sf_0 - x = 0, sf_{k+1} - x = (x + l_k) - x
            double K[NX * NZ];
            FK_UNROLL for (int r = 0; r < NX; ++r) {
                FK_UNROLL for (int c = 0; c < NZ; ++c) {
                    double acc = sWc[0] * (((x[r]) - x[r]) * h0[c]);
                    FK_UNROLL for (int k = 0; k < NX; ++k) {
                        const double dp = (x[r] - (-L[r * NX + k])) - x[r];
                        acc += sWc[1 + k] * (dp * hp[c * NX + k]);
                    }
                    FK_UNROLL for (int k = 0; k < NX; ++k) {
                        const double dm = (x[r] - L[r * NX + k]) - x[r];
                        acc += sWc[1 + NX + k] * (dm * hm[c * NX + k]);
                    }
                    K[r * NZ + c] = acc;
                }
                FK_STAGE();
            }
            // K = Pxz S^-1
            double Lf[NZ * NZ], d[NZ], dinv[NZ];
            FK_UNROLL for (int e = 0; e < NZ * NZ; ++e) Lf[e] = S[e];
            if (!ldlt2<NZ>(Lf, d, dinv)) st |= ST_NOT_PD;
            solve_rows_ldlt<NX, NZ>(Lf, dinv, K);
            // x += K (z - zp)
            FK_UNROLL for (int r = 0; r < NX; ++r) {
                double acc = 0.0;
                FK_UNROLL for (int c = 0; c < NZ; ++c) acc = (c == 0) ? K[r * NZ] * (z[0] - zp[0]) : fma(K[r * NZ + c], z[c] - zp[c], acc);
                x[r] += acc;
            }
            // P -= K (S K')
            double SK[NZ * NX];
            FK_UNROLL for (int c = 0; c < NZ; ++c)
                FK_UNROLL for (int r = 0; r < NX; ++r) {
                    double acc = S[c * NZ] * K[r * NZ];
                    FK_UNROLL for (int q = 1; q < NZ; ++q) acc = fma(S[c * NZ + q], K[r * NZ + q], acc);
                    SK[c * NX + r] = acc;
                }
            FK_UNROLL for (int r = 0; r < NX; ++r)
                FK_UNROLL for (int c = 0; c < NX; ++c) {
                    double acc = K[r * NZ] * SK[c];
                    FK_UNROLL for (int q = 1; q < NZ; ++q) acc = fma(K[r * NZ + q], SK[q * NX + c], acc);
                    P[r * NX + c] -= acc;
                }
        }
        if (live) {
            if (a.means) store_rec<NX, 1, LAYOUT, false>(x, a.means + t * N * n, ln, n, 1);
            if (a.covs) store_rec<NX, NX, LAYOUT, false>(P, a.covs + t * N * n * n, ln, n, n);
        }
    }
    if (live) {
        store_rec<NX, 1, LAYOUT, false>(x, a.x, ln, n, 1);
        store_rec<NX, NX, LAYOUT, false>(P, a.P, ln, n, n);
        if (a.status) {
            if (!all_finite<NX>(x) || !all_finite<NX * NX>(P)) st |= ST_NONFINITE;
            a.status[ln.blk0 + ln.tid] = st;
        }
    }
}

static int fail(int code, const char *msg)
{
    set_last_error(msg);
    return code;
}

}  // namespace fk

using namespace fk;

#define FK_BY_NX(n, CALL)            \
    do {                             \
        if ((n) <= 2) { CALL(2); }   \
        else if ((n) <= 4) { CALL(4); } \
        else if ((n) <= 6) { CALL(6); } \
        else if ((n) <= 8) { CALL(8); } \
        else if ((n) <= 12) { CALL(12); } \
        else { CALL(16); }           \
    } while (0)

extern "C" {

int fk_ut_sigma_points_f64(int32_t n, int64_t N, int32_t layout, double scale, const double *x,
                           const double *P, double *sigmas, int32_t *status, void *stream)
{
    if (n < 1 || n > 16) return fail(FK_ERR_UNSUPPORTED, "sigma points: dim_x must be 1..16");
    if (N < 0 || !x || !P || !sigmas) return fail(FK_ERR_BAD_ARG, "sigma points: bad argument");
    if ((double)N * (2 * n + 1) * n * 8.0 >= 4294967296.0) return fail(FK_ERR_UNSUPPORTED, "sigma points: record block >= 4 GiB, split the batch");
    if (N == 0) return FK_OK;
    const dim3 grid((unsigned)((N + BLOCK - 1) / BLOCK)), block(BLOCK);
#define CALL(NXV)                                                                                      \
    if (layout == FK_LAYOUT_SOA)                                                                       \
        hipLaunchKernelGGL((sigma_kernel<NXV, LAYOUT_SOA>), grid, block, 0, (hipStream_t)stream, n, N, \
                           scale, x, P, sigmas, status);                                               \
    else                                                                                               \
        hipLaunchKernelGGL((sigma_kernel<NXV, LAYOUT_AOS>), grid, block, 0, (hipStream_t)stream, n, N, \
                           scale, x, P, sigmas, status)
    FK_BY_NX(n, CALL);
#undef CALL
    return check_launch("sigma_kernel");
}

int fk_ut_transform_f64(int32_t n, int32_t k, int64_t N, int32_t layout, const double *sigmas,
                        const double *Wm, const double *Wc, const double *noise_cov, double *x_out,
                        double *P_out, void *stream)
{
    if (n < 1 || n > 16 || k < 1) return fail(FK_ERR_UNSUPPORTED, "unscented transform: dim must be 1..16");
    if (N < 0 || !sigmas || !Wm || !Wc || !x_out || !P_out) return fail(FK_ERR_BAD_ARG, "unscented transform: bad argument");
    if ((double)N * k * n * 8.0 >= 4294967296.0) return fail(FK_ERR_UNSUPPORTED, "unscented transform: record block >= 4 GiB, split the batch");
    if (N == 0) return FK_OK;
    const dim3 grid((unsigned)((N + BLOCK - 1) / BLOCK)), block(BLOCK);
    if (k == 2 * n + 1 && (n == 2 || n == 4 || n == 6)) {
#define REG(NXV)                                                                                          \
    if (layout == FK_LAYOUT_SOA)                                                                          \
        hipLaunchKernelGGL((ut_reg_kernel<NXV, LAYOUT_SOA>), grid, block, 0, (hipStream_t)stream, (long)N, \
                           sigmas, Wm, Wc, noise_cov, x_out, P_out);                                      \
    else                                                                                                  \
        hipLaunchKernelGGL((ut_reg_kernel<NXV, LAYOUT_AOS>), grid, block, 0, (hipStream_t)stream, (long)N, \
                           sigmas, Wm, Wc, noise_cov, x_out, P_out)
        if (n == 2) { REG(2); }
        else if (n == 4) { REG(4); }
        else { REG(6); }
#undef REG
        return check_launch("ut_reg_kernel");
    }
#define CALL(NXV)                                                                                   \
    if (layout == FK_LAYOUT_SOA)                                                                    \
        hipLaunchKernelGGL((ut_kernel<NXV, LAYOUT_SOA>), grid, block, 0, (hipStream_t)stream, n, k, \
                           N, sigmas, Wm, Wc, noise_cov, x_out, P_out);                             \
    else                                                                                            \
        hipLaunchKernelGGL((ut_kernel<NXV, LAYOUT_AOS>), grid, block, 0, (hipStream_t)stream, n, k, \
                           N, sigmas, Wm, Wc, noise_cov, x_out, P_out)
    FK_BY_NX(n, CALL);
#undef CALL
    return check_launch("ut_kernel");
}

int fk_ut_cross_variance_f64(int32_t n, int32_t m, int32_t k, int64_t N, int32_t layout,
                             const double *x, const double *z, const double *sigmas_f,
                             const double *sigmas_h, const double *Wc, double *Pxz, void *stream)
{
    if (n < 1 || n > 16 || m < 1 || k < 1) return fail(FK_ERR_UNSUPPORTED, "cross variance: dim_x must be 1..16");
    if (N < 0 || !x || !z || !sigmas_f || !sigmas_h || !Wc || !Pxz) return fail(FK_ERR_BAD_ARG, "cross variance: bad argument");
    if ((double)N * k * (n > m ? n : m) * 8.0 >= 4294967296.0) return fail(FK_ERR_UNSUPPORTED, "cross variance: record block >= 4 GiB, split the batch");
    if (N == 0) return FK_OK;
    const dim3 grid((unsigned)((N + BLOCK - 1) / BLOCK)), block(BLOCK);
#define CALL(NXV)                                                                                      \
    if (layout == FK_LAYOUT_SOA)                                                                       \
        hipLaunchKernelGGL((cross_kernel<NXV, LAYOUT_SOA>), grid, block, 0, (hipStream_t)stream, n, m, \
                           k, N, x, z, sigmas_f, sigmas_h, Wc, Pxz);                                   \
    else                                                                                               \
        hipLaunchKernelGGL((cross_kernel<NXV, LAYOUT_AOS>), grid, block, 0, (hipStream_t)stream, n, m, \
                           k, N, x, z, sigmas_f, sigmas_h, Wc, Pxz)
    FK_BY_NX(n, CALL);
#undef CALL
    return check_launch("cross_kernel");
}

int fk_ukf_correct_f64(int32_t n, int32_t m, int64_t N, int32_t layout, const double *Pxz, const double *zp,
                       const double *S, const double *z, double *x, double *P, double *K, int32_t *status,
                       void *stream)
{
    if (n < 1 || n > 16 || m < 1 || m > 8) return fail(FK_ERR_UNSUPPORTED, "ukf correct: dim_x 1..16, dim_z 1..8");
    if (N < 0 || !Pxz || !zp || !S || !z || !x || !P) return fail(FK_ERR_BAD_ARG, "ukf correct: bad argument");
    if ((double)N * n * n * 8.0 >= 4294967296.0) return fail(FK_ERR_UNSUPPORTED, "ukf correct: record block >= 4 GiB, split the batch");
    if (N == 0) return FK_OK;
    const dim3 grid((unsigned)((N + BLOCK - 1) / BLOCK)), block(BLOCK);
    hipStream_t s = (hipStream_t)stream;
#define GOZ(NXV, NZV)                                                                                      \
    if (layout == FK_LAYOUT_SOA)                                                                           \
        hipLaunchKernelGGL((ukf_correct_kernel<NXV, NZV, LAYOUT_SOA>), grid, block, 0, s, n, m, N, Pxz, zp, \
                           S, z, x, P, K, status);                                                         \
    else                                                                                                   \
        hipLaunchKernelGGL((ukf_correct_kernel<NXV, NZV, LAYOUT_AOS>), grid, block, 0, s, n, m, N, Pxz, zp, \
                           S, z, x, P, K, status)
#define CALL(NXV)                  \
    if (m <= 4) { GOZ(NXV, 4); }   \
    else { GOZ(NXV, 8); }
    FK_BY_NX(n, CALL);
#undef CALL
#undef GOZ
    return check_launch("ukf_correct_kernel");
}

int fk_ukf_linear_batch_f64(const fk_ukf_desc *d, const double *F, const double *H, const double *Q,
                            const double *R, const double *Wm, const double *Wc, const double *z,
                            const uint8_t *mask, double *x, double *P, double *means, double *covs,
                            int32_t *status, void *stream)
{
    if (!d) return fail(FK_ERR_BAD_ARG, "desc is NULL");
    if (d->n < 1 || d->n > 8 || d->m < 1 || d->m > 4) return fail(FK_ERR_UNSUPPORTED, "fused linear UKF: dim_x 1..8, dim_z 1..4");
    if (d->N < 0 || d->T < 0 || !F || !H || !Q || !R || !Wm || !Wc || !z || !x || !P)
        return fail(FK_ERR_BAD_ARG, "fused linear UKF: bad argument");
    if ((double)d->N * d->n * d->n * 8.0 >= 4294967296.0) return fail(FK_ERR_UNSUPPORTED, "fused linear UKF: record block >= 4 GiB, split the batch");
    if (d->N == 0 || d->T == 0) return FK_OK;
    UkfArgs a{};
    a.F = F; a.H = H; a.Q = Q; a.R = R; a.Wm = Wm; a.Wc = Wc; a.z = z; a.mask = mask;
    a.x = x; a.P = P; a.means = means; a.covs = covs; a.status = status;
    a.N = d->N; a.T = d->T; a.n = d->n; a.m = d->m; a.scale = d->scale;
    const dim3 grid((unsigned)((a.N + BLOCK - 1) / BLOCK)), block(BLOCK);
    hipStream_t s = (hipStream_t)stream;
#define GO(NXV, NZV)                                                                                    \
    if (d->layout == FK_LAYOUT_SOA)                                                                     \
        hipLaunchKernelGGL((ukf_linear_kernel<NXV, NZV, LAYOUT_SOA>), grid, block, 0, s, a, F, H, Q, R, \
                           Wm, Wc, z, mask);                                                            \
    else                                                                                                \
        hipLaunchKernelGGL((ukf_linear_kernel<NXV, NZV, LAYOUT_AOS>), grid, block, 0, s, a, F, H, Q, R, \
                           Wm, Wc, z, mask)
    if (d->n <= 2 && d->m <= 2) { GO(2, 2); }
    else if (d->n <= 4 && d->m <= 2) { GO(4, 2); }
    else if (d->n <= 6 && d->m <= 3) { GO(6, 3); }
    else { GO(8, 4); }
#undef GO
    return check_launch("ukf_linear_kernel");
}

}  // extern "C"
