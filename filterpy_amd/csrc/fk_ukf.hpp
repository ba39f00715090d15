// fk_ukf.hpp -- one predict + update step of the fused linear-model UKF, register-lean organisation ("V2" of
// ukf_kernels.hip, see there).  __host__ __device__: the kernel runs it per lane, tests/hostcheck runs the very
// same code on the host against the oracle.
//
//   fresh() returns a view {sm, Wm, Wc} of the shared model (rowF / rowQ / rowH / rowR policy) and the padded
//   sigma-point weights; the kernel hands out a new, optimiser-opaque view on every call so that the broadcast
//   LDS reads of the 2n+1 unrolled points are not hoisted and held, the host hands out the same plain arrays.
#pragma once

#include "fk_math.hpp"
#include "fk_math_sym.hpp"

#if defined(__HIP_DEVICE_COMPILE__)
#define FK_OPAQUE(v) asm volatile("" : "+v"(v))
#else
#define FK_OPAQUE(v) ((void)0)
#endif

namespace fk {

struct NoSweep {
    FK_HD void operator()() const {}
};

//   sweep() is called at the head of each of the four passes over the sigma points: a model policy that keeps F / H
//   in scalar registers for the length of one pass re-derives its (optimiser-opaque) base there, so that the rows
//   are re-fetched per pass instead of being hoisted out of the time loop and held.
template <int NX, int NZ, class Fresh, class Sweep = NoSweep>
FK_HD int ukf_linear_step_v2(double (&x)[NX], double (&P)[NX * (NX + 1) / 2], const double (&z)[NZ], bool has_z,
                             double scale, Fresh &&fresh, Sweep &&sweep = Sweep{})
{
    constexpr int KS = 2 * NX + 1;
    constexpr int PL = NX * (NX + 1) / 2;
    int st = 0;
    const auto mv0 = fresh();
    const auto &sm = mv0.sm;
    // ---------------- predict (UKF.py:400-411)
    double L[PL];
    if (!chol_packed<NX>(P, scale, L)) st |= ST_NOT_PD;
    // sweep 1: x- = sum_i Wm_i F sigma_i, one output component (row of F) at a time, points in
    // index order 0, x + L[:,k] (k = 0..n-1), x - L[:,k]
    double xm[NX];
    sweep();
    FK_UNROLL for (int i = 0; i < KS; ++i) {
        const auto mv = fresh();
        const auto &sm = mv.sm;
        const double *sWm = mv.Wm, *sWc = mv.Wc;
        (void)sWm; (void)sWc;
        FK_UNROLL for (int r = 0; r < NX; ++r) {
            double f[NX];
            sm.rowF(r, f);
            double v;
            if (i == 0) {
                v = dot<NX>(f, x);
            } else if (i <= NX) {
                v = f[0] * (x[0] - (-lcol<NX>(L, 0, i - 1)));
                FK_UNROLL for (int c = 1; c < NX; ++c) v = fma(f[c], x[c] - (-lcol<NX>(L, c, i - 1)), v);
            } else {
                v = f[0] * (x[0] - lcol<NX>(L, 0, i - 1 - NX));
                FK_UNROLL for (int c = 1; c < NX; ++c) v = fma(f[c], x[c] - lcol<NX>(L, c, i - 1 - NX), v);
            }
            xm[r] = (i == 0) ? sWm[0] * v : fma(sWm[i], v, xm[r]);
        }
        FK_STAGE();
    }
    // sweep 2: P- = sum_i Wc_i y_i y_i' + Q, y_i = F sigma_i - x-   (upper triangle).
    // The points are recomputed from copies the optimiser cannot relate to sweep 1 (otherwise it
    // common-subexpression-eliminates the recomputation by keeping all (2n+1) n values alive).
    double Pn[PL];
    FK_UNROLL for (int c = 0; c < NX; ++c) FK_OPAQUE(x[c]);
    FK_UNROLL for (int e = 0; e < PL; ++e) FK_OPAQUE(L[e]);
    sweep();
    FK_UNROLL for (int i = 0; i < KS; ++i) {
        const auto mv = fresh();
        const auto &sm = mv.sm;
        const double *sWm = mv.Wm, *sWc = mv.Wc;
        (void)sWm; (void)sWc;
        double y[NX], wy[NX];
        FK_UNROLL for (int r = 0; r < NX; ++r) {
            double f[NX];
            sm.rowF(r, f);
            double v;
            if (i == 0) {
                v = dot<NX>(f, x);
            } else if (i <= NX) {
                v = f[0] * (x[0] - (-lcol<NX>(L, 0, i - 1)));
                FK_UNROLL for (int c = 1; c < NX; ++c) v = fma(f[c], x[c] - (-lcol<NX>(L, c, i - 1)), v);
            } else {
                v = f[0] * (x[0] - lcol<NX>(L, 0, i - 1 - NX));
                FK_UNROLL for (int c = 1; c < NX; ++c) v = fma(f[c], x[c] - lcol<NX>(L, c, i - 1 - NX), v);
            }
            y[r] = v - xm[r];
        }
        FK_UNROLL for (int r = 0; r < NX; ++r) wy[r] = sWc[i] * y[r];
        FK_UNROLL for (int a2 = 0; a2 < NX; ++a2)
            FK_UNROLL for (int b = 0; b < NX; ++b)
                if (b >= a2)
                    Pn[sym_idx<NX>(a2, b)] = (i == 0) ? y[a2] * wy[b] : fma(y[a2], wy[b], Pn[sym_idx<NX>(a2, b)]);
        FK_STAGE();
    }
    FK_UNROLL for (int r = 0; r < NX; ++r) {
        double q[NX];
        sm.rowQ(r, q);
        FK_UNROLL for (int b = 0; b < NX; ++b)
            if (b >= r) P[sym_idx<NX>(r, b)] = Pn[sym_idx<NX>(r, b)] + q[b];
        x[r] = xm[r];
    }

    // ---------------- update (UKF.py:462-481), sigma points regenerated from the prior (:407)
    if (has_z) {
        if (!chol_packed<NX>(P, scale, L)) st |= ST_NOT_PD;
        // sweep 1: zp = sum_i Wm_i H sigma_i, point by point (index order 0, +k, -k)
        double zp[NZ];
        sweep();
        FK_UNROLL for (int i = 0; i < KS; ++i) {
            const auto mv = fresh();
            const auto &sm = mv.sm;
            const double *sWm = mv.Wm, *sWc = mv.Wc;
            (void)sWm; (void)sWc;
            FK_UNROLL for (int r = 0; r < NZ; ++r) {
                double h[NX];
                sm.rowH(r, h);
                double v;
                if (i == 0) {
                    v = dot<NX>(h, x);
                } else if (i <= NX) {
                    v = h[0] * (x[0] - (-lcol<NX>(L, 0, i - 1)));
                    FK_UNROLL for (int c = 1; c < NX; ++c) v = fma(h[c], x[c] - (-lcol<NX>(L, c, i - 1)), v);
                } else {
                    v = h[0] * (x[0] - lcol<NX>(L, 0, i - 1 - NX));
                    FK_UNROLL for (int c = 1; c < NX; ++c) v = fma(h[c], x[c] - lcol<NX>(L, c, i - 1 - NX), v);
                }
                zp[r] = (i == 0) ? sWm[0] * v : fma(sWm[i], v, zp[r]);
            }
            FK_STAGE();
        }
        // sweep 2: S = sum Wc_i d_i d_i' + R,  Pxz = sum Wc_i (sf_i - x) d_i',  d_i = H sigma_i - zp
        double S[NZ * NZ], K[NX * NZ];
        FK_UNROLL for (int c = 0; c < NX; ++c) FK_OPAQUE(x[c]);
        FK_UNROLL for (int e = 0; e < PL; ++e) FK_OPAQUE(L[e]);
        sweep();
        FK_UNROLL for (int i = 0; i < KS; ++i) {
            const auto mv = fresh();
            const auto &sm = mv.sm;
            const double *sWm = mv.Wm, *sWc = mv.Wc;
            (void)sWm; (void)sWc;
            double d[NZ], wd[NZ];
            FK_UNROLL for (int r = 0; r < NZ; ++r) {
                double h[NX];
                sm.rowH(r, h);
                double v;
                if (i == 0) {
                    v = dot<NX>(h, x);
                } else if (i <= NX) {
                    v = h[0] * (x[0] - (-lcol<NX>(L, 0, i - 1)));
                    FK_UNROLL for (int c = 1; c < NX; ++c) v = fma(h[c], x[c] - (-lcol<NX>(L, c, i - 1)), v);
                } else {
                    v = h[0] * (x[0] - lcol<NX>(L, 0, i - 1 - NX));
                    FK_UNROLL for (int c = 1; c < NX; ++c) v = fma(h[c], x[c] - lcol<NX>(L, c, i - 1 - NX), v);
                }
                d[r] = v - zp[r];
            }
            FK_UNROLL for (int r = 0; r < NZ; ++r) wd[r] = sWc[i] * d[r];
            FK_UNROLL for (int r = 0; r < NZ; ++r)
                FK_UNROLL for (int c = 0; c < NZ; ++c)
                    S[r * NZ + c] = (i == 0) ? d[r] * wd[c] : fma(d[r], wd[c], S[r * NZ + c]);
            FK_UNROLL for (int r = 0; r < NX; ++r) {
                double dx;
                if (i == 0) dx = x[r] - x[r];
                else if (i <= NX) dx = (x[r] - (-lcol<NX>(L, r, i - 1))) - x[r];
                else dx = (x[r] - lcol<NX>(L, r, i - 1 - NX)) - x[r];
                FK_UNROLL for (int c = 0; c < NZ; ++c) {
                    const double term = sWc[i] * (dx * d[c]);
                    K[r * NZ + c] = (i == 0) ? term : K[r * NZ + c] + term;
                }
            }
            FK_STAGE();
        }
        FK_UNROLL for (int r = 0; r < NZ; ++r) {
            double rr[NZ];
            sm.rowR(r, rr);
            FK_UNROLL for (int c = 0; c < NZ; ++c) S[r * NZ + c] += rr[c];
        }
        // K = Pxz S^-1
        double Lf[NZ * NZ], d[NZ], dinv[NZ];
        FK_UNROLL for (int e = 0; e < NZ * NZ; ++e) Lf[e] = S[e];
        if (!ldlt2<NZ>(Lf, d, dinv)) st |= ST_NOT_PD;
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
            FK_UNROLL for (int q = 0; q < NZ; ++q) FK_OPAQUE(sk[q]);
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
// in the reference's index order, the sigma points and their images are regenerated point by point exactly like
// ukf_linear_step_v2's sweeps.  inv(Pb) is applied by an L D L' solve (like every other gain here).
// Two halves, so that the caller may keep xn / Pn out of the registers during the first:
//   ukf_linear_rts_gain    (x, P) -> xb, Pb (packed), K            (the two sweeps and the solve)
//   ukf_linear_rts_correct x, P updated in place from xn, Pn, xb, Pb, K   (Pb is destroyed)
template <int NX, class Fresh, class Sweep = NoSweep>
FK_HD int ukf_linear_rts_gain(double (&x)[NX], const double (&P)[NX * (NX + 1) / 2], double scale, double (&xb)[NX],
                              double (&Pb)[NX * (NX + 1) / 2], double (&K)[NX * NX], Fresh &&fresh,
                              Sweep &&sweep = Sweep{})
{
    constexpr int KS = 2 * NX + 1;
    constexpr int PL = NX * (NX + 1) / 2;
    int st = 0;
    double L[PL];
    if (!chol_packed<NX>(P, scale, L)) st |= ST_NOT_PD;
    // sweep 1: xb = sum_i Wm_i F sigma_i
    sweep();
    FK_UNROLL for (int i = 0; i < KS; ++i) {
        const auto mv = fresh();
        const auto &sm = mv.sm;
        const double *sWm = mv.Wm;
        FK_UNROLL for (int r = 0; r < NX; ++r) {
            double f[NX];
            sm.rowF(r, f);
            double v;
            if (i == 0) {
                v = dot<NX>(f, x);
            } else if (i <= NX) {
                v = f[0] * (x[0] - (-lcol<NX>(L, 0, i - 1)));
                FK_UNROLL for (int c = 1; c < NX; ++c) v = fma(f[c], x[c] - (-lcol<NX>(L, c, i - 1)), v);
            } else {
                v = f[0] * (x[0] - lcol<NX>(L, 0, i - 1 - NX));
                FK_UNROLL for (int c = 1; c < NX; ++c) v = fma(f[c], x[c] - lcol<NX>(L, c, i - 1 - NX), v);
            }
            xb[r] = (i == 0) ? sWm[0] * v : fma(sWm[i], v, xb[r]);
        }
        FK_STAGE();
    }
    // sweep 2: Pb = sum Wc_i y_i y_i' (+ Q), Pxb = sum Wc_i z_i y_i', y_i = F sigma_i - xb, z_i = sigma_i - x
    FK_UNROLL for (int c = 0; c < NX; ++c) FK_OPAQUE(x[c]);
    FK_UNROLL for (int e = 0; e < PL; ++e) FK_OPAQUE(L[e]);
    sweep();
    FK_UNROLL for (int i = 0; i < KS; ++i) {
        const auto mv = fresh();
        const auto &sm = mv.sm;
        const double *sWc = mv.Wc;
        double y[NX], wy[NX];
        FK_UNROLL for (int r = 0; r < NX; ++r) {
            double f[NX];
            sm.rowF(r, f);
            double v;
            if (i == 0) {
                v = dot<NX>(f, x);
            } else if (i <= NX) {
                v = f[0] * (x[0] - (-lcol<NX>(L, 0, i - 1)));
                FK_UNROLL for (int c = 1; c < NX; ++c) v = fma(f[c], x[c] - (-lcol<NX>(L, c, i - 1)), v);
            } else {
                v = f[0] * (x[0] - lcol<NX>(L, 0, i - 1 - NX));
                FK_UNROLL for (int c = 1; c < NX; ++c) v = fma(f[c], x[c] - lcol<NX>(L, c, i - 1 - NX), v);
            }
            y[r] = v - xb[r];
        }
        FK_UNROLL for (int r = 0; r < NX; ++r) wy[r] = sWc[i] * y[r];
        FK_UNROLL for (int a2 = 0; a2 < NX; ++a2)
            FK_UNROLL for (int b = 0; b < NX; ++b)
                if (b >= a2)
                    Pb[sym_idx<NX>(a2, b)] = (i == 0) ? y[a2] * wy[b] : fma(y[a2], wy[b], Pb[sym_idx<NX>(a2, b)]);
        FK_UNROLL for (int r = 0; r < NX; ++r) {
            double z;
            if (i == 0) z = x[r] - x[r];
            else if (i <= NX) z = (x[r] - (-lcol<NX>(L, r, i - 1))) - x[r];
            else z = (x[r] - lcol<NX>(L, r, i - 1 - NX)) - x[r];
            FK_UNROLL for (int c = 0; c < NX; ++c) {
                const double term = sWc[i] * (z * y[c]);
                K[r * NX + c] = (i == 0) ? term : K[r * NX + c] + term;
            }
        }
        FK_STAGE();
    }
    {
        const auto mv = fresh();
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
        if (!ldlt_packed<NX>(Lp, d, dinv)) st |= ST_NOT_PD;
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
