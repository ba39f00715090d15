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

}  // namespace fk
