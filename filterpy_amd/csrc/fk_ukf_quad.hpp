// fk_ukf_quad.hpp -- one predict + update step of the fused linear-model UKF, and one backward step of its smoother, with SEVERAL
// LANES PER TRACK (LN = 4: a quad, DPP exchanges; LN = 8: ds_swizzle) -- the arithmetic of ukf_mlg.hip (dim_x 10..16).
// __host__ __device__ like fk_ukf.hpp: the kernels run it with the exchanges as cross-lane moves, tests/hostcheck runs the very same
// code on the host with the lanes of a track as fibers in lockstep (tests/test_hostcheck_ukf_quad.py).
//
// The filter step is ukf_linear_step_v4 (fk_ukf.hpp: UKF.py:400-411, 462-481 with fx = F x, hx = H x, sums regrouped over the +- pairs
// of sigma points; every sum below keeps v4's order of terms), distributed:
//   * lane q of the group holds rows q, q + LN, q + 2 LN, ... of P (CYCLIC, not kf_mlg's blocks: column j of the Cholesky factor only
//     touches rows below j, and with cyclic rows every lane has its share of them down to the last columns); a slot past row
//     n-1 duplicates row n-1 (same inputs, same instructions: same values, so nothing is predicated); x is replicated;
//   * the factor of scale * P column by column: the pivot and row j of the factor (final once column j-1 is done) are BROADCAST
//     from their owner, each lane eliminates its own rows -- and the same broadcast row is all the images need:
//         predict   F L (own rows) and F x (own rows) accumulate a row of L at a time:  FL[a][k] += F[a][j] L[j][k]
//         update    H L likewise, its rows dealt out like P's (dim_z <= 8), its columns gathered for S; the cross variance
//                   takes the lane's OWN rows of L
//   * P- = (sum Wc) y0 y0' + sum_k wp_k f_k f_k' + Q: column k of F L is gathered (pre-multiplied by its pair weight by the
//     owners), each lane accumulates its own rows;
//   * dim_z <= 4: S, its L D L' and z - zp are replicated; the gain's rows are solved by their owners and gathered one at a time
//     for x += K (z - zp) and P -= K (S K').  dim_z >= 5: S is distributed and factored like P, P -= W W' (see there).
// The only exchange is "the value lane o of my group holds" (quad.bcast<o>(v)); there is no cross-lane sum.  A row of P computed
// by one lane and its mirror image computed by another agree to a rounding (f_a (w f_b) against f_b (w f_a)), not bit for
// bit; the factorisation reads a row's own elements.
// A missing measurement runs the update half on z = 0 with the gain, S and the residual selected to zero (no branch).
#pragma once

#include "fk_ukf.hpp"

namespace fk {

struct UkfQuadModel {
    const double *F, *Q, *H, *R, *Wp;       // row-major [n][n], [n][n], [m][n], [m][m]; the pair table (make_pair_table)
};

// the value lane o of the track's lane group holds (o is a constant wherever this is used, once the loops are unrolled);
// LN: lanes per track -- 4 (a quad: DPP) or 8 (the smoother at dim_x >= 13: half the unrolled arithmetic per lane)
template <int LN = 4, class Quad>
FK_HD double quad_from(Quad &quad, double v, int o)
{
    if (o == 0) return quad.template bcast<0>(v);
    if (o == 1) return quad.template bcast<1>(v);
    if (o == 2) return quad.template bcast<2>(v);
    if constexpr (LN == 4) return quad.template bcast<3>(v);
    else {
        if (o == 3) return quad.template bcast<3>(v);
        if (o == 4) return quad.template bcast<4>(v);
        if (o == 5) return quad.template bcast<5>(v);
        if (o == 6) return quad.template bcast<6>(v);
        return quad.template bcast<7>(v);
    }
}

// Lower factor of scale * P, rows distributed cyclically (g[r]: the row slot r holds -- q + LN r, clamped to NX - 1).
// Lw[r][k], k <= LN r + LN - 1: the lane's rows (zero above the diagonal; entries past that are never written nor read).
// row_done(j, lrow, ljj, inv): called once row j is final -- lrow[0..j-1], the pivot's root and its reciprocal, replicated in the quad.
template <int NX, int LN = 4, class Quad, class RowDone>
FK_HD bool quad_chol_rows(const double (&P)[(NX + LN - 1) / LN][NX], const unsigned (&g)[(NX + LN - 1) / LN], double scale,
                          double (&Lw)[(NX + LN - 1) / LN][NX], Quad &quad, RowDone &&row_done)
{
    constexpr int R = (NX + LN - 1) / LN;
    bool pd = true;
    FK_UNROLL for (int j = 0; j < NX; ++j) {
        const int sj = j / LN, oj = j % LN;
        // the pivot: every lane forms it for its row of slot sj, the owner's counts
        double dl = scale * P[sj][j];
        FK_UNROLL for (int k = 0; k < NX; ++k)
            if (k < j) dl = fma(-Lw[sj][k], Lw[sj][k], dl);
        const double d = quad_from<LN>(quad, dl, oj);
        pd = pd && (d > 0.0);
        double ljj, inv;
        sqrt_rsqrt(d, ljj, inv);
        double lrow[NX];
        FK_UNROLL for (int k = 0; k < NX; ++k)
            if (k < j) lrow[k] = quad_from<LN>(quad, Lw[sj][k], oj);
        FK_UNROLL for (int r = 0; r < R; ++r) {
            if (LN * r + LN - 1 < j) continue;                      // the slot's rows all lie above row j
            double t = scale * P[r][j];
            FK_UNROLL for (int k = 0; k < NX; ++k)
                if (k < j) t = fma(-Lw[r][k], lrow[k], t);
            t *= inv;
            if (LN * r > j) Lw[r][j] = t;                      // all of them below it
            else Lw[r][j] = g[r] > (unsigned)j ? t : (g[r] == (unsigned)j ? ljj : 0.0);
        }
        row_done(j, lrow, ljj, inv);
        FK_STAGE();
    }
    return pd;
}

// zin / has_z_fn(): the step's measurement and whether there is one -- both are first looked at in the UPDATE half (has_z_fn is
// called there), so a kernel that carries them as loads in flight waits for them a predict half after it asked.
template <int NX, int NZ, int LN = 4, class Quad, class HasZ>
FK_HD int ukf_quad_step_v4(double (&x)[NX], double (&P)[(NX + LN - 1) / LN][NX], const unsigned (&g)[(NX + LN - 1) / LN],
                           const double (&zin)[NZ], HasZ &&has_z_fn, double scale, const UkfQuadModel &mv, Quad &quad)
{
    constexpr int R = (NX + LN - 1) / LN;
    static_assert((LN == 4 || LN == 8) && NX >= LN && NZ >= 1 && NZ <= 8, "four or eight lanes per track, every one holding a row; dim_z <= 8 (at most two rows of H L per lane)");
    int st = 0;
    // ---------------- predict (UKF.py:400-411)
    {
        double FL[R][NX], Fxo[R];
        {
            double Lw[R][NX], fc[R];                          // fc: column j of F at the lane's rows, requested a column ahead
            FK_UNROLL for (int r = 0; r < R; ++r) fc[r] = mv.F[g[r] * NX];
            const bool pd = quad_chol_rows<NX, LN>(P, g, scale, Lw, quad, [&](int j, const double (&lrow)[NX], double ljj, double) {
                double f[R];
                FK_UNROLL for (int r = 0; r < R; ++r) f[r] = fc[r];
                if (j + 1 < NX) {
                    FK_UNROLL for (int r = 0; r < R; ++r) fc[r] = mv.F[g[r] * NX + j + 1];
                }
                FK_UNROLL for (int r = 0; r < R; ++r) {
                    Fxo[r] = (j == 0) ? f[r] * x[0] : fma(f[r], x[j], Fxo[r]);
                    FK_UNROLL for (int k = 0; k < NX; ++k)
                        if (k < j) FL[r][k] = fma(f[r], lrow[k], FL[r][k]);
                    FL[r][j] = f[r] * ljj;
                }
            });
            if (!pd) st |= ST_NOT_PD;
        }
        const double wms = mv.Wp[0], wcs = mv.Wp[1];
        // mean (replicated) and the centre point's offset
        {
            double wy[NX];
            FK_UNROLL for (int b = 0; b < NX; ++b) {
                const double fx = quad_from<LN>(quad, Fxo[b / LN], b % LN);
                x[b] = wms * fx;
                wy[b] = wcs * (fx - x[b]);
            }
            FK_UNROLL for (int r = 0; r < R; ++r) {
                const double xa = wms * Fxo[r];
                const double ya = Fxo[r] - xa;
                FK_UNROLL for (int b = 0; b < NX; ++b) P[r][b] = ya * wy[b];
            }
            FK_STAGE();
        }
        // the pairs: column k of F L, weighted by its owners, gathered
        FK_UNROLL for (int k = 0; k < NX; ++k) {
            const double wp = mv.Wp[2 + k];
            double wfo[R], wf[NX];
            FK_UNROLL for (int r = 0; r < R; ++r) wfo[r] = wp * FL[r][k];
            FK_UNROLL for (int b = 0; b < NX; ++b) wf[b] = quad_from<LN>(quad, wfo[b / LN], b % LN);
            FK_UNROLL for (int r = 0; r < R; ++r)
                FK_UNROLL for (int b = 0; b < NX; ++b) P[r][b] = fma(FL[r][k], wf[b], P[r][b]);
            FK_STAGE();
        }
        // + Q last, like the reference: the lane's rows of Q, each requested while the one before it is added (all of them at
        // once are R * NX more registers at the point where P-, F L and x are all live)
        {
            double Qr[2][NX];
            FK_UNROLL for (int b = 0; b < NX; ++b) Qr[0][b] = mv.Q[g[0] * NX + b];
            FK_UNROLL for (int r = 0; r < R; ++r) {
                if (r + 1 < R) {
                    FK_UNROLL for (int b = 0; b < NX; ++b) Qr[(r + 1) & 1][b] = mv.Q[g[r + 1] * NX + b];
                }
                FK_UNROLL for (int b = 0; b < NX; ++b) P[r][b] += Qr[r & 1][b];
                FK_STAGE();
            }
        }
    }
    // ---------------- update (UKF.py:462-481), sigma points regenerated from the prior (:407)
    {
        // H L, its rows dealt out like P's: lane q accumulates rows q and q + 4 (dim_z <= 8; a slot past the last row duplicates it)
        constexpr int RZ = (NZ + LN - 1) / LN;
        double Lw[R][NX], HLo[RZ][NX];
        unsigned hrow[RZ];
        bool upd_pd;
        FK_UNROLL for (int rz = 0; rz < RZ; ++rz) {
            const unsigned hr = g[0] + (unsigned)LN * (unsigned)rz;                  // g[0] = q (NX >= LN)
            hrow[rz] = hr < (unsigned)NZ ? hr : (unsigned)NZ - 1u;
        }
        {
            double hc[RZ];
            FK_UNROLL for (int rz = 0; rz < RZ; ++rz) hc[rz] = mv.H[hrow[rz] * NX];
            const bool pd = quad_chol_rows<NX, LN>(P, g, scale, Lw, quad, [&](int j, const double (&lrow)[NX], double ljj, double) {
                double h[RZ];
                FK_UNROLL for (int rz = 0; rz < RZ; ++rz) h[rz] = hc[rz];
                if (j + 1 < NX) {
                    FK_UNROLL for (int rz = 0; rz < RZ; ++rz) hc[rz] = mv.H[hrow[rz] * NX + j + 1];
                }
                FK_UNROLL for (int rz = 0; rz < RZ; ++rz) {
                    FK_UNROLL for (int k = 0; k < NX; ++k)
                        if (k < j) HLo[rz][k] = fma(h[rz], lrow[k], HLo[rz][k]);
                    HLo[rz][j] = h[rz] * ljj;
                }
            });
            upd_pd = pd;
        }
        const bool has_z = has_z_fn();
        if (!upd_pd && has_z) st |= ST_NOT_PD;
        if constexpr (NZ >= 5) {
            // dim_z 5..8: the dim_z x dim_z block DISTRIBUTED as well (replicated it is S, its factor and their temporaries: 80
            // doubles at dim_z 8 on top of the rows of P-, L and H L -- 2.3-3.7 KB of scratch per lane at dim_x 16).  Lane q holds
            // the rows of S that it holds of H L; S = Ls Ls' is factored like P (quad_chol_rows over dim_z rows), the forward
            // substitution W = Pxz Ls^-T rides on the broadcast rows, and with it
            //     P -= K S K'  =  P - W W'        (K = W Ls^-1, so K S K' = W Ls^-1 Ls Ls' Ls^-T W')
            // needs neither S again nor the products S K'; K = W Ls^-1 (column gathers) follows for x += K (z - zp).
            double So[RZ][NZ], Ko[R * NZ], zp[NZ];
            const double wms = mv.Wp[0], wcs = mv.Wp[1];
            {
                double wd[NZ];
                FK_UNROLL for (int c = 0; c < NZ; ++c) {
                    double acc = mv.H[c * NX] * x[0];
                    FK_UNROLL for (int k = 1; k < NX; ++k) acc = fma(mv.H[c * NX + k], x[k], acc);
                    zp[c] = wms * acc;
                    wd[c] = wcs * (acc - zp[c]);
                }
                FK_UNROLL for (int rz = 0; rz < RZ; ++rz) {                              // the lane's own rows: the same arithmetic
                    double acc = mv.H[hrow[rz] * NX] * x[0];
                    FK_UNROLL for (int k = 1; k < NX; ++k) acc = fma(mv.H[hrow[rz] * NX + k], x[k], acc);
                    const double d0o = acc - wms * acc;
                    FK_UNROLL for (int c = 0; c < NZ; ++c) So[rz][c] = d0o * wd[c];
                }
            }
            FK_STAGE();
            FK_UNROLL for (int k = 0; k < NX; ++k) {
                const double wp = mv.Wp[2 + k];
                double wh[NZ];
                FK_UNROLL for (int c = 0; c < NZ; ++c) wh[c] = wp * quad_from<LN>(quad, HLo[c / LN][k], c % LN);
                FK_UNROLL for (int rz = 0; rz < RZ; ++rz)
                    FK_UNROLL for (int c = 0; c < NZ; ++c) So[rz][c] = fma(HLo[rz][k], wh[c], So[rz][c]);
                FK_UNROLL for (int r = 0; r < R; ++r) {
                    if (LN * r + LN - 1 < k) continue;
                    FK_UNROLL for (int c = 0; c < NZ; ++c)
                        Ko[r * NZ + c] = (k == 0) ? Lw[r][0] * wh[c] : fma(Lw[r][k], wh[c], Ko[r * NZ + c]);
                }
                if (k % 4 == 3) FK_STAGE();
            }
            FK_UNROLL for (int rz = 0; rz < RZ; ++rz)
                FK_UNROLL for (int c = 0; c < NZ; ++c) So[rz][c] += mv.R[hrow[rz] * NZ + c];   // + R last
            double Ls[RZ][NZ], invz[NZ];
            {
                const bool pd = quad_chol_rows<NZ, LN>(So, hrow, 1.0, Ls, quad, [&](int j, const double (&lrow)[NZ], double, double inv) {
                    invz[j] = inv;
                    FK_UNROLL for (int r = 0; r < R; ++r) {                              // W[a][j] = (Pxz[a][j] - sum_{k<j} W[a][k] Ls[j][k]) / Ls[j][j]
                        double acc = Ko[r * NZ + j];
                        FK_UNROLL for (int k = 0; k < NZ; ++k)
                            if (k < j) acc = fma(-Ko[r * NZ + k], lrow[k], acc);
                        Ko[r * NZ + j] = acc * inv;
                    }
                });
                if (!pd && has_z) st |= ST_NOT_PD;
            }
            FK_UNROLL for (int e = 0; e < R * NZ; ++e) Ko[e] = has_z ? Ko[e] : 0.0;
            // P -= W W': row b of W from its owner
            FK_UNROLL for (int b = 0; b < NX; ++b) {
                double Wb[NZ];
                FK_UNROLL for (int c = 0; c < NZ; ++c) Wb[c] = quad_from<LN>(quad, Ko[(b / LN) * NZ + c], b % LN);
                FK_UNROLL for (int r = 0; r < R; ++r) {
                    double acc = Ko[r * NZ] * Wb[0];
                    FK_UNROLL for (int qq = 1; qq < NZ; ++qq) acc = fma(Ko[r * NZ + qq], Wb[qq], acc);
                    P[r][b] -= acc;
                }
                if (b % 4 == 3) FK_STAGE();
            }
            // K = W Ls^-1 in place: K[a][i] = (W[a][i] - sum_{k>i} K[a][k] Ls[k][i]) / Ls[i][i]
            FK_UNROLL for (int i = NZ - 1; i >= 0; --i) {
                double col[NZ];
                FK_UNROLL for (int k = 0; k < NZ; ++k)
                    if (k > i) col[k] = quad_from<LN>(quad, Ls[k / LN][i], k % LN);
                FK_UNROLL for (int r = 0; r < R; ++r) {
                    double acc = Ko[r * NZ + i];
                    FK_UNROLL for (int k = NZ - 1; k >= 0; --k)
                        if (k > i) acc = fma(-Ko[r * NZ + k], col[k], acc);
                    Ko[r * NZ + i] = acc * invz[i];
                }
            }
            // x += K (z - zp): own rows, gathered  (selected to zero again: a factor with a NaN in it -- S not positive definite --
            // turns the zero rows of a missing measurement into NaN in the substitution, and 0 * NaN would reach x)
            {
                double zc[NZ], xo[R];
                FK_UNROLL for (int c = 0; c < NZ; ++c) zc[c] = has_z ? zin[c] - zp[c] : 0.0;
                FK_UNROLL for (int r = 0; r < R; ++r) {
                    double acc = Ko[r * NZ] * zc[0];
                    FK_UNROLL for (int c = 1; c < NZ; ++c) acc = fma(Ko[r * NZ + c], zc[c], acc);
                    xo[r] = has_z ? acc : 0.0;
                }
                FK_UNROLL for (int b = 0; b < NX; ++b) x[b] += quad_from<LN>(quad, xo[b / LN], b % LN);
            }
        } else {
            double zp[NZ], S[NZ * NZ], Ko[R * NZ];
            const double wms = mv.Wp[0], wcs = mv.Wp[1];
            {
                double d0[NZ], wd[NZ];
                FK_UNROLL for (int c = 0; c < NZ; ++c) {
                    double acc = mv.H[c * NX] * x[0];
                    FK_UNROLL for (int k = 1; k < NX; ++k) acc = fma(mv.H[c * NX + k], x[k], acc);
                    zp[c] = wms * acc;
                    d0[c] = acc - zp[c];
                }
                FK_UNROLL for (int c = 0; c < NZ; ++c) wd[c] = wcs * d0[c];
                FK_UNROLL for (int r = 0; r < NZ; ++r)
                    FK_UNROLL for (int c = 0; c < NZ; ++c)
                        if (c >= r) S[r * NZ + c] = d0[r] * wd[c];
            }
            FK_STAGE();
            FK_UNROLL for (int k = 0; k < NX; ++k) {
                const double wp = mv.Wp[2 + k];
                double hl[NZ], wh[NZ];                             // column k of H L, gathered from the lanes that hold its rows
                FK_UNROLL for (int c = 0; c < NZ; ++c) hl[c] = quad_from<LN>(quad, HLo[c / LN][k], c % LN);
                FK_UNROLL for (int c = 0; c < NZ; ++c) wh[c] = wp * hl[c];
                FK_UNROLL for (int r = 0; r < NZ; ++r)
                    FK_UNROLL for (int c = 0; c < NZ; ++c)
                        if (c >= r) S[r * NZ + c] = fma(hl[r], wh[c], S[r * NZ + c]);
                // the cross variance's own rows: l_k is zero above the diagonal (slots whose rows all lie above row k: skipped;
                // a row above it inside a slot: its element of L is a stored zero)
                FK_UNROLL for (int r = 0; r < R; ++r) {
                    if (LN * r + LN - 1 < k) continue;
                    FK_UNROLL for (int c = 0; c < NZ; ++c)
                        Ko[r * NZ + c] = (k == 0) ? Lw[r][0] * wh[c] : fma(Lw[r][k], wh[c], Ko[r * NZ + c]);
                }
                if (k % 4 == 3) FK_STAGE();
            }
            // S stays an UPPER triangle (its mirror image is the same bits: the reference's w * outer(d, d) is symmetric bit for
            // bit, and so is R as far as anybody reads it); the factorisation reads a lower one: its transpose.  At dim_z 8 the
            // full S next to a full copy for the factor was 56 doubles more at the point where P-, K and x are live.
            // (R is requested here, not at the head of the half: its upper triangle is dim_z (dim_z + 1) / 2 more doubles to hold
            //  through the sweep and the pairs -- 36 at dim_z 8 -- for one LDS round trip saved)
            FK_UNROLL for (int r = 0; r < NZ; ++r)
                FK_UNROLL for (int c = 0; c < NZ; ++c)
                    if (c >= r) S[r * NZ + c] += mv.R[r * NZ + c];                  // + R last
            // K = Pxz S^-1 (own rows)
            {
                double Lf[NZ * NZ], dd[NZ], dinv[NZ];
                FK_UNROLL for (int r = 0; r < NZ; ++r)
                    FK_UNROLL for (int c = 0; c < NZ; ++c)
                        if (c <= r) Lf[r * NZ + c] = S[c * NZ + r];
                if (!ldlt2_rs<NZ>(Lf, dd, dinv) && has_z) st |= ST_NOT_PD;
                solve_rows_ldlt<R, NZ>(Lf, dinv, Ko);
            }
            double zc[NZ];
            FK_UNROLL for (int e = 0; e < R * NZ; ++e) Ko[e] = has_z ? Ko[e] : 0.0;
            FK_UNROLL for (int r = 0; r < NZ; ++r)
                FK_UNROLL for (int c = 0; c < NZ; ++c)
                    if (c >= r) S[r * NZ + c] = has_z ? S[r * NZ + c] : 0.0;
            FK_UNROLL for (int c = 0; c < NZ; ++c) zc[c] = has_z ? zin[c] - zp[c] : 0.0;
            FK_STAGE();
            // x += K (z - zp) ; P -= K (S K') : row b of K from its owner serves x[b] and column b of P
            FK_UNROLL for (int b = 0; b < NX; ++b) {
                double Kb[NZ], sk[NZ];
                FK_UNROLL for (int c = 0; c < NZ; ++c) Kb[c] = quad_from<LN>(quad, Ko[(b / LN) * NZ + c], b % LN);
                {
                    double acc = Kb[0] * zc[0];
                    FK_UNROLL for (int c = 1; c < NZ; ++c) acc = fma(Kb[c], zc[c], acc);
                    x[b] += acc;
                }
                FK_UNROLL for (int qq = 0; qq < NZ; ++qq) {
                    double acc = S[0 * NZ + qq] * Kb[0];                             // S[qq][0] = S[0][qq]
                    FK_UNROLL for (int w = 1; w < NZ; ++w) acc = fma(w >= qq ? S[qq * NZ + w] : S[w * NZ + qq], Kb[w], acc);
                    sk[qq] = acc;
                }
                FK_UNROLL for (int r = 0; r < R; ++r) {
                    double acc = Ko[r * NZ] * sk[0];
                    FK_UNROLL for (int qq = 1; qq < NZ; ++qq) acc = fma(Ko[r * NZ + qq], sk[qq], acc);
                    P[r][b] -= acc;
                }
                if (b % 4 == 3) FK_STAGE();
            }
        }
    }
    return st;
}

// One backward step of UnscentedKalmanFilter.rts_smoother with fx(x, dt) = F x (UKF.py:714-737) on four lanes per track:
// ukf_linear_rts_gain_v4 + ukf_linear_rts_correct (fk_ukf.hpp) distributed like the filter step above, every sum in their order:
//   sweep      the factor of scale * Ps[k] column by column; F L and F x (own rows) from its broadcast rows
//   Pxb, Pb    two passes over the gathered, weighted columns of F L (one pass would hold both accumulators next to F L and
//              the factor: 232 doubles at dim_x 16): Pxb = sum_k wp_k l_k f_k' with the lane's own rows of L, then
//              Pb = (sum Wc) y0 y0' + sum_k wp_k f_k f_k' + Q
//   K          K = Pxb Pb^-1 with Pb = Lb Lb' (a Cholesky factor here, where v4 takes L D L': the same solve to a rounding):
//              the FORWARD substitution w Lb' = Pxb[a] rides on the rows of Lb the factorisation broadcasts anyway, the
//              backward one K[a] Lb = w gathers Lb's columns from their owners
//   correct    x += K (xn - xb) (own rows, gathered);  T1 = K (Pn - Pb): the rows of D = Pn - Pb are formed by their owners a
//              slot at a time and broadcast;  P += T1 K': K's rows gathered.  Full rows (v4: the upper triangle), so a row of
//              the result and its mirror image agree to a rounding.
// What the step keeps OUT of the registers: the smoothed covariance of step k+1 (next_row(r, out): the caller's copy of the
// lane's row of slot r -- the kernel's output tile still holds it) and the full rows of Ps[k] (own_row(r, out): requested
// again for the correction; the sweep only needs P[r][c], c <= 4 r + 3).
// Likewise the two means: the sweep spends x[j] at column j, so the correction asks for Xs[k] again (own_x) and for xs[k+1]
// (next_x) instead of carrying 2 n doubles through the passes.
// x: Xs[k] in, xs[k] out (replicated);  P: rows of Ps[k] in (lower part), of ps[k] out;  K: the lane's rows of the gain out.
// io: where the step's neighbours and its parked intermediates live --
//   io.next_x(out), io.next_row(r, out)   xs[k+1] (replicated) and the lane's row of slot r of ps[k+1]
//   io.own_x(out),  io.own_row(r, out)    Xs[k] and the lane's full row of slot r of Ps[k], again
//   Io::PARK (dim_x >= 13 in the kernel): a covariance-sized parking lot per track.  Pxb waits there through the Pb pass
//   (park_k / unpark_k), then Pb's full rows until the correction reads them back a slot at a time (park_pb / pb_row): without
//   it the second pass holds F L, Pb and Pxb -- 192 doubles -- and the step spills 2.4 KB per lane at dim_x 16, every reload a
//   vmcnt(0).  (The kernel's lot is its output tile, so with PARK the smoothed covariance of step k+1 comes from memory.)
template <int NX, int LN = 4, class Quad, class Io>
FK_HD int ukf_quad_rts_step_v4(double (&x)[NX], double (&P)[(NX + LN - 1) / LN][NX], const unsigned (&g)[(NX + LN - 1) / LN],
                               double scale, const UkfQuadModel &mv, Quad &quad, double (&K)[(NX + LN - 1) / LN][NX], Io &io)
{
    constexpr int R = (NX + LN - 1) / LN;
    static_assert((LN == 4 || LN == 8) && NX >= LN, "four or eight lanes per track, every one of them holding a row");
    int st = 0;
    double Pb[R][NX], xb[NX];
    constexpr bool FUSE = LN == 8 && !Io::PARK;
    {
        double FL[R][NX], Fxo[R];
        // xb (replicated), the centre point's offset, Pb's first term -- once the sweep has F x
        auto init_pb = [&] {
            const double wms = mv.Wp[0], wcs = mv.Wp[1];
            double wy[NX];
            FK_UNROLL for (int b = 0; b < NX; ++b) {
                const double fx = quad_from<LN>(quad, Fxo[b / LN], b % LN);
                xb[b] = wms * fx;
                wy[b] = wcs * (fx - xb[b]);
            }
            FK_UNROLL for (int r = 0; r < R; ++r) {
                const double xa = wms * Fxo[r];
                const double ya = Fxo[r] - xa;
                FK_UNROLL for (int b = 0; b < NX; ++b) Pb[r][b] = ya * wy[b];
            }
            FK_STAGE();
        };
        {
            double Lw[R][NX];
            {
                double fc[R];
                FK_UNROLL for (int r = 0; r < R; ++r) fc[r] = mv.F[g[r] * NX];
                const bool pd = quad_chol_rows<NX, LN>(P, g, scale, Lw, quad, [&](int j, const double (&lrow)[NX], double ljj, double) {
                    double f[R];
                    FK_UNROLL for (int r = 0; r < R; ++r) f[r] = fc[r];
                    if (j + 1 < NX) {
                        FK_UNROLL for (int r = 0; r < R; ++r) fc[r] = mv.F[g[r] * NX + j + 1];
                    }
                    FK_UNROLL for (int r = 0; r < R; ++r) {
                        Fxo[r] = (j == 0) ? f[r] * x[0] : fma(f[r], x[j], Fxo[r]);
                        FK_UNROLL for (int k = 0; k < NX; ++k)
                            if (k < j) FL[r][k] = fma(f[r], lrow[k], FL[r][k]);
                        FL[r][j] = f[r] * ljj;
                    }
                });
                if (!pd) st |= ST_NOT_PD;
            }
            // Pxb = sum_k wp_k l_k f_k' (own rows of L; l_k is zero above the diagonal) -> K;  with eight lanes per track (two
            // row slots: every accumulator fits) Pb takes its rank-one terms from the same gathered columns in the same pass
            if constexpr (FUSE) init_pb();
            FK_UNROLL for (int k = 0; k < NX; ++k) {
                const double wp = mv.Wp[2 + k];
                double wfo[R], wf[NX];
                FK_UNROLL for (int r = 0; r < R; ++r) wfo[r] = wp * FL[r][k];
                FK_UNROLL for (int b = 0; b < NX; ++b) wf[b] = quad_from<LN>(quad, wfo[b / LN], b % LN);
                FK_UNROLL for (int r = 0; r < R; ++r) {
                    if (LN * r + LN - 1 < k) continue;
                    FK_UNROLL for (int b = 0; b < NX; ++b) K[r][b] = (k == 0) ? Lw[r][0] * wf[b] : fma(Lw[r][k], wf[b], K[r][b]);
                }
                if constexpr (FUSE) {
                    FK_UNROLL for (int r = 0; r < R; ++r)
                        FK_UNROLL for (int b = 0; b < NX; ++b) Pb[r][b] = fma(FL[r][k], wf[b], Pb[r][b]);
                }
                FK_STAGE();
            }
        }
        if constexpr (Io::PARK) io.park_k(K);
        if constexpr (!FUSE) {
            init_pb();
            FK_UNROLL for (int k = 0; k < NX; ++k) {
                const double wp = mv.Wp[2 + k];
                double wfo[R], wf[NX];
                FK_UNROLL for (int r = 0; r < R; ++r) wfo[r] = wp * FL[r][k];
                FK_UNROLL for (int b = 0; b < NX; ++b) wf[b] = quad_from<LN>(quad, wfo[b / LN], b % LN);
                FK_UNROLL for (int r = 0; r < R; ++r)
                    FK_UNROLL for (int b = 0; b < NX; ++b) Pb[r][b] = fma(FL[r][k], wf[b], Pb[r][b]);
                FK_STAGE();
            }
        }
        {
            double Qr[2][NX];
            FK_UNROLL for (int b = 0; b < NX; ++b) Qr[0][b] = mv.Q[g[0] * NX + b];
            FK_UNROLL for (int r = 0; r < R; ++r) {
                if (r + 1 < R) {
                    FK_UNROLL for (int b = 0; b < NX; ++b) Qr[(r + 1) & 1][b] = mv.Q[g[r + 1] * NX + b];
                }
                FK_UNROLL for (int b = 0; b < NX; ++b) Pb[r][b] += Qr[r & 1][b];
                FK_STAGE();
            }
        }
    }
    if constexpr (Io::PARK) {
        io.unpark_k(K);
        io.park_pb(Pb);
    }
    // ---------------- K = Pxb Pb^-1
    {
        double Lb[R][NX], invd[NX];
        const bool pd = quad_chol_rows<NX, LN>(Pb, g, 1.0, Lb, quad, [&](int j, const double (&lrow)[NX], double, double inv) {
            invd[j] = inv;
            // forward substitution, column j of every own row: w[j] = (Pxb[a][j] - sum_{k<j} w[k] Lb[j][k]) / Lb[j][j]
            FK_UNROLL for (int r = 0; r < R; ++r) {
                double acc = K[r][j];
                FK_UNROLL for (int k = 0; k < NX; ++k)
                    if (k < j) acc = fma(-K[r][k], lrow[k], acc);
                K[r][j] = acc * inv;
            }
        });
        if (!pd) st |= ST_NOT_PD;
        // backward substitution: K[a][i] = (w[i] - sum_{k>i} K[a][k] Lb[k][i]) / Lb[i][i]
        FK_UNROLL for (int i = NX - 1; i >= 0; --i) {
            double col[NX];
            FK_UNROLL for (int k = 0; k < NX; ++k)
                if (k > i) col[k] = quad_from<LN>(quad, Lb[k / LN][i], k % LN);
            FK_UNROLL for (int r = 0; r < R; ++r) {
                double acc = K[r][i];
                FK_UNROLL for (int k = NX - 1; k >= 0; --k)
                    if (k > i) acc = fma(-K[r][k], col[k], acc);
                K[r][i] = acc * invd[i];
            }
            if (i % 4 == 0) FK_STAGE();
        }
    }
    // ---------------- correct: x += K (xn - xb)
    {
        double dx[NX], xo[R];
        io.next_x(dx);
        FK_UNROLL for (int c = 0; c < NX; ++c) dx[c] -= xb[c];
        FK_UNROLL for (int r = 0; r < R; ++r) {
            double acc = K[r][0] * dx[0];
            FK_UNROLL for (int c = 1; c < NX; ++c) acc = fma(K[r][c], dx[c], acc);
            xo[r] = acc;
        }
        io.own_x(x);
        FK_UNROLL for (int b = 0; b < NX; ++b) x[b] += quad_from<LN>(quad, xo[b / LN], b % LN);
    }
    FK_STAGE();
    // (eight lanes: the full rows of Ps[k] are requested HERE, a phase ahead of their use -- two row slots per lane leave the room)
    if constexpr (LN == 8) {
        FK_UNROLL for (int r = 0; r < R; ++r) io.own_row(r, P[r]);
    }
    // T1 = K D, D = Pn - Pb: the rows of D a slot at a time from their owners
    double T1[R][NX];
    {
        double Pn[2][NX], Pq[Io::PARK ? 2 : 1][NX];
        io.next_row(0, Pn[0]);
        if constexpr (Io::PARK) io.pb_row(0, Pq[0]);
        FK_UNROLL for (int s = 0; s < R; ++s) {
            if (s + 1 < R) {
                io.next_row(s + 1, Pn[(s + 1) & 1]);
                if constexpr (Io::PARK) io.pb_row(s + 1, Pq[(s + 1) & 1]);
            }
            double Ds[NX];
            FK_UNROLL for (int c = 0; c < NX; ++c) Ds[c] = Pn[s & 1][c] - (Io::PARK ? Pq[Io::PARK ? (s & 1) : 0][c] : Pb[s][c]);
            FK_UNROLL for (int q = 0; q < LN; ++q) {
                const int b = LN * s + q;
                if (b >= NX) continue;
                double Db[NX];
                FK_UNROLL for (int c = 0; c < NX; ++c) Db[c] = quad_from<LN>(quad, Ds[c], q);
                FK_UNROLL for (int r = 0; r < R; ++r)
                    FK_UNROLL for (int c = 0; c < NX; ++c) T1[r][c] = (b == 0) ? K[r][0] * Db[c] : fma(K[r][b], Db[c], T1[r][c]);
            }
            FK_STAGE();
        }
    }
    // P += T1 K': row j of K from its owner
    if constexpr (LN != 8) {
        FK_UNROLL for (int r = 0; r < R; ++r) io.own_row(r, P[r]);
    }
    FK_UNROLL for (int j = 0; j < NX; ++j) {
        double Kj[NX];
        FK_UNROLL for (int c = 0; c < NX; ++c) Kj[c] = quad_from<LN>(quad, K[j / LN][c], j % LN);
        FK_UNROLL for (int r = 0; r < R; ++r) {
            double acc = T1[r][0] * Kj[0];
            FK_UNROLL for (int c = 1; c < NX; ++c) acc = fma(T1[r][c], Kj[c], acc);
            P[r][j] += acc;
        }
        if (j % 4 == 3) FK_STAGE();
    }
    return st;
}

}  // namespace fk
