// ukf_kernels.hip -- fused linear-model unscented Kalman filter for gfx950 (MI355X).
//
//   fk_ukf_linear_batch_f64 <- UnscentedKalmanFilter.batch_filter (filterpy/kalman/UKF.py:524-632) with
//                              fx(x, dt) = F x, hx(x) = H x: the whole predict (UKF.py:400-411) / update
//                              (:462-481) loop stays in registers over the time steps.
// (Split from ut_kernels.hip so the two translation units compile in parallel.)
#include <stdlib.h>

#include <type_traits>

#include "../../include/filterhip.h"
#include "fk_chunks.hpp"
#include "fk_device.hpp"
#include "fk_kernel_args.hpp"
#include "fk_math_sym.hpp"
#include "fk_ukf.hpp"

namespace fk {

// --------------------------------------------------- fused linear-model UKF --
// Per step (UKF.py:400-411, 462-481) with fx(x) = F x, hx(x) = H x:
//   L  = chol(scale P);  sigma_i = x, x +- L[:,k]          (sigma_points.py:167-175)
//   sf_i = F sigma_i ;  (x,P) = UT(sf, Wm, Wc, Q)
//   L  = chol(scale P);  sf_i = sigma_i(x,P)  (regenerated, UKF.py:407)
//   sh_i = H sf_i ; (zp,S) = UT(sh, Wm, Wc, R) ; Pxz = sum Wc_i (sf_i-x)(sh_i-zp)'
//   K = Pxz S^-1 ; x += K (z-zp) ; P -= K (S K')
// The covariance and both Cholesky factors are packed triangles, and the propagated sigma points are
// never all materialised: the predict makes two sweeps over the 2n+1 points (mean, then covariance),
// regenerating each point x +- L[:,k] and pushing it through F on the fly -- ~70 live doubles at
// n = 6 instead of ~260 (the first version spilled 196 registers at one wave per SIMD).
// Two organisations of the kernel.  ukf_linear_kernel is the straightforward one; ukf_linear_kernel_v2 is the same
// arithmetic -- every sum accumulates over the sigma points in the same index order -- reorganised for registers
// (step arithmetic in fk_ukf.hpp, held against the oracle on the host; on the GPU it agrees with the first to 2e-13
// and with the oracle to 1e-13, profiles/r02/exp_ukf2.log):
//   * the mean sweeps run point by point like the covariance sweeps (row by row, the compiler kept all
//     (2n+1) n sigma-point values alive across the rows);
//   * the measurement update is two sweeps (zp, then S and Pxz) instead of holding all H sigma_i;
//   * every unrolled point re-reads its model rows through an LDS offset the optimiser cannot see through
//     (otherwise the broadcast reads of all points are hoisted and held);
//   => (6,3): 223 (SOA) / 256 (AOS) VGPRs, no scratch, two waves per SIMD, against 512 VGPRs + 44 spilled
//      registers at one wave per SIMD.
//   * SCALAR_FH (exact dims only): the rows of F and H do not come from LDS at all but from the kernel's uniform
//     pointers through the scalar cache into SGPRs, fetched once per pass over the sigma points and used as the
//     scalar operand of the FMAs.  A broadcast ds_read still delivers 64 x 16 bytes through the CU's one 128 B/clk
//     LDS port: 716 ds_read2_b64 per step and wave x 8 waves per CU was 46k LDS clocks per step against 30k VALU
//     clocks per SIMD -- the kernel was LDS-bound, not VALU-bound.
template <int NX, int NZ>
struct ScalarFHModel {
    const double *s;          // LDS model: Q, R (read once per step)
    const double *gF, *gH;    // uniform global pointers
    __device__ __forceinline__ void rowF(int i, double (&r)[NX]) const { FK_UNROLL for (int j = 0; j < NX; ++j) r[j] = gF[i * NX + j]; }
    __device__ __forceinline__ void rowH(int i, double (&r)[NX]) const { FK_UNROLL for (int j = 0; j < NX; ++j) r[j] = gH[i * NX + j]; }
    __device__ __forceinline__ void rowQ(int i, double (&r)[NX]) const { LdsModel<NX, NZ>{s}.rowQ(i, r); }
    __device__ __forceinline__ void rowR(int i, double (&r)[NZ]) const { LdsModel<NX, NZ>{s}.rowR(i, r); }
};

template <int NX, int NZ, int LAYOUT, bool SCALAR_FH, int VER = 3>
__global__ void __launch_bounds__(BLOCK, (NX <= 2 ? 4 : NX <= 6 ? 2 : 1))
ukf_linear_kernel_v2(const UkfArgs a, const double *__restrict__ pF, const double *__restrict__ pH,
                  const double *__restrict__ pQ, const double *__restrict__ pR,
                  const double *__restrict__ pWm, const double *__restrict__ pWc,
                  const double *__restrict__ pz, const uint8_t *__restrict__ pmask)
{
    constexpr int KS = 2 * NX + 1;
    constexpr int PL = NX * (NX + 1) / 2;
    using SharedModel = LdsModel<NX, NZ>;
    __shared__ double s_model[SharedModel::SIZE + 2 * KS];
    // NumPy order at the exact dims: the per-step outputs leave through a wave-private LDS tile, 1 KiB contiguous per store
    // instruction (wave_store_aos, fk_device.hpp) -- a lane-per-record store touches 64 lines per instruction, and at V3's
    // instruction count the 42 stores of a (6,3) step would make the address path, not the VALU, the bound.  (dim_x 9: the
    // tile of an 81-double record does not fit next to four waves.)
    constexpr bool COOP = LAYOUT == LAYOUT_AOS && NX <= 8 && NX % 2 == 0 && VER == 3;
    constexpr int TILE = 64 * NX * NX;                       // flat: wave_store_aos_flat
    __shared__ double s_tile[COOP ? (BLOCK / 64) * TILE : 1];
    const long N = a.N;
    const long blk0 = a.i0 + (long)blockIdx.x * BLOCK;
    const Lane ln{blk0, threadIdx.x, N};
    const bool live = blk0 + ln.tid < a.i0 + a.cnt;
    const Lane lr{blk0, live ? ln.tid : 0u, N};
    const int n = a.n, m = a.m;
    const int ks = 2 * n + 1;
    const bool coop = COOP && n == NX;
    double *tile = s_tile + (COOP ? (threadIdx.x >> 6) * TILE : 0);
    const unsigned lane = threadIdx.x & 63u, wave_row0 = (threadIdx.x >> 6) * 64u;
    const long left = a.i0 + a.cnt - blk0;
    const unsigned last_row = (unsigned)(left < BLOCK ? left : BLOCK) - 1u;

    lds_fill<NX, NX>(s_model + SharedModel::OFF_F, pF, n, n, 1.0, threadIdx.x);
    lds_fill<NX, NX>(s_model + SharedModel::OFF_Q, pQ, n, n, 1.0, threadIdx.x);   // padded block of P stays I
    lds_fill<NZ, NX>(s_model + SharedModel::OFF_H, pH, m, n, 0.0, threadIdx.x);
    lds_fill<NZ, NZ>(s_model + SharedModel::OFF_R, pR, m, m, 1.0, threadIdx.x);
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
    // a view of the LDS model the optimiser cannot relate to the previous one: keeps it from hoisting the
    // broadcast row reads of all 2n+1 unrolled points to the top (they are cheap to repeat, dear to hold)
    // (an offset is made opaque, not the pointer: that keeps the LDS address space)
    using StepModel = std::conditional_t<SCALAR_FH, ScalarFHModel<NX, NZ>, SharedModel>;
    struct View { StepModel sm; const double *Wm, *Wc; };
    int goff = 0;                                           // wave-uniform, re-made opaque at the head of every pass
    auto sweep = [&]() {
        if constexpr (SCALAR_FH) {
            int t;
            asm volatile("s_mov_b32 %0, 0" : "=s"(t));
            goff = t;
        }
    };
    auto fresh = [&]() {
        int off = 0;
        asm volatile("" : "+v"(off));
        const double *mb = s_model + off;
        if constexpr (SCALAR_FH) return View{StepModel{mb, pF + goff, pH + goff}, mb + SharedModel::SIZE, mb + SharedModel::SIZE + KS};
        else return View{StepModel{mb}, mb + SharedModel::SIZE, mb + SharedModel::SIZE + KS};
    };

    double x[NX], P[PL];
    load_rec<NX, 1, LAYOUT, false>(x, a.x, lr, n, 1, 0.0);
    {
        const RecView<LAYOUT> pv(a.P, lr, n * n);
        FK_UNROLL for (int i = 0; i < NX; ++i)
            FK_UNROLL for (int j = 0; j < NX; ++j)
                if (j >= i) P[sym_idx<NX>(i, j)] = (i < n && j < n) ? pv.load(i * n + j) : (i == j ? 1.0 : 0.0);
    }
    int st = 0;

    for (long t = 0; t < a.T; ++t) {
        double z[NZ];
        bool has_z = true;
        if (pmask) has_z = pmask[t * N + lr.blk0 + lr.tid] != 0;
        load_rec<NZ, 1, LAYOUT, false>(z, pz + t * N * m, lr, m, 1, 0.0);

        if constexpr (VER == 3) st |= ukf_linear_step_v3<NX, NZ>(x, P, z, has_z, a.scale, fresh, sweep);
        else st |= ukf_linear_step_v2<NX, NZ>(x, P, z, has_z, a.scale, fresh, sweep);
        if (COOP && coop) {
            if constexpr (COOP) {
                if (a.means) wave_store_aos_flat<NX>(x, a.means + t * N * NX + blk0 * NX, wave_row0, tile, lane, last_row);
                if (a.covs) {
                    double Pf[NX * NX];
                    FK_UNROLL for (int i = 0; i < NX; ++i)
                        FK_UNROLL for (int j = 0; j < NX; ++j) Pf[i * NX + j] = P[sym_idx<NX>(i, j)];
                    wave_store_aos_flat<NX * NX>(Pf, a.covs + t * N * (NX * NX) + blk0 * (NX * NX), wave_row0, tile, lane, last_row);
                }
            }
        } else if (live) {
            if (a.means) store_rec<NX, 1, LAYOUT, false>(x, a.means + t * N * n, ln, n, 1);
            if (a.covs) {
                double Pf[NX * NX];
                FK_UNROLL for (int i = 0; i < NX; ++i)
                    FK_UNROLL for (int j = 0; j < NX; ++j) Pf[i * NX + j] = P[sym_idx<NX>(i, j)];
                store_rec<NX, NX, LAYOUT, false>(Pf, a.covs + t * N * n * n, ln, n, n);
            }
        }
    }
    if (live) {
        store_rec<NX, 1, LAYOUT, false>(x, a.x, ln, n, 1);
        double Pf[NX * NX];
        FK_UNROLL for (int i = 0; i < NX; ++i)
            FK_UNROLL for (int j = 0; j < NX; ++j) Pf[i * NX + j] = P[sym_idx<NX>(i, j)];
        store_rec<NX, NX, LAYOUT, false>(Pf, a.P, ln, n, n);
        if (a.status) {
            if (!all_finite<NX>(x) || !all_finite<PL>(P)) st |= ST_NONFINITE;
            a.status[ln.blk0 + ln.tid] = a.status_or ? (a.status[ln.blk0 + ln.tid] | st) : st;
        }
    }
}
template <int NX, int NZ, int LAYOUT>
__global__ void __launch_bounds__(BLOCK, (NX <= 2 ? 4 : NX <= 4 ? 2 : 1))
ukf_linear_kernel(const UkfArgs a, const double *__restrict__ pF, const double *__restrict__ pH,
                  const double *__restrict__ pQ, const double *__restrict__ pR,
                  const double *__restrict__ pWm, const double *__restrict__ pWc,
                  const double *__restrict__ pz, const uint8_t *__restrict__ pmask)
{
    constexpr int KS = 2 * NX + 1;
    constexpr int PL = NX * (NX + 1) / 2;
    using SharedModel = LdsModel<NX, NZ>;
    __shared__ double s_model[SharedModel::SIZE + 2 * KS];
    const long N = a.N;
    const long blk0 = a.i0 + (long)blockIdx.x * BLOCK;
    const Lane ln{blk0, threadIdx.x, N};
    const bool live = blk0 + ln.tid < a.i0 + a.cnt;
    const Lane lr{blk0, live ? ln.tid : 0u, N};
    const int n = a.n, m = a.m;
    const int ks = 2 * n + 1;

    lds_fill<NX, NX>(s_model + SharedModel::OFF_F, pF, n, n, 1.0, threadIdx.x);
    lds_fill<NX, NX>(s_model + SharedModel::OFF_Q, pQ, n, n, 1.0, threadIdx.x);   // padded block of P stays I
    lds_fill<NZ, NX>(s_model + SharedModel::OFF_H, pH, m, n, 0.0, threadIdx.x);
    lds_fill<NZ, NZ>(s_model + SharedModel::OFF_R, pR, m, m, 1.0, threadIdx.x);
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

    double x[NX], P[PL];
    load_rec<NX, 1, LAYOUT, false>(x, a.x, lr, n, 1, 0.0);
    {
        const RecView<LAYOUT> pv(a.P, lr, n * n);
        FK_UNROLL for (int i = 0; i < NX; ++i)
            FK_UNROLL for (int j = 0; j < NX; ++j)
                if (j >= i) P[sym_idx<NX>(i, j)] = (i < n && j < n) ? pv.load(i * n + j) : (i == j ? 1.0 : 0.0);
    }
    int st = 0;

    for (long t = 0; t < a.T; ++t) {
        double z[NZ];
        bool has_z = true;
        if (pmask) has_z = pmask[t * N + lr.blk0 + lr.tid] != 0;
        load_rec<NZ, 1, LAYOUT, false>(z, pz + t * N * m, lr, m, 1, 0.0);

        // ---------------- predict (UKF.py:400-411)
        double L[PL];
        if (!chol_packed<NX>(P, a.scale, L)) st |= ST_NOT_PD;
        // sweep 1: x- = sum_i Wm_i F sigma_i, one output component (row of F) at a time, points in
        // index order 0, x + L[:,k] (k = 0..n-1), x - L[:,k]
        double xm[NX];
        FK_UNROLL for (int r = 0; r < NX; ++r) {
            double f[NX];
            sm.rowF(r, f);
            double acc = sWm[0] * dot<NX>(f, x);
            FK_UNROLL for (int k = 0; k < NX; ++k) {
                double v = f[0] * (x[0] - (-lcol<NX>(L, 0, k)));
                FK_UNROLL for (int c = 1; c < NX; ++c) v = fma(f[c], x[c] - (-lcol<NX>(L, c, k)), v);
                acc = fma(sWm[1 + k], v, acc);
            }
            FK_UNROLL for (int k = 0; k < NX; ++k) {
                double v = f[0] * (x[0] - lcol<NX>(L, 0, k));
                FK_UNROLL for (int c = 1; c < NX; ++c) v = fma(f[c], x[c] - lcol<NX>(L, c, k), v);
                acc = fma(sWm[1 + NX + k], v, acc);
            }
            xm[r] = acc;
            FK_STAGE();
        }
        // sweep 2: P- = sum_i Wc_i y_i y_i' + Q, y_i = F sigma_i - x-   (upper triangle).
        // The points are recomputed from copies the optimiser cannot relate to sweep 1 (otherwise it
        // common-subexpression-eliminates the recomputation by keeping all (2n+1) n values alive).
        double Pn[PL];
        FK_UNROLL for (int c = 0; c < NX; ++c) asm volatile("" : "+v"(x[c]));
        FK_UNROLL for (int e = 0; e < PL; ++e) asm volatile("" : "+v"(L[e]));
        FK_UNROLL for (int i = 0; i < KS; ++i) {
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
            if (!chol_packed<NX>(P, a.scale, L)) st |= ST_NOT_PD;
            double h0[NZ], hp[NZ * NX], hm[NZ * NX];
            FK_UNROLL for (int r = 0; r < NZ; ++r) {
                double h[NX];
                sm.rowH(r, h);
                h0[r] = dot<NX>(h, x);
                FK_UNROLL for (int k = 0; k < NX; ++k) {
                    double vp = h[0] * (x[0] - (-lcol<NX>(L, 0, k))), vm = h[0] * (x[0] - lcol<NX>(L, 0, k));
                    FK_UNROLL for (int c = 1; c < NX; ++c) {
                        vp = fma(h[c], x[c] - (-lcol<NX>(L, c, k)), vp);
                        vm = fma(h[c], x[c] - lcol<NX>(L, c, k), vm);
                    }
                    hp[r * NX + k] = vp;
                    hm[r * NX + k] = vm;
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
                        const double dp = (x[r] - (-lcol<NX>(L, r, k))) - x[r];
                        acc += sWc[1 + k] * (dp * hp[c * NX + k]);
                    }
                    FK_UNROLL for (int k = 0; k < NX; ++k) {
                        const double dm = (x[r] - lcol<NX>(L, r, k)) - x[r];
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
            }
        }
        if (live) {
            if (a.means) store_rec<NX, 1, LAYOUT, false>(x, a.means + t * N * n, ln, n, 1);
            if (a.covs) {
                double Pf[NX * NX];
                FK_UNROLL for (int i = 0; i < NX; ++i)
                    FK_UNROLL for (int j = 0; j < NX; ++j) Pf[i * NX + j] = P[sym_idx<NX>(i, j)];
                store_rec<NX, NX, LAYOUT, false>(Pf, a.covs + t * N * n * n, ln, n, n);
            }
        }
    }
    if (live) {
        store_rec<NX, 1, LAYOUT, false>(x, a.x, ln, n, 1);
        double Pf[NX * NX];
        FK_UNROLL for (int i = 0; i < NX; ++i)
            FK_UNROLL for (int j = 0; j < NX; ++j) Pf[i * NX + j] = P[sym_idx<NX>(i, j)];
        store_rec<NX, NX, LAYOUT, false>(Pf, a.P, ln, n, n);
        if (a.status) {
            if (!all_finite<NX>(x) || !all_finite<PL>(P)) st |= ST_NONFINITE;
            a.status[ln.blk0 + ln.tid] = a.status_or ? (a.status[ln.blk0 + ln.tid] | st) : st;
        }
    }
}

// ------------------------------------------------ fused linear-model UKF smoother --
//   fk_ukf_linear_rts_f64 <- UnscentedKalmanFilter.rts_smoother (filterpy/kalman/UKF.py:634-739) with fx(x, dt) = F x:
// the whole backward loop in one launch, one track per lane, the smoothed step k+1 carried in registers
// (fk_ukf.hpp, ukf_linear_rts_step).  Reads Xs[k], Ps[k]; writes xs[k], ps[k], Ks[k]: 8 (2n + 3n^2) bytes per
// track-step (1008 at n = 6).  Before: four kernel launches and nine host<->device copies per step.
struct UkfRtsArgs {
    const double *Xs, *Ps;
    double *xs, *ps, *Ks;
    int32_t *status;
    long N, T;
    int n;
    double scale;
};

// (at dim_x = 6 the gain's second sweep accumulates Pb and the full n x n Pxb side by side next to L, x and xb: ~115
// live doubles -- one wave per SIMD there; two spilled 90-220 registers)
template <int NX, int LAYOUT, bool SCALAR_F, int VER = 3>
__global__ void __launch_bounds__(BLOCK, (NX <= 2 ? 4 : NX <= 4 ? 2 : 1))
ukf_linear_rts_kernel(const UkfRtsArgs a, const double *__restrict__ pF, const double *__restrict__ pQ,
                      const double *__restrict__ pWm, const double *__restrict__ pWc)
{
    constexpr int KS = 2 * NX + 1;
    constexpr int PL = NX * (NX + 1) / 2;
    using SharedModel = LdsModel<NX, 1>;
    __shared__ double s_model[SharedModel::SIZE + 2 * KS];
    const long N = a.N;
    const long blk0 = (long)blockIdx.x * BLOCK;
    const Lane ln{blk0, threadIdx.x, N};
    const bool live = blk0 + ln.tid < N;
    const Lane lr{blk0, live ? ln.tid : 0u, N};
    const int n = a.n;
    const int ks = 2 * n + 1;
    lds_fill<NX, NX>(s_model + SharedModel::OFF_F, pF, n, n, 1.0, threadIdx.x);
    lds_fill<NX, NX>(s_model + SharedModel::OFF_Q, pQ, n, n, 1.0, threadIdx.x);
    for (unsigned q = ln.tid; q < (unsigned)(2 * KS); q += BLOCK) {           // weights re-indexed to the padded point set
        const int which = q / KS, i = q % KS;
        int src = -1;
        if (i == 0) src = 0;
        else if (i <= NX) { if (i <= n) src = i; }
        else { if (i - NX <= n) src = n + (i - NX); }
        const double *W = which ? pWc : pWm;
        s_model[SharedModel::SIZE + q] = (src >= 0 && src < ks) ? W[src] : 0.0;
    }
    __syncthreads();
    using StepModel = std::conditional_t<SCALAR_F, ScalarFHModel<NX, 1>, SharedModel>;
    struct View { StepModel sm; const double *Wm, *Wc; };
    int goff = 0;
    auto sweep = [&]() {
        if constexpr (SCALAR_F) {
            int t;
            asm volatile("s_mov_b32 %0, 0" : "=s"(t));
            goff = t;
        }
    };
    auto fresh = [&]() {
        int off = 0;
        asm volatile("" : "+v"(off));
        const double *mb = s_model + off;
        if constexpr (SCALAR_F) return View{StepModel{mb, pF + goff, pF + goff}, mb + SharedModel::SIZE, mb + SharedModel::SIZE + KS};
        else return View{StepModel{mb}, mb + SharedModel::SIZE, mb + SharedModel::SIZE + KS};
    };
    auto load_state = [&](long t, double (&x)[NX], double (&P)[PL]) {
        load_rec<NX, 1, LAYOUT, false>(x, a.Xs + t * N * n, lr, n, 1, 0.0);
        const RecView<LAYOUT> pv(a.Ps + t * N * n * n, lr, n * n);
        FK_UNROLL for (int i = 0; i < NX; ++i)
            FK_UNROLL for (int j = 0; j < NX; ++j)
                if (j >= i) P[sym_idx<NX>(i, j)] = (i < n && j < n) ? pv.load(i * n + j) : (i == j ? 1.0 : 0.0);
    };
    auto store_state = [&](long t, const double (&x)[NX], const double (&P)[PL]) {
        store_rec<NX, 1, LAYOUT, false>(x, a.xs + t * N * n, ln, n, 1);
        double Pf[NX * NX];
        FK_UNROLL for (int i = 0; i < NX; ++i)
            FK_UNROLL for (int j = 0; j < NX; ++j) Pf[i * NX + j] = P[sym_idx<NX>(i, j)];
        store_rec<NX, NX, LAYOUT, false>(Pf, a.ps + t * N * n * n, ln, n, n);
    };

    double xn[NX], Pn[PL];
    load_state(a.T - 1, xn, Pn);
    if (live) {
        // the last step is the filter's own output, copied as it is (both triangles: xs, ps = Xs.copy(), Ps.copy())
        store_rec<NX, 1, LAYOUT, false>(xn, a.xs + (a.T - 1) * N * n, ln, n, 1);
        {
            double Pf[NX * NX];
            load_rec<NX, NX, LAYOUT, false>(Pf, a.Ps + (a.T - 1) * N * n * n, lr, n, n, 0.0);
            store_rec<NX, NX, LAYOUT, false>(Pf, a.ps + (a.T - 1) * N * n * n, ln, n, n);
        }
        if (a.Ks) {
            double Z[NX * NX];
            FK_UNROLL for (int e = 0; e < NX * NX; ++e) Z[e] = 0.0;
            store_rec<NX, NX, LAYOUT, false>(Z, a.Ks + (a.T - 1) * N * n * n, ln, n, n);
        }
    }
    int st = 0;
    for (long t = a.T - 2; t >= 0; --t) {
        double x[NX], P[PL], K[NX * NX];
        load_state(t, x, P);
        {
            double xb[NX], Pb[PL];
            if constexpr (VER == 3) st |= ukf_linear_rts_gain_v3<NX>(x, P, a.scale, xb, Pb, K, fresh, sweep);
            else st |= ukf_linear_rts_gain<NX>(x, P, a.scale, xb, Pb, K, fresh, sweep);
            ukf_linear_rts_correct<NX>(x, P, xn, Pn, xb, Pb, K);
            FK_UNROLL for (int c = 0; c < NX; ++c) xn[c] = x[c];
            FK_UNROLL for (int e = 0; e < PL; ++e) Pn[e] = P[e];
        }
        if (live) {
            store_state(t, x, P);
            if (a.Ks) store_rec<NX, NX, LAYOUT, false>(K, a.Ks + t * N * n * n, ln, n, n);
        }
    }
    if (live && a.status) {
        if (!all_finite<NX>(xn) || !all_finite<PL>(Pn)) st |= ST_NONFINITE;
        a.status[ln.blk0 + ln.tid] = st;
    }
}

static int fail(int code, const char *msg)
{
    set_last_error(msg);
    return code;
}

}  // namespace fk

using namespace fk;

extern "C" {

int fk_ukf_linear_batch_f64(const fk_ukf_desc *d, const double *F, const double *H, const double *Q,
                            const double *R, const double *Wm, const double *Wc, const double *z,
                            const uint8_t *mask, double *x, double *P, double *means, double *covs,
                            int32_t *status, void *stream)
{
    if (!d) return fail(FK_ERR_BAD_ARG, "desc is NULL");
    if (d->n < 1 || d->n > 9 || d->m < 1 || d->m > 4 || (d->n <= 6 && d->m > 3))
        return fail(FK_ERR_UNSUPPORTED, "fused linear UKF: dim_x 1..6 with dim_z 1..3, dim_x 7..9 with dim_z 1..4");
    if (d->N < 0 || d->T < 0 || !F || !H || !Q || !R || !Wm || !Wc || !z || !x || !P)
        return fail(FK_ERR_BAD_ARG, "fused linear UKF: bad argument");
    if ((double)d->N * d->n * d->n * 8.0 >= 4294967296.0) return fail(FK_ERR_UNSUPPORTED, "fused linear UKF: record block >= 4 GiB, split the batch");
    if (d->N == 0 || d->T == 0) return FK_OK;
    UkfArgs a0{};
    a0.F = F; a0.H = H; a0.Q = Q; a0.R = R; a0.Wm = Wm; a0.Wc = Wc; a0.z = z; a0.mask = mask;
    a0.x = x; a0.P = P; a0.means = means; a0.covs = covs; a0.status = status;
    a0.N = d->N; a0.T = d->T; a0.n = d->n; a0.m = d->m; a0.scale = d->scale;
    a0.i0 = 0; a0.cnt = d->N; a0.status_or = 0;
    const int layout = d->layout;
    // Round 3: every class runs the factor-image step (fk_ukf.hpp, ukf_linear_step_v3) with the model in LDS -- it reads each
    // row of F and H once per step, so the scalar-operand form that V2's 4 x (2n+1) row sweeps needed no longer pays
    // (its SGPR spills cost more than the 83 broadcast reads).  A/B switches, exact (6,3) only unless noted:
    //   FK_UKF_SCALAR=1  F and H as scalar operands (also (8,4), (9,3));  FK_UKF_V2=1  round 2's point-by-point step;
    //   FK_UKF_V1=1  the straightforward kernel at dim_x <= 4.
    static const bool scalar_fh = getenv("FK_UKF_SCALAR") && getenv("FK_UKF_SCALAR")[0] == '1';
    static const bool step_v2 = getenv("FK_UKF_V2") && getenv("FK_UKF_V2")[0] == '1';
    static const bool kern_v1 = getenv("FK_UKF_V1") && getenv("FK_UKF_V1")[0] == '1';
    // one piece: tracks [a.i0, a.i0 + a.cnt), a.T steps from the pointers in a
    auto one = [&](const UkfArgs &a, hipStream_t s) -> int {
        const dim3 grid((unsigned)((a.cnt + BLOCK - 1) / BLOCK)), block(BLOCK);
#define GO(KERNEL, ...)                                                                                          \
    do {                                                                                                         \
        if (layout == FK_LAYOUT_SOA)                                                                             \
            hipLaunchKernelGGL((KERNEL<__VA_ARGS__, LAYOUT_SOA>), grid, block, 0, s, a, F, H, Q, R, Wm, Wc, a.z, a.mask); \
        else                                                                                                     \
            hipLaunchKernelGGL((KERNEL<__VA_ARGS__, LAYOUT_AOS>), grid, block, 0, s, a, F, H, Q, R, Wm, Wc, a.z, a.mask); \
    } while (0)
#define GO2(NXV, NZV, SC, VER)                                                                                   \
    do {                                                                                                         \
        if (layout == FK_LAYOUT_SOA)                                                                             \
            hipLaunchKernelGGL((ukf_linear_kernel_v2<NXV, NZV, LAYOUT_SOA, SC, VER>), grid, block, 0, s, a, F, H, Q, R, Wm, Wc, a.z, a.mask); \
        else                                                                                                     \
            hipLaunchKernelGGL((ukf_linear_kernel_v2<NXV, NZV, LAYOUT_AOS, SC, VER>), grid, block, 0, s, a, F, H, Q, R, Wm, Wc, a.z, a.mask); \
    } while (0)
        if (a.n <= 2 && a.m <= 2) {
            if (kern_v1) GO(ukf_linear_kernel, 2, 2);
            else GO2(2, 2, false, 3);
        } else if (a.n <= 4 && a.m <= 2) {
            if (kern_v1) GO(ukf_linear_kernel, 4, 2);
            else GO2(4, 2, false, 3);
        } else if (a.n <= 6 && a.m <= 3) {
            if (a.n == 6 && a.m == 3 && step_v2) GO2(6, 3, true, 2);
            else if (a.n == 6 && a.m == 3 && scalar_fh) GO2(6, 3, true, 3);
            else GO2(6, 3, false, 3);
        } else if (a.n <= 8) {                                             // round 3: dim_x 7..9 fused, one lane per track
            if (a.n == 8 && a.m == 4 && scalar_fh) GO2(8, 4, true, 3);
            else GO2(8, 4, false, 3);
        } else {
            if (a.m == 3 && scalar_fh) GO2(9, 3, true, 3);
            else if (a.m <= 3) GO2(9, 3, false, 3);
            else GO2(9, 4, false, 3);
        }
#undef GO
#undef GO2
        return check_launch("ukf_linear_kernel");
    };
    // tail filling (fk_chunks.hpp): FK_UKF_CHUNKS="G,H" cuts the call into G track groups x H time chunks on G streams, the
    // state handed from chunk to chunk through x / P in place (bit-identical results).  Default: one launch -- at
    // BASELINE configs[3] (1563 waves for 2048 wave slots) there is no last round to fill, and the forced decompositions
    // measured no faster (profiles/r03/ukf_chunking.jsonl).
    return ukf_chunked_call(a0, a0.n, a0.m, one, (hipStream_t)stream);
}

int fk_ukf_linear_rts_f64(const fk_ukf_desc *d, const double *F, const double *Q, const double *Wm, const double *Wc,
                          const double *Xs, const double *Ps, double *xs, double *Ps_out, double *K, int32_t *status,
                          void *stream)
{
    if (!d) return fail(FK_ERR_BAD_ARG, "desc is NULL");
    if (d->n < 1 || d->n > 9) return fail(FK_ERR_UNSUPPORTED, "fused linear UKF smoother: dim_x 1..9");
    if (d->N < 0 || d->T < 0 || !F || !Q || !Wm || !Wc || !Xs || !Ps || !xs || !Ps_out)
        return fail(FK_ERR_BAD_ARG, "fused linear UKF smoother: bad argument");
    if ((double)d->N * d->n * d->n * 8.0 >= 4294967296.0) return fail(FK_ERR_UNSUPPORTED, "fused linear UKF smoother: record block >= 4 GiB, split the batch");
    if (d->N == 0 || d->T == 0) return FK_OK;
    UkfRtsArgs a{};
    a.Xs = Xs; a.Ps = Ps; a.xs = xs; a.ps = Ps_out; a.Ks = K; a.status = status;
    a.N = d->N; a.T = d->T; a.n = d->n; a.scale = d->scale;
    const dim3 grid((unsigned)((a.N + BLOCK - 1) / BLOCK)), block(BLOCK);
    hipStream_t s = (hipStream_t)stream;
    const bool soa = d->layout == FK_LAYOUT_SOA;
#define GO(NXV, SC, VER)                                                                                         \
    do {                                                                                                         \
        if (soa) hipLaunchKernelGGL((ukf_linear_rts_kernel<NXV, LAYOUT_SOA, SC, VER>), grid, block, 0, s, a, F, Q, Wm, Wc); \
        else hipLaunchKernelGGL((ukf_linear_rts_kernel<NXV, LAYOUT_AOS, SC, VER>), grid, block, 0, s, a, F, Q, Wm, Wc);     \
    } while (0)
    // every class runs the factor-image gain (fk_ukf.hpp, ukf_linear_rts_gain_v3); FK_UKF_V2=1: round 2's point-by-point
    // gain with F as scalar operands, exact dim_x = 6 only (A/B)
    static const bool gain_v2 = getenv("FK_UKF_V2") && getenv("FK_UKF_V2")[0] == '1';
    if (d->n <= 2) GO(2, false, 3);
    else if (d->n <= 4) GO(4, false, 3);
    else if (d->n == 6 && gain_v2) GO(6, true, 2);
    else if (d->n <= 6) GO(6, false, 3);
    else if (d->n <= 8) GO(8, false, 3);                                   // round 3: dim_x 7..9
    else GO(9, false, 3);
#undef GO
    return check_launch("ukf_linear_rts_kernel");
}

}  // extern "C"
