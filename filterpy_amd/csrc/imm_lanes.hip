// imm_lanes.hip -- the batched Interacting Multiple Model estimator with ONE LANE PER FILTER of a bank (gfx950).
//
// filterpy/kalman/IMM.py: predict :188-222, update :160-186, _compute_state_estimate :224-237,
// _compute_mixing_probabilities :239-249.  imm_kernels.hip gives a lane a track's WHOLE bank; above (6,3) x 3 that bank --
// n_models x (dim_x + dim_x (dim_x + 1) / 2) doubles -- does not fit a lane's registers and the classes (9,4) / (16,8) ran out
// of scratch memory at 0.00-0.07 of HBM (round 5 / 6 tables).  Here a GROUP of G = 2, 4, 8 or 16 adjacent lanes owns a bank and
// lane j of the group owns filter j: its x and its packed P stay in that lane's registers for the whole time loop, its own
// F, Q, H, R come from LDS through a per-lane base address (the model blocks of a bank fall on different LDS banks), and
// predict / update are the one-filter routines of fk_math_sym.hpp, every lane on its own filter.  What the reference does
// ACROSS the filters goes through a wave-private LDS image ([element][lane]: every lane publishes its x, mu and P):
//   * mixing (IMM.py:200-219, :239-249): lane j adds up cbar_j, omega[.][j], its mixed x and -- element by element, the
//     filters i = 0 .. n_models-1 in the reference's order -- its mixed P;
//   * the bank's estimate (IMM.py:224-237): every lane forms x (nine sums), the dim_x (dim_x + 1) / 2 distinct elements of
//     P are dealt out over the G lanes of the group, each written to its two places of the record;
//   * the normalisation of mu (IMM.py:181-183): the sum over the group in filter order.
// One publication serves the posterior estimate of step t-1 AND the mixing of step t.  A wave works on 64 / G banks; the
// instruction stream is that of ONE filter plus the exchange, so a bank of eight (9,4) filters costs a wave about what a single
// (9,4) track costs the one-lane kernels -- for eight banks at once.
//
// n_models is a run-time value (2 .. G, idle lanes of a group duplicate the last filter and store nothing): four kernels per
// class and kind (plain; extended: MMAE, missing measurements, control input, the single-phase calls).  Exact arithmetic per
// element as in fk_imm.hpp (same operations, same order); parity against the oracle in tests/test_gpu_imm.py.  Not served here (imm_kernels.hip keeps them): the
// register-resident small banks and the (16, 8) class.
#include <type_traits>

#include "fk_device.hpp"
#include "fk_imm.hpp"
#include "fk_kernel_args.hpp"
#include "fk_ml.hpp"
#include "../../include/filterhip.h"

// FK_IL_STREAM_PREDICT / FK_IL_STREAM_UPDATE = 1 (build time): lanes_predict / lanes_update below -- the model's rows streamed from LDS,
// the results parked in the exchange image -- instead of kf_predict_sym / kf_update_sym with their register copy of the model.
// MEASURED SLOWER (profiles/r06/imm_lanes/streamed_vs_register_copy*.txt: (9,4) x 8 4.02 ms both streamed, 3.84 / 3.74 one of them,
// 3.55 neither): with one wave per SIMD the 1500 LDS operations of a streamed step are latency nobody hides, while the copy's
// v_accvgpr_read are issue slots the lone wave has to spare.  Kept as the A/B.
#ifndef FK_IL_STREAM_PREDICT
#define FK_IL_STREAM_PREDICT 0
#endif
#ifndef FK_IL_STREAM_UPDATE
#define FK_IL_STREAM_UPDATE 0
#endif

namespace fk {

// offset (in doubles) of element e of record `rec` in an [N][nelem] (NumPy order) or [nelem][N] (element-major) block: 32 bits
// (the entry point refuses record blocks of 4 GiB and more), so that an access is a uniform base plus one register
struct RecMap {
    unsigned rs, es;
    __device__ __forceinline__ unsigned at(unsigned rec, int e) const { return rec * rs + (unsigned)e * es; }
};
__device__ __forceinline__ RecMap rec_map(bool aos, long N, int nelem)
{
    return aos ? RecMap{(unsigned)nelem, 1u} : RecMap{1u, (unsigned)N};
}

// kf_predict_sym (fk_math_sym.hpp) with the working set of a lone wave in mind: the same operations in the same order, but F's rows
// STREAMED from LDS -- two rows ahead of the dot product that consumes them -- instead of held as a copy (81 doubles at dim_x 9,
// which the allocator kept in accumulator registers: 810 v_accvgpr_read per step, each an issue slot of the VALU, and scratch
// reloads inside the hottest loop on a bad day), and F P F' PARKED element by element in the lane's column of the exchange image
// (free between two exchanges; LDS traffic costs the VALU nothing).  park: element e at park[e * 64].
template <int NX, class Model>
__device__ __forceinline__ void lanes_predict(double (&x)[NX], double (&U)[NX * (NX + 1) / 2], const Model &M, double *park)
{
    double xn[NX];
    FK_UNROLL for (int i = 0; i < NX; ++i) {
        double f[NX], fp[NX], g[3][NX];
        M.rowF(i, f);
        if (i + 1 < NX) M.rowF(i + 1, g[0]);
        if (i + 2 < NX) M.rowF(i + 2, g[1]);
        xn[i] = dot<NX>(f, x);
        FK_UNROLL for (int j = 0; j < NX; ++j) {
            double acc = f[0] * U[sym_idx<NX>(0, j)];
            FK_UNROLL for (int k = 1; k < NX; ++k) acc = fma(f[k], U[sym_idx<NX>(k, j)], acc);
            fp[j] = acc;
        }
        park[sym_idx<NX>(i, i) * 64] = dot<NX>(fp, f);
        FK_STAGE();
        FK_UNROLL for (int j = i + 1; j < NX; ++j) {
            if (j + 2 < NX) M.rowF(j + 2, g[(j + 2 - (i + 1)) % 3]);
            park[sym_idx<NX>(i, j) * 64] = dot<NX>(fp, g[(j - (i + 1)) % 3]);
            FK_STAGE();
        }
    }
    FK_UNROLL for (int i = 0; i < NX; ++i) x[i] = xn[i];
    FK_UNROLL for (int i = 0; i < NX; ++i) {
        double q[NX];
        M.rowQ(i, q);
        FK_UNROLL for (int j = i; j < NX; ++j) U[sym_idx<NX>(i, j)] = fma(1.0, park[sym_idx<NX>(i, j) * 64], q[j]);
        FK_STAGE();
    }
}

// kf_update_sym<NX, NZ, true> (fk_math_sym.hpp) the same way: H's and R's rows read where they are used (no register copy of the
// model), the Joseph-form rows parked in the lane's column of the exchange image as they are produced instead of collected in a
// second packed matrix: what is live at the peak is U, P H', K and one row -- 150 doubles instead of 250.  Same operations, same
// order.  Returns status bits; y, Lf, dinv like kf_update_sym (the caller's likelihood).
template <int NX, int NZ, class Model>
__device__ __forceinline__ int lanes_update(double (&x)[NX], double (&U)[NX * (NX + 1) / 2], const double (&z)[NZ], const Model &M,
                                            double *park, double (&y)[NZ], double (&Lf)[NZ * NZ], double (&dinv)[NZ])
{
    int st = 0;
    double PHT[NX * NZ], K[NX * NZ], S[NZ * NZ];
    FK_UNROLL for (int r = 0; r < NZ; ++r) {
        double h[NX];
        M.rowH(r, h);
        y[r] = z[r] - dot<NX>(h, x);
        FK_UNROLL for (int i = 0; i < NX; ++i) {
            double acc = U[sym_idx<NX>(i, 0)] * h[0];
            FK_UNROLL for (int k = 1; k < NX; ++k) acc = fma(U[sym_idx<NX>(i, k)], h[k], acc);
            PHT[i * NZ + r] = acc;
        }
        FK_STAGE();
    }
    FK_UNROLL for (int r = 0; r < NZ; ++r) {
        double h[NX], rr[NZ];
        M.rowH(r, h);
        M.rowR(r, rr);
        FK_UNROLL for (int c = 0; c < NZ; ++c) {
            double acc = h[0] * PHT[c];
            FK_UNROLL for (int k = 1; k < NX; ++k) acc = fma(h[k], PHT[k * NZ + c], acc);
            S[r * NZ + c] = acc + rr[c];
        }
        FK_STAGE();
    }
    FK_UNROLL for (int i = 0; i < NX * NZ; ++i) K[i] = PHT[i];
    if constexpr (NZ == 1) {
        const double si = fk_rcp(S[0]);
        if (!(S[0] != 0.0)) st |= ST_NOT_PD;
        dinv[0] = si;
        Lf[0] = S[0];
        FK_UNROLL for (int i = 0; i < NX; ++i) K[i] = PHT[i] * si;
    } else {
        double d[NZ];
        FK_UNROLL for (int i = 0; i < NZ * NZ; ++i) Lf[i] = S[i];
        if (!ldlt2<NZ, true>(Lf, d, dinv)) st |= ST_NOT_PD;
        solve_rows_ldlt<NX, NZ>(Lf, dinv, K);
    }
    FK_STAGE();
    FK_UNROLL for (int i = 0; i < NX; ++i) {
        double acc = x[i];
        FK_UNROLL for (int k = 0; k < NZ; ++k) acc = fma(K[i * NZ + k], y[k], acc);
        x[i] = acc;
    }
    // Joseph form, one row of the result at a time
    FK_UNROLL for (int i = 0; i < NX; ++i) {
        double t1[NX];
        FK_UNROLL for (int k = 0; k < NX; ++k) {
            double acc = U[sym_idx<NX>(i, k)];
            FK_UNROLL for (int r = 0; r < NZ; ++r) acc = fma(-K[i * NZ + r], PHT[k * NZ + r], acc);
            t1[k] = acc;
        }
        double D[NZ];
        FK_UNROLL for (int c = 0; c < NZ; ++c) D[c] = 0.0;
        FK_UNROLL for (int r = 0; r < NZ; ++r) {
            double rr[NZ];
            M.rowR(r, rr);
            FK_UNROLL for (int c = 0; c < NZ; ++c) D[c] = (r == 0) ? K[i * NZ] * rr[c] : fma(K[i * NZ + r], rr[c], D[c]);
        }
        FK_UNROLL for (int r = 0; r < NZ; ++r) {
            double h[NX];
            M.rowH(r, h);
            D[r] -= dot<NX>(t1, h);
        }
        FK_UNROLL for (int j = i; j < NX; ++j) {
            double acc = t1[j];
            FK_UNROLL for (int r = 0; r < NZ; ++r) acc = fma(D[r], K[j * NZ + r], acc);
            park[sym_idx<NX>(i, j) * 64] = acc;
        }
        FK_STAGE();
    }
    FK_UNROLL for (int e = 0; e < NX * (NX + 1) / 2; ++e) U[e] = park[e * 64];
    return st;
}

// What a wave needs to know for an exchange (all of it the same from step to step)
struct LanesCtx {
    double *wX, *wP, *wH, *wMu;
    const double *sM;
    const unsigned short *s_rc;
    unsigned lane, g0, grp, j, jm, bank;
    int NM, n;
    bool live;
    RecMap ox, oP;
};

// One exchange: the estimate of the bank as it stands (est: x_dst / P_dst may each be NULL) and / or (MIX) the mixing of the next
// step.  (A function template, not a lambda of the kernel: as a generic lambda -- MIX has to be a compile-time choice, as a
// run-time flag every element of P carried a copy through the join of the two paths -- it kept the kernel's P in scratch memory.)
template <bool MIX, int NX, int G, int CH, int PH>
__device__ __forceinline__ void lanes_exchange(const LanesCtx &c, double (&x)[NX], double (&P)[NX * (NX + 1) / 2], const double mu,
                                               double &cbar, const bool est, double *x_dst, double *P_dst, const bool mmae = false)
{
    constexpr bool mix = MIX;
    constexpr int PL = NX * (NX + 1) / 2, GPW = 64 / G;
    double *const wX = c.wX, *const wP = c.wP, *const wH = c.wH, *const wMu = c.wMu;
    const double *const sM = c.sM;
    const unsigned short *const s_rc = c.s_rc;
    const unsigned lane = c.lane, g0 = c.g0, grp = c.grp, j = c.j, jm = c.jm, bank = c.bank;
    const int NM = c.NM, n = c.n;
    const bool live = c.live;
    const RecMap ox = c.ox, oP = c.oP;
        ml_wave_fence();
        FK_UNROLL for (int r = 0; r < NX; ++r) wX[r * 64 + lane] = x[r];
        wMu[lane] = mu;
        ml_wave_fence();
        double rc = 0.0;
        bool tiny = false;
        if constexpr (mix) {
            // cbar_j = sum_i mu_i M[i][j]  (IMM.py:244); one reciprocal per column like fk_imm.hpp
            double acc = 0.0;
            for (int i = 0; i < NM; ++i) acc = fma(wMu[g0 + i], sM[i * NM + (int)jm], acc);
            cbar = acc;
            tiny = cbar < 0x1p-500;
            rc = fk_rcp(tiny ? cbar * 0x1p600 : cbar);
        }
        {
            double xh[NX], xm[NX];
            FK_UNROLL for (int r = 0; r < NX; ++r) xh[r] = xm[r] = 0.0;
            for (int i = 0; i < NM; ++i) {
                const double mi = wMu[g0 + i];
                const double num = sM[i * NM + (int)jm] * mi;
                const double w = (tiny ? num * 0x1p600 : num) * rc;
                FK_UNROLL for (int r = 0; r < NX; ++r) {
                    const double xi = wX[r * 64 + g0 + i];
                    xh[r] = fma(xi, mi, xh[r]);
                    xm[r] = fma(xi, w, xm[r]);
                }
            }
            if (est) {
                // the estimate's x by every lane (IMM.py:224-237); lane j of the group stores elements j, j + G, ...
                FK_UNROLL for (int r = 0; r < NX; ++r) wH[r * GPW + grp] = xh[r];
                ml_wave_fence();
                if (x_dst) {
                    _Pragma("nounroll") for (int k = 0; k < (NX + G - 1) / G; ++k) {
                        const int r = (int)j + G * k;
                        if (r < n && live) x_dst[ox.at(bank, r)] = wH[r * GPW + grp];
                    }
                }
            }
            if constexpr (mix) { FK_UNROLL for (int r = 0; r < NX; ++r) x[r] = xm[r]; }       // (the old x stays published in wX)
        }
        if (!mix && !(est && P_dst)) return;
        FK_UNROLL for (int ph = 0; ph < PH; ++ph) {
            const int lo = ph * CH, hi = (lo + CH < PL) ? lo + CH : PL;
            ml_wave_fence();
            FK_UNROLL for (int e = 0; e < CH; ++e)
                if (lo + e < hi) wP[e * 64 + lane] = P[lo + e];
            ml_wave_fence();
            if (est && P_dst) {
                // the distinct elements of the estimate's P dealt out over the group, each to its two places of the record
                // (a loop, not unrolled: nothing in it is indexed by k but LDS, and unrolled it is 23 copies at G = 2)
                _Pragma("nounroll") for (int k = 0; k < (CH + G - 1) / G; ++k) {
                    const int q = (int)j + G * k;                      // element of this piece
                    const int qc = lo + q < hi ? q : hi - lo - 1;
                    const int rcw = s_rc[lo + qc], r = rcw & 255, c = rcw >> 8;
                    const double hr = wH[r * GPW + grp], hc = wH[c * GPW + grp];
                    double acc = 0.0;
                    if (mmae) {
                        // MMAEFilterBank's covariance (mmae.py:191-207) zips the COMPONENTS of x with the filters: filter k is centred on
                        // the scalar x[k], and only the first min(dim_x, n_models) filters contribute (fk_imm.hpp, mmae_estimate)
                        const int kmax = NM < n ? NM : n;
                        for (int i = 0; i < kmax; ++i) {
                            const double hk = wH[i * GPW + grp];
                            const double ya = wX[r * 64 + g0 + i] - hk, yb = wX[c * 64 + g0 + i] - hk;
                            acc = fma(wMu[g0 + i], fma(ya, yb, wP[qc * 64 + g0 + i]), acc);
                        }
                    } else {
                        for (int i = 0; i < NM; ++i) {
                            const double ya = wX[r * 64 + g0 + i] - hr, yb = wX[c * 64 + g0 + i] - hc;
                            acc = fma(wMu[g0 + i], fma(ya, yb, wP[qc * 64 + g0 + i]), acc);
                        }
                    }
                    if (lo + q < hi && c < n && live) {
                        P_dst[oP.at(bank, r * n + c)] = acc;
                        if (r != c) P_dst[oP.at(bank, c * n + r)] = acc;
                    }
                }
            }
            if constexpr (mix) {
                // mixed initial conditions (IMM.py:200-219): P0_j = sum_i omega[i][j] (outer(x_i - x0_j) + P_i), filter by filter
                // (x holds x0_j by now)
                FK_UNROLL for (int e = 0; e < CH; ++e)
                    if (lo + e < hi) P[lo + e] = 0.0;
                for (int i = 0; i < NM; ++i) {
                    const double num = sM[i * NM + (int)jm] * wMu[g0 + i];
                    const double w = (tiny ? num * 0x1p600 : num) * rc;
                    double d[NX];
                    FK_UNROLL for (int r = 0; r < NX; ++r) d[r] = wX[r * 64 + g0 + i] - x[r];
                    FK_UNROLL for (int r = 0; r < NX; ++r)
                        FK_UNROLL for (int c = r; c < NX; ++c) {
                            const int e = sym_idx<NX>(r, c);
                            if (e >= lo && e < hi) P[e] = fma(w, fma(d[r], d[c], wP[(e - lo) * 64 + g0 + i]), P[e]);
                        }
                }
            }
        }
}

// (waves per SIMD the register budget is held to: left to itself the compiler spreads a (4,2) filter over 280 registers)
// EXT: the instantiation that also carries MMAE (mmae.py:140-207: no mixing, p *= likelihood, its own estimate), missing measurements
// (update(None): the filters keep x, P; the likelihood is the density of a zero residual under the S of the filter's last real update,
// IMM.py:171-179 + kalman_filter.py:511-520) and the control input (x = F x + B u, kalman_filter.py:472-475) -- as run-time choices, kept
// out of the plain kernel's register allocation.
template <int NX, int NZ, int G, bool EXT>
__global__ void __launch_bounds__(BLOCK, (NX <= 4 ? 2 : 1))
imm_lanes_kernel(const ImmArgs a, const int NM, const int aos)
{
    using LM = LdsModel<NX, NZ>;
    constexpr int PL = NX * (NX + 1) / 2, WAVES = BLOCK / 64, GPW = 64 / G;         // GPW: banks (groups) per wave
    // The packed P is exchanged in PH pieces of at most CH elements (the image of a whole 16 x 16 bank would be 70 KB per wave):
    // as many as fit beside the G model blocks in 160 KB, element e of P in piece e / CH.
    constexpr int LDS_DOUBLES = 160 * 1024 / 8 - 64;                               // (- s_rc and alignment slack)
    constexpr int FIXED = NX * 64 + NX * GPW + 64 + 64;                            // X | xhat | mu | scratch
    constexpr int ROOM = (LDS_DOUBLES - G * LM::SIZE - G * G) / WAVES - FIXED;
    static_assert(ROOM >= 64 * 8, "no room for the exchange image");
    constexpr int CH = (ROOM / 64 >= PL) ? PL : ROOM / 64, PH = (PL + CH - 1) / CH;
    constexpr int WSZ = FIXED + CH * 64;
    __shared__ double smem[G * LM::SIZE + G * G + WAVES * WSZ];
    __shared__ unsigned short s_rc[PL];                                            // packed index -> row | col << 8
    const int n = a.n, m = a.m;
    const long N = a.N;
    for (int j = 0; j < NM; ++j) {
        double *s = smem + j * LM::SIZE;
        lds_fill<NX, NX>(s + LM::OFF_F, a.F + (long)j * n * n, n, n, 1.0, threadIdx.x);
        lds_fill<NX, NX>(s + LM::OFF_Q, a.Q + (long)j * n * n, n, n, 0.0, threadIdx.x);
        lds_fill<NZ, NX>(s + LM::OFF_H, a.H + (long)j * m * n, m, n, 0.0, threadIdx.x);
        lds_fill<NZ, NZ>(s + LM::OFF_R, a.R + (long)j * m * m, m, m, 1.0, threadIdx.x);
    }
    double *sM = smem + G * LM::SIZE;
    if ((int)threadIdx.x < NM * NM) sM[threadIdx.x] = a.Mt ? a.Mt[threadIdx.x] : 0.0;
    for (int p = threadIdx.x; p < PL; p += BLOCK) {
        int r = 0, base = 0;
        while (base + (NX - r) <= p) { base += NX - r; ++r; }
        s_rc[p] = (unsigned short)(r | ((r + (p - base)) << 8));
    }
    __syncthreads();

    const unsigned lane = threadIdx.x & 63u, wave = wave_index();
    const unsigned j = lane & (unsigned)(G - 1), g0 = lane & ~(unsigned)(G - 1), grp = lane / (unsigned)G;
    const bool active = (int)j < NM;
    const unsigned jm = active ? j : (unsigned)(NM - 1);
    const long end = a.i0 + a.cnt;
    const long w0 = a.i0 + ((long)blockIdx.x * WAVES + wave) * GPW;                // the wave's first bank
    if (w0 >= end) return;                                                         // (no block-wide barrier below)
    const bool live = w0 + grp < end;
    const unsigned bank = (unsigned)(live ? w0 + grp : end - 1);
    const bool writer = live && active;

    double *wX = smem + G * LM::SIZE + G * G + wave * WSZ, *wP = wX + NX * 64, *wH = wP + CH * 64, *wMu = wH + NX * GPW, *wS = wMu + 64;
    LM mod;
    mod.s = smem + jm * LM::SIZE;

    double x[NX], P[PL], mu;
    {
        const RecMap mx = rec_map(aos, N, NM * n), mP = rec_map(aos, N, NM * n * n), mm = rec_map(aos, N, NM);
        mu = a.mu[mm.at(bank, (int)jm)];
        // (padding: loads of a clamped element and a select, not a branch per element)
        FK_UNROLL for (int r = 0; r < NX; ++r) {
            const int rr = r < n ? r : n - 1;
            const double xv = a.xs[mx.at(bank, (int)jm * n + rr)];
            x[r] = (r < n) ? xv : 0.0;
            FK_UNROLL for (int c = r; c < NX; ++c) {
                const int cc = c < n ? c : n - 1;
                const double pv = a.Ps[mP.at(bank, ((int)jm * n + rr) * n + cc)];
                P[sym_idx<NX>(r, c)] = (r < n && c < n) ? pv : ((r == c) ? 1.0 : 0.0);
            }
        }
    }
    int st = 0;
    const RecMap ox = rec_map(aos, N, n), oP = rec_map(aos, N, n * n), oM = rec_map(aos, N, NM), oz = rec_map(aos, N, m);
    const int nn = n * n;
    double cbar = 0.0;

    const LanesCtx ctx{wX, wP, wH, wMu, sM, s_rc, lane, g0, grp, j, jm, bank, NM, n, live, ox, oP};

    // (2 pi)^(-m/2), m = 1..8
    const double cm = m == 1 ? 0.3989422804014327 : m == 2 ? 0.15915494309189535 : m == 3 ? 0.06349363593424097
                    : m == 4 ? 0.025330295910584444 : m == 5 ? 0.010105326013811644 : m == 6 ? 0.004031441804149937
                    : m == 7 ? 0.0016083125866532416 : m == 8 ? 0.000641623890917771 : 1.0;
    const bool want_post = a.x_out || a.P_out, want_prior = a.xp_out || a.Pp_out;
    const bool mmae = EXT && a.mmae;
    // missing measurements: the log-density of a zero residual under this filter's last S (before any update S = 0: the density
    // evaluates to 0 and is floored at DBL_MIN, kalman_filter.py:1221-1225)
    [[maybe_unused]] double ll0v = -__builtin_inf();
    if (EXT && a.ll0) ll0v = a.ll0[rec_map(aos, N, NM).at(bank, (int)jm)];
    [[maybe_unused]] const double log2pi_m = m * 1.8378770664093453;

    double zc[NZ];
    FK_UNROLL for (int r = 0; r < NZ; ++r) {
        const double zv = a.z[oz.at(bank, r < m ? r : m - 1)];
        zc[r] = (r < m) ? zv : 0.0;
    }
    for (long t = 0; t < a.T; ++t) {
        // posterior estimate of step t-1 and mixing for step t from one publication
        double z[NZ];
        FK_UNROLL for (int r = 0; r < NZ; ++r) z[r] = zc[r];
        {
            long tn = t + 1 < a.T ? t + 1 : t;
            asm volatile("" : "+s"(tn));
            const double *zt = a.z + tn * N * m;
            FK_UNROLL for (int r = 0; r < NZ; ++r) {
                const double zv = zt[oz.at(bank, r < m ? r : m - 1)];
                zc[r] = (r < m) ? zv : 0.0;
            }
        }
        // (the single-phase calls of the class API, EXT only: FK_IMM_UPDATE skips mixing and predict -- cbar = mu . M from the mode
        //  probabilities as they stand, IMM.py:244 --, FK_IMM_PREDICT leaves before the update; T = 1)
        const int phase = EXT ? a.phase : (int)FK_IMM_STEP;
        if (phase == FK_IMM_UPDATE) {
            if (mmae) {
                cbar = mu;
            } else {
                ml_wave_fence();
                wMu[lane] = mu;
                ml_wave_fence();
                double acc = 0.0;
                for (int i = 0; i < NM; ++i) acc = fma(wMu[g0 + i], sM[i * NM + (int)jm], acc);
                cbar = acc;
            }
        } else {
        if (mmae) {
                cbar = mu;                                          // p_i *= likelihood_i (mmae.py:186-187): no mixing
                if (t > 0 && want_post)
                    lanes_exchange<false, NX, G, CH, PH>(ctx, x, P, mu, cbar, true, a.x_out ? a.x_out + (t - 1) * N * n : nullptr,
                                                         a.P_out ? a.P_out + (t - 1) * N * nn : nullptr, true);
            } else {
                lanes_exchange<true, NX, G, CH, PH>(ctx, x, P, mu, cbar, t > 0 && want_post, a.x_out ? a.x_out + (t - 1) * N * n : nullptr,
                                                    a.P_out ? a.P_out + (t - 1) * N * nn : nullptr);
            }
            // (the model block's offset is made opaque once per step: F, Q, H, R are the same every step, and hoisted out of the time
            //  loop they would sit in 88 .. 400 registers across it)
            {
                unsigned moff = jm * (unsigned)LM::SIZE;
                asm volatile("" : "+v"(moff));
                mod.s = smem + moff;
            }
        if constexpr (PH == 1 && NX >= 7 && FK_IL_STREAM_PREDICT) {
            ml_wave_fence();
            lanes_predict<NX>(x, P, mod, wP + lane);
            ml_wave_fence();
        } else {
            kf_predict_sym<NX>(x, P, mod, 1.0);
        }
        if (EXT && a.nu > 0) {
            // every filter's predict(u): x = F x + B u, B u formed on its own like dot(B, u)
            const double *ut = a.u + t * N * a.nu;
            const RecMap ou = rec_map(aos, N, a.nu);
            double uu[4];
            FK_UNROLL for (int c = 0; c < 4; ++c) uu[c] = c < a.nu ? ut[ou.at(bank, c)] : 0.0;
            FK_UNROLL for (int r = 0; r < NX; ++r) {
                if (r < n) {
                    const double *Br = a.B + ((int)jm * n + r) * a.nu;
                    double bu = Br[0] * uu[0];
                    FK_UNROLL for (int c = 1; c < 4; ++c)
                        if (c < a.nu) bu = fma(Br[c], uu[c], bu);
                    x[r] += bu;
                }
            }
        }
        if (want_prior)
            lanes_exchange<false, NX, G, CH, PH>(ctx, x, P, mu, cbar, true, a.xp_out ? a.xp_out + t * N * n : nullptr,
                                                 a.Pp_out ? a.Pp_out + t * N * nn : nullptr);
        }      // phase != FK_IMM_UPDATE
        if (phase == FK_IMM_PREDICT) break;
        // this lane's filter: update, likelihood floored at DBL_MIN (kalman_filter.py:1213-1226; fk_imm.hpp, imm_update)
        double lj;
        bool has_z = true;
        if (EXT && a.mask) has_z = a.mask[t * N + bank] != 0;
        if (!has_z) {
            lj = exp(ll0v);
            if (lj == 0.0) lj = 2.2250738585072014e-308;
        } else {
            double y[NZ], Lf[NZ * NZ], dinv[NZ];
            if constexpr (PH == 1 && NX >= 7 && FK_IL_STREAM_UPDATE) {
                ml_wave_fence();
                st |= lanes_update<NX, NZ>(x, P, z, mod, wP + lane, y, Lf, dinv);
                ml_wave_fence();
            } else {
                double K[NX * NZ], S[NZ * NZ];
                st |= kf_update_sym<NX, NZ, true>(x, P, z, mod, K, y, S, Lf, dinv);
            }
            double q = 0.0;
            if constexpr (NZ == 1) {
                q = y[0] * y[0] * dinv[0];
            } else {
                double w[NZ];
                FK_UNROLL for (int i = 0; i < NZ; ++i) {
                    double acc = y[i];
                    FK_UNROLL for (int k2 = 0; k2 < NZ; ++k2)
                        if (k2 < i) acc = fma(-Lf[i * NZ + k2], w[k2], acc);
                    w[i] = acc;
                    if (i < m) q = fma(acc * acc, dinv[i], q);
                }
            }
            int e2;
            const double g = rsqrt_det_parts<NZ>(dinv, m, e2);
            lj = (cm * g) * exp(fma((double)e2, 0.6931471805599453, -0.5 * q));
            if (lj == 0.0) lj = 2.2250738585072014e-308;
            if constexpr (EXT) ll0v = -0.5 * (log2pi_m + logdet_from_dinv<NZ>(dinv, m));
        }
        // mu_j = cbar_j L_j / sum (IMM.py:181-183): the sum over the group in filter order
        {
            const double mj = cbar * lj;
            ml_wave_fence();
            wS[lane] = mj;
            ml_wave_fence();
            double sum = 0.0;
            for (int i = 0; i < NM; ++i) sum += wS[g0 + i];
            const bool tny = sum < 0x1p-500;
            const double rsum = fk_rcp(tny ? sum * 0x1p600 : sum);
            mu = (tny ? mj * 0x1p600 : mj) * rsum;
        }
        if (writer) {
            if (a.mu_out) (a.mu_out + t * N * NM)[oM.at(bank, (int)j)] = mu;
            if (a.L_out) (a.L_out + t * N * NM)[oM.at(bank, (int)j)] = lj;
        }
    }
    if (want_post && a.T > 0 && !(EXT && a.phase == FK_IMM_PREDICT))      // the last step's posterior estimate
        lanes_exchange<false, NX, G, CH, PH>(ctx, x, P, mu, cbar, true, a.x_out ? a.x_out + (a.T - 1) * N * n : nullptr,
                                             a.P_out ? a.P_out + (a.T - 1) * N * nn : nullptr, mmae);
    {
        const RecMap mx = rec_map(aos, N, NM * n), mP = rec_map(aos, N, NM * n * n), mm = rec_map(aos, N, NM);
        bool fin = all_finite<NX>(x) && all_finite<PL>(P) && (fabs(mu) <= 1.79769313486231570815e+308);
        if (writer) {
            a.mu[mm.at(bank, (int)j)] = mu;
            if (EXT && a.ll0) a.ll0[mm.at(bank, (int)j)] = ll0v;
            FK_UNROLL for (int r = 0; r < NX; ++r) {
                if (r < n) a.xs[mx.at(bank, (int)j * n + r)] = x[r];
                FK_UNROLL for (int c = 0; c < NX; ++c)
                    if (r < n && c < n) a.Ps[mP.at(bank, ((int)j * n + r) * n + c)] = P[sym_idx<NX>(r, c)];
            }
        }
        if (a.status) {
            // the bank's status: the OR over its filters
            int *wI = reinterpret_cast<int *>(wS);
            ml_wave_fence();
            wI[lane] = st | (fin ? 0 : ST_NONFINITE);
            ml_wave_fence();
            int sv = 0;
            for (int i = 0; i < NM; ++i) sv |= wI[g0 + i];
            if (live && j == 0) a.status[bank] = a.status_or ? (a.status[bank] | sv) : sv;
        }
    }
}

}  // namespace fk

using namespace fk;

#if !defined(FK_NX) || !defined(FK_NZ) || !defined(FK_IL_G) || !defined(FK_IL_EXT)
#error "compile with -DFK_NX= -DFK_NZ= (the class: every dim_x <= FK_NX, dim_z <= FK_NZ) -DFK_IL_G=2|4|8|16 (lanes per bank) -DFK_IL_EXT=0|1"
#endif
#define FK_IL_CAT_(a, b, c, d, e) a##b##_##c##_g##d##_x##e
#define FK_IL_CAT(a, b, c, d, e) FK_IL_CAT_(a, b, c, d, e)

// launch_imm_lanes_<NX>_<NZ>_g<G>_x<EXT>: banks of G/2 + 1 .. G filters of the class (one object per G and kind); x1 also serves MMAE,
// missing measurements and the control input; returns 1 when the call is not one this file serves
int FK_IL_CAT(launch_imm_lanes_, FK_NX, FK_NZ, FK_IL_G, FK_IL_EXT)(const ImmArgs &a, int n_models, int layout, hipStream_t s)
{
    if (a.n > FK_NX || a.m > FK_NZ || n_models < 2 || n_models > FK_IL_G) return 1;
    if (!FK_IL_EXT && (a.mmae || a.mask || a.ll0 || a.nu > 0 || a.phase != FK_IMM_STEP)) return 1;
    const int aos = layout == FK_LAYOUT_AOS ? 1 : 0;
    const long per_block = (BLOCK / 64) * (64 / FK_IL_G);
    const dim3 grid((unsigned)((a.cnt + per_block - 1) / per_block)), block(BLOCK);
    hipLaunchKernelGGL((imm_lanes_kernel<FK_NX, FK_NZ, FK_IL_G, (FK_IL_EXT != 0)>), grid, block, 0, s, a, n_models, aos);
    return 0;
}
