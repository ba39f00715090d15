// imm_quad.hip -- the batched Interacting Multiple Model estimator for dim_x = 10..16 / dim_z = 5..8: FOUR LANES PER FILTER (gfx950).
//
// filterpy/kalman/IMM.py: predict :188-222, update :160-186, _compute_state_estimate :224-237,
// _compute_mixing_probabilities :239-249; each filter's own step: kalman_filter.py:472-478 (predict), :533-556 (Joseph-form update).
// imm_lanes.hip gives every FILTER of a bank a lane (x and the packed P in registers); a 16 x 16 packed P is 272 registers, more
// than a lane addresses directly, and that kernel's (16,8) build lived in 9 KB of scratch memory per lane with 320-660 KB of code
// (40.8 ms for 5e6 bank-steps of (16,8) x 2; the one-lane-per-BANK rolled kernel before it: 247 ms).  Here a QUAD of lanes owns a
// filter the way kf_mlg.hip's quad owns a track: lane `sub` holds rows sub R .. sub R + R - 1 of the FULL P (R = NX / LPF: 64 doubles at
// dim_x 16), x is replicated in the quad, rows travel by quad-permute DPP moves, the filter's F, Q, H, R come from LDS through a
// per-lane base address.  A group of 4 G adjacent lanes (G = 2, 4, 8, 16 filters) owns a bank; a wave works on 16 / G banks.
//
//   predict  F P by rows (row l of P broadcast by its owner), then (F P) F' + Q on the lane's own rows: the reference's association
//   update   P H' rows local; S's rows dealt out over the quad (rows of P H' broadcast), gathered, L D L' and y replicated;
//            K rows local; T1 = P - K (P H')' in place (P symmetric: (H P)' = P H', the one-lane kernels' packed arithmetic makes the
//            same use of it), D = K R - T1 H', P+ = T1 + D K' with K's rows broadcast -- the Joseph form of fk_math_sym.hpp row by row
//   across the filters (mixing, the bank's estimate, the normalisation of mu): a wave-private LDS image [element][lane] like
//   imm_lanes.hip, the lane reading the columns of the lanes with ITS sub index (the same rows of the other filters); the
//   dim_x^2 elements of the estimate's P are dealt out over the 4 G lanes of the bank.
//
// The class is padded like every other (F, R with a unit diagonal, Q, H with zeros, P0 with a unit diagonal): run-time dim_x <= NX,
// dim_z <= NZ, n_models 2 .. G.  EXT: MMAE banks (mmae.py:140-207), missing measurements, the control input, the single-phase calls.
// Parity against the oracle: tests/test_gpu_imm.py (the banks above (9,4)).
#include <type_traits>

#include "fk_device.hpp"
#include "fk_imm.hpp"
#include "fk_kernel_args.hpp"
#include "fk_ml.hpp"
#include "../../include/filterhip.h"

#ifndef FK_IQ_LPF
#define FK_IQ_LPF 4          // lanes per filter: 4 (a quad) or 8 (a pair of quads: the second build of this file, imm_oct_*.o)
#endif
#define FK_IQ_NS_(l) iq##l
#define FK_IQ_NS(l) FK_IQ_NS_(l)

namespace fk {
namespace FK_IQ_NS(FK_IQ_LPF) {

#ifndef FK_IQ_OCC2
#define FK_IQ_OCC2 0
#endif
constexpr int LPF = FK_IQ_LPF;
static_assert(LPF == 4 || LPF == 8, "four or eight lanes per filter");

// offset (in doubles) of element e of record `rec` in an [N][nelem] (NumPy order) or [nelem][N] (element-major) block (32 bits:
// the entry point refuses record blocks of 4 GiB and more)
struct RecMap {
    unsigned rs, es;
    __device__ __forceinline__ unsigned at(unsigned rec, int e) const { return rec * rs + (unsigned)e * es; }
};
__device__ __forceinline__ RecMap rec_map(bool aos, long N, int nelem)
{
    return aos ? RecMap{(unsigned)nelem, 1u} : RecMap{1u, (unsigned)N};
}

// A filter's model block in LDS with PADDED pitches.  The lanes of a wave read it through per-lane addresses -- the quads of a wave
// belong to different filters, and the lane's own rows of F, Q, H, R start at sub * R rows --: with LdsModel's dense layout (row pitch
// NX, 704 doubles per (16,8) block) the blocks of two filters and the row groups of the four sub indices all start on the SAME LDS
// bank, an eight-way conflict on every per-lane read.  Row pitch NX + 2 (R's: NZ + 2) spreads the sub indices over the banks (16,8):
// 0 / 16 / 32 / 48, (12,4): 0 / 20 / 40 / 60), a block size = 2 (mod 32) doubles shifts every filter by 4 banks.
template <int NX, int NZ>
struct QuadModel {
    static constexpr int PX = NX + 2, PZ = NZ + 2;
    static constexpr int OFF_F = 0, OFF_Q = NX * PX, OFF_H = 2 * NX * PX, OFF_R = 2 * NX * PX + NZ * PX;
    static constexpr int SIZE0 = OFF_R + NZ * PZ, SIZE = SIZE0 + ((2 - SIZE0 % 32) + 32) % 32;
    const double *s;
    template <int LEN>
    __device__ __forceinline__ void row(int off, double (&r)[LEN]) const
    {
        FK_UNROLL for (int j = 0; j < LEN; ++j) r[j] = s[off + j];
    }
    __device__ __forceinline__ void rowF(int i, double (&r)[NX]) const { row<NX>(OFF_F + i * PX, r); }
    __device__ __forceinline__ void rowQ(int i, double (&r)[NX]) const { row<NX>(OFF_Q + i * PX, r); }
    __device__ __forceinline__ void rowH(int i, double (&r)[NX]) const { row<NX>(OFF_H + i * PX, r); }
    __device__ __forceinline__ void rowR(int i, double (&r)[NZ]) const { row<NZ>(OFF_R + i * PZ, r); }
};

// cooperative fill of one padded ROWS x COLS matrix (row pitch PITCH) from an r x c matrix in global memory; caller synchronises
template <int ROWS, int COLS, int PITCH>
__device__ __forceinline__ void quad_fill(double *dst, const double *__restrict__ src, int r, int c, double diag_pad, unsigned tid)
{
    for (unsigned k = tid; k < (unsigned)(ROWS * COLS); k += BLOCK) {
        const int a = (int)k / COLS, b = (int)k % COLS;
        dst[a * PITCH + b] = (a < r && b < c) ? src[a * c + b] : ((a == b) ? diag_pad : 0.0);
    }
}

// The value lane OWNER of the filter's group holds, in every lane of the group.  Four lanes: one quad-permute DPP move per half.  Eight
// lanes (a pair of quads, inside a DPP row of 16): every quad broadcasts ITS lane OWNER % 4, then the quad that does not hold the
// owner takes the other quad's value -- row_shr:4 into the odd quads (bank mask 0xA) or row_shl:4 into the even ones (0x5).
template <int OWNER>
__device__ __forceinline__ double grp_bcast(double v)
{
    static_assert(OWNER >= 0 && OWNER < LPF, "a lane of the group");
    if constexpr (LPF == 4) {
        return quad_bcast<OWNER>(v);
    } else {
        constexpr int q = OWNER & 3, h = OWNER >> 2;
        int lo = __double2loint(v), hi = __double2hiint(v);
        lo = __builtin_amdgcn_mov_dpp(lo, q * 0x55, 0xf, 0xf, true);
        hi = __builtin_amdgcn_mov_dpp(hi, q * 0x55, 0xf, 0xf, true);
        constexpr int ctrl = h == 0 ? 0x114 : 0x104, banks = h == 0 ? 0xA : 0x5;
        lo = __builtin_amdgcn_update_dpp(lo, lo, ctrl, 0xf, banks, false);
        hi = __builtin_amdgcn_update_dpp(hi, hi, ctrl, 0xf, banks, false);
        return __hiloint2double(hi, lo);
    }
}
// (k: a compile-time constant after unrolling; owner = lane k / R of the group)
#define FK_Q_OWN_(o, v) grp_bcast<((o) < LPF ? (o) : 0)>(v)
#define FK_Q_OWNER(k, R, v)                                                                                            \
    (((k) / (R)) == 0 ? FK_Q_OWN_(0, v) : ((k) / (R)) == 1 ? FK_Q_OWN_(1, v) : ((k) / (R)) == 2 ? FK_Q_OWN_(2, v) : ((k) / (R)) == 3 ? FK_Q_OWN_(3, v) \
     : ((k) / (R)) == 4 ? FK_Q_OWN_(4, v) : ((k) / (R)) == 5 ? FK_Q_OWN_(5, v) : ((k) / (R)) == 6 ? FK_Q_OWN_(6, v) : FK_Q_OWN_(7, v))

// element sub R + r of a replicated vector
template <int NX>
__device__ __forceinline__ double quad_pick(const double (&x)[NX], unsigned sub, int r)
{
    constexpr int R = NX / LPF;
    // (the candidates as VALUES before the selection: as a selection between loads the optimiser made it one load through a
    //  selected address, and the replicated vector an array in scratch memory)
    double c[LPF];
    FK_UNROLL for (int s = 0; s < LPF; ++s) {
        c[s] = x[s * R + r];
        asm volatile("" : "+v"(c[s]));
    }
    double v = c[0];
    FK_UNROLL for (int s = 1; s < LPF; ++s) v = sub == (unsigned)s ? c[s] : v;
    return v;
}

// The model block again through an offset the optimiser cannot see through: each phase of a step READS its rows of F, Q, H, R from
// LDS where it uses them -- left to itself the compiler merges the loads of the phases and keeps H (128 doubles at (16,8)) and the
// lane's rows of F (64) in registers across the step, i.e. the step's working set in scratch memory.
template <class LM>
__device__ __forceinline__ LM quad_fresh(const LM &M)
{
    unsigned zero = 0;
    asm volatile("" : "+v"(zero));
    LM r;
    r.s = M.s + zero;
    return r;
}

// x = F x ; P = (F P) F' + Q  (kalman_filter.py:472-478), the lane's rows of P
template <int NX, class LM>
__device__ __forceinline__ void quad_predict(double (&x)[NX], double (&P)[NX / LPF][NX], const LM &M0, unsigned sub)
{
    constexpr int R = NX / LPF;
    {
        const LM M = quad_fresh(M0);
        const double *fo = M.s + LM::OFF_F + sub * (unsigned)(R * LM::PX);      // the lane's own rows of F
        double xo[R];
        FK_UNROLL for (int r = 0; r < R; ++r) {
            double acc = fo[r * LM::PX] * x[0];
            FK_UNROLL for (int k = 1; k < NX; ++k) acc = fma(fo[r * LM::PX + k], x[k], acc);
            xo[r] = acc;
        }
        FK_UNROLL for (int k = 0; k < NX; ++k) x[k] = FK_Q_OWNER(k, R, xo[k % R]);
    }
    double FP[R][NX];
    {
    const LM M = quad_fresh(M0);
    const double *fo = M.s + LM::OFF_F + sub * (unsigned)(R * LM::PX);
    // (every LDS read of the step's loops is requested one iteration AHEAD of its use -- two buffers --: a lone wave per SIMD has nothing
    //  else to cover a ds_read -> s_waitcnt pair with; kf_mlg_predict.inc)
    double fc[2][R];
    FK_UNROLL for (int r = 0; r < R; ++r) fc[0][r] = fo[r * LM::PX];
    FK_UNROLL for (int l = 0; l < NX; ++l) {
        if (l + 1 < NX) { FK_UNROLL for (int r = 0; r < R; ++r) fc[(l + 1) & 1][r] = fo[r * LM::PX + l + 1]; }
        double prow[NX];
        FK_UNROLL for (int c = 0; c < NX; ++c) prow[c] = FK_Q_OWNER(l, R, P[l % R][c]);
        FK_UNROLL for (int r = 0; r < R; ++r) {
            const double f = fc[l & 1][r];
            FK_UNROLL for (int c = 0; c < NX; ++c) FP[r][c] = (l == 0) ? f * prow[c] : fma(f, prow[c], FP[r][c]);
        }
        FK_STAGE();
    }
    }
    FK_STAGE();
    const LM M = quad_fresh(M0);
    const double *qo = M.s + LM::OFF_Q + sub * (unsigned)(R * LM::PX);
    double fr[2][NX], qc[2][R];
    M.rowF(0, fr[0]);
    FK_UNROLL for (int r = 0; r < R; ++r) qc[0][r] = qo[r * LM::PX];
    FK_UNROLL for (int j = 0; j < NX; ++j) {
        if (j + 1 < NX) {
            M.rowF(j + 1, fr[(j + 1) & 1]);
            FK_UNROLL for (int r = 0; r < R; ++r) qc[(j + 1) & 1][r] = qo[r * LM::PX + j + 1];
        }
        FK_UNROLL for (int r = 0; r < R; ++r) P[r][j] = dot<NX>(FP[r], fr[j & 1]) + qc[j & 1][r];
        FK_STAGE();
    }
    FK_UNROLL for (int r = 0; r < R; ++r)
        FK_UNROLL for (int c = 0; c < NX; ++c) asm volatile("" : "+v"(P[r][c]));
    FK_STAGE();
}

// The update of the lane's filter; y, Lf (L D L' of S: strict lower part), dinv replicated in the quad for the caller's likelihood.
template <int NX, int NZ, class LM>
__device__ __forceinline__ int quad_update(double (&x)[NX], double (&P)[NX / LPF][NX], const double (&z)[NZ], const LM &M0, unsigned sub,
                                           double (&y)[NZ], double (&Lf)[NZ * NZ], double (&dinv)[NZ])
{
    constexpr int R = NX / LPF, ZR = NZ / LPF;
    int st = 0;
    LM M = quad_fresh(M0);
    FK_UNROLL for (int r = 0; r < NZ; ++r) {
        double h[NX];
        M.rowH(r, h);
        y[r] = z[r] - dot<NX>(h, x);
    }
    double PHT[R][NZ];
    M = quad_fresh(M0);
    {
        double hr[2][NX];
        M.rowH(0, hr[0]);
        FK_UNROLL for (int c = 0; c < NZ; ++c) {
            if (c + 1 < NZ) M.rowH(c + 1, hr[(c + 1) & 1]);
            const double (&h)[NX] = hr[c & 1];
            FK_UNROLL for (int r = 0; r < R; ++r) {
                double acc = P[r][0] * h[0];
                FK_UNROLL for (int k = 1; k < NX; ++k) acc = fma(P[r][k], h[k], acc);
                PHT[r][c] = acc;
            }
            FK_STAGE();
        }
    }
    FK_STAGE();
    {
        // S = H (P H') + R: lane `sub` forms rows sub ZR .. sub ZR + ZR - 1
        double So[ZR][NZ];
        M = quad_fresh(M0);
        const double *ho = M.s + LM::OFF_H + sub * (unsigned)(ZR * LM::PX);
        const double *ro = M.s + LM::OFF_R + sub * (unsigned)(ZR * LM::PZ);
        double hs[2][ZR][4], rq[ZR][NZ];
        FK_UNROLL for (int q = 0; q < ZR; ++q)
            FK_UNROLL for (int kk = 0; kk < 4; ++kk) hs[0][q][kk] = ho[q * LM::PX + kk];
        FK_UNROLL for (int k = 0; k < NX; ++k) {
            if (k % 4 == 0 && k + 4 < NX) {
                FK_UNROLL for (int q = 0; q < ZR; ++q)
                    FK_UNROLL for (int kk = 0; kk < 4; ++kk) hs[(k / 4 + 1) & 1][q][kk] = ho[q * LM::PX + k + 4 + kk];
            }
            if (k == NX - 4) {
                FK_UNROLL for (int q = 0; q < ZR; ++q)
                    FK_UNROLL for (int c = 0; c < NZ; ++c) rq[q][c] = ro[q * LM::PZ + c];
            }
            double prow[NZ];
            FK_UNROLL for (int c = 0; c < NZ; ++c) prow[c] = FK_Q_OWNER(k, R, PHT[k % R][c]);
            FK_UNROLL for (int q = 0; q < ZR; ++q) {
                const double h = hs[(k / 4) & 1][q][k % 4];
                FK_UNROLL for (int c = 0; c < NZ; ++c) So[q][c] = (k == 0) ? h * prow[c] : fma(h, prow[c], So[q][c]);
            }
            if (k % 4 == 3) FK_STAGE();
        }
        FK_UNROLL for (int q = 0; q < ZR; ++q)
            FK_UNROLL for (int c = 0; c < NZ; ++c) So[q][c] += rq[q][c];
        FK_UNROLL for (int a = 0; a < NZ; ++a)
            FK_UNROLL for (int c = 0; c < NZ; ++c) Lf[a * NZ + c] = (c <= a) ? FK_Q_OWNER(a, ZR, So[a % ZR][c]) : 0.0;
    }
    FK_STAGE();
    double K[R * NZ];
    {
        double d[NZ];
        if (!ldlt2<NZ, true>(Lf, d, dinv)) st |= ST_NOT_PD;
        FK_UNROLL for (int r = 0; r < R; ++r)
            FK_UNROLL for (int c = 0; c < NZ; ++c) K[r * NZ + c] = PHT[r][c];
        solve_rows_ldlt<R, NZ>(Lf, dinv, K);
    }
    FK_STAGE();
    {
        double xo[R];
        FK_UNROLL for (int r = 0; r < R; ++r) {
            double acc = quad_pick<NX>(x, sub, r);
            FK_UNROLL for (int k = 0; k < NZ; ++k) acc = fma(K[r * NZ + k], y[k], acc);
            xo[r] = acc;
        }
        FK_UNROLL for (int k = 0; k < NX; ++k) x[k] = FK_Q_OWNER(k, R, xo[k % R]);
    }
    // Joseph form.  T1 = (I - K H) P = P - K (P H')' in place
    // (the rows of P H' are broadcast a second time: as the same values as in S's phase the compiler kept the 128 broadcast doubles
    //  of that phase alive -- in scratch memory -- instead)
    FK_UNROLL for (int r = 0; r < R; ++r)
        FK_UNROLL for (int c = 0; c < NZ; ++c) asm volatile("" : "+v"(PHT[r][c]));
    FK_UNROLL for (int j = 0; j < NX; ++j) {
        double prow[NZ];
        FK_UNROLL for (int c = 0; c < NZ; ++c) prow[c] = FK_Q_OWNER(j, R, PHT[j % R][c]);
        FK_UNROLL for (int r = 0; r < R; ++r) {
            double acc = P[r][j];
            FK_UNROLL for (int c = 0; c < NZ; ++c) acc = fma(-K[r * NZ + c], prow[c], acc);
            P[r][j] = acc;
        }
        if (j % 2 == 1) FK_STAGE();
    }
    FK_STAGE();
    // D = K R - T1 H'
    double D[R][NZ];
    M = quad_fresh(M0);
    FK_UNROLL for (int a = 0; a < NZ; ++a) {
        double rr[NZ];
        M.rowR(a, rr);
        FK_UNROLL for (int r = 0; r < R; ++r)
            FK_UNROLL for (int c = 0; c < NZ; ++c) D[r][c] = (a == 0) ? K[r * NZ] * rr[c] : fma(K[r * NZ + a], rr[c], D[r][c]);
    }
    {
        double hr[2][NX];
        M.rowH(0, hr[0]);
        FK_UNROLL for (int c = 0; c < NZ; ++c) {
            if (c + 1 < NZ) M.rowH(c + 1, hr[(c + 1) & 1]);
            FK_UNROLL for (int r = 0; r < R; ++r) D[r][c] -= dot<NX>(P[r], hr[c & 1]);
            FK_STAGE();
        }
    }
    FK_STAGE();
    // P+ = T1 + D K'
    FK_UNROLL for (int j = 0; j < NX; ++j) {
        double krow[NZ];
        FK_UNROLL for (int c = 0; c < NZ; ++c) krow[c] = FK_Q_OWNER(j, R, K[(j % R) * NZ + c]);
        FK_UNROLL for (int r = 0; r < R; ++r) {
            double acc = P[r][j];
            FK_UNROLL for (int c = 0; c < NZ; ++c) acc = fma(D[r][c], krow[c], acc);
            P[r][j] = acc;
        }
        if (j % 2 == 1) FK_STAGE();
    }
    // (P+ is formed HERE: its broadcasts cannot leave this block, and the multiply-adds they feed sank past the likelihood's branches
    //  into a block of their own, with every broadcast value spilled on the way)
    FK_UNROLL for (int r = 0; r < R; ++r)
        FK_UNROLL for (int c = 0; c < NX; ++c) asm volatile("" : "+v"(P[r][c]));
    FK_STAGE();
    return st;
}

// What a wave needs to know for an exchange (the same from step to step)
struct QuadCtx {
    double *wX, *wP, *wH, *wMu;
    const double *sM;
    unsigned lane, g0, grp, bl, sub, jm, bank;
    int NM, n;
    bool live;
    RecMap ox, oP;
};

// One exchange: the estimate of the bank as it stands (est: x_dst / P_dst may each be NULL) and / or (MIX) the mixing of the next step
// (IMM.py:200-219, :224-237, :239-249; MMAE's estimate: mmae.py:191-207).  A function template taking x and P by reference
// (docs/KERNEL_NOTES.md: as a generic lambda the kernel's P stayed in scratch memory).
template <bool MIX, int NX, int G, int CH, int PH>
__device__ __forceinline__ void quad_exchange(const QuadCtx &c, double (&x)[NX], double (&P)[NX / LPF][NX], const double mu, double &cbar,
                                              const bool est, double *x_dst, double *P_dst, const bool mmae = false)
{
    constexpr int R = NX / LPF, PE = R * NX, LPB = LPF * G, GPW = 64 / LPB;
    double *const wX = c.wX, *const wP = c.wP, *const wH = c.wH, *const wMu = c.wMu;
    const double *const sM = c.sM;
    const unsigned lane = c.lane, grp = c.grp, bl = c.bl, sub = c.sub, jm = c.jm, bank = c.bank;
    const unsigned s0 = c.g0 + sub;                        // filter i of the bank, this lane's rows: lane s0 + 4 i
    const int NM = c.NM, n = c.n;
    const bool live = c.live;
    const RecMap ox = c.ox, oP = c.oP;
    ml_wave_fence();
    FK_UNROLL for (int r = 0; r < NX; ++r) wX[r * 64 + lane] = x[r];
    wMu[lane] = mu;
    ml_wave_fence();
    double rc = 0.0;
    bool tiny = false;
    if constexpr (MIX) {
        double acc = 0.0;                                  // cbar_j = sum_i mu_i M[i][j]  (IMM.py:244)
        for (int i = 0; i < NM; ++i) acc = fma(wMu[s0 + LPF * i], sM[i * NM + (int)jm], acc);
        cbar = acc;
        tiny = cbar < 0x1p-500;
        rc = fk_rcp(tiny ? cbar * 0x1p600 : cbar);
    }
    {
        double xh[NX], xm[NX];
        FK_UNROLL for (int r = 0; r < NX; ++r) xh[r] = xm[r] = 0.0;
        for (int i = 0; i < NM; ++i) {
            const double mi = wMu[s0 + LPF * i];
            const double num = sM[i * NM + (int)jm] * mi;
            const double w = (tiny ? num * 0x1p600 : num) * rc;
            FK_UNROLL for (int r = 0; r < NX; ++r) {
                const double xi = wX[r * 64 + s0 + LPF * i];
                xh[r] = fma(xi, mi, xh[r]);
                xm[r] = fma(xi, w, xm[r]);
            }
        }
        if (est) {
            FK_UNROLL for (int r = 0; r < NX; ++r) wH[r * GPW + grp] = xh[r];
            ml_wave_fence();
            if (x_dst) {
                _Pragma("nounroll") for (int k = 0; k < (NX + LPB - 1) / LPB; ++k) {
                    const int r = (int)bl + LPB * k;
                    if (r < n && live) x_dst[ox.at(bank, r)] = wH[r * GPW + grp];
                }
            }
        }
        if constexpr (MIX) { FK_UNROLL for (int r = 0; r < NX; ++r) x[r] = xm[r]; }       // (the old x stays published in wX)
    }
    if (!MIX && !(est && P_dst)) return;
    const unsigned fj = bl / (unsigned)LPF;
    FK_UNROLL for (int ph = 0; ph < PH; ++ph) {
        const int lo = ph * CH, hi = (lo + CH < PE) ? lo + CH : PE;
        ml_wave_fence();
        FK_UNROLL for (int e = 0; e < CH; ++e)
            if (lo + e < hi) wP[e * 64 + lane] = P[(lo + e) / NX][(lo + e) % NX];
        ml_wave_fence();
        if (est && P_dst) {
            // rows sub R .. of the estimate's P by the lanes with this sub index: their elements of the piece dealt out over the G
            // quads of the bank
            // (four elements per trip: their LDS reads in one batch -- as a loop of one it is a chain of dependent round trips, 2.3 of
            //  the 7.9 ms of a (16,8) x 2 launch that asks for P)
            _Pragma("unroll 4") for (int k = 0; k < (CH + G - 1) / G; ++k) {
                const int q = (int)fj + G * k;
                const int qc = lo + q < hi ? q : hi - lo - 1;
                const int e = lo + qc, r = (int)sub * R + e / NX, cc = e % NX;
                const double hr = wH[r * GPW + grp], hc = wH[cc * GPW + grp];
                double acc = 0.0;
                if (mmae) {
                    // MMAEFilterBank's covariance zips the COMPONENTS of x with the filters (fk_imm.hpp, mmae_estimate)
                    const int kmax = NM < n ? NM : n;
                    for (int i = 0; i < kmax; ++i) {
                        const double hk = wH[i * GPW + grp];
                        const double ya = wX[r * 64 + s0 + LPF * i] - hk, yb = wX[cc * 64 + s0 + LPF * i] - hk;
                        acc = fma(wMu[s0 + LPF * i], fma(ya, yb, wP[qc * 64 + s0 + LPF * i]), acc);
                    }
                } else if constexpr (G <= 4) {
                    // (small banks: the filters' terms unrolled under a predicate, their LDS reads in one batch)
                    FK_UNROLL for (int i = 0; i < G; ++i) {
                        const int ic = i < NM ? i : 0;
                        const double ya = wX[r * 64 + s0 + LPF * ic] - hr, yb = wX[cc * 64 + s0 + LPF * ic] - hc;
                        const double t = fma(wMu[s0 + LPF * ic], fma(ya, yb, wP[qc * 64 + s0 + LPF * ic]), acc);
                        acc = i < NM ? t : acc;
                    }
                } else {
                    for (int i = 0; i < NM; ++i) {
                        const double ya = wX[r * 64 + s0 + LPF * i] - hr, yb = wX[cc * 64 + s0 + LPF * i] - hc;
                        acc = fma(wMu[s0 + LPF * i], fma(ya, yb, wP[qc * 64 + s0 + LPF * i]), acc);
                    }
                }
                if (lo + q < hi && r < n && cc < n && live) P_dst[oP.at(bank, r * n + cc)] = acc;
            }
        }
        if constexpr (MIX) {
            // mixed initial conditions (IMM.py:200-219): P0_j = sum_i omega[i][j] (outer(x_i - x0_j) + P_i), filter by filter
            // (x holds x0_j by now)
            FK_UNROLL for (int e = 0; e < CH; ++e)
                if (lo + e < hi) P[(lo + e) / NX][(lo + e) % NX] = 0.0;
            double xo[R];
            FK_UNROLL for (int r = 0; r < R; ++r) xo[r] = quad_pick<NX>(x, sub, r);
            for (int i = 0; i < NM; ++i) {
                const double num = sM[i * NM + (int)jm] * wMu[s0 + LPF * i];
                const double w = (tiny ? num * 0x1p600 : num) * rc;
                double d[NX], dr[R];
                FK_UNROLL for (int r = 0; r < NX; ++r) d[r] = wX[r * 64 + s0 + LPF * i] - x[r];
                FK_UNROLL for (int r = 0; r < R; ++r) dr[r] = wX[(sub * R + r) * 64 + s0 + LPF * i] - xo[r];
                FK_UNROLL for (int r = 0; r < R; ++r)
                    FK_UNROLL for (int cc = 0; cc < NX; ++cc) {
                        const int e = r * NX + cc;
                        if (e >= lo && e < hi) P[r][cc] = fma(w, fma(dr[r], d[cc], wP[(e - lo) * 64 + s0 + LPF * i]), P[r][cc]);
                    }
            }
        }
    }
}

// (FK_IQ_OCC2=1, build time: eight lanes per filter, banks of up to four, TWO waves per SIMD -- 256 registers, 80 KB of LDS per workgroup.
//  MEASURED SLOWER, (16,8) x 2 without outputs 5.4 -> 9.6 ms: the step does not fit 256 registers (1.2 KB of spills per lane) and the
//  exchange goes in three pieces.  profiles/r06/imm_quad/oct_*.txt)
template <int G>
constexpr int IQ_WAVES = (LPF == 8 && G <= 4 && FK_IQ_OCC2) ? 2 : 1;

template <int NX, int NZ, int G, bool EXT>
__global__ void __launch_bounds__(BLOCK, IQ_WAVES<G>)
imm_quad_kernel(const ImmArgs a, const int NM, const int aos)
{
    using LM = QuadModel<NX, NZ>;
    static_assert(NX % LPF == 0 && NZ % LPF == 0 && LPF * G <= 64, "LPF lanes per filter, whole rows per lane");
    constexpr int R = NX / LPF, PE = R * NX, WAVES = BLOCK / 64, LPB = LPF * G, GPW = 64 / LPB;      // GPW: banks per wave
    // The lane's rows of P are exchanged in PH pieces of at most CH elements (as many as fit beside the G model blocks in 160 KB)
    constexpr int LDS_DOUBLES = 160 * 1024 / IQ_WAVES<G> / 8 - 64;
    constexpr int FIXED = NX * 64 + NX * GPW + 64 + 64;                            // X | xhat | mu | scratch
    constexpr int ROOM = (LDS_DOUBLES - G * LM::SIZE - G * G) / WAVES - FIXED;
    static_assert(ROOM >= 64 * 8, "no room for the exchange image");
    constexpr int CH = (ROOM / 64 >= PE) ? PE : ROOM / 64, PH = (PE + CH - 1) / CH;
    constexpr int WSZ = FIXED + CH * 64;
    __shared__ double smem[G * LM::SIZE + G * G + WAVES * WSZ];
    const int n = a.n, m = a.m;
    const long N = a.N;
    for (int j = 0; j < NM; ++j) {
        double *s = smem + j * LM::SIZE;
        quad_fill<NX, NX, LM::PX>(s + LM::OFF_F, a.F + (long)j * n * n, n, n, 1.0, threadIdx.x);
        quad_fill<NX, NX, LM::PX>(s + LM::OFF_Q, a.Q + (long)j * n * n, n, n, 0.0, threadIdx.x);
        quad_fill<NZ, NX, LM::PX>(s + LM::OFF_H, a.H + (long)j * m * n, m, n, 0.0, threadIdx.x);
        quad_fill<NZ, NZ, LM::PZ>(s + LM::OFF_R, a.R + (long)j * m * m, m, m, 1.0, threadIdx.x);
    }
    double *sM = smem + G * LM::SIZE;
    if ((int)threadIdx.x < NM * NM) sM[threadIdx.x] = a.Mt ? a.Mt[threadIdx.x] : 0.0;
    __syncthreads();

    const unsigned lane = threadIdx.x & 63u, wave = wave_index();
    const unsigned g0 = lane & ~(unsigned)(LPB - 1), grp = lane / (unsigned)LPB, bl = lane - g0, sub = lane & (unsigned)(LPF - 1), j = bl / (unsigned)LPF;
    const unsigned s0 = g0 + sub;
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

    double x[NX], P[R][NX], mu;
    {
        const RecMap mx = rec_map(aos, N, NM * n), mP = rec_map(aos, N, NM * n * n), mm = rec_map(aos, N, NM);
        mu = a.mu[mm.at(bank, (int)jm)];
        FK_UNROLL for (int r = 0; r < NX; ++r) {
            const int rr = r < n ? r : n - 1;
            const double xv = a.xs[mx.at(bank, (int)jm * n + rr)];
            x[r] = (r < n) ? xv : 0.0;
        }
        FK_UNROLL for (int r = 0; r < R; ++r) {
            const int row = (int)sub * R + r, rr = row < n ? row : n - 1;
            FK_UNROLL for (int c = 0; c < NX; ++c) {
                const int cc = c < n ? c : n - 1;
                const double pv = a.Ps[mP.at(bank, ((int)jm * n + rr) * n + cc)];
                P[r][c] = (row < n && c < n) ? pv : ((row == c) ? 1.0 : 0.0);
            }
        }
    }
    int st = 0;
    const RecMap ox = rec_map(aos, N, n), oP = rec_map(aos, N, n * n), oM = rec_map(aos, N, NM), oz = rec_map(aos, N, m);
    const int nn = n * n;
    double cbar = 0.0;

    const QuadCtx ctx{wX, wP, wH, wMu, sM, lane, g0, grp, bl, sub, jm, bank, NM, n, live, ox, oP};

    // (2 pi)^(-m/2), m = 1..8
    const double cm = m == 1 ? 0.3989422804014327 : m == 2 ? 0.15915494309189535 : m == 3 ? 0.06349363593424097
                    : m == 4 ? 0.025330295910584444 : m == 5 ? 0.010105326013811644 : m == 6 ? 0.004031441804149937
                    : m == 7 ? 0.0016083125866532416 : m == 8 ? 0.000641623890917771 : 1.0;
    const bool want_post = a.x_out || a.P_out, want_prior = a.xp_out || a.Pp_out;
    const bool mmae = EXT && a.mmae;
    // missing measurements: the log-density of a zero residual under this filter's last S (imm_lanes.hip)
    [[maybe_unused]] double ll0v = -__builtin_inf();
    if (EXT && a.ll0) ll0v = a.ll0[rec_map(aos, N, NM).at(bank, (int)jm)];
    [[maybe_unused]] const double log2pi_m = m * 1.8378770664093453;

    double zc[NZ];
    FK_UNROLL for (int r = 0; r < NZ; ++r) {
        const double zv = a.z[oz.at(bank, r < m ? r : m - 1)];
        zc[r] = (r < m) ? zv : 0.0;
    }
    for (long t = 0; t < a.T; ++t) {
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
                for (int i = 0; i < NM; ++i) acc = fma(wMu[s0 + LPF * i], sM[i * NM + (int)jm], acc);
                cbar = acc;
            }
        } else {
            // posterior estimate of step t-1 and mixing for step t from one publication
            if (mmae) {
                cbar = mu;                                          // p_i *= likelihood_i (mmae.py:186-187): no mixing
                if (t > 0 && want_post)
                    quad_exchange<false, NX, G, CH, PH>(ctx, x, P, mu, cbar, true, a.x_out ? a.x_out + (t - 1) * N * n : nullptr,
                                                        a.P_out ? a.P_out + (t - 1) * N * nn : nullptr, true);
            } else {
                quad_exchange<true, NX, G, CH, PH>(ctx, x, P, mu, cbar, t > 0 && want_post, a.x_out ? a.x_out + (t - 1) * N * n : nullptr,
                                                   a.P_out ? a.P_out + (t - 1) * N * nn : nullptr);
            }
            // (the model block's offset is made opaque once per step: hoisted out of the time loop the model would sit in registers)
            {
                unsigned moff = jm * (unsigned)LM::SIZE;
                asm volatile("" : "+v"(moff));
                mod.s = smem + moff;
            }
            quad_predict<NX>(x, P, mod, sub);
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
                quad_exchange<false, NX, G, CH, PH>(ctx, x, P, mu, cbar, true, a.xp_out ? a.xp_out + t * N * n : nullptr,
                                                    a.Pp_out ? a.Pp_out + t * N * nn : nullptr);
        }
        if (phase == FK_IMM_PREDICT) break;
        // this quad's filter: update, likelihood floored at DBL_MIN (kalman_filter.py:1213-1226; fk_imm.hpp, imm_update)
        double lj;
        bool has_z = true;
        if (EXT && a.mask) has_z = a.mask[t * N + bank] != 0;
        if (!has_z) {
            lj = exp(ll0v);
            if (lj == 0.0) lj = 2.2250738585072014e-308;
        } else {
            double y[NZ], Lf[NZ * NZ], dinv[NZ];
            st |= quad_update<NX, NZ>(x, P, z, mod, sub, y, Lf, dinv);
            double q = 0.0;
            {
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
        // mu_j = cbar_j L_j / sum (IMM.py:181-183): the sum over the bank in filter order
        {
            const double mj = cbar * lj;
            ml_wave_fence();
            wS[lane] = mj;
            ml_wave_fence();
            double sum = 0.0;
            for (int i = 0; i < NM; ++i) sum += wS[s0 + LPF * i];
            const bool tny = sum < 0x1p-500;
            const double rsum = fk_rcp(tny ? sum * 0x1p600 : sum);
            mu = (tny ? mj * 0x1p600 : mj) * rsum;
        }
        if (writer && sub == 0u) {
            if (a.mu_out) (a.mu_out + t * N * NM)[oM.at(bank, (int)j)] = mu;
            if (a.L_out) (a.L_out + t * N * NM)[oM.at(bank, (int)j)] = lj;
        }
    }
    if (want_post && a.T > 0 && !(EXT && a.phase == FK_IMM_PREDICT))      // the last step's posterior estimate
        quad_exchange<false, NX, G, CH, PH>(ctx, x, P, mu, cbar, true, a.x_out ? a.x_out + (a.T - 1) * N * n : nullptr,
                                            a.P_out ? a.P_out + (a.T - 1) * N * nn : nullptr, mmae);
    {
        const RecMap mx = rec_map(aos, N, NM * n), mP = rec_map(aos, N, NM * n * n), mm = rec_map(aos, N, NM);
        bool fin = all_finite<NX>(x) && (fabs(mu) <= 1.79769313486231570815e+308);
        FK_UNROLL for (int r = 0; r < R; ++r) fin = fin && all_finite<NX>(P[r]);
        if (writer) {
            if (sub == 0u) {
                a.mu[mm.at(bank, (int)j)] = mu;
                if (EXT && a.ll0) a.ll0[mm.at(bank, (int)j)] = ll0v;
                FK_UNROLL for (int r = 0; r < NX; ++r)
                    if (r < n) a.xs[mx.at(bank, (int)j * n + r)] = x[r];
            }
            FK_UNROLL for (int r = 0; r < R; ++r) {
                const int row = (int)sub * R + r;
                FK_UNROLL for (int c = 0; c < NX; ++c)
                    if (row < n && c < n) a.Ps[mP.at(bank, ((int)j * n + row) * n + c)] = P[r][c];
            }
        }
        if (a.status) {
            // the bank's status: the OR over the lanes of its filters
            int *wI = reinterpret_cast<int *>(wS);
            ml_wave_fence();
            wI[lane] = st | (fin ? 0 : ST_NONFINITE);
            ml_wave_fence();
            int sv = 0;
            for (int i = 0; i < LPF * NM; ++i) sv |= wI[g0 + i];
            if (live && bl == 0u) a.status[bank] = a.status_or ? (a.status[bank] | sv) : sv;
        }
    }
}

}  // namespace iq<LPF>
}  // namespace fk

using namespace fk;
using namespace fk::FK_IQ_NS(FK_IQ_LPF);

#if !defined(FK_NX) || !defined(FK_NZ) || !defined(FK_IL_G) || !defined(FK_IL_EXT)
#error "compile with -DFK_NX= -DFK_NZ= (the class: every dim_x <= FK_NX, dim_z <= FK_NZ; both multiples of 4) -DFK_IL_G=2|4|8|16 (filters per bank) -DFK_IL_EXT=0|1"
#endif
#define FK_IQ_CAT_(a, b, c, d, e) a##b##_##c##_g##d##_x##e
#define FK_IQ_CAT(a, b, c, d, e) FK_IQ_CAT_(a, b, c, d, e)

// launch_imm_quad_<NX>_<NZ>_g<G>_x<EXT>: banks of G/2 + 1 .. G filters of the class (one object per G and kind); x1 also serves MMAE,
// missing measurements, the control input and the single-phase calls; returns 1 when the call is not one this file serves
#if FK_IQ_LPF == 8
#define FK_IQ_LAUNCH launch_imm_oct_
#else
#define FK_IQ_LAUNCH launch_imm_quad_
#endif
int FK_IQ_CAT(FK_IQ_LAUNCH, FK_NX, FK_NZ, FK_IL_G, FK_IL_EXT)(const ImmArgs &a, int n_models, int layout, hipStream_t s)
{
    if (a.n > FK_NX || a.m > FK_NZ || n_models < 2 || n_models > FK_IL_G) return 1;
    if (!FK_IL_EXT && (a.mmae || a.mask || a.ll0 || a.nu > 0 || a.phase != FK_IMM_STEP)) return 1;
    const int aos = layout == FK_LAYOUT_AOS ? 1 : 0;
    const long per_block = (BLOCK / 64) * (64 / (FK_IQ_LPF * FK_IL_G));
    const dim3 grid((unsigned)((a.cnt + per_block - 1) / per_block)), block(BLOCK);
    hipLaunchKernelGGL((imm_quad_kernel<FK_NX, FK_NZ, FK_IL_G, (FK_IL_EXT != 0)>), grid, block, 0, s, a, n_models, aos);
    return 0;
}
