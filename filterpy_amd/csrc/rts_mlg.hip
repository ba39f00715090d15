// rts_mlg.hip -- KalmanFilter.rts_smoother for dim_x = 10..16 (and 8, 9 in NumPy order) with FOUR LANES PER TRACK (gfx950); the backward
// companion of kf_mlg.hip, same row ownership (lane L: rows L*R .. L*R+R-1, R = ceil(n/4), rows past n-1 clamped to
// row n-1 and recomputed -- nothing is predicated; the smoother has no cross-lane sum, so the duplicates need no
// special care).  Per step k = T-2 .. 0 (filterpy/kalman/kalman_filter.py:1066-1072):
//
//   Pp = F P F' + Q ;  K = (P F') Pp^-1 ;  x += K (xn - F x) ;  P += K (Pn - Pp) K'
//
// T = P F' and every row of Pp, K, E = K (Pn - Pp) and P live in the lane that owns the row; rows of T, of Pn - Pp and
// of K are broadcast by their owners (quad-permute DPP); the L D L' factor of Pp is row-distributed too (the owner of
// row j broadcasts it when column j is due) and each lane substitutes its own rows of K.  One wave per SIMD (the
// register file's AGPR half carries the overflow of the 256 VGPRs; the smoothed P of the next step and Pn - Pp wait in
// a wave-private LDS park, the filtered P is read twice).  Shared constant F, Q; K and Pp
// outputs both present; SOA and AOS (row blocks leave through the same wave-private LDS region as 1 KiB stores).  Everything else at these sizes stays on the padded kernel.
#include <stdlib.h>

#include "fk_device.hpp"
#include "fk_math_sym.hpp"
#include "fk_kernel_args.hpp"
#include "fk_ml.hpp"
#include "fk_chunks.hpp"
#include "../../include/filterhip.h"

#ifndef FK_NX
#error "compile with -DFK_NX=<dim_x>"
#endif

// occupancy bound at dim_x 8: three waves per SIMD leave the NumPy-order kernel 168 VGPRs + 204 B of scratch, two waves 234 and none --
// measured the same (14.45 vs 14.48-14.56 ms at N = 2.81e5 x 100, A/B/A/B in one lease, profiles/r05/dims/rmlg8_waves_ab.jsonl): kept at 3
#ifndef FK_RMLG8_WAVES
#define FK_RMLG8_WAVES 3
#endif
#define FK_RMLG_CAT_(a, b) a##b
#define FK_RMLG_CAT(a, b) FK_RMLG_CAT_(a, b)

namespace fk {
namespace FK_RMLG_CAT(rmlg_, FK_NX) {

#define FK_OWNER_VAL(v_, k) (((k) / R == 0) ? quad_bcast<0>(v_) : ((k) / R == 1) ? quad_bcast<1>(v_) \
                             : ((k) / R == 2) ? quad_bcast<2>(v_) : quad_bcast<3>(v_))
#define FK_OWNER_ROW(dst, M, k, LEN)                          \
    FK_UNROLL for (int j_ = 0; j_ < (LEN); ++j_) {            \
        const double v_ = M[(k) % R][j_];                     \
        dst[j_] = FK_OWNER_VAL(v_, k);                        \
    }

// AOS output of one row-block matrix (the wave's 16 x NX*NX slab) through the wave's LDS tile
template <int R, int NX>
__device__ __forceinline__ void store_rows_aos(const double (&M)[R][NX], const unsigned (&row)[R], double *dst, double *tile,
                                               unsigned lane, unsigned valid)
{
    constexpr int EP = NX * NX, UP = 16 * EP / 2;
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
    const unsigned q = lane >> 2;
    ml_wave_fence();
    FK_UNROLL for (int r = 0; r < R; ++r)
        FK_UNROLL for (int c = 0; c < NX; ++c) tile[q * EP + row[r] * NX + c] = M[r][c];
    ml_wave_fence();
    const rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc(dst, 0, (int)(valid * (unsigned)EP * 8u), 0x00020000);
    ml_copy_units<UP, 4>(lane, [&](unsigned unit) { return tile + 2u * unit; },
                         [&](unsigned unit, bool ok, const u32x4 &v) {
                             __builtin_amdgcn_raw_buffer_store_b128(v, rP, ok ? unit * 16u : ML_OFF_DROP, 0, 0);
                         });
}

template <int NX, int LAYOUT>
__global__ void __launch_bounds__(BLOCK, (NX <= 8 ? FK_RMLG8_WAVES : NX <= 9 ? 2 : 1))
rts_mlg_kernel(const RtsArgs a)
{
    constexpr int R = (NX + 3) / 4;
    constexpr bool AOS = LAYOUT == LAYOUT_AOS;
    constexpr bool SOA_SLAB = !AOS && NX <= FK_SOA_SLAB_MAX;      // SOA row blocks through the slab too (16-byte units): n = 10: 0.36 -> 0.45 of HBM, n = 14: 0.32 -> 0.30
    // One wave-private LDS region, used in turn as the AOS staging slab (16 x NX*NX doubles) and as the PARK of a
    // row block ([element][lane], conflict-free, R*NX x 64 doubles): the smoothed P of step k+1 waits there between
    // iterations and D = Pn - Pp across the factorisation.
    constexpr int REGION = (16 * NX * NX > R * NX * 64) ? 16 * NX * NX : R * NX * 64;
    __shared__ double smem[2 * NX * NX + (BLOCK / 64) * (REGION + 16 * NX)];
    double *tile = smem + 2 * NX * NX + (threadIdx.x >> 6) * REGION;
    // the smoothed x of step k+1 (replicated in the quad: one copy per track) waits here across the factorisation
    double *xpark = smem + 2 * NX * NX + (BLOCK / 64) * REGION + (threadIdx.x >> 2) * NX;
    lds_fill<NX, NX>(smem, a.F, NX, NX, 1.0, threadIdx.x);
    lds_fill<NX, NX>(smem + NX * NX, a.Q, NX, NX, 0.0, threadIdx.x);
    __syncthreads();
    const double *sF = smem, *sQ = smem + NX * NX;

    const long N = a.N, T = a.T;
    const unsigned L = threadIdx.x & 3u;
    const long i0 = a.cnt ? a.i0 : 0, iend = a.cnt ? a.i0 + a.cnt : N;     // this launch's track group (chunked calls, fk_chunks.hpp)
    long trk = i0 + (long)blockIdx.x * (BLOCK / 4) + (threadIdx.x >> 2);
    if (trk >= iend) trk = iend - 1;                           // tail quads recompute the last track
    unsigned estride = AOS ? 8u : (unsigned)N * 8u;
    asm volatile("" : "+s"(estride));
    const unsigned t8 = (unsigned)trk * (AOS ? (unsigned)NX * 8u : 8u);
    const unsigned trk32 = (unsigned)trk;
    // the rows this lane holds (clamped) and where they start in a covariance record: DERIVED AGAIN in every block that
    // needs them (from laundered copies of the lane and track numbers) -- kept live across the whole step these eight
    // values and what the compiler derives from them were spilled before the loop and reloaded inside it, and at
    // n >= 15 every such scratch reload waited (vmcnt) behind the step's output stores: 65 % of the wave time
#define FK_ROWS_HERE()                                                                                     \
    unsigned row[R], off_row[R];                                                                           \
    {                                                                                                      \
        unsigned Lx_ = L, tx_ = trk32;                                                                     \
        asm volatile("" : "+v"(Lx_), "+v"(tx_));                                                           \
        FK_UNROLL for (int r_ = 0; r_ < R; ++r_) {                                                         \
            const unsigned g_ = Lx_ * (unsigned)R + (unsigned)r_;                                          \
            row[r_] = g_ < (unsigned)NX ? g_ : (unsigned)NX - 1u;                                          \
            off_row[r_] = (AOS ? tx_ * (unsigned)(NX * NX) * 8u : tx_ * 8u) + row[r_] * (unsigned)NX * estride; \
        }                                                                                                  \
    }
    const unsigned lane = threadIdx.x & 63u;
    const long w0 = i0 + (long)blockIdx.x * (BLOCK / 4) + (long)wave_index() * 16;          // scalar: see wave_index()
    const unsigned valid = (unsigned)(iend - w0 >= 16 ? 16 : (iend - w0 > 0 ? iend - w0 : 0));
    const long xs_blk = N * NX, ps_blk = N * (long)NX * NX;
    double *park = tile + lane;                                // element e of this lane: park[e * 64]

#define FK_LOAD_ROWS(base, M)                                                          \
    FK_UNROLL for (int r_ = 0; r_ < R; ++r_) {                                         \
        const MlView v_((base), off_row[r_], estride);                                 \
        FK_UNROLL for (int c_ = 0; c_ < NX; ++c_) M[r_][c_] = v_.load(c_);             \
    }
#define FK_STORE_ROWS(base, step, M)                                                   \
    if constexpr (AOS) {                                                               \
        store_rows_aos<R, NX>(M, row, (base) + ((step) * N + w0) * NX * NX, tile, lane, valid); \
    } else if constexpr (SOA_SLAB) {                                                   \
        ml_store_rows_soa_slab<R, NX, 16>(M, row, (base) + (step) * ps_blk, N, w0, tile, lane, lane >> 2, valid); \
    } else {                                                                           \
        FK_UNROLL for (int r_ = 0; r_ < R; ++r_) {                                     \
            const MlView v_((base) + (step) * ps_blk, off_row[r_], estride);           \
            FK_UNROLL for (int c_ = 0; c_ < NX; ++c_) v_.store(c_, M[r_][c_]);         \
        }                                                                              \
    }
#define FK_PARK(M)                                                                     \
    ml_wave_fence();                                                                   \
    FK_UNROLL for (int r_ = 0; r_ < R; ++r_)                                           \
        FK_UNROLL for (int c_ = 0; c_ < NX; ++c_) park[(r_ * NX + c_) * 64] = M[r_][c_]; \
    ml_wave_fence();

    // k = T-1: smoothed == filtered; K = 0; Pp = Ps   (kalman_filter.py:1063-1065)
    if (a.cont) {
        // a later chunk of the call: step T-1 of this window was smoothed by the chunk that ran before (after it in time)
        FK_ROWS_HERE();
        double xn[NX], Pn[R][NX];
        const MlView vx(a.xs + (T - 1) * xs_blk, t8, estride);
        FK_UNROLL for (int k = 0; k < NX; ++k) xn[k] = vx.load(k);
        FK_LOAD_ROWS(a.Ps_out + (T - 1) * ps_blk, Pn);
        FK_PARK(Pn);
        FK_UNROLL for (int k = 0; k < NX; ++k) xpark[k] = xn[k];
    } else {
        FK_ROWS_HERE();
        double xn[NX], Pn[R][NX];
        const MlView vx(a.Xs + (T - 1) * xs_blk, t8, estride);
        FK_UNROLL for (int k = 0; k < NX; ++k) xn[k] = vx.load(k);
        FK_LOAD_ROWS(a.Ps + (T - 1) * ps_blk, Pn);
        const MlView ox(a.xs + (T - 1) * xs_blk, t8, estride);
        FK_UNROLL for (int k = 0; k < NX; ++k) ox.store(k, xn[k]);
        FK_STORE_ROWS(a.Ps_out, T - 1, Pn);
        FK_STORE_ROWS(a.Pp, T - 1, Pn);
        {
            double Z[R][NX];
            FK_UNROLL for (int r = 0; r < R; ++r)
                FK_UNROLL for (int c = 0; c < NX; ++c) Z[r][c] = 0.0;
            FK_STORE_ROWS(a.K, T - 1, Z);
        }
        FK_PARK(Pn);
        FK_UNROLL for (int k = 0; k < NX; ++k) xpark[k] = xn[k];
    }
    int st = 0;
    // (the unrolled step is 50 KB of code at n = 12, 95 KB at n = 14, 130 KB at n = 16 against a 64 KB instruction
    // cache; keeping the four waves of a workgroup in step with a barrier per stage was tried: no gain at n = 16,
    // slower at n = 14 -- the counters showed scratch reloads, not instruction fetch, see FK_ROWS_HERE)
#define FK_ISYNC()

    _Pragma("nounroll") for (long k = T - 2; k >= 0; --k) {
        double Tm[R][NX];
        FK_ISYNC();
        {
            FK_ROWS_HERE();
            double P[R][NX];
            FK_LOAD_ROWS(a.Ps + k * ps_blk, P);
            // T = P F'  (F's rows requested from LDS one iteration ahead -- two row buffers --: a ds_read -> s_waitcnt pair per
            // row exposed the LDS latency NX times per stage; same sums in the same order)
            double Fr[2][NX];
            FK_UNROLL for (int q = 0; q < NX; ++q) Fr[0][q] = sF[q];
            FK_UNROLL for (int i = 0; i < NX; ++i) {
                if (i + 1 < NX) {
                    FK_UNROLL for (int q = 0; q < NX; ++q) Fr[(i + 1) & 1][q] = sF[(i + 1) * NX + q];
                }
                FK_UNROLL for (int r = 0; r < R; ++r) {
                    double acc = P[r][0] * Fr[i & 1][0];
                    FK_UNROLL for (int q = 1; q < NX; ++q) acc = fma(P[r][q], Fr[i & 1][q], acc);
                    Tm[r][i] = acc;
                }
                FK_STAGE();
            }
        }
        FK_ISYNC();
        {
            // Pp = F T + Q (own rows)
            FK_ROWS_HERE();
            double Pp[R][NX];
            double fq[2][R];                                   // this lane's column q of F one iteration ahead
            FK_UNROLL for (int r = 0; r < R; ++r) fq[0][r] = sF[row[r] * NX];
            FK_UNROLL for (int q = 0; q < NX; ++q) {
                if (q + 1 < NX) {
                    FK_UNROLL for (int r = 0; r < R; ++r) fq[(q + 1) & 1][r] = sF[row[r] * NX + q + 1];
                }
                double Tq[NX];
                FK_OWNER_ROW(Tq, Tm, q, NX);
                FK_UNROLL for (int r = 0; r < R; ++r) {
                    const double f = fq[q & 1][r];
                    FK_UNROLL for (int j = 0; j < NX; ++j) Pp[r][j] = (q == 0) ? f * Tq[j] : fma(f, Tq[j], Pp[r][j]);
                }
                FK_STAGE();
            }
            // (a row of Q in one batch, then its sums)
            FK_UNROLL for (int r = 0; r < R; ++r) {
                double Qr[NX];
                FK_UNROLL for (int j = 0; j < NX; ++j) Qr[j] = sQ[row[r] * NX + j];
                FK_STAGE();
                FK_UNROLL for (int j = 0; j < NX; ++j) Pp[r][j] += Qr[j];
            }
            if constexpr (AOS || SOA_SLAB) {
                // the staging slab is the park: lift the parked Pn out, ship Pp, put D = Pn - Pp back
                double Pn[R][NX];
                FK_UNROLL for (int r = 0; r < R; ++r)
                    FK_UNROLL for (int j = 0; j < NX; ++j) Pn[r][j] = park[(r * NX + j) * 64];
                FK_STORE_ROWS(a.Pp, k, Pp);
                ml_wave_fence();
                FK_UNROLL for (int r = 0; r < R; ++r)
                    FK_UNROLL for (int j = 0; j < NX; ++j) park[(r * NX + j) * 64] = Pn[r][j] - Pp[r][j];
                ml_wave_fence();
            } else {
                FK_STORE_ROWS(a.Pp, k, Pp);
                // D = Pn - Pp (own rows) replaces the parked Pn
                FK_UNROLL for (int r = 0; r < R; ++r)
                    FK_UNROLL for (int j = 0; j < NX; ++j) park[(r * NX + j) * 64] -= Pp[r][j];
            }
            FK_STAGE();
            // K = T Pp^-1 with the factor ROW-DISTRIBUTED like everything else: Pp = L D L', row i of L (strict lower
            // part, overwriting Pp's own row i) lives with the owner of row i; d, 1/d are replicated.  Column by column:
            // the owner of row j broadcasts L[j][0..j-1] and Pp[j][j], every lane forms d[j] and finishes column j of
            // its own rows below j (a select, no branch: rows at or above j keep their entry).  A replicated packed copy
            // (n (n+1) / 2 doubles per lane: 136 at n = 16) spilled 40-100 doubles into scratch inside these loops and
            // cost 3-5x at n >= 13.
            double d[NX], dinv[NX];
            bool pd = true;
            FK_ISYNC();
            FK_UNROLL for (int j = 0; j < NX; ++j) {
                double Lj[NX];
                FK_UNROLL for (int q = 0; q < NX; ++q)
                    if (q < j) { const double v = Pp[j % R][q]; Lj[q] = FK_OWNER_VAL(v, j); }
                const double ajj = Pp[j % R][j];
                double dj = FK_OWNER_VAL(ajj, j);
                FK_UNROLL for (int q = 0; q < NX; ++q)
                    if (q < j) dj = fma(-Lj[q] * Lj[q], d[q], dj);
                pd = pd && (dj > 0.0);
                d[j] = dj;
                const double di = 1.0 / dj;
                dinv[j] = di;
                FK_UNROLL for (int r = 0; r < R; ++r) {
                    double t = Pp[r][j];
                    FK_UNROLL for (int q = 0; q < NX; ++q)
                        if (q < j) t = fma(-(Pp[r][q] * d[q]), Lj[q], t);
                    Pp[r][j] = row[r] > (unsigned)j ? t * di : Pp[r][j];
                }
                FK_STAGE();
            }
            if (!pd) st |= ST_NOT_PD;
            FK_ISYNC();
            // each lane solves  k Pp = t  for its own rows t of T (they become K's rows): forward with L's rows
            // broadcast, the diagonal, backward with L's columns gathered entry by entry
            FK_UNROLL for (int i = 1; i < NX; ++i) {
                double Li[NX];
                FK_UNROLL for (int q = 0; q < NX; ++q)
                    if (q < i) { const double v = Pp[i % R][q]; Li[q] = FK_OWNER_VAL(v, i); }
                FK_UNROLL for (int r = 0; r < R; ++r) {
                    double t = Tm[r][i];
                    FK_UNROLL for (int q = 0; q < NX; ++q)
                        if (q < i) t = fma(-Li[q], Tm[r][q], t);
                    Tm[r][i] = t;
                }
                FK_STAGE();
            }
            FK_UNROLL for (int r = 0; r < R; ++r)
                FK_UNROLL for (int i = 0; i < NX; ++i) Tm[r][i] *= dinv[i];
            FK_UNROLL for (int i = NX - 2; i >= 0; --i) {
                double t[R];
                FK_UNROLL for (int r = 0; r < R; ++r) t[r] = Tm[r][i];
                FK_UNROLL for (int q = 0; q < NX; ++q)
                    if (q > i) {
                        const double v = Pp[q % R][i];
                        const double lqi = FK_OWNER_VAL(v, q);
                        FK_UNROLL for (int r = 0; r < R; ++r) t[r] = fma(-lqi, Tm[r][q], t[r]);
                    }
                FK_UNROLL for (int r = 0; r < R; ++r) Tm[r][i] = t[r];
                FK_STAGE();
            }
        }
        // E = K D (own rows): rows of D broadcast by their owners
        FK_ISYNC();
        double E[R][NX];
        {
            double D[R][NX];
            FK_UNROLL for (int r = 0; r < R; ++r)
                FK_UNROLL for (int j = 0; j < NX; ++j) D[r][j] = park[(r * NX + j) * 64];
            FK_UNROLL for (int q = 0; q < NX; ++q) {
                double Dq[NX];
                FK_OWNER_ROW(Dq, D, q, NX);
                FK_UNROLL for (int r = 0; r < R; ++r) {
                    const double kq = Tm[r][q];
                    FK_UNROLL for (int j = 0; j < NX; ++j) E[r][j] = (q == 0) ? kq * Dq[j] : fma(kq, Dq[j], E[r][j]);
                }
                FK_STAGE();
            }
        }
        {
            FK_ROWS_HERE();
            FK_STORE_ROWS(a.K, k, Tm);
        }
        FK_STAGE();
        FK_ISYNC();
        {
            // x += K (xn - F x), replicated; G = E K' : column j needs K's row j from its owner.  G goes to the park (D is
            // dead), so that the filtered P (second read) only enters the registers once E and K have left them.
            double x[NX], dx[NX];
            {
                const MlView vx(a.Xs + k * xs_blk, t8, estride);
                FK_UNROLL for (int i = 0; i < NX; ++i) x[i] = vx.load(i);
            }
            ml_wave_fence();
            double Fr[2][NX], xpk[NX];
            FK_UNROLL for (int q = 0; q < NX; ++q) Fr[0][q] = sF[q];
            FK_UNROLL for (int i = 0; i < NX; ++i) xpk[i] = xpark[i];
            FK_UNROLL for (int i = 0; i < NX; ++i) {
                if (i + 1 < NX) {
                    FK_UNROLL for (int q = 0; q < NX; ++q) Fr[(i + 1) & 1][q] = sF[(i + 1) * NX + q];
                }
                double acc = Fr[i & 1][0] * x[0];
                FK_UNROLL for (int q = 1; q < NX; ++q) acc = fma(Fr[i & 1][q], x[q], acc);
                dx[i] = xpk[i] - acc;
                FK_STAGE();
            }
            double xn[NX];
            FK_UNROLL for (int j = 0; j < NX; ++j) {
                double Kj[NX];
                FK_OWNER_ROW(Kj, Tm, j, NX);
                double xa = x[j];
                FK_UNROLL for (int q = 0; q < NX; ++q) xa = fma(Kj[q], dx[q], xa);
                xn[j] = xa;
                FK_UNROLL for (int r = 0; r < R; ++r) {
                    double acc = E[r][0] * Kj[0];
                    FK_UNROLL for (int q = 1; q < NX; ++q) acc = fma(E[r][q], Kj[q], acc);
                    park[(r * NX + j) * 64] = acc;
                }
                FK_STAGE();
            }
            const MlView ox(a.xs + k * xs_blk, t8, estride);
            FK_UNROLL for (int i = 0; i < NX; ++i) ox.store(i, xn[i]);
            FK_UNROLL for (int i = 0; i < NX; ++i) xpark[i] = xn[i];
            FK_STAGE();
            FK_ROWS_HERE();
            double P[R][NX];
            FK_LOAD_ROWS(a.Ps + k * ps_blk, P);
            ml_wave_fence();
            FK_UNROLL for (int r = 0; r < R; ++r)
                FK_UNROLL for (int j = 0; j < NX; ++j) P[r][j] += park[(r * NX + j) * 64];      // P += (K (Pn - Pp)) K'  (kalman_filter.py:1071)
            FK_STORE_ROWS(a.Ps_out, k, P);
            FK_PARK(P);                                        // the smoothed P of this step: next iteration's Pn
        }
    }
    if (a.status) {
        double xn[NX];
        FK_UNROLL for (int i = 0; i < NX; ++i) xn[i] = xpark[i];
        bool fin = all_finite<NX>(xn);
        FK_UNROLL for (int r = 0; r < R; ++r)
            FK_UNROLL for (int c = 0; c < NX; ++c) fin = fin && (fabs(park[(r * NX + c) * 64]) <= 1.79769313486231570815e+308);
        int s = st | (fin ? 0 : ST_NONFINITE);
        s |= __builtin_amdgcn_mov_dpp(s, 0xB1, 0xf, 0xf, true);
        s |= __builtin_amdgcn_mov_dpp(s, 0x4E, 0xf, 0xf, true);
        if (L == 0) a.status[trk] = a.status_or ? (a.status[trk] | s) : s;
    }
#undef FK_ROWS_HERE
#undef FK_ISYNC
#undef FK_LOAD_ROWS
#undef FK_STORE_ROWS
#undef FK_PARK
}

}  // namespace (instantiation)

// returns 1 when this call is not one the four-lane smoother serves
int FK_RMLG_CAT(launch_rts_mlg_, FK_NX)(const RtsArgs &a, int layout, bool uniform, hipStream_t s)
{
    using namespace FK_RMLG_CAT(rmlg_, FK_NX);
    if (!uniform || a.model_t || a.n != FK_NX || !a.K || !a.Pp || a.T < 2) return 1;
    auto one = [layout](const RtsArgs &b, hipStream_t sb) -> int {
        const long cnt = b.cnt ? b.cnt : b.N;
        const dim3 grid((unsigned)((cnt + BLOCK / 4 - 1) / (BLOCK / 4))), block(BLOCK);
        if (layout == FK_LAYOUT_AOS) hipLaunchKernelGGL((rts_mlg_kernel<FK_NX, LAYOUT_AOS>), grid, block, 0, sb, b);
        else hipLaunchKernelGGL((rts_mlg_kernel<FK_NX, LAYOUT_SOA>), grid, block, 0, sb, b);
        return check_launch("rts_mlg_kernel");
    };
    // tail filling (fk_chunks.hpp): slots = one wave per SIMD above dim 9, two at 9, three below
    return rts_chunked_call(a, FK_NX, FK_NX <= 8 ? 3072 : FK_NX <= 9 ? 2048 : 1024, one, s);
}

}  // namespace fk
