// rts_mlx.hip -- KalmanFilter.rts_smoother for dim_x = 15, 16 with EIGHT LANES PER TRACK and every row exchange
// through LDS (gfx950).  Same arithmetic and row ownership idea as rts_mlg.hip (lane L of a track's group holds rows
// L*R .. L*R+R-1, R = ceil(n/8) = 2, rows past n-1 clamped to row n-1), per step k = T-2 .. 0
// (filterpy/kalman/kalman_filter.py:1066-1072):
//
//   Pp = F P F' + Q ;  K = (P F') Pp^-1 ;  x += K (xn - F x) ;  P += K (Pn - Pp) K'
//
// Why a second organisation: the unrolled step of the four-lane kernel is 72-77 KB of code at n = 15 and 95 KB at
// n = 16, the instruction cache holds 64 KB, and a loop that does not fit is refetched from L2 line by line every
// iteration -- a single wave alone on a CU took 85 / 148 us per step against 38 us at n = 14 (68 KB, fits), whatever
// the occupancy (profiles/r02/rts_mlg_icache.txt).  With eight lanes per track a lane holds two rows: half the FMAs
// per lane, all row blocks in 256 VGPRs (no AGPR shuffling: 2500 instructions at n = 16), and the rows other lanes
// need are read from two wave-private LDS parks ([element][lane], conflict-free; one ds_read per two doubles where the
// quad-permute DPP broadcast of the four-lane kernel needs four moves -- and a DPP cannot cross a quad anyway).
// Park A carries T, then the L D L' factor of Pp as it is formed (row-distributed: the owner of row j publishes column
// j of its rows, everybody reads row j when column j+1 is due), then K; park B carries the smoothed P of step k+1
// between iterations, then D = Pn - Pp.  Either park doubles as the AOS staging slab while it is idle.
// Shared constant F, Q; K and Pp outputs both present; SOA and AOS.
#include <stdlib.h>

#include "fk_device.hpp"
#include "fk_math_sym.hpp"
#include "fk_kernel_args.hpp"
#include "fk_ml.hpp"
#include "../../include/filterhip.h"

#ifndef FK_NX
#error "compile with -DFK_NX=<dim_x>"
#endif

#define FK_RMLX_CAT_(a, b) a##b
#define FK_RMLX_CAT(a, b) FK_RMLX_CAT_(a, b)

namespace fk {
namespace FK_RMLX_CAT(rmlx_, FK_NX) {

constexpr int LPT = 8;                        // lanes per track
constexpr int TPW = 64 / LPT;                 // tracks per wave

// AOS output of one row-block matrix (the wave's TPW x NX*NX slab) through an idle park
template <int R, int NX>
__device__ __forceinline__ void store_rows_aos(const double (&M)[R][NX], const unsigned (&row)[R], double *dst, double *tile,
                                               unsigned lane, unsigned valid)
{
    constexpr int EP = NX * NX, UP = TPW * EP / 2;
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
    const unsigned g = lane / LPT;
    ml_wave_fence();
    FK_UNROLL for (int r = 0; r < R; ++r)
        FK_UNROLL for (int c = 0; c < NX; ++c) tile[g * EP + row[r] * NX + c] = M[r][c];
    ml_wave_fence();
    const rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc(dst, 0, (int)(valid * (unsigned)EP * 8u), 0x00020000);
    ml_copy_units<UP, 4>(lane, [&](unsigned unit) { return tile + 2u * unit; },
                         [&](unsigned unit, bool ok, const u32x4 &v) {
                             __builtin_amdgcn_raw_buffer_store_b128(v, rP, ok ? unit * 16u : ML_OFF_DROP, 0, 0);
                         });
    ml_wave_fence();
}

template <int NX, int LAYOUT>
__global__ void __launch_bounds__(BLOCK, 1)
rts_mlx_kernel(const RtsArgs a)
{
    constexpr int R = (NX + LPT - 1) / LPT;
    constexpr bool AOS = LAYOUT == LAYOUT_AOS;
    constexpr int PK = R * NX;                                  // elements a lane parks
    constexpr int PARK = (PK * 64 > TPW * NX * NX) ? PK * 64 : TPW * NX * NX;     // doubles per park (>= the AOS slab)
    __shared__ double smem[2 * NX * NX + (BLOCK / 64) * 2 * PARK];
    lds_fill<NX, NX>(smem, a.F, NX, NX, 1.0, threadIdx.x);
    lds_fill<NX, NX>(smem + NX * NX, a.Q, NX, NX, 0.0, threadIdx.x);
    __syncthreads();
    const double *sF = smem, *sQ = smem + NX * NX;
    double *parkA = smem + 2 * NX * NX + (threadIdx.x >> 6) * 2 * PARK, *parkB = parkA + PARK;

    const long N = a.N, T = a.T;
    const unsigned lane = threadIdx.x & 63u;
    const unsigned L = lane % LPT;                              // lane within its track's group
    // (a.cnt != 0: this launch is a track window [a.i0, a.i0 + a.cnt) of a larger bank, N stays the array stride)
    const long i0 = a.cnt ? a.i0 : 0, iend = a.cnt ? a.i0 + a.cnt : N;
    long trk = i0 + (long)blockIdx.x * (BLOCK / LPT) + (threadIdx.x / LPT);
    if (trk >= iend) trk = iend - 1;                            // tail groups recompute the last track
    unsigned estride = AOS ? 8u : (unsigned)N * 8u;
    asm volatile("" : "+s"(estride));
    const unsigned t8 = (unsigned)trk * (AOS ? (unsigned)NX * 8u : 8u);
    unsigned row[R], off_row[R];
    FK_UNROLL for (int r = 0; r < R; ++r) {
        const unsigned g = L * (unsigned)R + (unsigned)r;
        row[r] = g < (unsigned)NX ? g : (unsigned)NX - 1u;
        off_row[r] = (AOS ? (unsigned)trk * (unsigned)(NX * NX) * 8u : (unsigned)trk * 8u) + row[r] * (unsigned)NX * estride;
    }
    const long w0 = i0 + (long)blockIdx.x * (BLOCK / LPT) + (long)wave_index() * TPW;      // scalar: see wave_index()
    const unsigned valid = (unsigned)(iend - w0 >= TPW ? TPW : (iend - w0 > 0 ? iend - w0 : 0));
    const long xs_blk = N * NX, ps_blk = N * (long)NX * NX;
    // element e (= slot * NX + col) of this lane in a park: [e * 64 + lane]; of the lane that owns row q: group base + q / R
    double *mineA = parkA + lane, *mineB = parkB + lane;
    const double *grpA = parkA + (lane - L), *grpB = parkB + (lane - L);
#define FK_FROM(grp, q, col) ((grp)[(((q) % R) * NX + (col)) * 64 + (q) / R])

#define FK_LOAD_ROWS(base, M)                                                          \
    FK_UNROLL for (int r_ = 0; r_ < R; ++r_) {                                         \
        const MlView v_((base), off_row[r_], estride);                                 \
        FK_UNROLL for (int c_ = 0; c_ < NX; ++c_) M[r_][c_] = v_.load(c_);             \
    }
#define FK_STORE_ROWS(base, step, M, tile)                                             \
    if constexpr (AOS) {                                                               \
        store_rows_aos<R, NX>(M, row, (base) + ((step) * N + w0) * NX * NX, tile, lane, valid); \
    } else {                                                                           \
        ml_store_rows_soa_slab<R, NX, TPW>(M, row, (base) + (step) * ps_blk, N, w0, tile, lane, lane / LPT, valid); \
    }
#define FK_PUT(mine, M)                                                                \
    ml_wave_fence();                                                                   \
    FK_UNROLL for (int r_ = 0; r_ < R; ++r_)                                           \
        FK_UNROLL for (int c_ = 0; c_ < NX; ++c_) mine[(r_ * NX + c_) * 64] = M[r_][c_]; \
    ml_wave_fence();

    // k = T-1: smoothed == filtered; K = 0; Pp = Ps   (kalman_filter.py:1063-1065)
    double xn[NX];
    {
        double Pn[R][NX];
        const MlView vx(a.Xs + (T - 1) * xs_blk, t8, estride);
        FK_UNROLL for (int k = 0; k < NX; ++k) xn[k] = vx.load(k);
        FK_LOAD_ROWS(a.Ps + (T - 1) * ps_blk, Pn);
        const MlView ox(a.xs + (T - 1) * xs_blk, t8, estride);
        FK_UNROLL for (int k = 0; k < NX; ++k) ox.store(k, xn[k]);
        FK_STORE_ROWS(a.Ps_out, T - 1, Pn, parkA);
        FK_STORE_ROWS(a.Pp, T - 1, Pn, parkA);
        {
            double Z[R][NX];
            FK_UNROLL for (int r = 0; r < R; ++r)
                FK_UNROLL for (int c = 0; c < NX; ++c) Z[r][c] = 0.0;
            FK_STORE_ROWS(a.K, T - 1, Z, parkA);
        }
        FK_PUT(mineB, Pn);
    }
    int st = 0;

    _Pragma("nounroll") for (long k = T - 2; k >= 0; --k) {
        double Tm[R][NX];
        {
            double P[R][NX];
            FK_LOAD_ROWS(a.Ps + k * ps_blk, P);
            // T = P F' (own rows), published in park A
            FK_UNROLL for (int i = 0; i < NX; ++i) {
                FK_UNROLL for (int r = 0; r < R; ++r) {
                    double acc = P[r][0] * sF[i * NX];
                    FK_UNROLL for (int q = 1; q < NX; ++q) acc = fma(P[r][q], sF[i * NX + q], acc);
                    Tm[r][i] = acc;
                }
                FK_STAGE();
            }
        }
        FK_PUT(mineA, Tm);
        double d[NX], dinv[NX];
        {
            // Pp = F T + Q (own rows): row q of T from its owner's park
            double Pp[R][NX];
            FK_UNROLL for (int q = 0; q < NX; ++q) {
                double Tq[NX];
                FK_UNROLL for (int j = 0; j < NX; ++j) Tq[j] = FK_FROM(grpA, q, j);
                FK_UNROLL for (int r = 0; r < R; ++r) {
                    const double f = sF[row[r] * NX + q];
                    FK_UNROLL for (int j = 0; j < NX; ++j) Pp[r][j] = (q == 0) ? f * Tq[j] : fma(f, Tq[j], Pp[r][j]);
                }
                FK_STAGE();
            }
            FK_UNROLL for (int r = 0; r < R; ++r)
                FK_UNROLL for (int j = 0; j < NX; ++j) Pp[r][j] += sQ[row[r] * NX + j];
            FK_STORE_ROWS(a.Pp, k, Pp, parkA);                  // (T is dead: park A is free)
            // D = Pn - Pp (own rows) replaces the parked Pn
            FK_UNROLL for (int r = 0; r < R; ++r)
                FK_UNROLL for (int j = 0; j < NX; ++j) mineB[(r * NX + j) * 64] -= Pp[r][j];
            // Pp = L D L', row-distributed: own rows in registers, published in park A column by column
            FK_PUT(mineA, Pp);
            bool pd = true;
            FK_UNROLL for (int j = 0; j < NX; ++j) {
                double Lj[NX];
                FK_UNROLL for (int q = 0; q < NX; ++q)
                    if (q < j) Lj[q] = FK_FROM(grpA, j, q);
                double dj = FK_FROM(grpA, j, j);
                FK_UNROLL for (int q = 0; q < NX; ++q)
                    if (q < j) dj = fma(-Lj[q] * Lj[q], d[q], dj);
                pd = pd && (dj > 0.0);
                d[j] = dj;
                const double di = 1.0 / dj;
                dinv[j] = di;
                if (j + 1 < NX) {
                    FK_UNROLL for (int r = 0; r < R; ++r) {
                        double t = Pp[r][j];
                        FK_UNROLL for (int q = 0; q < NX; ++q)
                            if (q < j) t = fma(-(Pp[r][q] * d[q]), Lj[q], t);
                        Pp[r][j] = row[r] > (unsigned)j ? t * di : Pp[r][j];
                        mineA[(r * NX + j) * 64] = Pp[r][j];
                    }
                    ml_wave_fence();
                }
                FK_STAGE();
            }
            if (!pd) st |= ST_NOT_PD;
        }
        // each lane solves  k Pp = t  for its own rows t of T (they become K's rows)
        FK_UNROLL for (int i = 1; i < NX; ++i) {
            double Li[NX];
            FK_UNROLL for (int q = 0; q < NX; ++q)
                if (q < i) Li[q] = FK_FROM(grpA, i, q);
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
                    const double lqi = FK_FROM(grpA, q, i);
                    FK_UNROLL for (int r = 0; r < R; ++r) t[r] = fma(-lqi, Tm[r][q], t[r]);
                }
            FK_UNROLL for (int r = 0; r < R; ++r) Tm[r][i] = t[r];
            FK_STAGE();
        }
        // E = K D (own rows): rows of D from their owners' park B
        double E[R][NX];
        ml_wave_fence();
        FK_UNROLL for (int q = 0; q < NX; ++q) {
            double Dq[NX];
            FK_UNROLL for (int j = 0; j < NX; ++j) Dq[j] = FK_FROM(grpB, q, j);
            FK_UNROLL for (int r = 0; r < R; ++r) {
                const double kq = Tm[r][q];
                FK_UNROLL for (int j = 0; j < NX; ++j) E[r][j] = (q == 0) ? kq * Dq[j] : fma(kq, Dq[j], E[r][j]);
            }
            FK_STAGE();
        }
        FK_STORE_ROWS(a.K, k, Tm, parkB);                       // (D is dead: park B is free)
        FK_PUT(mineA, Tm);                                      // K's rows for everybody (the factor is dead)
        FK_STAGE();
        {
            // x += K (xn - F x), replicated in the group; G = E K' : column j needs K's row j
            double x[NX], dx[NX], G[R][NX];
            {
                const MlView vx(a.Xs + k * xs_blk, t8, estride);
                FK_UNROLL for (int i = 0; i < NX; ++i) x[i] = vx.load(i);
            }
            FK_UNROLL for (int i = 0; i < NX; ++i) {
                double acc = sF[i * NX] * x[0];
                FK_UNROLL for (int q = 1; q < NX; ++q) acc = fma(sF[i * NX + q], x[q], acc);
                dx[i] = xn[i] - acc;
                FK_STAGE();
            }
            FK_UNROLL for (int j = 0; j < NX; ++j) {
                double Kj[NX];
                FK_UNROLL for (int q = 0; q < NX; ++q) Kj[q] = FK_FROM(grpA, j, q);
                double xa = x[j];
                FK_UNROLL for (int q = 0; q < NX; ++q) xa = fma(Kj[q], dx[q], xa);
                xn[j] = xa;
                FK_UNROLL for (int r = 0; r < R; ++r) {
                    double acc = E[r][0] * Kj[0];
                    FK_UNROLL for (int q = 1; q < NX; ++q) acc = fma(E[r][q], Kj[q], acc);
                    G[r][j] = acc;
                }
                FK_STAGE();
            }
            const MlView ox(a.xs + k * xs_blk, t8, estride);
            FK_UNROLL for (int i = 0; i < NX; ++i) ox.store(i, xn[i]);
            double P[R][NX];
            FK_LOAD_ROWS(a.Ps + k * ps_blk, P);
            FK_UNROLL for (int r = 0; r < R; ++r)
                FK_UNROLL for (int j = 0; j < NX; ++j) P[r][j] += G[r][j];      // P += (K (Pn - Pp)) K'  (kalman_filter.py:1071)
            FK_STORE_ROWS(a.Ps_out, k, P, parkA);               // (K's rows are dead)
            FK_PUT(mineB, P);                                   // the smoothed P of this step: next iteration's Pn
        }
    }
    if (a.status) {
        bool fin = all_finite<NX>(xn);
        FK_UNROLL for (int r = 0; r < R; ++r)
            FK_UNROLL for (int c = 0; c < NX; ++c) fin = fin && (fabs(mineB[(r * NX + c) * 64]) <= 1.79769313486231570815e+308);
        int s = st | (fin ? 0 : ST_NONFINITE);
        // OR over the eight lanes of the group (two quads): within the quad by DPP, across by a row shift
        s |= __builtin_amdgcn_mov_dpp(s, 0xB1, 0xf, 0xf, true);
        s |= __builtin_amdgcn_mov_dpp(s, 0x4E, 0xf, 0xf, true);
        s |= __builtin_amdgcn_mov_dpp(s, 0x104, 0xf, 0xf, true);      // row_shl:4 : lanes 0..3 of the group see lanes 4..7
        if (L == 0) a.status[trk] = s;
    }
#undef FK_FROM
#undef FK_LOAD_ROWS
#undef FK_STORE_ROWS
#undef FK_PUT
}

}  // namespace (instantiation)

// returns 1 when this call is not one the eight-lane smoother serves
int FK_RMLX_CAT(launch_rts_mlx_, FK_NX)(const RtsArgs &a, int layout, bool uniform, hipStream_t s)
{
    using namespace FK_RMLX_CAT(rmlx_, FK_NX);
    if (!uniform || a.model_t || a.n != FK_NX || !a.K || !a.Pp || a.T < 2) return 1;
    const dim3 grid((unsigned)(((a.cnt ? a.cnt : a.N) + BLOCK / LPT - 1) / (BLOCK / LPT))), block(BLOCK);
    if (layout == FK_LAYOUT_AOS) hipLaunchKernelGGL((rts_mlx_kernel<FK_NX, LAYOUT_AOS>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((rts_mlx_kernel<FK_NX, LAYOUT_SOA>), grid, block, 0, s, a);
    return check_launch("rts_mlx_kernel");
}

}  // namespace fk
