// ukf_mlg.hip -- the fused linear-model UKF (UnscentedKalmanFilter.batch_filter, filterpy/kalman/UKF.py:524-632, with
// fx(x, dt) = F x and hx(x) = H x) and its smoother (:634-739) for dim_x = 10..16, dim_z = 1..8 with FOUR -- where code size or
// registers ask for it, EIGHT -- LANES PER TRACK (gfx950).
//
// One lane per track ends at dim_x = 9 (ukf_kernels.hip); above it the step ran as five launches per epoch on resident blocks.
// Here a quad of lanes owns a track for the whole time loop, like kf_mlg.hip, but with the rows of P dealt out CYCLICALLY (lane q
// holds rows q, q + 4, ...; a slot past row n-1 duplicates row n-1): the step is two Cholesky factorisations, and cyclic rows
// keep all four lanes busy down to their last columns.  The arithmetic is ukf_quad_step_v4 (fk_ukf_quad.hpp: ukf_linear_step_v4
// distributed, every sum in v4's order; held against the oracle on the host with the lanes as fibers,
// tests/test_hostcheck_ukf_quad.py); the exchanges are quad-permute DPP moves.
//
// Pair-regrouped sums only (the caller's FK_UKF_FLAG_PAIR_WEIGHTS, verified in the prologue: FK_STATUS_BAD_WEIGHTS
// otherwise); exact dims; optional mask (branch-free: fk_ukf_quad.hpp); both layouts, the per-step outputs through the wave's
// LDS tile as 16-byte units (fk_ml.hpp).  One wave per SIMD.  One object per dim_x (-DFK_NX), dim_z = 1..8 inside.
#include <stdlib.h>
#include <type_traits>

#include "fk_device.hpp"
#include "fk_kernel_args.hpp"
#include "fk_ml.hpp"
#include "fk_chunks.hpp"
#include "fk_ukf_quad.hpp"
#include "../../include/filterhip.h"

#ifndef FK_NX
#error "compile with -DFK_NX=<dim_x>"
#endif

// FK_UMLG_ZDMA=0 (build time): the filter's measurement and mask byte as register loads again (A/B)
#ifndef FK_UMLG_ZDMA
#define FK_UMLG_ZDMA 1
#endif

#define FK_UMLG_CAT_(a, b) a##b
#define FK_UMLG_CAT(a, b) FK_UMLG_CAT_(a, b)

namespace fk {
namespace FK_UMLG_CAT(ukf_mlg_, FK_NX) {

struct QuadDpp {
    template <int O>
    __device__ __forceinline__ double bcast(double v) const { return quad_bcast<O>(v); }
};

// eight lanes per track: DPP cannot leave a quad; ds_swizzle's bit mode can -- within every group of 32 lanes the source
// lane is ((lane & and_mask) | or_mask) ^ xor_mask: and 0x18 keeps the group of eight, or O picks its lane O (the assembler
// prints these as swizzle(BROADCAST,8,O)).  The LDS crossbar, no memory access; two per double like the DPP moves, on the LDS pipe.
struct OctSwizzle {
    template <int O>
    __device__ __forceinline__ double bcast(double v) const
    {
        static_assert(O >= 0 && O < 8, "eight lanes per track");
        constexpr int pat = 0x18 | (O << 5);
        const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v), pat), hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v), pat);
        return __hiloint2double(hi, lo);
    }
};

// The same broadcast on the VALU (round 6, imm_quad.hip's): every quad broadcasts ITS lane O % 4 by quad-permute, then the quad of the pair that
// does not hold lane O takes the other quad's value -- row_shr:4 into the odd quads (bank mask 0xA) or row_shl:4 into the even ones
// (0x5); a group of eight lies inside a DPP row of 16.  Four VALU moves per double instead of two LDS-pipe operations: the eight-lane
// kernels kept the LDS pipe of a CU busy in 80 % of the cycles (four waves x 20 %, profiles/r06/ukf_mlg_counters/) with the VALU at 35 %.
// FK_UMLG_OCT_DPP=0 (build time): ds_swizzle again (A/B; the same bits either way).
#ifndef FK_UMLG_OCT_DPP
#define FK_UMLG_OCT_DPP 1
#endif
struct OctDpp {
    template <int O>
    __device__ __forceinline__ double bcast(double v) const
    {
        static_assert(O >= 0 && O < 8, "eight lanes per track");
        constexpr int q = O & 3, ctrl = (O >> 2) == 0 ? 0x114 : 0x104, banks = (O >> 2) == 0 ? 0xA : 0x5;
        int lo = __double2loint(v), hi = __double2hiint(v);
        lo = __builtin_amdgcn_mov_dpp(lo, q * 0x55, 0xf, 0xf, true);
        hi = __builtin_amdgcn_mov_dpp(hi, q * 0x55, 0xf, 0xf, true);
        lo = __builtin_amdgcn_update_dpp(lo, lo, ctrl, 0xf, banks, false);
        hi = __builtin_amdgcn_update_dpp(hi, hi, ctrl, 0xf, banks, false);
        return __hiloint2double(hi, lo);
    }
};
#if FK_UMLG_OCT_DPP
using OctBcast = OctDpp;
#else
using OctBcast = OctSwizzle;
#endif

template <int NX, int NZ, int LAYOUT, int LN = 4>
__global__ void __launch_bounds__(BLOCK, (NX <= 8 ? 2 : 1))
ukf_mlg_kernel(const UkfArgs a)
{
    constexpr int R = (NX + LN - 1) / LN, KS = 2 * NX + 1, TPW = 64 / LN;      // LN lanes per track, TPW tracks per wave
    static_assert(LN == 4 || LN == 8, "four or eight lanes per track");
    using LM = LdsModel<NX, NZ>;
    constexpr bool AOS = LAYOUT == LAYOUT_AOS;
    constexpr int EP = NX * NX;
    constexpr int TILE = TPW * EP;                               // one output set of a wave's tracks (x reuses its head)
    constexpr int MSZ = LM::SIZE + 2 * KS + 2 + NX;              // [F | Q | H | R | Wm | Wc | pair table]
    __shared__ double smem[MSZ + (BLOCK / 64) * TILE];
    double *tile = smem + MSZ + (threadIdx.x >> 6) * TILE;
    lds_fill<NX, NX>(smem + LM::OFF_F, a.F, NX, NX, 1.0, threadIdx.x);
    lds_fill<NX, NX>(smem + LM::OFF_Q, a.Q, NX, NX, 0.0, threadIdx.x);
    lds_fill<NZ, NX>(smem + LM::OFF_H, a.H, NZ, NX, 0.0, threadIdx.x);
    lds_fill<NZ, NZ>(smem + LM::OFF_R, a.R, NZ, NZ, 1.0, threadIdx.x);
    for (unsigned q = threadIdx.x; q < (unsigned)(2 * KS); q += BLOCK) smem[LM::SIZE + q] = (q < (unsigned)KS ? a.Wm[q] : a.Wc[q - KS]);
    __syncthreads();
    int st = 0;
    if (threadIdx.x == 0) make_pair_table<NX>(smem + LM::SIZE, smem + LM::SIZE + KS, smem + LM::SIZE + 2 * KS);
    if (!pair_weights_symmetric<NX>(smem + LM::SIZE, smem + LM::SIZE + KS)) st |= ST_BAD_WEIGHTS;
    __syncthreads();
    const UkfQuadModel mv{smem + LM::OFF_F, smem + LM::OFF_Q, smem + LM::OFF_H, smem + LM::OFF_R, smem + LM::SIZE + 2 * KS};

    const long N = a.N;
    const unsigned L = threadIdx.x & (unsigned)(LN - 1);
    const long iend = a.i0 + a.cnt;
    long trk = a.i0 + (long)blockIdx.x * (BLOCK / LN) + (threadIdx.x / (unsigned)LN);
    const bool owner = trk < iend;                               // tail quads duplicate the last track; they never write the final state
    if (trk >= iend) trk = iend - 1;
    unsigned row[R];                                             // slot r holds row L + 4 r (clamped: see the header)
    FK_UNROLL for (int r = 0; r < R; ++r) {
        const unsigned g = L + (unsigned)LN * (unsigned)r;
        row[r] = g < (unsigned)NX ? g : (unsigned)NX - 1u;
    }
    unsigned estride = AOS ? 8u : (unsigned)N * 8u;
    asm volatile("" : "+s"(estride));
    const unsigned t8 = (unsigned)trk * (AOS ? (unsigned)NX * 8u : 8u);
    const unsigned tz8 = (unsigned)trk * (AOS ? (unsigned)NZ * 8u : 8u);
    unsigned off_row[R];
    FK_UNROLL for (int r = 0; r < R; ++r)
        off_row[r] = (AOS ? (unsigned)trk * (unsigned)EP * 8u : (unsigned)trk * 8u) + row[r] * (unsigned)NX * estride;
    const long w0 = a.i0 + (long)blockIdx.x * (BLOCK / LN) + (long)wave_index() * TPW;    // scalar: wave_index()
    const unsigned valid = (unsigned)(iend - w0 >= TPW ? TPW : (iend - w0 > 0 ? iend - w0 : 0));
    const unsigned lane = threadIdx.x & 63u, g16 = lane / (unsigned)LN;       // g16: the track's index inside the wave
    const uint8_t *mask_or_dummy = a.mask ? a.mask : reinterpret_cast<const uint8_t *>(a.z);

    double P[R][NX], x[NX];
    {
        const MlView vx(a.x, t8, estride);
        FK_UNROLL for (int r = 0; r < R; ++r) {
            const MlView vP(a.P, off_row[r], estride);
            FK_UNROLL for (int c = 0; c < NX; ++c) P[r][c] = vP.load(c);
        }
        FK_UNROLL for (int k = 0; k < NX; ++k) x[k] = vx.load(k);
        FK_UNROLL for (int r = 0; r < R; ++r)
            FK_UNROLL for (int c = 0; c < NX; ++c) asm volatile("" ::"v"(P[r][c]));       // landed before the loop
        FK_UNROLL for (int k = 0; k < NX; ++k) asm volatile("" ::"v"(x[k]));
    }
    // z[t+1] and its mask byte travel HBM -> LDS by LDS-DMA while step t computes (ZDMA; LaneRecordDma, fk_device.hpp -- the very
    // scheme of kf_mlg.hip: requested at the top of step t, read at the top of step t+1 behind s_waitcnt vmcnt(k), k = the store
    // instructions a step issues (a lower bound, <= 63): vmcnt retires in order, so that wait covers the request and everything
    // older, and none of the step's own stores).  As register loads they are waited for at the end of the loop behind the step's
    // stores -- all but the last few of 35 KB per wave, one wave per SIMD, nothing to cover it; requested in front of the stores
    // and read in the update half (ukf_kernels.hip's way) they cost 116 registers at (12,3) and tipped (14,4) / (16,4) into
    // scratch.  Where the two images per wave do not fit next to the tiles (dim_x 16 with dim_z >= 6): the register loads.
    constexpr int ZIMGD = LaneRecordDma<NZ, LAYOUT>::IMG_DOUBLES + 32;        // + 64 dwords: the mask bytes' dwords
    constexpr bool ZDMA = FK_UMLG_ZDMA && (long)(MSZ + (BLOCK / 64) * TILE) * 8 + 64 + (long)(BLOCK / 64) * 2 * ZIMGD * 8 <= 160 * 1024;
    constexpr int ZST = AOS ? (TPW / 2 * NX + 63) / 64 + (TPW / 2 * EP + 63) / 64 : 2 * ((TPW / 2 * NX + 63) / 64 + (TPW / 2 * EP + 63) / 64);
    constexpr int ZWAIT = ZST < 63 ? ZST : 63;
    __shared__ double s_zd[ZDMA ? (BLOCK / 64) * 2 * ZIMGD : 1];
    LaneRecordDma<NZ, LAYOUT> zdma;
    [[maybe_unused]] auto zreq = [&](long tt, unsigned buf) {
        zdma.request(a.z + tt * N * NZ, (unsigned)N * (unsigned)NZ * 8u, buf);
        // the mask byte of (tt, trk): the aligned dword around it (base and its misalignment are wave-uniform)
        const unsigned long long mb = reinterpret_cast<unsigned long long>(mask_or_dummy) + (unsigned long long)(tt * N);
        const unsigned delta = (unsigned)(mb & 3ull);
        const dma_rsrc_t rm = make_dma_rsrc(reinterpret_cast<const void *>(mb & ~3ull), (unsigned)N + 8u);
        lds_dma4(rm, ((unsigned)trk + delta) & ~3u, 0u, zdma.lds + buf * (unsigned)(ZIMGD * 8) + (unsigned)(LaneRecordDma<NZ, LAYOUT>::IMG_DOUBLES * 8));
    };
    [[maybe_unused]] auto zread = [&](long tt, unsigned buf, double (&zd)[NZ]) -> unsigned {
        zdma.read(buf, zd);
        const unsigned long long mb = reinterpret_cast<unsigned long long>(mask_or_dummy) + (unsigned long long)(tt * N);
        const unsigned sh = (((unsigned)trk + (unsigned)(mb & 3ull)) & 3u) * 8u;
        const unsigned dw = zdma.img0[buf * (unsigned)(ZIMGD * 2) + (unsigned)(LaneRecordDma<NZ, LAYOUT>::IMG_DOUBLES * 2) + (threadIdx.x & 63u)];
        return (dw >> sh) & 0xffu;
    };
    double zn[NZ];
    unsigned hn = 1u;
    if constexpr (ZDMA) {
        zdma.init(s_zd + wave_index() * (2 * ZIMGD), (unsigned)trk, (unsigned)N, threadIdx.x & 63u);
        zdma.stride_doubles = ZIMGD;
        zreq(0, 0u);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        const MlView vz(a.z, tz8, estride);
        FK_UNROLL for (int c = 0; c < NZ; ++c) zn[c] = vz.load(c);
        const unsigned hb = mask_or_dummy[trk];
        hn = a.mask ? hb : 1u;
        FK_UNROLL for (int c = 0; c < NZ; ++c) asm volatile("" ::"v"(zn[c]));
        asm volatile("" ::"v"(hn));
    }
    std::conditional_t<LN == 4, QuadDpp, OctBcast> quad;
    const bool st_m = a.means != nullptr, st_c = a.covs != nullptr;
    _Pragma("nounroll") for (long t = 0; t < a.T; ++t) {
        if constexpr (ZDMA) {
            if (t > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ZWAIT) : "memory");
            const unsigned hb = zread(t, (unsigned)(t & 1), zn);
            hn = a.mask ? hb : 1u;
            zreq(t + 1 < a.T ? t + 1 : t, (unsigned)((t + 1) & 1));
        }
        const bool has_z = hn != 0u;
        double z[NZ];
        FK_UNROLL for (int c = 0; c < NZ; ++c) z[c] = has_z ? zn[c] : 0.0;
        if constexpr (!ZDMA) {
            long tn = t + 1 < a.T ? t + 1 : t;
            asm volatile("" : "+s"(tn));
            const MlView vz(a.z + tn * N * NZ, tz8, estride);
            FK_UNROLL for (int c = 0; c < NZ; ++c) zn[c] = vz.load(c);
            const unsigned hb = mask_or_dummy[tn * N + trk];
            hn = a.mask ? hb : 1u;
        }
        st |= ukf_quad_step_v4<NX, NZ, LN>(x, P, row, z, [&] { return has_z; }, a.scale, mv, quad);
        FK_STAGE();
        // the step's outputs through the wave's tile, 16-byte units (an output that was not asked for: a descriptor of zero
        // tracks -- issued and dropped, no branch in the loop)
        {
            double *md = st_m ? a.means : a.x, *cd = st_c ? a.covs : a.P;
            const unsigned vm = st_m ? valid : 0u, vc = st_c ? valid : 0u;
            if constexpr (AOS) {
                ml_wave_fence();
                FK_UNROLL for (int k = 0; k < NX; ++k) tile[g16 * NX + k] = x[k];          // the quad writes the same value
                ml_wave_fence();
                ml_tile_out_aos<NX, TPW>(md + (t * N + w0) * NX, tile, lane, vm);
                ml_wave_fence();
                FK_UNROLL for (int r = 0; r < R; ++r)
                    FK_UNROLL for (int c = 0; c < NX; ++c) tile[g16 * EP + row[r] * NX + c] = P[r][c];
                ml_wave_fence();
                ml_tile_out_aos<EP, TPW>(cd + (t * N + w0) * EP, tile, lane, vc);
                ml_wave_fence();
            } else {
                ml_wave_fence();
                FK_UNROLL for (int k = 0; k < NX; ++k) tile[k * TPW + g16] = x[k];
                ml_wave_fence();
                ml_tile_out_soa<NX, TPW>(md + t * N * NX, N, w0, tile, lane, vm);
                ml_wave_fence();
                FK_UNROLL for (int r = 0; r < R; ++r)
                    FK_UNROLL for (int c = 0; c < NX; ++c) tile[(row[r] * NX + c) * TPW + g16] = P[r][c];
                ml_wave_fence();
                ml_tile_out_soa<EP, TPW>(cd + t * N * EP, N, w0, tile, lane, vc);
                ml_wave_fence();
            }
        }
    }
    // the final state goes back in place: only a track's own quad writes it, and only once every wave of the workgroup has
    // consumed its initial state (a duplicating tail quad reads the last track's)
    __syncthreads();
    if (owner) {
        const MlView vx(a.x, t8, estride);
        bool fin = all_finite<NX>(x);
        FK_UNROLL for (int k = 0; k < NX; ++k) vx.store(k, x[k]);
        FK_UNROLL for (int r = 0; r < R; ++r) {
            const MlView vP(a.P, off_row[r], estride);
            FK_UNROLL for (int c = 0; c < NX; ++c) {
                vP.store(c, P[r][c]);
                fin = fin && (fabs(P[r][c]) <= 1.79769313486231570815e+308);
            }
        }
        if (a.status) {
            int s = st | (fin ? 0 : ST_NONFINITE);
            s |= __builtin_amdgcn_mov_dpp(s, 0xB1, 0xf, 0xf, true);
            s |= __builtin_amdgcn_mov_dpp(s, 0x4E, 0xf, 0xf, true);
            if constexpr (LN == 8) s |= __builtin_amdgcn_mov_dpp(s, 0x104, 0xf, 0xf, true);  // row_shl:4 : lanes 0..3 of the group see lanes 4..7
            if (L == 0) a.status[trk] = a.status_or ? (a.status[trk] | s) : s;
        }
    }
}

// FK_UMLG_RTS_PARK (build time): 1 = the smoother parks Pxb / Pb in its output tile where the step would spill otherwise (dim_x >= 13),
// 0 = nowhere, 2 = at every dim_x (A/B)
#ifndef FK_UMLG_RTS_PARK
#define FK_UMLG_RTS_PARK 1
#endif

// what ukf_quad_rts_step_v4 asks its caller for (fk_ukf_quad.hpp): the neighbours of the step and, with PARK, a parking lot
template <int NX, int LAYOUT, bool PARKV, int LN = 4>
struct RtsIo {
    static constexpr bool PARK = PARKV;
    static constexpr int R = (NX + LN - 1) / LN, EP = NX * NX, TPW = 64 / LN;     // TPW: tracks per wave
    static constexpr bool AOS = LAYOUT == LAYOUT_AOS;
    const double *Ps_t, *ps_next;            // Ps[t] and ps[t+1]: element 0 of the step's block
    unsigned estride;
    const unsigned (&row)[R];
    const unsigned (&off_row)[R];
    double *tile, *xt, *xpark;
    unsigned g16;                            // the track's index inside the wave
    __device__ __forceinline__ double &tile_at(unsigned e) const { return tile[AOS ? g16 * (unsigned)EP + e : e * (unsigned)TPW + g16]; }
    __device__ __forceinline__ double &lot_at(unsigned e) const { return tile[e * (unsigned)TPW + g16]; }      // the lot: [element][track], whatever the layout
    __device__ __forceinline__ void next_x(double (&out)[NX]) const       // xs[k+1]: still staged in xt
    {
        FK_UNROLL for (int c = 0; c < NX; ++c) out[c] = xt[AOS ? g16 * (unsigned)NX + (unsigned)c : (unsigned)c * (unsigned)TPW + g16];
    }
    __device__ __forceinline__ void next_row(int r, double (&out)[NX]) const
    {
        if constexpr (PARK) {
            // from memory: this wave's own copy-out of the step before (other lanes' stores: complete once vmcnt reaches 0;
            // the loads at agent scope, past the vector L1)
            if (r == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const MlView vP(ps_next, off_row[r], estride);
            FK_UNROLL for (int c = 0; c < NX; ++c) out[c] = vP.template load<16>(c);
        } else {
            FK_UNROLL for (int c = 0; c < NX; ++c) out[c] = tile_at(row[r] * (unsigned)NX + (unsigned)c);
        }
    }
    __device__ __forceinline__ void own_x(double (&out)[NX]) const
    {
        FK_UNROLL for (int c = 0; c < NX; ++c) out[c] = xpark[c * TPW];
    }
    __device__ __forceinline__ void own_row(int r, double (&out)[NX]) const
    {
        const MlView vP(Ps_t, off_row[r], estride);
        FK_UNROLL for (int c = 0; c < NX; ++c) out[c] = vP.load(c);
    }
    // the lot: every lane reads back exactly the addresses it wrote (a duplicated slot: its original's, with the same values)
    __device__ __forceinline__ void park_k(const double (&K)[R][NX]) const
    {
        FK_UNROLL for (int r = 0; r < R; ++r)
            FK_UNROLL for (int c = 0; c < NX; ++c) lot_at(row[r] * (unsigned)NX + (unsigned)c) = K[r][c];
    }
    __device__ __forceinline__ void unpark_k(double (&K)[R][NX]) const
    {
        FK_UNROLL for (int r = 0; r < R; ++r)
            FK_UNROLL for (int c = 0; c < NX; ++c) K[r][c] = lot_at(row[r] * (unsigned)NX + (unsigned)c);
    }
    __device__ __forceinline__ void park_pb(const double (&Pb)[R][NX]) const { park_k(Pb); }
    __device__ __forceinline__ void pb_row(int r, double (&out)[NX]) const
    {
        FK_UNROLL for (int c = 0; c < NX; ++c) out[c] = lot_at(row[r] * (unsigned)NX + (unsigned)c);
    }
};

// The smoother (UnscentedKalmanFilter.rts_smoother, UKF.py:634-739, with fx(x, dt) = F x) on the same four lanes per track:
// ukf_quad_rts_step_v4 per backward step.  What a step needs of its neighbours stays out of the registers:
//   * the smoothed covariance of step k+1 is still in the wave's output tile, where step k+1 staged it for its copy-out (the
//     LAST copy-out of a step, so that nothing overwrites it): the lanes read their rows back from there -- up to dim_x 12.
//     From 13 on (R = 4 row slots) the step would spill 1.5-2.7 KB per lane, every reload a vmcnt(0): there the tile is the
//     step's parking lot instead (RtsIo::PARK) and the covariance of step k+1 is read back from memory, behind the wave's own
//     stores of the step before;
//   * the full rows of Ps[k] are requested a second time for the correction (the factorisation takes the lower part only);
//   * the smoothed mean of step k+1 is replicated in the quad (dim_x registers).
// Reads Xs[k], Ps[k] (the latter 1.6 times); writes xs[k], ps[k], Ks[k]: 8 (2 n + 3 n^2) algorithmic bytes per track-step.
template <int NX, int LAYOUT, int LN = 4>
__global__ void __launch_bounds__(BLOCK, (NX <= 8 ? 2 : 1))
ukf_mlg_rts_kernel(const UkfRtsArgs a, const double *__restrict__ pF, const double *__restrict__ pQ,
                   const double *__restrict__ pWm, const double *__restrict__ pWc)
{
    constexpr int R = (NX + LN - 1) / LN, KS = 2 * NX + 1, TPW = 64 / LN;      // LN lanes per track, TPW tracks per wave
    static_assert(LN == 4 || LN == 8, "four or eight lanes per track");
    constexpr bool AOS = LAYOUT == LAYOUT_AOS;
    constexpr bool PARKV = LN == 4 && (FK_UMLG_RTS_PARK >= 2 || (FK_UMLG_RTS_PARK == 1 && NX >= 13));
    constexpr int EP = NX * NX;
    constexpr int TILE = TPW * EP, XT = 2 * TPW * NX;              // per wave: one covariance-sized output set; xs[k+1] as staged + Xs[k] parked
    constexpr int OFF_F = 0, OFF_Q = EP, OFF_W = 2 * EP;         // [F | Q | Wm | Wc | pair table]
    constexpr int MSZ = OFF_W + 2 * KS + 2 + NX;
    __shared__ double smem[MSZ + (BLOCK / 64) * (TILE + XT)];
    double *tile = smem + MSZ + (threadIdx.x >> 6) * (TILE + XT), *xt = tile + TILE;
    lds_fill<NX, NX>(smem + OFF_F, pF, NX, NX, 1.0, threadIdx.x);
    lds_fill<NX, NX>(smem + OFF_Q, pQ, NX, NX, 0.0, threadIdx.x);
    for (unsigned q = threadIdx.x; q < (unsigned)(2 * KS); q += BLOCK) smem[OFF_W + q] = (q < (unsigned)KS ? pWm[q] : pWc[q - KS]);
    __syncthreads();
    int st = 0;
    if (threadIdx.x == 0) make_pair_table<NX>(smem + OFF_W, smem + OFF_W + KS, smem + OFF_W + 2 * KS);
    if (!pair_weights_symmetric<NX>(smem + OFF_W, smem + OFF_W + KS)) st |= ST_BAD_WEIGHTS;
    __syncthreads();
    const UkfQuadModel mv{smem + OFF_F, smem + OFF_Q, nullptr, nullptr, smem + OFF_W + 2 * KS};

    const long N = a.N;
    const unsigned L = threadIdx.x & (unsigned)(LN - 1);
    const long iend = a.i0 + a.cnt;
    long trk = a.i0 + (long)blockIdx.x * (BLOCK / LN) + (threadIdx.x / (unsigned)LN);
    const bool owner = trk < iend;
    if (trk >= iend) trk = iend - 1;
    unsigned row[R];
    FK_UNROLL for (int r = 0; r < R; ++r) {
        const unsigned g = L + (unsigned)LN * (unsigned)r;
        row[r] = g < (unsigned)NX ? g : (unsigned)NX - 1u;
    }
    unsigned estride = AOS ? 8u : (unsigned)N * 8u;
    asm volatile("" : "+s"(estride));
    const unsigned t8 = (unsigned)trk * (AOS ? (unsigned)NX * 8u : 8u);
    unsigned off_row[R];
    FK_UNROLL for (int r = 0; r < R; ++r)
        off_row[r] = (AOS ? (unsigned)trk * (unsigned)EP * 8u : (unsigned)trk * 8u) + row[r] * (unsigned)NX * estride;
    const long w0 = a.i0 + (long)blockIdx.x * (BLOCK / LN) + (long)wave_index() * TPW;
    const unsigned valid = (unsigned)(iend - w0 >= TPW ? TPW : (iend - w0 > 0 ? iend - w0 : 0));
    const unsigned lane = threadIdx.x & 63u, g16 = lane / (unsigned)LN;       // g16: the track's index inside the wave
    // element e of the wave's track g16 in the staging tile (laid out like the wave's slab of the output array)
    auto tile_at = [&](unsigned e) -> double & { return tile[AOS ? g16 * (unsigned)EP + e : e * (unsigned)TPW + g16]; };
    auto xt_at = [&](unsigned e) -> double & { return xt[AOS ? g16 * (unsigned)NX + e : e * (unsigned)TPW + g16]; };
    double *xpark = xt + TPW * NX + g16;                          // Xs[k] of the step, [element][track]
    // copy-outs of the staged sets (a set that was not asked for, or must not be rewritten: zero tracks -- issued and dropped)
    auto out_cov = [&](double *arr, long t, unsigned vv) {
        double *dst = arr ? arr : a.ps;
        const unsigned v = arr ? vv : 0u;
        ml_wave_fence();
        if constexpr (AOS) ml_tile_out_aos<EP, TPW>(dst + (t * N + w0) * EP, tile, lane, v);
        else ml_tile_out_soa<EP, TPW>(dst + t * N * EP, N, w0, tile, lane, v);
        ml_wave_fence();
    };
    auto out_mean = [&](long t, unsigned vv) {
        ml_wave_fence();
        if constexpr (AOS) ml_tile_out_aos<NX, TPW>(a.xs + (t * N + w0) * NX, xt, lane, vv);
        else ml_tile_out_soa<NX, TPW>(a.xs + t * N * NX, N, w0, xt, lane, vv);
        ml_wave_fence();
    };
    std::conditional_t<LN == 4, QuadDpp, OctBcast> quad;

    // the last step is the filter's own output (xs, ps = Xs.copy(), Ps.copy(); K[T-1] = 0) -- unless this launch continues a
    // chunked call (a.cont): then the window's top step was smoothed by the piece before it and is read back, not rewritten
    {
        double xn[NX];
        const double *srcx = a.cont ? a.xs : a.Xs, *srcP = a.cont ? a.ps : a.Ps;
        const unsigned vtop = a.cont ? 0u : valid;
        const MlView vx(srcx + (a.T - 1) * N * NX, t8, estride);
        double Pt[R][NX];
        FK_UNROLL for (int r = 0; r < R; ++r) {
            const MlView vP(srcP + (a.T - 1) * N * EP, off_row[r], estride);
            FK_UNROLL for (int c = 0; c < NX; ++c) Pt[r][c] = vP.load(c);
        }
        FK_UNROLL for (int k = 0; k < NX; ++k) xn[k] = vx.load(k);
        ml_wave_fence();
        FK_UNROLL for (int r = 0; r < R; ++r)
            FK_UNROLL for (int c = 0; c < NX; ++c) tile_at(row[r] * (unsigned)NX + (unsigned)c) = 0.0;
        out_cov(a.Ks, a.T - 1, vtop);
        FK_UNROLL for (int k = 0; k < NX; ++k) xt_at((unsigned)k) = xn[k];
        out_mean(a.T - 1, vtop);
        FK_UNROLL for (int r = 0; r < R; ++r)
            FK_UNROLL for (int c = 0; c < NX; ++c) tile_at(row[r] * (unsigned)NX + (unsigned)c) = Pt[r][c];
        out_cov(a.ps, a.T - 1, vtop);                          // ... and stays in the tile for step T-2
    }
    _Pragma("nounroll") for (long t = a.T - 2; t >= 0; --t) {
        double x[NX], P[R][NX], K[R][NX];
        {
            const MlView vx(a.Xs + t * N * NX, t8, estride);
            FK_UNROLL for (int r = 0; r < R; ++r) {
                const MlView vP(a.Ps + t * N * EP, off_row[r], estride);
                FK_UNROLL for (int c = 0; c < NX; ++c)
                    if (c <= LN * r + LN - 1) P[r][c] = vP.load(c);
            }
            FK_UNROLL for (int k = 0; k < NX; ++k) x[k] = vx.load(k);
            FK_UNROLL for (int k = 0; k < NX; ++k) xpark[k * TPW] = x[k];
        }
        RtsIo<NX, LAYOUT, PARKV, LN> io{a.Ps + t * N * EP, a.ps + (t + 1) * N * EP, estride, row, off_row, tile, xt, xpark, g16};
        st |= ukf_quad_rts_step_v4<NX, LN>(x, P, row, a.scale, mv, quad, K, io);
        FK_STAGE();
        ml_wave_fence();
        FK_UNROLL for (int k = 0; k < NX; ++k) xt_at((unsigned)k) = x[k];
        out_mean(t, valid);
        FK_UNROLL for (int r = 0; r < R; ++r)
            FK_UNROLL for (int c = 0; c < NX; ++c) tile_at(row[r] * (unsigned)NX + (unsigned)c) = K[r][c];
        out_cov(a.Ks, t, valid);
        FK_UNROLL for (int r = 0; r < R; ++r)
            FK_UNROLL for (int c = 0; c < NX; ++c) tile_at(row[r] * (unsigned)NX + (unsigned)c) = P[r][c];
        out_cov(a.ps, t, valid);                               // last: the tile keeps ps[t] for step t-1
    }
    if (owner && a.status) {
        // the last smoothed state is what the wave's staging areas hold: finite?
        bool fin = true;
        FK_UNROLL for (int k = 0; k < NX; ++k) fin = fin && (fabs(xt_at((unsigned)k)) <= 1.79769313486231570815e+308);
        FK_UNROLL for (int r = 0; r < R; ++r)
            FK_UNROLL for (int c = 0; c < NX; ++c) fin = fin && (fabs(tile_at(row[r] * (unsigned)NX + (unsigned)c)) <= 1.79769313486231570815e+308);
        int s = st | (fin ? 0 : ST_NONFINITE);
        s |= __builtin_amdgcn_mov_dpp(s, 0xB1, 0xf, 0xf, true);
        s |= __builtin_amdgcn_mov_dpp(s, 0x4E, 0xf, 0xf, true);
        if constexpr (LN == 8) s |= __builtin_amdgcn_mov_dpp(s, 0x104, 0xf, 0xf, true);      // row_shl:4 : lanes 0..3 of the group see lanes 4..7
        if (L == 0) a.status[trk] = a.status_or ? (a.status[trk] | s) : s;
    }
}

}  // namespace (instantiation)

// returns 1 when this call is not one the four-lane kernel serves
int FK_UMLG_CAT(launch_ukf_mlg_, FK_NX)(const UkfArgs &a, int layout, hipStream_t s)
{
    using namespace FK_UMLG_CAT(ukf_mlg_, FK_NX);
    if (a.n != FK_NX || a.m < 1 || a.m > 8) return 1;
    // lanes per track: four; eight where four would spill kilobytes (dim_z >= 5 from dim_x 13: S, its factor and two rows of H L
    // per lane on top of the rows of P- and L).  FK_UKF_MLG_LANES=4 | 8 forces one (8: dim_x >= 9, dim_z as compiled below).
    static const int forced = [] { const char *v = getenv("FK_UKF_MLG_LANES"); return v ? atoi(v) : 0; }();
    [[maybe_unused]] const bool oct = FK_NX >= 9 && (forced == 8 || (forced != 4 && FK_NX >= 13 && a.m >= 5));
    const dim3 grid((unsigned)((a.cnt + BLOCK / 4 - 1) / (BLOCK / 4))), grid8((unsigned)((a.cnt + BLOCK / 8 - 1) / (BLOCK / 8))), block(BLOCK);
// (FK_UMLG_ONLY_NZ=<dim_z>: a one-off build with that filter instantiation alone, for looking at its code)
#ifndef FK_UMLG_ONLY_NZ
#define FK_UMLG_ONLY_NZ 0
#endif
#define GO(NZV)                                                                                                         \
    if constexpr (FK_UMLG_ONLY_NZ == 0 || FK_UMLG_ONLY_NZ == NZV) if (a.m == NZV) {                                      \
        if constexpr (FK_NX >= 9) if (oct) {                                                                            \
            if (layout == FK_LAYOUT_AOS) hipLaunchKernelGGL((ukf_mlg_kernel<FK_NX, NZV, LAYOUT_AOS, (FK_NX >= 9 ? 8 : 4)>), grid8, block, 0, s, a); \
            else hipLaunchKernelGGL((ukf_mlg_kernel<FK_NX, NZV, LAYOUT_SOA, (FK_NX >= 9 ? 8 : 4)>), grid8, block, 0, s, a); \
            return check_launch("ukf_mlg_kernel<8 lanes>");                                                             \
        }                                                                                                               \
        if (layout == FK_LAYOUT_AOS) hipLaunchKernelGGL((ukf_mlg_kernel<FK_NX, NZV, LAYOUT_AOS>), grid, block, 0, s, a); \
        else hipLaunchKernelGGL((ukf_mlg_kernel<FK_NX, NZV, LAYOUT_SOA>), grid, block, 0, s, a);                         \
    }
    GO(1) GO(2) GO(3) GO(4) GO(5) GO(6) GO(7) GO(8)
#undef GO
    return check_launch("ukf_mlg_kernel");
}

// the smoother's launch (fk_ukf_linear_rts_f64 at dim_x 10..16); returns 1 when this file does not serve the call
int FK_UMLG_CAT(launch_ukf_mlg_rts_, FK_NX)(const UkfRtsArgs &a, const double *F, const double *Q, const double *Wm, const double *Wc,
                                            int layout, hipStream_t s)
{
    using namespace FK_UMLG_CAT(ukf_mlg_, FK_NX);
    if (a.n != FK_NX) return 1;
    // lanes per track: four; eight from dim_x 13, where the four-lane step is 65-105 KB of code for a 64 KB instruction cache
    // (profiles/r04/lease_q: 85 us per wave-step at dim_x 16).  FK_UKF_MLG_RTS_LANES=4 | 8 forces one (8: dim_x >= 9 only).
    static const int forced = [] { const char *v = getenv("FK_UKF_MLG_RTS_LANES"); return v ? atoi(v) : 0; }();
    [[maybe_unused]] const bool oct = FK_NX >= 9 && (forced == 8 || (forced != 4 && FK_NX >= 13));
#if FK_NX >= 9
    if (oct) {
        const dim3 grid((unsigned)((a.cnt + BLOCK / 8 - 1) / (BLOCK / 8))), block(BLOCK);
        if (layout == FK_LAYOUT_AOS) hipLaunchKernelGGL((ukf_mlg_rts_kernel<FK_NX, LAYOUT_AOS, 8>), grid, block, 0, s, a, F, Q, Wm, Wc);
        else hipLaunchKernelGGL((ukf_mlg_rts_kernel<FK_NX, LAYOUT_SOA, 8>), grid, block, 0, s, a, F, Q, Wm, Wc);
        return check_launch("ukf_mlg_rts_kernel<8 lanes>");
    }
#endif
    const dim3 grid((unsigned)((a.cnt + BLOCK / 4 - 1) / (BLOCK / 4))), block(BLOCK);
    if (layout == FK_LAYOUT_AOS) hipLaunchKernelGGL((ukf_mlg_rts_kernel<FK_NX, LAYOUT_AOS>), grid, block, 0, s, a, F, Q, Wm, Wc);
    else hipLaunchKernelGGL((ukf_mlg_rts_kernel<FK_NX, LAYOUT_SOA>), grid, block, 0, s, a, F, Q, Wm, Wc);
    return check_launch("ukf_mlg_rts_kernel");
}

}  // namespace fk
