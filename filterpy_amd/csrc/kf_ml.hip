// kf_ml.hip -- batch_filter and rts_smoother for dim_x = 9 with THREE LANES PER TRACK (gfx950).
//
// One lane per track stops scaling at dim_x = 9: P alone is 81 doubles = 162 VGPRs, the step's
// temporaries push the kernel to one wave per SIMD with AGPR spills, and BASELINE config 3's 1e5
// tracks are only 1563 waves for 1024 SIMDs.  Here a QUAD of lanes owns a track: lanes 0..2 hold
// three rows of P each (27 doubles), lane 3 mirrors lane 2 (same loads, same stores: no lane is
// ever predicated).  Every matrix product is arranged so that each output element is produced
// whole by the lane that owns its row, in the reference's k = 0..n-1 order; the rows a lane does
// not own arrive by quad-permute DPP moves (v_mov_b32 quad_perm:[s,s,s,s], two per double, no LDS):
//
//   predict  T = P F' (rows local) ; P' = a2 (F T) + Q : row k of T is broadcast, lane accumulates
//            F[i][k] T[k][:] into its rows i       (kalman_filter.py:472-478, associated as F (P F'))
//   update   PHT = P H' rows local, broadcast -> S, its L D L' and y replicated in every lane ;
//            K rows local ; H P by a three-way sum across the quad ; T1 = P - K (H P) ;
//            D = K R - T1 H' ; P+ = T1 + D K' with K's rows broadcast
//            (Joseph form, kalman_filter.py:533-556, with the I - K H factors applied implicitly like
//            fk_math_sym.hpp does, but without assuming P symmetric)
//   x (9 doubles) is replicated in all lanes.
//
// Per lane and step: ~1250 FMAs and ~190 exchanged doubles, ~90 live doubles (one lane per track:
// ~3700 FMAs, > 250 live doubles).  A wave carries 16 tracks, so config 3 becomes 6250 waves.
// Layouts: SOA (element-major: 16-byte pair stores, see MlView::store_pair) and AOS (NumPy order: output
// sets staged through a wave-private LDS tile, ml_store_aos); exact dims (9, 3), optional mask (no
// branch: see MASK below), all four outputs or none.  The base instantiations serve the plain call (one
// constant model, predict -> update, no control input); the VAR instantiations (FK_ML_PART=2 of the build)
// add what KalmanFilter.batch_filter's other arguments ask for (kalman_filter.py:941-991): one model PER STEP
// shared by the bank (Fs / Qs / Hs / Rs lists: double-buffered in LDS, fetched a step ahead), a control input
// x = F x + B u (u[t] travels with z[t]) and update_first (UF: update -> predict inside a step).  Per-track
// models stay on kf_fast / kf_kernel.  The smoother (rts_ml_kernel) is at the end of the file.
#include <stdio.h>
#include <stdlib.h>

#include "fk_device.hpp"
#include "fk_math_sym.hpp"
#include "fk_kernel_args.hpp"
#include "fk_ml.hpp"
#include "fk_chunks.hpp"
#include "../../include/filterhip.h"

#ifndef FK_ML_WAVES
#define FK_ML_WAVES 2
#endif
// the build compiles this file twice (parallel): FK_ML_PART=1 the plain-call kernels + the smoother, FK_ML_PART=2
// the VAR instantiations; 0 = everything in one unit
#ifndef FK_ML_PART
#define FK_ML_PART 0
#endif

namespace fk {

// a lane's R x NX rows (R*NX consecutive elements) as pairs + one odd element
// MODE: 0 = 8-byte accesses; 1 = 16-byte pairs over the quad exchange (SOA, even N); 2 = 16-byte
// contiguous element pairs (AOS)
template <int R, int NX, int MODE>
__device__ __forceinline__ void store_rows(const MlView &v, const double (&M)[R][NX])
{
    if constexpr (MODE != 0) {
        FK_UNROLL for (int f = 0; f + 1 < R * NX; f += 2) {
            if constexpr (MODE == 1) v.store_pair(f, M[f / NX][f % NX], M[(f + 1) / NX][(f + 1) % NX]);
            else v.store2(f, M[f / NX][f % NX], M[(f + 1) / NX][(f + 1) % NX]);
        }
        if ((R * NX) % 2) v.store(R * NX - 1, M[R - 1][NX - 1]);
    } else {
        FK_UNROLL for (int r = 0; r < R; ++r)
            FK_UNROLL for (int c = 0; c < NX; ++c) v.store(r * NX + c, M[r][c]);
    }
}

template <int R, int NX, int MODE>
__device__ __forceinline__ void load_rows(const MlView &v, double (&M)[R][NX])
{
    if constexpr (MODE != 0) {
        FK_UNROLL for (int f = 0; f + 1 < R * NX; f += 2) {
            if constexpr (MODE == 1) v.load_pair(f, M[f / NX][f % NX], M[(f + 1) / NX][(f + 1) % NX]);
            else v.load2(f, M[f / NX][f % NX], M[(f + 1) / NX][(f + 1) % NX]);
        }
        if ((R * NX) % 2) M[R - 1][NX - 1] = v.load(R * NX - 1);
    } else {
        FK_UNROLL for (int r = 0; r < R; ++r)
            FK_UNROLL for (int c = 0; c < NX; ++c) M[r][c] = v.load(r * NX + c);
    }
}

template <int NX, int MODE>
__device__ __forceinline__ void load_x(const MlView &v, double (&x)[NX])
{
    if constexpr (MODE != 0) {
        FK_UNROLL for (int k = 0; k + 1 < NX; k += 2) {
            if constexpr (MODE == 1) v.load_pair(k, x[k], x[k + 1]);
            else v.load2(k, x[k], x[k + 1]);
        }
        if (NX % 2) x[NX - 1] = v.load(NX - 1);
    } else {
        FK_UNROLL for (int k = 0; k < NX; ++k) x[k] = v.load(k);
    }
}

template <int NX, int MODE>
__device__ __forceinline__ void store_x(const MlView &v, const double (&x)[NX])
{
    if constexpr (MODE != 0) {
        FK_UNROLL for (int k = 0; k + 1 < NX; k += 2) {
            if constexpr (MODE == 1) v.store_pair(k, x[k], x[k + 1]);
            else v.store2(k, x[k], x[k + 1]);
        }
        if (NX % 2) v.store(NX - 1, x[NX - 1]);
    } else {
        FK_UNROLL for (int k = 0; k < NX; ++k) v.store(k, x[k]);
    }
}

// AOS ([track][element], NumPy order) output of one (x, P) set: a wave's 16 tracks are one contiguous
// slab of 16 * E doubles; the quads write their rows into a wave-private LDS tile laid out like the
// slab, then the 64 lanes copy consecutive 16-byte units (1 KiB per store instruction).  The buffer
// descriptor is sized to the wave's valid tracks: the hardware range check drops the tail.
template <int R, int NX>
__device__ __forceinline__ void ml_store_aos(const double (&x)[NX], const double (&P)[R][NX], double *xdst, double *Pdst,
                                             double *tile, unsigned lane, unsigned Lc, unsigned valid)
{
    constexpr int EP = NX * NX, UP = 16 * EP / 2, UX = 16 * NX / 2;      // 16-byte units per wave
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
    const unsigned q = lane >> 2;
    double *tx = tile, *tP = tile + 16 * NX;
    ml_wave_fence();
    FK_UNROLL for (int k = 0; k < NX; ++k) tx[q * NX + k] = x[k];                         // the quad writes the same value
    FK_UNROLL for (int r = 0; r < R; ++r)
        FK_UNROLL for (int c = 0; c < NX; ++c) tP[q * EP + Lc * (R * NX) + r * NX + c] = P[r][c];
    ml_wave_fence();
    const rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(xdst, 0, (int)(valid * (unsigned)NX * 8u), 0x00020000);
    const rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc(Pdst, 0, (int)(valid * (unsigned)EP * 8u), 0x00020000);
    // units are read from the tile in batches AHEAD of their stores (ml_copy_units: read -> wait -> store per unit exposed the
    // LDS latency thirteen times per output set); no lane is predicated: a lane past the slab reads a clamped unit and its
    // store falls outside the descriptor (sized to the wave's valid tracks), which drops it
    ml_copy_units<UX, 6>(lane, [&](unsigned unit) { return tx + 2u * unit; },
                         [&](unsigned unit, bool ok, const u32x4 &v) {
                             __builtin_amdgcn_raw_buffer_store_b128(v, rx, ok ? unit * 16u : ML_OFF_DROP, 0, 0);
                         });
    ml_copy_units<UP, 6>(lane, [&](unsigned unit) { return tP + 2u * unit; },
                         [&](unsigned unit, bool ok, const u32x4 &v) {
                             __builtin_amdgcn_raw_buffer_store_b128(v, rP, ok ? unit * 16u : ML_OFF_DROP, 0, 0);
                         });
}

// element-major twin: the tile is [element][16 tracks] (x's NX planes, then P's NX*NX), copied out by ml_slab_out_soa;
// xfirst / Pfirst: element 0 of the wave's first track at this time step
template <int R, int NX>
__device__ __forceinline__ void ml_store_soa_slab(const double (&x)[NX], const double (&P)[R][NX], double *xfirst, double *Pfirst,
                                                  unsigned n8, double *tile, unsigned lane, unsigned Lc, unsigned valid)
{
    const unsigned q = lane >> 2;
    double *tx = tile, *tP = tile + 16 * NX;
    ml_wave_fence();
    FK_UNROLL for (int k = 0; k < NX; ++k) tx[k * 16 + q] = x[k];                          // the quad writes the same value
    FK_UNROLL for (int r = 0; r < R; ++r)
        FK_UNROLL for (int c = 0; c < NX; ++c) tP[(Lc * (R * NX) + r * NX + c) * 16 + q] = P[r][c];
    ml_wave_fence();
    ml_slab_out_soa<NX, 16, 6>(tx, xfirst, n8, lane, valid);
    ml_slab_out_soa<NX * NX, 16, 6>(tP, Pfirst, n8, lane, valid);
}

// VAR: per-step shared models (a.model_t), control input (a.nu > 0), mask by pointer -- all wave-uniform run-time
// switches of ONE extra instantiation family; UF (update_first) reorders the step and is compile-time.
// SLAB (element-major outputs, round 4): an output set leaves like the NumPy-order one -- staged in a wave-private LDS tile,
// laid out [element][16 tracks], and copied out as 16-byte units of two adjacent tracks (ml_slab_out_soa: 8 elements x 128
// contiguous bytes per store instruction) at the two points of the step where a set is complete -- instead of the PAIRS
// scheme's DPP half-exchange per pair (136 v_mov_dpp + the store-data hazard nops per step, in a kernel bound by VALU
// issue).  Any N (an odd tail's last track leaves as 8 bytes).  FK_ML_SLAB=0 selects the PAIRS / 8-byte instantiations.
// PERS (round 4): a persistent grid instead of one workgroup per 64 tracks for all T steps.  Every wave runs the same T steps, so
// W waves on S wave slots cost ceil(W / S) rounds: configs[2]'s 6250 waves on 2048 slots pay for a fourth round that is 5 %
// full (measured: 2 / 3 / 4 whole rounds 2.18 / 3.04 / 4.03 ms, N = 1e5 3.40 ms).  The multi-stream tail filling of
// fk_chunks.hpp returns 1-4 % of that -- every piece is a kernel with its own launch gap and its own tail.  Here the call is
// cut into G = ceil(N / 64) track groups x H time chunks and 512 resident workgroups draw TICKETS, chunk-major (all groups'
// chunk h before any chunk h + 1): a workgroup that finishes early simply takes the next ticket.  Chunk h of a group needs
// chunk h - 1 of the same group: its ticket is at least G >= 512 draws older, i.e. held by a workgroup that is running or
// done -- no deadlock whatever is resident; the state travels through an element-major hand-over block of the call's scratch
// allocation, written and read with agent-scope (sc1) accesses -- coherent across the XCDs' L2s without write-backs; NumPy-order
// x / P in place would make them 8-byte accesses to 64 different lines per instruction -- and announced by a completion word.  Same arithmetic per track: results are bit-identical to the single launch.
template <int R, int NZ, bool OUTS, int WAVES, bool PAIRS, bool MASK, int LAYOUT, bool VAR = false, bool UF = false, bool SLAB = false, bool PERS = false>
__global__ void __launch_bounds__(BLOCK, WAVES)
kf_ml_kernel(const KfArgs a_in)
{
    static_assert(!PERS || (OUTS && !VAR && !PAIRS && (LAYOUT == LAYOUT_AOS || SLAB)), "PERS: the plain call with outputs that leave in their own step");
    constexpr int HAUX = PERS ? 16 : 0;      // cache policy of the state hand-over: sc1 = agent scope (MlView::load / store)
    KfArgs a = a_in;
    constexpr int NX = 3 * R;
    using LM = LdsModel<NX, NZ>;
    constexpr bool AOS = LAYOUT == LAYOUT_AOS;
    constexpr int TILE = 16 * NX + 16 * NX * NX;                 // AOS: one (x, P) output set of a wave
    constexpr int NUC = 4;                                       // padded dim_u (VAR)
    constexpr int MLEN = LM::SIZE + NX * NUC;                    // VAR: [F | Q | H | R | B padded to NX x NUC]
    constexpr int MSTR = MLEN + 2;                               // ... + the pad slot idle threads publish into
    constexpr int MSZ = VAR ? 2 * MSTR : LM::SIZE;               // VAR: two model buffers (this step's, the next one's)
    static_assert(!VAR || MLEN <= BLOCK, "one thread per model element");
    static_assert(!UF || VAR, "update_first is a VAR instantiation");
    static_assert(!SLAB || (!AOS && !PAIRS && !VAR && OUTS), "SLAB: the plain element-major call with outputs");
    __shared__ double smem[MSZ + ((AOS || SLAB) && OUTS ? (BLOCK / 64) * TILE : 0)];
    double *tile = smem + MSZ + (threadIdx.x >> 6) * TILE;
    const double *sB = smem + LM::SIZE;
    lds_fill<NX, NX>(smem + LM::OFF_F, a.F, NX, NX, 1.0, threadIdx.x);
    lds_fill<NX, NX>(smem + LM::OFF_Q, a.Q, NX, NX, 0.0, threadIdx.x);
    lds_fill<NZ, NX>(smem + LM::OFF_H, a.H, NZ, NX, 0.0, threadIdx.x);
    lds_fill<NZ, NZ>(smem + LM::OFF_R, a.R, NZ, NZ, 1.0, threadIdx.x);
    if constexpr (VAR) lds_fill<NX, NUC>(smem + LM::SIZE, a.nu > 0 ? a.B : nullptr, NX, a.nu, 0.0, threadIdx.x);
    // element `tid` of the concatenated model [F | Q | H | R | B] of step tt (VAR; B [tt][NX][nu] goes with the model
    // mode like the others; no control input: a valid dummy address, the value selected to 0)
    const double *B_or_dummy = VAR && a.nu > 0 ? a.B : a.F;
    auto model_elem = [&](long tt) -> double {
        const int k = (int)threadIdx.x;
        if (k < LM::OFF_Q) return a.F[tt * (NX * NX) + k];
        if (k < LM::OFF_H) return a.Q[tt * (NX * NX) + (k - LM::OFF_Q)];
        if (k < LM::OFF_R) return a.H[tt * (NZ * NX) + (k - LM::OFF_H)];
        if (k < LM::SIZE) return a.R[tt * (NZ * NZ) + (k - LM::OFF_R)];
        const int i = (k - LM::SIZE) / NUC, j = (k - LM::SIZE) % NUC;
        const bool live = k < MLEN && j < a.nu;
        const double v = B_or_dummy[live ? tt * (long)(NX * a.nu) + i * a.nu + j : 0];
        return live ? v : 0.0;
    };
    // The run-time switches of the VAR family are BRANCH-FREE (a branch inside the time loop splits its one basic
    // block and the register allocation falls apart: +100 spilled doubles per lane measured): without per-step
    // models the hand-over republishes model[0] every step, without a control input B is zero-filled and u is
    // selected to 0, without a mask the byte is read from a valid dummy address and selected to 1.
    double mnext = 0.0;                        // this thread's element of the NEXT step's model
    if constexpr (VAR) {
        mnext = model_elem(a.model_t && a.T > 1 ? 1 : 0);
        asm volatile("" ::"v"(mnext));
    }
    __syncthreads();
    const double *sF = smem + LM::OFF_F, *sQ = smem + LM::OFF_Q, *sH = smem + LM::OFF_H, *sR = smem + LM::OFF_R;

    [[maybe_unused]] int pers_g = 0, pers_h = 0;
    __shared__ int s_task;
    for (;;) {                                                   // PERS: one trip per ticket; otherwise exactly one trip
    unsigned bid = blockIdx.x;
    if constexpr (PERS) {
        __syncthreads();                                         // the previous ticket's LDS traffic is over, s_task is free
        if (threadIdx.x == 0) s_task = atomicAdd(a_in.pers_ctl, 1);
        __syncthreads();
        const int task = __builtin_amdgcn_readfirstlane(s_task);
        const int G = a_in.pers_G, H = a_in.pers_H;
        if (task >= G * H) break;
        pers_h = task / G;
        pers_g = task - pers_h * G;
        bid = (unsigned)pers_g;
        const long t0 = a_in.T * pers_h / H, t1 = a_in.T * (pers_h + 1) / H, NN = a_in.N;
        a = a_in;
        a.T = t1 - t0;
        a.z = a_in.z + t0 * NN * NZ;
        a.mask = a_in.mask ? a_in.mask + t0 * NN : nullptr;
        a.means = a_in.means + t0 * NN * (3 * R);
        a.means_p = a_in.means_p + t0 * NN * (3 * R);
        a.covs = a_in.covs + t0 * NN * (9 * R * R);
        a.covs_p = a_in.covs_p + t0 * NN * (9 * R * R);
        a.status_or = t0 > 0 ? 1 : a_in.status_or;
        if (pers_h > 0) {
            // (the state is then read with agent-scope loads: no acquire fence, i.e. no L2 invalidation, is needed)
            if (threadIdx.x == 0) {
                while (__hip_atomic_load(&a_in.pers_ctl[1 + pers_g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < pers_h) __builtin_amdgcn_s_sleep(8);
            }
            __syncthreads();
        }
    }

    const long N = a.N;
    const unsigned L = threadIdx.x & 3u;
    const unsigned Lc = L < 3u ? L : 2u;                       // lane 3 mirrors lane 2
    // this launch handles tracks [i0, iend) of the bank (N stays the array stride): the whole bank, or one track group
    // of a chunked call (launch_kf_ml_9_3: groups start on multiples of 64 tracks)
    const long iend = a.i0 + a.cnt;
    long trk = a.i0 + (long)bid * (BLOCK / 4) + (threadIdx.x >> 2);
    const unsigned odd = (threadIdx.x >> 2) & 1u;              // odd quad of its pair (workgroups start on even tracks)
    const bool owner = trk < iend;                             // tail quads only duplicate: they never write the final state
    if (trk >= iend) trk = PAIRS ? iend - 2 + odd : iend - 1;  // tail quads recompute the last track (pair)
    // element e of this lane's track sits at  lane offset + e * estride:  SOA: track*8 + e*N*8 ;
    // AOS: track*E*8 + e*8 (E = elements per record of the array: the offsets below are per array)
    unsigned estride = AOS ? 8u : (unsigned)N * 8u;
    asm volatile("" : "+s"(estride));
    const unsigned t8 = (unsigned)trk * (AOS ? (unsigned)NX * 8u : 8u);                       // x-like arrays
    const unsigned tz8 = (unsigned)trk * (AOS ? (unsigned)NZ * 8u : 8u);                      // z
    const unsigned off_rows = (AOS ? (unsigned)trk * (unsigned)(NX * NX) * 8u : (unsigned)trk * 8u)
                              + Lc * (unsigned)(R * NX) * estride;                            // element (Lc*R + r)*NX + c
    const unsigned pair_rows = odd ? off_rows - 8u + estride : off_rows;
    const unsigned pair_x = odd ? t8 - 8u + estride : t8;
    // AOS output slabs: first track of this wave and how many of its 16 tracks exist
    const long w0 = a.i0 + (long)bid * (BLOCK / 4) + (long)wave_index() * 16;               // scalar: see wave_index()
    const unsigned valid = (unsigned)(iend - w0 >= 16 ? 16 : (iend - w0 > 0 ? iend - w0 : 0));
    const unsigned lane = threadIdx.x & 63u;
    const double *myF = sF + Lc * (R * NX);                     // this lane's rows of F and Q
    const double *myQ = sQ + Lc * (R * NX);
    const unsigned nu = VAR ? (unsigned)a.nu : 0u;
    const unsigned nu_idx = nu ? nu - 1u : 0u;
    const unsigned tu8 = (unsigned)trk * (AOS ? nu * 8u : 8u);                                // u
    // (dummies: valid addresses whose contents are never used -- z is at least as large as a mask / one u element per track)
    const uint8_t *mask_or_dummy = VAR && a.mask ? a.mask : reinterpret_cast<const uint8_t *>(a.z);
    const double *u_or_dummy = VAR && nu ? a.u : a.z;

    // H (27 doubles, replicated) lives in VGPRs instead of being re-read from LDS at each of its five uses
    // per step (measured: 2 % faster than LDS reads, no occupancy change)
    // (not in the AOS instantiation: its LDS staging needs the registers, H is read from LDS there)
    // (nor in the VAR instantiations: H may change every step)
    constexpr bool HLDS = AOS || VAR || !OUTS || SLAB;     // (the final-state-only instantiation: H from LDS too -- its registers go to the prefetched operands)
    double Hreg[HLDS ? 1 : NZ * NX];
    if constexpr (!HLDS) {
        FK_UNROLL for (int e = 0; e < NZ * NX; ++e) Hreg[e] = sH[e];
    }
#define HX(e) (HLDS ? sH[(e)] : Hreg[HLDS ? 0 : (e)])
    double P[R][NX], x[NX];
    {
        if (PERS && pers_h > 0) {
            // a later chunk of the group: the state the previous chunk left in the hand-over block (element-major whatever the
            // layout: 512 contiguous bytes per wave instruction; agent-scope loads)
            const unsigned n8 = (unsigned)N * 8u;
            const MlView wx(a_in.pers_ws, (unsigned)trk * 8u, n8), wP(a_in.pers_ws + (long)NX * N, (unsigned)trk * 8u + Lc * (unsigned)(R * NX) * n8, n8);
            FK_UNROLL for (int r = 0; r < R; ++r)
                FK_UNROLL for (int c = 0; c < NX; ++c) P[r][c] = wP.template load<HAUX>(r * NX + c);
            FK_UNROLL for (int k = 0; k < NX; ++k) x[k] = wx.template load<HAUX>(k);
        } else {
            const MlView vP(a.P, off_rows, estride), vx(a.x, t8, estride);
            FK_UNROLL for (int r = 0; r < R; ++r)
                FK_UNROLL for (int c = 0; c < NX; ++c) P[r][c] = vP.load(r * NX + c);
            FK_UNROLL for (int k = 0; k < NX; ++k) x[k] = vx.load(k);
        }
        // land the prologue loads here: left pending, the loop header's s_waitcnt (which must cover this
        // path too) makes every iteration wait for the previous step's stores
        FK_UNROLL for (int r = 0; r < R; ++r)
            FK_UNROLL for (int c = 0; c < NX; ++c) asm volatile("" ::"v"(P[r][c]));
        FK_UNROLL for (int k = 0; k < NX; ++k) asm volatile("" ::"v"(x[k]));
    }
    int st = 0;
    // z is fetched one step ahead: vmcnt retires in order, so waiting for a load also waits for every
    // store issued before it; with a whole step between a load and its use, the wait is behind stores
    // that had a full step to drain (the counter saturates at 63 anyway).
    // MASK: a missing measurement (mask byte 0; kalman_filter.py:515-520) leaves x, P at the prior.  No
    // branch: the step runs with z = 0 and K = 0, which makes T1 = P, P+ = P + D 0' = P and
    // x + 0 y = x exactly (every product with the zero gain is an exact zero).
    double zn[NZ];
    double un[VAR ? NUC : 1];
    unsigned hn = 1u;
    {
        const MlView vz(a.z, tz8, estride);
        FK_UNROLL for (int c = 0; c < NZ; ++c) zn[c] = vz.load(c);
        if constexpr (VAR) {
            const unsigned hb = mask_or_dummy[trk];
            hn = a.mask ? hb : 1u;
            const MlView vu(u_or_dummy, tu8, estride);
            FK_UNROLL for (int c = 0; c < NUC; ++c) {
                const double v = vu.load(uniform_int((unsigned)c < nu ? c : (int)nu_idx));   // clamped element index: no branch
                un[c] = nu ? v : 0.0;
            }
            FK_UNROLL for (int c = 0; c < NUC; ++c) asm volatile("" ::"v"(un[c]));
        } else if constexpr (MASK) hn = a.mask[trk];
        FK_UNROLL for (int c = 0; c < NZ; ++c) asm volatile("" ::"v"(zn[c]));     // landed, like x and P above
        if constexpr (MASK) asm volatile("" ::"v"(hn));
    }
    _Pragma("nounroll") for (long t = 0; t < a.T; ++t) {
        double z[NZ];
        double u[VAR ? NUC : 1];
        const bool has_z = !MASK || hn != 0u;
        FK_UNROLL for (int c = 0; c < NZ; ++c) z[c] = has_z ? zn[c] : 0.0;
        if constexpr (VAR) {
            FK_UNROLL for (int c = 0; c < NUC; ++c) u[c] = un[c];
        }
        {
            long tn = t + 1 < a.T ? t + 1 : t;
            asm volatile("" : "+s"(tn));      // opaque: keeps the compiler from re-deriving this load one iteration later
            const MlView vz(a.z + tn * N * NZ, tz8, estride);
            FK_UNROLL for (int c = 0; c < NZ; ++c) zn[c] = vz.load(c);
            if constexpr (VAR) {
                const unsigned hb = mask_or_dummy[tn * N + trk];
                hn = a.mask ? hb : 1u;
                const MlView vu(u_or_dummy + tn * N * (long)nu, tu8, estride);
                FK_UNROLL for (int c = 0; c < NUC; ++c) {
                    const double v = vu.load(uniform_int((unsigned)c < nu ? c : (int)nu_idx));
                    un[c] = nu ? v : 0.0;
                }
            } else if constexpr (MASK) hn = a.mask[tn * N + trk];
        }
        // VAR, per-step models: mnext holds this thread's element of model[t+1]; keep it for the hand-over at the
        // end of the step and request model[t+2] (consumed one step later: a counted wait)
        const double mpub = mnext;
        if constexpr (VAR) {
            long t2 = a.model_t ? (t + 2 < a.T ? t + 2 : a.T - 1) : 0;
            asm volatile("" : "+s"(t2));
            mnext = model_elem(t2);
        }
        // the two halves of a step, in the order update_first asks for (kalman_filter.py:966-991)
        if constexpr (!UF) {
#include "kf_ml_predict.inc"
        }
        {
        // ----------------------------------------------------------------- update --
        // Joseph form with the identity-minus-product factors applied implicitly (cf. fk_math_sym.hpp),
        // WITHOUT assuming P symmetric -- P is held by rows and its two triangles round differently;
        // replacing H P by (P H')' turns the contraction of that rounding asymmetry into an
        // amplification, 1e-16 -> 1e-8 in 40 steps of an unstable model:
        //   T1 = (I-KH) P = P - K (H P) ;  G = T1 H' ;  P+ = T1 (I-KH)' + K R K' = T1 + (K R - G) K'
        // PHT = P H', G and every row of T1 / P+ are lane-local; H P needs columns of P, i.e. one
        // three-way sum across the quad (27 values); PHT (for S) and K's rows are gathered.
        double y[NZ], K[R][NZ];
        const double *myH = sH + Lc * R;                        // this lane's columns of H (the H P stage)
        double hc[2][R];
        {
            if constexpr (!HLDS) {
                FK_UNROLL for (int c = 0; c < NZ; ++c) {
                    double acc = HX(c * NX) * x[0];
                    FK_UNROLL for (int k = 1; k < NX; ++k) acc = fma(HX(c * NX + k), x[k], acc);
                    y[c] = z[c] - acc;
                }
            }
            // (update_first: P is the prior the PREVIOUS step's predict left; at t == 0 the initial P lands in covs_p[0]
            // and is overwritten one step later)
            const MlView vPri(a.covs_p + (UF ? (t > 0 ? t - 1 : 0) : t) * N * NX * NX, off_rows, estride, pair_rows);
            double PHT[R][NZ], S[NZ * NZ];
            // the prior covariance's 27 stores in R groups, one per iteration of the P H' stage (element-major outputs)
#define FK_ML_PRIOR_STORES(rr)                                                                                          \
    if (OUTS && !AOS && !SLAB) {                                                                                        \
        if constexpr (PAIRS) {                                                                                          \
            /* 27 elements = 13 pairs + 1, pairs may straddle rows: group rr sends the pairs that END in row rr */      \
            FK_UNROLL for (int f0 = 0; f0 + 1 < R * NX; f0 += 2)                                                        \
                if ((f0 + 1) / NX == (rr)) vPri.store_pair(f0, P[f0 / NX][f0 % NX], P[(f0 + 1) / NX][(f0 + 1) % NX]);   \
            if ((rr) == R - 1) vPri.store(R * NX - 1, P[R - 1][NX - 1]);                                                \
        } else {                                                                                                        \
            FK_UNROLL for (int c_ = 0; c_ < NX; ++c_) vPri.store((rr) * NX + c_, P[(rr)][c_]);                          \
        }                                                                                                               \
    }
            if constexpr (HLDS) {
                // H from LDS (NumPy order, per-step models): row c serves y[c] and column c of P H' in one pass and is requested
                // one row ahead, like F's rows in the predict half (the same sums in the same order)
                double Hr[2][NX];
                FK_UNROLL for (int k = 0; k < NX; ++k) Hr[0][k] = sH[k];
                FK_UNROLL for (int c = 0; c < NZ; ++c) {
                    if (c + 1 < NZ) {
                        FK_UNROLL for (int k = 0; k < NX; ++k) Hr[(c + 1) & 1][k] = sH[(c + 1) * NX + k];
                    }
                    const double (&Hc)[NX] = Hr[c & 1];
                    double acc = Hc[0] * x[0];
                    FK_UNROLL for (int k = 1; k < NX; ++k) acc = fma(Hc[k], x[k], acc);
                    y[c] = z[c] - acc;
                    FK_UNROLL for (int r = 0; r < R; ++r) {
                        double pa = P[r][0] * Hc[0];
                        FK_UNROLL for (int k = 1; k < NX; ++k) pa = fma(P[r][k], Hc[k], pa);
                        PHT[r][c] = pa;
                    }
                    static_assert(NZ == R, "the prior's store groups (one per row block) ride on the NZ iterations");
                    FK_ML_PRIOR_STORES(c);
                    FK_STAGE();
                }
            }
            FK_UNROLL for (int r = 0; r < R && !HLDS; ++r) {
                FK_UNROLL for (int c = 0; c < NZ; ++c) {
                    double acc = P[r][0] * HX(c * NX);
                    FK_UNROLL for (int k = 1; k < NX; ++k) acc = fma(P[r][k], HX(c * NX + k), acc);
                    PHT[r][c] = acc;
                }
                FK_ML_PRIOR_STORES(r);
                FK_STAGE();
            }
            // S = H PHT + R, replicated in every lane: PHT's row k comes from its owner  (R requested in front of the sums)
            double Rs[NZ * NZ];
            FK_UNROLL for (int e = 0; e < NZ * NZ; ++e) Rs[e] = sR[e];
            FK_STAGE();
            double hk[2][NZ];                                   // HLDS: column k of H one iteration ahead
            if constexpr (HLDS) {
                // Round 6: lane Lc forms ROW Lc of S (NZ == R: one row per lane of the quad's three) -- the same nine products
                // in the same order as before, 27 fused multiply-adds instead of 81 -- and the rows are gathered afterwards
                static_assert(NZ == R, "one row of S per lane");
                const double *hrow = sH + Lc * NX;
                hk[0][0] = hrow[0];
                double Srow[NZ];
                FK_UNROLL for (int k = 0; k < NX; ++k) {
                    if (k + 1 < NX) hk[(k + 1) & 1][0] = hrow[k + 1];
                    double pk[NZ];
                    FK_UNROLL for (int c = 0; c < NZ; ++c) {
                        const double v = PHT[k % R][c];
                        pk[c] = (k / R == 0) ? quad_bcast<0>(v) : (k / R == 1) ? quad_bcast<1>(v) : quad_bcast<2>(v);
                    }
                    const double h = hk[k & 1][0];
                    FK_UNROLL for (int c = 0; c < NZ; ++c) Srow[c] = (k == 0) ? h * pk[c] : fma(h, pk[c], Srow[c]);
                    if (k % 3 == 2) FK_STAGE();
                }
                FK_UNROLL for (int r = 0; r < NZ; ++r)
                    FK_UNROLL for (int c = 0; c < NZ; ++c)
                        S[r * NZ + c] = (r == 0) ? quad_bcast<0>(Srow[c]) : (r == 1) ? quad_bcast<1>(Srow[c]) : quad_bcast<2>(Srow[c]);
            }
            FK_UNROLL for (int k = 0; k < NX && !HLDS; ++k) {
                double pk[NZ];
                FK_UNROLL for (int c = 0; c < NZ; ++c) {
                    const double v = PHT[k % R][c];
                    pk[c] = (k / R == 0) ? quad_bcast<0>(v) : (k / R == 1) ? quad_bcast<1>(v) : quad_bcast<2>(v);
                }
                FK_UNROLL for (int r = 0; r < NZ; ++r) {
                    const double h = HX(r * NX + k);
                    FK_UNROLL for (int c = 0; c < NZ; ++c)
                        S[r * NZ + c] = (k == 0) ? h * pk[c] : fma(h, pk[c], S[r * NZ + c]);
                }
            }
            FK_UNROLL for (int e = 0; e < NZ * NZ; ++e) S[e] += Rs[e];
            FK_STAGE();
            FK_UNROLL for (int r = 0; r < R; ++r) hc[0][r] = myH[r];     // requested in front of the factorisation that does not need them
            FK_STAGE();
            double Lf[NZ * NZ], d[NZ], dinv[NZ], Kr[R * NZ];
            FK_UNROLL for (int e = 0; e < NZ * NZ; ++e) Lf[e] = S[e];
            if (!ldlt2<NZ>(Lf, d, dinv) && has_z) st |= ST_NOT_PD;
            FK_UNROLL for (int r = 0; r < R; ++r)
                FK_UNROLL for (int c = 0; c < NZ; ++c) Kr[r * NZ + c] = PHT[r][c];
            solve_rows_ldlt<R, NZ>(Lf, dinv, Kr);
            FK_UNROLL for (int r = 0; r < R; ++r)
                FK_UNROLL for (int c = 0; c < NZ; ++c) K[r][c] = has_z ? Kr[r * NZ + c] : 0.0;
        }
        FK_STAGE();
        {
            // H P: this lane's rows contribute sum_r H[c][row0 + r] P[r][:]; the quad adds the three parts
            // (this lane's R coefficients of row c one iteration ahead of their use: see kf_ml_predict.inc)
            double HP[NZ][NX];
            FK_UNROLL for (int c = 0; c < NZ; ++c) {
                if (c + 1 < NZ) {
                    FK_UNROLL for (int r = 0; r < R; ++r) hc[(c + 1) & 1][r] = myH[(c + 1) * NX + r];
                }
                FK_UNROLL for (int j = 0; j < NX; ++j) {
                    double acc = hc[c & 1][0] * P[0][j];
                    FK_UNROLL for (int r = 1; r < R; ++r) acc = fma(hc[c & 1][r], P[r][j], acc);
                    HP[c][j] = acc;
                }
                FK_UNROLL for (int j = 0; j < NX; ++j) HP[c][j] = (HP[c][j] + quad_rot<0x09>(HP[c][j])) + quad_rot<0x52>(HP[c][j]);
                FK_STAGE();
            }
            // T1 = P - K (H P) (own rows, in place)
            double Rr[NZ * NZ];                                // R for the D stage, requested a stage early
            FK_UNROLL for (int r = 0; r < R; ++r) {
                if (r == R - 1) {
                    FK_UNROLL for (int e = 0; e < NZ * NZ; ++e) Rr[e] = sR[e];
                    FK_STAGE();
                }
                FK_UNROLL for (int j = 0; j < NX; ++j) {
                    double acc = P[r][j];
                    FK_UNROLL for (int c = 0; c < NZ; ++c) acc = fma(-K[r][c], HP[c][j], acc);
                    P[r][j] = acc;
                }
                FK_STAGE();
            }
            // D = K R - T1 H' (own rows)
            double D[R][NZ];
            if constexpr (HLDS) {
                double Hr[2][NX];
                FK_UNROLL for (int k = 0; k < NX; ++k) Hr[0][k] = sH[k];
                FK_UNROLL for (int c = 0; c < NZ; ++c) {
                    if (c + 1 < NZ) {
                        FK_UNROLL for (int k = 0; k < NX; ++k) Hr[(c + 1) & 1][k] = sH[(c + 1) * NX + k];
                    }
                    const double (&Hc)[NX] = Hr[c & 1];
                    FK_UNROLL for (int r = 0; r < R; ++r) {
                        double kr = K[r][0] * Rr[c];
                        FK_UNROLL for (int q = 1; q < NZ; ++q) kr = fma(K[r][q], Rr[q * NZ + c], kr);
                        double g = P[r][0] * Hc[0];
                        FK_UNROLL for (int k = 1; k < NX; ++k) g = fma(P[r][k], Hc[k], g);
                        D[r][c] = kr - g;
                    }
                    FK_STAGE();
                }
            }
            FK_UNROLL for (int r = 0; r < R && !HLDS; ++r)
                FK_UNROLL for (int c = 0; c < NZ; ++c) {
                    double kr = K[r][0] * Rr[c];
                    FK_UNROLL for (int q = 1; q < NZ; ++q) kr = fma(K[r][q], Rr[q * NZ + c], kr);
                    double g = P[r][0] * HX(c * NX);
                    FK_UNROLL for (int k = 1; k < NX; ++k) g = fma(P[r][k], HX(c * NX + k), g);
                    D[r][c] = kr - g;
                }
            FK_STAGE();
            // P+ = T1 + D K' : column j needs K's row j from its owner; the same row updates x[j]
            FK_UNROLL for (int j = 0; j < NX; ++j) {
                double Kj[NZ];
                FK_UNROLL for (int c = 0; c < NZ; ++c) {
                    const double v = K[j % R][c];
                    Kj[c] = (j / R == 0) ? quad_bcast<0>(v) : (j / R == 1) ? quad_bcast<1>(v) : quad_bcast<2>(v);
                }
                double xa = x[j];
                FK_UNROLL for (int c = 0; c < NZ; ++c) xa = fma(Kj[c], y[c], xa);
                x[j] = xa;
                FK_UNROLL for (int r = 0; r < R; ++r) {
                    double acc = P[r][j];
                    FK_UNROLL for (int c = 0; c < NZ; ++c) acc = fma(D[r][c], Kj[c], acc);
                    P[r][j] = acc;
                }
            }
        }
        if (OUTS && !AOS && !SLAB) {
            const MlView vx(a.means + t * N * NX, t8, estride, pair_x);
            store_x<NX, PAIRS ? 1 : 0>(vx, x);
        }
        if constexpr (OUTS && AOS)
            ml_store_aos<R, NX>(x, P, a.means + (t * N + w0) * NX, a.covs + (t * N + w0) * NX * NX, tile, lane, Lc, valid);
        if constexpr (OUTS && SLAB)
            ml_store_soa_slab<R, NX>(x, P, a.means + t * N * NX + w0, a.covs + t * N * NX * NX + w0, (unsigned)N * 8u, tile, lane, Lc, valid);
        }      // update half
        if constexpr (UF) {
#include "kf_ml_predict.inc"
        }
        if constexpr (VAR) {
            // publish model[t+1] into the other LDS buffer: nobody reads that buffer during step t, and one
            // barrier makes it visible for step t+1 (every thread of the workgroup runs all T steps)
            double *nb = smem + ((t + 1) & 1) * MSTR;
            nb[threadIdx.x < (unsigned)MLEN ? threadIdx.x : (unsigned)MLEN] = mpub;   // threads past the model: one pad slot
            __syncthreads();
            sF = nb + LM::OFF_F;
            sQ = nb + LM::OFF_Q;
            sH = nb + LM::OFF_H;
            sR = nb + LM::OFF_R;
            sB = nb + LM::SIZE;
            myF = sF + Lc * (R * NX);
            myQ = sQ + Lc * (R * NX);
        }
    }
    if (OUTS && !AOS && !SLAB && a.T > 0) {      // the last step's posterior (update_first: prior) covariance -- the others were stored one step late
        const MlView vP((UF ? a.covs_p : a.covs) + (a.T - 1) * N * NX * NX, off_rows, estride);
        FK_UNROLL for (int r = 0; r < R; ++r)
            FK_UNROLL for (int c = 0; c < NX; ++c) vP.store(r * NX + c, P[r][c]);
    }
    // (the final state goes back in place: only a track's own quad writes it -- a duplicating tail quad that loaded
    // x0 / P0 late must not find the final state there; these stores are outside the time loop)
    // ... and no lane of the workgroup may still be about to LOAD x0 / P0 when an owner overwrites them: every wave has
    // consumed its initial state once it arrives here (one barrier per launch, outside the time loop)
    __syncthreads();
    if (owner) {
        bool fin = all_finite<NX>(x);
        FK_UNROLL for (int r = 0; r < R; ++r)
            FK_UNROLL for (int c = 0; c < NX; ++c) fin = fin && (fabs(P[r][c]) <= 1.79769313486231570815e+308);
        if (PERS && pers_h + 1 < a_in.pers_H) {
            // not the group's last chunk: the state goes to the hand-over block (agent-scope stores), x / P stay untouched
            const unsigned n8 = (unsigned)N * 8u;
            const MlView wx(a_in.pers_ws, (unsigned)trk * 8u, n8), wP(a_in.pers_ws + (long)NX * N, (unsigned)trk * 8u + Lc * (unsigned)(R * NX) * n8, n8);
            FK_UNROLL for (int k = 0; k < NX; ++k) wx.template store<HAUX>(k, x[k]);
            FK_UNROLL for (int r = 0; r < R; ++r)
                FK_UNROLL for (int c = 0; c < NX; ++c) wP.template store<HAUX>(r * NX + c, P[r][c]);
        } else {
            const MlView vx(a.x, t8, estride), vP(a.P, off_rows, estride);
            FK_UNROLL for (int k = 0; k < NX; ++k) vx.store(k, x[k]);
            FK_UNROLL for (int r = 0; r < R; ++r)
                FK_UNROLL for (int c = 0; c < NX; ++c) vP.store(r * NX + c, P[r][c]);
        }
        if (a.status) {
            int s = st | (fin ? 0 : ST_NONFINITE);
            s |= __builtin_amdgcn_mov_dpp(s, 0x55 * 1, 0xf, 0xf, true) | __builtin_amdgcn_mov_dpp(s, 0x55 * 2, 0xf, 0xf, true);
            if constexpr (PERS) {
                if (L == 0) {
                    const int old = a.status_or ? __hip_atomic_load(&a.status[trk], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
                    __hip_atomic_store(&a.status[trk], old | s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else {
                if (L == 0) a.status[trk] = a.status_or ? (a.status[trk] | s) : s;
            }
        }
    }
    if constexpr (PERS) {
        // this chunk's final state (and status) is in place -- written with agent-scope stores, which are coherent across the
        // XCDs by themselves (a release FENCE here writes back the whole L2 the kernel is streaming 14 GB of outputs through:
        // measured 0.55 ms per chunk).  They are COMPLETE only once this wave's vmcnt has drained: a workgroup barrier does not
        // wait for VMEM on gfx950 (the compiler emits `s_waitcnt vmcnt(63)` -- nothing -- in front of s_barrier; ADVICE r4: with
        // status == NULL no other wait stood between the hand-over stores and the flag), so every wave drains explicitly --
        // inline asm, invisible to the pass that drops "redundant" waits -- then the barrier, then the chunk is published
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(&a_in.pers_ctl[1 + pers_g], pers_h + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        break;
    }
    }      // tickets
}

#if FK_ML_PART != 2
// AOS output of one R x NX row block per lane (the wave's 16 x NX*NX slab) staged through the wave's own
// columns of the smoother's parking buffer: element i of the slab lives at park[i / 64][64 * wave + i % 64].
// Only used at points of the step where the rows it touches (0 .. 16*NX*NX/64) hold nothing live.
template <int R, int NX, int PARK_COLS>
__device__ __forceinline__ void ml_store_rows_aos_park(const double (&M)[R][NX], double *dst, double (*park)[PARK_COLS],
                                                       unsigned lane, unsigned wave, unsigned Lc, unsigned valid)
{
    constexpr int EP = NX * NX, UP = 16 * EP / 2;
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
    const unsigned q = lane >> 2, c0 = 64u * wave;
    ml_wave_fence();
    FK_UNROLL for (int r = 0; r < R; ++r)
        FK_UNROLL for (int c = 0; c < NX; ++c) {
            const unsigned i = q * EP + Lc * (R * NX) + r * NX + c;
            park[i >> 6][c0 + (i & 63u)] = M[r][c];
        }
    ml_wave_fence();
    const rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc(dst, 0, (int)(valid * (unsigned)EP * 8u), 0x00020000);
    ml_copy_units<UP, 4>(lane, [&](unsigned unit) { return &park[(2u * unit) >> 6][c0 + ((2u * unit) & 63u)]; },
                         [&](unsigned unit, bool ok, const u32x4 &v) {
                             __builtin_amdgcn_raw_buffer_store_b128(v, rP, ok ? unit * 16u : ML_OFF_DROP, 0, 0);
                         });
    ml_wave_fence();
}

// row k of a rows-per-lane matrix, delivered to every lane of the quad by its owner
#define FK_ROW_FROM_OWNER(dst, M, k, LEN)                                                                     \
    FK_UNROLL for (int j_ = 0; j_ < (LEN); ++j_) {                                                            \
        const double v_ = M[(k) % R][j_];                                                                     \
        dst[j_] = ((k) / R == 0) ? quad_bcast<0>(v_) : ((k) / R == 1) ? quad_bcast<1>(v_) : quad_bcast<2>(v_); \
    }

// rts_smoother for dim_x = 9 on the same three-lanes-per-track layout (kalman_filter.py:1066-1072):
//   Pp = F P F' + Q ;  K = (P F') Pp^-1 ;  x += K (xn - F x) ;  P += K (Pn - Pp) K'
// T = P F' and every row of Pp, K, E = K (Pn - Pp) and P live in the lane that owns the row; rows of
// T, of Pn - Pp and of K are broadcast by their owners; the 9 x 9 factorisation of Pp is done by
// every lane on a gathered packed copy (45 doubles) and each lane back-substitutes its own three
// rows of K.  The filtered P is read twice (second time from L2) instead of being held across the
// factorisation.  Shared constant F, Q; SOA; K and Pp outputs both present.
// PERS (round 6): the forward kernel's persistent grid (see kf_ml_kernel) for the smoother -- G track groups x H time chunks drawn
// as tickets by 2 workgroups per CU, chunk-major from the END of the time axis (a chunk needs the smoothed state of the first step
// of the chunk after it in time: the ticket of that one is at least G draws older).  The state travels through the element-major
// hand-over block with agent-scope accesses and is announced by the group's completion word; the window of a later-drawn chunk
// overlaps its predecessor's by the one step it starts from (the `cont` convention of the chunked calls), whose outputs it leaves
// alone.  Same arithmetic per track: bit-identical to the single launch.
template <int R, int WAVES, int MODE, bool PERS = false>
__global__ void __launch_bounds__(BLOCK, WAVES)
rts_ml_kernel(const RtsArgs a_in)
{
    constexpr int HAUX = PERS ? 16 : 0;      // sc1 = agent scope (MlView::load / store)
    constexpr int NX = 3 * R, PL = NX * (NX + 1) / 2;
    __shared__ double smem[2 * NX * NX];
    // x and xn - F x are parked here ([element][lane]: conflict-free) across the E = K D stage: spilled
    // to scratch instead, their reloads would be vector-memory operations that retire in order behind
    // the step's stores (s_waitcnt vmcnt(0)); LDS traffic has its own counter.
    __shared__ double park[R * NX + NX][BLOCK];   // rows 0..26: D = Pn - Pp across the factorisation (the register peak), then x | dx; rows 27..35: x from the top of the step
    lds_fill<NX, NX>(smem, a_in.F, NX, NX, 1.0, threadIdx.x);
    lds_fill<NX, NX>(smem + NX * NX, a_in.Q, NX, NX, 0.0, threadIdx.x);
    __syncthreads();
    const double *sF = smem, *sQ = smem + NX * NX;

    RtsArgs a = a_in;
    [[maybe_unused]] int pers_g = 0, pers_c = 0;
    __shared__ int s_task;
    for (;;) {                                                   // PERS: one trip per ticket; otherwise exactly one trip
    unsigned bid = blockIdx.x;
    if constexpr (PERS) {
        __syncthreads();                                         // the previous ticket's LDS traffic is over, s_task is free
        if (threadIdx.x == 0) s_task = atomicAdd(a_in.pers_ctl, 1);
        __syncthreads();
        const int task = __builtin_amdgcn_readfirstlane(s_task);
        const int G = a_in.pers_G, H = a_in.pers_H;
        if (task >= G * H) break;
        pers_c = task / G;                                       // chunk number counted from the END of the time axis
        pers_g = task - pers_c * G;
        bid = (unsigned)pers_g;
        const long hh = H - 1 - pers_c;
        const long t0 = a_in.T * hh / H, t1 = a_in.T * (hh + 1) / H, NN = a_in.N;
        a = a_in;
        a.cont = pers_c > 0 ? 1 : 0;
        a.T = t1 - t0 + (pers_c > 0 ? 1 : 0);                    // (+ the step the chunk starts from, smoothed by the chunk before it)
        a.Xs = a_in.Xs + t0 * NN * NX;
        a.xs = a_in.xs + t0 * NN * NX;
        a.Ps = a_in.Ps + t0 * NN * (NX * NX);
        a.Ps_out = a_in.Ps_out + t0 * NN * (NX * NX);
        a.K = a_in.K + t0 * NN * (NX * NX);
        a.Pp = a_in.Pp + t0 * NN * (NX * NX);
        a.status_or = pers_c > 0 ? 1 : a_in.status_or;
        if (pers_c > 0) {
            if (threadIdx.x == 0) {
                while (__hip_atomic_load(&a_in.pers_ctl[1 + pers_g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < pers_c) __builtin_amdgcn_s_sleep(8);
            }
            __syncthreads();
        }
    }
    const long N = a.N, T = a.T;
    const unsigned L = threadIdx.x & 3u;
    const unsigned Lc = L < 3u ? L : 2u;
    const long i0 = a.cnt ? a.i0 : 0, iend = a.cnt ? a.i0 + a.cnt : N;     // this launch's track group (chunked calls)
    long trk = i0 + (long)bid * (BLOCK / 4) + (threadIdx.x >> 2);
    const unsigned odd = (threadIdx.x >> 2) & 1u;
    if (trk >= iend) trk = MODE == 1 ? iend - 2 + odd : iend - 1;
    // SOA: element e of a track at track*8 + e*N*8 ; AOS (MODE 2): track*E*8 + e*8
    constexpr bool AOS = MODE == 2;
    const unsigned estride = AOS ? 8u : (unsigned)N * 8u;
    const unsigned t8 = (unsigned)trk * (AOS ? (unsigned)NX * 8u : 8u);
    const unsigned off_rows = (AOS ? (unsigned)trk * (unsigned)(NX * NX) * 8u : (unsigned)trk * 8u) + Lc * (unsigned)(R * NX) * estride;
    const unsigned pair_rows = odd ? off_rows - 8u + estride : off_rows;
    const unsigned pair_x = odd ? t8 - 8u + estride : t8;
    const double *myF = sF + Lc * (R * NX), *myQ = sQ + Lc * (R * NX);
    // AOS: covariance-like outputs leave through the parking buffer as 1 KiB stores (first track of this
    // wave, how many of its 16 tracks exist)
    const unsigned lane = threadIdx.x & 63u, wave = wave_index();
    const long w0 = i0 + (long)bid * (BLOCK / 4) + (long)wave * 16;
    const unsigned valid = (unsigned)(iend - w0 >= 16 ? 16 : (iend - w0 > 0 ? iend - w0 : 0));
    const long xs_blk = N * NX, ps_blk = N * (long)NX * NX;

    // k = T-1: smoothed == filtered; K = 0; Pp = Ps   (kalman_filter.py:1063-1065)
    double xn[NX], Pn[R][NX];
    if (PERS && a.cont) {
        // a later ticket of the group: the smoothed state of step T-1 of this window, from the hand-over block
        const unsigned n8 = (unsigned)N * 8u;
        const MlView wx(a_in.pers_ws, (unsigned)trk * 8u, n8), wP(a_in.pers_ws + (long)NX * N, (unsigned)trk * 8u + Lc * (unsigned)(R * NX) * n8, n8);
        FK_UNROLL for (int k = 0; k < NX; ++k) xn[k] = wx.template load<HAUX>(k);
        FK_UNROLL for (int r = 0; r < R; ++r)
            FK_UNROLL for (int c = 0; c < NX; ++c) Pn[r][c] = wP.template load<HAUX>(r * NX + c);
    } else if (a.cont) {
        // a later chunk of the call: step T-1 of this window was smoothed by the chunk that ran before (after it in time)
        const MlView vx(a.xs + (T - 1) * xs_blk, t8, estride), vP(a.Ps_out + (T - 1) * ps_blk, off_rows, estride);
        FK_UNROLL for (int k = 0; k < NX; ++k) xn[k] = vx.load(k);
        FK_UNROLL for (int r = 0; r < R; ++r)
            FK_UNROLL for (int c = 0; c < NX; ++c) Pn[r][c] = vP.load(r * NX + c);
    } else {
        const MlView vx(a.Xs + (T - 1) * xs_blk, t8, estride), vP(a.Ps + (T - 1) * ps_blk, off_rows, estride);
        FK_UNROLL for (int k = 0; k < NX; ++k) xn[k] = vx.load(k);
        FK_UNROLL for (int r = 0; r < R; ++r)
            FK_UNROLL for (int c = 0; c < NX; ++c) Pn[r][c] = vP.load(r * NX + c);
        const MlView ox(a.xs + (T - 1) * xs_blk, t8, estride), oP(a.Ps_out + (T - 1) * ps_blk, off_rows, estride);
        const MlView oK(a.K + (T - 1) * ps_blk, off_rows, estride), oPp(a.Pp + (T - 1) * ps_blk, off_rows, estride);
        FK_UNROLL for (int k = 0; k < NX; ++k) ox.store(k, xn[k]);
        FK_UNROLL for (int r = 0; r < R; ++r)
            FK_UNROLL for (int c = 0; c < NX; ++c) {
                oP.store(r * NX + c, Pn[r][c]);
                oPp.store(r * NX + c, Pn[r][c]);
                oK.store(r * NX + c, 0.0);
            }
    }
    int st = 0;

    // The filtered P, x of step k are fetched at the END of step k+1, before that step's last stores
    // are issued: vmcnt retires in order, so a load issued behind 36 stores is only usable once they
    // have all been acknowledged -- a pipeline drain per step otherwise.
    double Pnx[R][NX], xnx[NX];
    {
        const long k0 = T >= 2 ? T - 2 : 0;
        const MlView vx(a.Xs + k0 * xs_blk, t8, estride, pair_x), vP(a.Ps + k0 * ps_blk, off_rows, estride, pair_rows);
        load_rows<R, NX, MODE>(vP, Pnx);
        load_x<NX, MODE>(vx, xnx);
        FK_UNROLL for (int r = 0; r < R; ++r)
            FK_UNROLL for (int c = 0; c < NX; ++c) asm volatile("" ::"v"(Pnx[r][c]));
        FK_UNROLL for (int i = 0; i < NX; ++i) asm volatile("" ::"v"(xnx[i]));
    }
    _Pragma("nounroll") for (long k = T - 2; k >= 0; --k) {
        const MlView vP(a.Ps + k * ps_blk, off_rows, estride, pair_rows);
        double Tm[R][NX];
        FK_UNROLL for (int i = 0; i < NX; ++i) park[R * NX + i][threadIdx.x] = xnx[i];
        {
            double P[R][NX];
            FK_UNROLL for (int r = 0; r < R; ++r)
                FK_UNROLL for (int c = 0; c < NX; ++c) P[r][c] = Pnx[r][c];
            // T = P F'  (F's rows requested from LDS one iteration ahead: see kf_ml_predict.inc)
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
        // Pp = F T + Q (own rows)
        double Pp[R][NX];
        double fq[2][R];                           // this lane's column q of F one iteration ahead
        FK_UNROLL for (int r = 0; r < R; ++r) fq[0][r] = myF[r * NX];
        FK_UNROLL for (int q = 0; q < NX; ++q) {
            if (q + 1 < NX) {
                FK_UNROLL for (int r = 0; r < R; ++r) fq[(q + 1) & 1][r] = myF[r * NX + q + 1];
            }
            double Tq[NX];
            FK_ROW_FROM_OWNER(Tq, Tm, q, NX);
            FK_UNROLL for (int r = 0; r < R; ++r) {
                const double f = fq[q & 1][r];
                FK_UNROLL for (int j = 0; j < NX; ++j) Pp[r][j] = (q == 0) ? f * Tq[j] : fma(f, Tq[j], Pp[r][j]);
            }
            FK_STAGE();
        }
        {
            const MlView oPp(a.Pp + k * ps_blk, off_rows, estride, pair_rows);
            // (Q's 27 values in one batch, then the sums: left to itself the compiler interleaves one read, one wait, one add)
            FK_UNROLL for (int r = 0; r < R; ++r) {
                double Qr[NX];
                FK_UNROLL for (int j = 0; j < NX; ++j) Qr[j] = myQ[r * NX + j];
                FK_STAGE();
                FK_UNROLL for (int j = 0; j < NX; ++j) Pp[r][j] += Qr[j];
            }
            if constexpr (AOS) ml_store_rows_aos_park<R, NX, BLOCK>(Pp, a.Pp + (k * N + w0) * NX * NX, park, lane, wave, Lc, valid);
            else store_rows<R, NX, MODE>(oPp, Pp);
        }
        FK_STAGE();
        // K = T Pp^-1: every lane factors a gathered packed copy of Pp and solves its own rows
        {
            double Pf[PL], d[NX], dinv[NX];
            // D = Pn - Pp (own rows) goes to LDS first: Pn is dead from here, Pp's rows after the gather
            FK_UNROLL for (int r = 0; r < R; ++r)
                FK_UNROLL for (int j = 0; j < NX; ++j) park[r * NX + j][threadIdx.x] = Pn[r][j] - Pp[r][j];
            FK_STAGE();
            FK_UNROLL for (int i = 0; i < NX; ++i)
                FK_UNROLL for (int j = i; j < NX; ++j) {
                    const double v = Pp[i % R][j];
                    Pf[sym_idx<NX>(i, j)] = (i / R == 0) ? quad_bcast<0>(v) : (i / R == 1) ? quad_bcast<1>(v) : quad_bcast<2>(v);
                }
            FK_STAGE();
            if (!ldlt_packed<NX>(Pf, d, dinv)) st |= ST_NOT_PD;
            FK_STAGE();
            FK_UNROLL for (int r = 0; r < R; ++r) {
                solve_row_packed<NX>(Pf, dinv, Tm[r]);      // Tm's rows become K's rows
                FK_STAGE();
            }
        }
        FK_UNROLL for (int r = 0; r < R; ++r)
            FK_UNROLL for (int j = 0; j < NX; ++j) Pn[r][j] = park[r * NX + j][threadIdx.x];
        FK_STAGE();
        {
            const MlView oK(a.K + k * ps_blk, off_rows, estride, pair_rows);
            if constexpr (AOS) ml_store_rows_aos_park<R, NX, BLOCK>(Tm, a.K + (k * N + w0) * NX * NX, park, lane, wave, Lc, valid);
            else store_rows<R, NX, MODE>(oK, Tm);
        }
        // x += K (xn - F x), replicated: K's rows come from their owners.  (x is only fetched now: nine
        // doubles less across the factorisation, which is where this kernel's register peak is.)
        {
            double x[NX], dx[NX];
            FK_UNROLL for (int i = 0; i < NX; ++i) x[i] = park[R * NX + i][threadIdx.x];
            // Round 6: F x by rows of the quad -- lane Lc forms rows 3 Lc .. 3 Lc + 2 from its own rows of F (the same products in
            // the same order as the nine rows every lane used to form: 27 fused multiply-adds instead of 81), gathered afterwards
            {
                double fxo[R];
                double Fo[2][NX];
                FK_UNROLL for (int q = 0; q < NX; ++q) Fo[0][q] = myF[q];
                FK_UNROLL for (int r = 0; r < R; ++r) {
                    if (r + 1 < R) {
                        FK_UNROLL for (int q = 0; q < NX; ++q) Fo[(r + 1) & 1][q] = myF[(r + 1) * NX + q];
                    }
                    double acc = Fo[r & 1][0] * x[0];
                    FK_UNROLL for (int q = 1; q < NX; ++q) acc = fma(Fo[r & 1][q], x[q], acc);
                    fxo[r] = acc;
                }
                FK_STAGE();
                FK_UNROLL for (int i = 0; i < NX; ++i) {
                    const double v = fxo[i % R];
                    const double fx = (i / R == 0) ? quad_bcast<0>(v) : (i / R == 1) ? quad_bcast<1>(v) : quad_bcast<2>(v);
                    dx[i] = xn[i] - fx;
                }
                FK_STAGE();
            }
            FK_UNROLL for (int i = 0; i < NX; ++i) {
                park[i][threadIdx.x] = x[i];
                park[NX + i][threadIdx.x] = dx[i];
            }
            FK_STAGE();
            // E = K D (own rows): rows of D broadcast
            double E[R][NX];
            FK_UNROLL for (int q = 0; q < NX; ++q) {
                double Dq[NX];
                FK_ROW_FROM_OWNER(Dq, Pn, q, NX);
                FK_UNROLL for (int r = 0; r < R; ++r) {
                    const double kq = Tm[r][q];
                    FK_UNROLL for (int j = 0; j < NX; ++j) E[r][j] = (q == 0) ? kq * Dq[j] : fma(kq, Dq[j], E[r][j]);
                }
                FK_STAGE();
            }
            // P (filtered, second read) += E K' : column j needs K's row j; the same row updates x[j]
            load_rows<R, NX, MODE>(vP, Pn);
            double dxr[NX];
            FK_UNROLL for (int i = 0; i < NX; ++i) dxr[i] = park[NX + i][threadIdx.x];
            // Round 6: x + K dx by rows of the quad too: this lane's rows of K are its own registers (Tm), the sums the ones every lane
            // used to form for all nine rows; gathered from their owners below
            double xao[R];
            FK_UNROLL for (int r = 0; r < R; ++r) {
                double xa = park[Lc * R + r][threadIdx.x];
                FK_UNROLL for (int q = 0; q < NX; ++q) xa = fma(Tm[r][q], dxr[q], xa);
                xao[r] = xa;
            }
            FK_STAGE();
            FK_UNROLL for (int j = 0; j < NX; ++j) {
                double Kj[NX];
                FK_ROW_FROM_OWNER(Kj, Tm, j, NX);
                {
                    const double v = xao[j % R];
                    xn[j] = (j / R == 0) ? quad_bcast<0>(v) : (j / R == 1) ? quad_bcast<1>(v) : quad_bcast<2>(v);
                }
                FK_UNROLL for (int r = 0; r < R; ++r) {
                    double acc = Pn[r][j];
                    FK_UNROLL for (int q = 0; q < NX; ++q) acc = fma(E[r][q], Kj[q], acc);
                    Pn[r][j] = acc;
                }
                FK_STAGE();
            }
        }
        {
            long kn = k > 0 ? k - 1 : 0;
            asm volatile("" : "+s"(kn));
            const MlView nx(a.Xs + kn * xs_blk, t8, estride, pair_x), nP(a.Ps + kn * ps_blk, off_rows, estride, pair_rows);
            load_rows<R, NX, MODE>(nP, Pnx);
            load_x<NX, MODE>(nx, xnx);
        }
        FK_STAGE();
        {
            const MlView ox(a.xs + k * xs_blk, t8, estride, pair_x), oP(a.Ps_out + k * ps_blk, off_rows, estride, pair_rows);
            store_x<NX, MODE>(ox, xn);
            if constexpr (AOS) ml_store_rows_aos_park<R, NX, BLOCK>(Pn, a.Ps_out + (k * N + w0) * NX * NX, park, lane, wave, Lc, valid);
            else store_rows<R, NX, MODE>(oP, Pn);
        }
    }
    if (a.status) {
        bool fin = all_finite<NX>(xn);
        FK_UNROLL for (int r = 0; r < R; ++r)
            FK_UNROLL for (int c = 0; c < NX; ++c) fin = fin && (fabs(Pn[r][c]) <= 1.79769313486231570815e+308);
        int s = st | (fin ? 0 : ST_NONFINITE);
        s |= __builtin_amdgcn_mov_dpp(s, 0x55 * 1, 0xf, 0xf, true) | __builtin_amdgcn_mov_dpp(s, 0x55 * 2, 0xf, 0xf, true);
        if constexpr (PERS) {
            if (L == 0) {
                const int old = a.status_or ? __hip_atomic_load(&a.status[trk], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
                __hip_atomic_store(&a.status[trk], old | s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else {
            if (L == 0) a.status[trk] = a.status_or ? (a.status[trk] | s) : s;
        }
    }
    if constexpr (PERS) {
        if (pers_c + 1 < a_in.pers_H) {
            // not the group's last ticket: the smoothed state of this window's first step goes to the hand-over block
            const unsigned n8 = (unsigned)N * 8u;
            const MlView wx(a_in.pers_ws, (unsigned)trk * 8u, n8), wP(a_in.pers_ws + (long)NX * N, (unsigned)trk * 8u + Lc * (unsigned)(R * NX) * n8, n8);
            FK_UNROLL for (int k = 0; k < NX; ++k) wx.template store<HAUX>(k, xn[k]);
            FK_UNROLL for (int r = 0; r < R; ++r)
                FK_UNROLL for (int c = 0; c < NX; ++c) wP.template store<HAUX>(r * NX + c, Pn[r][c]);
        }
        // (complete only once every wave's vmcnt has drained: see kf_ml_kernel)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(&a_in.pers_ctl[1 + pers_g], pers_c + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        break;
    }
    }      // tickets
}

static int launch_rts_ml_one(const RtsArgs &a, int layout, hipStream_t s)
{
    const long cnt = a.cnt ? a.cnt : a.N;
    const dim3 grid((unsigned)((cnt + BLOCK / 4 - 1) / (BLOCK / 4))), block(BLOCK);
    const char *pv = getenv("FK_ML_PAIRS");
    const bool pairs = (a.N % 2 == 0) && (cnt % 2 == 0) && cnt >= 2 && !(pv && atoi(pv) == 0);
    if (layout == FK_LAYOUT_AOS) hipLaunchKernelGGL((rts_ml_kernel<3, FK_ML_WAVES, 2>), grid, block, 0, s, a);
    else if (pairs) hipLaunchKernelGGL((rts_ml_kernel<3, FK_ML_WAVES, 1>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((rts_ml_kernel<3, FK_ML_WAVES, 0>), grid, block, 0, s, a);
    return check_launch("rts_ml_kernel");
}

static int launch_rts_ml_chunked(const RtsArgs &a, int layout, hipStream_t s);

// The smoother on the persistent grid (rts_ml_kernel<..., PERS>): same conditions, same scratch allocation scheme and the same
// chunk count as launch_kf_ml_persistent below.  OFF unless FK_RTS_PERSIST=1: measured at configs[2] (1e5 tracks x 100 steps,
// one lease, A/B/A/B): single launch 5.51 / 5.51 ms, tickets 5.51 / 5.61 (H = 3), 5.54 (2), 5.43 (4) -- nothing, where the forward
// kernel gained 3-7 % (profiles/r06/c3_rts_persist.txt).  The smoother's fourth round of workgroups is as empty as the forward
// kernel's, but its step is twice as long on the memory side (2736 against 1464 bytes per track-step): the three full rounds
// already keep the memory system as busy as this kernel can.  Kept for the bit-identity test and the next idea.  FK_RTS_PERSIST_H: chunks.
static int launch_rts_ml_persistent(const RtsArgs &a, int layout, hipStream_t s)
{
    const char *pv = getenv("FK_RTS_PERSIST");
    if (!pv || atoi(pv) == 0) return 1;
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                   ? prop.multiProcessorCount : 256;
    }
    const long cnt = a.cnt ? a.cnt : a.N;
    const long G = (cnt + BLOCK / 4 - 1) / (BLOCK / 4), slots = 2L * n_cu;
    long H = a.T >= 48 ? 3 : a.T / 16;
    if (const char *hv = getenv("FK_RTS_PERSIST_H")) H = atol(hv);
    if (G <= slots || H < 2 || a.T / H < 2 || G > (1L << 24)) return 1;
    int *ctl = nullptr;
    const size_t cbytes = ((size_t)(1 + G) * sizeof(int) + 255) & ~(size_t)255, wbytes = (size_t)90 * (size_t)a.N * sizeof(double);
    if (hipMallocAsync((void **)&ctl, cbytes + wbytes, s) != hipSuccess || !ctl) { (void)hipGetLastError(); return 1; }
    if (hipMemsetAsync(ctl, 0, cbytes, s) != hipSuccess) { (void)hipFreeAsync(ctl, s); (void)hipGetLastError(); return 1; }
    RtsArgs b = a;
    b.pers_ctl = ctl;
    b.pers_ws = reinterpret_cast<double *>(reinterpret_cast<char *>(ctl) + cbytes);
    b.pers_G = (int)G;
    b.pers_H = (int)H;
    const dim3 grid((unsigned)slots), block(BLOCK);
    const char *pp = getenv("FK_ML_PAIRS");
    const bool pairs = (a.N % 2 == 0) && (cnt % 2 == 0) && cnt >= 2 && !(pp && atoi(pp) == 0);
    if (layout == FK_LAYOUT_AOS) hipLaunchKernelGGL((rts_ml_kernel<3, FK_ML_WAVES, 2, true>), grid, block, 0, s, b);
    else if (pairs) hipLaunchKernelGGL((rts_ml_kernel<3, FK_ML_WAVES, 1, true>), grid, block, 0, s, b);
    else hipLaunchKernelGGL((rts_ml_kernel<3, FK_ML_WAVES, 0, true>), grid, block, 0, s, b);
    const int rc = check_launch("rts_ml_kernel<pers>");
    (void)hipFreeAsync(ctl, s);
    return rc;
}

// returns 1 when this call is not one the multi-lane smoother serves
int launch_rts_ml_9(const RtsArgs &a, int layout, bool uniform, hipStream_t s)
{
    if (!uniform || a.model_t || a.n != 9 || !a.K || !a.Pp || a.T < 2) return 1;
    if (layout == FK_LAYOUT_AOS && (double)a.N * 81.0 * 8.0 >= 4294967296.0) return 1;
    {
        const int rc = launch_rts_ml_persistent(a, layout, s);
        if (rc <= 0) return rc;                                // 1 = not a call the persistent grid takes
    }
    return launch_rts_ml_chunked(a, layout, s);
}

int launch_kf_ml_9_3_var(const KfArgs &a, int layout, hipStream_t s);
static int launch_kf_ml_one(const KfArgs &a, int layout, bool outs, hipStream_t s);
static int launch_kf_ml_chunked(const KfArgs &a, int layout, bool outs, hipStream_t s);
static int launch_kf_ml_persistent(const KfArgs &a, int layout, bool outs, hipStream_t s);

// returns 1 when this call is not one the multi-lane kernel serves
int launch_kf_ml_9_3(const KfArgs &a, int layout, bool outs, int model_mode, hipStream_t s)
{
    if ((model_mode != FK_MODEL_SHARED && model_mode != FK_MODEL_PER_STEP) || a.n != 9 || a.m != 3) return 1;
    if (layout == FK_LAYOUT_AOS && (double)a.N * 81.0 * 8.0 >= 4294967296.0) return 1;      // 32-bit record offsets
    if (model_mode == FK_MODEL_PER_STEP || a.nu > 0 || a.update_first) {
        // the VAR family: all four outputs, dim_u <= 4 (FK_ML_VAR=0 sends these calls back to kf_fast / kf_kernel)
        const char *vv = getenv("FK_ML_VAR");
        if (!outs || a.nu > 4 || (vv && atoi(vv) == 0)) return 1;
        return kf_chunked_call(a, 9, 3, 2048, [layout](const KfArgs &b, hipStream_t sb) { return launch_kf_ml_9_3_var(b, layout, sb); }, s);
    }
    {
        const int rc = launch_kf_ml_persistent(a, layout, outs, s);
        if (rc <= 0) return rc;                                // 1 = not a call the persistent grid takes
    }
    return launch_kf_ml_chunked(a, layout, outs, s);
}

// The persistent grid (PERS instantiations): where the bank is more workgroups than the chip holds at once and long enough
// to cut -- G = ceil(cnt / 64) track groups x H time chunks of >= 16 steps, 2 workgroups per CU drawing tickets.  The
// ticket counter and the G completion words live in a stream-ordered scratch allocation of this call (hipMallocAsync: no
// state shared between concurrent calls; capturable).  FK_ML_PERSIST=0: off (the multi-stream tail filling below).
static int launch_kf_ml_persistent(const KfArgs &a, int layout, bool outs, hipStream_t s)
{
    const char *pv = getenv("FK_ML_PERSIST");
    if (pv && atoi(pv) == 0) return 1;
    const char *sv = getenv("FK_ML_SLAB");
    const bool slab = !(sv && atoi(sv) == 0);
    if (!outs || (layout != FK_LAYOUT_AOS && !slab)) return 1;
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                   ? prop.multiProcessorCount : 256;
    }
    const long G = (a.cnt + BLOCK / 4 - 1) / (BLOCK / 4), slots = 2L * n_cu;
    // chunks: more of them shorten the tail (ceil(G H / slots) rounds of T / H steps) but each costs a state reload and a
    // pipeline start -- three measured best at configs[2] (3.10 ms; 4: 3.15, 6: 3.22, 10: 3.29; one launch 3.39)
    long H = a.T >= 48 ? 3 : a.T / 16;
    if (const char *hv = getenv("FK_ML_PERSIST_H")) H = atol(hv);
    if (G <= slots || H < 2 || G > (1L << 24)) return 1;
    int *ctl = nullptr;
    const size_t cbytes = ((size_t)(1 + G) * sizeof(int) + 255) & ~(size_t)255, wbytes = (size_t)90 * (size_t)a.N * sizeof(double);
    if (hipMallocAsync((void **)&ctl, cbytes + wbytes, s) != hipSuccess || !ctl) { (void)hipGetLastError(); return 1; }
    if (hipMemsetAsync(ctl, 0, cbytes, s) != hipSuccess) { (void)hipFreeAsync(ctl, s); (void)hipGetLastError(); return 1; }
    KfArgs b = a;
    b.pers_ctl = ctl;
    b.pers_ws = reinterpret_cast<double *>(reinterpret_cast<char *>(ctl) + cbytes);
    b.pers_G = (int)G;
    b.pers_H = (int)H;
    const dim3 grid((unsigned)slots), block(BLOCK);
#define GOP(M)                                                                                                              \
    if (layout == FK_LAYOUT_AOS)                                                                                            \
        hipLaunchKernelGGL((kf_ml_kernel<3, 3, true, FK_ML_WAVES, false, M, LAYOUT_AOS, false, false, false, true>), grid, block, 0, s, b); \
    else hipLaunchKernelGGL((kf_ml_kernel<3, 3, true, FK_ML_WAVES, false, M, LAYOUT_SOA, false, false, true, true>), grid, block, 0, s, b)
    if (a.mask) { GOP(true); } else { GOP(false); }
#undef GOP
    const int rc = check_launch("kf_ml_kernel<pers>");
    (void)hipFreeAsync(ctl, s);
    return rc;
}

// one launch over tracks [a.i0, a.i0 + a.cnt), a.T steps from the pointers in `a`
static int launch_kf_ml_one(const KfArgs &a, int layout, bool outs, hipStream_t s)
{
    const dim3 grid((unsigned)((a.cnt + BLOCK / 4 - 1) / (BLOCK / 4))), block(BLOCK);
    // 16-byte pair stores (SOA) need the track count even (a pair never straddles a plane); FK_ML_PAIRS=0 turns them off
    const char *pv = getenv("FK_ML_PAIRS");
    const bool pairs = (a.N % 2 == 0) && (a.cnt % 2 == 0) && a.cnt >= 2 && !(pv && atoi(pv) == 0);
    // FK_ML_SLAB=0: element-major outputs as DPP pair stores / 8-byte stores (rounds 2-3) instead of the LDS slab
    const char *sv = getenv("FK_ML_SLAB");
    const bool slab = !(sv && atoi(sv) == 0);
#define GO(M, LAY)                                                                                                          \
    if (outs && slab && LAY == LAYOUT_SOA)                                                                                  \
        hipLaunchKernelGGL((kf_ml_kernel<3, 3, true, FK_ML_WAVES, false, M, LAYOUT_SOA, false, false, true>), grid, block, 0, s, a); \
    else if (outs && pairs && LAY == LAYOUT_SOA)                                                                            \
        hipLaunchKernelGGL((kf_ml_kernel<3, 3, true, FK_ML_WAVES, true, M, LAYOUT_SOA>), grid, block, 0, s, a);             \
    else if (outs) hipLaunchKernelGGL((kf_ml_kernel<3, 3, true, FK_ML_WAVES, false, M, LAY>), grid, block, 0, s, a);        \
    else hipLaunchKernelGGL((kf_ml_kernel<3, 3, false, FK_ML_WAVES, false, M, LAY>), grid, block, 0, s, a)
    if (layout == FK_LAYOUT_SOA) {
        if (a.mask) { GO(true, LAYOUT_SOA); } else { GO(false, LAYOUT_SOA); }
    } else {
        if (a.mask) { GO(true, LAYOUT_AOS); } else { GO(false, LAYOUT_AOS); }
    }
#undef GO
    return check_launch("kf_ml_kernel");
}

// Tail filling.  The step is bound by arithmetic and latency, every wave runs the same T steps, so a bank of W waves on S
// wave slots takes ceil(W / S) rounds: BASELINE config 3 (1e5 tracks = 6250 waves on 2048 slots) pays 4 rounds for 3.05
// rounds of work.  A chunked call cuts the bank into G track groups (multiples of 64 tracks) and the T steps into H time
// chunks and launches the G x H pieces on G streams -- group g's chunks in order on stream g, the state handed from chunk
// to chunk through x / P in place (kernel boundaries of one stream: no protocol), different groups concurrently: while one
// group's piece tails off, the other groups' pieces fill the slots, and what is left at the very end is the tail of a
// piece 1/H as long.  Same arithmetic per track: results are bit-identical to the single launch.  The helper streams
// fork from and join the caller's stream with events (capturable); they and the events are created once.
static int launch_rts_ml_chunked(const RtsArgs &a, int layout, hipStream_t s)
{
    return rts_chunked_call(a, 9, 2048, [layout](const RtsArgs &b, hipStream_t sb) { return launch_rts_ml_one(b, layout, sb); }, s);
}

static int launch_kf_ml_chunked(const KfArgs &a, int layout, bool outs, hipStream_t s)
{
    if (!outs) return launch_kf_ml_one(a, layout, outs, s);
    return kf_chunked_call(a, 9, 3, 2048, [layout, outs](const KfArgs &b, hipStream_t sb) { return launch_kf_ml_one(b, layout, outs, sb); }, s);
}
#endif   // FK_ML_PART != 2

#if FK_ML_PART != 1
int launch_kf_ml_9_3_var(const KfArgs &a, int layout, hipStream_t s)
{
    const dim3 grid((unsigned)((a.cnt + BLOCK / 4 - 1) / (BLOCK / 4))), block(BLOCK);
    const char *pv = getenv("FK_ML_PAIRS");
    const bool pairs = (a.N % 2 == 0) && (a.cnt % 2 == 0) && a.cnt >= 2 && !(pv && atoi(pv) == 0);
#define GOV(UFV)                                                                                                            \
    if (layout == FK_LAYOUT_AOS)                                                                                            \
        hipLaunchKernelGGL((kf_ml_kernel<3, 3, true, FK_ML_WAVES, false, true, LAYOUT_AOS, true, UFV>), grid, block, 0, s, a); \
    else if (pairs)                                                                                                         \
        hipLaunchKernelGGL((kf_ml_kernel<3, 3, true, FK_ML_WAVES, true, true, LAYOUT_SOA, true, UFV>), grid, block, 0, s, a);  \
    else hipLaunchKernelGGL((kf_ml_kernel<3, 3, true, FK_ML_WAVES, false, true, LAYOUT_SOA, true, UFV>), grid, block, 0, s, a)
    if (a.update_first) { GOV(true); } else { GOV(false); }
#undef GOV
    return check_launch("kf_ml_kernel<var>");
}
#endif   // FK_ML_PART != 1

#undef HX
#undef FK_ML_PRIOR_STORES

}  // namespace fk
