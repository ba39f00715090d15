// ukf_kernels.hip -- fused linear-model unscented Kalman filter for gfx950 (MI355X).
//
//   fk_ukf_linear_batch_f64 <- UnscentedKalmanFilter.batch_filter (filterpy/kalman/UKF.py:524-632) with
//                              fx(x, dt) = F x, hx(x) = H x: the whole predict (UKF.py:400-411) / update
//                              (:462-481) loop stays in registers over the time steps.
//   fk_ukf_linear_rts_f64   <- UnscentedKalmanFilter.rts_smoother (UKF.py:634-739), likewise.
// FK_UKF_PART: the Makefile compiles this file eight times (parallel build) -- 1: forward kernels dim_x <= 6 + the forward
// entry point, 2: forward dim_x 7..9, 3: smoother dim_x <= 6 + the smoother's entry point, 4: smoother dim_x 7..9; 5..8: the
// same four sets of kernels with the sums regrouped over the +- pairs of sigma points (PAIRED, fk_ukf.hpp: the V4 steps).
#include <stdlib.h>

#ifndef FK_UKF_PART
#define FK_UKF_PART 0
#endif
#define FK_UKF_HAS(p) (FK_UKF_PART == 0 || FK_UKF_PART == (p))
// (parts 91 / 92: the forward / smoother kernel templates alone, for one-off instantiations: tools/ukf_one_kernel.py)
#define FK_UKF_FWD (FK_UKF_HAS(1) || FK_UKF_HAS(2) || FK_UKF_HAS(5) || FK_UKF_HAS(6) || FK_UKF_PART == 91)
#define FK_UKF_RTS (FK_UKF_HAS(3) || FK_UKF_HAS(4) || FK_UKF_HAS(7) || FK_UKF_HAS(8) || FK_UKF_PART == 92)

#include "../../include/filterhip.h"
#include "fk_chunks.hpp"
#include "fk_device.hpp"
#include "fk_kernel_args.hpp"
#include "fk_math_sym.hpp"
#include "fk_ukf.hpp"

namespace fk {


#if FK_UKF_FWD
// --------------------------------------------------- fused linear-model UKF --
// Per step (UKF.py:400-411, 462-481) with fx(x) = F x, hx(x) = H x:
//   L  = chol(scale P);  sigma_i = x, x +- L[:,k]          (sigma_points.py:167-175)
//   sf_i = F sigma_i ;  (x,P) = UT(sf, Wm, Wc, Q)
//   L  = chol(scale P);  sf_i = sigma_i(x,P)  (regenerated, UKF.py:407)
//   sh_i = H sf_i ; (zp,S) = UT(sh, Wm, Wc, R) ; Pxz = sum Wc_i (sf_i-x)(sh_i-zp)'
//   K = Pxz S^-1 ; x += K (z-zp) ; P -= K (S K')
// The arithmetic of a step is ukf_linear_step_v3 (fk_ukf.hpp, held against the oracle on the host by
// tests/test_hostcheck_ukf.py): packed triangles, the images of the sigma points formed from the image of the factor.
// Round 2's point-by-point step and the scalar-operand model are gone (A/B of the last lease that had them:
// profiles/r03/ukf_v3_ab.jsonl -- (6,3) 1.92 -> 1.27 ms SOA, 2.45 -> 1.33 ms NumPy order).
//
// The kernel around it, round 3 (the same rules as kf_fast / kf_ml):
//   * EXACT instantiations (n == NX, m == NZ) have no run-time element guards, so a step's stores are straight-line
//     code and the compiler can COUNT them in its s_waitcnt; the padded instantiations serve every smaller size.
//   * z[t] is requested at the HEAD OF THE UPDATE HALF of step t (behind the predict half: no registers are held across
//     it) and consumed at its end; the mask byte of step t+1 is requested at the top of step t (clamped index, no
//     branch) and landed at the end of its arithmetic, in front of step t's stores.  vmcnt retires in order,
//     so waiting for a load also waits for every store issued before it -- here the stores of step t-1, which have had
//     a predict half to drain.  Before, the loop header was `load z; s_waitcnt vmcnt(0)`: every step paid an HBM round
//     trip plus the drain of its predecessor's 42 stores with nothing to overlap.
//   * the prologue loads (x0, P0, z[0]) are landed before the loop: left pending, the loop header's wait (which must
//     cover the path from the prologue too) is a vmcnt(0) in every iteration.
//   * no store in the loop is predicated: the lanes past the bank's last track duplicate the block's last track
//     (same values to the same addresses); the wave-cooperative NumPy-order stores drop rows past the last track by
//     the descriptor's range check.  The in-place store of the final state IS predicated and sits behind a workgroup
//     barrier (a duplicate must have consumed x0 / P0 before its owner overwrites them).
//   * PAIRED (round 4): the step is ukf_linear_step_v4 -- the sums regrouped over the +- pairs of sigma points, for weights
//     equal within every pair (the caller's FK_UKF_FLAG_PAIR_WEIGHTS; the prologue checks it and reports
//     FK_STATUS_BAD_WEIGHTS on every track otherwise).  The pair table sits behind the weights in LDS.
//   * SP (round 4; element-major, exact even dims up to 6): the per-step outputs leave as 16-byte stores of two element rows
//     each (wave_store_soa_pairs: 21 instead of 42 vector-memory operations per step at (6,3)).  Compile-time, not a branch:
//     two store sequences of different lengths behind a run-time test make the compiler's s_waitcnt for z a vmcnt(0).
//     Full workgroups only -- the launcher hands the last partial workgroup to the plain instantiation.
//   * PERS (round 6): a persistent grid drawing TICKETS, like kf_ml's (kf_ml.hip).  Every wave runs the same T steps, and BASELINE
//     configs[3] is 391 workgroups for 512 slots (two per CU): 135 CUs carry two workgroups for the whole call, 121 carry one
//     and idle half their issue slots -- the launch takes what a two-workgroup CU takes (round 5: VALU executing 32 % of wave
//     residency, 49 % waiting for issue).  Here the call is cut into G = ceil(N / 256) track groups x H time chunks and 2 x CUs
//     resident workgroups draw tickets chunk-major; chunk h of a group waits for chunk h - 1 (a ticket at least G draws older:
//     held by a workgroup that runs or is done) through a completion word and picks the state up from an element-major
//     hand-over block ([NX + NX (NX + 1) / 2][N], agent-scope accesses: no fence, no L2 write-back).  Same arithmetic per
//     track: bit-identical to the single launch (tests/test_gpu_ukf.py).
template <int NX, int NZ, int LAYOUT, bool EXACT, bool PAIRED, bool SP = false, bool PERS = false>
__global__ void __launch_bounds__(BLOCK, (NX <= 2 ? 4 : NX <= 6 ? 2 : 1))
ukf_linear_kernel(const UkfArgs a_in, const double *__restrict__ pF, const double *__restrict__ pH,
                  const double *__restrict__ pQ, const double *__restrict__ pR,
                  const double *__restrict__ pWm, const double *__restrict__ pWc,
                  const double *__restrict__ pz_in, const uint8_t *__restrict__ pmask_in,
                  int *pers_ctl = nullptr, double *pers_ws = nullptr, const int pers_G = 1, const int pers_H = 1)
{
    static_assert(!PERS || EXACT, "PERS: the exact instantiations");
    UkfArgs a = a_in;
    const double *pz = pz_in;
    const uint8_t *pmask = pmask_in;
    constexpr int KS = 2 * NX + 1;
    constexpr int PL = NX * (NX + 1) / 2;
    using SharedModel = LdsModel<NX, NZ>;
    __shared__ double s_model[SharedModel::SIZE + 2 * KS + 2 + NX];
    // NumPy order at the exact dims: the per-step outputs leave through a wave-private LDS tile, 1 KiB contiguous per store
    // instruction (wave_store_aos_flat, fk_device.hpp) -- a lane-per-record store touches 64 lines per instruction.
    // (dim_x 9: the tile of an 81-double record does not fit next to four waves.)
    constexpr bool COOP = EXACT && LAYOUT == LAYOUT_AOS && NX <= 8 && NX % 2 == 0;
    // element-major at the exact even dims up to 6: the per-step outputs leave as 16-byte stores of two element rows each
    // (wave_store_soa_pairs: 21 instead of 42 vector-memory operations per step at (6,3)) where the wave is full and the rows
    // are 16-byte aligned; other waves take the 8-byte stores
    constexpr bool PAIRS = SP;
    static_assert(!SP || (EXACT && LAYOUT == LAYOUT_SOA && NX <= 6 && NX % 2 == 0), "SP: element-major, exact even dims up to 6");
    constexpr int TILE = 64 * NX * NX;
    __shared__ double s_tile[(COOP || PAIRS) ? (BLOCK / 64) * TILE : 1];
    const long N = a_in.N;
    const int n = EXACT ? NX : a_in.n, m = EXACT ? NZ : a_in.m;
    const int ks = 2 * n + 1;
    double *tile = s_tile + ((COOP || PAIRS) ? (threadIdx.x >> 6) * TILE : 0);
    const unsigned lane = threadIdx.x & 63u, wave_row0 = (threadIdx.x >> 6) * 64u;

    lds_fill<NX, NX>(s_model + SharedModel::OFF_F, pF, n, n, 1.0, threadIdx.x);
    lds_fill<NX, NX>(s_model + SharedModel::OFF_Q, pQ, n, n, 1.0, threadIdx.x);   // padded block of P stays I
    lds_fill<NZ, NX>(s_model + SharedModel::OFF_H, pH, m, n, 0.0, threadIdx.x);
    lds_fill<NZ, NZ>(s_model + SharedModel::OFF_R, pR, m, m, 1.0, threadIdx.x);
    // weights, re-indexed from the runtime point set (0, 1..n, n+1..2n) to the padded one
    // (0, 1..NX, NX+1..2NX); padded points get weight 0
    for (unsigned q = threadIdx.x; q < (unsigned)(2 * KS); q += BLOCK) {
        const int which = q / KS, i = q % KS;
        int src = -1;
        if (i == 0) src = 0;
        else if (i <= NX) { if (i <= n) src = i; }
        else { if (i - NX <= n) src = n + (i - NX); }
        const double *W = which ? pWc : pWm;
        s_model[SharedModel::SIZE + q] = (src >= 0 && src < ks) ? W[src] : 0.0;
    }
    __syncthreads();
    int st = 0;
    if constexpr (PAIRED) {
        if (threadIdx.x == 0) make_pair_table<NX>(s_model + SharedModel::SIZE, s_model + SharedModel::SIZE + KS, s_model + SharedModel::SIZE + 2 * KS);
        if (!pair_weights_symmetric<NX>(s_model + SharedModel::SIZE, s_model + SharedModel::SIZE + KS)) st |= ST_BAD_WEIGHTS;
        __syncthreads();
    }
    // a view of the LDS model the optimiser cannot relate to the previous one: keeps it from hoisting the
    // broadcast row reads of all 2n+1 unrolled points to the top (they are cheap to repeat, dear to hold)
    // (an offset is made opaque, not the pointer: that keeps the LDS address space)
    struct View { SharedModel sm; const double *Wm, *Wc, *Wp; };
    auto fresh = [&](double after = 0.0) {
        int off = 0;
        asm volatile("" : "+v"(off) : "v"(after));
        const double *mb = s_model + off;
        return View{SharedModel{mb}, mb + SharedModel::SIZE, mb + SharedModel::SIZE + KS, mb + SharedModel::SIZE + 2 * KS};
    };

    [[maybe_unused]] int pers_g = 0, pers_h = 0;
    __shared__ int s_task;
    const int st_prologue = st;
    for (;;) {                                                   // PERS: one trip per ticket; otherwise exactly one trip
    unsigned bid = blockIdx.x;
    if constexpr (PERS) {
        __syncthreads();                                         // the previous ticket's LDS traffic is over, s_task is free
        if (threadIdx.x == 0) s_task = atomicAdd(pers_ctl, 1);
        __syncthreads();
        const int task = __builtin_amdgcn_readfirstlane(s_task);
        if (task >= pers_G * pers_H) break;
        pers_h = task / pers_G;
        pers_g = task - pers_h * pers_G;
        bid = (unsigned)pers_g;
        const long t0 = a_in.T * pers_h / pers_H, t1 = a_in.T * (pers_h + 1) / pers_H;
        a = a_in;
        a.T = t1 - t0;
        pz = pz_in + t0 * N * m;
        pmask = pmask_in ? pmask_in + t0 * N : nullptr;
        a.means = a_in.means ? a_in.means + t0 * N * NX : nullptr;
        a.covs = a_in.covs ? a_in.covs + t0 * N * (NX * NX) : nullptr;
        a.status_or = t0 > 0 ? 1 : a_in.status_or;
        st = st_prologue;
        if (pers_h > 0) {
            if (threadIdx.x == 0) {
                while (__hip_atomic_load(&pers_ctl[1 + pers_g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < pers_h) __builtin_amdgcn_s_sleep(8);
            }
            __syncthreads();
        }
    }
    const long blk0 = a.i0 + (long)bid * BLOCK;
    const long left = a.i0 + a.cnt - blk0;
    const unsigned last_row = (unsigned)(left < BLOCK ? left : BLOCK) - 1u;
    const bool live = threadIdx.x <= last_row;
    const Lane ln{blk0, live ? threadIdx.x : last_row, N};      // lanes past the last track duplicate it
    double x[NX], P[PL];
    if (PERS && pers_h > 0) {
        // a later chunk of the group: the state the previous chunk left in the hand-over block (element-major: 512 contiguous
        // bytes per wave instruction; agent-scope loads -- coherent across the XCDs' L2s without a fence)
        const double *hx = pers_ws + (blk0 + ln.tid);
        FK_UNROLL for (int c = 0; c < NX; ++c) x[c] = __hip_atomic_load(hx + (long)c * N, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        FK_UNROLL for (int e = 0; e < PL; ++e) P[e] = __hip_atomic_load(hx + (long)(NX + e) * N, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
    load_rec<NX, 1, LAYOUT, EXACT>(x, a.x, ln, n, 1, 0.0);
    {
        const RecView<LAYOUT> pv(a.P, ln, n * n);
        FK_UNROLL for (int i = 0; i < NX; ++i)
            FK_UNROLL for (int j = 0; j < NX; ++j)
                if (j >= i) P[sym_idx<NX>(i, j)] = (EXACT || (i < n && j < n)) ? pv.load(i * n + j) : (i == j ? 1.0 : 0.0);
    }
    }
    // without a mask the byte comes from a valid dummy address (the measurements) and is selected away: no branch.  Read
    // through a descriptor (uniform base advanced per step + the lane's 32-bit offset): a per-lane 64-bit pointer that
    // advances by N per step is a loop-carried VGPR pair the allocator spills, and its reload at the loop's end is a
    // vmcnt(0) behind the step's stores.
    const uint8_t *mk0 = (pmask ? pmask : reinterpret_cast<const uint8_t *>(pz)) + blk0;
    unsigned hc = __builtin_amdgcn_raw_buffer_load_b8(make_rsrc(mk0), ln.tid, 0, 0);
    // z is carried from step to step: z[t+1] is requested at the end of step t IN FRONT of that step's stores (round 4).
    // vmcnt retires in order: requested at the head of the update half (round 3), the load queued behind the 42 stores of
    // the step before, and 1563 waves storing 21 KB each at the same moment take 5-7 us to drain -- the waves waited there
    // for 41 % of their residency (profiles/r04/ukf_sq_counters.jsonl).  In front of the stores it is waited for with a
    // counted vmcnt(#stores) a whole predict half later; the stores drain under the next step's arithmetic.  For the count
    // to be a constant the stores of a step are unconditional: an output that was not asked for gets a descriptor of zero
    // records (issued and dropped, RecView).
    double zc[NZ];
    load_rec<NZ, 1, LAYOUT, EXACT>(zc, pz, ln, m, 1, 0.0);
    // landed here (see above)
    FK_UNROLL for (int c = 0; c < NX; ++c) asm volatile("" ::"v"(x[c]));
    FK_UNROLL for (int e = 0; e < PL; ++e) asm volatile("" ::"v"(P[e]));
    FK_UNROLL for (int c = 0; c < NZ; ++c) asm volatile("" ::"v"(zc[c]));
    asm volatile("" ::"v"(hc));
    const bool st_m = a.means != nullptr, st_c = a.covs != nullptr;

    _Pragma("nounroll") for (long t = 0; t < a.T; ++t) {
        const bool has_z = pmask ? hc != 0u : true;
        auto load_z = [&](double (&z)[NZ]) { FK_UNROLL for (int c = 0; c < NZ; ++c) z[c] = zc[c]; };

        if constexpr (PAIRED) st |= ukf_linear_step_v4<NX, NZ>(x, P, load_z, has_z, a.scale, fresh);
        else st |= ukf_linear_step_v3<NX, NZ>(x, P, load_z, has_z, a.scale, fresh);
        FK_STAGE();
        // step t+1's measurement and mask byte (clamped index, no branch), in front of this step's stores
        {
            long tn = t + 1 < a.T ? t + 1 : t;
            asm volatile("" : "+s"(tn));      // opaque: keeps the compiler from re-deriving these loads one iteration later
            hc = __builtin_amdgcn_raw_buffer_load_b8(make_rsrc(mk0 + tn * N), ln.tid, 0, 0);
            load_rec<NZ, 1, LAYOUT, EXACT>(zc, pz + tn * N * m, ln, m, 1, 0.0);
        }
        FK_STAGE();
        if constexpr (COOP) {
            wave_store_aos_flat<NX>(x, a.means + t * N * NX + blk0 * NX, wave_row0, tile, lane, last_row, st_m);
            double Pf[NX * NX];
            FK_UNROLL for (int i = 0; i < NX; ++i)
                FK_UNROLL for (int j = 0; j < NX; ++j) Pf[i * NX + j] = P[sym_idx<NX>(i, j)];
            wave_store_aos_flat<NX * NX>(Pf, a.covs + t * N * (NX * NX) + blk0 * (NX * NX), wave_row0, tile, lane, last_row, st_c);
        } else {
            double Pf[NX * NX];
            FK_UNROLL for (int i = 0; i < NX; ++i)
                FK_UNROLL for (int j = 0; j < NX; ++j) Pf[i * NX + j] = P[sym_idx<NX>(i, j)];
            if constexpr (PAIRS) {
                // (w0 scalar: wave_index() -- no waterfall loop around the pair stores; a wave of the last workgroup may hold
                //  fewer than 64 tracks, or none: its missing pairs are dropped by offset, see wave_store_soa_pairs)
                const long w0 = blk0 + (long)wave_index() * 64, leftw = a.i0 + a.cnt - w0;
                const unsigned validw = leftw >= 64 ? 64u : (leftw > 0 ? (unsigned)leftw : 0u);
                wave_store_soa_pairs<NX>(x, a.means + t * N * NX + w0, (unsigned)N * 8u, tile, lane, st_m, validw);
                wave_store_soa_pairs<NX * NX>(Pf, a.covs + t * N * (NX * NX) + w0, (unsigned)N * 8u, tile, lane, st_c, validw);
            } else {
                store_rec<NX, 1, LAYOUT, EXACT>(x, a.means + t * N * n, ln, n, 1, st_m);
                store_rec<NX, NX, LAYOUT, EXACT>(Pf, a.covs + t * N * n * n, ln, n, n, st_c);
            }
        }
    }
    __syncthreads();
    if (live) {
        if (PERS && pers_h + 1 < pers_H) {
            // not the group's last chunk: the state goes to the hand-over block (agent-scope stores), x / P stay untouched
            double *hx = pers_ws + (blk0 + ln.tid);
            FK_UNROLL for (int c = 0; c < NX; ++c) __hip_atomic_store(hx + (long)c * N, x[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            FK_UNROLL for (int e = 0; e < PL; ++e) __hip_atomic_store(hx + (long)(NX + e) * N, P[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
        store_rec<NX, 1, LAYOUT, EXACT>(x, a.x, ln, n, 1);
        double Pf[NX * NX];
        FK_UNROLL for (int i = 0; i < NX; ++i)
            FK_UNROLL for (int j = 0; j < NX; ++j) Pf[i * NX + j] = P[sym_idx<NX>(i, j)];
        store_rec<NX, NX, LAYOUT, EXACT>(Pf, a.P, ln, n, n);
        }
        if (a.status) {
            if (!all_finite<NX>(x) || !all_finite<PL>(P)) st |= ST_NONFINITE;
            if constexpr (PERS) {
                const int old = a.status_or ? __hip_atomic_load(&a.status[ln.blk0 + ln.tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
                __hip_atomic_store(&a.status[ln.blk0 + ln.tid], old | st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                a.status[ln.blk0 + ln.tid] = a.status_or ? (a.status[ln.blk0 + ln.tid] | st) : st;
            }
        }
    }
    if constexpr (PERS) {
        // the chunk's state (and status) is in place once every wave's vmcnt has drained -- a workgroup barrier does not wait
        // for VMEM (kf_ml.hip) --, then the barrier, then the chunk is published
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(&pers_ctl[1 + pers_g], pers_h + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        break;
    }
    }      // tickets
}

#endif

// launchers of the instantiations this part holds (declared in every part)
int ukf_fwd_launch_small(const UkfArgs &a, int layout, bool exact, hipStream_t s);    // classes (2,2), (4,2), (6,3)
int ukf_fwd_launch_big(const UkfArgs &a, int layout, bool exact, hipStream_t s);      // classes (8,4), (9,3), (9,4)
int ukf_rts_launch_small(const UkfRtsArgs &a, const double *F, const double *Q, const double *Wm, const double *Wc, int layout, bool exact, hipStream_t s);   // 2, 4, 6
int ukf_rts_launch_big(const UkfRtsArgs &a, const double *F, const double *Q, const double *Wm, const double *Wc, int layout, bool exact, hipStream_t s);     // 8, 9
// ... and of their PAIRED twins (parts 5..8)
int ukf_fwd_launch_small_paired(const UkfArgs &a, int layout, bool exact, hipStream_t s);
int ukf_fwd_launch_big_paired(const UkfArgs &a, int layout, bool exact, hipStream_t s);
int ukf_rts_launch_small_paired(const UkfRtsArgs &a, const double *F, const double *Q, const double *Wm, const double *Wc, int layout, bool exact, hipStream_t s);
int ukf_rts_launch_big_paired(const UkfRtsArgs &a, const double *F, const double *Q, const double *Wm, const double *Wc, int layout, bool exact, hipStream_t s);

#if FK_UKF_FWD
#define FK_UKF_GO(NXV, NZV)                                                                                      \
    do {                                                                                                         \
        const dim3 grid((unsigned)((a.cnt + BLOCK - 1) / BLOCK)), block(BLOCK);                                  \
        const bool ex = exact && a.n == NXV && a.m == NZV;                                                       \
        if (layout == FK_LAYOUT_SOA) {                                                                           \
            if (ex && ukf_sp_ok<NXV>(a)) ukf_launch_sp<NXV, NZV, PV>(a, s);                                         \
            else if (ex) hipLaunchKernelGGL((ukf_linear_kernel<NXV, NZV, LAYOUT_SOA, true, PV>), grid, block, 0, s, a, a.F, a.H, a.Q, a.R, a.Wm, a.Wc, a.z, a.mask); \
            else hipLaunchKernelGGL((ukf_linear_kernel<NXV, NZV, LAYOUT_SOA, false, PV>), grid, block, 0, s, a, a.F, a.H, a.Q, a.R, a.Wm, a.Wc, a.z, a.mask);  \
        } else {                                                                                                 \
            if (ex) hipLaunchKernelGGL((ukf_linear_kernel<NXV, NZV, LAYOUT_AOS, true, PV>), grid, block, 0, s, a, a.F, a.H, a.Q, a.R, a.Wm, a.Wc, a.z, a.mask); \
            else hipLaunchKernelGGL((ukf_linear_kernel<NXV, NZV, LAYOUT_AOS, false, PV>), grid, block, 0, s, a, a.F, a.H, a.Q, a.R, a.Wm, a.Wc, a.z, a.mask);  \
        }                                                                                                        \
    } while (0)
// SP instantiations: even dims up to 6, element-major, N and the piece's track count even, 16-byte aligned histories (or none)
template <int NXV>
static bool ukf_sp_ok(const UkfArgs &a)
{
    if constexpr (NXV > 6 || NXV % 2 != 0) return false;
    return a.soa_pairs && (a.N & 1) == 0 && (a.cnt & 1) == 0 && (a.i0 & 63) == 0 &&
           ((reinterpret_cast<uintptr_t>(a.means) | reinterpret_cast<uintptr_t>(a.covs)) & 15u) == 0;
}
template <int NXV, int NZV, bool PV>
static void ukf_launch_sp(const UkfArgs &a, hipStream_t s)
{
    if constexpr (NXV <= 6 && NXV % 2 == 0) {
        // every workgroup on the SP kernel, the last partial one included (its missing track pairs are dropped by offset).
        // (Round 4 first launched the partial workgroup on the plain kernel behind the full ones: a lone workgroup stepping
        //  through T epochs is latency-bound -- 277 us behind the 780 us of configs[3]'s 390 full workgroups, also from a
        //  helper stream: 1.00-1.08 ms end to end where the 8-byte stores took 0.87.)
        hipLaunchKernelGGL((ukf_linear_kernel<NXV, NZV, LAYOUT_SOA, true, PV, true>), dim3((unsigned)((a.cnt + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s, a, a.F, a.H, a.Q, a.R, a.Wm, a.Wc, a.z, a.mask);
    }
}

// The persistent grid (PERS instantiation; the exact (6, 3) class with pair weights: BASELINE configs[3]): a whole-bank call of
// more workgroups than CUs and at least 32 steps is cut into G track groups x H time chunks, 2 x CUs resident workgroups draw
// tickets.  H: the count in [4, T / 6] whose last round of tickets is fullest (ceil(G H / slots) rounds of T / H steps), the
// larger chunk on a draw.  The ticket counter, the completion words and the hand-over block live in a stream-ordered scratch
// allocation of this call (hipMallocAsync: no state shared between concurrent calls).  FK_UKF_PERSIST=0: off;
// FK_UKF_PERSIST_H: the chunk count.  Returns 1 where the call is not one it takes.
template <bool PV>
static int ukf_fwd_persistent_6_3(const UkfArgs &a, int layout, hipStream_t s)
{
    if constexpr (!PV) return 1;
    else {
    // OFF unless FK_UKF_PERSIST=1.  Measured (profiles/r06/ukf_persist.txt, 1e5 x 100, same lease): one launch 0.81-0.85 ms, tickets
    // with H = 5 / 8 / 10 / 13 chunks 0.87 / 0.91 / 0.94 / 0.96 -- every chunk costs ~10 us (ticket, completion word, state reload,
    // drain) and the balance buys NOTHING: a CU that carries one workgroup takes as long as one that carries two.  The launch is
    // bound by the length of ONE wave's dependent instruction chain (100 steps x 1193 instructions at ~6.9 ns each), not by issue
    // slots: what would help is a shorter chain per step, not a better spread of the waves.
    const char *pv = getenv("FK_UKF_PERSIST");
    if (!(pv && atoi(pv) == 1)) return 1;
    if (a.i0 != 0 || a.cnt != a.N || a.T < 32) return 1;
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                   ? prop.multiProcessorCount : 256;
    }
    const long G = (a.cnt + BLOCK - 1) / BLOCK, slots = 2L * n_cu;
    if (G <= n_cu || G > (1L << 24)) return 1;
    long H = 0;
    double best = 0.0;
    for (long h = 4; h <= a.T / 6; ++h) {
        const double rounds = (double)(G * h) / (double)slots, fill = rounds / (double)((G * h + slots - 1) / slots);
        if (fill > best + 0.02) { best = fill; H = h; }
    }
    if (const char *hv = getenv("FK_UKF_PERSIST_H")) H = atol(hv);
    if (H < 2 || H > a.T) return 1;
    const bool sp = layout == FK_LAYOUT_SOA && ukf_sp_ok<6>(a);
    int *ctl = nullptr;
    const size_t cbytes = ((size_t)(1 + G) * sizeof(int) + 255) & ~(size_t)255, wbytes = (size_t)(6 + 21) * (size_t)a.N * sizeof(double);
    if (hipMallocAsync((void **)&ctl, cbytes + wbytes, s) != hipSuccess || !ctl) { (void)hipGetLastError(); return 1; }
    if (hipMemsetAsync(ctl, 0, cbytes, s) != hipSuccess) { (void)hipFreeAsync(ctl, s); (void)hipGetLastError(); return 1; }
    double *ws = reinterpret_cast<double *>(reinterpret_cast<char *>(ctl) + cbytes);
    const dim3 grid((unsigned)(G * H < slots ? G * H : slots)), block(BLOCK);
    if (layout == FK_LAYOUT_SOA) {
        if (sp) hipLaunchKernelGGL((ukf_linear_kernel<6, 3, LAYOUT_SOA, true, true, true, true>), grid, block, 0, s, a, a.F, a.H, a.Q, a.R, a.Wm, a.Wc, a.z, a.mask, ctl, ws, (int)G, (int)H);
        else hipLaunchKernelGGL((ukf_linear_kernel<6, 3, LAYOUT_SOA, true, true, false, true>), grid, block, 0, s, a, a.F, a.H, a.Q, a.R, a.Wm, a.Wc, a.z, a.mask, ctl, ws, (int)G, (int)H);
    } else {
        hipLaunchKernelGGL((ukf_linear_kernel<6, 3, LAYOUT_AOS, true, true, false, true>), grid, block, 0, s, a, a.F, a.H, a.Q, a.R, a.Wm, a.Wc, a.z, a.mask, ctl, ws, (int)G, (int)H);
    }
    const int rc = check_launch("ukf_linear_kernel<pers>");
    (void)hipFreeAsync(ctl, s);
    return rc;
    }
}

template <bool PV>
static int ukf_fwd_small_t(const UkfArgs &a, int layout, bool exact, hipStream_t s)
{
    if (PV && exact && a.n == 6 && a.m == 3) {
        const int rc = ukf_fwd_persistent_6_3<PV>(a, layout, s);
        if (rc <= 0) return rc;                                // 1 = not a call the persistent grid takes
    }
    if (a.n <= 2 && a.m <= 2) FK_UKF_GO(2, 2);
    else if (a.n <= 4 && a.m <= 2) FK_UKF_GO(4, 2);
    else FK_UKF_GO(6, 3);
    return check_launch("ukf_linear_kernel");
}
template <bool PV>
static int ukf_fwd_big_t(const UkfArgs &a, int layout, bool exact, hipStream_t s)
{
    if (a.n <= 8) FK_UKF_GO(8, 4);
    else if (a.m <= 3) FK_UKF_GO(9, 3);
    else FK_UKF_GO(9, 4);
    return check_launch("ukf_linear_kernel");
}
#if FK_UKF_HAS(1)
int ukf_fwd_launch_small(const UkfArgs &a, int layout, bool exact, hipStream_t s) { return ukf_fwd_small_t<false>(a, layout, exact, s); }
#endif
#if FK_UKF_HAS(2)
int ukf_fwd_launch_big(const UkfArgs &a, int layout, bool exact, hipStream_t s) { return ukf_fwd_big_t<false>(a, layout, exact, s); }
#endif
#if FK_UKF_HAS(5)
int ukf_fwd_launch_small_paired(const UkfArgs &a, int layout, bool exact, hipStream_t s) { return ukf_fwd_small_t<true>(a, layout, exact, s); }
#endif
#if FK_UKF_HAS(6)
int ukf_fwd_launch_big_paired(const UkfArgs &a, int layout, bool exact, hipStream_t s) { return ukf_fwd_big_t<true>(a, layout, exact, s); }
#endif
#undef FK_UKF_GO
#endif

#if FK_UKF_RTS
// ------------------------------------------------ fused linear-model UKF smoother --
//   fk_ukf_linear_rts_f64 <- UnscentedKalmanFilter.rts_smoother (filterpy/kalman/UKF.py:634-739) with fx(x, dt) = F x:
// the whole backward loop in one launch, one track per lane, the smoothed step k+1 carried in registers
// (fk_ukf.hpp, ukf_linear_rts_step).  Reads Xs[k], Ps[k]; writes xs[k], ps[k], Ks[k]: 8 (2n + 3n^2) bytes per
// track-step (1008 at n = 6).  Before: four kernel launches and nine host<->device copies per step.

template <int NX>
struct UpperTriangle {
    int flat[NX * (NX + 1) / 2];
    constexpr UpperTriangle() : flat{}
    {
        int k = 0;
        for (int i = 0; i < NX; ++i)
            for (int j = i; j < NX; ++j) flat[k++] = i * NX + j;
    }
};

// One wave per SIMD from dim_x = 5 (the gain's second sweep accumulates Pb and the full n x n Pxb side by side next to L, F L,
// x and xb).  Round 3, the kernel around the step (the forward kernel's rules):
//   * EXACT instantiations: straight-line loads and stores;
//   * the filtered state of step k is still requested at the top of step k (an HBM round trip behind the previous step's
//     stores, exposed at one wave per SIMD): a request one step ahead needs 27 doubles (84 registers of raw quads on the
//     cooperative path) that stay untouched for a whole step, and at 256 + 200 registers the allocator moves exactly those
//     to AGPRs or scratch the moment they are loaded -- which is a wait for the load (tried; an LDS-DMA fetch that
//     holds no register is the open item);
//   * NumPy order (COOP: exact, even dim_x <= 8): records move between registers and HBM through a wave-private LDS tile,
//     1 KiB contiguous per instruction, in both directions (a lane-per-record access of a 36-double record touches 64
//     lines per instruction: the n = 6 backward pass took 5.6 ms in NumPy order against 2.0 ms element-major);
//   * lanes past the last track duplicate it (or, on the cooperative path, compute on zeros that the descriptors never
//     store); no store is predicated.
//   * PAIRED (round 4): the gain is ukf_linear_rts_gain_v4 (sums regrouped over the +- pairs; see the forward kernel).
template <int NX, int LAYOUT, bool EXACT, bool PAIRED, bool DMA = false>
__global__ void __launch_bounds__(BLOCK, (NX <= 2 ? 4 : NX <= 4 ? 2 : 1))
ukf_linear_rts_kernel(const UkfRtsArgs a, const double *__restrict__ pF, const double *__restrict__ pQ,
                      const double *__restrict__ pWm, const double *__restrict__ pWc)
{
    constexpr int KS = 2 * NX + 1;
    constexpr int PL = NX * (NX + 1) / 2;
    constexpr int NN = NX * NX;
    constexpr bool COOP = EXACT && LAYOUT == LAYOUT_AOS && NX % 2 == 0 && NX <= 8;
    static_assert(!DMA || (EXACT && NX % 2 == 0 && NX <= 6), "LDS-DMA fetch: exact classes 2, 4, 6");
    // DMA: the filtered state of step k-1 travels HBM -> LDS while step k computes (lds_dma16, fk_device.hpp).  Its LDS image:
    // NumPy order -- the wave's slab of Ps in memory order in the cooperative tile (which the store path reuses afterwards),
    // the slab of Xs next to it; element-major -- [pair of elements][64 tracks], the upper triangle of P only.
    constexpr int DPAIRS = (PL + 1) / 2;                                       // element-major: pairs of P's upper triangle
    constexpr int DBUF = !DMA ? 1 : COOP ? 64 * NX : 64 * (NX + 2 * DPAIRS);   // doubles per wave next to the tile
    using SharedModel = LdsModel<NX, 1>;
    __shared__ double s_model[SharedModel::SIZE + 2 * KS + 2 + NX];
    __shared__ double s_tile[COOP ? (BLOCK / 64) * 64 * NN : 1];
    __shared__ double s_dma[(BLOCK / 64) * DBUF];
    // PARK (dim_x >= 7, wherever the LDS holds it): the smoothed state of step k+1 waits in LDS ([element][lane], wave-private)
    // between its birth at the end of step k+1 and its one use in step k's correction -- these classes spill, and 44 / 54
    // doubles less to carry is 0.5 KB less scratch per lane.  (At dim_x <= 6, where the state rides in AGPRs, parking it
    // removes 190 of 2530 VALU instructions per step and is NOT faster: 1.83 against 1.81 ms -- measured, not kept.)
    constexpr bool PARK = NX >= 7 && ((COOP ? (BLOCK / 64) * 64 * NN : 1) + (BLOCK / 64) * DBUF + (BLOCK / 64) * 64 * (NX + PL)) * 8 + 4096 <= 160 * 1024;
    __shared__ double s_park[PARK ? (BLOCK / 64) * 64 * (NX + PL) : 1];
    double *park = s_park + (PARK ? (threadIdx.x >> 6) * 64 * (NX + PL) + (threadIdx.x & 63u) : 0);
    const long N = a.N;
    const long blk0 = a.i0 + (long)blockIdx.x * BLOCK;
    const long left = a.i0 + a.cnt - blk0;
    const unsigned last_row = (unsigned)(left < BLOCK ? left : BLOCK) - 1u;
    const bool live = threadIdx.x <= last_row;
    const Lane ln{blk0, live ? threadIdx.x : last_row, N};
    const int n = EXACT ? NX : a.n;
    const int ks = 2 * n + 1;
    double *tile = s_tile + (COOP ? (threadIdx.x >> 6) * 64 * NN : 0);
    double *dbuf = s_dma + (threadIdx.x >> 6) * DBUF;
    const unsigned lane = threadIdx.x & 63u, wave_row0 = (threadIdx.x >> 6) * 64u;
    lds_fill<NX, NX>(s_model + SharedModel::OFF_F, pF, n, n, 1.0, threadIdx.x);
    lds_fill<NX, NX>(s_model + SharedModel::OFF_Q, pQ, n, n, 1.0, threadIdx.x);
    for (unsigned q = threadIdx.x; q < (unsigned)(2 * KS); q += BLOCK) {      // weights re-indexed to the padded point set
        const int which = q / KS, i = q % KS;
        int src = -1;
        if (i == 0) src = 0;
        else if (i <= NX) { if (i <= n) src = i; }
        else { if (i - NX <= n) src = n + (i - NX); }
        const double *W = which ? pWc : pWm;
        s_model[SharedModel::SIZE + q] = (src >= 0 && src < ks) ? W[src] : 0.0;
    }
    __syncthreads();
    int st = 0;
    if constexpr (PAIRED) {
        if (threadIdx.x == 0) make_pair_table<NX>(s_model + SharedModel::SIZE, s_model + SharedModel::SIZE + KS, s_model + SharedModel::SIZE + 2 * KS);
        if (!pair_weights_symmetric<NX>(s_model + SharedModel::SIZE, s_model + SharedModel::SIZE + KS)) st |= ST_BAD_WEIGHTS;
        __syncthreads();
    }
    struct View { SharedModel sm; const double *Wm, *Wc, *Wp; };
    auto fresh = [&](double after = 0.0) {
        int off = 0;
        asm volatile("" : "+v"(off) : "v"(after));
        const double *mb = s_model + off;
        return View{SharedModel{mb}, mb + SharedModel::SIZE, mb + SharedModel::SIZE + KS, mb + SharedModel::SIZE + 2 * KS};
    };
    // the filtered state of step t in flight / landed.  Element-major (and the padded classes): the lane's own registers;
    // cooperative: the wave's slab as register quads, then through the tile.
    struct Fetch {
        double x[COOP ? 1 : NX], P[COOP ? 1 : PL];
        WaveAosFetch<COOP ? NX : 2> fx;
        WaveAosFetch<COOP ? NN : 2> fP;
    };
    auto issue = [&](long t, Fetch &f) {
        if constexpr (COOP) {
            f.fx.issue(a.Xs + t * N * NX + blk0 * NX, wave_row0, lane, last_row);
            f.fP.issue(a.Ps + t * N * NN + blk0 * NN, wave_row0, lane, last_row);
        } else {
            load_rec<NX, 1, LAYOUT, EXACT>(f.x, a.Xs + t * N * n, ln, n, 1, 0.0);
            const RecView<LAYOUT> pv(a.Ps + t * N * n * n, ln, n * n);
            FK_UNROLL for (int i = 0; i < NX; ++i)
                FK_UNROLL for (int j = 0; j < NX; ++j)
                    if (j >= i) f.P[sym_idx<NX>(i, j)] = (EXACT || (i < n && j < n)) ? pv.load(i * n + j) : (i == j ? 1.0 : 0.0);
        }
    };
    auto land = [&](Fetch &f, double (&x)[NX], double (&P)[PL]) {
        if constexpr (COOP) {
            f.fx.to_tile(tile, lane);
            FK_UNROLL for (int c = 0; c < NX; ++c) x[c] = tile[lane * NX + c];
            f.fP.to_tile(tile, lane);
            FK_UNROLL for (int i = 0; i < NX; ++i)
                FK_UNROLL for (int j = 0; j < NX; ++j)
                    if (j >= i) P[sym_idx<NX>(i, j)] = tile[lane * NN + i * NX + j];
            wave_lds_fence();
        } else {
            FK_UNROLL for (int c = 0; c < NX; ++c) { asm volatile("" : "+v"(f.x[c])); x[c] = f.x[c]; }
            FK_UNROLL for (int e = 0; e < PL; ++e) { asm volatile("" : "+v"(f.P[e])); P[e] = f.P[e]; }
        }
    };
    // (DMA, element-major: a lane past the last track takes whatever the fetch left in its LDS slot -- not a copy of the last
    //  track like the register fetch gives it -- so its stores, which would land on the last track, are predicated; hipcc
    //  counts no vector-memory operation in that loop, so the branch costs no wait)
    auto store_x = [&](long t, const double (&x)[NX]) {
        if constexpr (COOP) wave_store_aos_flat<NX>(x, a.xs + t * N * NX + blk0 * NX, wave_row0, tile, lane, last_row);
        else if (!DMA || live) store_rec<NX, 1, LAYOUT, EXACT>(x, a.xs + t * N * n, ln, n, 1);
    };
    auto store_full = [&](double *arr, long t, const double (&M)[NN]) {
        if constexpr (COOP) wave_store_aos_flat<NN>(M, arr + t * N * NN + blk0 * NN, wave_row0, tile, lane, last_row);
        else if (!DMA || live) store_rec<NX, NX, LAYOUT, EXACT>(M, arr + t * N * n * n, ln, n, n);
    };

    // ---- LDS-DMA fetch of the filtered state of step t (DMA instantiations)
    auto dma_issue = [&](long t) {
        if constexpr (DMA) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // every read of the regions about to be overwritten is done
            if constexpr (COOP) {
                const dma_rsrc_t rx = make_dma_rsrc(a.Xs + t * N * NX + blk0 * NX, (last_row + 1u) * (unsigned)NX * 8u);
                const dma_rsrc_t rp = make_dma_rsrc(a.Ps + t * N * NN + blk0 * NN, (last_row + 1u) * (unsigned)NN * 8u);
                const unsigned lx = lds_address(dbuf), lp = lds_address(tile);
                FK_UNROLL for (int it = 0; it < NX / 2; ++it)
                    lds_dma16(rx, (wave_row0 * NX + lane * 2u) * 8u, (unsigned)(it * 1024), lx + (unsigned)(it * 1024));
                FK_UNROLL for (int it = 0; it < NN / 2; ++it)
                    lds_dma16(rp, (wave_row0 * NN + lane * 2u) * 8u, (unsigned)(it * 1024), lp + (unsigned)(it * 1024));
            } else {
                // lanes 0..31 fetch tracks (2l, 2l+1) of the pair's first element, lanes 32..63 of its second: 1 KiB per
                // instruction, lane-linear in LDS = [first element: 64 tracks][second element: 64 tracks]
                const unsigned n8 = (unsigned)N * 8u;
                const dma_rsrc_t rx = make_dma_rsrc(a.Xs + t * N * NX, (unsigned)NX * n8);
                const dma_rsrc_t rp = make_dma_rsrc(a.Ps + t * N * NN, (unsigned)NN * n8);
                const unsigned lb = lds_address(dbuf);
                const unsigned vlo = ((unsigned)blk0 + wave_row0 + (lane & 31u) * 2u) * 8u, hi = lane >> 5;
                FK_UNROLL for (int p = 0; p < NX / 2; ++p)
                    lds_dma16(rx, vlo + hi * n8, (unsigned)(2 * p) * n8, lb + (unsigned)(p * 1024));
                FK_UNROLL for (int p = 0; p < DPAIRS; ++p) {
                    constexpr UpperTriangle<NX> TRI{};          // k-th element of the upper triangle as a flat index i * NX + j
                    const int e1 = TRI.flat[2 * p], e2 = TRI.flat[2 * p + 1 < PL ? 2 * p + 1 : 2 * p];
                    lds_dma16(rp, vlo + hi * ((unsigned)(e2 - e1) * n8), (unsigned)e1 * n8, lb + (unsigned)((NX / 2 + p) * 1024));
                }
            }
        }
    };
    // waits for the fetch and takes the lane's record out of the LDS image (the tile is free for the store path afterwards)
    auto dma_take = [&](double (&x)[NX], double (&P)[PL]) {
        if constexpr (DMA) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if constexpr (COOP) {
                FK_UNROLL for (int c = 0; c < NX; ++c) x[c] = dbuf[lane * NX + c];
                FK_UNROLL for (int i = 0; i < NX; ++i)
                    FK_UNROLL for (int j = 0; j < NX; ++j)
                        if (j >= i) P[sym_idx<NX>(i, j)] = tile[lane * NN + i * NX + j];
            } else {
                FK_UNROLL for (int c = 0; c < NX; ++c) x[c] = dbuf[(c >> 1) * 128 + (c & 1) * 64 + lane];
                FK_UNROLL for (int k = 0; k < PL; ++k) P[k] = dbuf[(NX / 2 + (k >> 1)) * 128 + (k & 1) * 64 + lane];
            }
            wave_lds_fence();
        }
    };

    // the last step is the filter's own output, copied as it is (both triangles: xs, ps = Xs.copy(), Ps.copy()) -- unless this
    // launch continues a chunked call (a.cont): then the window's top step was smoothed by the piece before it and is read
    // back from xs / ps, not rewritten
    double xn[NX], Pn[PL];
    {
        const double *srcx = a.cont ? a.xs : a.Xs, *srcP = a.cont ? a.ps : a.Ps;
        double Pf[NN];
        if constexpr (COOP) {
            WaveAosFetch<NX> fx;
            WaveAosFetch<NN> fP;
            fx.issue(srcx + (a.T - 1) * N * NX + blk0 * NX, wave_row0, lane, last_row);
            fP.issue(srcP + (a.T - 1) * N * NN + blk0 * NN, wave_row0, lane, last_row);
            fx.to_tile(tile, lane);
            FK_UNROLL for (int c = 0; c < NX; ++c) xn[c] = tile[lane * NX + c];
            fP.to_tile(tile, lane);
            FK_UNROLL for (int e = 0; e < NN; ++e) Pf[e] = tile[lane * NN + e];
            wave_lds_fence();
        } else {
            load_rec<NX, 1, LAYOUT, EXACT>(xn, srcx + (a.T - 1) * N * n, ln, n, 1, 0.0);
            load_rec<NX, NX, LAYOUT, EXACT>(Pf, srcP + (a.T - 1) * N * n * n, ln, n, n, 1.0);
        }
        FK_UNROLL for (int i = 0; i < NX; ++i)
            FK_UNROLL for (int j = 0; j < NX; ++j)
                if (j >= i) Pn[sym_idx<NX>(i, j)] = Pf[i * NX + j];
        if (!a.cont) {
            store_x(a.T - 1, xn);
            store_full(a.ps, a.T - 1, Pf);
            if (a.Ks) {
                double Z[NN];
                FK_UNROLL for (int e = 0; e < NN; ++e) Z[e] = 0.0;
                store_full(a.Ks, a.T - 1, Z);
            }
        }
    }
    double xc[DMA ? NX : 1], Pc[DMA ? PL : 1];
    if constexpr (DMA) {
        if (a.T >= 2) {
            dma_issue(a.T - 2);
            double xq[NX], Pq[PL];
            dma_take(xq, Pq);
            FK_UNROLL for (int c = 0; c < NX; ++c) xc[c] = xq[c];
            FK_UNROLL for (int e = 0; e < PL; ++e) Pc[e] = Pq[e];
        }
    }
    // landed: nothing of the prologue is pending at the loop header
    FK_UNROLL for (int c = 0; c < NX; ++c) asm volatile("" ::"v"(xn[c]));
    FK_UNROLL for (int e = 0; e < PL; ++e) asm volatile("" ::"v"(Pn[e]));
    if constexpr (PARK) {
        FK_UNROLL for (int c = 0; c < NX; ++c) park[c * 64] = xn[c];
        FK_UNROLL for (int e = 0; e < PL; ++e) park[(NX + e) * 64] = Pn[e];
    }
    _Pragma("nounroll") for (long t = a.T - 2; t >= 0; --t) {
        double x[NX], P[PL], K[NN];
        if constexpr (DMA) {
            FK_UNROLL for (int c = 0; c < NX; ++c) x[c] = xc[c];
            FK_UNROLL for (int e = 0; e < PL; ++e) P[e] = Pc[e];
            long tp = t > 0 ? t - 1 : 0;
            asm volatile("" : "+s"(tp));
            dma_issue(tp);                                  // in flight during this step's arithmetic, no register held
        } else {
            Fetch f;
            issue(t, f);
            land(f, x, P);
        }
        {
            double xb[NX], Pb[PL];
            if constexpr (PAIRED) st |= ukf_linear_rts_gain_v4<NX>(x, P, a.scale, xb, Pb, K, fresh);
            else st |= ukf_linear_rts_gain_v3<NX>(x, P, a.scale, xb, Pb, K, fresh);
            if constexpr (PARK) {
                double xq[NX], Pq[PL];
                FK_UNROLL for (int c = 0; c < NX; ++c) xq[c] = park[c * 64];
                FK_UNROLL for (int e = 0; e < PL; ++e) Pq[e] = park[(NX + e) * 64];
                ukf_linear_rts_correct<NX>(x, P, xq, Pq, xb, Pb, K);
                FK_UNROLL for (int c = 0; c < NX; ++c) park[c * 64] = x[c];
                FK_UNROLL for (int e = 0; e < PL; ++e) park[(NX + e) * 64] = P[e];
            } else {
                ukf_linear_rts_correct<NX>(x, P, xn, Pn, xb, Pb, K);
                FK_UNROLL for (int c = 0; c < NX; ++c) xn[c] = x[c];
                FK_UNROLL for (int e = 0; e < PL; ++e) Pn[e] = P[e];
            }
        }
        if constexpr (DMA) {
            // in front of this step's stores: nothing else of this wave is in flight, so the wait is for the fetch alone
            double xq[NX], Pq[PL];
            dma_take(xq, Pq);
            FK_UNROLL for (int c = 0; c < NX; ++c) xc[c] = xq[c];
            FK_UNROLL for (int e = 0; e < PL; ++e) Pc[e] = Pq[e];
        }
        store_x(t, x);
        {
            double Pf[NN];
            FK_UNROLL for (int i = 0; i < NX; ++i)
                FK_UNROLL for (int j = 0; j < NX; ++j) Pf[i * NX + j] = P[sym_idx<NX>(i, j)];
            store_full(a.ps, t, Pf);
        }
        if (a.Ks) store_full(a.Ks, t, K);
    }
    if constexpr (PARK) {
        FK_UNROLL for (int c = 0; c < NX; ++c) xn[c] = park[c * 64];
        FK_UNROLL for (int e = 0; e < PL; ++e) Pn[e] = park[(NX + e) * 64];
    }
    if (live && a.status) {
        if (!all_finite<NX>(xn) || !all_finite<PL>(Pn)) st |= ST_NONFINITE;
        a.status[ln.blk0 + ln.tid] = a.status_or ? (a.status[ln.blk0 + ln.tid] | st) : st;
    }
}


#define FK_UKF_GO(NXV, DMAV)                                                                                     \
    do {                                                                                                         \
        const dim3 grid((unsigned)((a.cnt + BLOCK - 1) / BLOCK)), block(BLOCK);                                  \
        const bool ex = exact && a.n == NXV;                                                                     \
        if (layout == FK_LAYOUT_SOA) {                                                                           \
            if (ex && DMAV && dma) hipLaunchKernelGGL((ukf_linear_rts_kernel<NXV, LAYOUT_SOA, true, PV, DMAV>), grid, block, 0, s, a, F, Q, Wm, Wc); \
            else if (ex) hipLaunchKernelGGL((ukf_linear_rts_kernel<NXV, LAYOUT_SOA, true, PV>), grid, block, 0, s, a, F, Q, Wm, Wc);  \
            else hipLaunchKernelGGL((ukf_linear_rts_kernel<NXV, LAYOUT_SOA, false, PV>), grid, block, 0, s, a, F, Q, Wm, Wc);   \
        } else {                                                                                                 \
            if (ex && DMAV && dma) hipLaunchKernelGGL((ukf_linear_rts_kernel<NXV, LAYOUT_AOS, true, PV, DMAV>), grid, block, 0, s, a, F, Q, Wm, Wc); \
            else if (ex) hipLaunchKernelGGL((ukf_linear_rts_kernel<NXV, LAYOUT_AOS, true, PV>), grid, block, 0, s, a, F, Q, Wm, Wc);  \
            else hipLaunchKernelGGL((ukf_linear_rts_kernel<NXV, LAYOUT_AOS, false, PV>), grid, block, 0, s, a, F, Q, Wm, Wc);   \
        }                                                                                                        \
    } while (0)
template <bool PV>
static int ukf_rts_small_t(const UkfRtsArgs &a, const double *F, const double *Q, const double *Wm, const double *Wc, int layout, bool exact, hipStream_t s)
{
    // FK_UKF_DMA=0: the exact classes without the LDS-DMA fetch of the next state (A/B).  The fetch moves 16-byte units: it
    // needs 16-byte aligned arrays and, element-major, an even track count -- with an odd one every element row starts 8
    // bytes off and the last unit of the last row straddles the end of the array (the range check drops it whole, and the last
    // track's last element with it).  Other calls take the same kernel with the register fetch.
    static const bool dma_on = [] { const char *v = getenv("FK_UKF_DMA"); return !(v && v[0] == '0'); }();
    const bool dma = dma_on && reinterpret_cast<uintptr_t>(a.Xs) % 16 == 0 && reinterpret_cast<uintptr_t>(a.Ps) % 16 == 0 &&
                     (layout != FK_LAYOUT_SOA || a.N % 2 == 0);
    if (a.n <= 2) FK_UKF_GO(2, true);
    else if (a.n <= 4) FK_UKF_GO(4, true);
    else FK_UKF_GO(6, true);
    return check_launch("ukf_linear_rts_kernel");
}
template <bool PV>
static int ukf_rts_big_t(const UkfRtsArgs &a, const double *F, const double *Q, const double *Wm, const double *Wc, int layout, bool exact, hipStream_t s)
{
    const bool dma = false;
    if (a.n <= 8) FK_UKF_GO(8, false);
    else FK_UKF_GO(9, false);
    return check_launch("ukf_linear_rts_kernel");
}
#if FK_UKF_HAS(3)
int ukf_rts_launch_small(const UkfRtsArgs &a, const double *F, const double *Q, const double *Wm, const double *Wc, int layout, bool exact, hipStream_t s) { return ukf_rts_small_t<false>(a, F, Q, Wm, Wc, layout, exact, s); }
#endif
#if FK_UKF_HAS(4)
int ukf_rts_launch_big(const UkfRtsArgs &a, const double *F, const double *Q, const double *Wm, const double *Wc, int layout, bool exact, hipStream_t s) { return ukf_rts_big_t<false>(a, F, Q, Wm, Wc, layout, exact, s); }
#endif
#if FK_UKF_HAS(7)
int ukf_rts_launch_small_paired(const UkfRtsArgs &a, const double *F, const double *Q, const double *Wm, const double *Wc, int layout, bool exact, hipStream_t s) { return ukf_rts_small_t<true>(a, F, Q, Wm, Wc, layout, exact, s); }
#endif
#if FK_UKF_HAS(8)
int ukf_rts_launch_big_paired(const UkfRtsArgs &a, const double *F, const double *Q, const double *Wm, const double *Wc, int layout, bool exact, hipStream_t s) { return ukf_rts_big_t<true>(a, F, Q, Wm, Wc, layout, exact, s); }
#endif
#undef FK_UKF_GO

#endif
// ukf_mlg.hip, one object per dim_x
#define FK_UMLG_DECL(NXV)                                         \
    int launch_ukf_mlg_##NXV(const UkfArgs &, int, hipStream_t); \
    int launch_ukf_mlg_rts_##NXV(const UkfRtsArgs &, const double *, const double *, const double *, const double *, int, hipStream_t);
FK_UMLG_DECL(7) FK_UMLG_DECL(8) FK_UMLG_DECL(9) FK_UMLG_DECL(10) FK_UMLG_DECL(11) FK_UMLG_DECL(12) FK_UMLG_DECL(13) FK_UMLG_DECL(14) FK_UMLG_DECL(15) FK_UMLG_DECL(16)
#undef FK_UMLG_DECL
#if FK_UKF_HAS(1)
static int ukf_mlg_launch(const UkfArgs &a, int layout, hipStream_t s)
{
    int rc = 1;
    switch (a.n) {
        case 7: rc = launch_ukf_mlg_7(a, layout, s); break;
        case 8: rc = launch_ukf_mlg_8(a, layout, s); break;
        case 9: rc = launch_ukf_mlg_9(a, layout, s); break;
        case 10: rc = launch_ukf_mlg_10(a, layout, s); break;
        case 11: rc = launch_ukf_mlg_11(a, layout, s); break;
        case 12: rc = launch_ukf_mlg_12(a, layout, s); break;
        case 13: rc = launch_ukf_mlg_13(a, layout, s); break;
        case 14: rc = launch_ukf_mlg_14(a, layout, s); break;
        case 15: rc = launch_ukf_mlg_15(a, layout, s); break;
        case 16: rc = launch_ukf_mlg_16(a, layout, s); break;
        default: break;
    }
    if (rc == 1) {
        set_last_error("fused linear UKF: no four-lane instantiation for these dims");
        return FK_ERR_UNSUPPORTED;
    }
    return rc;
}
#endif

#if FK_UKF_HAS(3)
static int ukf_mlg_rts_launch(const UkfRtsArgs &a, const double *F, const double *Q, const double *Wm, const double *Wc, int layout, hipStream_t s)
{
    int rc = 1;
    switch (a.n) {
        case 7: rc = launch_ukf_mlg_rts_7(a, F, Q, Wm, Wc, layout, s); break;
        case 8: rc = launch_ukf_mlg_rts_8(a, F, Q, Wm, Wc, layout, s); break;
        case 9: rc = launch_ukf_mlg_rts_9(a, F, Q, Wm, Wc, layout, s); break;
        case 10: rc = launch_ukf_mlg_rts_10(a, F, Q, Wm, Wc, layout, s); break;
        case 11: rc = launch_ukf_mlg_rts_11(a, F, Q, Wm, Wc, layout, s); break;
        case 12: rc = launch_ukf_mlg_rts_12(a, F, Q, Wm, Wc, layout, s); break;
        case 13: rc = launch_ukf_mlg_rts_13(a, F, Q, Wm, Wc, layout, s); break;
        case 14: rc = launch_ukf_mlg_rts_14(a, F, Q, Wm, Wc, layout, s); break;
        case 15: rc = launch_ukf_mlg_rts_15(a, F, Q, Wm, Wc, layout, s); break;
        case 16: rc = launch_ukf_mlg_rts_16(a, F, Q, Wm, Wc, layout, s); break;
        default: break;
    }
    if (rc == 1) {
        set_last_error("fused linear UKF smoother: no four-lane instantiation for this dim_x");
        return FK_ERR_UNSUPPORTED;
    }
    return rc;
}
#endif

// Which calls the several-lanes-per-track kernels (ukf_mlg.hip) serve, decided by measurement (profiles/r05/ukf_mlg/):
//   filter:   dim_x 10..16 (no one-lane kernel exists there); at 7..9 the one-lane classes are faster (8: 1.9 vs 2.8 ms, 9: 2.2
//             vs 4.4 ms at N = 1e5, T = 100) and keep the call;
//   smoother: dim_x 10..16, and 7..9 too for pair-weight callers (8: 5.1 / 8.2 -> 4.4 / 4.1 ms, 9: 10.3 / 13.8 -> 7.8 / 7.4 ms
//             element-major / NumPy order): the one-lane smoother classes run one wave per SIMD with 0.4-1.3 KB of scratch.
// A/B knobs, read once per process: FK_UKF_MLG=0 takes the several-lane kernels out altogether (dim_x >= 10 then answers
// FK_ERR_UNSUPPORTED and the host takes the building blocks); FK_UKF_MLG_MIN_NX / FK_UKF_MLG_RTS_MIN_NX = 7..10 move the
// filter's / the smoother's lower bound.
// (ONE snapshot of the environment for the whole library -- fk_host.cpp: this file is compiled as eight objects, and a copy per
// object, each initialised at its own first call, let fk_ukf_linear_supported and the launchers disagree when a knob changed in
// between)
struct UkfMlgRoute { int fwd_min, rts_min; };
UkfMlgRoute ukf_mlg_route();
static int ukf_mlg_min_nx() { return ukf_mlg_route().fwd_min; }
static int ukf_mlg_rts_min_nx() { return ukf_mlg_route().rts_min; }

static int fail(int code, const char *msg)
{
    set_last_error(msg);
    return code;
}

// FK_UKF_PADDED=1: the padded instantiations also at the exact dims (A/B, and the parity tests of the padded path)
static bool ukf_exact()
{
    static const bool padded = [] { const char *v = getenv("FK_UKF_PADDED"); return v && v[0] == '1'; }();
    return !padded;
}

// The pair-regrouped kernels run when the caller asserts weights equal within every +- pair (FK_UKF_FLAG_PAIR_WEIGHTS);
// FK_UKF_PAIRED=0 keeps the index-order sums for every call (A/B, and the parity tests of that path).
static bool ukf_paired(const fk_ukf_desc *d)
{
    static const bool off = [] { const char *v = getenv("FK_UKF_PAIRED"); return v && v[0] == '0'; }();
    return !off && (d->flags & FK_UKF_FLAG_PAIR_WEIGHTS) != 0;
}

}  // namespace fk

using namespace fk;

extern "C" {

#if FK_UKF_HAS(1)
int fk_ukf_linear_supported(int32_t n, int32_t m, int32_t flags, int32_t smoother)
{
    const bool pairw = (flags & FK_UKF_FLAG_PAIR_WEIGHTS) != 0;
    if (smoother) {
        if (n >= 10 && n <= 16) return pairw && ukf_mlg_rts_min_nx() <= 10;
        return n >= 1 && n <= 9;
    }
    if (n >= 10 && n <= 16) return m >= 1 && m <= 8 && pairw && ukf_mlg_min_nx() <= 10;
    return (n >= 1 && n <= 6 && m >= 1 && m <= 3) || (n >= 7 && n <= 9 && m >= 1 && m <= 4);
}
#endif

#if FK_UKF_HAS(1)
int fk_ukf_linear_batch_f64(const fk_ukf_desc *d, const double *F, const double *H, const double *Q,
                            const double *R, const double *Wm, const double *Wc, const double *z,
                            const uint8_t *mask, double *x, double *P, double *means, double *covs,
                            int32_t *status, void *stream)
{
    if (!d) return fail(FK_ERR_BAD_ARG, "desc is NULL");
    // dim_x 10..16 (dim_z 1..8): four / eight lanes per track (ukf_mlg.hip), the pair-regrouped sums only
    const bool big = d->n >= 10 && d->n <= 16 && d->m >= 1 && d->m <= 8;
    const bool quad = big || (d->n >= ukf_mlg_min_nx() && d->n <= 9 && d->m >= 1 && d->m <= 4 && (d->flags & FK_UKF_FLAG_PAIR_WEIGHTS) && ukf_paired(d));
    if (big) {
        if (ukf_mlg_min_nx() > 10 || !(d->flags & FK_UKF_FLAG_PAIR_WEIGHTS))
            return fail(FK_ERR_UNSUPPORTED, "fused linear UKF at dim_x 10..16: needs weights equal within every +- pair (FK_UKF_FLAG_PAIR_WEIGHTS; and FK_UKF_MLG != 0)");
    } else if (d->n < 1 || d->n > 9 || d->m < 1 || d->m > 4 || (d->n <= 6 && d->m > 3))
        return fail(FK_ERR_UNSUPPORTED, "fused linear UKF: dim_x 1..6 with dim_z 1..3, dim_x 7..9 with dim_z 1..4, dim_x 10..16 with dim_z 1..8");
    if (d->N < 0 || d->T < 0 || !F || !H || !Q || !R || !Wm || !Wc || !z || !x || !P)
        return fail(FK_ERR_BAD_ARG, "fused linear UKF: bad argument");
    if ((double)d->N * d->n * d->n * 8.0 >= 4294967296.0 - 32.0) return fail(FK_ERR_UNSUPPORTED, "fused linear UKF: record block >= 4 GiB, split the batch");
    if (d->N == 0 || d->T == 0) return FK_OK;
    UkfArgs a0{};
    a0.F = F; a0.H = H; a0.Q = Q; a0.R = R; a0.Wm = Wm; a0.Wc = Wc; a0.z = z; a0.mask = mask;
    a0.x = x; a0.P = P; a0.means = means; a0.covs = covs; a0.status = status;
    a0.N = d->N; a0.T = d->T; a0.n = d->n; a0.m = d->m; a0.scale = d->scale;
    a0.i0 = 0; a0.cnt = d->N; a0.status_or = 0;
    {   // FK_UKF_SOA_PAIRS=0: the element-major outputs as 8-byte stores everywhere (A/B)
        const char *pv = getenv("FK_UKF_SOA_PAIRS");
        a0.soa_pairs = !(pv && pv[0] == '0');
    }
    const int layout = d->layout;
    const bool exact = ukf_exact(), paired = ukf_paired(d);
    // one piece: tracks [a.i0, a.i0 + a.cnt), a.T steps from the pointers in a.  Classes (2,2), (4,2), (6,3), (8,4), (9,3),
    // (9,4): the exact instantiation where the dims are the class's own, the padded one otherwise.
    auto one = [&](const UkfArgs &a, hipStream_t s) -> int {
        if (quad) return ukf_mlg_launch(a, layout, s);
        if (paired) return (a.n <= 6 && a.m <= 3) ? ukf_fwd_launch_small_paired(a, layout, exact, s) : ukf_fwd_launch_big_paired(a, layout, exact, s);
        return (a.n <= 6 && a.m <= 3) ? ukf_fwd_launch_small(a, layout, exact, s) : ukf_fwd_launch_big(a, layout, exact, s);
    };
    // tail filling (fk_chunks.hpp): FK_UKF_CHUNKS="G,H" cuts the call into G track groups x H time chunks on G streams, the
    // state handed from chunk to chunk through x / P in place (bit-identical results).  Default: one launch -- at
    // BASELINE configs[3] (1563 waves for 2048 wave slots) there is no last round to fill, and the forced decompositions
    // measured no faster (profiles/r03/ukf_chunking.jsonl).
    return ukf_chunked_call(a0, a0.n, a0.m, one, (hipStream_t)stream);
}
#endif

#if FK_UKF_HAS(3)
int fk_ukf_linear_rts_f64(const fk_ukf_desc *d, const double *F, const double *Q, const double *Wm, const double *Wc,
                          const double *Xs, const double *Ps, double *xs, double *Ps_out, double *K, int32_t *status,
                          void *stream)
{
    if (!d) return fail(FK_ERR_BAD_ARG, "desc is NULL");
    // dim_x 10..16 -- and 7..9 for pair-weight callers (ukf_mlg_route) --: four / eight lanes per track (ukf_mlg.hip), pair-regrouped sums only
    const bool big = d->n >= 10 && d->n <= 16;
    const bool quad = big || (d->n >= ukf_mlg_rts_min_nx() && d->n <= 9 && (d->flags & FK_UKF_FLAG_PAIR_WEIGHTS) && ukf_paired(d));
    if (big) {
        if (ukf_mlg_rts_min_nx() > 10 || !(d->flags & FK_UKF_FLAG_PAIR_WEIGHTS))
            return fail(FK_ERR_UNSUPPORTED, "fused linear UKF smoother at dim_x 10..16: needs weights equal within every +- pair (FK_UKF_FLAG_PAIR_WEIGHTS; and FK_UKF_MLG != 0)");
    } else if (d->n < 1 || d->n > 9) return fail(FK_ERR_UNSUPPORTED, "fused linear UKF smoother: dim_x 1..9, 10..16");
    if (quad && (double)d->N * d->n * d->n * 8.0 >= 4294967296.0 - 32.0) return fail(FK_ERR_UNSUPPORTED, "fused linear UKF smoother: record block >= 4 GiB, split the batch");
    if (d->N < 0 || d->T < 0 || !F || !Q || !Wm || !Wc || !Xs || !Ps || !xs || !Ps_out)
        return fail(FK_ERR_BAD_ARG, "fused linear UKF smoother: bad argument");
    if ((double)d->N * d->n * d->n * 8.0 >= 4294967296.0) return fail(FK_ERR_UNSUPPORTED, "fused linear UKF smoother: record block >= 4 GiB, split the batch");
    if (d->N == 0 || d->T == 0) return FK_OK;
    UkfRtsArgs a0{};
    a0.Xs = Xs; a0.Ps = Ps; a0.xs = xs; a0.ps = Ps_out; a0.Ks = K; a0.status = status;
    a0.N = d->N; a0.T = d->T; a0.n = d->n; a0.scale = d->scale;
    a0.i0 = 0; a0.cnt = d->N; a0.cont = 0; a0.status_or = 0;
    const int layout = d->layout;
    if (quad) return d->T >= 1 ? ukf_mlg_rts_launch(a0, F, Q, Wm, Wc, layout, (hipStream_t)stream) : FK_OK;
    const bool exact = ukf_exact(), paired = ukf_paired(d);
    auto one = [&](const UkfRtsArgs &a, hipStream_t s) -> int {
        if (paired) return a.n <= 6 ? ukf_rts_launch_small_paired(a, F, Q, Wm, Wc, layout, exact, s) : ukf_rts_launch_big_paired(a, F, Q, Wm, Wc, layout, exact, s);
        return a.n <= 6 ? ukf_rts_launch_small(a, F, Q, Wm, Wc, layout, exact, s) : ukf_rts_launch_big(a, F, Q, Wm, Wc, layout, exact, s);
    };
    // Tail filling (fk_chunks.hpp): the classes of dim_x >= 5 run one wave per SIMD, so BASELINE configs[3]'s 1563 waves are
    // two rounds for 1.53 rounds of work; cut into track groups x backward time windows on helper streams the pieces of
    // different groups fill each other's tails (bit-identical results: a window's top step is read back from the smoothed
    // outputs).  FK_UKF_RTS_CHUNKS="G,H" forces a decomposition ("1,1": one launch).
    const long slots = 1024L * (d->n <= 2 ? 4 : d->n <= 4 ? 2 : 1);
    return ukf_rts_chunked_call(a0, a0.n, slots, one, (hipStream_t)stream);
}
#endif

}  // extern "C"
