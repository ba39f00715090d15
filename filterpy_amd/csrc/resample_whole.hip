// resample_whole.hip -- systematic / stratified resampling of SHORT weight vectors (gfx950): one workgroup takes a
// filter's WHOLE vector (Np <= 8 weights x 1024 threads) and does it in one round.
//
//   fk_resample_systematic_f64  <- systematic_resample (filterpy/monte_carlo/resampling.py:117-150)
//   fk_resample_stratified_f64  <- stratified_resample (:80-114)
//
// Round 2's short-vector kernel (resample_local_kernel) walked a vector's 2048-weight chunks in sequence -- four
// dependent chunk rounds for BASELINE configs[4]'s 8000 particles, each with its own exact scan, ~75k dependent clocks
// per filter, 39 us for 125 filters.  Here the exact scan sees the whole vector at once (fk_resample_whole.hpp: bounds
// from plain prefix sums -> clean / dirty elements -> segments -> a chain of two adds per segment -> cs_j), the weights
// never leave the registers (8 B read per particle, once), every weight computes its slot boundary n(cs_j) straight
// from its cumulative sum (fk_resample_math.hpp: no division, no search), and ONE window of Np slots in LDS turns the
// boundaries into indices: run heads -> inclusive max-scan -> 16-byte coalesced stores (4 B written per particle).
// Seven workgroup barriers per filter in all; a thread whose eight weights share a binade (nearly all do) spends ~40 VALU
// instructions per weight from load to store.  A vector the round cannot take -- a negative / NaN / huge weight, more
// than WH_DMAX dirty elements (half-ulp ties by the hundred, running sums below 2^-900), a failed binade check -- is
// handed to the reference's merge loop, run literally by one thread: slow, but still the reference's answer.
//
// This unit is compiled with -ffp-contract=off: positions and sums must be single IEEE operations.
#include <stdlib.h>
#include <string.h>

#include "../../include/filterhip.h"
#include "fk_device.hpp"
#include "fk_exact_scan.hpp"
#include "fk_resample_math.hpp"
#include "fk_resample_whole.hpp"
#include "resample_dev.hpp"
#include "resample_onepass.hpp"

namespace fk {

struct WholeArgs {
    int Np, force_exact;
    const double *w, *u;
    int32_t *idx, *status;
};

// Build-time instrumentation (tools/op_phase.py --whole builds a separate library with -DFK_OP_CLOCKS; the shipped library
// has none of it): thread 0 of every workgroup adds the shader-clock ticks between the barriers to fk_wh_phase[].
#ifdef FK_OP_CLOCKS
__device__ unsigned long long fk_wh_phase[16];
#define WH_CLOCK_START() long long t_prev = clock64()
#define WH_CLOCK(slot) do { if (threadIdx.x == 0) { const long long t_now = clock64(); atomicAdd(&fk_wh_phase[slot], (unsigned long long)(t_now - t_prev)); t_prev = t_now; } } while (0)
#define WH_COUNT(slot, n) do { if (threadIdx.x == 0) atomicAdd(&fk_wh_phase[slot], (unsigned long long)(n)); } while (0)
#else
#define WH_CLOCK_START() do { } while (0)
#define WH_CLOCK(slot) do { } while (0)
#define WH_COUNT(slot, n) do { } while (0)
#endif

template <int NT>
struct WholeShared {
    static constexpr int NW = NT / 64, CAP = NT * WH_ITEMS;
    int win[CAP];                          // slot window: the index of the weight whose run starts there, else -1
    int nlast[NT];                         // slot boundary after each thread's last element
    double wtot[NW];                       // per-wave partials: plain sums,
    u64 ptot[NW];                          //   increment sums,
    int dtot[NW];                          //   dirty counts,
    int wmax[NW];                          //   running maxima of the window
    double d_w[WH_DMAX];                   // dirty element r: its weight, the increment prefix up to it
    u64 d_ps[WH_DMAX];
    double d_cs[WH_DMAX];                  //   and (from the chain) the running sum after its real add
    int seg_e[WH_DMAX + 1];                // segment r: claimed ulp exponent, increment prefix and running sum at its start
    u64 seg_ps0[WH_DMAX + 1];
    double seg_c[WH_DMAX + 1];
    double carry_out;
    int fail;
    int lit_status;
};

// ---- weights: eight consecutive ones per thread, straight from HBM into registers (padding: +0.0) ----
__device__ __forceinline__ void wh_fetch(const double *wf, int j0, int Np, double (&w)[WH_ITEMS])
{
    if ((((uintptr_t)wf) & 15) == 0 && j0 + WH_ITEMS <= Np) {
        const double *src = wf + j0;
        FK_UNROLL for (int q = 0; q < WH_ITEMS; q += 2) {
            const f64x2 t = *reinterpret_cast<const f64x2 *>(src + q);
            w[q] = t.x;
            w[q + 1] = t.y;
        }
    } else {
        FK_UNROLL for (int q = 0; q < WH_ITEMS; ++q) {
            const int j = j0 + q;
            const double t = wf[j < Np ? j : 0];                           // Np >= 1: always a valid address
            w[q] = j < Np ? t : 0.0;
        }
    }
}

// The whole algorithm on ONE filter: plain-prefix boundaries, the exact round where an estimate is inside the error band, the
// literal loop for garbage.  Rounds 3 / 4 ran it as the kernel (resample_whole_kernel below: FK_WHOLE_QUICK=0); since round 5 it
// is the rare tail of resample_whole_quick_kernel.  One filter, start to finish, by the whole workgroup (every barrier inside is
// reached by all of its threads).
template <bool STRATIFIED, int NT>
__device__ __forceinline__ void wh_full_one(const WholeArgs &a, const int f_in, WholeShared<NT> &sh)
{
    using Sh = WholeShared<NT>;
    constexpr int NW = Sh::NW, CAP = Sh::CAP;
    // (opaque copies: as the quick kernel's tail this code must share NO value with the common path in front of it -- a shared
    //  one would be live across the whole tail and be spilled by the path that cannot afford it)
    int tid = threadIdx.x, Np = a.Np, f = f_in;
    asm volatile("" : "+v"(tid));
    asm volatile("" : "+s"(Np), "+s"(f));
    const int lane = tid & 63, wave = tid >> 6;
    const double Nd = (double)Np, halfNd = 0.5 * Nd;
    const int j0 = tid * WH_ITEMS;
    WH_CLOCK_START();
    const double *wf = a.w + (long)f * Np;
    int32_t *of = a.idx + (long)f * Np;
    const double u_sys = STRATIFIED ? 0.0 : a.u[f];
    const double *u_str = STRATIFIED ? a.u + (long)f * Np : nullptr;
    double w[WH_ITEMS];
    wh_fetch(wf, j0, Np, w);
    // the slot window and the segment claims are reset while the loads are in flight
    FK_UNROLL for (int g = 0; g < WH_ITEMS / 4; ++g) *reinterpret_cast<i32x4 *>(&sh.win[j0 + 4 * g]) = i32x4{-1, -1, -1, -1};
    for (int r = tid; r <= WH_DMAX; r += NT) sh.seg_e[r] = WH_NONE;
    if (tid == 0) sh.fail = 0;

    double run = 0.0, mn = 0.0;
    FK_UNROLL for (int q = 0; q < WH_ITEMS; ++q) {
        run += w[q];
        mn = w[q] < mn ? w[q] : mn;                                        // a negative weight (NaN / Inf show in the sum)
    }
    const double winc = wave_incl_sum(run);
    if (lane == 63) sh.wtot[wave] = winc;
    const int any_neg = __syncthreads_or(mn < 0.0 ? 1 : 0);                                   // (1)
    WH_CLOCK(0);                                                           // weights landed, sums
    double before = __shfl_up(winc, 1, 64);
    if (lane == 0) before = 0.0;
    double S = 0.0;
    FK_UNROLL for (int wv = 0; wv < NW; ++wv) {
        const double t = sh.wtot[wv];
        if (wv < wave) before += t;
        S += t;
    }
    bool literal = any_neg || !(S < 0x1p1000);                             // negative, NaN, Inf or absurdly large: uniform

    WhPos<STRATIFIED> px;
    px.Np = Np;
    px.Nd = Nd;
    px.halfNd = halfNd;
    px.u_sys = u_sys;
    px.u_str = u_str;
    // ---- step 0: the slot boundaries from the plain prefix sums (fk_resample_whole.hpp, wh_approx_boundaries): exact unless
    // an estimate lies within the error band of an integer -- one vector of 8000 weights in ~500 --, and only then does the
    // workgroup run the exact round below.  FK_WHOLE_EXACT=1 (a.force_exact) runs it for every vector (tests, A/B timing).
    int nb[WH_ITEMS];
    bool exact = true;
    if (!literal) {
        const unsigned unsure = wh_approx_boundaries<STRATIFIED>(w, before, px, nb);
        sh.nlast[tid] = nb[WH_ITEMS - 1];
        exact = __syncthreads_or((unsure != 0 || a.force_exact) ? 1 : 0) != 0;                // (A) (also publishes nlast)
        WH_CLOCK(8);                                                       // plain-prefix boundaries
        WH_COUNT(12, exact ? 1 : 0);
    }

    int D = 0;
    WhThread th;
    int dbase = 0;
    u64 pbase = 0, ptotal = 0;
    if (exact) {
    if (!literal) {
        // ---- clean / dirty, increments; their prefix sums over the workgroup --------------------------------
        wh_classify(w, before, j0, Np, th);
        const int dincl = wave_incl_sum_i32(th.ndirty);
        const u64 pincl = wave_incl_sum_u64(th.psum);
        if (lane == 63) {
            sh.dtot[wave] = dincl;
            sh.ptot[wave] = pincl;
        }
        __syncthreads();                                                                      // (2)
        WH_CLOCK(1);                                                       // classification + scans
        dbase = dincl - th.ndirty;
        pbase = pincl - th.psum;
        FK_UNROLL for (int wv = 0; wv < NW; ++wv) {
            const int dt = sh.dtot[wv];
            const u64 pt = sh.ptot[wv];
            if (wv < wave) {
                dbase += dt;
                pbase += pt;
            }
            D += dt;
            ptotal += pt;
        }
        literal = D > WH_DMAX;                                             // uniform
    }
    if (!literal) {
        wh_lists(w, th, dbase, pbase, sh.seg_e, sh.d_w, sh.d_ps);
        __syncthreads();                                                                      // (3)
        WH_CLOCK(2);                                                       // lists
        if (th.overflow || wh_claims_bad(th, dbase, sh.seg_e)) sh.fail = 1;
        // ---- the chain over the segments (wave 0; lane l holds segment b0 + l and dirty element b0 + l).  The serial
        // part is kept to what IS serial -- two dependent fp64 adds per segment, their operands fetched from the owning
        // lane -- because a lone wave issues one instruction every 4-8 clocks: everything else (the segment's increment
        // sum as a double, the check that the running sum enters and leaves the segment in the claimed binade, the
        // running sum behind the dirty element) is done by the lanes in parallel before and after it -------------------
        if (wave == 0) {
            double c = 0.0;
            int fail = 0;
            for (int b0 = 0; b0 <= D; b0 += 64) {                          // uniform
                const int me = b0 + lane;
                WhSeg sg;
                sg.add = 0.0;
                sg.xf = -1;
                sg.bad = false;
                sg.ps0 = 0;
                if (me <= D) sg = wh_segment(me, D, ptotal, sh.seg_e, sh.d_ps);
                const double my_w = me < D ? sh.d_w[me < WH_DMAX ? me : 0] : 0.0;
                double r_c = 0.0;
                const int cnt = D + 1 - b0 < 64 ? D + 1 - b0 : 64;
                for (int r = 0; r < cnt; ++r) {                            // uniform, serial
                    const double add = lane_bcast(sg.add, r);
                    const double wr = lane_bcast(my_w, r);                 // (+0.0 behind the last segment: c + 0 = c)
                    if (lane == r) r_c = c;
                    c = (c + add) + wr;        // exact add of the segment's increments, real IEEE add of the dirty element
                }
                if (me <= D) {
                    bool fl = sg.bad;
                    const double after = wh_chain_segment(r_c, sg.add, sg.xf, fl);     // the same add, now with its checks
                    fail |= fl ? 1 : 0;
                    sh.seg_c[me] = r_c;
                    sh.seg_ps0[me] = sg.ps0;
                    if (me < D) sh.d_cs[me] = after + my_w;
                }
            }
            if (__builtin_amdgcn_ballot_w64(fail != 0) != 0 && lane == 0) sh.fail = 1;
            if (lane == 0) sh.carry_out = c;
        }
        __syncthreads();                                                                      // (4)
        WH_CLOCK(3);                                                       // claims check + chain
        WH_COUNT(10, D);
        literal = sh.fail != 0;                                            // uniform
    }
    if (literal) {
        // the reference's loop, literally (resampling.py:106-112 / :142-149), one thread
        if (tid == 0) {
            const int st = literal_merge<STRATIFIED>(wf, STRATIFIED ? u_str : a.u + f, (long)Np, of);
            if (a.status) a.status[f] = st;
        }
        __syncthreads();                     // nobody still reads this filter's shared state when the next trip resets it
        return;
    }

    // ---- cumulative sums -> slot boundaries: weight j owns the slots [n_{j-1}, n_j), n_j = n(cs_j) ------------------
    wh_boundaries<STRATIFIED>(th, dbase, pbase, sh.seg_e, sh.seg_c, sh.seg_ps0, sh.d_cs, px, nb);
    sh.nlast[tid] = nb[WH_ITEMS - 1];       // (everybody read step 0's values? nobody has: they are read behind the next barrier)
    __syncthreads();                                                                          // (5)
    WH_CLOCK(4);                                                           // boundaries
    }   // exact
    int nprev = tid == 0 ? 0 : sh.nlast[tid - 1];
    const int u_hi = __builtin_amdgcn_readfirstlane(sh.nlast[NT - 1]);     // = n(carry-out): slots [0, u_hi) get an index
    // window position p = slot + sft: with sft = (address of slot 0 in ints) mod 4 a thread's two quads are 16-byte
    // aligned stores.  A vector that would not fit the window shifted (Np > CAP - 3 at an odd address) goes unshifted
    // and leaves as 4-byte stores.
    const int mis = (int)(((uintptr_t)of >> 2) & 3);
    const int sft = Np + mis <= CAP ? mis : 0;
    const bool vec_ok = sft == mis;
    FK_UNROLL for (int q = 0; q < WH_ITEMS; ++q) {
        if (nb[q] > nprev) {
            sh.win[nprev + sft] = j0 + q;                                  // (nprev < nb[q] <= Np: inside the window)
            nprev = nb[q];
        }
    }
    __syncthreads();                                                                          // (6)
    WH_CLOCK(5);                                                           // heads
    int x[WH_ITEMS];
    FK_UNROLL for (int g = 0; g < WH_ITEMS / 4; ++g) {
        const i32x4 t = *reinterpret_cast<const i32x4 *>(&sh.win[j0 + 4 * g]);
        x[4 * g + 0] = t.x;
        x[4 * g + 1] = t.y;
        x[4 * g + 2] = t.z;
        x[4 * g + 3] = t.w;
    }
    FK_UNROLL for (int e = 1; e < WH_ITEMS; ++e) x[e] = x[e] > x[e - 1] ? x[e] : x[e - 1];
    const int wincl = wave_incl_max(x[WH_ITEMS - 1]);
    if (lane == 63) sh.wmax[wave] = wincl;
    __syncthreads();                                                                          // (7)
    WH_CLOCK(6);                                                           // window read + scan
    int pre = __shfl_up(wincl, 1, 64);
    if (lane == 0) pre = -1;
    FK_UNROLL for (int wv = 0; wv < NW; ++wv) {
        const int t = sh.wmax[wv];
        if (wv < wave) pre = pre > t ? pre : t;
    }
    const int last = Np - 1;
    FK_UNROLL for (int g = 0; g < WH_ITEMS / 4; ++g) {
        int v[4];
        FK_UNROLL for (int e = 0; e < 4; ++e) {
            const int s = j0 + 4 * g + e - sft;                            // the slot of this window position
            const int m = x[4 * g + e] > pre ? x[4 * g + e] : pre;
            v[e] = s < u_hi ? m : last;                                    // positions >= cumsum[-1]: resampling.py:109,145
        }
        const int s0 = j0 + 4 * g - sft;
        if (vec_ok && s0 >= 0 && s0 + 3 < Np) *reinterpret_cast<i32x4 *>(&of[s0]) = i32x4{v[0], v[1], v[2], v[3]};
        else {
            FK_UNROLL for (int e = 0; e < 4; ++e)
                if (s0 + e >= 0 && s0 + e < Np) of[s0 + e] = v[e];
        }
    }
    if (tid == 0 && a.status) a.status[f] = u_hi < Np ? ST_OVERRUN : 0;
    WH_CLOCK(7);                                                           // stores issued
    WH_COUNT(11, 1);
    __syncthreads();                         // (a workgroup that takes another filter: this one's window is done with)
}

// EU = waves per SIMD the register allocation must allow (a 1024-thread workgroup is four waves per SIMD)
template <bool STRATIFIED, int NT, int EU>
__global__ void __launch_bounds__(NT, EU)
resample_whole_kernel(const WholeArgs a)
{
    __shared__ WholeShared<NT> sh;
    wh_full_one<STRATIFIED, NT>(a, (int)blockIdx.x, sh);
}

// Round 5: the QUICK kernel.  With the exact round inside its one code path the kernel above needs 87 VGPRs: ONE 1024-thread
// workgroup per CU, so nothing covers a workgroup's load latency (4.2k of its 12.7k clocks), its seven barriers or its store tail
// (profiles/r03/resample_whole_phase_clocks.jsonl) -- and a persistent grid that prefetched the next filter's weights into
// registers measured 35.1 against 36.2 us at 1000 x 8000: the latency is not the weights' alone (profiles/r05/c5/).  Here the
// common path -- weights -> plain prefix sums -> slot boundaries from the estimates -> heads -> max-scan -> stores -- is
// straight-line code that fits the 64 VGPRs at which TWO 1024-thread workgroups share a CU (32 waves), each filling the other's
// stalls: 25.0 against 36.3 us on the same GPU, 24.1-24.9 us once its two __syncthreads_or (three s_barrier each) were single
// barriers.  A vector with an estimate inside the error band (one in ~5000 at 8000 weights
// since the band is priced at the prefix sums' real depth, fk_resample_whole.hpp), a negative / NaN / huge weight, or
// FK_WHOLE_EXACT=1 leaves through wh_full_one, inlined behind the common path's `return` and marked unlikely: the register
// allocator spills there (128-190 bytes per lane) and nowhere in the common path (checked in the ISA: no scratch instruction in
// front of the first s_endpgm; tests/test_host_logic.py holds that).  First tried as two launches (quick kernel + the full kernel
// on marked filters): the second launch cost 4.8 us + the boundary even with nothing to do, 19 us with one deferred filter.
template <bool STRATIFIED, int NT, int EU>
__global__ void __launch_bounds__(NT, EU)
resample_whole_quick_kernel(const WholeArgs a)
{
    constexpr int NW = NT / 64, CAP = NT * WH_ITEMS;
    __shared__ WholeShared<NT> sh;
    int (&s_win)[CAP] = sh.win;
    int (&s_nlast)[NT] = sh.nlast;
    double (&s_wtot)[NW] = sh.wtot;
    int (&s_wmax)[NW] = sh.wmax;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Np = a.Np;
    const int f = blockIdx.x;
    const double *wf = a.w + (long)f * Np;
    int32_t *of = a.idx + (long)f * Np;
    const int j0 = tid * WH_ITEMS;
    double w[WH_ITEMS];
    wh_fetch(wf, j0, Np, w);
    WhPos<STRATIFIED> px;
    px.Np = Np;
    px.Nd = (double)Np;
    px.halfNd = 0.5 * px.Nd;
    px.u_sys = STRATIFIED ? 0.0 : a.u[f];
    px.u_str = STRATIFIED ? a.u + (long)f * Np : nullptr;
    FK_UNROLL for (int g = 0; g < WH_ITEMS / 4; ++g) *reinterpret_cast<i32x4 *>(&s_win[j0 + 4 * g]) = i32x4{-1, -1, -1, -1};
    double run = 0.0, mn = 0.0;
    FK_UNROLL for (int q = 0; q < WH_ITEMS; ++q) {
        run += w[q];
        mn = w[q] < mn ? w[q] : mn;
    }
    const double winc = wave_incl_sum(run);
    // (one s_barrier here and one at (A) instead of the three a __syncthreads_or costs: a wave that holds a negative weight
    //  publishes NaN as its total -- the sum then fails the sanity test like any other garbage --, and (A)'s verdict travels in
    //  one flag word per wave)
    const bool wave_neg = __builtin_amdgcn_ballot_w64(mn < 0.0) != 0;
    if (lane == 63) s_wtot[wave] = wave_neg ? __builtin_nan("") : winc;
    __syncthreads();                                                                          // (1)
    double before = __shfl_up(winc, 1, 64);
    if (lane == 0) before = 0.0;
    double S = 0.0;
    FK_UNROLL for (int wv = 0; wv < NW; ++wv) {
        const double t = s_wtot[wv];
        if (wv < wave) before += t;
        S += t;
    }
    const bool garbage = !(S < 0x1p1000);                                  // uniform: negative, NaN, Inf or absurdly large
    int nb[WH_ITEMS];
    unsigned unsure = 1;
    if (!garbage) {
        unsure = wh_approx_boundaries<STRATIFIED>(w, before, px, nb);
        s_nlast[tid] = nb[WH_ITEMS - 1];
    }
    {
        const bool wave_unsure = __builtin_amdgcn_ballot_w64(unsure != 0 || a.force_exact != 0) != 0;
        if (lane == 0) sh.dtot[wave] = wave_unsure ? 1 : 0;
    }
    __syncthreads();                                                                          // (A) (also publishes nlast)
    int tail = 0;
    FK_UNROLL for (int wv = 0; wv < NW; ++wv) tail |= sh.dtot[wv];
    if (__builtin_expect(tail == 0, 1)) {
    int nprev = tid == 0 ? 0 : s_nlast[tid - 1];
    const int u_hi = __builtin_amdgcn_readfirstlane(s_nlast[NT - 1]);
    const int mis = (int)(((uintptr_t)of >> 2) & 3);
    const int sft = Np + mis <= CAP ? mis : 0;
    const bool vec_ok = sft == mis;
    FK_UNROLL for (int q = 0; q < WH_ITEMS; ++q) {
        if (nb[q] > nprev) {
            s_win[nprev + sft] = j0 + q;
            nprev = nb[q];
        }
    }
    __syncthreads();                                                                          // (6)
    int x[WH_ITEMS];
    FK_UNROLL for (int g = 0; g < WH_ITEMS / 4; ++g) {
        const i32x4 t = *reinterpret_cast<const i32x4 *>(&s_win[j0 + 4 * g]);
        x[4 * g + 0] = t.x;
        x[4 * g + 1] = t.y;
        x[4 * g + 2] = t.z;
        x[4 * g + 3] = t.w;
    }
    FK_UNROLL for (int e = 1; e < WH_ITEMS; ++e) x[e] = x[e] > x[e - 1] ? x[e] : x[e - 1];
    const int wincl = wave_incl_max(x[WH_ITEMS - 1]);
    if (lane == 63) s_wmax[wave] = wincl;
    __syncthreads();                                                                          // (7)
    int pre = __shfl_up(wincl, 1, 64);
    if (lane == 0) pre = -1;
    FK_UNROLL for (int wv = 0; wv < NW; ++wv) {
        const int t = s_wmax[wv];
        if (wv < wave) pre = pre > t ? pre : t;
    }
    const int last = Np - 1;
    FK_UNROLL for (int g = 0; g < WH_ITEMS / 4; ++g) {
        int v[4];
        FK_UNROLL for (int e = 0; e < 4; ++e) {
            const int s = j0 + 4 * g + e - sft;
            const int m = x[4 * g + e] > pre ? x[4 * g + e] : pre;
            v[e] = s < u_hi ? m : last;
        }
        const int s0 = j0 + 4 * g - sft;
        if (vec_ok && s0 >= 0 && s0 + 3 < Np) *reinterpret_cast<i32x4 *>(&of[s0]) = i32x4{v[0], v[1], v[2], v[3]};
        else {
            FK_UNROLL for (int e = 0; e < 4; ++e)
                if (s0 + e >= 0 && s0 + e < Np) of[s0 + e] = v[e];
        }
    }
    if (tid == 0 && a.status) a.status[f] = u_hi < Np ? ST_OVERRUN : 0;
    return;
    }
    // ---- the rare way out: the whole algorithm on this filter, from its weights (inline, at the END of the kernel: nothing of
    // the common path is live across it, so what it spills under the 64-VGPR budget it spills inside itself) ----
    __syncthreads();                         // (every thread has read what step 0 left in the shared arrays)
    wh_full_one<STRATIFIED, NT>(a, f, sh);
}

// Np <= 8192: one workgroup of 256 / 512 / 1024 threads per filter (8 weights per thread)
bool whole_supported(int64_t Np) { return Np >= 1 && Np <= 1024 * WH_ITEMS; }

int whole_launch(bool stratified, int64_t Fn, int64_t Np, const double *w, const double *u, int32_t *idx,
                 int32_t *status, hipStream_t s)
{
    if (Fn > 0x7fffffffL || !whole_supported(Np)) return FK_ERR_UNSUPPORTED;
    WholeArgs a;
    a.Np = (int)Np;
    const char *fe = getenv("FK_WHOLE_EXACT");
    a.force_exact = (fe && fe[0] == '1') ? 1 : 0;
    a.w = w;
    a.u = u;
    a.idx = idx;
    a.status = status;
    // FK_WHOLE_QUICK=0: the round-3 / round-4 kernel (one code path, one workgroup of <= 128 VGPRs per filter): A/B and tests
    const char *qv = getenv("FK_WHOLE_QUICK");
    const bool quick = !(qv && qv[0] == '0');
#define GO(K, NTV, EUV)                                                                                          \
    do {                                                                                                         \
        if (stratified) hipLaunchKernelGGL((K<true, NTV, EUV>), dim3((unsigned)Fn), dim3(NTV), 0, s, a);         \
        else hipLaunchKernelGGL((K<false, NTV, EUV>), dim3((unsigned)Fn), dim3(NTV), 0, s, a);                   \
    } while (0)
    if (quick) {
        // 64 VGPRs in the common path -- 32 waves per CU: two 1024-, four 512-thread workgroups (256 threads: with the tail
        // 97 VGPRs, five workgroups -- 20 waves -- per CU)
        if (Np <= 256 * WH_ITEMS) GO(resample_whole_quick_kernel, 256, 5);
        else if (Np <= 512 * WH_ITEMS) GO(resample_whole_quick_kernel, 512, 8);
        else GO(resample_whole_quick_kernel, 1024, 8);
    } else {
        if (Np <= 256 * WH_ITEMS) GO(resample_whole_kernel, 256, 4);
        else if (Np <= 512 * WH_ITEMS) GO(resample_whole_kernel, 512, 4);
        else GO(resample_whole_kernel, 1024, 4);
    }
#undef GO
    return check_launch("resample_whole_kernel");
}

}  // namespace fk

#ifdef FK_OP_CLOCKS
extern "C" int fk_debug_wh_phases(unsigned long long *out)      // only in the instrumented build (tools/op_phase.py --whole)
{
    unsigned long long host[16];
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(fk::fk_wh_phase), sizeof(host)) != hipSuccess) return FK_ERR_LAUNCH;
    for (int q = 0; q < 16; ++q) out[q] = host[q];
    memset(host, 0, sizeof(host));
    return hipMemcpyToSymbol(HIP_SYMBOL(fk::fk_wh_phase), host, sizeof(host)) == hipSuccess ? FK_OK : FK_ERR_LAUNCH;
}
#endif
