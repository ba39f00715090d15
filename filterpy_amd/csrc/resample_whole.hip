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
    int Np, force_exact, Fn;
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

// EU = waves per SIMD the register allocation must allow (a 1024-thread workgroup is four waves per SIMD)
template <bool STRATIFIED, int NT, int EU>
__global__ void __launch_bounds__(NT, EU)
resample_whole_kernel(const WholeArgs a)
{
    using Sh = WholeShared<NT>;
    constexpr int NW = Sh::NW, CAP = Sh::CAP;
    __shared__ Sh sh;
    const int Np = a.Np;
    const double Nd = (double)Np, halfNd = 0.5 * Nd;
    WH_CLOCK_START();

    // ---- weights: eight consecutive ones per thread, straight from HBM into registers (padding: +0.0) ----
    auto fetch = [&](int f, int j0, double (&dst)[WH_ITEMS]) {
        const double *wf = a.w + (long)f * Np;
        if ((((uintptr_t)wf) & 15) == 0 && j0 + WH_ITEMS <= Np) {
            const double *src = wf + j0;
            FK_UNROLL for (int q = 0; q < WH_ITEMS; q += 2) {
                const f64x2 t = *reinterpret_cast<const f64x2 *>(src + q);
                dst[q] = t.x;
                dst[q + 1] = t.y;
            }
        } else {
            FK_UNROLL for (int q = 0; q < WH_ITEMS; ++q) {
                const int j = j0 + q;
                const double t = wf[j < Np ? j : 0];                       // Np >= 1: always a valid address
                dst[q] = j < Np ? t : 0.0;
            }
        }
    };
    // Round 5: the grid is PERSISTENT -- a workgroup takes the filters blockIdx.x, + gridDim.x, ... (whole_launch: one grid's
    // worth of resident workgroups) and fetches the NEXT filter's weights into a second register set before it starts on the
    // current one: of a filter's 12.7k clocks 4.2k were the HBM latency of its weights with nothing else resident on the CU to
    // cover them (one 1024-thread workgroup per CU; profiles/r03/resample_whole_phase_clocks.jsonl), and every later round of a
    // 1000-filter call paid them again.  The prefetched set (and the next filter's u) is landed just before the index stores,
    // i.e. behind ~8k clocks of work, and NO other global load sits in the common path in between: vmcnt retires in order, so
    // a wait for any younger load would wait for the prefetch too.  The exact round (one vector in ~500) needs the registers:
    // the prefetched set is dead across it and simply fetched again behind it (from L2 by then) -- no spill in either path.
    double w[WH_ITEMS], wn[WH_ITEMS];
    FK_UNROLL for (int q = 0; q < WH_ITEMS; ++q) w[q] = wn[q] = 0.0;
    int f = blockIdx.x;
    double u_sys = 0.0, u_next = 0.0;
    if (f < a.Fn) {
        fetch(f, (int)threadIdx.x * WH_ITEMS, w);
        if (!STRATIFIED) u_sys = a.u[f];
    }
    // (landed before the loop is entered: with loads pending on ONE of the two ways into the loop head, the wait in front of the
    //  first use of w becomes a vmcnt(0) on both -- and in every later trip it would wait for the prefetch issued just above it)
    FK_UNROLL for (int q = 0; q < WH_ITEMS; ++q) asm volatile("" ::"v"(w[q]));
    asm volatile("" ::"v"(u_sys));
    for (; f < a.Fn; f += gridDim.x) {
    // the thread's coordinates are re-derived in every trip from an opaque copy of threadIdx.x: as loop invariants the two
    // dozen addresses and masks derived from them were hoisted, kept live across the whole body and SPILLED (104 bytes of
    // scratch, reloaded behind a vmcnt(0) in front of two of the barriers); re-deriving them is a handful of VALU instructions
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6, j0 = tid * WH_ITEMS;
    const double *wf = a.w + (long)f * Np;
    int32_t *of = a.idx + (long)f * Np;
    const double *u_str = STRATIFIED ? a.u + (long)f * Np : nullptr;
    const int f_next = f + (int)gridDim.x;
    const bool has_next = f_next < a.Fn;                                   // uniform
    auto prefetch = [&]() {
        if (has_next) {
            fetch(f_next, j0, wn);
            if (!STRATIFIED) u_next = a.u[f_next];
        } else {                             // (a definition on every path: the set is DEAD across the exact round, not carried)
            FK_UNROLL for (int q = 0; q < WH_ITEMS; ++q) wn[q] = 0.0;
            u_next = 0.0;
        }
    };
    // systematic: in flight from here until the end of this trip.  Stratified: its boundaries GATHER u[f][.] (a wait for those
    // loads would be a wait for the prefetch issued before them), so there the prefetch starts behind the boundaries
    if (!STRATIFIED) prefetch();
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
        __syncthreads();                     // nobody still reads this filter's shared state when the next one resets it
        if (STRATIFIED) prefetch();
        FK_UNROLL for (int q = 0; q < WH_ITEMS; ++q) w[q] = wn[q];
        u_sys = u_next;
        continue;
    }

    // ---- cumulative sums -> slot boundaries: weight j owns the slots [n_{j-1}, n_j), n_j = n(cs_j) ------------------
    wh_boundaries<STRATIFIED>(th, dbase, pbase, sh.seg_e, sh.seg_c, sh.seg_ps0, sh.d_cs, px, nb);
    sh.nlast[tid] = nb[WH_ITEMS - 1];       // (everybody read step 0's values? nobody has: they are read behind the next barrier)
    __syncthreads();                                                                          // (5)
    WH_CLOCK(4);                                                           // boundaries
    }   // exact
    if (STRATIFIED) prefetch();
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
    FK_UNROLL for (int q = 0; q < WH_ITEMS; ++q) asm volatile("" ::"v"(wn[q]));   // the next filter's weights land HERE
    asm volatile("" ::"v"(u_next));
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
    FK_UNROLL for (int q = 0; q < WH_ITEMS; ++q) w[q] = wn[q];
    u_sys = u_next;
    }   // filters of this workgroup
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
    a.Fn = (int)Fn;
    // one grid's worth of resident workgroups (16 waves per CU at <= 128 VGPRs: one 1024-, two 512-, four 256-thread workgroups),
    // each walking its filters f, f + grid, ...  FK_WHOLE_GRID=<workgroups> forces a grid (0: one workgroup per filter as before)
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    const char *gv = getenv("FK_WHOLE_GRID");                              // (per call, like FK_WHOLE_EXACT: the tests flip it)
    const long forced = gv ? atol(gv) : -1L;
    const int per_cu = Np <= 256 * WH_ITEMS ? 4 : Np <= 512 * WH_ITEMS ? 2 : 1;
    long G = forced > 0 ? forced : forced == 0 ? Fn : (long)cus * per_cu;
    if (G > Fn) G = Fn;
    const dim3 grid((unsigned)G);
#define GO(NTV, EUV)                                                                                             \
    do {                                                                                                         \
        if (stratified) hipLaunchKernelGGL((resample_whole_kernel<true, NTV, EUV>), grid, dim3(NTV), 0, s, a);   \
        else hipLaunchKernelGGL((resample_whole_kernel<false, NTV, EUV>), grid, dim3(NTV), 0, s, a);             \
    } while (0)
    // (an instantiation budgeted for two 1024-thread workgroups per CU -- 64 VGPRs -- was measured and dropped: it spills,
    // 36 against 32 us at 1000 x 8000, 11.8 against 8.1 us at 125 x 8000; profiles/r03/resample_whole_variants.txt)
    if (Np <= 256 * WH_ITEMS) GO(256, 4);
    else if (Np <= 512 * WH_ITEMS) GO(512, 4);
    else GO(1024, 4);
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
