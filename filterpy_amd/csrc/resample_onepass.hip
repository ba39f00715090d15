// resample_onepass.hip -- systematic / stratified resampling in ONE pass over the weights (gfx950).
//
//   fk_resample_systematic_f64  <- systematic_resample (filterpy/monte_carlo/resampling.py:117-150)
//   fk_resample_stratified_f64  <- stratified_resample (:80-114)
//
// The reference compares the positions against numpy.cumsum(weights), a strictly sequential fp64 add chain, and
// the indices must match it bit for bit.  This kernel reads every weight from HBM exactly once (8 B) and writes
// every index once (4 B): the algorithmic traffic.  One workgroup owns one chunk of 2048 weights of one filter;
// the sequential dependency between chunks is carried by TWO chained "decoupled look-back" hand-offs:
//
//   stage 1 (approximate): plain fp64 chunk sums, summed in any order, tell every chunk -- with a rigorous error
//            bound -- which binade the exact running sum is in while it crosses the chunk;
//   stage 2 (exact): inside one binade every add of the sequential chain is the integer map C -> C + inc
//            (fk_exact_scan.hpp, fast_inc), so a chunk whose binade is known publishes the integer sum of its
//            increments BEFORE its own carry-in is known, and the exact carry of a run of such chunks is an
//            integer sum -- associative.  Chunks that cross a binade, hold a half-ulp tie or start a vector
//            resolve their carry-out from the exact carry-in with the general segmented scan below and publish
//            it; nobody ever trusts the approximation: every shortcut is re-verified against the exact carry.
//
// All hand-off words are single 8-byte granules {state, payload} moved with relaxed agent-scope atomics (the
// data IS the flag: no fences, no L2 write-backs; MI355X guide, Guideline 16 form R2).  Workgroups take their
// chunk from an atomic ticket, so a chunk only ever waits for chunks taken earlier; spins are bounded and set an
// abort word instead of hanging.
//
// Output side: slot i receives #{ j : cs_j <= pos_i }.  Seen from the weights, weight j owns the slots
// [n(cs_{j-1}), n(cs_j)), n(c) = #{ i : pos_i < c } (fk_resample_math.hpp: exact, division-free).  Each weight
// drops its index at the head of its run in an LDS window; an inclusive max-scan fills the runs; the window
// leaves as 16-byte coalesced stores.  No per-output search, no division.
//
// This unit is compiled with -ffp-contract=off: positions and sums must be single IEEE operations.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/filterhip.h"
#include "fk_device.hpp"
#include "fk_exact_scan.hpp"
#include "fk_resample_math.hpp"
#include "resample_onepass.hpp"
#include "resample_dev.hpp"

namespace fk {

#ifndef FK_OP_ITEMS
#define FK_OP_ITEMS 8          // weights per thread (A/B builds: tools/exp_rs_variants.py)
#endif
constexpr int OP_THREADS = 256;
constexpr int OP_ITEMS = FK_OP_ITEMS;
constexpr int OP_TILE = OP_THREADS * OP_ITEMS;             // 2048 weights per chunk
constexpr int OP_WIN = 12 * OP_THREADS;                     // 3072 output slots per window, 12 consecutive per thread
constexpr int OP_REGIONS = 32;                             // ticket heads, each in a cache line of its own: ONE word takes
                                                           // ~88 atomics per microsecond, and so do eight words in one line
                                                           // (measured: 488k tickets -> 6.1 ms whatever the kernel did)
constexpr int OP_SERIAL_RUN = 256;                         // elements per sequential run of the general scan
constexpr int OP_MAX_TIES = 8;                             // more half-ulp ties than this in a chunk: sequential
constexpr unsigned OP_SPIN_LIMIT = 1u << 22;
#ifndef FK_OP_LOOKBACK
#define FK_OP_LOOKBACK 16      // predecessors polled per look-back step (lanes of wave 0 that load a hand-off word).  64 until
                               // round 5; A/B/A in one lease (profiles/r05/c5/onepass_lookback_window.txt): 125 x 8e6 3221 / 3209 /
                               // 3202 / 3228 us at 64 / 32 / 16 / 64, 8 x 8e6 733 / 721 / 712 / 730, 1 x 8e6 267.8 / 263.0 / 262.8 /
                               // 267.8 -- the "look-back over-fetch" is worth 0.7-2.7 %
#endif
#ifndef FK_OP_SLEEP
#define FK_OP_SLEEP 2          // s_sleep argument between two polls
#endif
constexpr int OP_LB = FK_OP_LOOKBACK;

// per-chunk hand-off record (zeroed by the launcher before every call)
struct OpDesc {
    u64 approx;     // stage 1: bits of a double; low 2 bits = state (0 empty, 1 chunk sum, 2 inclusive prefix)
    u64 exact;      // stage 2: state:2 | eu9:9 | C:53   (1 = increment sum of the chunk, 2 = carry-out = C * 2^eu,
                    //          3 = carry-out only in `raw`)
    u64 raw;        // bits of the exact carry-out
    u64 raw_chk;    // ~raw: the pair is valid when raw_chk == ~raw
    u64 exact2;     // speculation (round 3): state 1 = increment sum of the chunk at a SECOND ulp exponent (same packing)
    u64 pad;
};

struct OpHead {
    unsigned next;      // next ticket of the region
    unsigned pad[31];   // 128 bytes apart
};
struct OpCtl {      // head of the workspace (zeroed with it)
    OpHead head[OP_REGIONS];
    unsigned abort;
    unsigned pad[31];
};

constexpr u64 ST_MASK = 3;
constexpr double OP_SANE_LO = 0x1p-900, OP_SANE_HI = 0x1p900;
constexpr u64 OP_NAN_BITS = 0x7ff8000000000000ull;

__device__ __forceinline__ u64 ld_agent(const u64 *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(u64 *p, u64 v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ int pad8(int j) { return j + (j >> 3); }   // one spare slot per eight: the
                                                                     // thread-owns-8-consecutive reads are conflict-free

// scratch of segmented_cumsum (below): the elements of a chunk that cannot be taken as integer increments (`dirty`:
// a binade crossing, the first weight of a vector, a half-ulp tie, an ambiguous prediction) cut it into SEGMENTS of
// clean elements
#ifndef FK_SEG_DMAX
#define FK_SEG_DMAX 64
#endif
constexpr int SEG_DMAX = FK_SEG_DMAX;              // dirty elements a chunk may hold before the scan declines
constexpr int SEG_NONE = 0x7fffffff;      // a segment without a non-zero clean element claims no binade
struct SegShared {
    double wtot[OP_THREADS / 64];         // per-wave partial sums (approximate prefix)
    u64 ptot[OP_THREADS / 64];            // per-wave increment sums
    int dtot[OP_THREADS / 64];            // per-wave dirty counts
    int d_pos[SEG_DMAX];                  // dirty element r: its position,
    double d_w[SEG_DMAX];                 //   its weight,
    u64 d_ps[SEG_DMAX];                   //   the increment prefix up to it,
    double d_cs[SEG_DMAX];                //   and (from the chain) the running sum after its real add
    int seg_e[SEG_DMAX + 1];              // segment r (the clean elements before dirty element r): claimed ulp exponent,
    u64 seg_ps0[SEG_DMAX + 1];            //   increment prefix at its start,
    double seg_c[SEG_DMAX + 1];           //   running sum at its start (what a segment of zeros keeps)
    double carry_out;
    int fail;
#ifdef FK_OP_CLOCKS
    long long dbg_t[5];                   // thread 0's clock after each barrier of segmented_cumsum (tools/op_phase.py)
    long long dbg_in;
    int dbg_D;
#endif
};

struct OpShared {
    double tile[OP_TILE + OP_TILE / 8];   // weights (padded), later their cumulative sums; then aliased by the window
    int nlast[OP_THREADS];                // slot boundary after each thread's last element
    double wsum[OP_THREADS / 64];         // per-wave partials (stage-1 sum, increment sums)
    int wmax[OP_THREADS / 64];
    double bc_d[2];                       // broadcast slots written by wave 0 / thread 0
    int bc_i[6];
    SegShared seg;
    __device__ __forceinline__ int *win() { return reinterpret_cast<int *>(tile); }
};
static_assert(sizeof(OpShared) <= 24576 * OP_ITEMS / 8, "six workgroups per CU need <= 24 KiB of LDS each (the kernel runs five)");

// ---- hand-off words --------------------------------------------------------------------------------------
__device__ __forceinline__ u64 pack_approx(double v, u64 state) { return (double_to_bits(v) & ~ST_MASK) | state; }
__device__ __forceinline__ double unpack_approx(u64 w) { return bits_to_double(w & ~ST_MASK); }
__device__ __forceinline__ u64 pack_exact(u64 state, int eu, double C) { return state | ((u64)(eu & 511) << 2) | ((u64)C << 11); }
__device__ __forceinline__ int exact_eu9(u64 w) { return (int)((w >> 2) & 511); }
__device__ __forceinline__ double exact_C(u64 w) { return (double)(w >> 11); }

__device__ __forceinline__ bool spin_fail(unsigned &spins, unsigned *abort_word)
{
    __builtin_amdgcn_s_sleep(FK_OP_SLEEP);
    ++spins;
    if ((spins & 63u) == 0u) {
        if (spins > OP_SPIN_LIMIT) __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return true;
    }
    return false;
}

// stage 1, wave 0 only: approximate running sum entering chunk k (k >= 1) = sum of the predecessors' chunk sums
// down to the nearest published inclusive prefix.  NaN = poison (a bad weight upstream) or abort.
__device__ double lookback_approx(const OpDesc *d, int k, int lane, unsigned *abort_word, const int lbw = OP_LB)
{
    double A = 0.0;
    unsigned spins = 0;
    for (int j = k - 1;; j -= lbw) {
        const int jj = j - lane;
        u64 word, incl;
        for (;;) {
            // before the vector: inclusive prefix 0; lanes beyond the window: a chunk sum of 0 (never waited for)
            word = lane >= lbw ? (u64)1 : (jj >= 0 ? ld_agent(&d[jj].approx) : (u64)2);
            incl = __ballot((word & ST_MASK) == 2);
            const u64 empty = __ballot((word & ST_MASK) == 0);
            const u64 need = incl ? ((incl & (0 - incl)) << 1) - 1 : ~(u64)0;   // lanes up to the nearest inclusive one
            if ((empty & need) == 0) break;
            if (spin_fail(spins, abort_word)) return __builtin_nan("");
        }
        const int L = incl ? __builtin_ctzll(incl) : 63;
        const double v = lane <= L ? unpack_approx(word) : 0.0;
        A += lane_bcast(wave_incl_sum(v), 63);
        if (incl) return A;
    }
}

// exact carry-out pair of chunk jj (valid once its stage-2 word says state >= 2)
__device__ double read_raw(const OpDesc *d, int jj, unsigned *abort_word)
{
    unsigned spins = 0;
    for (;;) {
        const u64 r = ld_agent(&d[jj].raw), c = ld_agent(&d[jj].raw_chk);
        if (c == ~r) return bits_to_double(r);
        if (spin_fail(spins, abort_word)) return __builtin_nan("");
    }
}

// stage 2, wave 0 only: EXACT running sum entering chunk k (k >= 1).  A chunk that knows its binade (`clean`,
// ulp exponent eu) may add up the increment sums of predecessors in the same binade down to the nearest
// published carry-out; anything else waits for the carry-out of chunk k-1.  `ok` = false on abort.
__device__ double lookback_exact(const OpDesc *d, int k, bool clean, int eu, int lane, unsigned *abort_word, bool &ok, const int lbw = OP_LB)
{
    ok = true;
    double acc = 0.0;            // integer sum of increments, in units of 2^eu
    unsigned spins = 0;
    const int my9 = eu & 511;
    for (int j = k - 1;; j -= lbw) {
        const int jj = j - lane;
        u64 word, term;
        for (;;) {
            // before the vector: carry-out 0 (state 3: value in `raw`, handled below without a load); lanes beyond
            // the window: an increment sum of 0
            word = lane >= lbw ? (u64)1 : (jj >= 0 ? ld_agent(&d[jj].exact) : (u64)3);
            const u64 st = word & ST_MASK;
            const bool is_term = st >= 2;
            const bool compat = (clean || lane >= lbw) && st == 1 && (exact_eu9(word) == my9 || (word >> 11) == 0);
            term = __ballot(is_term);
            const u64 blocked = __ballot(!(is_term || compat));
            const u64 below = term ? (term & (0 - term)) - 1 : ~(u64)0;   // lanes nearer than the nearest carry-out
            if ((blocked & below) == 0) break;
            if (spin_fail(spins, abort_word)) { ok = false; return 0.0; }
        }
        const int L = term ? __builtin_ctzll(term) : 64;
        acc += lane_bcast(wave_incl_sum(lane < L ? exact_C(word) : 0.0), 63);
        if (!term) continue;
        const u64 tw = lane_bcast_u64(word, L);
        const int tj = j - L;
        if (L == 0 && j == k - 1) {                    // the carry-out of chunk k-1 itself
            if ((tw & ST_MASK) == 2 && clean && exact_eu9(tw) == my9) return scale2(exact_C(tw), eu);
            return tj >= 0 ? read_raw(d, tj, abort_word) : 0.0;
        }
        // a farther carry-out plus the increment sums in between: valid only inside my binade
        double Ct = -1.0;
        if ((tw & ST_MASK) == 2 && exact_eu9(tw) == my9) Ct = exact_C(tw);
        else if (tj >= 0) {
            const double c = read_raw(d, tj, abort_word);
            if (c > OP_SANE_LO && c < OP_SANE_HI && ulp_exp(c) == eu) Ct = scale2(c, -eu);
        }
        if (Ct >= 0x1p52 && Ct + acc < 0x1p53) return scale2(Ct + acc, eu);
        // inconsistent with what stage 1 promised (never observed; kept so that nothing rests on it): wait for k-1
        unsigned spins2 = 0;
        for (;;) {
            const u64 w1 = ld_agent(&d[k - 1].exact);
            if ((w1 & ST_MASK) >= 2) return read_raw(d, k - 1, abort_word);
            if (spin_fail(spins2, abort_word)) { ok = false; return 0.0; }
        }
    }
}

// ---- speculation (round 3) ---------------------------------------------------------------------------------------------------
// Round 2's chunk walked a chain of dependent global round trips: chunk sum -> stage 1 (approximate carry-in: WHICH binade)
// -> increments in that binade -> stage 2 (exact carry-in).  A published pair (ulp exponent e, increment sum I) is a
// CONDITIONAL truth -- "if my chunk lies entirely in the binade with ulp 2^e, its adds raise the running sum by I ulps" -- and
// publishing it needs no knowledge of the carry-in at all.  So a chunk GUESSES its binade from its own sum (the k chunks
// before it weigh about k times as much), publishes the sums for the guess g and for g + 1 the moment its weights have
// landed, and looks back ONCE: increment sums of the predecessors down to the nearest published exact carry-out.  If that
// carry-out lies in binade g (or g + 1), every chunk in between published a sum for it, and carry-out + sums + the chunk's
// own sum stays inside the binade, then -- running sums are monotone -- every one of those chunks lies entirely in that
// binade, the conditional truths all apply, and the exact carry-in is an integer sum: verified with exact arithmetic, no
// approximation involved.  Anything else (a guess that missed, a crossing, a tie, the start of a vector, garbage) is a
// miss: the chunk takes round 2's two stages unchanged.  Wave 0 only.  which = 0 / 1: hit at g / g + 1, -1: miss.
#ifndef FK_OP_SPEC_SINGLE
#define FK_OP_SPEC_SINGLE 1     // 0: always speculate on two binades (rounds 3 / 4; A/B build)
#endif
#ifndef FK_OP_SPEC_POLLS
#define FK_OP_SPEC_POLLS 24
#endif
#ifndef FK_OP_GUESS_LO
#define FK_OP_GUESS_LO 0.96    // the guess g is the binade of this fraction of k times the chunk's own sum
#endif
__device__ double lookback_spec(const OpDesc *d, int k, int g, double I0, double I1, bool v0, bool v1, int lane, int &which)
{
    which = -1;
    double acc0 = 0.0, acc1 = 0.0;
    bool ok0 = v0, ok1 = v1;
    const int g9 = g & 511, h9 = (g + 1) & 511;
    unsigned polls = 0;
    for (int j = k - 1; j >= 0; j -= 64) {
        const int jj = j - lane;
        u64 e1, e2, rw, term;
        for (;;) {
            e1 = jj >= 0 ? ld_agent(&d[jj].exact) : (u64)3;               // before the vector: a carry-out nobody can use
            e2 = jj >= 0 ? ld_agent(&d[jj].exact2) : (u64)0;
            rw = jj >= 0 ? ld_agent(&d[jj].raw) : (u64)0;
            term = __ballot((e1 & ST_MASK) >= 2);
            const u64 unpub = __ballot((e1 & ST_MASK) == 0 && (e2 & ST_MASK) == 0);
            const u64 below = term ? (term & (0 - term)) - 1 : ~(u64)0;   // lanes nearer than the nearest carry-out
            if ((unpub & below) == 0) break;
            if (++polls > FK_OP_SPEC_POLLS) return 0.0;                   // somebody is not speculating: do not wait here
            __builtin_amdgcn_s_sleep(FK_OP_SLEEP);
        }
        const int L = term ? __builtin_ctzll(term) : 64;
        const bool mine = lane < L;
        const bool a0 = ((e1 & ST_MASK) == 1 && exact_eu9(e1) == g9), b0 = ((e2 & ST_MASK) == 1 && exact_eu9(e2) == g9);
        const bool a1 = ((e1 & ST_MASK) == 1 && exact_eu9(e1) == h9), b1 = ((e2 & ST_MASK) == 1 && exact_eu9(e2) == h9);
        const double m0 = a0 ? exact_C(e1) : (b0 ? exact_C(e2) : 0.0), m1 = a1 ? exact_C(e1) : (b1 ? exact_C(e2) : 0.0);
        ok0 = ok0 && __ballot(mine && !(a0 || b0)) == 0;
        ok1 = ok1 && __ballot(mine && !(a1 || b1)) == 0;
        if (!ok0 && !ok1) return 0.0;
        acc0 += lane_bcast(wave_incl_sum(mine ? m0 : 0.0), 63);            // integers: exact below 2^53 (checked at the end)
        acc1 += lane_bcast(wave_incl_sum(mine ? m1 : 0.0), 63);
        if (!term) continue;
        // the carry-out: state 2 = (C, eu mod 512) in the word, the double itself in `raw` -- taken only if the two agree
        // (they are separate stores: a stale `raw` shows as a mismatch), and the exponent comes from the double
        const u64 tw = lane_bcast_u64(e1, L), tr = lane_bcast_u64(rw, L);
        if ((tw & ST_MASK) != 2 || j - L < 0) return 0.0;
        const double c = bits_to_double(tr);
        if (!(c > OP_SANE_LO && c < OP_SANE_HI)) return 0.0;
        const int ec = ulp_exp(c);
        const double Ct = scale2(c, -ec);
        if (Ct != exact_C(tw) || (ec & 511) != exact_eu9(tw)) return 0.0;
        if (ok0 && ec == g && acc0 < 0x1p53 && Ct + acc0 + I0 < 0x1p53) {
            which = 0;
            return scale2(Ct + acc0, g);
        }
        if (ok1 && ec == g + 1 && acc1 < 0x1p53 && Ct + acc1 + I1 < 0x1p53) {
            which = 1;
            return scale2(Ct + acc1, g + 1);
        }
        return 0.0;
    }
    return 0.0;
}

// publish the exact carry-out of a chunk (one lane)
__device__ __forceinline__ void publish_carry(OpDesc *dk, double c)
{
    const u64 r = (c == c) ? double_to_bits(c) : OP_NAN_BITS;
    st_agent(&dk->raw, r);
    st_agent(&dk->raw_chk, ~r);
    if (c > OP_SANE_LO && c < OP_SANE_HI) {
        const int eu = ulp_exp(c);
        st_agent(&dk->exact, pack_exact(2, eu, scale2(c, -eu)));
    } else {
        st_agent(&dk->exact, (u64)3);
    }
}

// plain sequential adds of elements [pos, stop) by ONE thread (the start of a vector, odd values, tie-ridden chunks);
// the result goes to sh.bc_d[0].  Kept out of line: inlined, its loop costs the whole kernel 20 registers.
__device__ __attribute__((noinline)) void serial_run(OpShared &sh, int pos, int stop, double carry)
{
    double c = carry;
    _Pragma("nounroll") for (int j = pos; j < stop; j += 4) {              // 4 independent LDS reads, then the add chain
        double v0 = sh.tile[pad8(j + 0)];
        double v1 = sh.tile[pad8(j + 1 < OP_TILE ? j + 1 : j)];
        double v2 = sh.tile[pad8(j + 2 < OP_TILE ? j + 2 : j)];
        double v3 = sh.tile[pad8(j + 3 < OP_TILE ? j + 3 : j)];
        v0 = c + v0;
        v1 = v0 + v1;
        v2 = v1 + v2;
        v3 = v2 + v3;
        sh.tile[pad8(j + 0)] = v0;
        c = v0;
        if (j + 1 < stop) { sh.tile[pad8(j + 1)] = v1; c = v1; }
        if (j + 2 < stop) { sh.tile[pad8(j + 2)] = v2; c = v2; }
        if (j + 3 < stop) { sh.tile[pad8(j + 3)] = v3; c = v3; }
    }
    sh.bc_d[0] = c;
}

// ---- the general exact scan of one chunk (binade crossings, ties, vector start, odd values) ---------------
// In: weights in sh.tile (padded), exact carry-in.  Out: cumulative sums in sh.tile, returns the carry-out.
// Rounds: inside the binade of the running sum the adds are integer increments -> one block scan; the round
// ends at the first element that would leave the binade or that holds a half-ulp tie -- that single element is
// added with a real IEEE add and the next round starts behind it.  A running sum that is not a positive normal
// number (start of a vector, garbage) is advanced by plain sequential adds.
__device__ double general_cumsum(OpShared &sh, int len, double carry)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int pos = 0;
    bool force_serial = false;
    while (pos < len) {                                                   // uniform
        if (force_serial || !(carry > OP_SANE_LO && carry < OP_SANE_HI)) {
            const int stop = force_serial ? len : (pos + OP_SERIAL_RUN < len ? pos + OP_SERIAL_RUN : len);
            __syncthreads();
            if (tid == 0) serial_run(sh, pos, stop, carry);
            __syncthreads();
            carry = sh.bc_d[0];
            pos = stop;
            continue;
        }
        // one binade: integer increments (a half-ulp tie or a weight this binade cannot take counts 2^54 and so
        // ends the round like a crossing), block scan, first stop element
        const int eu = ulp_exp(carry);
        const double ulp = ulp_of(carry);
        const double C0 = scale2(carry, -eu);
        double E[OP_ITEMS];
        double run = 0.0;
        int nties = 0;
        FK_UNROLL for (int q = 0; q < OP_ITEMS; ++q) {
            const int j = tid * OP_ITEMS + q;
            bool tk = false;
            const double e = fast_inc(sh.tile[pad8(j)], eu, tk);          // (slots >= len hold +0.0)
            nties += (tk && j >= pos && j < len) ? 1 : 0;
            run += (j >= pos && j < len) ? (tk ? 0x1p54 : e) : 0.0;
            E[q] = run;
            FK_STAGE();                                                    // one element at a time: short live ranges
        }
        const double winc = wave_incl_sum(run);
        if (tid == 0) sh.bc_i[0] = OP_TILE;
        if (lane == 63) sh.wsum[wave] = winc;
        const int tie_threads = __syncthreads_count(nties > 0 ? 1 : 0);
        if (tie_threads > OP_MAX_TIES) {                                  // tie-ridden chunk: sequential
            force_serial = true;
            continue;
        }
        // running C before this thread's elements.  NOT winc - run: a 2^54 marker among the own elements would
        // absorb the low bits of the lanes before it
        const double before = __shfl_up(winc, 1, 64);
        double acc = C0 + (lane == 0 ? 0.0 : before);
        FK_UNROLL for (int wv = 0; wv < OP_THREADS / 64; ++wv)
            if (wv < wave) acc += sh.wsum[wv];
        // (exact below 2^53; a sum that reaches 2^53 stays >= 2^53, which is all the stop test needs)
        int my_stop = OP_TILE;
        FK_UNROLL for (int q = OP_ITEMS - 1; q >= 0; --q) {
            const int j = tid * OP_ITEMS + q;
            E[q] += acc;
            if (j >= pos && j < len && !(E[q] < 0x1p53)) my_stop = j;     // descending: the first one wins
        }
        if (my_stop < OP_TILE) atomicMin(&sh.bc_i[0], my_stop);
        __syncthreads();
        const int first_stop = __builtin_amdgcn_readfirstlane(sh.bc_i[0]);
        const int stop = first_stop < len ? first_stop : len;             // first element NOT covered by this round
        const double w_stop = stop < len ? sh.tile[pad8(stop)] : 0.0;     // still the weight: read before the writes
        FK_UNROLL for (int q = 0; q < OP_ITEMS; ++q) {
            const int j = tid * OP_ITEMS + q;
            if (j >= pos && j < stop) sh.tile[pad8(j)] = E[q] * ulp;      // exact: no tie, no crossing before `stop`
        }
        __syncthreads();
        if (stop > pos) carry = sh.tile[pad8(stop - 1)];
        if (stop < len) {
            carry = carry + w_stop;                                       // the real IEEE add
            if (tid == 0) sh.tile[pad8(stop)] = carry;
            pos = stop + 1;
        } else {
            pos = len;
        }
        __syncthreads();
    }
    __syncthreads();
    return carry;
}

// ---- the exact scan of one chunk in ONE round (binade crossings, vector start, rare ties) ---------------------------
// general_cumsum above spends one block scan and four barriers per binade the running sum passes through; the start
// of a vector passes through a dozen (the sum doubles every time the element count does), which made the first chunk
// of every vector cost ~45k clocks and a short vector -- four dependent tiles -- mostly that.  Here the binades are
// PREDICTED for all elements at once and the prediction is then verified, so nothing rests on it:
//   1. plain fp64 prefix sums (any order, all terms >= 0) bound the exact running sum before and after every add to
//      a factor (1 +- delta); an element whose two bounds lie in one binade e is CLEAN: its add is the integer map
//      C -> C + inc_e(w) (fk_exact_scan.hpp, fast_inc), a zero weight is clean in any binade (inc = 0); everything
//      else -- the crossings themselves, the first weight of a vector (running sum 0), half-ulp ties, ambiguous
//      bounds -- is DIRTY and will be added with a real IEEE add;
//   2. one wrapping 64-bit prefix sum of the increments and one of the dirty flags: the dirty elements (a few dozen
//      at most, else the scan declines) cut the chunk into segments of clean elements that share one binade;
//   3. wave 0 walks the segments in order, two dependent adds per segment: the exact running sum entering a segment
//      must have the claimed ulp and, with the segment's whole increment sum added, stay in that binade -- then every
//      prefix inside did -- else the scan declines; a dirty element is one real add;
//   4. every element reads its segment's start and adds its scaled increment prefix: cs_j = c_start + (PS_j - PS0) 2^e.
// Declining (returns false; the tile still holds the weights) hands the chunk to general_cumsum.
// In: weights in sh.tile (padded; slots >= len hold +0.0, no negative / NaN weight), exact carry-in.
// Out: cumulative sums in sh.tile, *c_out = carry-out.
// The scan comes in three pieces so that a chunk of the one-pass kernel can do everything that does not need the EXACT carry-in
// before that carry arrives (op_chunk_slow): seg_prepare (steps 1-2 and the lists of step 3; its bounds only need the carry
// to a relative `slack`, e.g. the approximate carry-in of stage 1 -- the classification is a prediction, step 3 verifies it),
// seg_chain (step 3's walk, wave 0: the only part behind the carry -- a few dependent adds), seg_finish (step 4).
struct SegRegs {
    int eq[OP_ITEMS];
    unsigned dirty, claims;
    int dbase, D;
    u64 pbase, ptotal;
};

// SYNC_CLAIMS: one more barrier at the end, behind which sg.fail holds every thread's verdict on the claims (the caller's wave 0
// decides on it alone, before the next barrier).  Returns false (uniform) if the chunk holds too many dirty elements.
template <bool SYNC_CLAIMS>
__device__ __forceinline__ bool seg_prepare(OpShared &sh, int len, double carry, double slack, SegRegs &R)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    SegShared &sg = sh.seg;
#ifdef FK_OP_CLOCKS
    const long long t_in = clock64();
#endif
    if (tid <= SEG_DMAX) sg.seg_e[tid] = SEG_NONE;
    if (tid == 0) sg.fail = 0;
    // (registers: the weights and one exponent per element are kept; the increments are recomputed from them at each
    // of their three uses -- two instructions -- instead of being held)
    double w[OP_ITEMS];
    int (&eq)[OP_ITEMS] = R.eq;
    double run = 0.0;
    FK_UNROLL for (int q = 0; q < OP_ITEMS; ++q) {
        w[q] = sh.tile[pad8(tid * OP_ITEMS + q)];
        run += w[q];
    }
    const double winc = wave_incl_sum(run);
    if (lane == 63) sg.wtot[wave] = winc;
    __syncthreads();                                                                          // (1)
#ifdef FK_OP_CLOCKS
    if (tid == 0) { sg.dbg_t[0] = clock64(); sg.dbg_in = t_in; }
#endif
    double excl = __shfl_up(winc, 1, 64);
    if (lane == 0) excl = 0.0;
    FK_UNROLL for (int wv = 0; wv < OP_THREADS / 64; ++wv)
        if (wv < wave) excl += sg.wtot[wv];
    // classification + increments
    unsigned dirty = 0, claims = 0;       // bit q: element q is dirty / claims a binade (clean and non-zero)
    int dl = 0;
    u64 psum = 0;
    double prev = carry + excl, arun = 0.0;
    FK_UNROLL for (int q = 0; q < OP_ITEMS; ++q) {
        const int j = tid * OP_ITEMS + q;
        arun += w[q];
        const double cur = carry + (excl + arun);
        const double lo = prev * (1.0 - slack), hi = cur * (1.0 + slack);
        const bool known = lo > OP_SANE_LO && hi < OP_SANE_HI && ulp_exp(lo) == ulp_exp(hi);
        const int e = ulp_exp(lo);
        const double x = scale2(w[q], -e) + 0.5;
        const double i = floor(x);
        const bool zero = w[q] == 0.0;
        const bool ok = known && i != x && i < 0x1p53;        // no half-ulp tie; (i < 2^53 always holds: w <= hi)
        const bool in = j < len;
        if (in && !zero && ok) claims |= 1u << q;
        if (in && !zero && !ok) dirty |= 1u << q;
        eq[q] = e;
        psum += (in && !zero && ok) ? (u64)i : (u64)0;
        prev = cur;
    }
    dl = __builtin_popcount(dirty);
    const int dincl = wave_incl_sum_i32(dl);
    const u64 pincl = wave_incl_sum_u64(psum);
    if (lane == 63) {
        sg.dtot[wave] = dincl;
        sg.ptot[wave] = pincl;
    }
    __syncthreads();                                                                          // (2)
#ifdef FK_OP_CLOCKS
    if (tid == 0) sg.dbg_t[1] = clock64();
#endif
    int dbase = dincl - dl, D = 0;
    u64 pbase = pincl - psum, ptotal = 0;
    FK_UNROLL for (int wv = 0; wv < OP_THREADS / 64; ++wv) {
        if (wv < wave) {
            dbase += sg.dtot[wv];
            pbase += sg.ptot[wv];
        }
        D += sg.dtot[wv];
        ptotal += sg.ptot[wv];
    }
    R.dirty = dirty;
    R.claims = claims;
    R.dbase = dbase;
    R.D = D;
    R.pbase = pbase;
    R.ptotal = ptotal;
    if (D > SEG_DMAX) return false;                                        // uniform
    // the dirty list and the segments' claims (all claimants of a segment write the same value -- checked below)
    {
        int r = dbase;
        u64 ps = pbase;
        FK_UNROLL for (int q = 0; q < OP_ITEMS; ++q) {
            if (claims & (1u << q)) {
                sg.seg_e[r] = eq[q];
                ps += (u64)floor(scale2(w[q], -eq[q]) + 0.5);
            }
            if (dirty & (1u << q)) {
                sg.d_pos[r] = tid * OP_ITEMS + q;
                sg.d_w[r] = w[q];
                sg.d_ps[r] = ps;
                ++r;
            }
        }
    }
    __syncthreads();                                                                          // (3)
#ifdef FK_OP_CLOCKS
    if (tid == 0) { sg.dbg_t[2] = clock64(); sg.dbg_D = D; }
#endif
    {
        int r = dbase, bad = 0;
        FK_UNROLL for (int q = 0; q < OP_ITEMS; ++q) {
            if ((claims & (1u << q)) && sg.seg_e[r] != eq[q]) bad = 1;    // two binades claimed inside one segment
            if (dirty & (1u << q)) ++r;
        }
        if (bad) sg.fail = 1;
    }
    if (SYNC_CLAIMS) __syncthreads();                                                         // (3b)
#ifdef FK_OP_CLOCKS
    if (tid == 0) sg.dbg_t[3] = clock64();
#endif
    return true;
}

// ---- the chain over the segments (wave 0; lane r holds segment r and dirty element r) -------------------------
// Everything that does not depend on the running sum is prepared by the lanes in parallel -- the segment's whole
// increment sum as a double in units of ONE (I 2^e: exact, a power-of-two scaling) and the exponent field its
// running sum must show -- so that the serial part is two dependent fp64 adds per segment:
//   c + I 2^e  is exact when C0 + I < 2^53 (the sum is representable), and when it is not the rounded result is
//   >= 2^(e+53), the next binade (rounding is monotone, the bound is representable): the exponent field of the
//   result tells; the exponent field of c before the add is the claim itself.
// Returns the carry-out; failed (wave-uniform) = the walk contradicts a claim (the caller declines the scan).
__device__ __forceinline__ double seg_chain(OpShared &sh, double carry, int D, u64 ptotal, int lane, bool &failed)
{
    SegShared &sg = sh.seg;
    const u64 my_end = lane < D ? sg.d_ps[lane < SEG_DMAX ? lane : 0] : ptotal;
    const u64 my_start = (lane >= 1 && lane <= D) ? sg.d_ps[lane - 1] : 0;
    const int my_e = lane <= D ? sg.seg_e[lane] : SEG_NONE;
    const double my_w = lane < D ? sg.d_w[lane < SEG_DMAX ? lane : 0] : 0.0;
    const u64 I = my_end - my_start;                                   // (wrapping; < 2^53 for a segment that passes)
    const bool claim = my_e != SEG_NONE;
    int fail = (lane <= D && ((claim && !(I < (1ull << 53))) || (!claim && I != 0))) ? 1 : 0;
    const double my_add = claim ? scale2((double)I, my_e) : 0.0;      // I 2^e
    const int my_xf = claim ? my_e + 1075 : -1;                        // biased exponent of a sum with ulp 2^e (-1: any)
    double c = carry, r_c = 0.0, r_dcs = 0.0;
    for (int r = 0; r <= D; ++r) {                                     // uniform
        const double add = lane_bcast(my_add, r);
        const double wr = lane_bcast(my_w, r);
        const int xf = __builtin_amdgcn_readlane(my_xf, r);
        const int x0 = (int)((double_to_bits(c) >> 52) & 0x7ffu);
        if (lane == r) r_c = c;
        c = c + add;                                                   // exact, or out of the binade (see above)
        const int x1 = (int)((double_to_bits(c) >> 52) & 0x7ffu);
        fail |= (xf >= 0 && (x0 != xf || x1 != xf || xf <= 1075 - 900 || xf >= 1075 + 900 - 52)) ? 1 : 0;
        if (r < D) {
            c = c + wr;                                                // the real IEEE add of the dirty element
            if (lane == r) r_dcs = c;
        }
    }
    if (lane <= D) {
        sg.seg_c[lane] = r_c;
        sg.seg_ps0[lane] = my_start;
        if (lane < D) sg.d_cs[lane] = r_dcs;
    }
    failed = __builtin_amdgcn_ballot_w64(fail != 0) != 0;
    if (failed && lane == 0) sg.fail = 1;
    if (lane == 0) sg.carry_out = c;
    return c;
}

// step 4 (behind a barrier that follows seg_chain; sg.fail == 0): the cumulative sums into the tile, one more barrier
__device__ __forceinline__ void seg_finish(OpShared &sh, int len, const SegRegs &R)
{
    const int tid = threadIdx.x;
    SegShared &sg = sh.seg;
    int r = R.dbase;
    u64 ps = R.pbase;
    FK_UNROLL for (int q = 0; q < OP_ITEMS; ++q) {
        const int j = tid * OP_ITEMS + q;
        // (the weight once more from the thread's own tile slot, overwritten below: not kept in registers across the wait for the carry)
        if (R.claims & (1u << q)) ps += (u64)floor(scale2(sh.tile[pad8(j)], -R.eq[q]) + 0.5);
        double cs;
        if (R.dirty & (1u << q)) {
            cs = sg.d_cs[r];
            ++r;
        } else {
            // (C0 + dPS) 2^e = c_start + dPS 2^e: exact for the same reason as in the chain
            const int e = sg.seg_e[r];
            cs = e == SEG_NONE ? sg.seg_c[r] : sg.seg_c[r] + scale2((double)(ps - sg.seg_ps0[r]), e);
        }
        if (j < len) sh.tile[pad8(j)] = cs;
    }
    __syncthreads();                                                                          // (5)
#ifdef FK_OP_CLOCKS
    if (tid == 0) sg.dbg_t[4] = clock64();
#endif
}

__device__ __forceinline__ bool segmented_cumsum(OpShared &sh, int len, double carry, double *c_out)
{
    // bound on the relative distance between the plain prefix sums and the sequential fp64 sums: both are within
    // (OP_TILE + 1) rounding errors of the real sum of non-negative terms, 2 * 2049 * 2^-53 < 2^-41; 2^-38 is safe
    constexpr double DELTA = 0x1p-38;
    SegRegs R;
    if (!seg_prepare<false>(sh, len, carry, DELTA, R)) return false;       // uniform
    if (threadIdx.x < 64) {
        bool failed;
        seg_chain(sh, carry, R.D, R.ptotal, (int)threadIdx.x, failed);
    }
    __syncthreads();                                                                          // (4)
#ifdef FK_OP_CLOCKS
    if (threadIdx.x == 0) sh.seg.dbg_t[3] = clock64();
#endif
    if (sh.seg.fail) return false;                                         // uniform; the tile is untouched
    seg_finish(sh, len, R);
    *c_out = sh.seg.carry_out;
    return true;
}

// Build-time instrumentation (tools/op_phase.py builds a separate library with -DFK_OP_CLOCKS; the shipped library
// has none of it): thread 0 of every workgroup adds the shader-clock ticks of each phase, and the polls its wave
// spent in the two look-backs, to fk_op_phase[]; fk_debug_op_phases() reads and clears them.
#ifdef FK_OP_CLOCKS
constexpr int OP_PHASE_SLOTS = 16, OP_PHASE_BUCKETS = 1024;
__device__ unsigned long long fk_op_phase[OP_PHASE_SLOTS][OP_PHASE_BUCKETS];
__device__ double *g_dbg_cs;     // [Fn][Np] dump of the cumulative sums / boundaries the kernel worked with (or null)
__device__ int *g_dbg_n;
__device__ unsigned long long *g_dbg_tl;   // [Fn * nch][16] wall-clock stamps (100 MHz) of a chunk's way through the kernel (or null)
#define OP_STAMP(slot) do { if (threadIdx.x == 0 && g_dbg_tl) g_dbg_tl[((long)f * nch + k) * 16 + (slot)] = wall_clock64(); } while (0)
#define OP_NOTE(slot, v) do { if (threadIdx.x == 0 && g_dbg_tl) g_dbg_tl[((long)f * nch + k) * 16 + (slot)] = (unsigned long long)(v); } while (0)
// (accumulators live in LDS, touched by thread 0 only: sixteen 64-bit counters in registers would spill)
#define OP_CLOCK_START() __shared__ long long t_acc[OP_PHASE_SLOTS]; long long t_prev = clock64(); \
    if (threadIdx.x == 0) for (int q_ = 0; q_ < OP_PHASE_SLOTS; ++q_) t_acc[q_] = 0
#define OP_CLOCK(slot) do { if (threadIdx.x == 0) { const long long t_now = clock64(); t_acc[slot] += t_now - t_prev; t_prev = t_now; } } while (0)
#define OP_COUNT(slot, n) do { if (threadIdx.x == 0) t_acc[slot] += (n); } while (0)
#define OP_CLOCK_FLUSH() do { if (threadIdx.x == 0) for (int q_ = 0; q_ < OP_PHASE_SLOTS; ++q_) \
        if (t_acc[q_]) atomicAdd(&fk_op_phase[q_][blockIdx.x % OP_PHASE_BUCKETS], (unsigned long long)t_acc[q_]); } while (0)
#else
#define OP_CLOCK_START() do { } while (0)
#define OP_CLOCK(slot) do { } while (0)
#define OP_COUNT(slot, n) do { } while (0)
#define OP_CLOCK_FLUSH() do { } while (0)
#define OP_STAMP(slot) do { } while (0)
#define OP_NOTE(slot, v) do { } while (0)
#endif

// ---- the kernel -------------------------------------------------------------------------------------------
struct OpArgs {
    long Np, nch;
    int Fn, nregions;
    const double *w, *u;
    int32_t *idx, *status;
    OpCtl *ctl;
    OpDesc *desc;
    int *bad;           // [Fn]: filter holds a weight the fast path does not take -> resample_literal_kernel
    double delta;       // relative error bound of the stage-1 prefix
    const unsigned *only_if;   // resample_local_kernel as the one-pass kernel's repair pass: run only if this word is non-zero
    int literal_first;         // ... and, in the same launch, in front of it: the filters the one-pass kernel declined (bad[f]) redone literally
};

#ifndef FK_OP_WAVES
#define FK_OP_WAVES 5          // workgroups per CU the register allocation aims at (<= 96 VGPRs, no scratch)
#endif

__device__ __forceinline__ void init_window(int *win, int tid)
{
    FK_UNROLL for (int g = 0; g < 3; ++g) *reinterpret_cast<i32x4 *>(&win[12 * tid + 4 * g]) = i32x4{-1, -1, -1, -1};
}

// TICKET = false (the default since round 3): chunk (f, k) is a function of blockIdx alone -- chunk-major over the filters --
// instead of an atomic ticket.  A chunk only waits for chunks with a smaller blockIdx.  Forward progress: every hardware
// dispatcher (one per XCD) starts its share of the grid in increasing blockIdx order, so the unfinished workgroup with the
// SMALLEST index is always resident (anything resident on its dispatcher was started before it, i.e. has a smaller index
// and would be the smallest) -- and it waits for nobody.  The order of dispatch is how the hardware walks a 1-D grid, not
// an architectural promise: the bounded spins + abort word (status ST_INTERNAL) stay, FK_OP_STATIC=0 brings the tickets back.
template <bool STRATIFIED, bool TICKET = true, bool SPEC = false>
__global__ void __launch_bounds__(OP_THREADS, FK_OP_WAVES)
resample_onepass_kernel(const OpArgs a)
{
    __shared__ OpShared sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long Np = a.Np, nch = a.nch;
    OP_CLOCK_START();

    // ---- ticket: chunk (f, k).  Filters are split into contiguous regions, one head each; a workgroup starts at
    // its home region (blockIdx % regions ~ its XCD) and moves on when that one is exhausted.  Inside a region the
    // tickets run chunk-major (chunk k of every filter of the region before chunk k + 1 of any): the resident
    // workgroups then advance all the region's filters together, and a chunk that has to resolve its carry with the
    // general scan (a binade crossing: ~20 per 8e6-particle vector) holds up its own filter's chain only.
    int *win = sh.win();
    int f, k;
    if constexpr (TICKET) {
    if (tid == 0) {
        const int R = a.nregions;
        int tf = -1, tk = 0;
        for (int s = 0; s < R; ++s) {
            const int r = (int)((blockIdx.x + (unsigned)s) % (unsigned)R);
            const long f_lo = (long)a.Fn * r / R, f_hi = (long)a.Fn * (r + 1) / R;
            const unsigned long cnt = (unsigned long)(f_hi - f_lo) * (unsigned long)nch;
            const unsigned t = __hip_atomic_fetch_add(&a.ctl->head[r].next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t < cnt) {
                const unsigned nf = (unsigned)(f_hi - f_lo);
                tf = (int)(f_lo + t % nf);
                tk = (int)(t / nf);
                break;
            }
        }
        sh.bc_i[1] = tf;
        sh.bc_i[2] = tk;
    }
    init_window(win, tid);                 // the output window (it shares its LDS with the general scan's tile)
    __syncthreads();
    f = __builtin_amdgcn_readfirstlane(sh.bc_i[1]);
    k = __builtin_amdgcn_readfirstlane(sh.bc_i[2]);
    } else {
        f = (int)(blockIdx.x % (unsigned)a.Fn);
        k = (int)(blockIdx.x / (unsigned)a.Fn);
        init_window(win, tid);             // (first read far behind barrier (A))
    }
    if (f < 0) return;                                                     // cannot happen: one workgroup per chunk
    OP_CLOCK(0);                                                           // ticket

    const double *wf = a.w + (long)f * Np;
    int32_t *of = a.idx + (long)f * Np;
    const double u_sys = STRATIFIED ? 0.0 : a.u[f];
    const double *u_str = STRATIFIED ? a.u + (long)f * Np : nullptr;
    const double Nd = (double)Np, halfNd = 0.5 * Nd;
    OpDesc *d = a.desc + (long)f * nch;
    unsigned *abort_word = &a.ctl->abort;
    const long base = (long)k * OP_TILE;
    const int len = (int)((Np - base) < OP_TILE ? (Np - base) : OP_TILE);

    // ---- weights: every thread reads its eight CONSECUTIVE weights straight from HBM (a wave covers 4 KiB with four
    // 16-byte loads per lane: every byte of every line is used, nothing is staged or transposed through LDS) ----
    double w8[OP_ITEMS];
    {
        const double *src = wf + base + tid * OP_ITEMS;
        if (len == OP_TILE && (((uintptr_t)(wf + base)) & 15) == 0) {                          // uniform
            FK_UNROLL for (int q = 0; q < OP_ITEMS; q += 2) {
                const f64x2 t = *reinterpret_cast<const f64x2 *>(src + q);
                w8[q] = t.x;
                w8[q + 1] = t.y;
            }
        } else if (len == OP_TILE) {
            FK_UNROLL for (int q = 0; q < OP_ITEMS; ++q) w8[q] = src[q];
        } else {
            FK_UNROLL for (int q = 0; q < OP_ITEMS; ++q) {
                const int j = tid * OP_ITEMS + q;
                const double t = wf[base + (j < len ? j : 0)];
                w8[q] = j < len ? t : 0.0;
            }
        }
    }
    double s = 0.0, mn = 0.0;
    FK_UNROLL for (int q = 0; q < OP_ITEMS; ++q) {
        s += w8[q];
        mn = w8[q] < mn ? w8[q] : mn;                                      // a negative weight (NaN / Inf show in the sum)
    }
    s = lane_bcast(wave_incl_sum(s), 63);
    if (lane == 0) sh.wsum[wave] = s;
    const int any_neg = __syncthreads_or(mn < 0.0 ? 1 : 0);                                  // (A)
    double S = (sh.wsum[0] + sh.wsum[1]) + (sh.wsum[2] + sh.wsum[3]);
    const bool any_bad = any_neg || !(S < 0x1p1000);                       // negative, NaN, Inf or absurdly large
    if (any_bad) S = __builtin_nan("");
    OP_CLOCK(1);                                                           // weights landed

    // ---- speculation (lookback_spec above): publish the increment sums for a guessed binade and the next one, look back once ----
    double E[OP_ITEMS];
    double excl = 0.0, I = 0.0;
    int eu = 0;
    bool hit = false;
    if constexpr (SPEC) {
        int g = 0;
        bool guess = false;
        if (k > 0 && !any_bad && S > 0.0) {                                // uniform
            // the k chunks before this one weigh about k times as much: the running sum enters near k S and leaves near
            // (k + 1) S; the lower end picks g, the next binade is the second candidate
            const double glo = (double)k * S * FK_OP_GUESS_LO, ghi = ((double)k + 1.0) * S * 1.04;
            guess = glo > OP_SANE_LO && ghi < OP_SANE_HI;
            g = ulp_exp(glo);
        }
        if (guess) {                                                       // uniform
            // Round 5 (the lever DESIGN section 9 carried since round 3): where the whole guess interval lies inside ONE binade
            // -- two chunks in three once k is past a few dozen: away from power-of-two carries -- the second candidate g + 1
            // cannot be the binade of this chunk's adds, so its increments, their tie test and their wave scan are not formed and
            // nothing is published at g + 1 (a successor that would have needed this chunk's sum there falls back to the
            // two-stage path like after any other miss; every shortcut is re-verified against the exact carry either way).
            const bool single = FK_OP_SPEC_SINGLE && __builtin_amdgcn_readfirstlane((int)(ulp_exp(((double)k + 1.0) * S * 1.04) == g)) != 0;   // uniform
            bool tie0 = false, tie1 = single;
            double run0 = 0.0, run1 = 0.0;
            if (single) {
                FK_UNROLL for (int q = 0; q < OP_ITEMS; ++q) {
                    const double x0 = scale2(w8[q], -g) + 0.5;
                    const double i0 = floor(x0);
                    tie0 = tie0 || (i0 == x0);
                    run0 += i0;
                    E[q] = run0;
                }
            } else {
                FK_UNROLL for (int q = 0; q < OP_ITEMS; ++q) {
                    const double t0 = scale2(w8[q], -g);
                    const double x0 = t0 + 0.5, x1 = t0 * 0.5 + 0.5;       // (w / 2^(g+1) = t0 / 2: exact)
                    const double i0 = floor(x0), i1 = floor(x1);
                    tie0 = tie0 || (i0 == x0);
                    tie1 = tie1 || (i1 == x1);
                    run0 += i0;
                    run1 += i1;
                    E[q] = run0;
                }
            }
            // (the same loop on multiplications by 2^-g and 2^52-rounding instead of v_ldexp / v_floor was measured: four more
            // VGPRs, 32 B of scratch, 3.36 -> 3.71 ms at 125 x 8e6 -- a spill in this kernel is a vmcnt(0) behind its stores)
            const double winc0 = wave_incl_sum(run0);
            const double tot1 = single ? 0.0 : lane_bcast(wave_incl_sum(run1), 63);
            const int tf = (__ballot(tie0) != 0 ? 1 : 0) | ((single || __ballot(tie1) != 0) ? 2 : 0);
            if (lane == 63) sh.wsum[wave] = winc0;
            if (lane == 0) {
                sh.seg.wtot[wave] = tot1;
                sh.wmax[wave] = tf;
            }
            __syncthreads();                                                                  // (S1)
            excl = winc0 - run0;
            double I0 = 0.0, I1 = 0.0;
            int ties = 0;
            FK_UNROLL for (int wv = 0; wv < OP_THREADS / 64; ++wv) {
                if (wv < wave) excl += sh.wsum[wv];
                I0 += sh.wsum[wv];
                I1 += sh.seg.wtot[wv];
                ties |= sh.wmax[wv];
            }
            const bool v0 = !(ties & 1) && I0 < 0x1p52 && I0 > 0.0, v1 = !(ties & 2) && I1 < 0x1p52 && I1 > 0.0;
            if (wave == 0) {
                if (lane == 0) {
                    st_agent(&d[k].approx, pack_approx(S, 1));
                    if (v0) st_agent(&d[k].exact, pack_exact(1, g, I0));
                    if (v1) st_agent(&d[k].exact2, pack_exact(1, g + 1, I1));
                }
                int which = -1;
                double c_in = 0.0;
                if (v0 || v1) c_in = lookback_spec(d, k, g, I0, I1, v0, v1, lane, which);
                int out_lo = 0;
                if (which >= 0) {
                    const int e = g + which;
                    const double Cout = scale2(c_in, -e) + (which ? I1 : I0);                  // < 2^53: lookback_spec checked it
                    if (lane == 0) {
                        publish_carry(&d[k], scale2(Cout, e));
                        st_agent(&d[k].approx, pack_approx(c_in + S, 2));  // what a successor on the two-stage path looks for
                    }
                    out_lo = n_boundary_fast<STRATIFIED>(c_in, (int)Np, Nd, halfNd, u_sys, u_str);
                }
                if (lane == 0) {
                    sh.bc_d[1] = c_in;
                    sh.bc_i[3] = which >= 0 ? 1 : 0;
                    sh.bc_i[4] = out_lo;
                    sh.bc_i[5] = which;
                }
            }
            __syncthreads();                                                                  // (S2)
            const int which = __builtin_amdgcn_readfirstlane(sh.bc_i[5]);
            hit = which >= 0;
            if (hit) {                                                     // uniform
                eu = g + which;
                I = which ? I1 : I0;
                if (which == 1) {                                          // the thread's inclusive sums at the other ulp
                    double run = 0.0;
                    FK_UNROLL for (int q = 0; q < OP_ITEMS; ++q) {
                        run += floor(scale2(w8[q], -eu) + 0.5);
                        E[q] = run;
                    }
                    const double winc = wave_incl_sum(run);
                    __syncthreads();                                       // (everybody has read wsum above)
                    if (lane == 63) sh.wsum[wave] = winc;
                    __syncthreads();                                                          // (S3)
                    excl = winc - run;
                    FK_UNROLL for (int wv = 0; wv < OP_THREADS / 64; ++wv)
                        if (wv < wave) excl += sh.wsum[wv];
                }
            }
            OP_CLOCK(2);                                                   // (speculation: counted as stage 1)
        }
    }
    bool at_start = false;
    if (!hit) {
    // ---- stage 1: approximate carry-in --------------------------------------------------------------------
    if (wave == 0) {
        double A = 0.0;
        if (k > 0) {
            if (lane == 0) st_agent(&d[k].approx, pack_approx(S, 1));
            A = lookback_approx(d, k, lane, abort_word);
        }
        if (lane == 0) {
            st_agent(&d[k].approx, pack_approx(A + S, 2));
            sh.bc_d[0] = A;
        }
    }
    __syncthreads();                                                                          // (B)
    const double A = sh.bc_d[0];
    OP_CLOCK(2);                                                           // stage 1

    // ---- what stage 1 tells this chunk --------------------------------------------------------------------
    const bool poison = !(A >= 0.0 && S >= 0.0 && A + S < 0x1p1000);
    if (poison) {                                                          // uniform
        // a weight this path does not take (negative, NaN, huge) here or upstream: the whole filter is redone by
        // resample_literal_kernel; successors only need to learn that quickly
        if (tid == 0) {
            publish_carry(&d[k], __builtin_nan(""));
            if (any_bad) __hip_atomic_store(&a.bad[f], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (a.status && __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                atomicOr(&a.status[f], ST_INTERNAL);
        }
        return;
    }
    // A == 0 means EXACTLY: every weight before this chunk is +0.0 (they are all >= 0), so the carry-in is 0 and
    // needs no stage 2; with S == 0 as well the chunk is empty-handed: carry-out 0, no slots
    at_start = (k == 0) || (A == 0.0);
    bool clean = false;
    {
        const double lo = A * (1.0 - a.delta), hi = (A + S) * (1.0 + a.delta);
        if (lo > OP_SANE_LO && hi < OP_SANE_HI && ulp_exp(lo) == ulp_exp(hi)) {
            clean = true;
            eu = ulp_exp(lo);
        }
    }
    // increments in the promised binade (fk_exact_scan.hpp, fast_inc): inc = floor(w / ulp + 1/2) unless the
    // remainder is exactly half an ulp -- such a chunk, and one whose sums reach 2^53 (which is also where t + 1/2
    // stops being exact, and then the test fires by itself), leaves this path.  E[q] = inclusive sums of the thread.
    excl = 0.0;
    I = 0.0;
    bool fast = false;
    if (clean) {                                                           // uniform
        bool tie = false;
        double run = 0.0;
        FK_UNROLL for (int q = 0; q < OP_ITEMS; ++q) {
            const double x = scale2(w8[q], -eu) + 0.5;
            const double i = floor(x);
            tie = tie || (i == x);
            run += i;
            E[q] = run;
        }
        const double winc = wave_incl_sum(run);
        if (lane == 63) sh.wsum[wave] = winc;
        const int any_tie = __syncthreads_or(tie ? 1 : 0);                                    // (C)
        excl = winc - run;                                                 // exact: everything is an integer < 2^53 ...
        FK_UNROLL for (int wv = 0; wv < OP_THREADS / 64; ++wv) {
            if (wv < wave) excl += sh.wsum[wv];
            I += sh.wsum[wv];
        }
        fast = !any_tie && I < 0x1p53;                                     // ... or this says so
    } else {
        FK_UNROLL for (int q = 0; q < OP_ITEMS; ++q) E[q] = 0.0;
    }
    OP_CLOCK(3);                                                           // increments

    // ---- stage 2: exact carry-in ---------------------------------------------------------------------------
    if (wave == 0) {
        double c_in = 0.0;
        bool ok = true;
        if (!at_start) {
            if (fast && lane == 0) st_agent(&d[k].exact, pack_exact(1, eu, I));
            c_in = lookback_exact(d, k, fast, eu, lane, abort_word, ok);
        }
        // carry-out of a chunk that stayed inside its binade: known right now, successors need not wait for the scan
        int quick = 0;
        if (ok && fast && c_in > OP_SANE_LO && c_in < OP_SANE_HI && ulp_exp(c_in) == eu) {
            const double Cout = scale2(c_in, -eu) + I;
            if (Cout < 0x1p53) {
                quick = 1;
                if (lane == 0) publish_carry(&d[k], scale2(Cout, eu));
            }
        }
        if (ok && at_start && S == 0.0) {
            quick = 2;                                                     // still nothing but zeros
            if (lane == 0) publish_carry(&d[k], 0.0);
        }
        // first slot of this chunk: everything below n(carry-in) belongs to earlier chunks
        const int out_lo = at_start ? 0 : n_boundary_fast<STRATIFIED>(c_in, (int)Np, Nd, halfNd, u_sys, u_str);
        if (lane == 0) {
            sh.bc_d[1] = c_in;
            sh.bc_i[3] = ok ? quick : -1;
            sh.bc_i[4] = out_lo;
        }
    }
    __syncthreads();                                                                          // (D)
    }   // !hit
    const double c_in = sh.bc_d[1];
    const int quick = __builtin_amdgcn_readfirstlane(sh.bc_i[3]);
    const int u_lo = __builtin_amdgcn_readfirstlane(sh.bc_i[4]);
    OP_CLOCK(4);                                                           // stage 2
    OP_COUNT(8 + (quick < 0 ? 3 : quick), 1);                              // 8: general, 9: quick, 10: zeros
    if (quick < 0) {                                                       // abort: a predecessor never published
        if (tid == 0 && a.status) atomicOr(&a.status[f], ST_INTERNAL);
        return;
    }

    // ---- slot boundaries: weight j owns the slots [n_{j-1}, n_j), n_j = n(cs_j)  (fk_resample_math.hpp) -----------
    int nb[OP_ITEMS];
    bool win_ready = true;
    if (quick == 1) {                                                      // uniform
        // cs_j = (C0 + E_j) ulp exactly, so N cs_j - u = fma(E_j, N ulp, C0 N ulp - u): ONE fma per weight gives the
        // estimate whose ceiling is n_j whenever it is not within eps of an integer (n_boundary_fast's argument:
        // here two roundings of 2^-22 slots each, the same budget); the rare rest takes the exact tests on cs_j.
        const double ulp = scale2(1.0, eu), C0 = scale2(c_in, -eu), Nu = scale2(Nd, eu);
        const double K = __builtin_fma(C0, Nu, STRATIFIED ? 0.0 : -u_sys);
        unsigned unsure = 0;
        FK_UNROLL for (int q = 0; q < OP_ITEMS; ++q) {
            const double Et = excl + E[q];                                 // exact
            const double e = __builtin_fma(Et, Nu, K);
            // floor(e) + 1 without a v_floor_f64 / v_cvt_i32_f64 pair: m = e + 1.5 2^52 holds the nearest integer
            // of e in its low mantissa bits (two's complement; -1 < e < Nd < 2^31 where the result is used), dd = e - nearest
            const double m = e + 0x1.8p52;
            const double dd = e - (m - 0x1.8p52);                          // exact, |dd| <= 1/2
            const int ri = (int)(unsigned)double_to_bits(m);
            bool sure = fabs(dd) > N_BOUNDARY_EPS && e < Nd;               // (= fr in (eps, 1 - eps))
            int n = ri + (dd < 0.0 ? 0 : 1);
            if (STRATIFIED) {
                const int fl = ri - (dd < 0.0 ? 1 : 0);
                const double fr = dd < 0.0 ? dd + 1.0 : dd;                // e - floor(e), exact
                const double uf = u_str[e < Nd ? fl : 0];                  // e >= 0 here
                const double gap = uf - fr;
                sure = sure && (gap > N_BOUNDARY_EPS || gap < -N_BOUNDARY_EPS);
                n = fl + (gap > 0.0 ? 0 : 1);
            }
            unsure |= sure ? 0u : (1u << q);
            nb[q] = n;
        }
        if (unsure) {                                                      // about one weight in 10^5: the exact tests
            // ONE inlined copy in a rolled loop; E[q] / nb[q] are picked and put back with selects (registers
            // cannot be indexed)
            _Pragma("nounroll") for (int q = 0; q < OP_ITEMS; ++q) {
                if (!(unsure & (1u << q))) continue;
                double Eq = E[0];
                FK_UNROLL for (int r = 1; r < OP_ITEMS; ++r) Eq = (q == r) ? E[r] : Eq;
                const int n = n_boundary<STRATIFIED>((C0 + (excl + Eq)) * ulp, (int)Np, Nd, halfNd, u_sys, u_str);
                FK_UNROLL for (int r = 0; r < OP_ITEMS; ++r) nb[r] = (q == r) ? n : nb[r];
            }
        }
#ifdef FK_OP_CLOCKS
        FK_UNROLL for (int q = 0; q < OP_ITEMS; ++q) {
            if (g_dbg_cs && tid * OP_ITEMS + q < len) {
                g_dbg_cs[(long)f * Np + base + tid * OP_ITEMS + q] = (C0 + (excl + E[q])) * ulp;
                g_dbg_n[(long)f * Np + base + tid * OP_ITEMS + q] = nb[q];
            }
        }
#endif
        OP_CLOCK(5);
    } else if (quick == 2) {
        FK_UNROLL for (int q = 0; q < OP_ITEMS; ++q) nb[q] = 0;
        OP_CLOCK(5);
    } else {
        // general scan: the weights once more (L2), now into the padded LDS tile the scan works in
        FK_UNROLL for (int q = 0; q < OP_ITEMS; ++q) {
            const int j = tid * OP_ITEMS + q;
            sh.tile[pad8(j)] = j < len ? wf[base + j] : 0.0;
        }
        __syncthreads();
        const double c_out = general_cumsum(sh, len, c_in);
        if (tid == 0) publish_carry(&d[k], c_out);
        OP_CLOCK(5);                                                       // general scan
        // (each thread reads back only its own slots, behind general_cumsum's final barrier; n_j goes into the low
        // half of cs_j's slot so that this loop stays rolled)
        int *nslot = reinterpret_cast<int *>(&sh.tile[pad8(tid * OP_ITEMS)]);
        _Pragma("nounroll") for (int q = 0; q < OP_ITEMS; ++q) {
            const int j = tid * OP_ITEMS + q;
            const double c = j < len ? sh.tile[pad8(j)] : c_out;
            const int n = n_boundary_fast<STRATIFIED>(c, (int)Np, Nd, halfNd, u_sys, u_str);
            nslot[2 * q] = n;
#ifdef FK_OP_CLOCKS
            if (g_dbg_cs && j < len) {
                g_dbg_cs[(long)f * Np + base + j] = c;
                g_dbg_n[(long)f * Np + base + j] = n;
            }
#endif
        }
        FK_UNROLL for (int q = 0; q < OP_ITEMS; ++q) nb[q] = nslot[2 * q];
        win_ready = false;                                                 // the tile overwrote the window
    }
    sh.nlast[tid] = nb[OP_ITEMS - 1];
    __syncthreads();                                                                          // (E)
    int nprev = tid == 0 ? u_lo : sh.nlast[tid - 1];
    const int u_hi = __builtin_amdgcn_readfirstlane(sh.nlast[OP_THREADS - 1]);
    // heads: the slot where each non-empty run starts (boundaries are non-decreasing for valid input)
    int head[OP_ITEMS];
    FK_UNROLL for (int q = 0; q < OP_ITEMS; ++q) {
        head[q] = nb[q] > nprev ? nprev : -1;
        nprev = nb[q] > nprev ? nb[q] : nprev;
    }
    OP_CLOCK(6);                                                           // boundaries

    // ---- emission: windows of OP_WIN slots; heads -> inclusive max-scan -> coalesced 16-byte stores --------------
    // (slot numbers fit an int: Np < 2^31; everything about a window is wave-uniform and lives in SGPRs)
    const int mis = (int)(((uintptr_t)of >> 2) & 3);
    int seed = -1;
    for (int wb = u_lo - ((mis + u_lo) & 3); u_hi > u_lo && wb < u_hi; wb += OP_WIN) {        // uniform
        if (!win_ready) {
            __syncthreads();                                               // everybody is done with the tile / the last window
            init_window(win, tid);
            __syncthreads();
        }
        win_ready = false;
        int have = 0;
        FK_UNROLL for (int q = 0; q < OP_ITEMS; ++q) {
            const unsigned h = (unsigned)(head[q] - wb);                   // a head below wb wraps to a huge number
            if (head[q] >= 0 && h < (unsigned)OP_WIN) {
                win[h] = (int)base + tid * OP_ITEMS + q;
                have = 1;
            }
        }
        const int any_head = __syncthreads_or(have);                                          // (G)
        const int lim = u_hi - wb;                                         // slots [max(first, 0), lim) are ours
        const int first = u_lo - wb;
        int32_t *ow = of + wb;                                             // uniform; 16-byte aligned
        const int s0 = 12 * tid;
        int x[12];
        if (any_head) {                                                    // uniform
            FK_UNROLL for (int g = 0; g < 3; ++g) {
                const i32x4 t = *reinterpret_cast<const i32x4 *>(&win[s0 + 4 * g]);
                x[4 * g + 0] = t.x;
                x[4 * g + 1] = t.y;
                x[4 * g + 2] = t.z;
                x[4 * g + 3] = t.w;
            }
            FK_UNROLL for (int e = 1; e < 12; ++e) x[e] = x[e] > x[e - 1] ? x[e] : x[e - 1];
            const int wincl = wave_incl_max(x[11]);
            if (lane == 63) sh.wmax[wave] = wincl;
            __syncthreads();
            const int up = __shfl_up(wincl, 1, 64);
            int pre = (lane == 0 || up < seed) ? seed : up;               // everything before this thread, this wave
            FK_UNROLL for (int wv = 0; wv < OP_THREADS / 64; ++wv) {
                const int t = sh.wmax[wv];
                if (wv < wave) pre = pre > t ? pre : t;
                seed = seed > t ? seed : t;                                // every thread: running max after this window
            }
            FK_UNROLL for (int e = 0; e < 12; ++e) x[e] = x[e] > pre ? x[e] : pre;
        } else {
            // the whole window lies inside one run (a weight owning more than OP_WIN slots): constant fill
            FK_UNROLL for (int e = 0; e < 12; ++e) x[e] = seed;
        }
        if (s0 < lim) {
            FK_UNROLL for (int g = 0; g < 3; ++g) {
                const int sg = s0 + 4 * g;
                if (sg >= first && sg + 3 < lim) *reinterpret_cast<i32x4 *>(&ow[sg]) = i32x4{x[4 * g], x[4 * g + 1], x[4 * g + 2], x[4 * g + 3]};
                else {
                    FK_UNROLL for (int e = 0; e < 4; ++e)
                        if (sg + e >= first && sg + e < lim) ow[sg + e] = x[4 * g + e];
                }
            }
        }
    }
    OP_CLOCK(7);                                                           // emission
    OP_COUNT(11, 1);
    OP_CLOCK_FLUSH();

    // ---- end of the vector: positions >= cumsum[-1] (the reference raises IndexError, resampling.py:109,145) --
    if (k == nch - 1) {
        for (long i = (long)u_hi + tid; i < Np; i += OP_THREADS) of[i] = (int32_t)(Np - 1);
        if (tid == 0 && a.status && u_hi < (int)Np) atomicOr(&a.status[f], ST_OVERRUN);
    }
}

#include "resample_onepass2.inc"

// ---- the chunk as round 3's kernel resolves it when the speculation does not apply: stage 1 -> stage 2 -> boundaries (quick /
// zeros / general scan) -> emission -> end of the vector.  A verbatim copy of resample_onepass_kernel's body from "stage 1" on
// (that kernel stays as it is: FK_OP_V2=0 is the A/B), out of line: resample_onepass2_kernel calls it for the chunks its fast
// path declines -- a miss of the guess, a tie, the start of a vector, the last chunk, garbage -- so that those never cost the
// fast path a register.  The eight weights are read again (L2); everything else is recomputed from them.
template <bool STRATIFIED, int BUDGET>
__device__ OP2_SLOW_INLINE void op_chunk_slow(OpShared &sh, const double *a_w, const double *a_u, int32_t *a_idx, int32_t *a_status,
                                                        OpCtl *a_ctl, OpDesc *a_desc, int *a_bad, const double a_delta, const long a_Np,
                                                        const long a_nch, const int f, const int k, const double S_in, const int any_bad_in,
                                                        const int lb_mode)
{
    // (the arguments one by one: an OpArgs by value would travel through scratch memory, written at the top of the CALLER)
    OpArgs a = {};
    a.w = a_w; a.u = a_u; a.idx = a_idx; a.status = a_status; a.ctl = a_ctl; a.desc = a_desc; a.bad = a_bad;
    a.delta = a_delta; a.Np = a_Np; a.nch = a_nch;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long Np = a.Np, nch = a.nch;
    int *win = sh.win();
    const double *wf = a.w + (long)f * Np;
    int32_t *of = a.idx + (long)f * Np;
    const double u_sys = STRATIFIED ? 0.0 : a.u[f];
    const double *u_str = STRATIFIED ? a.u + (long)f * Np : nullptr;
    const double Nd = (double)Np, halfNd = 0.5 * Nd;
    OpDesc *d = a.desc + (long)f * nch;
    unsigned *abort_word = &a.ctl->abort;
    const long base = (long)k * OP_TILE;
    const int len = (int)((Np - base) < OP_TILE ? (Np - base) : OP_TILE);
    // (the weights once more, from L2.  Handing the caller's registers over -- the path is inlined -- was tried: the allocator then
    //  spills in the FAST path's emission loop, 62 -> 90 spilled registers; the second trip costs a slow chunk ~2 us)
    double w8[OP_ITEMS];
    FK_UNROLL for (int q = 0; q < OP_ITEMS; ++q) {
        const int j = tid * OP_ITEMS + q;
        const double t = wf[base + (j < len ? j : 0)];
        w8[q] = j < len ? t : 0.0;
    }
    // (lb_mode & 8: 64 predecessors per look-back step instead of 16 -- calls with few long vectors walk hundreds of chunks back.
    //  Measured with and without, and with s_setprio 3 on this path: 1 x 8e6 108 / 110 / 111 us, nothing -- profiles/r06/onepass/r06o)
    const int lbw = (lb_mode & 8) ? 64 : OP_LB;
    const bool any_bad = any_bad_in != 0;
    const double S = any_bad ? __builtin_nan("") : S_in;
    double E[OP_ITEMS];
    double excl = 0.0, I = 0.0;
    int eu = 0;
    const bool hit = false;
    __syncthreads();                                                       // everybody has left the fast path's LDS slots
    OP_STAMP(5);
    init_window(win, tid);
    bool at_start = false;
    SegRegs R;
    bool prepared = false, scan_tried = false;
    if (!hit) {
    // ---- stage 1: approximate carry-in --------------------------------------------------------------------
    if (wave == 0) {
        double A = 0.0;
        if (k > 0) {
            if (lane == 0) st_agent(&d[k].approx, pack_approx(S, 1));
            A = lookback_approx(d, k, lane, abort_word, lbw);
        }
        if (lane == 0) {
            st_agent(&d[k].approx, pack_approx(A + S, 2));
            sh.bc_d[0] = A;
        }
    }
    __syncthreads();                                                                          // (B)
    OP_STAMP(6);
    const double A = sh.bc_d[0];

    // ---- what stage 1 tells this chunk --------------------------------------------------------------------
    const bool poison = !(A >= 0.0 && S >= 0.0 && A + S < 0x1p1000);
    if (poison) {                                                          // uniform
        // a weight this path does not take (negative, NaN, huge) here or upstream: the whole filter is redone by
        // resample_literal_kernel; successors only need to learn that quickly
        if (tid == 0) {
            publish_carry(&d[k], __builtin_nan(""));
            if (any_bad) __hip_atomic_store(&a.bad[f], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (a.status && __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                atomicOr(&a.status[f], ST_INTERNAL);
        }
        return;
    }
    // A == 0 means EXACTLY: every weight before this chunk is +0.0 (they are all >= 0), so the carry-in is 0 and
    // needs no stage 2; with S == 0 as well the chunk is empty-handed: carry-out 0, no slots
    at_start = (k == 0) || (A == 0.0);
    bool clean = false;
    {
        const double lo = A * (1.0 - a.delta), hi = (A + S) * (1.0 + a.delta);
        if (lo > OP_SANE_LO && hi < OP_SANE_HI && ulp_exp(lo) == ulp_exp(hi)) {
            clean = true;
            eu = ulp_exp(lo);
        }
    }
    // increments in the promised binade (fk_exact_scan.hpp, fast_inc): inc = floor(w / ulp + 1/2) unless the
    // remainder is exactly half an ulp -- such a chunk, and one whose sums reach 2^53 (which is also where t + 1/2
    // stops being exact, and then the test fires by itself), leaves this path.  E[q] = inclusive sums of the thread.
    excl = 0.0;
    I = 0.0;
    bool fast = false;
    if (clean) {                                                           // uniform
        bool tie = false;
        double run = 0.0;
        FK_UNROLL for (int q = 0; q < OP_ITEMS; ++q) {
            const double x = scale2(w8[q], -eu) + 0.5;
            const double i = floor(x);
            tie = tie || (i == x);
            run += i;
            E[q] = run;
        }
        const double winc = wave_incl_sum(run);
        if (lane == 63) sh.wsum[wave] = winc;
        const int any_tie = __syncthreads_or(tie ? 1 : 0);                                    // (C)
        excl = winc - run;                                                 // exact: everything is an integer < 2^53 ...
        FK_UNROLL for (int wv = 0; wv < OP_THREADS / 64; ++wv) {
            if (wv < wave) excl += sh.wsum[wv];
            I += sh.wsum[wv];
        }
        fast = !any_tie && I < 0x1p53;                                     // ... or this says so
    } else {
        FK_UNROLL for (int q = 0; q < OP_ITEMS; ++q) E[q] = 0.0;
    }

    // ---- a chunk that needs the general scan prepares it NOW (lb_mode & 4): the weights into the tile, seg_prepare on the
    // APPROXIMATE carry-in of stage 1 (relative error <= a.delta; exactly 0 at the start of a vector) -- all of this before the
    // exact carry-in has arrived, while the chunks below are still being resolved.  What is left behind the carry is
    // seg_chain, a few dependent adds by wave 0, and the carry-out goes out at once: with few long vectors every crossing of
    // a binade and every tie-holder sits on the one chain all later chunks wait for, and it used to hold that chain for a
    // reload of its weights, five barriers and the whole scan -- 7 us per crossing, 19 crossings in a row at 1 x 2e6
    // (tools/op_timeline.py; profiles/r06/onepass/timeline_before.txt).
    if (!fast && (lb_mode & 4) && !(at_start && S == 0.0)) {               // uniform
        FK_UNROLL for (int q = 0; q < OP_ITEMS; ++q) sh.tile[pad8(tid * OP_ITEMS + q)] = w8[q];   // (0.0 behind len)
        __syncthreads();
        scan_tried = true;
        prepared = seg_prepare<true>(sh, len, at_start ? 0.0 : A, (at_start ? 0.0 : a.delta) + 0x1p-38, R);
#ifdef FK_OP_CLOCKS
        {   // (tools/op_timeline.py: core clocks / 16 between the barriers of seg_prepare, 16 bits each)
            const SegShared &sg = sh.seg;
            auto d16 = [](long long d) -> unsigned long long { d >>= 4; return (unsigned long long)(d < 0 ? 0 : (d > 65535 ? 65535 : d)); };
            OP_NOTE(2, d16(sg.dbg_t[0] - sg.dbg_in) | (d16(sg.dbg_t[1] - sg.dbg_t[0]) << 16) | (d16(sg.dbg_t[2] - sg.dbg_t[1]) << 32) | (d16(sg.dbg_t[3] - sg.dbg_t[2]) << 48));
        }
#endif
    }
    // ---- stage 2: exact carry-in ---------------------------------------------------------------------------
    OP_STAMP(7);
    if (wave == 0) {
        double c_in = 0.0;
        bool ok = true;
        if (!at_start) {
            if (fast && lane == 0) st_agent(&d[k].exact, pack_exact(1, eu, I));
            // (lb_mode & 2: a chunk that does NOT stay inside one binade -- a crossing -- still knows the binade of its carry-in from
            //  stage 1, and may add up the predecessors' increment sums in THAT binade down to the nearest carry-out like anybody
            //  else (the argument of lookback_exact is about the chunks in between, not about this one) instead of waiting for
            //  chunk k-1 to see that carry, add the same sums and publish: one global round trip less per binade on the chain
            //  every chunk of the binade above waits for)
            bool lb_ok = fast;
            int lb_eu = eu;
            if (!fast && (lb_mode & 2)) {
                const double lo = A * (1.0 - a.delta), hi = A * (1.0 + a.delta);
                if (lo > OP_SANE_LO && hi < OP_SANE_HI && ulp_exp(lo) == ulp_exp(hi)) {
                    lb_ok = true;
                    lb_eu = ulp_exp(lo);
                }
            }
            c_in = lookback_exact(d, k, lb_ok, lb_eu, lane, abort_word, ok, lbw);
        }
        // carry-out of a chunk that stayed inside its binade: known right now, successors need not wait for the scan
        int quick = 0;
        if (ok && fast && c_in > OP_SANE_LO && c_in < OP_SANE_HI && ulp_exp(c_in) == eu) {
            const double Cout = scale2(c_in, -eu) + I;
            if (Cout < 0x1p53) {
                quick = 1;
                if (lane == 0) publish_carry(&d[k], scale2(Cout, eu));
            }
        }
        if (ok && at_start && S == 0.0) {
            quick = 2;                                                     // still nothing but zeros
            if (lane == 0) publish_carry(&d[k], 0.0);
        }
        if (ok && prepared && sh.seg.fail == 0) {                          // (behind seg_prepare's last barrier: every thread's verdict)
            bool failed;
            const double c_out = seg_chain(sh, c_in, R.D, R.ptotal, lane, failed);
            if (!failed && lane == 0) publish_carry(&d[k], c_out);
            OP_STAMP(14);
        }
        // first slot of this chunk: everything below n(carry-in) belongs to earlier chunks
        const int out_lo = at_start ? 0 : n_boundary_fast<STRATIFIED>(c_in, (int)Np, Nd, halfNd, u_sys, u_str);
        if (lane == 0) {
            sh.bc_d[1] = c_in;
            sh.bc_i[3] = ok ? quick : -1;
            sh.bc_i[4] = out_lo;
        }
    }
    __syncthreads();                                                                          // (D)
    OP_STAMP(8);
    }   // !hit
    const double c_in = sh.bc_d[1];
    const int quick = __builtin_amdgcn_readfirstlane(sh.bc_i[3]);
    const int u_lo = __builtin_amdgcn_readfirstlane(sh.bc_i[4]);
    if (quick < 0) {                                                       // abort: a predecessor never published
        if (tid == 0 && a.status) atomicOr(&a.status[f], ST_INTERNAL);
        return;
    }

    // ---- slot boundaries: weight j owns the slots [n_{j-1}, n_j), n_j = n(cs_j)  (fk_resample_math.hpp) -----------
    int nb[OP_ITEMS];
    bool win_ready = true;
    if (quick == 1) {                                                      // uniform
        // cs_j = (C0 + E_j) ulp exactly, so N cs_j - u = fma(E_j, N ulp, C0 N ulp - u): ONE fma per weight gives the
        // estimate whose ceiling is n_j whenever it is not within eps of an integer (n_boundary_fast's argument:
        // here two roundings of 2^-22 slots each, the same budget); the rare rest takes the exact tests on cs_j.
        const double ulp = scale2(1.0, eu), C0 = scale2(c_in, -eu), Nu = scale2(Nd, eu);
        const double K = __builtin_fma(C0, Nu, STRATIFIED ? 0.0 : -u_sys);
        unsigned unsure = 0;
        FK_UNROLL for (int q = 0; q < OP_ITEMS; ++q) {
            const double Et = excl + E[q];                                 // exact
            const double e = __builtin_fma(Et, Nu, K);
            // floor(e) + 1 without a v_floor_f64 / v_cvt_i32_f64 pair: m = e + 1.5 2^52 holds the nearest integer
            // of e in its low mantissa bits (two's complement; -1 < e < Nd < 2^31 where the result is used), dd = e - nearest
            const double m = e + 0x1.8p52;
            const double dd = e - (m - 0x1.8p52);                          // exact, |dd| <= 1/2
            const int ri = (int)(unsigned)double_to_bits(m);
            bool sure = fabs(dd) > N_BOUNDARY_EPS && e < Nd;               // (= fr in (eps, 1 - eps))
            int n = ri + (dd < 0.0 ? 0 : 1);
            if (STRATIFIED) {
                const int fl = ri - (dd < 0.0 ? 1 : 0);
                const double fr = dd < 0.0 ? dd + 1.0 : dd;                // e - floor(e), exact
                const double uf = u_str[e < Nd ? fl : 0];                  // e >= 0 here
                const double gap = uf - fr;
                sure = sure && (gap > N_BOUNDARY_EPS || gap < -N_BOUNDARY_EPS);
                n = fl + (gap > 0.0 ? 0 : 1);
            }
            unsure |= sure ? 0u : (1u << q);
            nb[q] = n;
        }
        if (unsure) {                                                      // about one weight in 10^5: the exact tests
            // ONE inlined copy in a rolled loop; E[q] / nb[q] are picked and put back with selects (registers
            // cannot be indexed)
            _Pragma("nounroll") for (int q = 0; q < OP_ITEMS; ++q) {
                if (!(unsure & (1u << q))) continue;
                double Eq = E[0];
                FK_UNROLL for (int r = 1; r < OP_ITEMS; ++r) Eq = (q == r) ? E[r] : Eq;
                const int n = n_boundary<STRATIFIED>((C0 + (excl + Eq)) * ulp, (int)Np, Nd, halfNd, u_sys, u_str);
                FK_UNROLL for (int r = 0; r < OP_ITEMS; ++r) nb[r] = (q == r) ? n : nb[r];
            }
        }
#ifdef FK_OP_CLOCKS
        FK_UNROLL for (int q = 0; q < OP_ITEMS; ++q) {
            if (g_dbg_cs && tid * OP_ITEMS + q < len) {
                g_dbg_cs[(long)f * Np + base + tid * OP_ITEMS + q] = (C0 + (excl + E[q])) * ulp;
                g_dbg_n[(long)f * Np + base + tid * OP_ITEMS + q] = nb[q];
            }
        }
#endif
    } else if (quick == 2) {
        FK_UNROLL for (int q = 0; q < OP_ITEMS; ++q) nb[q] = 0;
    } else {
        double c_out = 0.0;
        if (prepared && sh.seg.fail == 0) {                                // uniform (behind barrier (D))
            seg_finish(sh, len, R);
            c_out = sh.seg.carry_out;                                      // (published by wave 0 the moment it was known)
        } else {
            // general scan: the weights into the padded LDS tile the scan works in (a prepared scan that declined left them there)
            if (!prepared) {
                __syncthreads();
                FK_UNROLL for (int q = 0; q < OP_ITEMS; ++q) sh.tile[pad8(tid * OP_ITEMS + q)] = w8[q];
            }
            __syncthreads();
            // (round 6: the one-round scan first -- binades predicted for all elements at once, then verified; round 3 went straight
            //  to general_cumsum, one block scan and four barriers per binade the running sum passes through: ~45 k clocks for the
            //  first chunk of a vector, ~20 k for a crossing)
            if (scan_tried || !segmented_cumsum(sh, len, c_in, &c_out)) c_out = general_cumsum(sh, len, c_in);
            if (tid == 0) publish_carry(&d[k], c_out);
        }
        OP_STAMP(9);
        OP_NOTE(15, (prepared ? 1 : 0) | (sh.seg.fail ? 2 : 0) | ((unsigned)R.D << 8));
        // (each thread reads back only its own slots, behind general_cumsum's final barrier; n_j goes into the low
        // half of cs_j's slot so that this loop stays rolled)
        int *nslot = reinterpret_cast<int *>(&sh.tile[pad8(tid * OP_ITEMS)]);
        _Pragma("nounroll") for (int q = 0; q < OP_ITEMS; ++q) {
            const int j = tid * OP_ITEMS + q;
            const double c = j < len ? sh.tile[pad8(j)] : c_out;
            const int n = n_boundary_fast<STRATIFIED>(c, (int)Np, Nd, halfNd, u_sys, u_str);
            nslot[2 * q] = n;
#ifdef FK_OP_CLOCKS
            if (g_dbg_cs && j < len) {
                g_dbg_cs[(long)f * Np + base + j] = c;
                g_dbg_n[(long)f * Np + base + j] = n;
            }
#endif
        }
        FK_UNROLL for (int q = 0; q < OP_ITEMS; ++q) nb[q] = nslot[2 * q];
        win_ready = false;                                                 // the tile overwrote the window
    }
    sh.nlast[tid] = nb[OP_ITEMS - 1];
    __syncthreads();                                                                          // (E)
    int nprev = tid == 0 ? u_lo : sh.nlast[tid - 1];
    const int u_hi = __builtin_amdgcn_readfirstlane(sh.nlast[OP_THREADS - 1]);
    // heads: the slot where each non-empty run starts (boundaries are non-decreasing for valid input)
    int head[OP_ITEMS];
    FK_UNROLL for (int q = 0; q < OP_ITEMS; ++q) {
        head[q] = nb[q] > nprev ? nprev : -1;
        nprev = nb[q] > nprev ? nb[q] : nprev;
    }

    // ---- emission: windows of OP_WIN slots; heads -> inclusive max-scan -> coalesced 16-byte stores --------------
    // (slot numbers fit an int: Np < 2^31; everything about a window is wave-uniform and lives in SGPRs)
    const int mis = (int)(((uintptr_t)of >> 2) & 3);
    int seed = -1;
    for (int wb = u_lo - ((mis + u_lo) & 3); u_hi > u_lo && wb < u_hi; wb += OP_WIN) {        // uniform
        if (!win_ready) {
            __syncthreads();                                               // everybody is done with the tile / the last window
            init_window(win, tid);
            __syncthreads();
        }
        win_ready = false;
        int have = 0;
        FK_UNROLL for (int q = 0; q < OP_ITEMS; ++q) {
            const unsigned h = (unsigned)(head[q] - wb);                   // a head below wb wraps to a huge number
            if (head[q] >= 0 && h < (unsigned)OP_WIN) {
                win[h] = (int)base + tid * OP_ITEMS + q;
                have = 1;
            }
        }
        const int any_head = __syncthreads_or(have);                                          // (G)
        const int lim = u_hi - wb;                                         // slots [max(first, 0), lim) are ours
        const int first = u_lo - wb;
        int32_t *ow = of + wb;                                             // uniform; 16-byte aligned
        const int s0 = 12 * tid;
        int x[12];
        if (any_head) {                                                    // uniform
            FK_UNROLL for (int g = 0; g < 3; ++g) {
                const i32x4 t = *reinterpret_cast<const i32x4 *>(&win[s0 + 4 * g]);
                x[4 * g + 0] = t.x;
                x[4 * g + 1] = t.y;
                x[4 * g + 2] = t.z;
                x[4 * g + 3] = t.w;
            }
            FK_UNROLL for (int e = 1; e < 12; ++e) x[e] = x[e] > x[e - 1] ? x[e] : x[e - 1];
            const int wincl = wave_incl_max(x[11]);
            if (lane == 63) sh.wmax[wave] = wincl;
            __syncthreads();
            const int up = __shfl_up(wincl, 1, 64);
            int pre = (lane == 0 || up < seed) ? seed : up;               // everything before this thread, this wave
            FK_UNROLL for (int wv = 0; wv < OP_THREADS / 64; ++wv) {
                const int t = sh.wmax[wv];
                if (wv < wave) pre = pre > t ? pre : t;
                seed = seed > t ? seed : t;                                // every thread: running max after this window
            }
            FK_UNROLL for (int e = 0; e < 12; ++e) x[e] = x[e] > pre ? x[e] : pre;
        } else {
            // the whole window lies inside one run (a weight owning more than OP_WIN slots): constant fill
            FK_UNROLL for (int e = 0; e < 12; ++e) x[e] = seed;
        }
        if (s0 < lim) {
            FK_UNROLL for (int g = 0; g < 3; ++g) {
                const int sg = s0 + 4 * g;
                if (sg >= first && sg + 3 < lim) *reinterpret_cast<i32x4 *>(&ow[sg]) = i32x4{x[4 * g], x[4 * g + 1], x[4 * g + 2], x[4 * g + 3]};
                else {
                    FK_UNROLL for (int e = 0; e < 4; ++e)
                        if (sg + e >= first && sg + e < lim) ow[sg + e] = x[4 * g + e];
                }
            }
        }
    }

    OP_STAMP(10);
    OP_NOTE(11, 100 + quick);
    // ---- end of the vector: positions >= cumsum[-1] (the reference raises IndexError, resampling.py:109,145) --
    if (k == nch - 1) {
        for (long i = (long)u_hi + tid; i < Np; i += OP_THREADS) of[i] = (int32_t)(Np - 1);
        if (tid == 0 && a.status && u_hi < (int)Np) atomicOr(&a.status[f], ST_OVERRUN);
    }
}

// ---- short vectors: one workgroup per filter, its chunks in sequence, the carry in a register ------------------------
// (Np < RS_PAR_MIN.  No tickets, no look-back, no workspace: the exact carry-in of a chunk is the carry-out of the one
// before, so its binade is KNOWN and most chunks take the integer-increment path straight away; the chunks that cross
// a binade, the start of the vector and tie-holders take segmented_cumsum, and only what that declines the round-by-
// round general_cumsum.  Before: one
// workgroup per filter walking 2048-weight tiles with an int64 Mono scan per binade and one binary search per output
// slot, 140 us for 1000 x 8000.)
template <bool STRATIFIED>
struct ChunkCtx {
    long Np;
    double Nd, halfNd, u_sys;
    const double *u_str;
};

// slot boundaries of a chunk that stayed inside the binade of its carry-in: cs_j = (C0 + excl + E_j) 2^eu
template <bool STRATIFIED>
__device__ __forceinline__ void quick_boundaries(const double (&E)[OP_ITEMS], double excl, double c_in, int eu,
                                                 const ChunkCtx<STRATIFIED> &cx, int (&nb)[OP_ITEMS])
{
    const double ulp = scale2(1.0, eu), C0 = scale2(c_in, -eu), Nu = scale2(cx.Nd, eu);
    const double K = __builtin_fma(C0, Nu, STRATIFIED ? 0.0 : -cx.u_sys);
    unsigned unsure = 0;
    FK_UNROLL for (int q = 0; q < OP_ITEMS; ++q) {
        const double Et = excl + E[q];                                     // exact
        const double e = __builtin_fma(Et, Nu, K);
        const double m = e + 0x1.8p52;                                     // (as in resample_onepass_kernel)
        const double dd = e - (m - 0x1.8p52);
        const int ri = (int)(unsigned)double_to_bits(m);
        bool sure = fabs(dd) > N_BOUNDARY_EPS && e < cx.Nd;
        int n = ri + (dd < 0.0 ? 0 : 1);
        if (STRATIFIED) {
            const int fl = ri - (dd < 0.0 ? 1 : 0);
            const double fr = dd < 0.0 ? dd + 1.0 : dd;
            const double uf = cx.u_str[e < cx.Nd ? fl : 0];                // e >= 0 here
            const double gap = uf - fr;
            sure = sure && (gap > N_BOUNDARY_EPS || gap < -N_BOUNDARY_EPS);
            n = fl + (gap > 0.0 ? 0 : 1);
        }
        unsure |= sure ? 0u : (1u << q);
        nb[q] = n;
    }
    if (unsure) {                                                          // about one weight in 10^5: the exact tests
        _Pragma("nounroll") for (int q = 0; q < OP_ITEMS; ++q) {
            if (!(unsure & (1u << q))) continue;
            double Eq = E[0];
            FK_UNROLL for (int r = 1; r < OP_ITEMS; ++r) Eq = (q == r) ? E[r] : Eq;
            const int n = n_boundary<STRATIFIED>((C0 + (excl + Eq)) * ulp, (int)cx.Np, cx.Nd, cx.halfNd, cx.u_sys, cx.u_str);
            FK_UNROLL for (int r = 0; r < OP_ITEMS; ++r) nb[r] = (q == r) ? n : nb[r];
        }
    }
}

// slot boundaries from the cumulative sums a scan left in the tile (slots >= len take the carry-out: no slots)
template <bool STRATIFIED>
__device__ __forceinline__ void tile_boundaries(OpShared &sh, int len, double c_out, const ChunkCtx<STRATIFIED> &cx, int tid,
                                                int (&nb)[OP_ITEMS])
{
    int *nslot = reinterpret_cast<int *>(&sh.tile[pad8(tid * OP_ITEMS)]);
    _Pragma("nounroll") for (int q = 0; q < OP_ITEMS; ++q) {
        const int j = tid * OP_ITEMS + q;
        const double c = j < len ? sh.tile[pad8(j)] : c_out;
        nslot[2 * q] = n_boundary_fast<STRATIFIED>(c, (int)cx.Np, cx.Nd, cx.halfNd, cx.u_sys, cx.u_str);
    }
    FK_UNROLL for (int q = 0; q < OP_ITEMS; ++q) nb[q] = nslot[2 * q];
}

// heads -> windows -> coalesced stores (the emission of resample_onepass_kernel); returns the chunk's last boundary
__device__ __forceinline__ int emit_slots(OpShared &sh, int *win, const int (&nb)[OP_ITEMS], int u_lo, long base,
                                          int32_t *of, int tid, bool win_ready)
{
    const int lane = tid & 63, wave = tid >> 6;
    sh.nlast[tid] = nb[OP_ITEMS - 1];
    __syncthreads();
    int nprev = tid == 0 ? u_lo : sh.nlast[tid - 1];
    const int u_hi = __builtin_amdgcn_readfirstlane(sh.nlast[OP_THREADS - 1]);
    int head[OP_ITEMS];
    FK_UNROLL for (int q = 0; q < OP_ITEMS; ++q) {
        head[q] = nb[q] > nprev ? nprev : -1;
        nprev = nb[q] > nprev ? nb[q] : nprev;
    }
    const int mis = (int)(((uintptr_t)of >> 2) & 3);
    int seed = -1;
    for (int wb = u_lo - ((mis + u_lo) & 3); u_hi > u_lo && wb < u_hi; wb += OP_WIN) {        // uniform
        if (!win_ready) {
            __syncthreads();
            init_window(win, tid);
            __syncthreads();
        }
        win_ready = false;
        int have = 0;
        FK_UNROLL for (int q = 0; q < OP_ITEMS; ++q) {
            const unsigned h = (unsigned)(head[q] - wb);
            if (head[q] >= 0 && h < (unsigned)OP_WIN) {
                win[h] = (int)base + tid * OP_ITEMS + q;
                have = 1;
            }
        }
        const int any_head = __syncthreads_or(have);
        const int lim = u_hi - wb;
        const int first = u_lo - wb;
        int32_t *ow = of + wb;
        const int s0 = 12 * tid;
        int x[12];
        if (any_head) {                                                    // uniform
            FK_UNROLL for (int g = 0; g < 3; ++g) {
                const i32x4 t = *reinterpret_cast<const i32x4 *>(&win[s0 + 4 * g]);
                x[4 * g + 0] = t.x;
                x[4 * g + 1] = t.y;
                x[4 * g + 2] = t.z;
                x[4 * g + 3] = t.w;
            }
            FK_UNROLL for (int e = 1; e < 12; ++e) x[e] = x[e] > x[e - 1] ? x[e] : x[e - 1];
            const int wincl = wave_incl_max(x[11]);
            if (lane == 63) sh.wmax[wave] = wincl;
            __syncthreads();
            const int up = __shfl_up(wincl, 1, 64);
            int pre = (lane == 0 || up < seed) ? seed : up;
            FK_UNROLL for (int wv = 0; wv < OP_THREADS / 64; ++wv) {
                const int t = sh.wmax[wv];
                if (wv < wave) pre = pre > t ? pre : t;
                seed = seed > t ? seed : t;
            }
            FK_UNROLL for (int e = 0; e < 12; ++e) x[e] = x[e] > pre ? x[e] : pre;
        } else {
            FK_UNROLL for (int e = 0; e < 12; ++e) x[e] = seed;
        }
        if (s0 < lim) {
            FK_UNROLL for (int g = 0; g < 3; ++g) {
                const int s4 = s0 + 4 * g;
                if (s4 >= first && s4 + 3 < lim) *reinterpret_cast<i32x4 *>(&ow[s4]) = i32x4{x[4 * g], x[4 * g + 1], x[4 * g + 2], x[4 * g + 3]};
                else {
                    FK_UNROLL for (int e = 0; e < 4; ++e)
                        if (s4 + e >= first && s4 + e < lim) ow[s4 + e] = x[4 * g + e];
                }
            }
        }
    }
    return u_hi;
}

__device__ __forceinline__ void load_chunk(double (&w8)[OP_ITEMS], const double *wf, long base, long Np, int tid)
{
    const int len = (int)((Np - base) < OP_TILE ? (Np - base) : OP_TILE);
    const double *src = wf + base + tid * OP_ITEMS;
    if (len == OP_TILE && (((uintptr_t)(wf + base)) & 15) == 0) {                              // uniform
        FK_UNROLL for (int q = 0; q < OP_ITEMS; q += 2) {
            const f64x2 t = *reinterpret_cast<const f64x2 *>(src + q);
            w8[q] = t.x;
            w8[q + 1] = t.y;
        }
    } else if (len == OP_TILE) {
        FK_UNROLL for (int q = 0; q < OP_ITEMS; ++q) w8[q] = src[q];
    } else {
        FK_UNROLL for (int q = 0; q < OP_ITEMS; ++q) {
            const int j = tid * OP_ITEMS + q;
            const double t = wf[base + (j < len ? j : 0)];
            w8[q] = j < len ? t : 0.0;
        }
    }
}

// WAVES = workgroups per CU the registers are budgeted for.  Looped, the chunk body wants ~165 VGPRs: three per CU run
// without a spill and are the faster ones while the chip is not full (125 x 8000: 39 vs 46 us); four per CU spill ~30
// registers but hold 1024 filters at once -- one round instead of two for BASELINE configs[4]'s 1000 x 8000 (68 vs
// 80 us; profiles/r02/resample_local_variants.log).  The launcher picks by the filter count.
template <bool STRATIFIED>
__device__ void literal_filter(const OpArgs &a, const int f);
#ifndef FK_LOCAL_PREFETCH
#define FK_LOCAL_PREFETCH 0    // 1: the next chunk's weights are requested before the current chunk is worked on (measured: no gain)
#endif
template <bool STRATIFIED, int WAVES>
__global__ void __launch_bounds__(OP_THREADS, WAVES)
resample_local_kernel(const OpArgs a)
{
    __shared__ OpShared sh;
    // behind the one-pass kernel, ONE launch for both follow-ups (round 6; two launches before: ~5 us of a call whose kernel takes
    // 100): first the filter the one-pass kernel declined, redone literally by one thread (what resample_literal_kernel does) ...
    if (a.literal_first && a.bad[blockIdx.x]) {                            // uniform
        if (threadIdx.x == 0) literal_filter<STRATIFIED>(a, (int)blockIdx.x);
        __syncthreads();
    }
    // ... then the repair pass of onepass_launch: nothing to do unless a hand-off of the one-pass kernel timed out (uniform exit)
    if (a.only_if && __hip_atomic_load(a.only_if, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;
    const long Np = a.Np, nch = a.nch;
    const int f = blockIdx.x;
    const double *wf = a.w + (long)f * Np;
    int32_t *of = a.idx + (long)f * Np;
    ChunkCtx<STRATIFIED> cx;
    cx.Np = Np;
    cx.Nd = (double)Np;
    cx.halfNd = 0.5 * cx.Nd;
    cx.u_sys = STRATIFIED ? 0.0 : a.u[f];
    cx.u_str = STRATIFIED ? a.u + (long)f * Np : nullptr;
    int *win = sh.win();
    double carry = 0.0;
    int u_lo = 0;
    bool bad = false;
    OP_CLOCK_START();
#if FK_LOCAL_PREFETCH
    double wn[OP_ITEMS];
    load_chunk(wn, wf, 0, Np, (int)threadIdx.x);
#endif
    for (long k = 0; k < nch; ++k) {                                       // uniform
        // (the thread index is re-made opaque every round: otherwise everything derived from it -- a dozen offsets and
        // masks -- is hoisted out of the loop and held in registers across it)
        int tid_opaque = threadIdx.x;
        asm volatile("" : "+v"(tid_opaque));
        const int tid = tid_opaque, lane = tid & 63, wave = tid >> 6;
        const long base = k * OP_TILE;
        const int len = (int)((Np - base) < OP_TILE ? (Np - base) : OP_TILE);
        double w8[OP_ITEMS];
#if FK_LOCAL_PREFETCH
        FK_UNROLL for (int q = 0; q < OP_ITEMS; ++q) w8[q] = wn[q];
        if (k + 1 < nch) load_chunk(wn, wf, base + OP_TILE, Np, tid);
#else
        load_chunk(w8, wf, base, Np, tid);
#endif

        init_window(win, tid);
        double s = 0.0, mn = 0.0;
        FK_UNROLL for (int q = 0; q < OP_ITEMS; ++q) {
            s += w8[q];
            mn = w8[q] < mn ? w8[q] : mn;
        }
        s = lane_bcast(wave_incl_sum(s), 63);
        if (lane == 0) sh.seg.wtot[wave] = s;
        const int any_neg = __syncthreads_or(mn < 0.0 ? 1 : 0);
        OP_CLOCK(1);                                                       // weights landed, chunk sum
        const double S = (sh.seg.wtot[0] + sh.seg.wtot[1]) + (sh.seg.wtot[2] + sh.seg.wtot[3]);
        if (any_neg || !(S < 0x1p1000)) {                                  // negative, NaN, Inf or absurdly large
            bad = true;
            break;
        }
        int nb[OP_ITEMS];
        bool win_ready = true;
        double c_out = carry;
        if (carry == 0.0 && S == 0.0) {
            // nothing but zeros so far (carry == 0 means exactly that: the weights are >= 0): no slots
            FK_UNROLL for (int q = 0; q < OP_ITEMS; ++q) nb[q] = 0;
        } else {
            bool fast = false;
            double E[OP_ITEMS], excl = 0.0, I = 0.0;
            int eu = 0;
            if (carry > OP_SANE_LO && carry < OP_SANE_HI) {                // uniform
                eu = ulp_exp(carry);
                bool tie = false;
                double run = 0.0;
                FK_UNROLL for (int q = 0; q < OP_ITEMS; ++q) {
                    const double x = scale2(w8[q], -eu) + 0.5;
                    const double i = floor(x);
                    tie = tie || (i == x);
                    run += i;
                    E[q] = run;
                }
                const double winc = wave_incl_sum(run);
                if (lane == 63) sh.wsum[wave] = winc;
                const int any_tie = __syncthreads_or(tie ? 1 : 0);
                excl = winc - run;
                FK_UNROLL for (int wv = 0; wv < OP_THREADS / 64; ++wv) {
                    if (wv < wave) excl += sh.wsum[wv];
                    I += sh.wsum[wv];
                }
                fast = !any_tie && I < 0x1p53 && scale2(carry, -eu) + I < 0x1p53;
            }
            OP_CLOCK(3);                                                   // increments
            if (fast) {                                                    // uniform
                quick_boundaries<STRATIFIED>(E, excl, carry, eu, cx, nb);
                c_out = scale2(scale2(carry, -eu) + I, eu);
                OP_CLOCK(5);                                               // quick boundaries
                OP_COUNT(9, 1);
            } else {
                __syncthreads();                                           // (the window shares the tile's LDS)
                FK_UNROLL for (int q = 0; q < OP_ITEMS; ++q) sh.tile[pad8(tid * OP_ITEMS + q)] = w8[q];   // slots >= len: +0.0
                __syncthreads();
                if (!segmented_cumsum(sh, len, carry, &c_out)) {
                    c_out = general_cumsum(sh, len, carry);
                    OP_COUNT(10, 1);
                }
                OP_CLOCK(4);                                               // exact scan of the chunk
#ifdef FK_OP_CLOCKS
                if (tid == 0) {
                    t_acc[12] += sh.seg.dbg_t[1] - sh.seg.dbg_t[0];        // classification + scans
                    t_acc[13] += sh.seg.dbg_t[2] - sh.seg.dbg_t[1];        // lists
                    t_acc[14] += sh.seg.dbg_t[3] - sh.seg.dbg_t[2];        // claims check + chain
                    t_acc[15] += sh.seg.dbg_t[4] - sh.seg.dbg_t[3];        // cumulative sums into the tile
                    t_acc[0] += sh.seg.dbg_D;                              // (slot 0 is unused here: dirty elements)
                }
#endif
                tile_boundaries<STRATIFIED>(sh, len, c_out, cx, tid, nb);
                win_ready = false;
                OP_CLOCK(2);                                               // boundaries from the tile
                OP_COUNT(8, 1);
            }
        }
        u_lo = emit_slots(sh, win, nb, u_lo, base, of, tid, win_ready);
        carry = c_out;
        __syncthreads();                                                   // the LDS slots are free for the next chunk
        OP_CLOCK(7);                                                       // emission
        OP_COUNT(11, 1);
    }
    OP_CLOCK_FLUSH();
    const int tid = threadIdx.x;
    if (bad) {                                                             // uniform
        if (tid == 0) {
            const int st = literal_merge<STRATIFIED>(wf, STRATIFIED ? cx.u_str : a.u + f, Np, of);
            if (a.status) a.status[f] = st;
        }
        return;
    }
    // ---- end of the vector: positions >= cumsum[-1] (the reference raises IndexError, resampling.py:109,145) --
    for (long i = (long)u_lo + tid; i < Np; i += OP_THREADS) of[i] = (int32_t)(Np - 1);
    if (tid == 0 && a.status) a.status[f] = u_lo < (int)Np ? ST_OVERRUN : 0;
}

// ---- filters the one-pass kernel declined (a negative / NaN / huge weight): the reference's loop, literally -----
// cumulative_sum = np.cumsum(weights); i, j = 0, 0; while i < N: positions[i] < cumulative_sum[j] ? indexes[i] = j,
// i += 1 : j += 1   (resampling.py:106-112 / 142-149; j == N is the IndexError).  One thread: such input is
// garbage, but the answer is still the reference's.
template <bool STRATIFIED>
__device__ void literal_filter(const OpArgs &a, const int f)              // ONE thread
{
    const long Np = a.Np;
    const double *wf = a.w + (long)f * Np;
    int32_t *of = a.idx + (long)f * Np;
    const double Nd = (double)Np;
    long i = 0, j = 0;
    double c = wf[0];
    while (i < Np) {
        const double ui = STRATIFIED ? a.u[(long)f * Np + i] : a.u[f];
        const double p = (ui + (double)i) / Nd;
        if (p < c) {
            of[i] = (int32_t)j;
            ++i;
        } else {
            ++j;
            if (j == Np) break;
            c = c + wf[j];
        }
    }
    int st = 0;
    for (; i < Np; ++i) {
        of[i] = (int32_t)(Np - 1);
        st = ST_OVERRUN;
    }
    if (a.status) a.status[f] = st;
}

template <bool STRATIFIED>
__global__ void __launch_bounds__(64)
resample_literal_kernel(const OpArgs a)
{
    const int f = blockIdx.x;
    const long Np = a.Np;
    if (a.bad) {
        if (!a.bad[f]) return;
    } else {
        // no verdict from the one-pass kernel (short vectors take resample_kernel): look for a weight that path is
        // not defined on -- negative, NaN, Inf or absurdly large
        int bad = 0;
        for (long j = threadIdx.x; j < Np; j += 64) {
            const double v = a.w[(long)f * Np + j];
            bad |= !(v >= 0.0 && v < 0x1p1000);
        }
        if (!__syncthreads_or(bad)) return;
    }
    if (threadIdx.x != 0) return;
    literal_filter<STRATIFIED>(a, f);
}

#ifdef FK_OP_CLOCKS
__global__ void debug_n_boundary_kernel(int K, const double *c, int Np, double u, int *out, double *aux)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K) return;
    const double Nd = (double)Np;
    out[i] = n_boundary<false>(c[i], Np, Nd, 0.5 * Nd, u, nullptr);
    const double e = c[i] * Nd - u;
    const int b = e > 0.0 ? (int)e : 0;
    aux[4 * i + 0] = e;
    aux[4 * i + 1] = pos_ge(u + (double)b, c[i], Nd, 0.5 * Nd) ? 1.0 : 0.0;
    aux[4 * i + 2] = __builtin_fma(Nd, c[i], -(u + (double)b));
    aux[4 * i + 3] = (c[i] - bits_to_double(double_to_bits(c[i]) - 1)) * (0.5 * Nd);
}
}  // namespace fk
extern "C" int fk_debug_n_boundary(int K, const double *c, int Np, double u, int *out, double *aux)
{
    hipLaunchKernelGGL(fk::debug_n_boundary_kernel, dim3((K + 63) / 64), dim3(64), 0, 0, K, c, Np, u, out, aux);
    return hipDeviceSynchronize() == hipSuccess ? 0 : -1;
}
extern "C" int fk_debug_set_dump(double *cs, int *n)
{
    if (hipMemcpyToSymbol(HIP_SYMBOL(fk::g_dbg_cs), &cs, sizeof(cs)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(fk::g_dbg_n), &n, sizeof(n)) == hipSuccess ? 0 : -1;
}
extern "C" int fk_debug_set_timeline(unsigned long long *tl)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(fk::g_dbg_tl), &tl, sizeof(tl)) == hipSuccess ? 0 : -1;
}
extern "C" int fk_debug_op_phases(unsigned long long *out)     // only in the instrumented build (tools/op_phase.py)
{
    using namespace fk;
    static unsigned long long host[OP_PHASE_SLOTS][OP_PHASE_BUCKETS];
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(fk_op_phase), sizeof(host)) != hipSuccess) return FK_ERR_LAUNCH;
    for (int q = 0; q < OP_PHASE_SLOTS; ++q) {
        out[q] = 0;
        for (int b = 0; b < OP_PHASE_BUCKETS; ++b) out[q] += host[q][b];
    }
    memset(host, 0, sizeof(host));
    return hipMemcpyToSymbol(HIP_SYMBOL(fk_op_phase), host, sizeof(host)) == hipSuccess ? FK_OK : FK_ERR_LAUNCH;
}
namespace fk {
#endif

static size_t op_align(size_t v) { return (v + 255) & ~(size_t)255; }

// short vectors (resample_kernel): redo the filters that hold a weight that kernel is not defined on, literally
int literal_fixup_launch(bool stratified, int64_t Fn, int64_t Np, const double *w, const double *u, int32_t *idx,
                         int32_t *status, hipStream_t s)
{
    OpArgs a = {};
    a.Np = (long)Np;
    a.Fn = (int)Fn;
    a.w = w;
    a.u = u;
    a.idx = idx;
    a.status = status;
    a.bad = nullptr;
    if (stratified) hipLaunchKernelGGL((resample_literal_kernel<true>), dim3((unsigned)Fn), dim3(64), 0, s, a);
    else hipLaunchKernelGGL((resample_literal_kernel<false>), dim3((unsigned)Fn), dim3(64), 0, s, a);
    return check_launch("resample_literal_kernel");
}

int local_launch(bool stratified, int64_t Fn, int64_t Np, const double *w, const double *u, int32_t *idx,
                 int32_t *status, hipStream_t s)
{
    if (Fn > 0x7fffffffL) return FK_ERR_UNSUPPORTED;
    OpArgs a = {};
    a.Np = (long)Np;
    a.nch = (long)((Np + OP_TILE - 1) / OP_TILE);
    a.Fn = (int)Fn;
    a.w = w;
    a.u = u;
    a.idx = idx;
    a.status = status;
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                   ? prop.multiProcessorCount : 256;
    }
    const char *ev = getenv("FK_LOCAL_WAVES");
    const bool four = ev ? atoi(ev) == 4 : Fn > 3L * n_cu;                 // more filters than three per CU hold at once
    const dim3 grid((unsigned)Fn), block(OP_THREADS);
    if (stratified) {
        if (four) hipLaunchKernelGGL((resample_local_kernel<true, 4>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((resample_local_kernel<true, 3>), grid, block, 0, s, a);
    } else {
        if (four) hipLaunchKernelGGL((resample_local_kernel<false, 4>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((resample_local_kernel<false, 3>), grid, block, 0, s, a);
    }
    return check_launch("resample_local_kernel");
}

__global__ void __launch_bounds__(256) op_clear_kernel(u32x4 *ws, size_t n16, int32_t *status, long Fn)
{
    const size_t stride = (size_t)gridDim.x * 256u;
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n16; i += stride) ws[i] = u32x4{0u, 0u, 0u, 0u};
    if (status)
        for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < (size_t)Fn; i += stride) status[i] = 0;
}

size_t onepass_workspace_bytes(int64_t Fn, int64_t Np)
{
    if (Fn <= 0 || Np <= 0) return 0;
    const size_t nch = (size_t)((Np + OP_TILE - 1) / OP_TILE);
    return op_align(sizeof(OpCtl)) + op_align((size_t)Fn * nch * sizeof(OpDesc)) + op_align((size_t)Fn * sizeof(int));
}

int onepass_launch(bool stratified, int64_t Fn, int64_t Np, const double *w, const double *u, int32_t *idx,
                   int32_t *status, void *ws, size_t ws_bytes, hipStream_t s)
{
    const long nch = (long)((Np + OP_TILE - 1) / OP_TILE);
    const size_t need = onepass_workspace_bytes(Fn, Np);
    if (!ws || ws_bytes < need || (reinterpret_cast<uintptr_t>(ws) & 15u)) return FK_ERR_WORKSPACE;
    const unsigned long total = (unsigned long)Fn * (unsigned long)nch;
    if (total >= 0x7fffffffUL || Fn > 0x7fffffffL / 2) return FK_ERR_UNSUPPORTED;
    char *p = (char *)ws;
    OpArgs a = {};
    a.Np = (long)Np;
    a.nch = nch;
    a.Fn = (int)Fn;
    a.nregions = Fn < OP_REGIONS ? (int)Fn : OP_REGIONS;
    a.w = w;
    a.u = u;
    a.idx = idx;
    a.status = status;
    a.ctl = (OpCtl *)p;
    a.desc = (OpDesc *)(p + op_align(sizeof(OpCtl)));
    a.bad = (int *)(p + op_align(sizeof(OpCtl)) + op_align((size_t)Fn * nch * sizeof(OpDesc)));
    a.delta = (8.0 * (double)(Np + 4096) + 16.0 * (double)nch) * 0x1p-53;
    // the hand-off words and the status cleared in ONE launch (two memsets before: a launch each, ~3 us of a 100 us call)
    {
        const size_t n16 = need / 16;                                      // (need is a multiple of 256)
        const size_t blocks = (n16 + 1023) / 1024;
        hipLaunchKernelGGL(op_clear_kernel, dim3((unsigned)(blocks < 4096 ? (blocks ? blocks : 1) : 4096)), dim3(256), 0, s,
                           reinterpret_cast<u32x4 *>(ws), n16, status, (long)Fn);
    }
    // chunk assignment: static (blockIdx -> chunk, chunk-major over the filters) unless FK_OP_STATIC=0 asks for the atomic
    // tickets of round 2.  125 x 8e6: 4.43 -> 3.47 ms (profiles/r03/onepass_static_vs_ticket.jsonl): the ticket was a
    // dependent global round trip in front of the weight loads.
    const char *sv = getenv("FK_OP_STATIC");
    const bool stat = !(sv && sv[0] == '0');
    // speculation (lookback_spec): on unless FK_OP_SPEC=0 (A/B timing; the tests run both)
    const char *pv = getenv("FK_OP_SPEC");
    const bool spec = !(pv && pv[0] == '0');
    const dim3 grid((unsigned)total), block(OP_THREADS);
#define GO(STRAT)                                                                                                     \
    do {                                                                                                              \
        if (stat && spec) hipLaunchKernelGGL((resample_onepass_kernel<STRAT, false, true>), grid, block, 0, s, a);    \
        else if (stat) hipLaunchKernelGGL((resample_onepass_kernel<STRAT, false, false>), grid, block, 0, s, a);      \
        else if (spec) hipLaunchKernelGGL((resample_onepass_kernel<STRAT, true, true>), grid, block, 0, s, a);        \
        else hipLaunchKernelGGL((resample_onepass_kernel<STRAT, true, false>), grid, block, 0, s, a);                 \
        hipLaunchKernelGGL((resample_local_kernel<STRAT, 3>), dim3((unsigned)Fn), block, 0, s, r);                    \
    } while (0)
    // The repair pass (round 4).  The hand-offs between chunks are bounded spins; one that times out sets the abort word and
    // FK_STATUS_INTERNAL instead of hanging.  With tickets that cannot happen (a chunk only waits for chunks taken earlier);
    // with the static assignment it rests on the order in which the hardware starts the workgroups of a 1-D grid, which is
    // an observation, not a promise (CU masking, a co-tenant, a priority queue).  Instead of reporting the failure, the
    // call repairs itself: resample_local_kernel -- one workgroup per filter, chunks in sequence, no workgroup ever waits
    // for another -- runs behind the one-pass kernel and exits at once unless the abort word is set; then it recomputes
    // every filter of the call (indices and status, bit-identical to the one-pass result by the same exact arithmetic).
    // FK_OP_FORCE_ABORT=1 (tests) presets the abort word, so that every chunk that has to wait gives up.
    OpArgs r = a;
    r.only_if = &a.ctl->abort;
    r.literal_first = 1;
    if (const char *fv = getenv("FK_OP_FORCE_ABORT"); fv && fv[0] == '1') {
        static const unsigned one = 1u;
        if (hipMemcpyAsync(&a.ctl->abort, &one, sizeof(one), hipMemcpyHostToDevice, s) != hipSuccess) return FK_ERR_LAUNCH;
    }
    // Round 6: systematic calls on the static assignment with speculation run resample_onepass2_kernel (resample_onepass2.inc:
    // the same protocol, the common chunk in fewer instructions and barriers); FK_OP_V2=0 keeps round 3's kernel (A/B, tests).
    // pred_back: how far back a chunk looks for a PUBLISHED inclusive prefix to predict its binade from -- a chunk of the same
    // vector that finished before this one started: 1.25 rounds of resident workgroups back, at least 16 chunks.
    const char *v2 = getenv("FK_OP_V2");
    if (!stratified && stat && spec && !(v2 && v2[0] == '0')) {
        static int n_cu = 0;
        if (!n_cu) {
            int dev = 0;
            hipDeviceProp_t prop;
            n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                       ? prop.multiProcessorCount : 256;
        }
        long back = (5L * 7 * n_cu / 4 + Fn - 1) / Fn + 2;                 // (up to seven workgroups per CU resident)
        if (back < 16) back = 16;
        if (const char *pb = getenv("FK_OP_PRED_BACK")) back = atol(pb);   // (0: never predict)
        int polls = 48;                                                    // look-back polls before a chunk gives its guess up
        if (const char *pv2 = getenv("FK_OP_POLLS")) polls = atoi(pv2);
        // workgroups per CU (FK_OP_WAVES = 5 / 6 / 7): the kernel is latency-bound -- every chunk is a chain of load -> sums ->
        // publish -> look back -> boundaries -> emission --, so what a CU retires per microsecond is how many such chains it
        // interleaves
        int waves = FK_OP2_WAVES;
        if (const char *wv = getenv("FK_OP_WAVES")) waves = atoi(wv);
        if (waves < 5) waves = 5;
        if (waves > 7) waves = 7;
        const int pb = (int)(back > 0x3fffffff ? 0x3fffffff : back);
        // (FK_OP_LDS_PAD: bytes of dynamic LDS nobody uses -- lowers the workgroups a CU holds; occupancy experiments only:
        //  profiles/r06/onepass_occupancy.txt, time = 1.62 + 7.97 / (workgroups per CU) ms at 125 x 8e6)
        const unsigned lds_pad = getenv("FK_OP_LDS_PAD") ? (unsigned)atol(getenv("FK_OP_LDS_PAD")) : 0u;
        // FK_OP_LB: bit 0 -- a look-back waits for predecessors that published at another binade (they become carry-outs) instead of
        // missing; bit 1 -- a crossing chunk adds up the increment sums of the binade it enters from (op_chunk_slow); bit 2 -- the general scan
        // of a chunk is prepared on the approximate carry-in, the carry-out published a few adds behind the exact one.  The polls
        // are then bounded by the spin limit of the slow path, not by FK_OP_POLLS.
        // bit 2 costs a chunk that takes the general scan ~5 us of its own time (the scan in two pieces holds more across the wait)
        // and takes ~6 us per crossing off the chain behind it: on where many chunks of a vector are in flight at once (the chain
        // is what the call waits for), off where the chunks of a vector come one or two at a time -- 1000 x 1e5: 0.41 ms without,
        // 0.46 with; 32 x 1e6: 0.28 / 0.25; 1 x 8e6: 0.174 / 0.108 (profiles/r06/onepass/r06o/rs_ab.txt)
        int lb_mode = 3 | (Fn <= 128 ? 4 : 0);
        if (const char *lbv = getenv("FK_OP_LB")) lb_mode = atoi(lbv);
        if ((lb_mode & 1) && !getenv("FK_OP_POLLS")) polls = 1 << 16;
#define GO2(W) hipLaunchKernelGGL((resample_onepass2_kernel<W>), grid, block, lds_pad, s, a, pb, polls, lb_mode)
        if (waves == 5) GO2(5); else if (waves == 6) GO2(6); else GO2(7);
#undef GO2
        hipLaunchKernelGGL((resample_local_kernel<false, 3>), dim3((unsigned)Fn), block, 0, s, r);
        return check_launch("resample_onepass2_kernel");
    }
    if (stratified) GO(true);
    else GO(false);
#undef GO
    return check_launch("resample_onepass_kernel");
}

}  // namespace fk
