// resample_kernels.hip -- particle-filter resampling for gfx950 (MI355X).
//
//   fk_resample_systematic_f64  <- systematic_resample (filterpy/monte_carlo/resampling.py:117-150)
//   fk_resample_stratified_f64  <- stratified_resample (:80-114)
//   fk_resample_multinomial_f64 <- multinomial_resample (:153-176) and the random tail of
//                                  residual_resample (:72-76)
//   fk_cumsum_exact_f64         <- numpy.cumsum of a float64 vector, bit-for-bit
//
// One workgroup per filter walks the weight vector in tiles of RS_TILE weights.  Per tile:
//   1. coalesced load of the weights into LDS;
//   2. exact prefix sum (fk_exact_scan.hpp): a block-wide associative scan over int64 pairs
//      reproduces the sequential fp64 add chain bit-for-bit, restarting at binade crossings;
//      the tile's cumulative sums stay in LDS -- they are never written to HBM;
//   3. the positions are implicit, pos_i = fl(fl(u + i)/Np): the tile's last cumulative sum
//      determines the contiguous range of output slots it covers; every slot binary-searches
//      the LDS tile (idx_i = #{ j : cs_j <= pos_i }) and the int32 indices are written coalesced.
// Algorithmic HBM traffic: 8 B read + 4 B written per particle (+ 8 B per particle for the
// stratified uniforms).  This translation unit is compiled with -ffp-contract=off: positions
// and sums must be single IEEE operations.
#include <stdlib.h>
#include <string.h>

#include "../../include/filterhip.h"
#include "fk_device.hpp"
#include "fk_exact_scan.hpp"
#include "resample_onepass.hpp"

namespace fk {

constexpr int RS_THREADS = 256;
constexpr int RS_ITEMS = 8;
constexpr int RS_TILE = RS_THREADS * RS_ITEMS;   // 2048 weights per tile
constexpr int RS_PRELUDE = 128;                  // first non-zero-sum elements of a vector: sequential adds

struct ScanShared {
    // weights of the tile, then (in place) their cumulative sums, between the guards tile_upper_bound reads
    // without a bounds test: w()[-1] = -inf, w()[len ..] = +inf (load_tile)
    double tile[2 + RS_TILE + TILE_GUARD];
    __device__ __forceinline__ double *w() { return tile + 2; }
    Mono wave_tot[RS_THREADS / 64];
    double wave_sum[RS_THREADS / 64];
    double carry;               // exact running sum entering the next segment
    int first_cross;            // first tile index whose add leaves the binade (RS_TILE = none)
    int first_nonzero;
};

// weights of one tile -> LDS.  Slots past `len` and the guards hold +inf / -inf: the scan never reads them
// (every read there is bounded by len), the output search relies on them.
__device__ __forceinline__ void fetch_tile(double (&v)[RS_ITEMS], const double *__restrict__ src, int len)
{
    const int tid = threadIdx.x;
    // all RS_ITEMS loads are issued back to back: a load under a per-element bounds branch is followed by
    // its own vmcnt(0), i.e. RS_ITEMS serial HBM round trips per tile
    if (len == RS_TILE) {                                         // uniform
        FK_UNROLL for (int k = 0; k < RS_ITEMS; ++k) v[k] = src[tid + k * RS_THREADS];
    } else {
        FK_UNROLL for (int k = 0; k < RS_ITEMS; ++k) {
            const int j = tid + k * RS_THREADS;
            const double t = src[j < len ? j : 0];                // len >= 1: always a valid address
            v[k] = j < len ? t : __builtin_inf();
        }
    }
}

__device__ __forceinline__ void stage_tile(ScanShared &sh, const double (&v)[RS_ITEMS])
{
    const int tid = threadIdx.x;
    FK_UNROLL for (int k = 0; k < RS_ITEMS; ++k) sh.w()[tid + k * RS_THREADS] = v[k];
    if (tid < TILE_GUARD) sh.w()[RS_TILE + tid] = __builtin_inf();
    if (tid == 0) sh.w()[-1] = -__builtin_inf();
}

__device__ __forceinline__ void load_tile(ScanShared &sh, const double *__restrict__ src, int len)
{
    double v[RS_ITEMS];
    fetch_tile(v, src, len);
    stage_tile(sh, v);
}

__device__ __forceinline__ Mono shfl_up_mono(const Mono &m, int delta)
{
    Mono r;
    r.ae = __shfl_up(m.ae, delta, 64);
    r.ao = __shfl_up(m.ao, delta, 64);
    return r;
}

// Block-wide scan of Mono elements laid out ITEMS-per-thread (thread t owns elements
// t*ITEMS .. t*ITEMS+ITEMS-1).  In: loc[k] = the thread's running composites (inclusive within
// the thread).  Out: excl = composite of everything before the thread's first element.
// `wave_tot` is LDS scratch (RS_THREADS/64 entries); contains one __syncthreads().
template <int ITEMS>
__device__ __forceinline__ Mono block_mono_excl(const Mono (&loc)[ITEMS], Mono *wave_tot)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    Mono inc = loc[ITEMS - 1];
    FK_UNROLL for (int d = 1; d < 64; d <<= 1) {
        const Mono up = shfl_up_mono(inc, d);
        if (lane >= d) inc = mono_compose(up, inc);
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    Mono excl = shfl_up_mono(inc, 1);
    if (lane == 0) excl = mono_identity();
    Mono wprefix = mono_identity();
    for (int wv = 0; wv < wave; ++wv) wprefix = mono_compose(wprefix, wave_tot[wv]);
    return mono_compose(wprefix, excl);
}

// In-place exact inclusive prefix sum of sh.w()[0..len) continuing from the running sum `carry`
// (`started` = false means no element has been summed yet: cs[0] = w[0], like numpy.cumsum).
// All RS_THREADS threads participate.  Returns the running sum after the tile.
__device__ double tile_cumsum_exact(ScanShared &sh, int len, double carry, bool &started, int &prelude)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int pos = 0;
    while (pos < len) {   // uniform
        // --- serial fall-backs (one element), executed by every thread redundantly ----------
        const bool finite_pos = started && carry > 0.0 && carry <= 1.79769313486231570815e+308;
        if (!finite_pos) {
            if (!started || carry == 0.0) {
                // cumulative sum is still exactly zero (or empty): skip the run of zeros in parallel
                if (tid == 0) sh.first_nonzero = len;
                __syncthreads();
                int mine = len;
                for (int j = pos + tid; j < len; j += RS_THREADS)
                    if (sh.w()[j] != 0.0) { mine = j; break; }
                if (mine < len) atomicMin(&sh.first_nonzero, mine);
                __syncthreads();
                const int j0 = sh.first_nonzero;
                __syncthreads();
                // elements pos..j0-1 are zero weights: their cumulative sum is +0.0 (0+0, or w itself)
                if (j0 < len) {
                    // 0 + w = w exactly (also for negative / NaN w)
                    carry = started ? carry + sh.w()[j0] : sh.w()[j0];
                    started = true;
                    pos = j0 + 1;      // sh.w()[j0] already holds its own cumulative sum
                } else {
                    started = started || len > pos;
                    if (started && !(carry == 0.0)) carry = 0.0;
                    pos = len;
                }
                continue;
            }
            // negative, NaN or infinite running sum: plain sequential adds
            carry = carry + sh.w()[pos];
            __syncthreads();
            if (tid == 0) sh.w()[pos] = carry;
            __syncthreads();
            ++pos;
            continue;
        }

        // --- start of a vector: the running sum doubles after 1, 2, 4, ... elements, so nearly every
        // add crosses a binade; a short plain sequential chain is cheaper than one scan per crossing
        if (prelude > 0) {
            const int stop = (pos + prelude < len) ? pos + prelude : len;
            __syncthreads();
            if (tid == 0) {
                double c = carry;
                // blocks of 16: the LDS reads of a block are independent (issued back to back), only the
                // adds are a dependent chain; padding with +0.0 leaves the sum unchanged
                for (int j = pos; j < stop; j += 16) {
                    double v[16];
                    FK_UNROLL for (int k = 0; k < 16; ++k) v[k] = (j + k < stop) ? sh.w()[j + k] : 0.0;
                    FK_UNROLL for (int k = 0; k < 16; ++k) {
                        c = c + v[k];
                        v[k] = c;
                    }
                    FK_UNROLL for (int k = 0; k < 16; ++k)
                        if (j + k < stop) sh.w()[j + k] = v[k];
                }
                sh.carry = c;
            }
            __syncthreads();
            carry = sh.carry;
            prelude -= stop - pos;
            pos = stop;
            continue;
        }

        // --- one binade: parallel exact scan over [pos, len) ---------------------------------
        const double u = ulp_of(carry);
        const int eu = ulp_exp(carry);
        // fast path (fk_exact_scan.hpp, fast_inc): no half-ulp tie in the segment -> integer increments
        // held in doubles, the scan is a plain fp64 prefix sum
        {
            const double C0d = scale2(carry, -eu);
            double incl[RS_ITEMS];
            double runs = 0.0;
            bool tie = false;
            FK_UNROLL for (int k = 0; k < RS_ITEMS; ++k) {
                const int j = tid * RS_ITEMS + k;
                bool tk = false;
                const double e = (j >= pos && j < len) ? fast_inc(sh.w()[j], eu, tk) : 0.0;
                tie = tie || tk;
                runs += e;
                incl[k] = runs;
            }
            if (tid == 0) sh.first_cross = RS_TILE;
            const int any_tie = __syncthreads_or(tie ? 1 : 0);
            if (!any_tie) {
                double inc = runs;
                FK_UNROLL for (int d = 1; d < 64; d <<= 1) {
                    const double up = __shfl_up(inc, d, 64);
                    if (lane >= d) inc += up;
                }
                if (lane == 63) sh.wave_sum[wave] = inc;
                __syncthreads();
                double excl = __shfl_up(inc, 1, 64);
                if (lane == 0) excl = 0.0;
                FK_UNROLL for (int wv = 0; wv < RS_THREADS / 64; ++wv)
                    if (wv < wave) excl += sh.wave_sum[wv];
                double Cd[RS_ITEMS];
                int my_cross = RS_TILE;
                FK_UNROLL for (int k = 0; k < RS_ITEMS; ++k) {
                    const int j = tid * RS_ITEMS + k;
                    Cd[k] = C0d + (excl + incl[k]);
                    if (j >= pos && j < len && !(Cd[k] < 0x1p53) && my_cross == RS_TILE) my_cross = j;
                }
                if (my_cross < RS_TILE) atomicMin(&sh.first_cross, my_cross);
                __syncthreads();
                const int cross = sh.first_cross < len ? sh.first_cross : len;   // first element NOT covered
                FK_UNROLL for (int k = 0; k < RS_ITEMS; ++k) {
                    const int j = tid * RS_ITEMS + k;
                    if (j >= pos && j < cross) sh.w()[j] = Cd[k] * u;               // exact
                }
                __syncthreads();
                if (cross > pos) carry = sh.w()[cross - 1];
                if (cross < len) {
                    carry = carry + sh.w()[cross];      // the add that leaves the binade: a real IEEE add
                    __syncthreads();
                    if (tid == 0) sh.w()[cross] = carry;
                    __syncthreads();
                    pos = cross + 1;
                } else {
                    pos = len;
                }
                continue;
            }
        }
        // general path: Mono scan (handles exact ties)
        const long long C0 = (long long)scale2(carry, -eu);
        Mono loc[RS_ITEMS];
        Mono run = mono_identity();
        FK_UNROLL for (int k = 0; k < RS_ITEMS; ++k) {
            const int j = tid * RS_ITEMS + k;
            const Mono e = (j >= pos && j < len) ? mono_elem(sh.w()[j], u, eu) : mono_identity();
            run = mono_compose(run, e);
            loc[k] = run;
        }
        if (tid == 0) sh.first_cross = RS_TILE;
        const Mono excl = block_mono_excl<RS_ITEMS>(loc, sh.wave_tot);
        // element results, crossing detection
        long long Cj[RS_ITEMS];
        int my_cross = RS_TILE;
        FK_UNROLL for (int k = 0; k < RS_ITEMS; ++k) {
            const int j = tid * RS_ITEMS + k;
            Cj[k] = mono_apply(C0, mono_compose(excl, loc[k]));
            if (j >= pos && j < len && Cj[k] >= MONO_LIMIT && my_cross == RS_TILE) my_cross = j;
        }
        if (my_cross < RS_TILE) atomicMin(&sh.first_cross, my_cross);
        __syncthreads();
        const int cross = sh.first_cross < len ? sh.first_cross : len;   // first element NOT covered
        FK_UNROLL for (int k = 0; k < RS_ITEMS; ++k) {
            const int j = tid * RS_ITEMS + k;
            if (j >= pos && j < cross) sh.w()[j] = (double)Cj[k] * u;       // exact
        }
        __syncthreads();
        if (cross > pos) carry = sh.w()[cross - 1];
        if (cross < len) {
            // the add that leaves the binade: a real IEEE add
            carry = carry + sh.w()[cross];
            __syncthreads();
            if (tid == 0) sh.w()[cross] = carry;
            __syncthreads();
            pos = cross + 1;
        } else {
            pos = len;
        }
    }
    __syncthreads();
    return carry;
}

// Np < 2^31 (checked by the entry point: the indices are int32), so slot numbers are ints and
// (double)i is one conversion instead of the four of an int64
template <bool STRATIFIED>
__device__ __forceinline__ double position(int i, double dNp, double u_sys, const double *__restrict__ u_str)
{
    const double ui = STRATIFIED ? u_str[i] : u_sys;
    return (ui + (double)i) / dNp;      // fl(fl(u + i) / Np): resampling.py:103,139
}

// number of output slots whose position is < c  (= first i with pos_i >= c), searched upward
// from `lo` around the estimate c*Np - u.  The estimate is off by a slot or two at most: the loops
// stay rolled (an unrolled-by-four body runs four divisions where one is needed).
template <bool STRATIFIED>
__device__ int count_below(double c, int lo, int Np, double dNp, double u_sys, const double *__restrict__ u_str)
{
    const double est = c * dNp - (STRATIFIED ? 0.5 : u_sys);
    int k = est <= (double)lo ? lo : (est >= (double)Np ? Np : (int)est);
    _Pragma("nounroll") while (k > lo && !(position<STRATIFIED>(k - 1, dNp, u_sys, u_str) < c)) --k;
    _Pragma("nounroll") while (k < Np && position<STRATIFIED>(k, dNp, u_sys, u_str) < c) ++k;
    return k;
}

// systematic / stratified: one workgroup per filter
template <bool STRATIFIED>
__global__ void __launch_bounds__(RS_THREADS)
resample_kernel(long Np, const double *__restrict__ w, const double *__restrict__ u, int32_t *__restrict__ idx,
                int32_t *__restrict__ status)
{
    __shared__ ScanShared sh;
    const long f = blockIdx.x;
    const double *wf = w + f * Np;
    int32_t *of = idx + f * Np;
    const double u_sys = STRATIFIED ? 0.0 : u[f];
    const double *u_str = STRATIFIED ? u + f * Np : nullptr;
    const double dNp = (double)Np;
    const int tid = threadIdx.x;

    double carry = 0.0;
    bool started = false;
    int prelude = RS_PRELUDE;
    int out_lo = 0;      // output slots [0, out_lo) are done
    // the next tile's weights are fetched while this tile is scanned and searched
    double v[RS_ITEMS];
    fetch_tile(v, wf, (int)(Np < RS_TILE ? Np : RS_TILE));
    for (long base = 0; base < Np; base += RS_TILE) {
        const int len = (int)((Np - base) < RS_TILE ? (Np - base) : RS_TILE);
        stage_tile(sh, v);
        __syncthreads();
        const long nbase = base + RS_TILE;
        if (nbase < Np) fetch_tile(v, wf + nbase, (int)((Np - nbase) < RS_TILE ? (Np - nbase) : RS_TILE));
        const double c_in = carry;
        carry = tile_cumsum_exact(sh, len, carry, started, prelude);
        // slots covered by this tile: pos_i < cs_last  (cs is non-decreasing for weights >= 0)
        const int out_hi = count_below<STRATIFIED>(carry, out_lo, (int)Np, dNp, u_sys, u_str);
        const double inv_span = (double)len / (carry - c_in);
        for (int i = out_lo + tid; i < out_hi; i += RS_THREADS) {
            const double p = position<STRATIFIED>(i, dNp, u_sys, u_str);
            // idx = #{ j : cs_j <= p }  (upper bound; the two-pointer merge of resampling.py:143-149)
            of[i] = (int32_t)(base + tile_upper_bound(sh.w(), len, p, c_in, inv_span));
        }
        out_lo = out_hi;
        __syncthreads();
    }
    // positions >= cumsum[-1]: the reference raises IndexError (resampling.py:109,145)
    for (int i = out_lo + tid; i < (int)Np; i += RS_THREADS) of[i] = (int32_t)(Np - 1);
    if (tid == 0 && status) status[f] = out_lo < (int)Np ? ST_OVERRUN : 0;
}

// exact cumulative sums to HBM (multinomial needs random access to them)
__global__ void __launch_bounds__(RS_THREADS)
cumsum_kernel(long Np, const double *__restrict__ w, double *__restrict__ cs, int force_last_one)
{
    __shared__ ScanShared sh;
    const long f = blockIdx.x;
    const double *wf = w + f * Np;
    double *cf = cs + f * Np;
    const int tid = threadIdx.x;
    double carry = 0.0;
    bool started = false;
    int prelude = RS_PRELUDE;
    for (long base = 0; base < Np; base += RS_TILE) {
        const int len = (int)((Np - base) < RS_TILE ? (Np - base) : RS_TILE);
        load_tile(sh, wf + base, len);
        __syncthreads();
        carry = tile_cumsum_exact(sh, len, carry, started, prelude);
        for (int j = tid; j < len; j += RS_THREADS) cf[base + j] = sh.w()[j];
        __syncthreads();
    }
    // cumulative_sum[-1] = 1.  (resampling.py:74,175)
    if (force_last_one && tid == 0 && Np > 0) cf[Np - 1] = 1.0;
}

// idx = searchsorted(cs, u)  (side 'left': #{ j : cs_j < u })
__global__ void __launch_bounds__(RS_THREADS)
searchsorted_left_kernel(long Np, long Nu, const double *__restrict__ cs, const double *__restrict__ u,
                         int64_t *__restrict__ idx)
{
    const long f = blockIdx.y;
    const long i = (long)blockIdx.x * RS_THREADS + threadIdx.x;
    if (i >= Nu) return;
    const double *cf = cs + f * Np;
    const double v = u[f * Nu + i];
    long lo = 0, hi = Np;
    while (lo < hi) {
        const long mid = (lo + hi) >> 1;
        if (cf[mid] < v) lo = mid + 1;
        else hi = mid;
    }
    idx[f * Nu + i] = lo;
}

// =================================================================================================
// Chunk-parallel path for long weight vectors (Np >= RS_PAR_MIN): the sequential dependency of the
// exact cumulative sum is reduced to one O(1) step per 2048-element chunk.
//   P1 chunk_sum_kernel     : plain fp64 sum of every chunk (any order) + a "bad weight" flag
//   P2 chunk_plan_kernel    : per filter, approximate running sum at every chunk start -> the binade
//                             the exact running sum MUST be in for the whole chunk (error-bounded), or
//                             "dirty" when the chunk may cross a binade / sits at the vector start
//   P3 chunk_compose_kernel : clean chunks: the chunk's composite rounding map (Mono) in that binade
//   P4 chain_kernel         : per filter, walk the chunks: clean -> c_out = map(c_in) after VERIFYING
//                             the binade and that no crossing happened; dirty or unverified -> the exact
//                             tile algorithm.  Stores the exact carry-in (+ scan state) of every chunk.
//   P5 resample_chunk_kernel: every chunk independently: exact tile cumsum from its carry-in, output
//                             range from the implicit positions, binary search, coalesced int32 stores.
// Correctness never rests on the approximation of P1/P2 -- it only decides how much of P4 runs as O(1)
// steps; every shortcut is verified against the exact carry.
constexpr long RS_PAR_MIN = 16L * RS_TILE;

struct ChunkPlan {          // one per (filter, chunk), in the caller's workspace
    double approx_sum;      // P1
    double cin;             // P4: exact running sum entering the chunk
    Mono F;                 // P3: composite map of the chunk (clean chunks)
    int eu;                 // P2: ulp exponent of the binade, RS_DIRTY when not clean
    int bad;                // P1: chunk holds a negative / NaN / Inf weight
    int started;            // P4: scan state entering the chunk
    int prelude;            // P4
    int todo;               // P5: the lean output kernel could not finish this chunk -> the general one does
};
constexpr int RS_DIRTY = -100000;

__global__ void __launch_bounds__(RS_THREADS)
chunk_sum_kernel(long Np, long nch, const double *__restrict__ w, ChunkPlan *__restrict__ plan)
{
    __shared__ double red[RS_THREADS / 64];
    __shared__ int badf;
    const long f = blockIdx.y, k = blockIdx.x;
    const long base = k * RS_TILE;
    const int len = (int)((Np - base) < RS_TILE ? (Np - base) : RS_TILE);
    const int tid = threadIdx.x;
    if (tid == 0) badf = 0;
    __syncthreads();
    double acc = 0.0;
    int bad = 0;
    for (int j = tid; j < len; j += RS_THREADS) {
        const double v = w[f * Np + base + j];
        bad |= !(v >= 0.0 && v < 0x1p1000);
        acc += v;
    }
    FK_UNROLL for (int d = 32; d > 0; d >>= 1) acc += __shfl_down(acc, d, 64);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    if (bad) atomicOr(&badf, 1);
    __syncthreads();
    if (tid == 0) {
        ChunkPlan &p = plan[f * nch + k];
        p.approx_sum = (red[0] + red[1]) + (red[2] + red[3]);
        p.bad = badf;
    }
}

// one workgroup per filter: every thread owns a contiguous slice of the chunks; slice totals are
// scanned across the workgroup, then each thread walks its slice (the prefix is only approximate
// anyway, so its association order is free)
__global__ void __launch_bounds__(RS_THREADS)
chunk_plan_kernel(long Np, long nch, ChunkPlan *__restrict__ plan)
{
    __shared__ double tot[RS_THREADS];
    __shared__ int badpos[RS_THREADS];
    const long f = blockIdx.x;
    const int tid = threadIdx.x;
    const long per = (nch + RS_THREADS - 1) / RS_THREADS;
    const long k0 = tid * per, k1 = (k0 + per < nch) ? k0 + per : nch;
    double acc = 0.0;
    long firstbad = nch;
    for (long k = k0; k < k1; ++k) {
        const ChunkPlan &p = plan[f * nch + k];
        const double S = p.approx_sum;
        if ((p.bad != 0 || !(S >= 0.0) || !(S < 0x1p1000)) && firstbad == nch) firstbad = k;
        acc += S;
    }
    tot[tid] = acc;
    badpos[tid] = (int)(firstbad < nch ? firstbad : nch);
    __syncthreads();
    // serial exclusive prefix over 256 slice totals by every thread up to its own slot (cheap, LDS)
    double A = 0.0;
    long poison_from = nch;
    for (int t = 0; t < RS_THREADS; ++t) {
        if (t < tid) A += tot[t];
        if (badpos[t] < poison_from) poison_from = badpos[t];
    }
    // |approx prefix - exact sequential prefix| <= delta * prefix for non-negative weights
    const double delta = 8.0 * (double)(Np + 4096) * 0x1p-53;
    for (long k = k0; k < k1; ++k) {
        ChunkPlan &p = plan[f * nch + k];
        const double S = p.approx_sum;
        int eu = RS_DIRTY;
        if (k < poison_from && k > 0 && A + S < 0x1p1000) {
            const double lo = A * (1.0 - delta), hi = (A + S) * (1.0 + delta);
            if (lo > 0x1p-900 && ulp_exp(lo) == ulp_exp(hi)) eu = ulp_exp(lo);
        }
        p.eu = eu;
        A += S;
    }
}

__global__ void __launch_bounds__(RS_THREADS)
chunk_compose_kernel(long Np, long nch, const double *__restrict__ w, ChunkPlan *__restrict__ plan)
{
    __shared__ Mono wave_tot[RS_THREADS / 64];
    const long f = blockIdx.y, k = blockIdx.x;
    ChunkPlan &p = plan[f * nch + k];
    const int eu = p.eu;
    if (eu == RS_DIRTY) return;            // uniform
    const long base = k * RS_TILE;
    const int len = (int)((Np - base) < RS_TILE ? (Np - base) : RS_TILE);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double u = scale2(1.0, eu);
    double wv[RS_ITEMS];
    FK_UNROLL for (int q = 0; q < RS_ITEMS; ++q) {
        const int j = tid * RS_ITEMS + q;
        wv[q] = j < len ? w[f * Np + base + j] : 0.0;
    }
    // fast path: no half-ulp tie in the chunk -> its composite map is "add S", S a plain fp64 sum of
    // integer increments (fast_inc); an S that reaches 2^53 is inexact but still >= 2^53, which the
    // chain kernel treats as a binade crossing and redoes exactly
    {
        __shared__ double wsum[RS_THREADS / 64];
        double S = 0.0;
        bool tie = false;
        FK_UNROLL for (int q = 0; q < RS_ITEMS; ++q) {
            bool tk = false;
            S += (tid * RS_ITEMS + q < len) ? fast_inc(wv[q], eu, tk) : 0.0;
            tie = tie || tk;
        }
        const int any_tie = __syncthreads_or(tie ? 1 : 0);
        if (!any_tie) {
            FK_UNROLL for (int d = 32; d > 0; d >>= 1) S += __shfl_down(S, d, 64);
            if (lane == 0) wsum[wave] = S;
            __syncthreads();
            if (tid == 0) {
                double t = wsum[0];
                for (int k2 = 1; k2 < RS_THREADS / 64; ++k2) t += wsum[k2];
                const long long Sl = t >= 0x1p60 ? MONO_SAT : (long long)t;
                p.F = Mono{Sl, Sl};
            }
            return;
        }
    }
    Mono run = mono_identity();
    FK_UNROLL for (int q = 0; q < RS_ITEMS; ++q) {
        const int j = tid * RS_ITEMS + q;
        if (j < len) run = mono_compose(run, mono_elem(wv[q], u, eu));
    }
    FK_UNROLL for (int d = 1; d < 64; d <<= 1) {
        const Mono up = shfl_up_mono(run, d);
        if (lane >= d) run = mono_compose(up, run);
    }
    if (lane == 63) wave_tot[wave] = run;
    __syncthreads();
    if (tid == 0) {
        Mono t = wave_tot[0];
        for (int wv = 1; wv < RS_THREADS / 64; ++wv) t = mono_compose(t, wave_tot[wv]);
        p.F = t;
    }
}

constexpr int RS_CHAIN_BATCH = 1024;   // chunks staged in LDS per batch of the chain walk

__global__ void __launch_bounds__(RS_THREADS)
chain_kernel(long Np, long nch, const double *__restrict__ w, ChunkPlan *__restrict__ plan)
{
    __shared__ ScanShared sh;
    // the plan of a batch of chunks, staged in LDS so the sequential walk never waits on HBM
    __shared__ Mono b_F[RS_CHAIN_BATCH];
    __shared__ double b_cin[RS_CHAIN_BATCH];
    __shared__ int b_eu[RS_CHAIN_BATCH], b_started[RS_CHAIN_BATCH], b_prelude[RS_CHAIN_BATCH];
    const long f = blockIdx.x;
    const int tid = threadIdx.x;
    double carry = 0.0;
    bool started = false;
    int prelude = RS_PRELUDE;
    for (long k0 = 0; k0 < nch; k0 += RS_CHAIN_BATCH) {
        const int nb = (int)((nch - k0) < RS_CHAIN_BATCH ? (nch - k0) : RS_CHAIN_BATCH);
        __syncthreads();
        for (int q = tid; q < nb; q += RS_THREADS) {
            const ChunkPlan &p = plan[f * nch + k0 + q];
            b_eu[q] = p.eu;
            b_F[q] = p.F;
        }
        __syncthreads();
        int q = 0;
        while (q < nb) {                               // uniform
            const int eu = b_eu[q];
            if (eu != RS_DIRTY && started && prelude == 0 && carry > 0.0 && ulp_exp(carry) == eu) {
                // A run of chunks planned for this binade: one integer step per chunk,
                // C -> C + (C odd ? ao : ae), exact while C stays < 2^53 -- evaluated for the whole
                // rest of the batch at once by a scan over the chunks' composite maps.  Chunks of
                // another binade / dirty chunks are poisoned so that they end the run.
                const long long C0 = (long long)scale2(carry, -eu);
                constexpr int CI = RS_CHAIN_BATCH / RS_THREADS;
                Mono loc[CI];
                Mono run = mono_identity();
                FK_UNROLL for (int k = 0; k < CI; ++k) {
                    const int j = tid * CI + k;
                    Mono e = mono_identity();
                    if (j >= q && j < nb) e = (b_eu[j] == eu) ? b_F[j] : Mono{MONO_BIG, MONO_BIG};
                    run = mono_compose(run, e);
                    loc[k] = run;
                }
                if (tid == 0) sh.first_cross = RS_CHAIN_BATCH;
                const Mono excl = block_mono_excl<CI>(loc, sh.wave_tot);
                long long Cin[CI];
                int my_bad = RS_CHAIN_BATCH;
                FK_UNROLL for (int k = 0; k < CI; ++k) {
                    const int j = tid * CI + k;
                    Cin[k] = mono_apply(C0, k == 0 ? excl : mono_compose(excl, loc[k - 1]));
                    const long long Cout = mono_apply(C0, mono_compose(excl, loc[k]));
                    if (j >= q && j < nb && Cout >= MONO_LIMIT && my_bad == RS_CHAIN_BATCH) my_bad = j;
                }
                if (my_bad < RS_CHAIN_BATCH) atomicMin(&sh.first_cross, my_bad);
                __syncthreads();
                const int stop = sh.first_cross < nb ? sh.first_cross : nb;   // first chunk NOT covered by the run
                FK_UNROLL for (int k = 0; k < CI; ++k) {
                    const int j = tid * CI + k;
                    if (j >= q && j < stop) {
                        b_cin[j] = scale2((double)Cin[k], eu);
                        b_started[j] = 1;
                        b_prelude[j] = 0;
                    }
                    if (j == stop - 1 && stop > q) sh.carry = scale2((double)mono_apply(C0, mono_compose(excl, loc[k])), eu);
                }
                __syncthreads();
                if (stop > q) {
                    carry = sh.carry;
                    q = stop;
                    continue;
                }
                // the very first chunk of the run leaves the binade: general path below
            }
            if (tid == 0) {
                b_cin[q] = carry;
                b_started[q] = started ? 1 : 0;
                b_prelude[q] = prelude;
            }
            const long base = (k0 + q) * RS_TILE;
            const int len = (int)((Np - base) < RS_TILE ? (Np - base) : RS_TILE);
            __syncthreads();
            load_tile(sh, w + f * Np + base, len);
            __syncthreads();
            carry = tile_cumsum_exact(sh, len, carry, started, prelude);
            ++q;
        }
        __syncthreads();
        for (int q = tid; q < nb; q += RS_THREADS) {
            ChunkPlan &p = plan[f * nch + k0 + q];
            p.cin = b_cin[q];
            p.started = b_started[q];
            p.prelude = b_prelude[q];
        }
    }
}

// Build-time instrumentation (tools/rs_phase.py builds a separate library with -DFK_RS_PHASE_CLOCKS; the
// shipped library has none of it): wave 0 of every workgroup of resample_chunk_kernel adds the s_memtime
// ticks of each phase to fk_rs_phase[], fk_debug_rs_phases() reads and clears them.
#ifdef FK_RS_PHASE_CLOCKS
constexpr int RS_PHASE_BUCKETS = 4096;
__device__ unsigned long long fk_rs_phase[8][RS_PHASE_BUCKETS];
// ticks are kept in registers and reach memory once, after the last phase (an atomic per phase would sit in
// the same in-order vmcnt queue as the tile loads and be measured as "waiting for the tile"), spread over
// RS_PHASE_BUCKETS addresses per slot (half a million workgroups adding to ONE address serialize in L2 and slow
// the very kernel being measured)
#define RS_CLOCK_START()                                \
    long long t_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};      \
    long long t_prev = __builtin_readcyclecounter()
#define RS_CLOCK(slot)                                              \
    do {                                                            \
        const long long t_now = __builtin_readcyclecounter();       \
        t_acc[slot] += t_now - t_prev;                              \
        t_prev = t_now;                                             \
    } while (0)
#define RS_CLOCK_FLUSH()                                                                           \
    do {                                                                                           \
        if (threadIdx.x == 0)                                                                      \
            for (int q_ = 0; q_ < 8; ++q_)                                                         \
                atomicAdd(&fk_rs_phase[q_][(blockIdx.x + 977u * blockIdx.y) % RS_PHASE_BUCKETS], (unsigned long long)t_acc[q_]); \
    } while (0)
#else
#define RS_CLOCK(slot) do { } while (0)
#define RS_CLOCK_START() do { } while (0)
#define RS_CLOCK_FLUSH() do { } while (0)
#endif

// P5, lean route: the chunk as ONE tie-free binade segment (the fast path of tile_cumsum_exact with pos = 0 and
// no crossing -- the same operations in the same order, so the same bits), which is what nearly every chunk
// of a long vector is.  Without the int64 Mono scan, the serial fall-backs and the prelude this kernel needs
// 64 VGPRs where the general one needs 94: eight workgroups per CU instead of five.  Anything
// it cannot prove (scan not started, prelude pending, running sum not a positive finite number, a half-ulp
// tie, a sum that leaves the binade, a negative / non-finite weight) is left untouched and flagged in
// plan.todo for resample_chunk_kernel.
struct LeanShared {
    double tile[2 + RS_TILE + TILE_GUARD];      // guarded like ScanShared::tile
    double wave_sum[RS_THREADS / 64];
    __device__ __forceinline__ double *w() { return tile + 2; }
};

template <bool STRATIFIED>
__global__ void __launch_bounds__(RS_THREADS)
resample_chunk_lean_kernel(long Np, long nch, const double *__restrict__ w, const double *__restrict__ u,
                           ChunkPlan *__restrict__ plan, int32_t *__restrict__ idx, int32_t *__restrict__ status)
{
    __shared__ LeanShared sh;
    const long f = blockIdx.y, k = blockIdx.x;
    ChunkPlan &p = plan[f * nch + k];
    const double *wf = w + f * Np;
    int32_t *of = idx + f * Np;
    const double u_sys = STRATIFIED ? 0.0 : u[f];
    const double *u_str = STRATIFIED ? u + f * Np : nullptr;
    const double dNp = (double)Np;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long base = k * RS_TILE;
    const int len = (int)((Np - base) < RS_TILE ? (Np - base) : RS_TILE);
    RS_CLOCK_START();
    // the tile's address does not depend on the plan entry: its loads go out before that entry is waited for
    // (a chunk that turns out not to be ours costs one unused fetch)
    double v[RS_ITEMS];
    fetch_tile(v, wf + base, len);
    RS_CLOCK(5);                    // loads issued
    const double c_in = p.cin;
    const bool can = p.started != 0 && p.prelude == 0 && c_in > 0.0 && c_in <= 1.79769313486231570815e+308 && k > 0;
    if (!can) {                                                       // uniform
        if (tid == 0) p.todo = 1;
        return;
    }
    // everything that does not need the tile runs while it is in flight: the binade of the running sum and
    // the first output slot (slots below the previous chunk's last cumulative sum belong to earlier chunks)
    const double ulp = ulp_of(c_in);
    const int eu = ulp_exp(c_in);
    const double C0d = scale2(c_in, -eu);
    const int out_lo = count_below<STRATIFIED>(c_in, 0, (int)Np, dNp, u_sys, u_str);
    RS_CLOCK(1);
    FK_UNROLL for (int q = 0; q < RS_ITEMS; ++q) sh.w()[tid + q * RS_THREADS] = v[q];
    if (tid < TILE_GUARD) sh.w()[RS_TILE + tid] = __builtin_inf();
    if (tid == 0) sh.w()[-1] = -__builtin_inf();
    RS_CLOCK(6);                    // loads landed, LDS written
    __syncthreads();
    RS_CLOCK(0);                    // the other waves arrived
    // tie-free exact scan (fk_exact_scan.hpp, fast_inc): thread t owns elements t*RS_ITEMS ..
    double incl[RS_ITEMS];
    double runs = 0.0;
    bool odd = false;               // a tie, or (below) a sum that reaches 2^53
    FK_UNROLL for (int q = 0; q < RS_ITEMS; ++q) {
        const int j = tid * RS_ITEMS + q;
        bool tk = false;
        const double e = j < len ? fast_inc(sh.w()[j], eu, tk) : 0.0;
        odd = odd || tk;
        runs += e;
        incl[q] = runs;
    }
    double inc = runs;
    FK_UNROLL for (int d = 1; d < 64; d <<= 1) {
        const double up = __shfl_up(inc, d, 64);
        if (lane >= d) inc += up;
    }
    if (lane == 63) sh.wave_sum[wave] = inc;
    __syncthreads();
    double excl = __shfl_up(inc, 1, 64);
    if (lane == 0) excl = 0.0;
    FK_UNROLL for (int wv = 0; wv < RS_THREADS / 64; ++wv)
        if (wv < wave) excl += sh.wave_sum[wv];
    FK_UNROLL for (int q = 0; q < RS_ITEMS; ++q) {
        const int j = tid * RS_ITEMS + q;
        incl[q] = C0d + (excl + incl[q]);
        odd = odd || (j < len && !(incl[q] < 0x1p53));
    }
    if (__syncthreads_or(odd ? 1 : 0)) {                              // uniform
        if (tid == 0) p.todo = 1;
        return;
    }
    FK_UNROLL for (int q = 0; q < RS_ITEMS; ++q) {
        const int j = tid * RS_ITEMS + q;
        if (j < len) sh.w()[j] = incl[q] * ulp;                       // exact
    }
    __syncthreads();
    const double carry = sh.w()[len - 1];
    RS_CLOCK(2);
    const int out_hi = count_below<STRATIFIED>(carry, out_lo, (int)Np, dNp, u_sys, u_str);
    RS_CLOCK(3);
    const double inv_span = (double)len / (carry - c_in);
    for (int i = out_lo + tid; i < out_hi; i += RS_THREADS) {
        const double ps = position<STRATIFIED>(i, dNp, u_sys, u_str);
        of[i] = (int32_t)(base + tile_upper_bound(sh.w(), len, ps, c_in, inv_span));
    }
    RS_CLOCK(4);
    if (k == nch - 1) {
        for (int i = out_hi + tid; i < (int)Np; i += RS_THREADS) of[i] = (int32_t)(Np - 1);
        if (tid == 0 && status) status[f] = out_hi < (int)Np ? ST_OVERRUN : 0;
    }
    if (tid == 0) p.todo = 0;
    RS_CLOCK_FLUSH();
}

// P5, general route: the chunks the lean kernel flagged (first chunk of a filter, binade crossings, ties,
// invalid weights), every case of tile_cumsum_exact
template <bool STRATIFIED>
__global__ void __launch_bounds__(RS_THREADS)
resample_chunk_kernel(long Np, long nch, const double *__restrict__ w, const double *__restrict__ u,
                      const ChunkPlan *__restrict__ plan, int32_t *__restrict__ idx, int32_t *__restrict__ status)
{
    __shared__ ScanShared sh;
    const long f = blockIdx.y, k = blockIdx.x;
    const ChunkPlan &p = plan[f * nch + k];
    if (p.todo == 0) return;                                          // uniform
    const double *wf = w + f * Np;
    int32_t *of = idx + f * Np;
    const double u_sys = STRATIFIED ? 0.0 : u[f];
    const double *u_str = STRATIFIED ? u + f * Np : nullptr;
    const double dNp = (double)Np;
    const int tid = threadIdx.x;
    const long base = k * RS_TILE;
    const int len = (int)((Np - base) < RS_TILE ? (Np - base) : RS_TILE);
    load_tile(sh, wf + base, len);
    __syncthreads();
    double carry = p.cin;
    bool started = p.started != 0;
    int prelude = p.prelude;
    // slots below the previous chunk's last cumulative sum belong to earlier chunks
    const int out_lo = (k == 0) ? 0 : count_below<STRATIFIED>(carry, 0, (int)Np, dNp, u_sys, u_str);
    const double c_in = carry;
    carry = tile_cumsum_exact(sh, len, carry, started, prelude);
    const int out_hi = count_below<STRATIFIED>(carry, out_lo, (int)Np, dNp, u_sys, u_str);
    const double inv_span = (double)len / (carry - c_in);
    for (int i = out_lo + tid; i < out_hi; i += RS_THREADS) {
        const double ps = position<STRATIFIED>(i, dNp, u_sys, u_str);
        of[i] = (int32_t)(base + tile_upper_bound(sh.w(), len, ps, c_in, inv_span));
    }
    if (k == nch - 1) {
        for (int i = out_hi + tid; i < (int)Np; i += RS_THREADS) of[i] = (int32_t)(Np - 1);
        if (tid == 0 && status) status[f] = out_hi < (int)Np ? ST_OVERRUN : 0;
    }
}


// Posterior mean of the resampled set: mean[f][k] = (1/Np) sum_i particles[f][idx[f][i]][k].  Not a filterpy
// function -- it is the "resample from index" + mean every caller of the resamplers writes
// (particles[:] = particles[indexes]; docs/monte_carlo/resampling.rst), fused so that the resampled copy is
// never materialised; BASELINE configs[4] all-gathers these means.  Partial sums meet in fp64 atomic adds:
// the result is the mean up to summation-order rounding (not bit-reproducible run to run).
constexpr int GM_CHUNK = 16384;
template <int D>
__global__ void __launch_bounds__(RS_THREADS)
gather_mean_kernel(long Np, const double *__restrict__ particles, const int32_t *__restrict__ idx, double *__restrict__ mean,
                   int d)
{
    __shared__ double red[RS_THREADS / 64][D];
    const long f = blockIdx.y;
    const long i0 = (long)blockIdx.x * GM_CHUNK;
    const long i1 = (i0 + GM_CHUNK < Np) ? i0 + GM_CHUNK : Np;
    const double *pf = particles + f * Np * d;
    const int32_t *xf = idx + f * Np;
    double acc[D];
    FK_UNROLL for (int k = 0; k < D; ++k) acc[k] = 0.0;
    for (long i = i0 + threadIdx.x; i < i1; i += RS_THREADS) {
        const double *src = pf + (long)xf[i] * d;
        FK_UNROLL for (int k = 0; k < D; ++k)
            if (k < d) acc[k] += src[k];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    FK_UNROLL for (int k = 0; k < D; ++k) {
        FK_UNROLL for (int s = 32; s > 0; s >>= 1) acc[k] += __shfl_down(acc[k], s, 64);
        if (lane == 0) red[wave][k] = acc[k];
    }
    __syncthreads();
    if (threadIdx.x < (unsigned)d) {
        double t = 0.0;
        for (int w = 0; w < RS_THREADS / 64; ++w) t += red[w][threadIdx.x];
        atomicAdd(&mean[f * d + threadIdx.x], t / (double)Np);
    }
}

static int fail(int code, const char *msg)
{
    set_last_error(msg);
    return code;
}

}  // namespace fk

using namespace fk;

extern "C" {

size_t fk_resample_workspace_bytes(int64_t Fn, int64_t Np)
{
    // systematic / stratified: the hand-off records of the one-pass path (resample_onepass.hip); the multi-pass
    // path kept for comparison (FK_RESAMPLE_PATH=chunk) needs one ChunkPlan per 2048 weights
    if (Fn <= 0 || Np <= 0) return 0;
    const size_t nch = (size_t)((Np + RS_TILE - 1) / RS_TILE);
    const size_t a = (size_t)Fn * nch * sizeof(ChunkPlan), b = onepass_workspace_bytes(Fn, Np);
    return a > b ? a : b;
}

size_t fk_multinomial_workspace_bytes(int64_t Fn, int64_t Np)
{
    // the cumulative sums: Fn * Np doubles
    return (Fn > 0 && Np > 0) ? (size_t)Fn * (size_t)Np * sizeof(double) : 0;
}

static int resample_common(bool stratified, int64_t Fn, int64_t Np, const double *w, const double *u,
                           int32_t *idx, int32_t *status, void *ws, size_t ws_bytes, void *stream)
{
    if (Fn < 0 || Np < 0) return fail(FK_ERR_BAD_ARG, "resample: negative size");
    if (Np >= 2147483647LL) return fail(FK_ERR_UNSUPPORTED, "resample: Np must fit int32 (the reference returns int32 indices)");
    if (Fn == 0 || Np == 0) return FK_OK;
    if (!w || !u || !idx) return fail(FK_ERR_BAD_ARG, "resample: w, u, idx must not be NULL");
    hipStream_t s = (hipStream_t)stream;
    const long nch = (long)((Np + RS_TILE - 1) / RS_TILE);
    const size_t plan_bytes = (size_t)Fn * (size_t)nch * sizeof(ChunkPlan);
    // Long vectors (Np >= RS_PAR_MIN) take the one-pass path (resample_onepass.hip); short ones are a handful of
    // dependent tiles whatever is done and stay with one workgroup per filter (resample_kernel above).
    // FK_RESAMPLE_PATH overrides: onepass (any length) | chunk (the multi-pass path of round 1) | serial
    const char *path = getenv("FK_RESAMPLE_PATH");
    const bool want_onepass = path ? !strcmp(path, "onepass") : Np >= RS_PAR_MIN;
    const bool want_chunk = path && !strcmp(path, "chunk");
    if (want_onepass && ws && ws_bytes >= onepass_workspace_bytes(Fn, Np) && !getenv("FK_RESAMPLE_SERIAL")) {
        const int rc = onepass_launch(stratified, Fn, Np, w, u, idx, status, ws, ws_bytes, s);
        if (rc == FK_ERR_UNSUPPORTED) return fail(rc, "resample: too many chunks for one launch");
        if (rc != FK_OK && rc != FK_ERR_LAUNCH) return fail(rc, "resample: one-pass launch failed");
        return rc;
    }
    if (want_chunk && Np >= RS_PAR_MIN && ws && ws_bytes >= plan_bytes && nch <= 65535 * 32L && Fn <= 65535 && !getenv("FK_RESAMPLE_SERIAL")) {
        // chunk-parallel path: the caller's workspace holds the plan
        ChunkPlan *plan = (ChunkPlan *)ws;
        const dim3 gch((unsigned)nch, (unsigned)Fn), block(RS_THREADS);
        hipLaunchKernelGGL(chunk_sum_kernel, gch, block, 0, s, (long)Np, nch, w, plan);
        hipLaunchKernelGGL(chunk_plan_kernel, dim3((unsigned)Fn), block, 0, s, (long)Np, nch, plan);
        hipLaunchKernelGGL(chunk_compose_kernel, gch, block, 0, s, (long)Np, nch, w, plan);
        hipLaunchKernelGGL(chain_kernel, dim3((unsigned)Fn), block, 0, s, (long)Np, nch, w, plan);
        if (stratified) {
            hipLaunchKernelGGL((resample_chunk_lean_kernel<true>), gch, block, 0, s, (long)Np, nch, w, u, plan, idx, status);
            hipLaunchKernelGGL((resample_chunk_kernel<true>), gch, block, 0, s, (long)Np, nch, w, u, plan, idx, status);
        } else {
            hipLaunchKernelGGL((resample_chunk_lean_kernel<false>), gch, block, 0, s, (long)Np, nch, w, u, plan, idx, status);
            hipLaunchKernelGGL((resample_chunk_kernel<false>), gch, block, 0, s, (long)Np, nch, w, u, plan, idx, status);
        }
        return check_launch("resample_chunk_kernel");
    }
    const dim3 grid((unsigned)Fn), block(RS_THREADS);
    if (stratified)
        hipLaunchKernelGGL((resample_kernel<true>), grid, block, 0, s, (long)Np, w, u, idx, status);
    else
        hipLaunchKernelGGL((resample_kernel<false>), grid, block, 0, s, (long)Np, w, u, idx, status);
    if (int rc = check_launch("resample_kernel")) return rc;
    return literal_fixup_launch(stratified, Fn, Np, w, u, idx, status, s);
}

#ifdef FK_RS_PHASE_CLOCKS
extern "C" int fk_debug_rs_phases(unsigned long long *out)     // only in the instrumented build (tools/rs_phase.py)
{
    static unsigned long long host[8][RS_PHASE_BUCKETS];
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(fk_rs_phase), sizeof(host)) != hipSuccess) return FK_ERR_LAUNCH;
    for (int q = 0; q < 8; ++q) {
        out[q] = 0;
        for (int b = 0; b < RS_PHASE_BUCKETS; ++b) out[q] += host[q][b];
    }
    memset(host, 0, sizeof(host));
    return hipMemcpyToSymbol(HIP_SYMBOL(fk_rs_phase), host, sizeof(host)) == hipSuccess ? FK_OK : FK_ERR_LAUNCH;
}
#endif

int fk_resample_systematic_f64(int64_t Fn, int64_t Np, const double *w, const double *u, int32_t *idx,
                               int32_t *status, void *ws, size_t ws_bytes, void *stream)
{
    return resample_common(false, Fn, Np, w, u, idx, status, ws, ws_bytes, stream);
}

int fk_resample_stratified_f64(int64_t Fn, int64_t Np, const double *w, const double *u, int32_t *idx,
                               int32_t *status, void *ws, size_t ws_bytes, void *stream)
{
    return resample_common(true, Fn, Np, w, u, idx, status, ws, ws_bytes, stream);
}

int fk_cumsum_exact_f64(int64_t Fn, int64_t Np, const double *w, double *cs, int32_t force_last_one, void *stream)
{
    if (Fn < 0 || Np < 0) return fail(FK_ERR_BAD_ARG, "cumsum: negative size");
    if (Fn == 0 || Np == 0) return FK_OK;
    if (!w || !cs) return fail(FK_ERR_BAD_ARG, "cumsum: w and cs must not be NULL");
    hipLaunchKernelGGL(cumsum_kernel, dim3((unsigned)Fn), dim3(RS_THREADS), 0, (hipStream_t)stream, (long)Np, w, cs,
                       (int)force_last_one);
    return check_launch("cumsum_kernel");
}

int fk_resample_multinomial_f64(int64_t Fn, int64_t Np, int64_t Nu, const double *w, const double *u,
                                int64_t *idx, void *ws, size_t ws_bytes, void *stream)
{
    if (Fn < 0 || Np < 0 || Nu < 0) return fail(FK_ERR_BAD_ARG, "multinomial: negative size");
    if (Fn == 0 || Np == 0 || Nu == 0) return FK_OK;
    if (!w || !u || !idx) return fail(FK_ERR_BAD_ARG, "multinomial: w, u, idx must not be NULL");
    if (!ws || ws_bytes < fk_multinomial_workspace_bytes(Fn, Np)) return fail(FK_ERR_WORKSPACE, "multinomial: workspace too small");
    double *cs = (double *)ws;
    if (int rc = fk_cumsum_exact_f64(Fn, Np, w, cs, 1, stream)) return rc;
    const dim3 grid((unsigned)((Nu + RS_THREADS - 1) / RS_THREADS), (unsigned)Fn), block(RS_THREADS);
    hipLaunchKernelGGL(searchsorted_left_kernel, grid, block, 0, (hipStream_t)stream, (long)Np, (long)Nu, cs, u, idx);
    return check_launch("searchsorted_left_kernel");
}

int fk_resample_gather_mean_f64(int64_t Fn, int64_t Np, int32_t d, const double *particles, const int32_t *idx,
                                double *mean, void *stream)
{
    if (Fn < 0 || Np < 0 || d < 1 || d > 8) return fail(FK_ERR_BAD_ARG, "gather_mean: Fn, Np >= 0 and 1 <= d <= 8");
    if (Fn == 0) return FK_OK;
    if (!mean || (Np > 0 && (!particles || !idx))) return fail(FK_ERR_BAD_ARG, "gather_mean: NULL argument");
    if (Fn > 65535) return fail(FK_ERR_UNSUPPORTED, "gather_mean: at most 65535 filters per call");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(mean, 0, (size_t)Fn * d * sizeof(double), s) != hipSuccess) return fail(FK_ERR_LAUNCH, "gather_mean: memset failed");
    if (Np == 0) return FK_OK;
    const dim3 grid((unsigned)((Np + GM_CHUNK - 1) / GM_CHUNK), (unsigned)Fn), block(RS_THREADS);
    if (d <= 4) hipLaunchKernelGGL((gather_mean_kernel<4>), grid, block, 0, s, (long)Np, particles, idx, mean, (int)d);
    else hipLaunchKernelGGL((gather_mean_kernel<8>), grid, block, 0, s, (long)Np, particles, idx, mean, (int)d);
    return check_launch("gather_mean_kernel");
}

}  // extern "C"
