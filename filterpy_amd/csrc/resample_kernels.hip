// resample_kernels.hip -- particle-filter resampling for gfx950 (MI355X).
//
//   fk_resample_systematic_f64  <- systematic_resample (filterpy/monte_carlo/resampling.py:117-150)
//   fk_resample_stratified_f64  <- stratified_resample (:80-114)
//   fk_resample_multinomial_f64 <- multinomial_resample (:153-176) and the random tail of
//                                  residual_resample (:72-76)
//   fk_cumsum_exact_f64         <- numpy.cumsum of a float64 vector, bit-for-bit
//
// Short vectors (and the cumulative sums the multinomial resampler needs in HBM): one workgroup per filter walks the
// weight vector in tiles of RS_TILE weights; long vectors take the one-pass path of resample_onepass.hip.  Per tile:
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

// ---- residual_resample (filterpy/monte_carlo/resampling.py:27-76) ---------------------------------------------------------
// Part 1, one workgroup per filter:
//   num_copies = floor(N w)  (:61);  indexes[k++] = i, num_copies[i] times, i ascending  (:63-66) -- an exclusive integer scan
//   of the copy counts gives every weight the start of its run;  residual = w - num_copies  (:70 -- not N w - num_copies:
//   restated literally, so every weight that earned a copy has a NEGATIVE residual);  residual /= sum(residual)  (:71: the
//   Python builtin, i.e. 0 + r_0 + r_1 + ... strictly left to right);  cumulative_sum = cumsum(residual); [-1] = 1.  (:72-74).
// The two add chains are walked by ONE thread over tiles staged in LDS: with negative terms the running sum wanders through
// binades in both directions and fk_exact_scan.hpp's monoid (non-negative terms) does not apply -- 2 x Np dependent adds,
// ~70 us for 8000 particles, every filter on its own workgroup.  k > Np (the reference's fill loop then raises IndexError)
// sets ST_OVERRUN and leaves the rest undone.
// Part 2: indexes[k:N] = searchsorted(cumulative_sum, random(N - k))  (:75-76), one thread per draw.
__global__ void __launch_bounds__(RS_THREADS)
residual_fill_kernel(long Np, const double *__restrict__ w, int32_t *__restrict__ idx, long *__restrict__ kout,
                     double *__restrict__ cs, int32_t *__restrict__ status)
{
    __shared__ double tile[RS_TILE];
    __shared__ long wtot[RS_THREADS / 64];
    __shared__ double bc;
    const long f = blockIdx.x;
    const double *wf = w + f * Np;
    int32_t *of = idx + f * Np;
    double *cf = cs + f * Np;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double dN = (double)Np;
    long carry = 0;                                                        // copies before the tile
    for (long base = 0; base < Np; base += RS_TILE) {                      // uniform
        const int len = (int)((Np - base) < RS_TILE ? (Np - base) : RS_TILE);
        long cnt[RS_ITEMS], run = 0;
        FK_UNROLL for (int q = 0; q < RS_ITEMS; ++q) {                     // thread t owns RS_ITEMS consecutive weights
            const int j = tid * RS_ITEMS + q;
            const double wj = j < len ? wf[base + j] : 0.0;
            const double c = floor(dN * wj);                               // (np.floor(N * w)).astype(int): one rounding, then floor
            // negative / NaN: no copies (the reference's range(negative) is empty).  More copies than slots: the reference's
            // fill loop raises IndexError at slot Np -- saturate at Np + 1, so that the total exceeds Np and ST_OVERRUN is
            // reported, whatever the size of N w (un-normalised weights: N w ~ 1e9 used to spin here for minutes, and counts
            // of 2^40 and more were dropped silently: ADVICE r3)
            const long ci = !(c >= 0.0) ? 0 : (c > dN ? Np + 1 : (long)c);
            cnt[q] = ci;
            run += ci;
            if (j < len) cf[base + j] = wj - c;                            // the residual, for the two passes below
        }
        long incl = run;
        FK_UNROLL for (int d = 1; d < 64; d <<= 1) {
            const long up = __shfl_up(incl, d, 64);
            if (lane >= d) incl += up;
        }
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        long off = carry + incl - run, total = 0;
        FK_UNROLL for (int wv = 0; wv < RS_THREADS / 64; ++wv) {
            if (wv < wave) off += wtot[wv];
            total += wtot[wv];
        }
        FK_UNROLL for (int q = 0; q < RS_ITEMS; ++q) {
            const int j = tid * RS_ITEMS + q;
            const long room = off < Np ? Np - off : 0;                     // copies that still land inside the vector
            const long todo = cnt[q] < room ? cnt[q] : room;
            for (long c = 0; c < todo; ++c) of[off + c] = (int32_t)(base + j);
            off += cnt[q];
        }
        carry += total;
        __syncthreads();
    }
    if (tid == 0) {
        kout[f] = carry;
        if (status) status[f] = carry > Np ? ST_OVERRUN : 0;
    }
    if (carry > Np) return;                                                // uniform: IndexError in the reference
    __threadfence_block();
    __syncthreads();
    // sum(residual): the builtin's chain 0 + r_0 + r_1 + ...
    double total = 0.0;
    for (int pass = 0; pass < 2; ++pass) {                                 // 0: the sum; 1: the cumulative sums of r / sum
        double c = 0.0;
        for (long base = 0; base < Np; base += RS_TILE) {
            const int len = (int)((Np - base) < RS_TILE ? (Np - base) : RS_TILE);
            for (int j = tid; j < len; j += RS_THREADS) tile[j] = pass ? cf[base + j] / total : cf[base + j];   // (:71: one division each)
            __syncthreads();
            if (tid == 0) {
                for (int j = 0; j < len; j += 8) {                         // eight independent LDS reads, then the add chain
                    double v[8];
                    FK_UNROLL for (int e = 0; e < 8; ++e) v[e] = j + e < len ? tile[j + e] : 0.0;
                    FK_UNROLL for (int e = 0; e < 8; ++e) {
                        if (j + e < len) {
                            c = c + v[e];
                            v[e] = c;
                        }
                    }
                    if (pass) { FK_UNROLL for (int e = 0; e < 8; ++e) if (j + e < len) tile[j + e] = v[e]; }
                }
                bc = c;
            }
            __syncthreads();
            if (pass)
                for (int j = tid; j < len; j += RS_THREADS) cf[base + j] = tile[j];
            c = bc;
            __syncthreads();
        }
        if (pass == 0) total = c;
    }
    if (tid == 0) cf[Np - 1] = 1.0;                                        // :74
}

__global__ void __launch_bounds__(RS_THREADS)
residual_draw_kernel(long Np, const double *__restrict__ cs, const long *__restrict__ k, const long *__restrict__ uoff,
                     const double *__restrict__ u, int32_t *__restrict__ idx, long f0)
{
    const long f = f0 + blockIdx.y;
    const long kf = k[f];
    const long i = (long)blockIdx.x * RS_THREADS + threadIdx.x;
    if (kf > Np || i >= Np - kf) return;
    const double *cf = cs + f * Np;
    const double v = u[uoff[f] + i];
    long lo = 0, hi = Np;
    while (lo < hi) {                                                      // numpy.searchsorted, side 'left'
        const long mid = (lo + hi) >> 1;
        if (cf[mid] < v) lo = mid + 1;
        else hi = mid;
    }
    idx[f * Np + kf + i] = (int32_t)lo;
}

// vectors of at least RS_PAR_MIN weights take the one-pass path (resample_onepass.hip): many workgroups per filter
constexpr long RS_PAR_MIN = 16L * RS_TILE;

// Posterior mean of the resampled set: mean[f][k] = (1/Np) sum_i particles[f][idx[f][i]][k].  Not a filterpy
// function -- it is the "resample from index" + mean every caller of the resamplers writes
// (particles[:] = particles[indexes]; docs/monte_carlo/resampling.rst), fused so that the resampled copy is
// never materialised; BASELINE configs[4] all-gathers these means.  Partial sums meet in fp64 atomic adds:
// the result is the mean up to summation-order rounding (not bit-reproducible run to run).
constexpr int GM_CHUNK = 16384;
template <int D>
__global__ void __launch_bounds__(RS_THREADS)
gather_mean_kernel(long Np, const double *__restrict__ particles, const int32_t *__restrict__ idx, double *__restrict__ mean,
                   int d, long f0)
{
    __shared__ double red[RS_THREADS / 64][D];
    const long f = f0 + blockIdx.y;
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

// The same for 32-byte records (d = 4, 16-byte aligned: BASELINE configs[4]'s harness), round 4.  The indices a resampler
// returns are non-decreasing, so a workgroup's slice of them points into ONE short contiguous range of particles and the
// gather is a stream -- what the first kernel lacked was memory-level parallelism: one dependent (index -> record) pair in
// flight per lane and four 8-byte loads per record (8.0 ms for 36 GB at 125 x 8e6 = 0.56 of HBM; this one: 5.8 ms = 0.78).  Here every lane has GM_U
// independent index loads in flight, then 2 GM_U independent 16-byte record loads; a wave's 64 lanes take 64 consecutive
// indices per load, i.e. (with duplicates) one or two KiB of consecutive records.
constexpr int GM_CHUNK4 = 16384;
template <int GM_U>
__global__ void __launch_bounds__(RS_THREADS)
gather_mean4_kernel(long Np, const double *__restrict__ particles, const int32_t *__restrict__ idx, double *__restrict__ mean, long f0)
{
    using f64x2 = __attribute__((ext_vector_type(2))) double;
    __shared__ double red[RS_THREADS / 64][4];
    const long f = f0 + blockIdx.y;
    const long i0 = (long)blockIdx.x * GM_CHUNK4;
    const long i1 = (i0 + GM_CHUNK4 < Np) ? i0 + GM_CHUNK4 : Np;
    const f64x2 *pf = reinterpret_cast<const f64x2 *>(particles + f * Np * 4);
    const int32_t *xf = idx + f * Np;
    f64x2 a0 = {0.0, 0.0}, a1 = {0.0, 0.0}, b0 = {0.0, 0.0}, b1 = {0.0, 0.0};      // two accumulator pairs: shorter add chains
    long i = i0 + threadIdx.x;
    for (; i + (long)(GM_U - 1) * RS_THREADS < i1; i += (long)GM_U * RS_THREADS) {
        int j[GM_U];
        FK_UNROLL for (int q = 0; q < GM_U; ++q) j[q] = __builtin_nontemporal_load(xf + i + (long)q * RS_THREADS);
        f64x2 lo[GM_U], hi[GM_U];
        FK_UNROLL for (int q = 0; q < GM_U; ++q) {
            lo[q] = pf[2L * j[q]];
            hi[q] = pf[2L * j[q] + 1];
        }
        FK_UNROLL for (int q = 0; q < GM_U; q += 2) {
            a0 += lo[q];
            a1 += hi[q];
            b0 += lo[q + 1];
            b1 += hi[q + 1];
        }
    }
    for (; i < i1; i += RS_THREADS) {
        const long j = xf[i];
        a0 += pf[2 * j];
        a1 += pf[2 * j + 1];
    }
    double acc[4] = {a0.x + b0.x, a0.y + b0.y, a1.x + b1.x, a1.y + b1.y};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    FK_UNROLL for (int k = 0; k < 4; ++k) {
        FK_UNROLL for (int s = 32; s > 0; s >>= 1) acc[k] += __shfl_down(acc[k], s, 64);
        if (lane == 0) red[wave][k] = acc[k];
    }
    __syncthreads();
    if (threadIdx.x < 4u) {
        double t = 0.0;
        for (int w = 0; w < RS_THREADS / 64; ++w) t += red[w][threadIdx.x];
        atomicAdd(&mean[f * 4 + threadIdx.x], t / (double)Np);
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
    // systematic / stratified: the hand-off records of the one-pass path (resample_onepass.hip)
    return onepass_workspace_bytes(Fn, Np);
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
    // Three kernels by vector length: up to 8192 weights one workgroup takes the WHOLE vector in one round
    // (resample_whole.hip); from RS_PAR_MIN on the one-pass path (resample_onepass.hip: many workgroups per filter);
    // in between one workgroup per filter walks 2048-weight chunks in sequence (resample_local_kernel).
    // FK_RESAMPLE_PATH=whole|local|onepass forces a kernel where it applies, FK_RESAMPLE_SERIAL=1 the round-1
    // tile-by-tile kernel (the tests' independent cross-check).
    const char *path = getenv("FK_RESAMPLE_PATH");
    const bool serial = getenv("FK_RESAMPLE_SERIAL") != nullptr;
    const bool forced_other = path && (!strcmp(path, "onepass") || !strcmp(path, "local"));
    if (!serial && !forced_other && whole_supported(Np)) {
        const int rc = whole_launch(stratified, Fn, Np, w, u, idx, status, s);
        if (rc == FK_ERR_UNSUPPORTED) return fail(rc, "resample: too many filters for one launch");
        return rc;
    }
    const bool want_onepass = (path && !strcmp(path, "onepass")) || (Np >= RS_PAR_MIN && !(path && !strcmp(path, "local")));
    if (want_onepass && ws && ws_bytes >= onepass_workspace_bytes(Fn, Np) && !getenv("FK_RESAMPLE_SERIAL")) {
        const int rc = onepass_launch(stratified, Fn, Np, w, u, idx, status, ws, ws_bytes, s);
        if (rc == FK_ERR_UNSUPPORTED) return fail(rc, "resample: too many chunks for one launch");
        if (rc != FK_OK && rc != FK_ERR_LAUNCH) return fail(rc, "resample: one-pass launch failed");
        return rc;
    }
    // short vectors: one workgroup per filter, the carry local (resample_onepass.hip, resample_local_kernel);
    // FK_RESAMPLE_PATH=local sends every length there, FK_RESAMPLE_SERIAL=1 keeps the tile-by-tile kernel below (tests)
    if (!getenv("FK_RESAMPLE_SERIAL")) {
        const int rc = local_launch(stratified, Fn, Np, w, u, idx, status, s);
        if (rc == FK_ERR_UNSUPPORTED) return fail(rc, "resample: too many filters for one launch");
        return rc;
    }
    const dim3 grid((unsigned)Fn), block(RS_THREADS);
    if (stratified)
        hipLaunchKernelGGL((resample_kernel<true>), grid, block, 0, s, (long)Np, w, u, idx, status);
    else
        hipLaunchKernelGGL((resample_kernel<false>), grid, block, 0, s, (long)Np, w, u, idx, status);
    if (int rc = check_launch("resample_kernel")) return rc;
    return literal_fixup_launch(stratified, Fn, Np, w, u, idx, status, s);
}


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

int fk_resample_residual_fill_f64(int64_t Fn, int64_t Np, const double *w, int32_t *idx, int64_t *k, double *cs,
                                  int32_t *status, void *stream)
{
    if (Fn < 0 || Np < 0) return fail(FK_ERR_BAD_ARG, "residual: negative size");
    if (Np >= 2147483647LL || Fn > 0x7fffffffL) return fail(FK_ERR_UNSUPPORTED, "residual: Np must fit int32");
    if (Fn == 0 || Np == 0) return FK_OK;
    if (!w || !idx || !k || !cs) return fail(FK_ERR_BAD_ARG, "residual: w, idx, k, cs must not be NULL");
    hipLaunchKernelGGL(residual_fill_kernel, dim3((unsigned)Fn), dim3(RS_THREADS), 0, (hipStream_t)stream, (long)Np, w, idx,
                       (long *)k, cs, status);
    return check_launch("residual_fill_kernel");
}

int fk_resample_residual_draw_f64(int64_t Fn, int64_t Np, const double *cs, const int64_t *k, const int64_t *uoff,
                                  const double *u, int32_t *idx, void *stream)
{
    if (Fn < 0 || Np < 0) return fail(FK_ERR_BAD_ARG, "residual: negative size");
    if (Fn == 0 || Np == 0) return FK_OK;
    if (!cs || !k || !uoff || !u || !idx) return fail(FK_ERR_BAD_ARG, "residual: NULL argument");
    for (long f0 = 0; f0 < Fn; f0 += 65535) {                              // grid.y holds 65535 filters: larger banks in slices
        const long fc = Fn - f0 < 65535 ? Fn - f0 : 65535;
        const dim3 grid((unsigned)((Np + RS_THREADS - 1) / RS_THREADS), (unsigned)fc), block(RS_THREADS);
        hipLaunchKernelGGL(residual_draw_kernel, grid, block, 0, (hipStream_t)stream, (long)Np, cs, (const long *)k, (const long *)uoff, u, idx, f0);
    }
    return check_launch("residual_draw_kernel");
}

int fk_resample_gather_mean_f64(int64_t Fn, int64_t Np, int32_t d, const double *particles, const int32_t *idx,
                                double *mean, void *stream)
{
    if (Fn < 0 || Np < 0 || d < 1 || d > 8) return fail(FK_ERR_BAD_ARG, "gather_mean: Fn, Np >= 0 and 1 <= d <= 8");
    if (Fn == 0) return FK_OK;
    if (!mean || (Np > 0 && (!particles || !idx))) return fail(FK_ERR_BAD_ARG, "gather_mean: NULL argument");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(mean, 0, (size_t)Fn * d * sizeof(double), s) != hipSuccess) return fail(FK_ERR_LAUNCH, "gather_mean: memset failed");
    if (Np == 0) return FK_OK;
    // FK_GATHER_MEAN_WIDE=0: the first kernel also for 32-byte records (A/B)
    const char *wv = getenv("FK_GATHER_MEAN_WIDE");
    const bool wide = d == 4 && reinterpret_cast<uintptr_t>(particles) % 16 == 0 && !(wv && wv[0] == '0');
    for (long f0 = 0; f0 < Fn; f0 += 65535) {                              // grid.y holds 65535 filters: larger banks in slices
        const long fc = Fn - f0 < 65535 ? Fn - f0 : 65535;
        const dim3 block(RS_THREADS);
        if (wide) {
            const dim3 grid((unsigned)((Np + GM_CHUNK4 - 1) / GM_CHUNK4), (unsigned)fc);
            // sixteen independent index / record pairs per lane in flight (FK_GATHER_MEAN_U=8: eight -- 5.95 against 5.79 ms at
            // 125 x 8e6, profiles/r04/gather_mean.txt)
            const char *uv = getenv("FK_GATHER_MEAN_U");
            if (uv && atoi(uv) == 8) hipLaunchKernelGGL((gather_mean4_kernel<8>), grid, block, 0, s, (long)Np, particles, idx, mean, f0);
            else hipLaunchKernelGGL((gather_mean4_kernel<16>), grid, block, 0, s, (long)Np, particles, idx, mean, f0);
            continue;
        }
        const dim3 grid((unsigned)((Np + GM_CHUNK - 1) / GM_CHUNK), (unsigned)fc);
        if (d <= 4) hipLaunchKernelGGL((gather_mean_kernel<4>), grid, block, 0, s, (long)Np, particles, idx, mean, (int)d, f0);
        else hipLaunchKernelGGL((gather_mean_kernel<8>), grid, block, 0, s, (long)Np, particles, idx, mean, (int)d, f0);
    }
    return check_launch("gather_mean_kernel");
}

}  // extern "C"
