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
#include "../../include/filterhip.h"
#include "fk_device.hpp"
#include "fk_exact_scan.hpp"

namespace fk {

constexpr int RS_THREADS = 256;
constexpr int RS_ITEMS = 8;
constexpr int RS_TILE = RS_THREADS * RS_ITEMS;   // 2048 weights per tile

struct ScanShared {
    double w[RS_TILE];          // weights of the tile, then (in place) their cumulative sums
    Mono wave_tot[RS_THREADS / 64];
    double carry;               // exact running sum entering the next segment
    int first_cross;            // first tile index whose add leaves the binade (RS_TILE = none)
    int first_nonzero;
};

__device__ __forceinline__ Mono shfl_up_mono(const Mono &m, int delta)
{
    Mono r;
    r.ae = __shfl_up(m.ae, delta, 64);
    r.ao = __shfl_up(m.ao, delta, 64);
    return r;
}

// In-place exact inclusive prefix sum of sh.w[0..len) continuing from the running sum `carry`
// (`started` = false means no element has been summed yet: cs[0] = w[0], like numpy.cumsum).
// All RS_THREADS threads participate.  Returns the running sum after the tile.
__device__ double tile_cumsum_exact(ScanShared &sh, int len, double carry, bool &started)
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
                    if (sh.w[j] != 0.0) { mine = j; break; }
                if (mine < len) atomicMin(&sh.first_nonzero, mine);
                __syncthreads();
                const int j0 = sh.first_nonzero;
                __syncthreads();
                // elements pos..j0-1 are zero weights: their cumulative sum is +0.0 (0+0, or w itself)
                if (j0 < len) {
                    // 0 + w = w exactly (also for negative / NaN w)
                    carry = started ? carry + sh.w[j0] : sh.w[j0];
                    started = true;
                    pos = j0 + 1;      // sh.w[j0] already holds its own cumulative sum
                } else {
                    started = started || len > pos;
                    if (started && !(carry == 0.0)) carry = 0.0;
                    pos = len;
                }
                continue;
            }
            // negative, NaN or infinite running sum: plain sequential adds
            carry = carry + sh.w[pos];
            __syncthreads();
            if (tid == 0) sh.w[pos] = carry;
            __syncthreads();
            ++pos;
            continue;
        }

        // --- one binade: parallel exact scan over [pos, len) ---------------------------------
        const double u = ulp_of(carry);
        const long long C0 = (long long)(carry / u);
        Mono loc[RS_ITEMS];
        Mono run = mono_identity();
        FK_UNROLL for (int k = 0; k < RS_ITEMS; ++k) {
            const int j = tid * RS_ITEMS + k;
            const Mono e = (j >= pos && j < len) ? mono_elem(sh.w[j], u) : mono_identity();
            run = mono_compose(run, e);
            loc[k] = run;
        }
        // wave-level inclusive scan of the thread totals
        Mono inc = run;
        FK_UNROLL for (int d = 1; d < 64; d <<= 1) {
            const Mono up = shfl_up_mono(inc, d);
            if (lane >= d) inc = mono_compose(up, inc);
        }
        if (tid == 0) sh.first_cross = RS_TILE;
        if (lane == 63) sh.wave_tot[wave] = inc;
        __syncthreads();
        Mono excl = shfl_up_mono(inc, 1);
        if (lane == 0) excl = mono_identity();
        Mono wprefix = mono_identity();
        for (int wv = 0; wv < wave; ++wv) wprefix = mono_compose(wprefix, sh.wave_tot[wv]);
        excl = mono_compose(wprefix, excl);
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
            if (j >= pos && j < cross) sh.w[j] = (double)Cj[k] * u;       // exact
        }
        __syncthreads();
        if (cross > pos) carry = sh.w[cross - 1];
        if (cross < len) {
            // the add that leaves the binade: a real IEEE add
            carry = carry + sh.w[cross];
            __syncthreads();
            if (tid == 0) sh.w[cross] = carry;
            __syncthreads();
            pos = cross + 1;
        } else {
            pos = len;
        }
    }
    __syncthreads();
    return carry;
}

template <bool STRATIFIED>
__device__ __forceinline__ double position(long i, double dNp, double u_sys, const double *__restrict__ u_str)
{
    const double ui = STRATIFIED ? u_str[i] : u_sys;
    return (ui + (double)i) / dNp;      // fl(fl(u + i) / Np): resampling.py:103,139
}

// number of output slots whose position is < c  (= first i with pos_i >= c), searched upward
// from `lo` around the estimate c*Np - u
template <bool STRATIFIED>
__device__ long count_below(double c, long lo, long Np, double dNp, double u_sys, const double *__restrict__ u_str)
{
    double est = c * dNp - (STRATIFIED ? 0.5 : u_sys);
    long k = est <= (double)lo ? lo : (est >= (double)Np ? Np : (long)est);
    while (k > lo && !(position<STRATIFIED>(k - 1, dNp, u_sys, u_str) < c)) --k;
    while (k < Np && position<STRATIFIED>(k, dNp, u_sys, u_str) < c) ++k;
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
    long out_lo = 0;     // output slots [0, out_lo) are done
    for (long base = 0; base < Np; base += RS_TILE) {
        const int len = (int)((Np - base) < RS_TILE ? (Np - base) : RS_TILE);
        for (int j = tid; j < RS_TILE; j += RS_THREADS) sh.w[j] = j < len ? wf[base + j] : 0.0;
        __syncthreads();
        carry = tile_cumsum_exact(sh, len, carry, started);
        // slots covered by this tile: pos_i < cs_last  (cs is non-decreasing for weights >= 0)
        const long out_hi = count_below<STRATIFIED>(carry, out_lo, Np, dNp, u_sys, u_str);
        for (long i = out_lo + tid; i < out_hi; i += RS_THREADS) {
            const double p = position<STRATIFIED>(i, dNp, u_sys, u_str);
            // idx = #{ j : cs_j <= p }  (upper bound; the two-pointer merge of resampling.py:143-149)
            int lo = 0, hi = len;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (sh.w[mid] <= p) lo = mid + 1;
                else hi = mid;
            }
            of[i] = (int32_t)(base + lo);
        }
        out_lo = out_hi;
        __syncthreads();
    }
    // positions >= cumsum[-1]: the reference raises IndexError (resampling.py:109,145)
    for (long i = out_lo + tid; i < Np; i += RS_THREADS) of[i] = (int32_t)(Np - 1);
    if (tid == 0 && status) status[f] = out_lo < Np ? ST_OVERRUN : 0;
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
    for (long base = 0; base < Np; base += RS_TILE) {
        const int len = (int)((Np - base) < RS_TILE ? (Np - base) : RS_TILE);
        for (int j = tid; j < RS_TILE; j += RS_THREADS) sh.w[j] = j < len ? wf[base + j] : 0.0;
        __syncthreads();
        carry = tile_cumsum_exact(sh, len, carry, started);
        for (int j = tid; j < len; j += RS_THREADS) cf[base + j] = sh.w[j];
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
    // only multinomial needs scratch: the cumulative sums, Fn * Np doubles
    return (Fn > 0 && Np > 0) ? (size_t)Fn * (size_t)Np * sizeof(double) : 0;
}

static int resample_common(bool stratified, int64_t Fn, int64_t Np, const double *w, const double *u,
                           int32_t *idx, int32_t *status, void *stream)
{
    if (Fn < 0 || Np < 0) return fail(FK_ERR_BAD_ARG, "resample: negative size");
    if (Np >= 2147483647LL) return fail(FK_ERR_UNSUPPORTED, "resample: Np must fit int32 (the reference returns int32 indices)");
    if (Fn == 0 || Np == 0) return FK_OK;
    if (!w || !u || !idx) return fail(FK_ERR_BAD_ARG, "resample: w, u, idx must not be NULL");
    const dim3 grid((unsigned)Fn), block(RS_THREADS);
    if (stratified)
        hipLaunchKernelGGL((resample_kernel<true>), grid, block, 0, (hipStream_t)stream, (long)Np, w, u, idx, status);
    else
        hipLaunchKernelGGL((resample_kernel<false>), grid, block, 0, (hipStream_t)stream, (long)Np, w, u, idx, status);
    return check_launch("resample_kernel");
}

int fk_resample_systematic_f64(int64_t Fn, int64_t Np, const double *w, const double *u, int32_t *idx,
                               int32_t *status, void *, size_t, void *stream)
{
    return resample_common(false, Fn, Np, w, u, idx, status, stream);
}

int fk_resample_stratified_f64(int64_t Fn, int64_t Np, const double *w, const double *u, int32_t *idx,
                               int32_t *status, void *, size_t, void *stream)
{
    return resample_common(true, Fn, Np, w, u, idx, status, stream);
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
    if (!ws || ws_bytes < fk_resample_workspace_bytes(Fn, Np)) return fail(FK_ERR_WORKSPACE, "multinomial: workspace too small");
    double *cs = (double *)ws;
    if (int rc = fk_cumsum_exact_f64(Fn, Np, w, cs, 1, stream)) return rc;
    const dim3 grid((unsigned)((Nu + RS_THREADS - 1) / RS_THREADS), (unsigned)Fn), block(RS_THREADS);
    hipLaunchKernelGGL(searchsorted_left_kernel, grid, block, 0, (hipStream_t)stream, (long)Np, (long)Nu, cs, u, idx);
    return check_launch("searchsorted_left_kernel");
}

}  // extern "C"
