// EXPERIMENTAL -- not part of libfilterhip.so, not validated on a GPU yet (written at the end of round 1 when the
// GPU budget was spent).  Built only by tools/exp_lean2.py into build/libfk_exp.so, next to an unmodified copy
// of the resampling unit, so that the next round can compare it bit for bit and time it against the shipped path
// in one call.
//
// resample_chunk_lean2_kernel: the lean output kernel (resample_chunk_lean_kernel, ../resample_kernels.hip)
// restructured after what tools/rs_phase.py measured on it (output loop 44 %, scan 34 %, count_below 16 %):
//   * slab ownership: thread t owns elements t, t+256, ... of the tile -- they come straight from coalesced
//     global loads into registers, the raw weights never visit LDS, and the cumulative sums are written to the
//     LDS tile with stride-1 (conflict-free) stores.  (The shipped kernel's thread-owns-8-consecutive layout
//     makes each of its 16 scan accesses a 16-way bank conflict.)  The price is eight wave scans instead of one;
//     they run on DPP row shifts / row broadcasts, not on the LDS crossbar.
//   * the increments are integers held in doubles (fast_inc) and every partial sum is below 2^53 or the chunk
//     is handed to the general kernel, so the different association order gives the same bits.
//   * eight CONSECUTIVE output slots per thread: one interpolated search (tile_upper_bound) for the first, then
//     a walk -- positions and cumulative sums are both non-decreasing, the next index is the previous one plus
//     the few elements in between (one on average).
#include "../resample_kernels.hip"

namespace fk {

// inclusive scan over the 64 lanes of a wave, DPP only (gfx9 row_shr / row_bcast), for values whose sum is exact
__device__ __forceinline__ double wave_inclusive_sum_dpp(double v)
{
    auto dpp_add = [](double acc, auto ctrl_tag, int row_mask) {
        constexpr int CTRL = decltype(ctrl_tag)::value;
        const int lo = __double2loint(acc), hi = __double2hiint(acc);
        // lanes without a source (or outside row_mask) receive `old` = 0 -> +0.0
        const int slo = row_mask == 0xf ? __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false)
                      : row_mask == 0xa ? __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xa, 0xf, false)
                                        : __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xc, 0xf, false);
        const int shi = row_mask == 0xf ? __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false)
                      : row_mask == 0xa ? __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xa, 0xf, false)
                                        : __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xc, 0xf, false);
        return acc + __hiloint2double(shi, slo);
    };
    v = dpp_add(v, std::integral_constant<int, 0x111>{}, 0xf);   // row_shr:1
    v = dpp_add(v, std::integral_constant<int, 0x112>{}, 0xf);   // row_shr:2
    v = dpp_add(v, std::integral_constant<int, 0x114>{}, 0xf);   // row_shr:4
    v = dpp_add(v, std::integral_constant<int, 0x118>{}, 0xf);   // row_shr:8   -> inclusive within rows of 16
    v = dpp_add(v, std::integral_constant<int, 0x142>{}, 0xa);   // row_bcast:15 into rows 1, 3
    v = dpp_add(v, std::integral_constant<int, 0x143>{}, 0xc);   // row_bcast:31 into rows 2, 3
    return v;
}

struct Lean2Shared {
    double tile[2 + RS_TILE + TILE_GUARD];          // guarded like ScanShared::tile
    double slab_wave[RS_ITEMS][RS_THREADS / 64];    // total of (slab q, wave)
    __device__ __forceinline__ double *w() { return tile + 2; }
};

template <bool STRATIFIED>
__global__ void __launch_bounds__(RS_THREADS)
resample_chunk_lean2_kernel(long Np, long nch, const double *__restrict__ w, const double *__restrict__ u,
                            ChunkPlan *__restrict__ plan, int32_t *__restrict__ idx, int32_t *__restrict__ status)
{
    __shared__ Lean2Shared sh;
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
    double v[RS_ITEMS];
    fetch_tile(v, wf + base, len);                  // v[q] = element q*256 + tid (+inf past len)
    const double c_in = p.cin;
    const bool can = p.started != 0 && p.prelude == 0 && c_in > 0.0 && c_in <= 1.79769313486231570815e+308 && k > 0;
    if (!can) {                                                       // uniform
        if (tid == 0) p.todo = 1;
        return;
    }
    const double ulp = ulp_of(c_in);
    const int eu = ulp_exp(c_in);
    const double C0d = scale2(c_in, -eu);
    const int out_lo = count_below<STRATIFIED>(c_in, 0, (int)Np, dNp, u_sys, u_str);

    // per slab: increments, inclusive scan inside the wave, wave totals to LDS
    double s[RS_ITEMS];
    bool odd = false;
    FK_UNROLL for (int q = 0; q < RS_ITEMS; ++q) {
        const int j = q * RS_THREADS + tid;
        bool tk = false;
        const double e = j < len ? fast_inc(v[q], eu, tk) : 0.0;
        odd = odd || tk;
        s[q] = wave_inclusive_sum_dpp(e);
        if (lane == 63) sh.slab_wave[q][wave] = s[q];
    }
    if (tid < TILE_GUARD) sh.w()[RS_TILE + tid] = __builtin_inf();
    if (tid == 0) sh.w()[-1] = -__builtin_inf();
    __syncthreads();
    // running offset in tile order: slab 0 wave 0..3, slab 1 wave 0..3, ...  (broadcast LDS reads)
    double run = 0.0;
    FK_UNROLL for (int q = 0; q < RS_ITEMS; ++q) {
        double mine = 0.0;
        FK_UNROLL for (int wv = 0; wv < RS_THREADS / 64; ++wv) {
            if (wv == wave) mine = run;
            run += sh.slab_wave[q][wv];
        }
        const int j = q * RS_THREADS + tid;
        s[q] = C0d + (mine + s[q]);                                   // C of element j, an integer < 2^53 or:
        odd = odd || (j < len && !(s[q] < 0x1p53));
    }
    if (__syncthreads_or(odd ? 1 : 0)) {                              // uniform
        if (tid == 0) p.todo = 1;
        return;
    }
    FK_UNROLL for (int q = 0; q < RS_ITEMS; ++q) {
        const int j = q * RS_THREADS + tid;
        sh.w()[j] = j < len ? s[q] * ulp : __builtin_inf();           // stride-1: conflict-free
    }
    __syncthreads();
    const double carry = sh.w()[len - 1];
    const int out_hi = count_below<STRATIFIED>(carry, out_lo, (int)Np, dNp, u_sys, u_str);
    const double inv_span = (double)len / (carry - c_in);
    // RS_ITEMS consecutive slots per thread; a tile can cover more than RS_TILE slots (heavy weights elsewhere
    // make the positions denser than the sums here), hence the outer loop
    for (int i0 = out_lo + tid * RS_ITEMS; i0 < out_hi; i0 += RS_TILE) {
        int r = 0;
        FK_UNROLL for (int e = 0; e < RS_ITEMS; ++e) {
            const int i = i0 + e;
            if (i < out_hi) {
                const double ps = position<STRATIFIED>(i, dNp, u_sys, u_str);
                if (e == 0) {
                    r = tile_upper_bound(sh.w(), len, ps, c_in, inv_span);
                } else {
                    while (sh.w()[r] <= ps) ++r;                       // +inf past len stops it
                }
                of[i] = (int32_t)(base + r);
            }
        }
    }
    if (k == nch - 1) {
        for (int i = out_hi + tid; i < (int)Np; i += RS_THREADS) of[i] = (int32_t)(Np - 1);
        if (tid == 0 && status) status[f] = out_hi < (int)Np ? ST_OVERRUN : 0;
    }
    if (tid == 0) p.todo = 0;
}

}  // namespace fk

// systematic / stratified resampling with the experimental lean kernel in place of the shipped one (long vectors
// only: the caller's workspace is required); everything else -- plan, chain, general kernel -- is the shipped code
extern "C" int fk_exp_resample_lean2_f64(int32_t stratified, int64_t Fn, int64_t Np, const double *w, const double *u,
                                         int32_t *idx, int32_t *status, void *ws, size_t ws_bytes, void *stream)
{
    using namespace fk;
    hipStream_t s = (hipStream_t)stream;
    const long nch = (long)((Np + RS_TILE - 1) / RS_TILE);
    if (Np < RS_PAR_MIN || !ws || ws_bytes < (size_t)Fn * (size_t)nch * sizeof(ChunkPlan)) return FK_ERR_WORKSPACE;
    ChunkPlan *plan = (ChunkPlan *)ws;
    const dim3 gch((unsigned)nch, (unsigned)Fn), block(RS_THREADS);
    hipLaunchKernelGGL(chunk_sum_kernel, gch, block, 0, s, (long)Np, nch, w, plan);
    hipLaunchKernelGGL(chunk_plan_kernel, dim3((unsigned)Fn), block, 0, s, (long)Np, nch, plan);
    hipLaunchKernelGGL(chunk_compose_kernel, gch, block, 0, s, (long)Np, nch, w, plan);
    hipLaunchKernelGGL(chain_kernel, dim3((unsigned)Fn), block, 0, s, (long)Np, nch, w, plan);
    if (stratified) {
        hipLaunchKernelGGL((resample_chunk_lean2_kernel<true>), gch, block, 0, s, (long)Np, nch, w, u, plan, idx, status);
        hipLaunchKernelGGL((resample_chunk_kernel<true>), gch, block, 0, s, (long)Np, nch, w, u, plan, idx, status);
    } else {
        hipLaunchKernelGGL((resample_chunk_lean2_kernel<false>), gch, block, 0, s, (long)Np, nch, w, u, plan, idx, status);
        hipLaunchKernelGGL((resample_chunk_kernel<false>), gch, block, 0, s, (long)Np, nch, w, u, plan, idx, status);
    }
    return check_launch("resample_chunk_lean2_kernel");
}
