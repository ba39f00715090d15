// kf_fast.hip -- the headline kernel: KalmanFilter.batch_filter for a bank of tracks with ONE
// model shared by every track and step (BASELINE.json configs[1] / SURVEY.md §8 row K3).
//
// Same arithmetic as kf_kernels.hip (fk_math.hpp: filterpy/kalman/kalman_filter.py:472-478,
// 533-556, loop :980-991), specialised for the common call -- predict then update, no control
// input, constant shared F/Q/H/R -- so that the per-step instruction stream is nothing but
//   1 measurement load (software-pipelined one step ahead, counted vmcnt: the wave never
//     drains its own stores), ~480 fp64 FMAs, 22 LDS broadcast reads of the model,
//   the four output records (x-, P-, x+, P+) streamed straight to HBM.
// Layouts:
//   SOA  a[t][e][i]: each store instruction covers 64 consecutive tracks of one element
//        (512 B contiguous per wave);
//   AOS  a[t][i][e] (NumPy C order): records are transposed through a wave-private LDS tile so
//        every store instruction still writes 1 KiB of contiguous memory (16 B per lane).
// HBM-bound: 8*(m + 2n + 2n^2) algorithmic bytes per track-step (336 B at n=4, m=2).
#include "fk_device.hpp"
#include "fk_kernel_args.hpp"
#include "fk_math_sym.hpp"

#ifndef FK_NX
#error "compile with -DFK_NX=<dim_x> -DFK_NZ=<dim_z> -DFK_VARIANT=<v> -DFK_FAST_WAVES=<w>"
#endif
#ifndef FK_FAST_WAVES
#define FK_FAST_WAVES 0
#endif
// FK_FAST_SYM=1: covariance kept as a packed upper triangle (fk_math_sym.hpp) -- fewer registers
// and flops; reads only the upper triangle of P0 and writes a mirrored P.
#ifndef FK_FAST_SYM
#define FK_FAST_SYM 0
#endif

#ifndef FK_VARIANT
#define FK_VARIANT 0
#endif
#define FK_CAT_(a, b, c, d) a##b##_##c##_v##d
#define FK_CAT(a, b, c, d) FK_CAT_(a, b, c, d)

namespace fk {
namespace FK_CAT(fastv_, FK_NX, FK_NZ, FK_VARIANT) {

constexpr int fast_min_waves(int nx)
{
    return FK_FAST_WAVES ? FK_FAST_WAVES : (nx <= 2 ? 8 : nx <= 4 ? 3 : nx <= 6 ? 2 : 1);
}

using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

__device__ __forceinline__ void wave_lds_fence()
{
    // LDS operations of one wave execute in order; this only stops the compiler from
    // reordering the tile writes and the transposed reads around each other.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Wave-cooperative store of one LEN-double record per lane into an AOS block:
// lane l owns the record of track (block_first + wave_row0 + l); `rs` is the descriptor of the
// workgroup's slab.  Rows past `last_row` (the block's last valid track; only in the tail
// workgroup, where those lanes carry a duplicate of that track) are redirected onto it:
// the same bytes are written twice instead of predicating the store.
// The tile is written row-per-lane (row stride LEN|1 doubles: conflict-free ds_write_b64) and read
// back in memory order, two consecutive doubles per lane per pass -> buffer_store_dwordx4,
// 1 KiB contiguous per instruction.
template <int LEN>
__device__ __forceinline__ void wave_store_aos(const double (&v)[LEN], rsrc_t rs, unsigned wave_row0,
                                               double *tile, unsigned lane, unsigned last_row)
{
    constexpr int LENP = LEN | 1;
    FK_UNROLL for (int e = 0; e < LEN; ++e) tile[lane * LENP + e] = v[e];
    wave_lds_fence();
    if constexpr (LEN % 2 == 0) {
        constexpr int PASSES = LEN / 2;          // 64*LEN doubles, 128 per pass
        FK_UNROLL for (int it = 0; it < PASSES; ++it) {
            const unsigned q = it * 128u + lane * 2u;
            const unsigned row = q / LEN, col = q % LEN;
            const double a = tile[row * LENP + col], b = tile[row * LENP + col + 1];
            const unsigned grow = min(wave_row0 + row, last_row);
            const u32x2 ua = __builtin_bit_cast(u32x2, a), ub = __builtin_bit_cast(u32x2, b);
            const u32x4 w = {ua.x, ua.y, ub.x, ub.y};
            __builtin_amdgcn_raw_buffer_store_b128(w, rs, (grow * LEN + col) * 8u, 0, 0);
        }
    } else {
        FK_UNROLL for (int it = 0; it < LEN; ++it) {
            const unsigned q = it * 64u + lane;
            const unsigned row = q / LEN, col = q % LEN;
            const double a = tile[row * LENP + col];
            const unsigned grow = min(wave_row0 + row, last_row);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, a), rs, (grow * LEN + col) * 8u, 0, 0);
        }
    }
    wave_lds_fence();
}

// OUTS: true = all four outputs (means, covs, means_p, covs_p) are stored every step; false =
// none (only the final state).  No store is ever predicated: in the tail workgroup the lanes past
// the last track recompute that track and write the same bytes again.  The number of stores
// between a measurement load and its use is therefore a compile-time constant and the wait for
// it is a counted vmcnt that never drains the store queue.
// expand the covariance state (full or packed upper triangle) to a row-major NX x NX array;
// pure register renaming after optimisation
template <int NX, bool SYM, int PLEN>
__device__ __forceinline__ void cov_full(const double (&P)[PLEN], double (&M)[NX * NX])
{
    FK_UNROLL for (int a = 0; a < NX; ++a)
        FK_UNROLL for (int b = 0; b < NX; ++b) M[a * NX + b] = SYM ? P[sym_idx<NX>(a, b)] : P[a * NX + b];
}

template <int NX, int NZ, int LAYOUT, bool HAS_MASK, bool OUTS, bool SYM>
__global__ void __launch_bounds__(BLOCK, fast_min_waves(NX))
kf_fast_kernel(const KfArgs a, const double *__restrict__ pF, const double *__restrict__ pQ,
               const double *__restrict__ pH, const double *__restrict__ pR,
               const double *__restrict__ pz, const uint8_t *__restrict__ pmask)
{
    using SharedModel = LdsModel<NX, NZ>;
    // AOS records go through the LDS transpose while a wave tile fits (dim_x <= 6); larger
    // records fall back to per-lane stores (TODO: chunked tiles)
    constexpr bool COOP = (LAYOUT == LAYOUT_AOS) && (NX * NX <= 36);
    constexpr int TILE = COOP ? 64 * ((NX * NX) | 1) : 0;   // doubles per wave
    __shared__ double s_mem[SharedModel::SIZE + (BLOCK / 64) * TILE + 1];

    const long N = a.N, T = a.T;
    const long blk0 = a.i0 + (long)blockIdx.x * BLOCK;
    const unsigned tid = threadIdx.x;
    const long left = a.i0 + a.cnt - blk0;                       // >= 1
    const unsigned last_row = (unsigned)(left < BLOCK ? left : BLOCK) - 1u;
    const Lane ln{blk0, min(threadIdx.x, last_row), N};          // tail lanes duplicate the last track
    const Lane lr = ln;
    const unsigned lane = tid & 63u, wave = tid >> 6;
    double *tile = s_mem + SharedModel::SIZE + wave * TILE;

    lds_fill<NX, NX>(s_mem + SharedModel::OFF_F, pF, NX, NX, 1.0, tid);
    lds_fill<NX, NX>(s_mem + SharedModel::OFF_Q, pQ, NX, NX, 0.0, tid);
    lds_fill<NZ, NX>(s_mem + SharedModel::OFF_H, pH, NZ, NX, 0.0, tid);
    lds_fill<NZ, NZ>(s_mem + SharedModel::OFF_R, pR, NZ, NZ, 1.0, tid);
    __syncthreads();
    const SharedModel sm{s_mem};

    constexpr int PLEN = SYM ? NX * (NX + 1) / 2 : NX * NX;
    double x[NX], P[PLEN];      // SYM: P is the packed upper triangle
    load_rec<NX, 1, LAYOUT, true>(x, a.x, lr, NX, 1, 0.0);
    {
        const RecView<LAYOUT> pv(a.P, lr, NX * NX);
        FK_UNROLL for (int i = 0; i < NX; ++i)
            FK_UNROLL for (int j = (SYM ? i : 0); j < NX; ++j)
                P[SYM ? sym_idx<NX>(i, j) : i * NX + j] = pv.load(i * NX + j);
    }

    // measurement pipeline: z[t+1] is requested at the top of step t
    double zc[NZ], zn[NZ];
    bool hc = true, hn = true;
    load_rec<NZ, 1, LAYOUT, true>(zc, pz, lr, NZ, 1, 0.0);
    if (HAS_MASK) hc = pmask[lr.blk0 + lr.tid] != 0;
    // Land every prologue load before the loop: a load still pending at the loop header would
    // make the compiler wait vmcnt(0) inside the loop on every iteration (draining the stores).
    FK_UNROLL for (int i = 0; i < NX; ++i) asm volatile("" ::"v"(x[i]));
    FK_UNROLL for (int i = 0; i < PLEN; ++i) asm volatile("" ::"v"(P[i]));
    FK_UNROLL for (int i = 0; i < NZ; ++i) asm volatile("" ::"v"(zc[i]));
    FK_UNROLL for (int i = 0; i < NZ; ++i) zn[i] = zc[i];

    int st = 0;
    for (long t = 0; t < T; ++t) {
        if (t + 1 < T) {
            load_rec<NZ, 1, LAYOUT, true>(zn, pz + (t + 1) * N * NZ, lr, NZ, 1, 0.0);
            if (HAS_MASK) hn = pmask[(t + 1) * N + lr.blk0 + lr.tid] != 0;
        }
        if constexpr (SYM) kf_predict_sym<NX>(x, P, sm, a.alpha_sq);
        else kf_predict<NX>(x, P, sm, a.alpha_sq);
        double Pf[NX * NX];
        cov_full<NX, SYM, PLEN>(P, Pf);
        if (!OUTS) {
        } else if (!COOP) {
            store_rec<NX, 1, LAYOUT, true>(x, a.means_p + t * N * NX, ln, NX, 1);
            store_rec<NX, NX, LAYOUT, true>(Pf, a.covs_p + t * N * NX * NX, ln, NX, NX);
        } else {
            wave_store_aos<NX>(x, make_rsrc(a.means_p + (t * N + blk0) * NX), wave * 64u, tile, lane, last_row);
            wave_store_aos<NX * NX>(Pf, make_rsrc(a.covs_p + (t * N + blk0) * NX * NX), wave * 64u, tile, lane, last_row);
        }
        if (hc) {
            double K[NX * NZ], y[NZ], S[NZ * NZ], Lf[NZ * NZ], dinv[NZ];
            if constexpr (SYM) st |= kf_update_sym<NX, NZ>(x, P, zc, sm, K, y, S, Lf, dinv);
            else st |= kf_update<NX, NZ>(x, P, zc, sm, K, y, S, Lf, dinv);
        }
        cov_full<NX, SYM, PLEN>(P, Pf);
        if (!OUTS) {
        } else if (!COOP) {
            store_rec<NX, 1, LAYOUT, true>(x, a.means + t * N * NX, ln, NX, 1);
            store_rec<NX, NX, LAYOUT, true>(Pf, a.covs + t * N * NX * NX, ln, NX, NX);
        } else {
            wave_store_aos<NX>(x, make_rsrc(a.means + (t * N + blk0) * NX), wave * 64u, tile, lane, last_row);
            wave_store_aos<NX * NX>(Pf, make_rsrc(a.covs + (t * N + blk0) * NX * NX), wave * 64u, tile, lane, last_row);
        }
        FK_UNROLL for (int i = 0; i < NZ; ++i) zc[i] = zn[i];
        hc = hn;
    }

    store_rec<NX, 1, LAYOUT, true>(x, a.x, ln, NX, 1);
    double Pl[NX * NX];
    cov_full<NX, SYM, PLEN>(P, Pl);
    store_rec<NX, NX, LAYOUT, true>(Pl, a.P, ln, NX, NX);
    if (a.status) {
        if (!all_finite<NX>(x) || !all_finite<PLEN>(P)) st |= ST_NONFINITE;
        a.status[blk0 + ln.tid] = st;
    }
}



}  // namespace (variant)
using namespace FK_CAT(fastv_, FK_NX, FK_NZ, FK_VARIANT);

// Handles tracks [a.i0, a.i0 + a.cnt).  outs: all four outputs
// non-NULL (true) or all NULL (false).
int FK_CAT(launch_kf_fast_, FK_NX, FK_NZ, FK_VARIANT)(const KfArgs &a, int layout, bool outs, hipStream_t stream)
{
    const dim3 grid((unsigned)((a.cnt + BLOCK - 1) / BLOCK)), block(BLOCK);
#define FK_GO(LAY, MSK, OUT)                                                                               \
    hipLaunchKernelGGL((kf_fast_kernel<FK_NX, FK_NZ, LAY, MSK, OUT, (FK_FAST_SYM != 0)>), grid, block, 0, stream, a, a.F, a.Q, \
                       a.H, a.R, a.z, a.mask)
#define FK_GO2(LAY)                          \
    do {                                     \
        if (a.mask) {                        \
            if (outs) FK_GO(LAY, true, true); \
            else FK_GO(LAY, true, false);    \
        } else {                             \
            if (outs) FK_GO(LAY, false, true); \
            else FK_GO(LAY, false, false);   \
        }                                    \
    } while (0)
    if (layout == LAYOUT_SOA) FK_GO2(LAYOUT_SOA);
    else FK_GO2(LAYOUT_AOS);
#undef FK_GO2
#undef FK_GO
    return check_launch("kf_fast_kernel");
}

}  // namespace fk
