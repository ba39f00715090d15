// rts_kernels.hip -- Rauch-Tung-Striebel smoother kernels for gfx950 (MI355X).
//
// One track per lane, the backward time loop inside the kernel; the smoothed
// (x, P) of step k+1 stays in VGPRs for step k.  Per step the kernel reads the
// filtered (x_k, P_k) and writes the smoothed x, P, the gain K and the predicted
// covariance Pp: 8*(2n + 4n^2) algorithmic bytes per track-step.
//
// Replaces the backward loop of KalmanFilter.rts_smoother
// (filterpy/kalman/kalman_filter.py:1066-1072; F[k+1], Q[k+1]) and of the module
// function rts_smoother (:1851-1856; F[k], Q[k]) -- selected by conv_off.
#include "fk_device.hpp"
#include "fk_kernel_args.hpp"
#include "fk_math_sym.hpp"

#ifndef FK_NX
#error "compile with -DFK_NX=<dim_x> -DFK_EXACT=<0|1>"
#endif
// Exact-dimension instantiations keep every covariance as a packed upper triangle and stream the
// products row by row (fk_math_sym.hpp: no n x n temporary besides the gain K itself); the padded
// instantiations use the general full-matrix arithmetic of fk_math.hpp.
#ifndef FK_RTS_SYM
#define FK_RTS_SYM FK_EXACT
#endif

namespace fk {

constexpr int rts_min_waves(int nx) { return nx <= 2 ? 4 : nx <= 4 ? 2 : 1; }

// covariance state (full or packed upper triangle) <-> row-major NX x NX
template <int NX, bool SYM, int PL>
__device__ __forceinline__ void cov_expand(const double (&P)[PL], double (&M)[NX * NX])
{
    FK_UNROLL for (int a = 0; a < NX; ++a)
        FK_UNROLL for (int b = 0; b < NX; ++b) M[a * NX + b] = SYM ? P[sym_idx<NX>(a, b)] : P[a * NX + b];
}

template <int NX, bool SYM, int PL, int LAYOUT, bool EXACT>
__device__ __forceinline__ void cov_load(double (&P)[PL], const double *blk, const Lane &ln, int n)
{
    if constexpr (SYM) {
        const RecView<LAYOUT> pv(blk, ln, NX * NX);
        FK_UNROLL for (int i = 0; i < NX; ++i)
            FK_UNROLL for (int j = 0; j < NX; ++j)
                if (j >= i) P[sym_idx<NX>(i, j)] = pv.load(i * NX + j);
    } else {
        load_rec<NX, NX, LAYOUT, EXACT>(P, blk, ln, n, n, 1.0);
    }
}

template <int NX, bool EXACT, int LAYOUT, bool UNIFORM>
__global__ void __launch_bounds__(BLOCK, rts_min_waves(NX))
rts_kernel(const RtsArgs a, const double *__restrict__ pF, const double *__restrict__ pQ,
           const double *__restrict__ pXs, const double *__restrict__ pPs)
{
    using SharedModel = LdsModel<NX, 1>;
    using TrackModel = RegModel<NX, 1>;
    __shared__ double s_model[UNIFORM ? SharedModel::SIZE : 1];
    // AOS (NumPy order) outputs of the exact instantiations up to dim_x = 8 go through a wave-private LDS
    // tile and leave as 1 KiB stores (wave_store_aos, fk_device.hpp); tail lanes then duplicate the last
    // track and the buffer descriptor clips what they would write
    constexpr bool COOP = (LAYOUT == LAYOUT_AOS) && EXACT && NX <= 8;
    constexpr int TILE = COOP ? 64 * ((NX * NX) | 1) : 0;
    __shared__ double s_tile[COOP ? (BLOCK / 64) * TILE : 1];
    double *tile = s_tile + (threadIdx.x >> 6) * TILE;
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;

    const long N = a.N, T = a.T;
    const long cnt = a.cnt ? a.cnt : N;        // a track window of a larger bank (kf_dispatch.cpp: N stays the array stride)
    const long blk0 = (long)blockIdx.x * BLOCK;
    const unsigned last_row = (unsigned)(cnt - blk0 < BLOCK ? cnt - blk0 : BLOCK) - 1u;
    const Lane ln{blk0, COOP ? min(threadIdx.x, last_row) : threadIdx.x, N};
    const bool live = COOP || blk0 + ln.tid < cnt;
    const Lane lr{blk0, live ? ln.tid : 0u, N};
    // one record array of the step: x-like (LEN = NX) or covariance-like (LEN = NX * NX)
    auto put_x = [&](const double (&v)[NX], double *dst, long k) {
        if constexpr (COOP) wave_store_aos<NX>(v, dst + (k * N + blk0) * NX, wave * 64u, tile, lane, last_row);
        else store_rec<NX, 1, LAYOUT, EXACT>(v, dst + k * N * (EXACT ? NX : a.n), ln, EXACT ? NX : a.n, 1);
    };
    auto put_P = [&](const double (&v)[NX * NX], double *dst, long k) {
        if constexpr (COOP) wave_store_aos<NX * NX>(v, dst + (k * N + blk0) * NX * NX, wave * 64u, tile, lane, last_row);
        else store_rec<NX, NX, LAYOUT, EXACT>(v, dst + k * N * (long)(EXACT ? NX : a.n) * (EXACT ? NX : a.n), ln, EXACT ? NX : a.n, EXACT ? NX : a.n);
    };
    const int n = EXACT ? NX : a.n;
    const long xs_blk = N * n, ps_blk = N * (long)n * n;

    // k = T-1: smoothed == filtered; K = 0; Pp = Ps   (kalman_filter.py:1063-1065)
    constexpr bool SYM = (FK_RTS_SYM != 0) && EXACT;
    constexpr int PL = SYM ? NX * (NX + 1) / 2 : NX * NX;
    double xn[NX], Pn[PL];
    load_rec<NX, 1, LAYOUT, EXACT>(xn, pXs + (T - 1) * xs_blk, lr, n, 1, 0.0);
    cov_load<NX, SYM, PL, LAYOUT, EXACT>(Pn, pPs + (T - 1) * ps_blk, lr, n);
    if (live) {
        double Pf[NX * NX];
        cov_expand<NX, SYM, PL>(Pn, Pf);
        put_x(xn, a.xs, T - 1);
        put_P(Pf, a.Ps_out, T - 1);
        if (a.Pp) put_P(Pf, a.Pp, T - 1);
        if (a.K) {
            double Z[NX * NX];
            FK_UNROLL for (int i = 0; i < NX * NX; ++i) Z[i] = 0.0;
            put_P(Z, a.K, T - 1);
        }
    }

    TrackModel tm;
    const SharedModel sm{s_model};
    int st = 0;
    bool first = true;

    for (long k = T - 2; k >= 0; --k) {
        if (first || a.model_t) {
            const long mt = a.model_t ? k + a.conv_off : 0;
            if (UNIFORM) {
                if (!first) __syncthreads();
                lds_fill<NX, NX>(s_model + SharedModel::OFF_F, pF + mt * n * n, n, n, 1.0, threadIdx.x);
                lds_fill<NX, NX>(s_model + SharedModel::OFF_Q, pQ + mt * n * n, n, n, 0.0, threadIdx.x);
                __syncthreads();
            } else {
                load_rec<NX, NX, LAYOUT, EXACT>(tm.F, pF + mt * ps_blk, lr, n, n, 1.0);
                load_rec<NX, NX, LAYOUT, EXACT>(tm.Q, pQ + mt * ps_blk, lr, n, n, 0.0);
            }
            first = false;
        }
        double x[NX], P[PL], K[NX * NX];
        load_rec<NX, 1, LAYOUT, EXACT>(x, pXs + k * xs_blk, lr, n, 1, 0.0);
        cov_load<NX, SYM, PL, LAYOUT, EXACT>(P, pPs + k * ps_blk, lr, n);
        if constexpr (SYM) {
            // Pp is streamed out the moment it is complete (its registers are reused by the solve)
            auto pp_sink = [&](const double (&Ppk)[PL]) {
                if (live && a.Pp) {
                    double Pf[NX * NX];
                    cov_expand<NX, SYM, PL>(Ppk, Pf);
                    put_P(Pf, a.Pp, k);
                }
            };
            if (UNIFORM) st |= rts_step_sym<NX>(x, P, xn, Pn, sm, K, pp_sink);
            else st |= rts_step_sym<NX>(x, P, xn, Pn, tm, K, pp_sink);
        } else {
            double Pp[PL];
            if (UNIFORM) st |= rts_step<NX>(x, P, xn, Pn, sm, K, Pp);
            else st |= rts_step<NX>(x, P, xn, Pn, tm, K, Pp);
            if (live && a.Pp) put_P(Pp, a.Pp, k);
        }
        if (live) {
            put_x(x, a.xs, k);
            if (a.K) put_P(K, a.K, k);
            double Pf[NX * NX];
            cov_expand<NX, SYM, PL>(P, Pf);
            put_P(Pf, a.Ps_out, k);
        }
        FK_UNROLL for (int i = 0; i < NX; ++i) xn[i] = x[i];
        FK_UNROLL for (int i = 0; i < PL; ++i) Pn[i] = P[i];
    }
    if (live && a.status) {
        if (!all_finite<NX>(xn) || !all_finite<PL>(Pn)) st |= ST_NONFINITE;
        a.status[ln.blk0 + ln.tid] = st;
    }
}

template <int NX, bool EXACT>
static int launch(const RtsArgs &a, int layout, bool uniform, hipStream_t stream)
{
    const dim3 grid((unsigned)(((a.cnt ? a.cnt : a.N) + BLOCK - 1) / BLOCK)), block(BLOCK);
#define FK_GO(LAY, UNI) \
    hipLaunchKernelGGL((rts_kernel<NX, EXACT, LAY, UNI>), grid, block, 0, stream, a, a.F, a.Q, a.Xs, a.Ps)
    if (layout == LAYOUT_SOA) {
        if (uniform) FK_GO(LAYOUT_SOA, true);
        else FK_GO(LAYOUT_SOA, false);
    } else {
        if (uniform) FK_GO(LAYOUT_AOS, true);
        else FK_GO(LAYOUT_AOS, false);
    }
#undef FK_GO
    return check_launch("rts_kernel");
}

#define FK_CAT_(a, b, c) a##b##_##c
#define FK_CAT(a, b, c) FK_CAT_(a, b, c)

int FK_CAT(launch_rts_, FK_NX, FK_EXACT)(const RtsArgs &a, int layout, bool uniform, hipStream_t stream)
{
    return launch<FK_NX, (FK_EXACT != 0)>(a, layout, uniform, stream);
}

}  // namespace fk
