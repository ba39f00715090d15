// kf_given_inv.hip -- update() and rts_smoother() around a CALLER-SUPPLIED inverse (gfx950).
//
// filterpy.kalman.KalmanFilter documents `kf.inv = numpy.linalg.pinv` (kalman_filter.py:363, 434) and uses whatever the
// attribute names on S (:541); rts_smoother takes `inv=` (:995, 1069).  A Python callable cannot run inside a kernel, so the
// two recursions are cut at that call:
//
//   fk_kf_update_f64 + FK_KF_FLAG_S_ONLY      y = z - Hx, S = H P H' + R stored; the state is left alone       (:533-540)
//   fk_kf_update_f64 + FK_KF_FLAG_SI_GIVEN    `SI` is an input: K = P H' SI, x += K y, Joseph form of P          (:545-556)
//   fk_kf_rts_f64 + FK_KF_FLAG_PP_ONLY        Pp[k] = F P[k] F' + Q for every k (filtered P only: no recursion)  (:1067)
//   fk_kf_rts_f64 + FK_KF_FLAG_PPINV_GIVEN    `K` holds inv(Pp[k]) on entry and the gain on exit; the sweep      (:1069-1072)
//
// with the host applying the callable in between (filterpy_amd/kalman/kalman_filter.py).  This is the reference's escape
// hatch for singular / indefinite S, not a hot path: ONE padded instantiation (16, 8), rolled loops (arrays in scratch),
// one track per lane, every dim_x <= 16 / dim_z <= 8, both layouts and every model mode.  No factorisation happens here,
// so FK_STATUS_NOT_PD is never set; FK_STATUS_NONFINITE is.
#define FK_ROLLED 1
#include "fk_device.hpp"
#include "fk_kernel_args.hpp"

namespace fk {

constexpr int GX = 16, GZ = 8;

template <int LAYOUT, bool UNIFORM>
__global__ void __launch_bounds__(BLOCK)
kf_given_kernel(const KfArgs a, const int mode)      // mode 1: S only, 2: SI given
{
    using SharedModel = LdsModel<GX, GZ>;
    using TrackModel = RegModel<GX, GZ>;
    __shared__ double s_model[UNIFORM ? SharedModel::SIZE : 1];
    const long N = a.N;
    const long blk0 = a.i0 + (long)blockIdx.x * BLOCK;
    const Lane ln{blk0, threadIdx.x, N};
    const bool live = blk0 + ln.tid < a.i0 + a.cnt;
    const Lane lr{blk0, live ? ln.tid : 0u, N};
    const int n = a.n, m = a.m;

    double x[GX], P[GX * GX], z[GZ];
    load_rec<GX, 1, LAYOUT, false>(x, a.x, lr, n, 1, 0.0);
    load_rec<GX, GX, LAYOUT, false>(P, a.P, lr, n, n, 1.0);
    load_rec<GZ, 1, LAYOUT, false>(z, a.z, lr, m, 1, 0.0);
    const bool has_z = a.mask ? a.mask[lr.blk0 + lr.tid] != 0 : true;
    TrackModel tm;
    const SharedModel sm{s_model};
    if (UNIFORM) {
        lds_fill<GX, GX>(s_model + SharedModel::OFF_F, nullptr, n, n, 1.0, threadIdx.x);
        lds_fill<GX, GX>(s_model + SharedModel::OFF_Q, nullptr, n, n, 0.0, threadIdx.x);
        lds_fill<GZ, GX>(s_model + SharedModel::OFF_H, a.H, m, n, 0.0, threadIdx.x);
        lds_fill<GZ, GZ>(s_model + SharedModel::OFF_R, a.R, m, m, 1.0, threadIdx.x);
        __syncthreads();
    } else {
        load_rec<GZ, GX, LAYOUT, false>(tm.H, a.H, lr, m, n, 0.0);
        load_rec<GZ, GZ, LAYOUT, false>(tm.R, a.R, lr, m, m, 1.0);
    }
    double K[GX * GZ], y[GZ], S[GZ * GZ];
    if (mode == 1) {
        double PHT[GX * GZ];
        if (UNIFORM) kf_innovation<GX, GZ>(x, P, z, sm, PHT, y, S);
        else kf_innovation<GX, GZ>(x, P, z, tm, PHT, y, S);
        if (live && has_z) {
            if (a.y_out) store_rec<GZ, 1, LAYOUT, false>(y, a.y_out, ln, m, 1);
            if (a.S_out) store_rec<GZ, GZ, LAYOUT, false>(S, a.S_out, ln, m, m);
        }
        return;
    }
    double SI[GZ * GZ];
    load_rec<GZ, GZ, LAYOUT, false>(SI, a.SI_out, lr, m, m, 1.0);
    if (!has_z) return;                          // update(None): nothing changes (kalman_filter.py:515-520)
    if (UNIFORM) kf_update_given_si<GX, GZ>(x, P, z, sm, SI, K, y, S, a.rj_diag != 0);
    else kf_update_given_si<GX, GZ>(x, P, z, tm, SI, K, y, S, a.rj_diag != 0);
    if (live) {
        store_rec<GX, 1, LAYOUT, false>(x, a.x, ln, n, 1);
        store_rec<GX, GX, LAYOUT, false>(P, a.P, ln, n, n);
        if (a.y_out) store_rec<GZ, 1, LAYOUT, false>(y, a.y_out, ln, m, 1);
        if (a.K_out) store_rec<GX, GZ, LAYOUT, false>(K, a.K_out, ln, n, m);
        if (a.S_out) store_rec<GZ, GZ, LAYOUT, false>(S, a.S_out, ln, m, m);
        if (a.status) a.status[ln.blk0 + ln.tid] = (all_finite<GX>(x) && all_finite<GX * GX>(P)) ? 0 : ST_NONFINITE;
    }
}

template <int LAYOUT, bool UNIFORM>
__global__ void __launch_bounds__(BLOCK)
rts_given_kernel(const RtsArgs a, const int mode)    // mode 1: Pp only, 2: inverses given in K
{
    using SharedModel = LdsModel<GX, 1>;
    using TrackModel = RegModel<GX, 1>;
    __shared__ double s_model[UNIFORM ? SharedModel::SIZE : 1];
    const long N = a.N, T = a.T;
    const long cnt = a.cnt ? a.cnt : N;
    const long blk0 = (long)blockIdx.x * BLOCK;
    const Lane ln{blk0, threadIdx.x, N};
    const bool live = blk0 + ln.tid < cnt;
    const Lane lr{blk0, live ? ln.tid : 0u, N};
    const int n = a.n;
    const long xs_blk = N * n, ps_blk = N * (long)n * n;

    double xn[GX], Pn[GX * GX];
    load_rec<GX, 1, LAYOUT, false>(xn, a.Xs + (T - 1) * xs_blk, lr, n, 1, 0.0);
    load_rec<GX, GX, LAYOUT, false>(Pn, a.Ps + (T - 1) * ps_blk, lr, n, n, 1.0);
    if (live) {
        // k = T-1: smoothed == filtered; K = 0; Pp = Ps   (kalman_filter.py:1063-1065)
        if (a.Pp) store_rec<GX, GX, LAYOUT, false>(Pn, a.Pp + (T - 1) * ps_blk, ln, n, n);
        if (mode == 2) {
            store_rec<GX, 1, LAYOUT, false>(xn, a.xs + (T - 1) * xs_blk, ln, n, 1);
            store_rec<GX, GX, LAYOUT, false>(Pn, a.Ps_out + (T - 1) * ps_blk, ln, n, n);
            double Z[GX * GX];
            FK_UNROLL for (int i = 0; i < GX * GX; ++i) Z[i] = 0.0;
            store_rec<GX, GX, LAYOUT, false>(Z, a.K + (T - 1) * ps_blk, ln, n, n);
        }
    }
    TrackModel tm;
    const SharedModel sm{s_model};
    bool first = true;
    for (long k = T - 2; k >= 0; --k) {
        if (first || a.model_t) {
            const long mt = a.model_t ? k + a.conv_off : 0;
            if (UNIFORM) {
                if (!first) __syncthreads();
                lds_fill<GX, GX>(s_model + SharedModel::OFF_F, a.F + mt * n * n, n, n, 1.0, threadIdx.x);
                lds_fill<GX, GX>(s_model + SharedModel::OFF_Q, a.Q + mt * n * n, n, n, 0.0, threadIdx.x);
                __syncthreads();
            } else {
                load_rec<GX, GX, LAYOUT, false>(tm.F, a.F + mt * ps_blk, lr, n, n, 1.0);
                load_rec<GX, GX, LAYOUT, false>(tm.Q, a.Q + mt * ps_blk, lr, n, n, 0.0);
            }
            first = false;
        }
        double x[GX], P[GX * GX], Pp[GX * GX];
        load_rec<GX, GX, LAYOUT, false>(P, a.Ps + k * ps_blk, lr, n, n, 1.0);
        if (mode == 1) {
            if (UNIFORM) rts_pp_only<GX>(P, sm, Pp);
            else rts_pp_only<GX>(P, tm, Pp);
            if (live) store_rec<GX, GX, LAYOUT, false>(Pp, a.Pp + k * ps_blk, ln, n, n);
            continue;
        }
        double K[GX * GX];
        load_rec<GX, 1, LAYOUT, false>(x, a.Xs + k * xs_blk, lr, n, 1, 0.0);
        load_rec<GX, GX, LAYOUT, false>(K, a.K + k * ps_blk, lr, n, n, 1.0);
        if (UNIFORM) rts_step_given<GX>(x, P, xn, Pn, sm, K, Pp);
        else rts_step_given<GX>(x, P, xn, Pn, tm, K, Pp);
        if (live) {
            store_rec<GX, 1, LAYOUT, false>(x, a.xs + k * xs_blk, ln, n, 1);
            store_rec<GX, GX, LAYOUT, false>(P, a.Ps_out + k * ps_blk, ln, n, n);
            store_rec<GX, GX, LAYOUT, false>(K, a.K + k * ps_blk, ln, n, n);
            if (a.Pp) store_rec<GX, GX, LAYOUT, false>(Pp, a.Pp + k * ps_blk, ln, n, n);
        }
        FK_UNROLL for (int i = 0; i < GX; ++i) xn[i] = x[i];
        FK_UNROLL for (int i = 0; i < GX * GX; ++i) Pn[i] = P[i];
    }
    if (live && a.status && mode == 2)
        a.status[ln.blk0 + ln.tid] = (all_finite<GX>(xn) && all_finite<GX * GX>(Pn)) ? 0 : ST_NONFINITE;
}

int launch_kf_given(const KfArgs &a, int layout, bool uniform, int mode, hipStream_t stream)
{
    const dim3 grid((unsigned)((a.cnt + BLOCK - 1) / BLOCK)), block(BLOCK);
    if (layout == LAYOUT_SOA) {
        if (uniform) hipLaunchKernelGGL((kf_given_kernel<LAYOUT_SOA, true>), grid, block, 0, stream, a, mode);
        else hipLaunchKernelGGL((kf_given_kernel<LAYOUT_SOA, false>), grid, block, 0, stream, a, mode);
    } else {
        if (uniform) hipLaunchKernelGGL((kf_given_kernel<LAYOUT_AOS, true>), grid, block, 0, stream, a, mode);
        else hipLaunchKernelGGL((kf_given_kernel<LAYOUT_AOS, false>), grid, block, 0, stream, a, mode);
    }
    return check_launch("kf_given_kernel");
}

int launch_rts_given(const RtsArgs &a, int layout, bool uniform, int mode, hipStream_t stream)
{
    const dim3 grid((unsigned)(((a.cnt ? a.cnt : a.N) + BLOCK - 1) / BLOCK)), block(BLOCK);
    if (layout == LAYOUT_SOA) {
        if (uniform) hipLaunchKernelGGL((rts_given_kernel<LAYOUT_SOA, true>), grid, block, 0, stream, a, mode);
        else hipLaunchKernelGGL((rts_given_kernel<LAYOUT_SOA, false>), grid, block, 0, stream, a, mode);
    } else {
        if (uniform) hipLaunchKernelGGL((rts_given_kernel<LAYOUT_AOS, true>), grid, block, 0, stream, a, mode);
        else hipLaunchKernelGGL((rts_given_kernel<LAYOUT_AOS, false>), grid, block, 0, stream, a, mode);
    }
    return check_launch("rts_given_kernel");
}

}  // namespace fk
