// kf_kernels.hip -- fused linear-Kalman-filter kernels for gfx950 (MI355X).
//
// One track per lane, the whole time loop inside the kernel: x and P never
// leave the VGPRs between steps; per step the kernel reads z (and the model if
// it varies) and streams out the prior/posterior records.  The kernel is
// HBM-bound by its output stream (336 B per track-step at dim_x=4, dim_z=2).
//
// Replaces the per-epoch Python loop of KalmanFilter.batch_filter
// (filterpy/kalman/kalman_filter.py:980-991) and, with T=1, predict()/update()
// (:437-482 / :485-561).  Compiled once per -DFK_NX/-DFK_NZ/-DFK_EXACT (see Makefile,
// fk_dims.def); each object exports one launcher used by kf_dispatch.cpp.
#include "fk_device.hpp"
#include "fk_kernel_args.hpp"

#ifndef FK_KF_WAVES4
#define FK_KF_WAVES4 4
#endif
#ifndef FK_NX
#error "compile with -DFK_NX=<dim_x> -DFK_NZ=<dim_z> -DFK_EXACT=<0|1>"
#endif

namespace fk {

// Occupancy target (waves per SIMD): the live state is ~3 n^2 doubles per lane.
constexpr int kf_min_waves(int nx) { return nx <= 2 ? 8 : nx <= 4 ? FK_KF_WAVES4 : nx <= 6 ? 2 : 1; }

template <int NX, int NZ, bool EXACT, int LAYOUT, bool UNIFORM>
__global__ void __launch_bounds__(BLOCK, kf_min_waves(NX))
kf_kernel(const KfArgs a,
          // read-only inputs as separate __restrict__ kernel arguments (noalias)
          const double *__restrict__ pF, const double *__restrict__ pQ,
          const double *__restrict__ pH, const double *__restrict__ pR,
          const double *__restrict__ pB, const double *__restrict__ pu,
          const double *__restrict__ pz, const uint8_t *__restrict__ pmask)
{
    using SharedModel = LdsModel<NX, NZ>;
    using TrackModel = RegModel<NX, NZ>;
    __shared__ double s_model[UNIFORM ? SharedModel::SIZE : 1];

    const long N = a.N;
    const long blk0 = a.i0 + (long)blockIdx.x * BLOCK;
    const Lane ln{blk0, threadIdx.x, N};
    // Lanes past N stay in the kernel (the shared-model refill uses workgroup barriers)
    // but never touch memory: they are redirected to the workgroup's first track for loads
    // and predicated off for stores.
    const bool live = blk0 + ln.tid < a.i0 + a.cnt;
    const Lane lr{blk0, live ? ln.tid : 0u, N};
    const int n = EXACT ? NX : a.n;
    const int m = EXACT ? NZ : a.m;

    double x[NX], P[NX * NX];
    load_rec<NX, 1, LAYOUT, EXACT>(x, a.x, lr, n, 1, 0.0);
    load_rec<NX, NX, LAYOUT, EXACT>(P, a.P, lr, n, n, 1.0);

    TrackModel tm;
    const SharedModel sm{s_model};
    int st = 0;
    // carried K, S, SI, log det S for the per-step histories (zeros before the first update, like the
    // attributes of a fresh KalmanFilter)
    double cK[NX * NZ], cS[NZ * NZ], cSI[NZ * NZ], c_logdet = 0.0;
    FK_UNROLL for (int i = 0; i < NX * NZ; ++i) cK[i] = 0.0;
    FK_UNROLL for (int i = 0; i < NZ * NZ; ++i) { cS[i] = 0.0; cSI[i] = 0.0; }

    for (long t = 0; t < a.T; ++t) {
        if (t == 0 || a.model_t) {
            const long mt = a.model_t ? t : 0;
            if (UNIFORM) {
                if (t != 0) __syncthreads();   // everyone done reading the previous step's model
                lds_fill<NX, NX>(s_model + SharedModel::OFF_F, a.do_predict ? pF + mt * n * n : nullptr, n, n, 1.0, threadIdx.x);
                lds_fill<NX, NX>(s_model + SharedModel::OFF_Q, a.do_predict ? pQ + mt * n * n : nullptr, n, n, 0.0, threadIdx.x);
                lds_fill<NZ, NX>(s_model + SharedModel::OFF_H, a.do_update ? pH + mt * m * n : nullptr, m, n, 0.0, threadIdx.x);
                lds_fill<NZ, NZ>(s_model + SharedModel::OFF_R, a.do_update ? pR + mt * m * m : nullptr, m, m, 1.0, threadIdx.x);
                __syncthreads();
            } else {
                if (a.do_predict) {
                    load_rec<NX, NX, LAYOUT, EXACT>(tm.F, pF + mt * N * n * n, lr, n, n, 1.0);
                    load_rec<NX, NX, LAYOUT, EXACT>(tm.Q, pQ + mt * N * n * n, lr, n, n, 0.0);
                }
                if (a.do_update) {
                    load_rec<NZ, NX, LAYOUT, EXACT>(tm.H, pH + mt * N * m * n, lr, m, n, 0.0);
                    load_rec<NZ, NZ, LAYOUT, EXACT>(tm.R, pR + mt * N * m * m, lr, m, m, 1.0);
                }
            }
        }

        // measurement of this step (issued early: independent of the predict arithmetic)
        double z[NZ];
        bool has_z = a.do_update != 0;
        if (has_z) {
            if (pmask) has_z = pmask[t * N + lr.blk0 + lr.tid] != 0;
            load_rec<NZ, 1, LAYOUT, EXACT>(z, pz + t * N * m, lr, m, 1, 0.0);
        }

        for (int phase = 0; phase < 2; ++phase) {
            const bool is_predict = (phase == 0) != (a.update_first != 0);
            if (is_predict) {
                if (!a.do_predict) continue;
                if (UNIFORM) kf_predict<NX>(x, P, sm, a.alpha_sq);
                else kf_predict<NX>(x, P, tm, a.alpha_sq);
                if (a.nu > 0) {
                    // x = F x + B u   (kalman_filter.py:472-473)
                    const long bt = a.model_t ? t : 0;
                    double bu[NX];
                    FK_UNROLL for (int r = 0; r < NX; ++r) bu[r] = 0.0;
                    const RecView<LAYOUT> uv(pu + t * N * a.nu, lr, a.nu);
                    const RecView<LAYOUT> bv(pB + bt * N * n * a.nu, lr, n * a.nu);
                    for (int j = 0; j < a.nu; ++j) {
                        const double uj = uv.load(j);
                        FK_UNROLL for (int r = 0; r < NX; ++r) {
                            if (EXACT || r < n) {
                                const double b = UNIFORM ? pB[bt * n * a.nu + r * a.nu + j]
                                                         : bv.load(r * a.nu + j);
                                bu[r] = (j == 0) ? b * uj : fma(b, uj, bu[r]);
                            }
                        }
                    }
                    FK_UNROLL for (int r = 0; r < NX; ++r) x[r] += bu[r];
                }
                if (live) {
                    if (a.means_p) store_rec<NX, 1, LAYOUT, EXACT>(x, a.means_p + t * N * n, ln, n, 1);
                    if (a.covs_p) store_rec<NX, NX, LAYOUT, EXACT>(P, a.covs_p + t * N * n * n, ln, n, n);
                }
            } else {
                if (!a.do_update) continue;
                // extras: per-step histories or the single record of update()
                const long et = a.extras_per_step ? t : 0;
                const bool want_extras = a.y_out || a.K_out || a.S_out || a.SI_out || a.ll_out || a.maha_out;
                if (has_z) {
                    double K[NX * NZ], y[NZ], S[NZ * NZ], Lf[NZ * NZ], dinv[NZ];
                    if (UNIFORM) st |= kf_update<NX, NZ>(x, P, z, sm, K, y, S, Lf, dinv, a.rj_diag != 0);
                    else st |= kf_update<NX, NZ>(x, P, z, tm, K, y, S, Lf, dinv, a.rj_diag != 0);
                    if (want_extras) {
                        double SI[NZ * NZ];
                        inv_from_ldlt<NZ>(Lf, dinv, SI);
                        // log det S and y' S^-1 y from the factorisation
                        double logdet = 0.0, q = 0.0;
                        if constexpr (NZ == 1) {
                            logdet = log(S[0]);
                            q = y[0] * y[0] * dinv[0];
                        } else {
                            double w[NZ];
                            FK_UNROLL for (int i = 0; i < NZ; ++i) {
                                double acc = y[i];
                                FK_UNROLL for (int k2 = 0; k2 < NZ; ++k2)
                                    if (k2 < i) acc = fma(-Lf[i * NZ + k2], w[k2], acc);
                                w[i] = acc;
                                if (EXACT || i < m) q = fma(acc * acc, dinv[i], q);
                            }
                            logdet = logdet_from_dinv<NZ>(dinv, EXACT ? NZ : m);
                        }
                        FK_UNROLL for (int i = 0; i < NX * NZ; ++i) cK[i] = K[i];
                        FK_UNROLL for (int i = 0; i < NZ * NZ; ++i) { cS[i] = S[i]; cSI[i] = SI[i]; }
                        c_logdet = logdet;
                        if (live) {
                            if (a.y_out) store_rec<NZ, 1, LAYOUT, EXACT>(y, a.y_out + et * N * m, ln, m, 1);
                            if (a.ll_out) a.ll_out[t * N + ln.blk0 + ln.tid] = -0.5 * (m * 1.8378770664093453 + logdet + q);
                            if (a.maha_out) a.maha_out[t * N + ln.blk0 + ln.tid] = sqrt(q);
                        }
                    }
                } else if (want_extras && live) {
                    // missing measurement (kalman_filter.py:515-520): y = 0, K / S / SI keep their last values
                    double y0[NZ];
                    FK_UNROLL for (int i = 0; i < NZ; ++i) y0[i] = 0.0;
                    if (a.y_out) store_rec<NZ, 1, LAYOUT, EXACT>(y0, a.y_out + et * N * m, ln, m, 1);
                    if (a.ll_out) a.ll_out[t * N + ln.blk0 + ln.tid] = -0.5 * (m * 1.8378770664093453 + c_logdet);
                    if (a.maha_out) a.maha_out[t * N + ln.blk0 + ln.tid] = 0.0;
                }
                if (want_extras && live) {
                    if (a.K_out) store_rec<NX, NZ, LAYOUT, EXACT>(cK, a.K_out + et * N * n * m, ln, n, m);
                    if (a.S_out) store_rec<NZ, NZ, LAYOUT, EXACT>(cS, a.S_out + et * N * m * m, ln, m, m);
                    if (a.SI_out) store_rec<NZ, NZ, LAYOUT, EXACT>(cSI, a.SI_out + et * N * m * m, ln, m, m);
                }
                if (live) {
                    if (a.means) store_rec<NX, 1, LAYOUT, EXACT>(x, a.means + t * N * n, ln, n, 1);
                    if (a.covs) store_rec<NX, NX, LAYOUT, EXACT>(P, a.covs + t * N * n * n, ln, n, n);
                }
            }
        }
    }

    if (live) {
        store_rec<NX, 1, LAYOUT, EXACT>(x, a.x, ln, n, 1);
        store_rec<NX, NX, LAYOUT, EXACT>(P, a.P, ln, n, n);
        if (a.status) {
            if (!all_finite<NX>(x) || !all_finite<NX * NX>(P)) st |= ST_NONFINITE;
            a.status[ln.blk0 + ln.tid] = st;
        }
    }
}

template <int NX, int NZ, bool EXACT>
static int launch(const KfArgs &a, int layout, bool uniform, hipStream_t stream)
{
    const dim3 grid((unsigned)((a.cnt + BLOCK - 1) / BLOCK)), block(BLOCK);
#define FK_GO(LAY, UNI)                                                                             \
    hipLaunchKernelGGL((kf_kernel<NX, NZ, EXACT, LAY, UNI>), grid, block, 0, stream, a, a.F, a.Q, \
                       a.H, a.R, a.B, a.u, a.z, a.mask)
    if (layout == LAYOUT_SOA) {
        if (uniform) FK_GO(LAYOUT_SOA, true);
        else FK_GO(LAYOUT_SOA, false);
    } else {
        if (uniform) FK_GO(LAYOUT_AOS, true);
        else FK_GO(LAYOUT_AOS, false);
    }
#undef FK_GO
    return check_launch("kf_kernel");
}

#define FK_CAT_(a, b, c, d) a##b##_##c##_##d
#define FK_CAT(a, b, c, d) FK_CAT_(a, b, c, d)

int FK_CAT(launch_kf_, FK_NX, FK_NZ, FK_EXACT)(const KfArgs &a, int layout, bool uniform, hipStream_t stream)
{
    return launch<FK_NX, FK_NZ, (FK_EXACT != 0)>(a, layout, uniform, stream);
}

}  // namespace fk
