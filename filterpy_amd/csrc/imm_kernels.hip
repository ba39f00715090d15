// imm_kernels.hip -- batched Interacting Multiple Model estimator for gfx950.
//
// the reference's only "many filters, same z" caller is
// filterpy/kalman/IMM.py (IMMEstimator: predict :188-222, update :160-186,
// _compute_state_estimate :224-237, _compute_mixing_probabilities :239-249), which loops over
// its filter bank in Python.  Here one lane owns one track's whole bank (NM linear filters of
// the same dim_x/dim_z) and runs T x { predict(); update(z) } without leaving registers:
//
//   cbar = mu . M ;  omega[i][j] = M[i][j] mu[i] / cbar[j]
//   x0_j = sum_i omega[i][j] x_i ;  P0_j = sum_i omega[i][j] (outer(x_i - x0_j) + P_i)     (mixing)
//   (x_j, P_j) = KF_j.predict(x0_j, P0_j) ; prior estimate = mu-weighted moments
//   (x_j, P_j) = KF_j.update(z) ;  L_j = max(exp(logpdf(y_j; 0, S_j)), DBL_MIN)
//   mu_j = cbar_j L_j / sum ;  posterior estimate = mu-weighted moments
//
// The NM models (F,Q,H,R each) are shared by all tracks and sit in LDS (NM x LdsModel);
// the transition matrix M (NM x NM) is read from LDS too.  Records: the bank state is
// xs [N][NM*n], Ps [N][NM*n*n], mu [N][NM] (AOS) or element-major (SOA); per-step outputs
// x [T][N][n], P [T][N][n*n], mu [T][N][NM], optional prior x/P and likelihoods [T][N][NM].
#include "fk_device.hpp"
#include "fk_imm.hpp"
#include "fk_kernel_args.hpp"
#include "../../include/filterhip.h"

namespace fk {

struct ImmArgs {
    const double *F, *Q, *H, *R, *Mt, *z;
    double *xs, *Ps, *mu;
    double *x_out, *P_out, *mu_out, *xp_out, *Pp_out, *L_out;
    int32_t *status;
    long N, T;
    int n, m;
    int phase;
};

template <int NX, int NZ, int NM, int LAYOUT>
__global__ void __launch_bounds__(BLOCK, 1)
imm_kernel(const ImmArgs a)
{
    using LM = LdsModel<NX, NZ>;
    __shared__ double smem[NM * LM::SIZE + NM * NM];
    const int n = a.n, m = a.m;
    const long N = a.N;
    FK_UNROLL for (int j = 0; j < NM; ++j) {
        double *s = smem + j * LM::SIZE;
        lds_fill<NX, NX>(s + LM::OFF_F, a.F + (long)j * n * n, n, n, 1.0, threadIdx.x);
        lds_fill<NX, NX>(s + LM::OFF_Q, a.Q + (long)j * n * n, n, n, 0.0, threadIdx.x);
        lds_fill<NZ, NX>(s + LM::OFF_H, a.H + (long)j * m * n, m, n, 0.0, threadIdx.x);
        lds_fill<NZ, NZ>(s + LM::OFF_R, a.R + (long)j * m * m, m, m, 1.0, threadIdx.x);
    }
    if (threadIdx.x < NM * NM) smem[NM * LM::SIZE + threadIdx.x] = a.Mt[threadIdx.x];
    __syncthreads();
    const double *sM = smem + NM * LM::SIZE;

    Lane ln{(long)blockIdx.x * BLOCK, threadIdx.x, N};
    const bool live = ln.blk0 + ln.tid < N;
    if (!live) return;

    constexpr int PL = NX * (NX + 1) / 2;
    double xs[NM][NX], Ps[NM][PL], mu[NM];   // covariances packed (upper triangle): fk_math_sym.hpp
    {
        const RecView<LAYOUT> vx(a.xs, ln, NM * n), vP(a.Ps, ln, NM * n * n), vm(a.mu, ln, NM);
        FK_UNROLL for (int j = 0; j < NM; ++j) {
            mu[j] = vm.load(j);
            FK_UNROLL for (int r = 0; r < NX; ++r) {
                xs[j][r] = (r < n) ? vx.load(j * n + r) : 0.0;
                FK_UNROLL for (int c = r; c < NX; ++c)
                    Ps[j][sym_idx<NX>(r, c)] = (r < n && c < n) ? vP.load((j * n + r) * n + c) : ((r == c) ? 1.0 : 0.0);
            }
        }
    }
    int st = 0;
    LM mods[NM];
    FK_UNROLL for (int j = 0; j < NM; ++j) mods[j].s = smem + j * LM::SIZE;

    for (long t = 0; t < a.T; ++t) {
        double z[NZ];
        {
            const RecView<LAYOUT> vz(a.z + t * N * m, ln, m);
            FK_UNROLL for (int r = 0; r < NZ; ++r) z[r] = (r < m && a.phase != FK_IMM_PREDICT) ? vz.load(r) : 0.0;
        }
        double cbar[NM];
        imm_mixing_cbar<NM>(mu, sM, cbar);
        if (a.phase != FK_IMM_UPDATE) {
        imm_predict<NX, NM>(xs, Ps, mu, cbar, sM, mods);
        if (a.xp_out || a.Pp_out) {
            double x[NX], P[NX * NX];
            imm_estimate<NX, NM>(xs, Ps, mu, x, P);
            if (a.xp_out) store_rec<NX, 1, LAYOUT, false>(x, a.xp_out + t * N * n, ln, n, 1);
            if (a.Pp_out) store_rec<NX, NX, LAYOUT, false>(P, a.Pp_out + t * N * n * n, ln, n, n);
        }
        }
        if (a.phase == FK_IMM_PREDICT) break;
        double L[NM];
        st |= imm_update<NX, NZ, NM>(xs, Ps, mu, cbar, z, m, mods, L);
        {
            double x[NX], P[NX * NX];
            imm_estimate<NX, NM>(xs, Ps, mu, x, P);
            if (a.x_out) store_rec<NX, 1, LAYOUT, false>(x, a.x_out + t * N * n, ln, n, 1);
            if (a.P_out) store_rec<NX, NX, LAYOUT, false>(P, a.P_out + t * N * n * n, ln, n, n);
        }
        if (a.mu_out) {
            const RecView<LAYOUT> v(a.mu_out + t * N * NM, ln, NM);
            FK_UNROLL for (int j = 0; j < NM; ++j) v.store(j, mu[j]);
        }
        if (a.L_out) {
            const RecView<LAYOUT> v(a.L_out + t * N * NM, ln, NM);
            FK_UNROLL for (int j = 0; j < NM; ++j) v.store(j, L[j]);
        }
    }
    {
        const RecView<LAYOUT> vx(a.xs, ln, NM * n), vP(a.Ps, ln, NM * n * n), vm(a.mu, ln, NM);
        bool fin = true;
        FK_UNROLL for (int j = 0; j < NM; ++j) {
            vm.store(j, mu[j]);
            fin = fin && all_finite<NX>(xs[j]) && all_finite<PL>(Ps[j]) && (fabs(mu[j]) <= 1.79769313486231570815e+308);
            FK_UNROLL for (int r = 0; r < NX; ++r) {
                if (r < n) vx.store(j * n + r, xs[j][r]);
                FK_UNROLL for (int c = 0; c < NX; ++c)
                    if (r < n && c < n) vP.store((j * n + r) * n + c, Ps[j][sym_idx<NX>(r, c)]);
            }
        }
        if (a.status) a.status[ln.blk0 + ln.tid] = st | (fin ? 0 : ST_NONFINITE);
    }
}

static int fail(int code, const char *msg)
{
    set_last_error(msg);
    return code;
}

template <int NX, int NZ, int NM>
static void launch(const ImmArgs &a, int layout, hipStream_t s)
{
    const dim3 grid((unsigned)((a.N + BLOCK - 1) / BLOCK)), block(BLOCK);
    if (layout == FK_LAYOUT_SOA) hipLaunchKernelGGL((imm_kernel<NX, NZ, NM, LAYOUT_SOA>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((imm_kernel<NX, NZ, NM, LAYOUT_AOS>), grid, block, 0, s, a);
}

}  // namespace fk

using namespace fk;

extern "C" int fk_imm_batch_f64(const fk_imm_desc *d, const double *F, const double *Q, const double *H,
                                const double *R, const double *M, const double *z, double *xs, double *Ps,
                                double *mu, double *x_out, double *P_out, double *mu_out, double *x_prior_out,
                                double *P_prior_out, double *likelihood_out, int32_t *status, void *stream)
{
    if (!d) return fail(FK_ERR_BAD_ARG, "desc is NULL");
    if (d->n < 1 || d->n > 6 || d->m < 1 || d->m > 3 || d->n_models < 2 || d->n_models > 3)
        return fail(FK_ERR_UNSUPPORTED, "IMM: dim_x 1..6, dim_z 1..3, 2..3 models");
    if (d->layout != FK_LAYOUT_AOS && d->layout != FK_LAYOUT_SOA) return fail(FK_ERR_BAD_ARG, "IMM: bad layout");
    if (d->phase < FK_IMM_STEP || d->phase > FK_IMM_UPDATE) return fail(FK_ERR_BAD_ARG, "IMM: bad phase");
    const bool needs_z = (d->phase == FK_IMM_STEP && d->T > 0) || d->phase == FK_IMM_UPDATE;
    if (d->N < 0 || d->T < 0 || !F || !Q || !H || !R || !M || !xs || !Ps || !mu || (needs_z && !z))
        return fail(FK_ERR_BAD_ARG, "IMM: bad argument");
    if ((double)d->N * d->n_models * d->n * d->n * 8.0 >= 4294967296.0)
        return fail(FK_ERR_UNSUPPORTED, "IMM: record block >= 4 GiB, split the batch");
    if (d->N == 0) return FK_OK;
    ImmArgs a{};
    a.F = F; a.Q = Q; a.H = H; a.R = R; a.Mt = M; a.z = z; a.xs = xs; a.Ps = Ps; a.mu = mu;
    a.x_out = x_out; a.P_out = P_out; a.mu_out = mu_out; a.xp_out = x_prior_out; a.Pp_out = P_prior_out;
    a.L_out = likelihood_out; a.status = status; a.N = d->N; a.T = d->phase == FK_IMM_STEP ? d->T : 1;
    a.n = d->n; a.m = d->m; a.phase = d->phase;
    hipStream_t s = (hipStream_t)stream;
    const int cls = (d->n <= 2 && d->m <= 1) ? 0 : (d->n <= 4 && d->m <= 2) ? 1 : 2;
    if (d->n_models == 2) {
        if (cls == 0) launch<2, 1, 2>(a, d->layout, s);
        else if (cls == 1) launch<4, 2, 2>(a, d->layout, s);
        else launch<6, 3, 2>(a, d->layout, s);
    } else {
        if (cls == 0) launch<2, 1, 3>(a, d->layout, s);
        else if (cls == 1) launch<4, 2, 3>(a, d->layout, s);
        else launch<6, 3, 3>(a, d->layout, s);
    }
    return check_launch("imm_kernel");
}
