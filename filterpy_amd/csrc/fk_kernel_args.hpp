// fk_kernel_args.hpp -- by-value argument blocks of the gfx950 kernels (kernarg segment).
#pragma once
#include <stdint.h>

namespace fk {

struct KfArgs {
    const double *F, *Q, *H, *R, *B, *u, *z;
    const uint8_t *mask;
    double *x, *P, *means, *covs, *means_p, *covs_p;
    double *y_out, *K_out, *S_out, *SI_out;   // update() extras: one record (T == 1) or one per step
    double *ll_out, *maha_out;                // per-step log-likelihood / mahalanobis [T][N]
    int extras_per_step;                      // 1: the extras are [T][N][..] histories (batch_filter_ex)
    int32_t *status;
    long N, T;
    long i0, cnt;       // this launch handles tracks [i0, i0+cnt); N stays the array stride
    int n, m, nu;
    int model_t;        // 1: model records advance with the time step
    int update_first;
    int do_predict, do_update;
    int xcd_swizzle;    // kf_fast: give each XCD (blockIdx % 8) one contiguous range of tracks
    int rj_diag;        // FK_KF_FLAG_R_JOSEPH_DIAG: K R K' uses only R's diagonal (generic kernel only)
    int status_or;      // kf_ml: OR the status bits into status[] instead of storing them (later time chunks of one call)
    double alpha_sq;
    // FK_KF_FLAG_COV_INTERLEAVED (kf_fast only): covs / covs_p are the two halves of ONE array, so that a step writes one
    // contiguous region.  cov_step: doubles between the slabs of consecutive steps (N n^2, or 2 N n^2 interleaved);
    // cov_pitch: doubles between the records of consecutive tracks in NumPy order (n^2, or 2 n^2 interleaved).
    long cov_step;
    int cov_pitch;
    // kf_ml PERS instantiations (persistent grid: one ticket per (time chunk, workgroup of tracks), kf_ml.hip): ctl[0] the
    // ticket counter, ctl[1 + g] how many time chunks of track group g are complete; G groups x H chunks
    int *pers_ctl;
    int pers_G, pers_H;
    double *pers_ws;    // [n + n*n][N] element-major: the state between the chunks of a group (coalesced in both layouts)
};

struct RtsArgs {
    const double *F, *Q, *Xs, *Ps;
    double *xs, *Ps_out, *K, *Pp;
    int32_t *status;
    long N, T;
    int n;
    int model_t;
    int conv_off;       // 1: class method uses model[k+1]; 0: module function uses model[k]
    // rts_ml (chunked calls, kf_ml.hip): this launch handles tracks [i0, i0 + cnt) (cnt == 0: all N); cont: the last step
    // of this launch's window is ALREADY smoothed in xs / Ps_out (written by the chunk after it in time) -- read it
    // from there and leave it alone; status_or: OR the status bits into status[]
    long i0, cnt;
    int cont, status_or;
    // the persistent grid of rts_ml_kernel (PERS instantiations; same scheme as KfArgs'): ticket counter + one completion word per
    // track group, the smoothed state between the time chunks of a group [n + n*n][N] element-major
    int *pers_ctl;
    int pers_G, pers_H;
    double *pers_ws;
};

struct UkfArgs {
    const double *F, *H, *Q, *R, *Wm, *Wc, *z;
    const uint8_t *mask;
    double *x, *P, *means, *covs;
    int32_t *status;
    long N, T;
    int n, m;
    double scale;
    long i0, cnt;        // the launch covers tracks [i0, i0 + cnt) of the N (a piece of a chunked call, fk_chunks.hpp)
    int status_or;       // 1: OR the status into what an earlier time chunk left
    int soa_pairs;       // element-major outputs as 16-byte stores of two element rows (wave_store_soa_pairs) where a wave allows it
};

// the fused linear UKF smoothers (ukf_kernels.hip, ukf_mlg.hip)
struct UkfRtsArgs {
    const double *Xs, *Ps;
    double *xs, *ps, *Ks;
    int32_t *status;
    long N, T;
    int n;
    double scale;
    long i0, cnt;        // the launch covers tracks [i0, i0 + cnt) of the N (a piece of a chunked call, fk_chunks.hpp)
    int cont;            // 1: the window's top step was smoothed by the piece before it -- read it from xs / ps, do not copy
    int status_or;       // 1: OR the status into what an earlier piece left
};

struct ImmArgs {
    const double *F, *Q, *H, *R, *Mt, *z;
    double *xs, *Ps, *mu;
    double *x_out, *P_out, *mu_out, *xp_out, *Pp_out, *L_out;
    const double *B, *u;     // control input (general kernel only): B [n_models][n*nu], u [T][N][nu]
    int nu;
    const uint8_t *mask;     // [T][N], 0 = update(None); NULL = every measurement present (general kernel only)
    double *ll0;             // [N][n_models] in/out: log-density of a zero residual under each filter's last S (or NULL)
    int32_t *status;
    long N, T;
    int n, m;
    int phase;
    int mmae;
    long i0, cnt;            // the launch covers banks [i0, i0 + cnt) of the N (a piece of a chunked call, fk_chunks.hpp)
    int status_or;           // 1: OR the status into what an earlier time chunk left
};

}  // namespace fk
