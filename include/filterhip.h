/* filterhip.h -- C ABI of libfilterhip.so, the MI355X (gfx950) batched-filter engine.
 *
 * This is the drop-in boundary for ONE hot path of rlabbe/filterpy (v1.4.5):
 * many independent small filters stepped in lock-step.  The reference has no
 * FFI of its own (it is pure Python); each entry point below names the
 * reference function (file:line under /root/reference) whose arithmetic it
 * replaces.  INTEGRATION.md shows the ctypes binding a filterpy maintainer
 * would add.
 *
 * Conventions (all entry points):
 *   - extern "C", plain pointers and sizes, no C++/torch types;
 *   - every data pointer is a DEVICE pointer owned by the caller (HBM resident);
 *     the library never allocates, frees or copies user-visible memory;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); calls
 *     only enqueue work, they do not synchronise;
 *   - return value: FK_OK (0) or a negative FK_ERR_* for argument / launch
 *     errors.  Per-track numerical trouble is reported in the caller's
 *     `status[N]` array (bit flags FK_STATUS_*), never by faulting;
 *   - all arithmetic is IEEE fp64 ("f64" suffix).
 *
 * Array layouts.  `layout` selects how per-track records are laid out:
 *   FK_LAYOUT_AOS (0): NumPy C order, track-major records:  a[t][i][e]
 *                      (e.g. covariances[T][N][n][n]) -- what np.asarray() of the
 *                      reference's return values would look like with a track axis;
 *   FK_LAYOUT_SOA (1): track-minor (coalesced) records:      a[t][e][i]
 * with t = time step, i = track, e = element index in C order (row*cols+col).
 *
 * Model matrices F,Q (n x n), H (m x n), R (m x m), B (n x nu) follow
 * `model_mode`:
 *   FK_MODEL_SHARED           (0): one matrix for every track and step   M[e]
 *   FK_MODEL_PER_TRACK        (1): one per track                          AOS M[i][e] / SOA M[e][i]
 *   FK_MODEL_PER_TRACK_STEP   (2): one per track and step (Fs/Qs/Hs/Rs lists,
 *                                  kalman_filter.py:941-952)              AOS M[t][i][e] / SOA M[t][e][i]
 *   FK_MODEL_PER_STEP         (3): one per step shared by all tracks     M[t][e]
 */
#ifndef FILTERHIP_H
#define FILTERHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FK_ABI_VERSION 4      /* 4 (round 6): the FK_KF_FLAG_*_ONLY / *_GIVEN flags (a caller-supplied inverse); fk_ukf_linear_supported */

enum {
    FK_OK = 0,
    FK_ERR_BAD_ARG = -1,      /* NULL where data is required, negative sizes, ... */
    FK_ERR_UNSUPPORTED = -2,  /* dims outside the compiled range (dim_x <= 16, dim_z <= 8) */
    FK_ERR_LAUNCH = -3,       /* HIP reported a launch / runtime error */
    FK_ERR_WORKSPACE = -4     /* workspace too small: see the *_workspace_bytes query */
};

enum { FK_LAYOUT_AOS = 0, FK_LAYOUT_SOA = 1 };

enum {
    FK_MODEL_SHARED = 0,
    FK_MODEL_PER_TRACK = 1,
    FK_MODEL_PER_TRACK_STEP = 2,
    FK_MODEL_PER_STEP = 3
};

/* bits of status[i] */
enum {
    FK_STATUS_NOT_PD = 1,      /* S (or Pp / P for RTS / sigma points) not positive definite:
                                  the reference would raise numpy.linalg.LinAlgError or return junk */
    FK_STATUS_NONFINITE = 2,   /* NaN/Inf in the track's state after the call */
    FK_STATUS_OVERRUN = 4,     /* resample: a position >= cumsum[-1] (reference: IndexError,
                                  resampling.py:109,145); the index is clamped to Np-1 */
    FK_STATUS_BAD_WEIGHTS = 16,/* fused UKF: FK_UKF_FLAG_PAIR_WEIGHTS was given but the weights of a +- pair of sigma points differ;
                                  the call's outputs are not meaningful */
    FK_STATUS_INTERNAL = 8     /* resample: an in-launch hand-off timed out (never expected; the indices of this
                                  filter are not valid) */
};

/* ------------------------------------------------------------------ */
/* Linear Kalman filter                                               */
/* ------------------------------------------------------------------ */
typedef struct fk_kf_desc {
    int32_t n;            /* dim_x  (1..16) */
    int32_t m;            /* dim_z  (1..8, <= n not required) */
    int32_t nu;           /* dim_u  (0 = no control input) */
    int32_t model_mode;   /* FK_MODEL_* for F,Q,H,R,B */
    int64_t N;            /* tracks */
    int64_t T;            /* time steps */
    int32_t layout;       /* FK_LAYOUT_* for z,u,mask-free records, x,P, outputs, per-track models */
    int32_t update_first; /* kalman_filter.py:966-978: update then predict */
    double  alpha_sq;     /* fading memory alpha^2 (kalman_filter.py:478), 1.0 = off */
    int32_t flags;        /* FK_KF_FLAG_* (0 = none) */
    int32_t reserved;     /* 0 */
} fk_kf_desc;

/* The reference with a SCALAR `R` attribute and dim_z > 1 (kalman_filter.py:540, 556: `S = dot(H, PHT) + R` adds r to
 * every element of S, `dot(dot(K, R), K.T)` is r K K'): pass R = r * ones(m, m) and set this flag -- the innovation
 * covariance uses R as given, the Joseph term K R K' only its diagonal. */
#define FK_KF_FLAG_R_JOSEPH_DIAG 1

/* fk_kf_batch_filter_f64 with all four outputs: `covs` and `covs_p` are the two halves of ONE array,
 *     FK_LAYOUT_AOS:  cov2[T][N][2][n*n]   covs = cov2, covs_p = cov2 + n*n        (record pitch 2 n*n)
 *     FK_LAYOUT_SOA:  cov2[T][2][n*n][N]   covs = cov2, covs_p = cov2 + n*n*N      (step stride 2 n*n*N)
 * so that the posterior and the prior covariance of a step leave as ONE contiguous write front.  Why: on MI355X two large
 * arrays written side by side at full rate run 2-25 % slower when the driver happened to back both with the same class of
 * physical memory, which user space cannot control (docs/PLACEMENT.md); one front has no partner to interfere with.  The
 * two histories are then strided views of cov2 (what KalmanFilterBank.batch_filter(device_outputs=True) returns).  The
 * pointers must stand in exactly that relation (FK_ERR_BAD_ARG otherwise); calls the specialised kernel does not serve
 * (dim_x >= 9, per-step extras, final-state-only) return FK_ERR_UNSUPPORTED -- use two arrays there. */
#define FK_KF_FLAG_COV_INTERLEAVED 2

/* A caller-supplied inverse.  The reference applies whatever `KalmanFilter.inv` names to S (kalman_filter.py:363, 434, 541;
 * documented use: numpy.linalg.pinv for a singular S) and rts_smoother takes `inv=` (:995, 1069).  A callable of the host
 * language cannot run inside a kernel, so both recursions can be cut at that call -- the caller runs two launches with its own
 * inverse in between (filterpy_amd/kalman/kalman_filter.py does, whenever `inv` is not numpy.linalg.inv):
 *   fk_kf_update_f64, FK_KF_FLAG_S_ONLY      y = z - Hx and S = H P H' + R are stored (y, S required); x, P, K, SI untouched;
 *   fk_kf_update_f64, FK_KF_FLAG_SI_GIVEN    `SI` is an INPUT record array [N][m][m] (the caller's inv(S); need not be
 *                                            symmetric): K = P H' SI, x += K y, P = (I-KH) P (I-KH)' + K R K'; y, K, S stored;
 *   fk_kf_rts_f64, FK_KF_FLAG_PP_ONLY        Pp[k] = F P[k] F' + Q for k < T-1, Pp[T-1] = Ps[T-1] (Pp required); nothing else
 *                                            is written (Pp depends on the FILTERED covariances only: no recursion);
 *   fk_kf_rts_f64, FK_KF_FLAG_PPINV_GIVEN    `K` (required) holds the caller's inv(Pp[k]) in K[k] on entry and the gain on
 *                                            exit; xs, Ps_out, Pp as in the plain call.
 * No factorisation runs in these calls: FK_STATUS_NOT_PD is never set.  One padded kernel serves every size (a single
 * filter's escape hatch, not a throughput path); banks whose step block reaches 4 GiB are refused (FK_ERR_UNSUPPORTED). */
#define FK_KF_FLAG_S_ONLY 4
#define FK_KF_FLAG_SI_GIVEN 8
#define FK_KF_FLAG_PP_ONLY 16
#define FK_KF_FLAG_PPINV_GIVEN 32

/* KalmanFilter.batch_filter (filterpy/kalman/kalman_filter.py:826-993; module twin :1664-1788)
 * for N independent filters: T x { predict (:472-478) ; update (:533-556, Joseph form) },
 * state kept in registers across the whole time loop.
 *
 *   F,Q,H,R : models per desc->model_mode.          B,u : control (NULL when nu == 0);
 *             u is a record array u[T][N][nu] in `layout`.
 *   z       : measurements, record array [T][N][m] in `layout`.
 *   mask    : uint8 [T][N] (always t-major, track-minor), 0 = missing measurement
 *             (the reference's `None`, kalman_filter.py:515-520: update skipped); NULL = none missing.
 *   x, P    : in: initial state [N][n] / [N][n][n] (in `layout`: AOS x[i][e], SOA x[e][i]);
 *             out: final state.
 *   means, covs     : posterior per step, records [T][N][n] / [T][N][n*n]; may be NULL (not stored).
 *   means_p, covs_p : prior per step; may be NULL.
 *   status  : int32 [N] (OR-ed FK_STATUS_* bits); may be NULL.
 *
 * The call is asynchronous on `stream` and ordered like one kernel on it; at dim_x >= 7 it may fan out over up to three
 * helper streams of the library that fork from and join back into `stream` with events (same results bit for bit;
 * INTEGRATION.md, "Streams").
 *
 * S^-1 is applied by an in-lane LDL^T (square-root-free Cholesky) solve, not an
 * explicit inverse; S must be symmetric positive definite (status bit otherwise).
 */
int fk_kf_batch_filter_f64(const fk_kf_desc *desc,
                           const double *F, const double *Q, const double *H, const double *R,
                           const double *B, const double *u,
                           const double *z, const uint8_t *mask,
                           double *x, double *P,
                           double *means, double *covs, double *means_p, double *covs_p,
                           int32_t *status, void *stream);

/* Per-step histories of the update's by-products: what filterpy.common.Saver
 * (filterpy/common/helpers.py:121-152) records when it is attached to batch_filter (kalman_filter.py:990-991)
 * and the lazily computed properties log_likelihood / mahalanobis (kalman_filter.py:1203-1239).
 * Every pointer may be NULL (not stored).  Records per (step, track) in desc->layout:
 *   y [T][N][m], K [T][N][n*m], S [T][N][m*m], SI [T][N][m*m];  scalars [T][N]: log_likelihood
 *   = log N(y; 0, S), mahalanobis = sqrt(y' S^-1 y).  A missing measurement stores y = 0 and repeats
 *   the previous K, S, SI (zeros before the first update), exactly like the attributes of the
 *   reference object after update(None). */
typedef struct fk_kf_extras {
    double *y, *K, *S, *SI;
    double *log_likelihood, *mahalanobis;
} fk_kf_extras;

/* fk_kf_batch_filter_f64 plus the histories above (ex may be NULL).  Which kernel serves the call (all are tested against
 * the same oracle): the plain call (shared constant model, predict -> update, all four outputs) runs on the specialised kernels'
 * extras instantiations -- one lane per track at dim_x <= 8 (and at dim_x 9 with a mask), four lanes per track at dim_x 9..16
 * without a mask --; every other combination (per-track / per-step models, control input, update_first, a subset of the outputs)
 * on the generic kernel. */
int fk_kf_batch_filter_ex_f64(const fk_kf_desc *desc,
                              const double *F, const double *Q, const double *H, const double *R,
                              const double *B, const double *u,
                              const double *z, const uint8_t *mask,
                              double *x, double *P,
                              double *means, double *covs, double *means_p, double *covs_p,
                              const fk_kf_extras *extras, int32_t *status, void *stream);

/* KalmanFilter.predict (kalman_filter.py:437-482; module twin :1571-1621) on a resident batch:
 * one step, x/P updated in place.  desc->T is ignored (treated as 1). */
int fk_kf_predict_f64(const fk_kf_desc *desc, const double *F, const double *Q,
                      const double *B, const double *u, double *x, double *P,
                      int32_t *status, void *stream);

/* KalmanFilter.update (kalman_filter.py:485-561; module twin :1401-1508) on a resident batch.
 * Optional extra outputs (NULL = skip), records per track in `layout`:
 *   y [N][m], K [N][n*m], S [N][m*m], SI [N][m*m]  (attributes kalman_filter.py:533-544). */
int fk_kf_update_f64(const fk_kf_desc *desc, const double *H, const double *R,
                     const double *z, const uint8_t *mask, double *x, double *P,
                     double *y, double *K, double *S, double *SI,
                     int32_t *status, void *stream);

/* KalmanFilter.rts_smoother (kalman_filter.py:995-1074) / module rts_smoother (:1792-1858).
 *   Xs [T][N][n], Ps [T][N][n*n] : filter output (read only)
 *   xs, Ps_out, K, Pp            : smoothed means / covariances / gains / predicted covariances,
 *                                  same shapes ([T][N][n*n] for K, Pp); K[T-1] = 0, Pp[T-1] = Ps[T-1].
 *   index_convention 0: class method, uses F[k+1],Q[k+1] (:1067);  1: module function, F[k],Q[k] (:1851).
 *   F,Q per desc->model_mode (only PER_STEP / PER_TRACK_STEP make the convention visible).
 * Pp must be SPD (LDL^T solve); status bit otherwise. */
int fk_kf_rts_f64(const fk_kf_desc *desc, const double *F, const double *Q,
                  const double *Xs, const double *Ps,
                  double *xs, double *Ps_out, double *K, double *Pp,
                  int32_t index_convention, int32_t *status, void *stream);

/* ------------------------------------------------------------------ */
/* Unscented transform path                                           */
/* ------------------------------------------------------------------ */

/* MerweScaledSigmaPoints.sigma_points (filterpy/kalman/sigma_points.py:124-177) and
 * JulierSigmaPoints.sigma_points (:289-357): U = chol_upper(scale * P);
 * sigma_0 = x, sigma_{k+1} = x + U[k], sigma_{n+k+1} = x - U[k].
 *   scale = lambda + n (Merwe) or n + kappa (Julier), computed by the caller.
 *   x [N][n], P [N][n*n] -> sigmas [N][(2n+1)*n] (record = the (2n+1, n) C-order array). */
int fk_ut_sigma_points_f64(int32_t n, int64_t N, int32_t layout, double scale,
                           const double *x, const double *P, double *sigmas,
                           int32_t *status, void *stream);

/* unscented_transform (filterpy/kalman/unscented_transform.py:99-128), default mean/residual:
 *   x = Wm . sigmas ; y = sigmas - x ; P = y' diag(Wc) y (+ noise_cov).
 *   sigmas [N][k*n] records; Wm, Wc [k] shared (device); noise_cov [n*n] shared or NULL.
 *   -> x_out [N][n], P_out [N][n*n]. */
int fk_ut_transform_f64(int32_t n, int32_t k, int64_t N, int32_t layout,
                        const double *sigmas, const double *Wm, const double *Wc,
                        const double *noise_cov, double *x_out, double *P_out, void *stream);

/* UnscentedKalmanFilter.cross_variance (filterpy/kalman/UKF.py:493-504), default residuals:
 *   Pxz = sum_i Wc[i] (sigmas_f[i]-x)(sigmas_h[i]-z)'.
 *   sigmas_f [N][k*n], sigmas_h [N][k*m], x [N][n], z [N][m] -> Pxz [N][n*m].
 * Custom residual_x / residual_z callables (UKF.py:500-501; unscented_transform.py:120-123 is the same sum with
 * sigmas_f = sigmas_h = the residuals): the caller applies them and passes x = NULL and / or z = NULL -- that
 * operand then already holds dx (dz) and nothing is subtracted; the accumulation order is the reference loop's. */
int fk_ut_cross_variance_f64(int32_t n, int32_t m, int32_t k, int64_t N, int32_t layout,
                             const double *x, const double *z,
                             const double *sigmas_f, const double *sigmas_h,
                             const double *Wc, double *Pxz, void *stream);

/* A linear process / measurement model given as a matrix -- UnscentedKalmanFilter(fx=F, hx=H) -- applied to the sigma
 * points where the fused kernels do not reach: what the reference's lambda `F @ s` does once per point (filterpy/kalman/
 * UKF.py:521-522, :462-466), for every point of every track in one launch.
 *   M [n_out*n_in] shared (device); in [N][k*n_in] records -> out [N][k*n_out] records.  Dims 1..16. */
int fk_ut_linear_map_f64(int32_t n_in, int32_t n_out, int32_t k, int64_t N, int32_t layout, const double *M,
                         const double *in, double *out, void *stream);

/* The correction at the end of UnscentedKalmanFilter.update (filterpy/kalman/UKF.py:470-481) for
 * arbitrary measurement functions:  K = Pxz S^-1 (Cholesky solve) ; x += K (z - zp) ; P -= K (S K').
 *   Pxz [N][n*m], zp [N][m], S [N][m*m], z [N][m] ; x [N][n], P [N][n*n] updated in place ;
 *   K [N][n*m] out (may be NULL) ; status [N] or NULL.   dim_x 1..16, dim_z 1..8.
 * zp = NULL: z already holds y = residual_z(z, zp) (UKF.py:474 with a custom callable).  For a custom state_add
 * (UKF.py:477) pass x = 0 and read back K y. */
int fk_ukf_correct_f64(int32_t n, int32_t m, int64_t N, int32_t layout,
                       const double *Pxz, const double *zp, const double *S, const double *z,
                       double *x, double *P, double *K, int32_t *status, void *stream);

/* fk_ukf_desc.flags */
enum {
    FK_UKF_FLAG_PAIR_WEIGHTS = 1   /* the caller asserts Wm[1+k] == Wm[1+n+k] and Wc[1+k] == Wc[1+n+k] for k = 0..n-1 (true of
                                      MerweScaledSigmaPoints and JulierSigmaPoints, sigma_points.py:180-192, :358-372): the sums
                                      of unscented_transform.py:104-126 and UKF.py:483-497 are then formed over the n +- PAIRS of
                                      sigma points (a re-association of the reference's index-order sums: every factor, image and
                                      weighted sum is still formed, with n + 1 instead of 2n + 1 rank-one terms per covariance).
                                      Without the flag -- and for every other weight set -- the sums run in the reference's index
                                      order.  Environment FK_UKF_PAIRED=0 ignores the flag (A/B).  A violated assertion is
                                      reported as FK_STATUS_BAD_WEIGHTS on every track. */
};

typedef struct fk_ukf_desc {
    int32_t n, m;         /* dim_x, dim_z */
    int64_t N, T;
    int32_t layout;
    int32_t flags;        /* FK_UKF_FLAG_* (0: index-order sums) */
    double  scale;        /* lambda + n */
} fk_ukf_desc;

/* UnscentedKalmanFilter.batch_filter (UKF.py:524-632) with LINEAR fx(x,dt)=F x, hx(x)=H x,
 * fused per track: T x { predict (UKF.py:400-411: sigma points, F sigma, UT+Q, regenerate
 * sigma points) ; update (UKF.py:462-481: H sigma, UT+R, cross variance, K = Pxz S^-1,
 * x += K y, P -= K S K') }.
 *   F [n*n], H [m*n], Q [n*n], R [m*m], Wm, Wc [2n+1] : shared device arrays.
 *   z [T][N][m], mask [T][N] or NULL; x [N][n], P [N][n*n] in/out;
 *   means [T][N][n], covs [T][N][n*n] posterior per step (NULL = not stored).
 * Sizes: dim_x 1..6 with dim_z 1..3, dim_x 7..9 with dim_z 1..4 (one track per lane, ukf_kernels.hip); dim_x 10..16 with
 * dim_z 1..8 on four / eight lanes per track (ukf_mlg.hip) for FK_UKF_FLAG_PAIR_WEIGHTS callers; FK_ERR_UNSUPPORTED
 * otherwise (the building blocks serve every size) -- fk_ukf_linear_supported answers without a launch. */
int fk_ukf_linear_batch_f64(const fk_ukf_desc *desc,
                            const double *F, const double *H, const double *Q, const double *R,
                            const double *Wm, const double *Wc,
                            const double *z, const uint8_t *mask,
                            double *x, double *P, double *means, double *covs,
                            int32_t *status, void *stream);

/* UnscentedKalmanFilter.rts_smoother (filterpy/kalman/UKF.py:634-739) with LINEAR fx(x, dt) = F x, fused per track:
 * the whole backward loop in one launch (per step: sigma points of (xs[k], ps[k]) -> F sigma -> unscented transform
 * + Q -> cross variance around Xs[k] / xb -> K = Pxb inv(Pb) -> xs[k] += K (xs[k+1] - xb), ps[k] += K (ps[k+1] - Pb) K').
 * desc: n (1..9; 10..16 on four / eight lanes per track for FK_UKF_FLAG_PAIR_WEIGHTS callers, like
 *   fk_ukf_linear_batch_f64; pair-weight callers get those kernels from dim_x 7 on), N, T, layout, scale = lambda + n (m is ignored).
 *   F [n*n], Q [n*n] (the filter's Q: the reference never reads its Qs argument, UKF.py:717-722), Wm, Wc [2n+1];
 *   Xs [T][N][n], Ps [T][N][n*n]: the filter output; xs, Ps_out likewise: the smoothed output (distinct arrays); K
 *   [T][N][n*n] or NULL (K of the last step is zero, like the reference's); status [N] or NULL.
 * Asynchronous on `stream` and ordered like one kernel on it; at dim_x >= 5 a call whose last round of waves would be
 * mostly idle fans out over up to three helper streams like fk_kf_batch_filter_f64 (bit-identical; FK_UKF_RTS_CHUNKS=1,1
 * turns it off; INTEGRATION.md, "Streams"). */
int fk_ukf_linear_rts_f64(const fk_ukf_desc *desc, const double *F, const double *Q, const double *Wm, const double *Wc,
                          const double *Xs, const double *Ps, double *xs, double *Ps_out, double *K, int32_t *status,
                          void *stream);

/* Will fk_ukf_linear_batch_f64 (smoother = 0) / fk_ukf_linear_rts_f64 (smoother != 0; m ignored) take this size with these
 * fk_ukf_desc.flags?  1 / 0; no launch, no device needed.  The host side of UnscentedKalmanFilter.batch_filter / rts_smoother
 * (filterpy/kalman/UKF.py:524-739) asks before choosing between the fused launch and the per-step building blocks, so that the
 * two sides cannot disagree about the library's A/B switches (FK_UKF_MLG, read once per process by the library). */
int fk_ukf_linear_supported(int32_t n, int32_t m, int32_t flags, int32_t smoother);

/* ------------------------------------------------------------------ */
/* API variants of the linear filter               */
/* ------------------------------------------------------------------ */

/* KalmanFilter.predict_steadystate (filterpy/kalman/kalman_filter.py:563-593) and
 * update_steadystate (:595-668) for N tracks: only x moves, the gain K is fixed.
 * T x { x = F x (+ B u) ; means_p[t] = x ; y = z - H x ; x += K y ; means[t] = x }.
 *   desc: n (1..16), m (1..8), nu (0..4), N, T, layout; model_mode FK_MODEL_SHARED: K [n*m] shared,
 *   FK_MODEL_PER_TRACK: K [N][n*m] records (every track its own converged gain).
 *   F [n*n], H [m*n], B [n*nu]: shared.  F == NULL: no predict (update_steadystate alone);
 *   z == NULL: no update (predict_steadystate alone).  u [T][N][nu], z [T][N][m], mask [T][N] or NULL
 *   (0 = z is None: x unchanged, y = 0, kalman_filter.py:647-652).
 *   x [N][n] in/out; means, means_p [T][N][n], y_out [T][N][m]: optional outputs. */
int fk_kf_steadystate_f64(const fk_kf_desc *desc,
                          const double *F, const double *H, const double *K,
                          const double *B, const double *u,
                          const double *z, const uint8_t *mask,
                          double *x, double *means, double *means_p, double *y_out,
                          void *stream);

/* KalmanFilter.update_correlated (kalman_filter.py:670-752): process and measurement noise correlated
 * through M (dim_x x dim_z):  y = z - H x ; S = H P H' + H M + M' H' + R ; K = (P H' + M) S^-1 ;
 * x += K y ; P -= K (H P + M').   One update of N tracks.
 *   desc: n (1..16), m (1..8), N, layout; model_mode FK_MODEL_SHARED: M [n*m] shared,
 *   FK_MODEL_PER_TRACK: M [N][n*m].  H [m*n], R [m*m]: shared.  z [N][m], mask [N] or NULL.
 *   x [N][n], P [N][n*n] in/out; y [N][m], K [N][n*m], S, SI [N][m*m]: optional outputs; status [N] or NULL. */
int fk_kf_update_correlated_f64(const fk_kf_desc *desc,
                                const double *H, const double *R, const double *M,
                                const double *z, const uint8_t *mask,
                                double *x, double *P,
                                double *y, double *K, double *S, double *SI,
                                int32_t *status, void *stream);

/* The gain / correction of one backward step of UnscentedKalmanFilter.rts_smoother
 * (filterpy/kalman/UKF.py:726-733):  K = Pxb inv(Pb) ; x += K (xn - xb) ; P += K (Pn - Pb) K'.
 *   Pxb [N][n*n] (cross variance of the sigma points around Xs[k] and their images around xb),
 *   xb [N][n], Pb [N][n*n] (unscented transform of the propagated sigma points + Q),
 *   xn [N][n], Pn [N][n*n] (smoothed step k+1); x [N][n], P [N][n*n]: filtered step k in, smoothed out;
 *   K [N][n*n] out (may be NULL); status [N] or NULL.  dim_x 1..9.
 * xb = NULL: xn already holds residual_x(xs[k+1], xb) (UKF.py:731 with a custom callable).
 * The sigma points, their transform and the cross variance are fk_ut_sigma_points_f64,
 * fk_ut_transform_f64 and fk_ut_cross_variance_f64. */
int fk_ukf_rts_correct_f64(int32_t n, int64_t N, int32_t layout,
                           const double *Pxb, const double *xb, const double *Pb,
                           const double *xn, const double *Pn,
                           double *x, double *P, double *K,
                           int32_t *status, void *stream);

/* ------------------------------------------------------------------ */
/* Interacting multiple models                     */
/* ------------------------------------------------------------------ */

typedef struct fk_imm_desc {
    int32_t n, m;         /* dim_x (1..16), dim_z (1..8): the same for every filter of the bank */
    int32_t n_models;     /* filters per track: 2 .. 16 (9 .. 16: the rolled general kernel) */
    int32_t layout;
    int64_t N, T;
    int32_t phase;        /* FK_IMM_STEP: T x {predict; update}; FK_IMM_PREDICT / FK_IMM_UPDATE: that half once */
    int32_t flags;        /* FK_IMM_FLAG_MMAE: MMAEFilterBank arithmetic (below) */
} fk_imm_desc;

enum { FK_IMM_STEP = 0, FK_IMM_PREDICT = 1, FK_IMM_UPDATE = 2 };
enum { FK_IMM_FLAG_MMAE = 1 };

/* filterpy.kalman.IMMEstimator (filterpy/kalman/IMM.py) for N independent tracks, each with its own
 * bank of n_models linear Kalman filters, T x { predict() (IMM.py:188-222) ; update(z) (IMM.py:160-186) }
 * in one launch: mixing probabilities (IMM.py:239-249), mixed initial conditions, each filter's
 * KalmanFilter.predict/update (kalman_filter.py:472-478, 533-556), likelihood = exp(logpdf(y; 0, S))
 * floored at DBL_MIN (kalman_filter.py:1213-1226), mode probabilities, mixed estimate (IMM.py:224-237).
 *   F, Q [n_models][n*n], H [n_models][m*n], R [n_models][m*m], M [n_models][n_models] (mode
 *   transition probabilities): device arrays shared by all tracks.
 *   z [T][N][m] (every measurement present).
 *   xs [N][n_models*n], Ps [N][n_models*n*n], mu [N][n_models]: the banks' filter states and mode
 *   probabilities (mu normalised by the caller like IMM.py:129), updated in place.
 *   x_out [T][N][n], P_out [T][N][n*n], mu_out [T][N][n_models]: IMMEstimator.x/.P/.mu after each
 *   update; x_prior_out, P_prior_out: after each predict; likelihood_out [T][N][n_models].
 *   Any output may be NULL.  status [N] or NULL.
 * phase FK_IMM_PREDICT runs IMMEstimator.predict() once (mixing + every filter's predict; writes
 * xs, Ps and the prior outputs [N][..]), FK_IMM_UPDATE runs IMMEstimator.update(z) once (z [N][m];
 * writes xs, Ps, mu and x_out/P_out/mu_out/likelihood_out [N][..]); T is ignored for both.
 * Asynchronous on `stream` and ordered like one kernel on it; a whole-step call whose last round of waves would be mostly
 * idle fans out over up to three helper streams like fk_kf_batch_filter_f64 (bit-identical; FK_IMM_CHUNKS=1,1 turns it
 * off; INTEGRATION.md, "Streams").
 *
 * flags & FK_IMM_FLAG_MMAE: filterpy.kalman.MMAEFilterBank (filterpy/kalman/mmae.py:140-212) on the
 * same records: no mixing (every filter predicts from its own state, mmae.py:153-154; M is unused and
 * may be NULL), p_i *= likelihood_i then normalised (mmae.py:185-189; mu holds p), x = sum p_i x_i and
 * the covariance exactly as mmae.py:205-207 computes it (the loop zips the COMPONENTS of x with the
 * filters: P = sum over k < min(dim_x, n_models) of p_k (outer(x_k - x[k]) + P_k), reproduced as is).
 * The prior outputs are not defined for MMAE (the reference's x_prior is a copy of the last x). */
/* fk_imm_batch_f64 with MISSING measurements (IMMEstimator.update(None) / MMAEFilterBank.update(None): IMM.py:171-186,
 * mmae.py:160-212 on top of kalman_filter.py:511-520, :1203-1226).
 *   zmask [T][N] (t-major), 0 = the measurement of that track and step is None: the filters keep x and P, the
 *         likelihood of each is the density of a ZERO residual under the S of ITS last real update, and the mode
 *         probabilities are re-weighted with those numbers, mixed and re-estimated -- like the reference; NULL = none.
 *   ll0   [N][n_models] in/out record, or NULL: -(m ln 2 pi + ln |S_j|) / 2 of each filter's last real update,
 *         -inf for a filter that has not seen one (S = 0: the reference's density evaluates to 0 and is floored at
 *         DBL_MIN like every likelihood).  Keeps the bookkeeping across launches (the call-by-call API); NULL starts
 *         every filter at -inf and drops the result.
 * ... and with a CONTROL input (IMMEstimator.predict(u) / MMAEFilterBank.predict(u) hand u to every filter's predict,
 * kalman_filter.py:472-475: x = F x + B u):
 *   nu    dim_u (0 = none); B [n_models][n*nu]: every filter's own B; u [T][N][nu] record array in `layout`
 *         ([N][nu] for a single phase). */
int fk_imm_batch_ex_f64(const fk_imm_desc *desc,
                            const double *F, const double *Q, const double *H, const double *R, const double *M,
                            const double *z, const uint8_t *zmask, double *ll0,
                            int32_t nu, const double *B, const double *u,
                            double *xs, double *Ps, double *mu,
                            double *x_out, double *P_out, double *mu_out,
                            double *x_prior_out, double *P_prior_out, double *likelihood_out,
                            int32_t *status, void *stream);

int fk_imm_batch_f64(const fk_imm_desc *desc,
                     const double *F, const double *Q, const double *H, const double *R,
                     const double *M, const double *z,
                     double *xs, double *Ps, double *mu,
                     double *x_out, double *P_out, double *mu_out,
                     double *x_prior_out, double *P_prior_out, double *likelihood_out,
                     int32_t *status, void *stream);

/* ------------------------------------------------------------------ */
/* Particle-filter resampling                                         */
/* ------------------------------------------------------------------ */

/* systematic_resample (filterpy/monte_carlo/resampling.py:117-150) for Fn independent
 * filters of Np particles:  pos_i = fl(fl(u_f + i)/Np); cs = cumsum(w_f) reproduced
 * bit-for-bit as a sequential fp64 add chain; idx_i = #{ j : cs_j <= pos_i }.
 *   w   [Fn][Np] weights;  u [Fn] one uniform per filter (drawn by the host from numpy.random);
 *   idx [Fn][Np] int32 (np.zeros(N,'i'), resampling.py:141);  status [Fn] or NULL.
 *   ws / ws_bytes : scratch from fk_resample_workspace_bytes(Fn, Np); with it, weight vectors of
 *                   >= 32768 particles are processed chunk-parallel (many workgroups per filter);
 *                   NULL / too small = one workgroup per filter.  16-byte aligned (FK_ERR_WORKSPACE otherwise;
 *                   any device allocation is). */
int fk_resample_systematic_f64(int64_t Fn, int64_t Np, const double *w, const double *u,
                               int32_t *idx, int32_t *status,
                               void *ws, size_t ws_bytes, void *stream);

/* stratified_resample (resampling.py:80-114): pos_i = fl(fl(u_{f,i} + i)/Np), u [Fn][Np]. */
int fk_resample_stratified_f64(int64_t Fn, int64_t Np, const double *w, const double *u,
                               int32_t *idx, int32_t *status,
                               void *ws, size_t ws_bytes, void *stream);

/* multinomial_resample (resampling.py:153-176) and the random tail of residual_resample
 * (:72-76):  cs = cumsum(w_f) sequential, cs[-1] = 1.0, idx_i = searchsorted(cs, u_{f,i}) (side
 * 'left'), for Nu draws per filter.   u [Fn][Nu] -> idx [Fn][Nu] int64 (np.intp, as the
 * reference returns). */
int fk_resample_multinomial_f64(int64_t Fn, int64_t Np, int64_t Nu, const double *w, const double *u,
                                int64_t *idx, void *ws, size_t ws_bytes, void *stream);

/* residual_resample (resampling.py:27-76) for Fn filters, in two launches around the host's draw of the uniforms (how many
 * the reference consumes from numpy.random depends on the weights: random(N - k)).
 *   fill: num_copies = floor(Np w) (:61); the deterministic copies idx[f][0 .. k_f) (:63-66); k_f -> k [Fn] (int64);
 *         cs [Fn][Np] (caller's buffer) <- cumsum((w - num_copies) / sum(w - num_copies)) with cs[-1] = 1 (:70-74; the sum
 *         is the Python builtin's left-to-right chain); status[f] = FK_STATUS_OVERRUN when k_f > Np (the reference's
 *         IndexError), nothing else is written for that filter.
 *   draw: idx[f][k_f + i] = searchsorted(cs_f, u[uoff[f] + i]) for i < Np - k_f (:75-76); u holds every filter's
 *         Np - k_f uniforms back to back, uoff [Fn] (int64) their offsets.   idx is int32 like the reference's. */
int fk_resample_residual_fill_f64(int64_t Fn, int64_t Np, const double *w, int32_t *idx, int64_t *k, double *cs,
                                  int32_t *status, void *stream);
int fk_resample_residual_draw_f64(int64_t Fn, int64_t Np, const double *cs, const int64_t *k, const int64_t *uoff,
                                  const double *u, int32_t *idx, void *stream);

size_t fk_resample_workspace_bytes(int64_t Fn, int64_t Np);     /* systematic / stratified */
size_t fk_multinomial_workspace_bytes(int64_t Fn, int64_t Np);  /* multinomial: Fn*Np doubles */

/* Posterior mean of a resampled particle set, mean[f][k] = (1/Np) sum_i particles[f][idx[f][i]][k], without
 * materialising the resampled copy.  NOT a filterpy function: it is the "resample from index" step every
 * caller of the resamplers writes (particles[:] = particles[indexes], docs/monte_carlo/resampling.rst) fused
 * with the mean that BASELINE configs[4] all-gathers across GPUs.
 *   particles [Fn][Np][d] (d = 1..8), idx [Fn][Np] (output of fk_resample_*), mean [Fn][d] out.
 * Partial sums meet in fp64 atomic adds: the mean is exact up to summation-order rounding. */
int fk_resample_gather_mean_f64(int64_t Fn, int64_t Np, int32_t d,
                                const double *particles, const int32_t *idx,
                                double *mean, void *stream);

/* numpy.cumsum(w_f) for Fn float64 vectors of length Np, bit-for-bit (NumPy adds strictly left to
 * right; a re-associated parallel scan differs in the last bits -- here an associative scan over
 * integer rounding maps reproduces the sequential result exactly, see csrc/fk_exact_scan.hpp).
 * The building block of every resampler (resampling.py:72,106,142,174).
 *   w [Fn][Np] -> cs [Fn][Np];  force_last_one != 0 also sets cs[f][Np-1] = 1.0 (resampling.py:74,175). */
int fk_cumsum_exact_f64(int64_t Fn, int64_t Np, const double *w, double *cs, int32_t force_last_one,
                        void *stream);

/* ------------------------------------------------------------------ */
/* Utilities                                                          */
/* ------------------------------------------------------------------ */
int fk_abi_version(void);
/* name of the gfx target the kernels were compiled for ("gfx950") */
const char *fk_build_arch(void);
/* last HIP error string seen by this library on the calling thread ("" if none) */
const char *fk_last_error(void);
/* How a call would be cut for tail filling (INTEGRATION.md "Streams", csrc/fk_chunks.hpp) -- host arithmetic only:
 * n_tracks tracks at tracks_per_wave per wave on wave_slots resident waves, n_steps time steps (the smoother: T - 1
 * backward steps).  Writes the time windows [w0, w1) of track group `group` into windows[2 * i], windows[2 * i + 1]
 * (room for 2 * 65 values), the group and chunk counts into n_groups / n_chunks (1, 1: one launch); returns the number
 * of windows, -1 on a bad argument.  FK_ML_CHUNKS="G,H" in the environment overrides the policy. */
int fk_chunk_plan(int64_t n_tracks, int64_t n_steps, int32_t tracks_per_wave, int64_t wave_slots, int32_t group,
                  int64_t *windows, int32_t *n_groups, int32_t *n_chunks);

#ifdef __cplusplus
}
#endif
#endif /* FILTERHIP_H */
