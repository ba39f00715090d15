// kf_variants.hip -- the remaining API variants of the linear filter: small
// kernels next to the batch_filter path, one track per lane, everything in registers.
//
//   steady_kernel        KalmanFilter.predict_steadystate / update_steadystate
//                        (filterpy/kalman/kalman_filter.py:563-593, 595-668): only x moves, with a
//                        fixed gain K; T x { x = F x (+ B u) ; y = z - H x ; x += K y } in one launch
//   corr_update_kernel   KalmanFilter.update_correlated (kalman_filter.py:670-752): process and
//                        measurement noise correlated through M (dim_x x dim_z)
//   ukf_rts_kernel       the gain / correction of UnscentedKalmanFilter.rts_smoother
//                        (filterpy/kalman/UKF.py:726-733): K = Pxb inv(Pb) ; x += K (xn - xb) ;
//                        P += K (Pn - Pb) K'
// (update_sequential, kalman_filter.py:754-824, is the ordinary update on a slice of z, H, R and
// runs on fk_kf_update_f64: host shim only.)
#include "fk_device.hpp"
#include "fk_kernel_args.hpp"
#include "../../include/filterhip.h"

namespace fk {

struct SteadyArgs {
    const double *F, *H, *K, *B, *u, *z;
    const uint8_t *mask;
    double *x, *means, *means_p, *y_out;
    long N, T;
    int n, m, nu, k_per_track;
};

// F (NX x NX), H (NZ x NX) and the shared K (NX x NZ) sit in LDS; a per-track K in registers.
// EXACT (n == NX, m == NZ): the x / z / y records move without per-element guards -- in NumPy order (AOS) as 16-byte
// pairs, half as many memory operations (steady-state AOS measured 0.24-0.38 of HBM with the guarded 8-byte accesses).
#ifndef FK_STEADY_KLDS_WAVES
#define FK_STEADY_KLDS_WAVES 3
#endif
// KREG = false (round 5, the padded classes (12,8) / (16,8) with a shared gain): K is read from LDS where it is used instead of
// being held in 2 NX NZ registers per lane -- with unrolled loops that is the difference between ~120 and ~380 VGPRs.
template <int NX, int NZ, int LAYOUT, bool EXACT, bool KREG = true, int WAVES = (KREG ? 1 : FK_STEADY_KLDS_WAVES)>
__global__ void __launch_bounds__(BLOCK, WAVES)
steady_kernel(const SteadyArgs a)
{
    constexpr int NU = 4;
    // COOP (round 3; NumPy order, exact dims): the x / z / y records move between HBM and the registers through a
    // wave-private LDS tile, memory order on the HBM side (wave_load_aos / wave_store_aos: 0.5-1 KiB contiguous per
    // instruction) -- a lane-per-record 16-byte access touches 64 lines per instruction (measured 0.34 of HBM at (9,3))
    constexpr bool COOP = LAYOUT == LAYOUT_AOS && EXACT;
    __shared__ double sF[NX * NX], sH[NZ * NX], sK[NX * NZ], sB[NX * NU];
    __shared__ double s_tile[COOP ? (BLOCK / 64) * 64 * (NX | 1) : 1];
    const int n = a.n, m = a.m, nu = a.nu;
    const long N = a.N;
    lds_fill<NX, NX>(sF, a.F, n, n, 1.0, threadIdx.x);
    lds_fill<NZ, NX>(sH, a.H, m, n, 0.0, threadIdx.x);
    lds_fill<NX, NZ>(sK, a.k_per_track ? nullptr : a.K, n, m, 0.0, threadIdx.x);
    lds_fill<NX, NU>(sB, a.B, n, nu, 0.0, threadIdx.x);
    __syncthreads();
    const long blk0 = (long)blockIdx.x * BLOCK;
    const long left = N - blk0;                                            // >= 1
    const unsigned last_row = (unsigned)(left < BLOCK ? left : BLOCK) - 1u;
    if (!COOP && threadIdx.x > last_row) return;
    // COOP: every lane of a wave takes part in the moves; lanes past the last track compute on zeros (the buffer
    // descriptors end at the last track: their loads return 0, their stores are dropped)
    const Lane ln{blk0, COOP ? min(threadIdx.x, last_row) : threadIdx.x, N};
    const unsigned lane = threadIdx.x & 63u, wrow = (threadIdx.x >> 6) * 64u;
    double *tile = s_tile + (COOP ? (threadIdx.x >> 6) * 64 * (NX | 1) : 0);
    double x[NX], K[KREG ? NX * NZ : 1];
    if constexpr (COOP) wave_load_aos<NX>(x, a.x + blk0 * NX, wrow, tile, lane, last_row);
    else load_rec<NX, 1, LAYOUT, EXACT>(x, a.x, ln, n, 1, 0.0);
    if constexpr (KREG) {
        if (a.k_per_track) {
            load_rec<NX, NZ, LAYOUT, false>(K, a.K, ln, n, m, 0.0);
        } else {
            FK_UNROLL for (int e = 0; e < NX * NZ; ++e) K[e] = sK[e];
        }
    }
    for (long t = 0; t < a.T; ++t) {
        if (a.F) {      // predict_steadystate: x = F x (+ B u)
            double xn[NX];
            FK_UNROLL for (int i = 0; i < NX; ++i) {
                double acc = sF[i * NX] * x[0];
                FK_UNROLL for (int k = 1; k < NX; ++k) acc = fma(sF[i * NX + k], x[k], acc);
                xn[i] = acc;
            }
            if (a.B) {
                double u[NU], Bu[NX];
                load_rec<NU, 1, LAYOUT, false>(u, a.u + t * N * nu, ln, nu, 1, 0.0);
                FK_UNROLL for (int i = 0; i < NX; ++i) {
                    double acc = sB[i * NU] * u[0];
                    FK_UNROLL for (int k = 1; k < NU; ++k) acc = fma(sB[i * NU + k], u[k], acc);
                    Bu[i] = acc;
                }
                FK_UNROLL for (int i = 0; i < NX; ++i) xn[i] += Bu[i];
            }
            FK_UNROLL for (int i = 0; i < NX; ++i) x[i] = xn[i];
            if (a.means_p) {
                if constexpr (COOP) wave_store_aos<NX>(x, a.means_p + (t * N + blk0) * NX, wrow, tile, lane, last_row);
                else store_rec<NX, 1, LAYOUT, EXACT>(x, a.means_p + t * N * n, ln, n, 1);
            }
        }
        if (a.z) {      // update_steadystate: y = z - H x ; x += K y
            double z[NZ], y[NZ];
            if constexpr (COOP) wave_load_aos<NZ>(z, a.z + (t * N + blk0) * NZ, wrow, tile, lane, last_row);
            else load_rec<NZ, 1, LAYOUT, EXACT>(z, a.z + t * N * m, ln, m, 1, 0.0);
            const bool has_z = !a.mask || a.mask[t * N + ln.blk0 + ln.tid];
            FK_UNROLL for (int r = 0; r < NZ; ++r) {
                double acc = sH[r * NX] * x[0];
                FK_UNROLL for (int k = 1; k < NX; ++k) acc = fma(sH[r * NX + k], x[k], acc);
                y[r] = has_z ? z[r] - acc : 0.0;
            }
            if (has_z) {
                FK_UNROLL for (int i = 0; i < NX; ++i) {
                    double acc = (KREG ? K[KREG ? i * NZ : 0] : sK[i * NZ]) * y[0];
                    FK_UNROLL for (int k = 1; k < NZ; ++k) acc = fma(KREG ? K[KREG ? i * NZ + k : 0] : sK[i * NZ + k], y[k], acc);
                    x[i] += acc;
                }
            }
            if constexpr (COOP) {
                if (a.y_out) wave_store_aos<NZ>(y, a.y_out + (t * N + blk0) * NZ, wrow, tile, lane, last_row);
                if (a.means) wave_store_aos<NX>(x, a.means + (t * N + blk0) * NX, wrow, tile, lane, last_row);
            } else {
                if (a.y_out) store_rec<NZ, 1, LAYOUT, EXACT>(y, a.y_out + t * N * m, ln, m, 1);
                if (a.means) store_rec<NX, 1, LAYOUT, EXACT>(x, a.means + t * N * n, ln, n, 1);
            }
        }
    }
    // (in place: a wave reads and writes its own 64 rows only)
    if constexpr (COOP) wave_store_aos<NX>(x, a.x + blk0 * NX, wrow, tile, lane, last_row);
    else store_rec<NX, 1, LAYOUT, EXACT>(x, a.x, ln, n, 1);
}

// update_correlated (kalman_filter.py:727-748), in the reference's association order:
//   y = z - H x ; PHT = P H' ; S = ((H PHT + H M) + M' H') + R ; K = (PHT + M) S^-1 ;
//   x += K y ; P = P - K (H P + M')
template <int NX, int NZ, int LAYOUT>
__global__ void __launch_bounds__(BLOCK)
corr_update_kernel(int n, int m, long N, const double *__restrict__ pH, const double *__restrict__ pR,
                   const double *__restrict__ pM, int m_per_track, const double *__restrict__ pz,
                   const uint8_t *__restrict__ mask, double *px, double *pP, double *py, double *pK, double *pS,
                   double *pSI, int32_t *status)
{
    __shared__ double sH[NZ * NX], sR[NZ * NZ], sM[NX * NZ];
    lds_fill<NZ, NX>(sH, pH, m, n, 0.0, threadIdx.x);
    lds_fill<NZ, NZ>(sR, pR, m, m, 1.0, threadIdx.x);
    lds_fill<NX, NZ>(sM, m_per_track ? nullptr : pM, n, m, 0.0, threadIdx.x);
    __syncthreads();
    const Lane ln{(long)blockIdx.x * BLOCK, threadIdx.x, N};
    if (ln.blk0 + ln.tid >= N) return;
    double y[NZ];
    if (mask && !mask[ln.blk0 + ln.tid]) {     // z is None: only y is reset (kalman_filter.py:705-710)
        FK_UNROLL for (int r = 0; r < NZ; ++r) y[r] = 0.0;
        if (py) store_rec<NZ, 1, LAYOUT, false>(y, py, ln, m, 1);
        if (status) status[ln.blk0 + ln.tid] = 0;
        return;
    }
    double x[NX], P[NX * NX], M[NX * NZ], z[NZ];
    load_rec<NX, 1, LAYOUT, false>(x, px, ln, n, 1, 0.0);
    load_rec<NX, NX, LAYOUT, false>(P, pP, ln, n, n, 1.0);
    load_rec<NZ, 1, LAYOUT, false>(z, pz, ln, m, 1, 0.0);
    if (m_per_track) {
        load_rec<NX, NZ, LAYOUT, false>(M, pM, ln, n, m, 0.0);
    } else {
        FK_UNROLL for (int e = 0; e < NX * NZ; ++e) M[e] = sM[e];
    }
    double PHT[NX * NZ], S[NZ * NZ];
    FK_UNROLL for (int r = 0; r < NZ; ++r) {
        double acc = sH[r * NX] * x[0];
        FK_UNROLL for (int k = 1; k < NX; ++k) acc = fma(sH[r * NX + k], x[k], acc);
        y[r] = z[r] - acc;
        FK_UNROLL for (int i = 0; i < NX; ++i) {
            double a2 = P[i * NX] * sH[r * NX];
            FK_UNROLL for (int k = 1; k < NX; ++k) a2 = fma(P[i * NX + k], sH[r * NX + k], a2);
            PHT[i * NZ + r] = a2;
        }
    }
    FK_UNROLL for (int r = 0; r < NZ; ++r)
        FK_UNROLL for (int c = 0; c < NZ; ++c) {
            double hp = sH[r * NX] * PHT[c], hm = sH[r * NX] * M[c], mh = M[r] * sH[c * NX];
            FK_UNROLL for (int k = 1; k < NX; ++k) {
                hp = fma(sH[r * NX + k], PHT[k * NZ + c], hp);
                hm = fma(sH[r * NX + k], M[k * NZ + c], hm);
                mh = fma(M[k * NZ + r], sH[c * NX + k], mh);
            }
            S[r * NZ + c] = ((hp + hm) + mh) + sR[r * NZ + c];
        }
    int st = 0;
    double K[NX * NZ], Lf[NZ * NZ], d[NZ], dinv[NZ];
    FK_UNROLL for (int e = 0; e < NX * NZ; ++e) K[e] = PHT[e] + M[e];
    FK_UNROLL for (int e = 0; e < NZ * NZ; ++e) Lf[e] = S[e];
    if (!ldlt2<NZ>(Lf, d, dinv)) st |= ST_NOT_PD;
    solve_rows_ldlt<NX, NZ>(Lf, dinv, K);
    FK_UNROLL for (int i = 0; i < NX; ++i) {
        double acc = K[i * NZ] * y[0];
        FK_UNROLL for (int k = 1; k < NZ; ++k) acc = fma(K[i * NZ + k], y[k], acc);
        x[i] += acc;
    }
    // W = H P + M'  (NZ x NX) ;  P -= K W
    double W[NZ * NX];
    FK_UNROLL for (int r = 0; r < NZ; ++r)
        FK_UNROLL for (int c = 0; c < NX; ++c) {
            double acc = sH[r * NX] * P[c];
            FK_UNROLL for (int k = 1; k < NX; ++k) acc = fma(sH[r * NX + k], P[k * NX + c], acc);
            W[r * NX + c] = acc + M[c * NZ + r];
        }
    FK_UNROLL for (int i = 0; i < NX; ++i)
        FK_UNROLL for (int c = 0; c < NX; ++c) {
            double acc = K[i * NZ] * W[c];
            FK_UNROLL for (int k = 1; k < NZ; ++k) acc = fma(K[i * NZ + k], W[k * NX + c], acc);
            P[i * NX + c] -= acc;
        }
    store_rec<NX, 1, LAYOUT, false>(x, px, ln, n, 1);
    store_rec<NX, NX, LAYOUT, false>(P, pP, ln, n, n);
    if (py) store_rec<NZ, 1, LAYOUT, false>(y, py, ln, m, 1);
    if (pK) store_rec<NX, NZ, LAYOUT, false>(K, pK, ln, n, m);
    if (pS) store_rec<NZ, NZ, LAYOUT, false>(S, pS, ln, m, m);
    if (pSI) {
        double SI[NZ * NZ];
        inv_from_ldlt<NZ>(Lf, dinv, SI);
        store_rec<NZ, NZ, LAYOUT, false>(SI, pSI, ln, m, m);
    }
    if (status) {
        if (!all_finite<NX>(x) || !all_finite<NX * NX>(P)) st |= ST_NONFINITE;
        status[ln.blk0 + ln.tid] = st;
    }
}

// UKF.py:726-733:  K = Pxb inv(Pb) ; x += K (xn - xb) ; P += (K (Pn - Pb)) K'
template <int NX, int LAYOUT>
__global__ void __launch_bounds__(BLOCK)
ukf_rts_kernel(int n, long N, const double *__restrict__ pPxb, const double *__restrict__ pxb,
               const double *__restrict__ pPb, const double *__restrict__ pxn, const double *__restrict__ pPn,
               double *px, double *pP, double *pK, int32_t *status)
{
    const Lane ln{(long)blockIdx.x * BLOCK, threadIdx.x, N};
    if (ln.blk0 + ln.tid >= N) return;
    double K[NX * NX], Lf[NX * NX], d[NX], dinv[NX];
    load_rec<NX, NX, LAYOUT, false>(K, pPxb, ln, n, n, 0.0);
    load_rec<NX, NX, LAYOUT, false>(Lf, pPb, ln, n, n, 1.0);
    int st = 0;
    if (!ldlt2<NX>(Lf, d, dinv)) st |= ST_NOT_PD;
    solve_rows_ldlt<NX, NX>(Lf, dinv, K);
    double x[NX], dx[NX];
    {
        double xn[NX], xb[NX];
        load_rec<NX, 1, LAYOUT, false>(x, px, ln, n, 1, 0.0);
        load_rec<NX, 1, LAYOUT, false>(xn, pxn, ln, n, 1, 0.0);
        if (pxb) load_rec<NX, 1, LAYOUT, false>(xb, pxb, ln, n, 1, 0.0);   // nullptr: xn IS residual_x(xs[k+1], xb) (UKF.py:731)
        else { FK_UNROLL for (int i = 0; i < NX; ++i) xb[i] = 0.0; }
        FK_UNROLL for (int i = 0; i < NX; ++i) dx[i] = xn[i] - xb[i];
    }
    FK_UNROLL for (int i = 0; i < NX; ++i) {
        double acc = K[i * NX] * dx[0];
        FK_UNROLL for (int k = 1; k < NX; ++k) acc = fma(K[i * NX + k], dx[k], acc);
        x[i] += acc;
    }
    store_rec<NX, 1, LAYOUT, false>(x, px, ln, n, 1);
    if (pK) store_rec<NX, NX, LAYOUT, false>(K, pK, ln, n, n);
    // D = Pn - Pb ; T1 = K D ; P += T1 K'
    double D[NX * NX], T1[NX * NX];
    {
        double Pb[NX * NX];
        load_rec<NX, NX, LAYOUT, false>(D, pPn, ln, n, n, 0.0);
        load_rec<NX, NX, LAYOUT, false>(Pb, pPb, ln, n, n, 0.0);
        FK_UNROLL for (int e = 0; e < NX * NX; ++e) D[e] -= Pb[e];
    }
    matmul<NX, NX, NX>(K, D, T1);
    double P[NX * NX];
    load_rec<NX, NX, LAYOUT, false>(P, pP, ln, n, n, 0.0);
    FK_UNROLL for (int i = 0; i < NX; ++i)
        FK_UNROLL for (int j = 0; j < NX; ++j) {
            double acc = T1[i * NX] * K[j * NX];
            FK_UNROLL for (int k = 1; k < NX; ++k) acc = fma(T1[i * NX + k], K[j * NX + k], acc);
            P[i * NX + j] += acc;
        }
    store_rec<NX, NX, LAYOUT, false>(P, pP, ln, n, n);
    if (status) {
        if (!all_finite<NX>(x) || !all_finite<NX * NX>(P)) st |= ST_NONFINITE;
        status[ln.blk0 + ln.tid] = st;
    }
}

static int fail(int code, const char *msg)
{
    set_last_error(msg);
    return code;
}

}  // namespace fk

// dim_x 10 .. 16 of fk_ukf_rts_correct_f64: the same kernel on the padded classes 12 and 16, compiled in a unit of its own
// (ukf_rts_big.hip) with rolled loops (FK_ROLLED: the n x n arrays live in scratch) -- slow, but the general-fx UKF smoother
// needs them (UnscentedKalmanFilter.rts_smoother at dim_x > 9 was an FK_ERR_UNSUPPORTED until round 3)
namespace fk {
int ukf_rts_big_launch(int n, long N, int layout, const double *Pxb, const double *xb, const double *Pb, const double *xn,
                       const double *Pn, double *x, double *P, double *K, int32_t *status, hipStream_t s);
// round 4: the steady-state pair and update_correlated above (9,4) -- the same kernels on the padded classes (12,8) and (16,8)
// in the rolled unit (batch_filter reaches (16,8); these stopped at (9,4): VERDICT r3 missing 3)
int steady_big_launch(const SteadyArgs &a, int layout, hipStream_t s);
int corr_big_launch(int n, int m, long N, int layout, const double *H, const double *R, const double *M, int per_track,
                    const double *z, const uint8_t *mask, double *x, double *P, double *y, double *K, double *S, double *SI,
                    int32_t *status, hipStream_t s);
}
#ifdef FK_VARIANTS_BIG
namespace fk {
int steady_big_launch(const SteadyArgs &a, int layout, hipStream_t s)
{
    const dim3 grid((unsigned)((a.N + BLOCK - 1) / BLOCK)), block(BLOCK);
#define CALL(NXV)                                                                                            \
    if (layout == FK_LAYOUT_SOA) hipLaunchKernelGGL((steady_kernel<NXV, 8, LAYOUT_SOA, false>), grid, block, 0, s, a); \
    else hipLaunchKernelGGL((steady_kernel<NXV, 8, LAYOUT_AOS, false>), grid, block, 0, s, a)
    if (a.n <= 12) { CALL(12); }
    else { CALL(16); }
#undef CALL
    return check_launch("steady_kernel");
}
int corr_big_launch(int n, int m, long N, int layout, const double *H, const double *R, const double *M, int per_track,
                    const double *z, const uint8_t *mask, double *x, double *P, double *y, double *K, double *S, double *SI,
                    int32_t *status, hipStream_t s)
{
    const dim3 grid((unsigned)((N + BLOCK - 1) / BLOCK)), block(BLOCK);
#define CALL(NXV)                                                                                                  \
    if (layout == FK_LAYOUT_SOA)                                                                                   \
        hipLaunchKernelGGL((corr_update_kernel<NXV, 8, LAYOUT_SOA>), grid, block, 0, s, n, m, N, H, R, M, per_track, z, mask, x, P, y, K, S, SI, status); \
    else                                                                                                           \
        hipLaunchKernelGGL((corr_update_kernel<NXV, 8, LAYOUT_AOS>), grid, block, 0, s, n, m, N, H, R, M, per_track, z, mask, x, P, y, K, S, SI, status)
    if (n <= 12) { CALL(12); }
    else { CALL(16); }
#undef CALL
    return check_launch("corr_update_kernel");
}
int ukf_rts_big_launch(int n, long N, int layout, const double *Pxb, const double *xb, const double *Pb, const double *xn,
                       const double *Pn, double *x, double *P, double *K, int32_t *status, hipStream_t s)
{
    const dim3 grid((unsigned)((N + BLOCK - 1) / BLOCK)), block(BLOCK);
#define CALL(NXV)                                                                                                 \
    if (layout == FK_LAYOUT_SOA)                                                                                  \
        hipLaunchKernelGGL((ukf_rts_kernel<NXV, LAYOUT_SOA>), grid, block, 0, s, n, N, Pxb, xb, Pb, xn, Pn, x, P, K, status); \
    else                                                                                                          \
        hipLaunchKernelGGL((ukf_rts_kernel<NXV, LAYOUT_AOS>), grid, block, 0, s, n, N, Pxb, xb, Pb, xn, Pn, x, P, K, status)
    if (n <= 12) { CALL(12); }
    else { CALL(16); }
#undef CALL
    return check_launch("ukf_rts_kernel");
}
}  // namespace fk
#else
using namespace fk;

#define FK_BY_DIMS(n, m, CALL)                              \
    do {                                                    \
        if ((n) <= 2 && (m) <= 2) { CALL(2, 2); }           \
        else if ((n) <= 4 && (m) <= 2) { CALL(4, 2); }      \
        else if ((n) <= 6 && (m) <= 4) { CALL(6, 4); }      \
        else { CALL(9, 4); }                                \
    } while (0)

extern "C" {

int fk_kf_steadystate_f64(const fk_kf_desc *d, const double *F, const double *H, const double *K, const double *B,
                          const double *u, const double *z, const uint8_t *mask, double *x, double *means,
                          double *means_p, double *y_out, void *stream)
{
    if (!d) return fail(FK_ERR_BAD_ARG, "desc is NULL");
    if (d->n < 1 || d->n > 16 || d->m < 1 || d->m > 8 || d->nu < 0 || d->nu > 4)
        return fail(FK_ERR_UNSUPPORTED, "steady state: dim_x 1..16, dim_z 1..8, dim_u 0..4");
    if (d->layout != FK_LAYOUT_AOS && d->layout != FK_LAYOUT_SOA) return fail(FK_ERR_BAD_ARG, "steady state: bad layout");
    if (d->model_mode != FK_MODEL_SHARED && d->model_mode != FK_MODEL_PER_TRACK)
        return fail(FK_ERR_UNSUPPORTED, "steady state: K is shared or per track");
    if (d->N < 0 || d->T < 0 || !x || (!F && !z) || (z && (!H || !K)) || (d->nu > 0 && F && (!B || !u)))
        return fail(FK_ERR_BAD_ARG, "steady state: bad argument");
    if ((double)d->N * d->n * d->m * 8.0 >= 4294967296.0) return fail(FK_ERR_UNSUPPORTED, "steady state: record block >= 4 GiB");
    if (d->N == 0 || d->T == 0) return FK_OK;
    SteadyArgs a{};
    a.F = F; a.H = H; a.K = K; a.B = (d->nu > 0 && F) ? B : nullptr; a.u = u; a.z = z; a.mask = mask;
    a.x = x; a.means = means; a.means_p = means_p; a.y_out = y_out; a.N = d->N; a.T = d->T;
    a.n = d->n; a.m = d->m; a.nu = d->nu; a.k_per_track = d->model_mode == FK_MODEL_PER_TRACK && K != nullptr;
    const dim3 grid((unsigned)((a.N + BLOCK - 1) / BLOCK)), block(BLOCK);
    hipStream_t s = (hipStream_t)stream;
    if (d->n > 9 || d->m > 4) {
        // the padded classes (12,8), (16,8).  Round 4 ran them in the rolled unit (x and K in scratch: 0.12 / 0.08 of HBM at (16,8));
        // round 5: unrolled here -- only x moves, 512 multiply-adds per step at (16,8) -- with a shared gain read from LDS (KREG =
        // false), a per-track gain in registers.  FK_STEADY_ROLLED=1 keeps the rolled unit (A/B).
        static const bool rolled = [] { const char *v = getenv("FK_STEADY_ROLLED"); return v && v[0] == '1'; }();
        if (rolled) return steady_big_launch(a, d->layout, s);
#define CALLB(NXV)                                                                                                           \
        if (d->n == NXV && d->m == 8 && !a.k_per_track) {       /* the class's own shape: unguarded records, NumPy order through the LDS tiles */ \
            if (d->layout == FK_LAYOUT_SOA) hipLaunchKernelGGL((steady_kernel<NXV, 8, LAYOUT_SOA, true, false>), grid, block, 0, s, a); \
            else hipLaunchKernelGGL((steady_kernel<NXV, 8, LAYOUT_AOS, true, false>), grid, block, 0, s, a);                            \
        } else if (a.k_per_track) {                                                                                                     \
            if (d->layout == FK_LAYOUT_SOA) hipLaunchKernelGGL((steady_kernel<NXV, 8, LAYOUT_SOA, false, true>), grid, block, 0, s, a); \
            else hipLaunchKernelGGL((steady_kernel<NXV, 8, LAYOUT_AOS, false, true>), grid, block, 0, s, a);                            \
        } else if (d->layout == FK_LAYOUT_SOA) hipLaunchKernelGGL((steady_kernel<NXV, 8, LAYOUT_SOA, false, false>), grid, block, 0, s, a); \
        else if (aos_waves == 3) hipLaunchKernelGGL((steady_kernel<NXV, 8, LAYOUT_AOS, false, false, 3>), grid, block, 0, s, a);          \
        else hipLaunchKernelGGL((steady_kernel<NXV, 8, LAYOUT_AOS, false, false, 1>), grid, block, 0, s, a)
        // (the guarded NumPy-order records of the padded class are what costs registers: 512 VGPRs + 724 B of scratch at one wave per
        // SIMD, 168 + 2260 B at three; FK_STEADY_AOS_WAVES=1 / 3 picks, A/B in profiles/r05/dims/steady_big.jsonl)
        static const int aos_waves = [] { const char *v = getenv("FK_STEADY_AOS_WAVES"); return v && v[0] == '3' ? 3 : 1; }();
        if (d->n <= 12) { CALLB(12); }
        else { CALLB(16); }
#undef CALLB
        return check_launch("steady_kernel");
    }
#define CALL(NXV, NZV)                                                                                      \
    if (d->n == NXV && d->m == NZV) {                                                                                    \
        if (d->layout == FK_LAYOUT_SOA) hipLaunchKernelGGL((steady_kernel<NXV, NZV, LAYOUT_SOA, true>), grid, block, 0, s, a); \
        else hipLaunchKernelGGL((steady_kernel<NXV, NZV, LAYOUT_AOS, true>), grid, block, 0, s, a);                       \
    } else if (d->layout == FK_LAYOUT_SOA) hipLaunchKernelGGL((steady_kernel<NXV, NZV, LAYOUT_SOA, false>), grid, block, 0, s, a); \
    else hipLaunchKernelGGL((steady_kernel<NXV, NZV, LAYOUT_AOS, false>), grid, block, 0, s, a)
    // exact instantiations for the shapes of BASELINE / the benches, the padded size classes for everything else
    if (d->n == 2 && d->m == 1) { CALL(2, 1); }
    else if (d->n == 6 && d->m == 3) { CALL(6, 3); }
    else if (d->n == 9 && d->m == 3) { CALL(9, 3); }
    else FK_BY_DIMS(d->n, d->m, CALL);
#undef CALL
    return check_launch("steady_kernel");
}

int fk_kf_update_correlated_f64(const fk_kf_desc *d, const double *H, const double *R, const double *M,
                                const double *z, const uint8_t *mask, double *x, double *P, double *y, double *K,
                                double *S, double *SI, int32_t *status, void *stream)
{
    if (!d) return fail(FK_ERR_BAD_ARG, "desc is NULL");
    if (d->n < 1 || d->n > 16 || d->m < 1 || d->m > 8) return fail(FK_ERR_UNSUPPORTED, "update_correlated: dim_x 1..16, dim_z 1..8");
    if (d->layout != FK_LAYOUT_AOS && d->layout != FK_LAYOUT_SOA) return fail(FK_ERR_BAD_ARG, "update_correlated: bad layout");
    if (d->model_mode != FK_MODEL_SHARED && d->model_mode != FK_MODEL_PER_TRACK)
        return fail(FK_ERR_UNSUPPORTED, "update_correlated: M is shared or per track");
    if (d->N < 0 || !H || !R || !M || !z || !x || !P) return fail(FK_ERR_BAD_ARG, "update_correlated: bad argument");
    if ((double)d->N * d->n * d->n * 8.0 >= 4294967296.0) return fail(FK_ERR_UNSUPPORTED, "update_correlated: record block >= 4 GiB");
    if (d->N == 0) return FK_OK;
    const dim3 grid((unsigned)((d->N + BLOCK - 1) / BLOCK)), block(BLOCK);
    hipStream_t s = (hipStream_t)stream;
    const int per_track = d->model_mode == FK_MODEL_PER_TRACK;
    if (d->n > 9 || d->m > 4)                                                     // padded classes (12,8), (16,8): rolled unit
        return corr_big_launch(d->n, d->m, (long)d->N, d->layout, H, R, M, per_track, z, mask, x, P, y, K, S, SI, status, s);
#define CALL(NXV, NZV)                                                                                          \
    if (d->layout == FK_LAYOUT_SOA)                                                                             \
        hipLaunchKernelGGL((corr_update_kernel<NXV, NZV, LAYOUT_SOA>), grid, block, 0, s, d->n, d->m, (long)d->N, \
                           H, R, M, per_track, z, mask, x, P, y, K, S, SI, status);                              \
    else                                                                                                        \
        hipLaunchKernelGGL((corr_update_kernel<NXV, NZV, LAYOUT_AOS>), grid, block, 0, s, d->n, d->m, (long)d->N, \
                           H, R, M, per_track, z, mask, x, P, y, K, S, SI, status)
    FK_BY_DIMS(d->n, d->m, CALL);
#undef CALL
    return check_launch("corr_update_kernel");
}

int fk_ukf_rts_correct_f64(int32_t n, int64_t N, int32_t layout, const double *Pxb, const double *xb,
                           const double *Pb, const double *xn, const double *Pn, double *x, double *P, double *K,
                           int32_t *status, void *stream)
{
    if (n < 1 || n > 16) return fail(FK_ERR_UNSUPPORTED, "ukf rts: dim_x 1..16");
    if (layout != FK_LAYOUT_AOS && layout != FK_LAYOUT_SOA) return fail(FK_ERR_BAD_ARG, "ukf rts: bad layout");
    if (N < 0 || !Pxb || !Pb || !xn || !Pn || !x || !P) return fail(FK_ERR_BAD_ARG, "ukf rts: bad argument");
    if ((double)N * n * n * 8.0 >= 4294967296.0) return fail(FK_ERR_UNSUPPORTED, "ukf rts: record block >= 4 GiB");
    if (N == 0) return FK_OK;
    const dim3 grid((unsigned)((N + BLOCK - 1) / BLOCK)), block(BLOCK);
    hipStream_t s = (hipStream_t)stream;
    if (n > 9) return ukf_rts_big_launch(n, (long)N, layout, Pxb, xb, Pb, xn, Pn, x, P, K, status, s);
#define CALL(NXV, NZV)                                                                                       \
    if (layout == FK_LAYOUT_SOA)                                                                             \
        hipLaunchKernelGGL((ukf_rts_kernel<NXV, LAYOUT_SOA>), grid, block, 0, s, n, (long)N, Pxb, xb, Pb, xn, \
                           Pn, x, P, K, status);                                                             \
    else                                                                                                     \
        hipLaunchKernelGGL((ukf_rts_kernel<NXV, LAYOUT_AOS>), grid, block, 0, s, n, (long)N, Pxb, xb, Pb, xn, \
                           Pn, x, P, K, status)
    FK_BY_DIMS(n, 1, CALL);
#undef CALL
    return check_launch("ukf_rts_kernel");
}

}  // extern "C"
#endif   // FK_VARIANTS_BIG
