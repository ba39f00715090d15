// ut_kernels.hip -- sigma-point / unscented-transform kernels for gfx950 (MI355X).
//
// One track per lane.  Stand-alone kernels mirror the reference's building blocks so
// that a UKF with arbitrary (host-side, vectorised) fx/hx can be assembled from them:
//   fk_ut_sigma_points_f64   <- MerweScaledSigmaPoints.sigma_points (sigma_points.py:124-177)
//                               JulierSigmaPoints.sigma_points      (sigma_points.py:289-357)
//   fk_ut_transform_f64      <- unscented_transform (unscented_transform.py:99-128)
//   fk_ut_cross_variance_f64 <- UnscentedKalmanFilter.cross_variance (UKF.py:493-504)
// (the fused linear-model filter lives in ukf_kernels.hip)
// Algorithmic bytes: sigma points 8(n + n^2 + (2n+1)n), UT 8((2n+1)n + n + n^2) per track;
// fused step 8(m + n + n^2) per track-step.
#include <stdlib.h>

#include "../../include/filterhip.h"
#include "fk_device.hpp"
#include "fk_kernel_args.hpp"
#include "fk_math_sym.hpp"

// FK_UT_PART: the Makefile compiles this file twice (1: sigma points + transform, 2: cross variance +
// correction) so that the two halves build in parallel; kernels unused by a half are never instantiated.
#ifndef FK_UT_PART
#define FK_UT_PART 0
#endif

namespace fk {

// ------------------------------------------------------------ sigma points --
// PAIRS (NumPy order, n == NX even): x, P and the points move as 16-byte pairs -- two consecutive components of one
// point are adjacent in the record -- half as many memory operations as the guarded 8-byte accesses (AOS measured
// 0.20-0.24 of HBM with those).
template <int NX, int LAYOUT, bool PAIRS = false>
__global__ void __launch_bounds__(BLOCK)
sigma_kernel(int n, long N, double scale, const double *__restrict__ px, const double *__restrict__ pP,
             double *sig, int32_t *status)
{
    const long blk0 = (long)blockIdx.x * BLOCK;
    const Lane ln{blk0, threadIdx.x, N};
    if (blk0 + ln.tid >= N) return;
    double x[NX], P[NX * NX], L[NX * NX];
    load_rec<NX, 1, LAYOUT, PAIRS>(x, px, ln, n, 1, 0.0);
    load_rec<NX, NX, LAYOUT, PAIRS>(P, pP, ln, n, n, 1.0 / scale);   // padded block -> L = I
    const bool pd = chol_lower<NX>(P, scale, L);
    const RecView<LAYOUT> out(sig, ln, (2 * n + 1) * n);
    if constexpr (PAIRS) {
        static_assert(NX % 2 == 0 && LAYOUT == LAYOUT_AOS, "pairs: NumPy order, even dim_x");
        FK_UNROLL for (int c = 0; c < NX; c += 2) out.store2(c, x[c], x[c + 1]);
        FK_UNROLL for (int k = 0; k < NX; ++k)
            FK_UNROLL for (int c = 0; c < NX; c += 2) {
                out.store2((k + 1) * NX + c, x[c] - (-L[c * NX + k]), x[c + 1] - (-L[(c + 1) * NX + k]));
                out.store2((NX + k + 1) * NX + c, x[c] - L[c * NX + k], x[c + 1] - L[(c + 1) * NX + k]);
            }
        if (status) status[blk0 + ln.tid] = pd ? 0 : ST_NOT_PD;
        return;
    }
    // sigma_0 = x ; sigma_{k+1} = x + U[k] ; sigma_{n+k+1} = x - U[k]   (U[k][c] = L[c][k])
    FK_UNROLL for (int c = 0; c < NX; ++c)
        if (c < n) out.store(c, x[c]);
    FK_UNROLL for (int k = 0; k < NX; ++k) {
        if (k < n) {
            FK_UNROLL for (int c = 0; c < NX; ++c) {
                if (c < n) {
                    // np.subtract(x, -U[k]) and np.subtract(x, U[k])  (sigma_points.py:174-175)
                    out.store((k + 1) * n + c, x[c] - (-L[c * NX + k]));
                    out.store((n + k + 1) * n + c, x[c] - L[c * NX + k]);
                }
            }
        }
    }
    if (status) status[blk0 + ln.tid] = pd ? 0 : ST_NOT_PD;
}

// Register-resident variant for the standard point sets (k == 2n+1, n == NX <= 6): every sigma
// point is read from HBM exactly once (the generic kernel below re-reads them for the second pass).
template <int NX, int LAYOUT>
__global__ void __launch_bounds__(BLOCK, (NX <= 4 ? 4 : 1))
ut_reg_kernel(long N, const double *__restrict__ sig, const double *__restrict__ Wm,
              const double *__restrict__ Wc, const double *__restrict__ noise, double *xo, double *Po)
{
    constexpr int K = 2 * NX + 1;
    const long blk0 = (long)blockIdx.x * BLOCK;
    const Lane ln{blk0, threadIdx.x, N};
    if (blk0 + ln.tid >= N) return;
    double s[K * NX];
    load_rec<K, NX, LAYOUT, true>(s, sig, ln, K, NX, 0.0);
    double x[NX];
    FK_UNROLL for (int c = 0; c < NX; ++c) {
        double acc = Wm[0] * s[c];
        FK_UNROLL for (int i = 1; i < K; ++i) acc = fma(Wm[i], s[i * NX + c], acc);
        x[c] = acc;
    }
    FK_UNROLL for (int i = 0; i < K; ++i)
        FK_UNROLL for (int c = 0; c < NX; ++c) s[i * NX + c] -= x[c];
    store_rec<NX, 1, LAYOUT, true>(x, xo, ln, NX, 1);
    // P = sum_i y_i (Wc_i y_i)': accumulate the upper triangle point by point (P is symmetric by
    // construction: the same two factors commute), mirror on store
    double U[NX * (NX + 1) / 2];
    FK_UNROLL for (int i = 0; i < K; ++i) {
        double wy[NX];
        FK_UNROLL for (int c = 0; c < NX; ++c) wy[c] = Wc[i] * s[i * NX + c];
        int t = 0;
        FK_UNROLL for (int a = 0; a < NX; ++a)
            FK_UNROLL for (int b2 = a; b2 < NX; ++b2, ++t)
                U[t] = (i == 0) ? s[a] * wy[b2] : fma(s[i * NX + a], wy[b2], U[t]);
        FK_STAGE();
    }
    const RecView<LAYOUT> pv(Po, ln, NX * NX);
    int t = 0;
    FK_UNROLL for (int a = 0; a < NX; ++a)
        FK_UNROLL for (int b2 = a; b2 < NX; ++b2, ++t) {
            const double v = noise ? U[t] + noise[a * NX + b2] : U[t];
            pv.store(a * NX + b2, v);
            if (b2 != a) pv.store(b2 * NX + a, noise ? U[t] + noise[b2 * NX + a] : U[t]);
        }
}

// ---- NumPy order, exact dims 2 / 4 / 6: one WAVE per workgroup, records through a wave-private LDS tile -------------------
// In NumPy order (AOS) a track's record is contiguous -- 78 doubles of sigma points at n = 6 -- and a lane-per-record access
// touches 64 different cache lines per instruction: the round-2 kernels measured 0.32-0.43 of HBM there against 0.6-0.86
// in the element-major layout.  Here the sigma points (and x, P) move like kf_fast's outputs: 1 KiB contiguous per
// instruction between HBM and a tile, row-per-lane between the tile and the registers (wave_load_aos / wave_store_aos,
// fk_device.hpp).  The tile is 64 x (LEN | 1) doubles -- 40 KiB at n = 6 --, hence one wave per workgroup.
constexpr int UT_WAVE = 64;
template <int NX>
__global__ void __launch_bounds__(UT_WAVE)
sigma_coop_kernel(long N, double scale, const double *__restrict__ px, const double *__restrict__ pP, double *sig,
                  int32_t *status)
{
    constexpr int K = 2 * NX + 1, LEN = K * NX;
    __shared__ double tile[UT_WAVE * (LEN | 1)];
    const long blk0 = (long)blockIdx.x * UT_WAVE;
    const unsigned lane = threadIdx.x;
    const long left = N - blk0;
    const unsigned last_row = (unsigned)(left < UT_WAVE ? left : UT_WAVE) - 1u;
    double x[NX], P[NX * NX], L[NX * NX];
    if constexpr (NX >= 2) {
        wave_load_aos<NX>(x, px + blk0 * NX, 0u, tile, lane, last_row);
        wave_load_aos<NX * NX>(P, pP + blk0 * NX * NX, 0u, tile, lane, last_row);
    }
    if (lane > last_row) {                                                 // rows past the end read as zeros: keep the Cholesky sane
        FK_UNROLL for (int e = 0; e < NX * NX; ++e) P[e] = (e / NX == e % NX) ? 1.0 : 0.0;
    }
    const bool pd = chol_lower<NX>(P, scale, L);
    double s[LEN];
    FK_UNROLL for (int c = 0; c < NX; ++c) s[c] = x[c];
    FK_UNROLL for (int k = 0; k < NX; ++k)
        FK_UNROLL for (int c = 0; c < NX; ++c) {
            // np.subtract(x, -U[k]) and np.subtract(x, U[k])  (sigma_points.py:174-175); U[k][c] = L[c][k]
            s[(k + 1) * NX + c] = x[c] - (-L[c * NX + k]);
            s[(NX + k + 1) * NX + c] = x[c] - L[c * NX + k];
        }
    wave_store_aos<LEN>(s, sig + blk0 * LEN, 0u, tile, lane, last_row);
    if (status && lane <= last_row) status[blk0 + lane] = pd ? 0 : ST_NOT_PD;
}

template <int NX>
__global__ void __launch_bounds__(UT_WAVE)
ut_coop_kernel(long N, const double *__restrict__ sig, const double *__restrict__ Wm, const double *__restrict__ Wc,
               const double *__restrict__ noise, double *xo, double *Po)
{
    constexpr int K = 2 * NX + 1, LEN = K * NX;
    __shared__ double tile[UT_WAVE * (LEN | 1)];
    const long blk0 = (long)blockIdx.x * UT_WAVE;
    const unsigned lane = threadIdx.x;
    const long left = N - blk0;
    const unsigned last_row = (unsigned)(left < UT_WAVE ? left : UT_WAVE) - 1u;
    double s[LEN];
    wave_load_aos<LEN>(s, sig + blk0 * LEN, 0u, tile, lane, last_row);
    // the arithmetic of ut_reg_kernel: x = Wm . sigmas; P = sum_i y_i (Wc_i y_i)' (+ noise), points in index order
    double x[NX];
    FK_UNROLL for (int c = 0; c < NX; ++c) {
        double acc = Wm[0] * s[c];
        FK_UNROLL for (int i = 1; i < K; ++i) acc = fma(Wm[i], s[i * NX + c], acc);
        x[c] = acc;
    }
    FK_UNROLL for (int i = 0; i < K; ++i)
        FK_UNROLL for (int c = 0; c < NX; ++c) s[i * NX + c] -= x[c];
    double U[NX * (NX + 1) / 2];
    FK_UNROLL for (int i = 0; i < K; ++i) {
        double wy[NX];
        FK_UNROLL for (int c = 0; c < NX; ++c) wy[c] = Wc[i] * s[i * NX + c];
        int t = 0;
        FK_UNROLL for (int a = 0; a < NX; ++a)
            FK_UNROLL for (int b2 = a; b2 < NX; ++b2, ++t)
                U[t] = (i == 0) ? s[a] * wy[b2] : fma(s[i * NX + a], wy[b2], U[t]);
        FK_STAGE();
    }
    double Pf[NX * NX];
    int t = 0;
    FK_UNROLL for (int a = 0; a < NX; ++a)
        FK_UNROLL for (int b2 = a; b2 < NX; ++b2, ++t) {
            Pf[a * NX + b2] = noise ? U[t] + noise[a * NX + b2] : U[t];
            if (b2 != a) Pf[b2 * NX + a] = noise ? U[t] + noise[b2 * NX + a] : U[t];
        }
    wave_store_aos<NX>(x, xo + blk0 * NX, 0u, tile, lane, last_row);
    wave_store_aos<NX * NX>(Pf, Po + blk0 * NX * NX, 0u, tile, lane, last_row);
}

// ------------------------------------------------------ unscented transform --
template <int NX, int LAYOUT>
__global__ void __launch_bounds__(BLOCK)
ut_kernel(int n, int k, long N, const double *__restrict__ sig, const double *__restrict__ Wm,
          const double *__restrict__ Wc, const double *__restrict__ noise, double *xo, double *Po)
{
    const long blk0 = (long)blockIdx.x * BLOCK;
    const Lane ln{blk0, threadIdx.x, N};
    if (blk0 + ln.tid >= N) return;
    const RecView<LAYOUT> sv(sig, ln, k * n);
    // x = Wm . sigmas   (unscented_transform.py:104)
    double x[NX];
    FK_UNROLL for (int c = 0; c < NX; ++c) x[c] = 0.0;
    for (int i = 0; i < k; ++i) {
        const double w = Wm[i];
        FK_UNROLL for (int c = 0; c < NX; ++c)
            if (c < n) x[c] = (i == 0) ? w * sv.load(c) : fma(w, sv.load(i * n + c), x[c]);
    }
    // P = y' (diag(Wc) y)   (unscented_transform.py:117-118)
    double P[NX * NX];
    FK_UNROLL for (int e = 0; e < NX * NX; ++e) P[e] = 0.0;
    for (int i = 0; i < k; ++i) {
        const double w = Wc[i];
        double y[NX];
        FK_UNROLL for (int c = 0; c < NX; ++c) y[c] = (c < n) ? sv.load(i * n + c) - x[c] : 0.0;
        FK_UNROLL for (int a = 0; a < NX; ++a)
            FK_UNROLL for (int b = 0; b < NX; ++b) P[a * NX + b] = fma(y[a], w * y[b], P[a * NX + b]);
    }
    if (noise) {
        FK_UNROLL for (int a = 0; a < NX; ++a)
            FK_UNROLL for (int b = 0; b < NX; ++b)
                if (a < n && b < n) P[a * NX + b] += noise[a * n + b];
    }
    store_rec<NX, 1, LAYOUT, false>(x, xo, ln, n, 1);
    store_rec<NX, NX, LAYOUT, false>(P, Po, ln, n, n);
}

// ---------------------------------------------------------- cross variance --
template <int NX, int LAYOUT>
__global__ void __launch_bounds__(BLOCK)
cross_kernel(int n, int m, int k, long N, const double *__restrict__ px, const double *__restrict__ pz,
             const double *__restrict__ sf, const double *__restrict__ sh, const double *__restrict__ Wc,
             double *Pxz)
{
    const long blk0 = (long)blockIdx.x * BLOCK;
    const Lane ln{blk0, threadIdx.x, N};
    if (blk0 + ln.tid >= N) return;
    const RecView<LAYOUT> fv(sf, ln, k * n), hv(sh, ln, k * m), zv(pz, ln, m), ov(Pxz, ln, n * m);
    // px / pz == nullptr: the caller already applied its own residual_x / residual_z (UKF.py:500-501 with custom
    // callables), sf / sh hold dx / dz; v - 0.0 is v exactly
    double x[NX];
    if (px) load_rec<NX, 1, LAYOUT, false>(x, px, ln, n, 1, 0.0);
    else { FK_UNROLL for (int a = 0; a < NX; ++a) x[a] = 0.0; }
    for (int c = 0; c < m; ++c) {
        const double zc = pz ? zv.load(c) : 0.0;
        double acc[NX];
        FK_UNROLL for (int a = 0; a < NX; ++a) acc[a] = 0.0;
        for (int i = 0; i < k; ++i) {
            // Pxz += Wc[i] * outer(dx, dz)   (UKF.py:500-503)
            const double dz = hv.load(i * m + c) - zc;
            const double w = Wc[i];
            FK_UNROLL for (int a = 0; a < NX; ++a)
                if (a < n) acc[a] += w * ((fv.load(i * n + a) - x[a]) * dz);
        }
        FK_UNROLL for (int a = 0; a < NX; ++a)
            if (a < n) ov.store(a * m + c, acc[a]);
    }
}

// ------------------------------------------------------------ linear map --
// A linear fx / hx handed over as a matrix (UnscentedKalmanFilter(fx=F, hx=H)), applied to every sigma point of every
// track where the fused kernels do not reach (dim_x > 9, per-epoch Rs / dts, hooks): out[i] = M in[i], the products
// accumulated over the columns in order (what numpy.dot of a matrix and a vector does in the reference's lambda,
// UKF.py:521-522 / :462-466).  One track per lane, M in LDS.  (Round 2 called torch.matmul -- rocBLAS -- here.)
template <int NX, int LAYOUT>
__global__ void __launch_bounds__(BLOCK)
linear_map_kernel(int n_in, int n_out, int k, long N, const double *__restrict__ pM, const double *__restrict__ in,
                  double *__restrict__ out)
{
    __shared__ double sM[16 * 16];
    for (int q = threadIdx.x; q < n_out * n_in; q += BLOCK) sM[q] = pM[q];
    __syncthreads();
    const long blk0 = (long)blockIdx.x * BLOCK;
    const Lane ln{blk0, threadIdx.x, N};
    if (blk0 + ln.tid >= N) return;
    const RecView<LAYOUT> iv(in, ln, k * n_in), ov(out, ln, k * n_out);
    for (int i = 0; i < k; ++i) {
        double x[NX];
        FK_UNROLL for (int c = 0; c < NX; ++c) x[c] = c < n_in ? iv.load(i * n_in + c) : 0.0;
        for (int r = 0; r < n_out; ++r) {
            const double *row = sM + r * n_in;
            double acc = row[0] * x[0];
            FK_UNROLL for (int c = 1; c < NX; ++c)
                if (c < n_in) acc = fma(row[c], x[c], acc);
            ov.store(i * n_out + r, acc);
        }
    }
}

// ------------------------------------------------------------- UKF correct --
// The tail of UnscentedKalmanFilter.update (UKF.py:470-481) for arbitrary hx:
//   K = Pxz S^-1 ; x += K (z - zp) ; P -= K (S K')
template <int NX, int NZ, int LAYOUT>
__global__ void __launch_bounds__(BLOCK)
ukf_correct_kernel(int n, int m, long N, const double *__restrict__ pPxz, const double *__restrict__ pzp,
                   const double *__restrict__ pS, const double *__restrict__ pz, double *px, double *pP,
                   double *pK, int32_t *status)
{
    const long blk0 = (long)blockIdx.x * BLOCK;
    const Lane ln{blk0, threadIdx.x, N};
    if (blk0 + ln.tid >= N) return;
    double x[NX], P[NX * NX], K[NX * NZ], S[NZ * NZ], z[NZ], zp[NZ];
    load_rec<NX, 1, LAYOUT, false>(x, px, ln, n, 1, 0.0);
    load_rec<NX, NX, LAYOUT, false>(P, pP, ln, n, n, 0.0);
    load_rec<NX, NZ, LAYOUT, false>(K, pPxz, ln, n, m, 0.0);
    load_rec<NZ, NZ, LAYOUT, false>(S, pS, ln, m, m, 1.0);
    load_rec<NZ, 1, LAYOUT, false>(z, pz, ln, m, 1, 0.0);
    if (pzp) load_rec<NZ, 1, LAYOUT, false>(zp, pzp, ln, m, 1, 0.0);      // nullptr: z IS residual_z(z, zp) (UKF.py:474)
    else { FK_UNROLL for (int c = 0; c < NZ; ++c) zp[c] = 0.0; }
    // the padded diagonal of K (load_rec pads a==b with diag_pad=0) is zero: nothing to undo
    double Lf[NZ * NZ], d[NZ], dinv[NZ];
    FK_UNROLL for (int e = 0; e < NZ * NZ; ++e) Lf[e] = S[e];
    int st = 0;
    if (!ldlt2<NZ>(Lf, d, dinv)) st |= ST_NOT_PD;
    solve_rows_ldlt<NX, NZ>(Lf, dinv, K);
    FK_UNROLL for (int r = 0; r < NX; ++r) {
        double acc = K[r * NZ] * (z[0] - zp[0]);
        FK_UNROLL for (int c = 1; c < NZ; ++c) acc = fma(K[r * NZ + c], z[c] - zp[c], acc);
        x[r] += acc;
    }
    double SK[NZ * NX];
    FK_UNROLL for (int c = 0; c < NZ; ++c)
        FK_UNROLL for (int r = 0; r < NX; ++r) {
            double acc = S[c * NZ] * K[r * NZ];
            FK_UNROLL for (int q = 1; q < NZ; ++q) acc = fma(S[c * NZ + q], K[r * NZ + q], acc);
            SK[c * NX + r] = acc;
        }
    FK_UNROLL for (int r = 0; r < NX; ++r)
        FK_UNROLL for (int c = 0; c < NX; ++c) {
            double acc = K[r * NZ] * SK[c];
            FK_UNROLL for (int q = 1; q < NZ; ++q) acc = fma(K[r * NZ + q], SK[q * NX + c], acc);
            P[r * NX + c] -= acc;
        }
    store_rec<NX, 1, LAYOUT, false>(x, px, ln, n, 1);
    store_rec<NX, NX, LAYOUT, false>(P, pP, ln, n, n);
    if (pK) store_rec<NX, NZ, LAYOUT, false>(K, pK, ln, n, m);
    if (status) {
        if (!all_finite<NX>(x) || !all_finite<NX * NX>(P)) st |= ST_NONFINITE;
        status[blk0 + ln.tid] = st;
    }
}

static int fail(int code, const char *msg)
{
    set_last_error(msg);
    return code;
}

}  // namespace fk

using namespace fk;

#define FK_BY_NX(n, CALL)            \
    do {                             \
        if ((n) <= 2) { CALL(2); }   \
        else if ((n) <= 4) { CALL(4); } \
        else if ((n) <= 6) { CALL(6); } \
        else if ((n) <= 8) { CALL(8); } \
        else if ((n) <= 12) { CALL(12); } \
        else { CALL(16); }           \
    } while (0)

extern "C" {

#if FK_UT_PART != 2
int fk_ut_sigma_points_f64(int32_t n, int64_t N, int32_t layout, double scale, const double *x,
                           const double *P, double *sigmas, int32_t *status, void *stream)
{
    if (n < 1 || n > 16) return fail(FK_ERR_UNSUPPORTED, "sigma points: dim_x must be 1..16");
    if (N < 0 || !x || !P || !sigmas) return fail(FK_ERR_BAD_ARG, "sigma points: bad argument");
    if ((double)N * (2 * n + 1) * n * 8.0 >= 4294967296.0) return fail(FK_ERR_UNSUPPORTED, "sigma points: record block >= 4 GiB, split the batch");
    if (N == 0) return FK_OK;
    const dim3 grid((unsigned)((N + BLOCK - 1) / BLOCK)), block(BLOCK);
    // NumPy order at the exact dims 2 / 4 / 6: the wave-cooperative kernel (FK_UT_COOP=0: the per-lane 16-byte pairs of round 2)
    static const bool coop = [] { const char *v = getenv("FK_UT_COOP"); return !(v && v[0] == '0'); }();
    if (coop && layout == FK_LAYOUT_AOS && (n == 2 || n == 4 || n == 6)) {
        const dim3 g1((unsigned)((N + UT_WAVE - 1) / UT_WAVE)), b1(UT_WAVE);
        if (n == 2) hipLaunchKernelGGL((sigma_coop_kernel<2>), g1, b1, 0, (hipStream_t)stream, (long)N, scale, x, P, sigmas, status);
        else if (n == 4) hipLaunchKernelGGL((sigma_coop_kernel<4>), g1, b1, 0, (hipStream_t)stream, (long)N, scale, x, P, sigmas, status);
        else hipLaunchKernelGGL((sigma_coop_kernel<6>), g1, b1, 0, (hipStream_t)stream, (long)N, scale, x, P, sigmas, status);
        return check_launch("sigma_coop_kernel");
    }
#define CALL(NXV)                                                                                      \
    if (layout == FK_LAYOUT_SOA)                                                                       \
        hipLaunchKernelGGL((sigma_kernel<NXV, LAYOUT_SOA>), grid, block, 0, (hipStream_t)stream, n, N, \
                           scale, x, P, sigmas, status);                                               \
    else if (n == NXV && NXV <= 8)                                                                     \
        hipLaunchKernelGGL((sigma_kernel<NXV, LAYOUT_AOS, true>), grid, block, 0, (hipStream_t)stream, n, N, \
                           scale, x, P, sigmas, status);                                               \
    else                                                                                               \
        hipLaunchKernelGGL((sigma_kernel<NXV, LAYOUT_AOS>), grid, block, 0, (hipStream_t)stream, n, N, \
                           scale, x, P, sigmas, status)
    FK_BY_NX(n, CALL);
#undef CALL
    return check_launch("sigma_kernel");
}

int fk_ut_transform_f64(int32_t n, int32_t k, int64_t N, int32_t layout, const double *sigmas,
                        const double *Wm, const double *Wc, const double *noise_cov, double *x_out,
                        double *P_out, void *stream)
{
    if (n < 1 || n > 16 || k < 1) return fail(FK_ERR_UNSUPPORTED, "unscented transform: dim must be 1..16");
    if (N < 0 || !sigmas || !Wm || !Wc || !x_out || !P_out) return fail(FK_ERR_BAD_ARG, "unscented transform: bad argument");
    if ((double)N * k * n * 8.0 >= 4294967296.0) return fail(FK_ERR_UNSUPPORTED, "unscented transform: record block >= 4 GiB, split the batch");
    if (N == 0) return FK_OK;
    const dim3 grid((unsigned)((N + BLOCK - 1) / BLOCK)), block(BLOCK);
    static const bool coop = [] { const char *v = getenv("FK_UT_COOP"); return !(v && v[0] == '0'); }();
    if (coop && layout == FK_LAYOUT_AOS && k == 2 * n + 1 && (n == 2 || n == 4 || n == 6)) {
        const dim3 g1((unsigned)((N + UT_WAVE - 1) / UT_WAVE)), b1(UT_WAVE);
        if (n == 2) hipLaunchKernelGGL((ut_coop_kernel<2>), g1, b1, 0, (hipStream_t)stream, (long)N, sigmas, Wm, Wc, noise_cov, x_out, P_out);
        else if (n == 4) hipLaunchKernelGGL((ut_coop_kernel<4>), g1, b1, 0, (hipStream_t)stream, (long)N, sigmas, Wm, Wc, noise_cov, x_out, P_out);
        else hipLaunchKernelGGL((ut_coop_kernel<6>), g1, b1, 0, (hipStream_t)stream, (long)N, sigmas, Wm, Wc, noise_cov, x_out, P_out);
        return check_launch("ut_coop_kernel");
    }
    if (k == 2 * n + 1 && (n == 2 || n == 4 || n == 6)) {
#define REG(NXV)                                                                                          \
    if (layout == FK_LAYOUT_SOA)                                                                          \
        hipLaunchKernelGGL((ut_reg_kernel<NXV, LAYOUT_SOA>), grid, block, 0, (hipStream_t)stream, (long)N, \
                           sigmas, Wm, Wc, noise_cov, x_out, P_out);                                      \
    else                                                                                                  \
        hipLaunchKernelGGL((ut_reg_kernel<NXV, LAYOUT_AOS>), grid, block, 0, (hipStream_t)stream, (long)N, \
                           sigmas, Wm, Wc, noise_cov, x_out, P_out)
        if (n == 2) { REG(2); }
        else if (n == 4) { REG(4); }
        else { REG(6); }
#undef REG
        return check_launch("ut_reg_kernel");
    }
#define CALL(NXV)                                                                                   \
    if (layout == FK_LAYOUT_SOA)                                                                    \
        hipLaunchKernelGGL((ut_kernel<NXV, LAYOUT_SOA>), grid, block, 0, (hipStream_t)stream, n, k, \
                           N, sigmas, Wm, Wc, noise_cov, x_out, P_out);                             \
    else                                                                                            \
        hipLaunchKernelGGL((ut_kernel<NXV, LAYOUT_AOS>), grid, block, 0, (hipStream_t)stream, n, k, \
                           N, sigmas, Wm, Wc, noise_cov, x_out, P_out)
    FK_BY_NX(n, CALL);
#undef CALL
    return check_launch("ut_kernel");
}

#endif   // FK_UT_PART
#if FK_UT_PART != 1
int fk_ut_cross_variance_f64(int32_t n, int32_t m, int32_t k, int64_t N, int32_t layout,
                             const double *x, const double *z, const double *sigmas_f,
                             const double *sigmas_h, const double *Wc, double *Pxz, void *stream)
{
    if (n < 1 || n > 16 || m < 1 || k < 1) return fail(FK_ERR_UNSUPPORTED, "cross variance: dim_x must be 1..16");
    if (N < 0 || !sigmas_f || !sigmas_h || !Wc || !Pxz) return fail(FK_ERR_BAD_ARG, "cross variance: bad argument");
    if ((double)N * k * (n > m ? n : m) * 8.0 >= 4294967296.0) return fail(FK_ERR_UNSUPPORTED, "cross variance: record block >= 4 GiB, split the batch");
    if (N == 0) return FK_OK;
    const dim3 grid((unsigned)((N + BLOCK - 1) / BLOCK)), block(BLOCK);
#define CALL(NXV)                                                                                      \
    if (layout == FK_LAYOUT_SOA)                                                                       \
        hipLaunchKernelGGL((cross_kernel<NXV, LAYOUT_SOA>), grid, block, 0, (hipStream_t)stream, n, m, \
                           k, N, x, z, sigmas_f, sigmas_h, Wc, Pxz);                                   \
    else                                                                                               \
        hipLaunchKernelGGL((cross_kernel<NXV, LAYOUT_AOS>), grid, block, 0, (hipStream_t)stream, n, m, \
                           k, N, x, z, sigmas_f, sigmas_h, Wc, Pxz)
    FK_BY_NX(n, CALL);
#undef CALL
    return check_launch("cross_kernel");
}

int fk_ut_linear_map_f64(int32_t n_in, int32_t n_out, int32_t k, int64_t N, int32_t layout, const double *M,
                         const double *in, double *out, void *stream)
{
    if (n_in < 1 || n_in > 16 || n_out < 1 || n_out > 16 || k < 1) return fail(FK_ERR_UNSUPPORTED, "linear map: dims must be 1..16");
    if (N < 0 || !M || !in || !out) return fail(FK_ERR_BAD_ARG, "linear map: bad argument");
    if ((double)N * k * (n_in > n_out ? n_in : n_out) * 8.0 >= 4294967296.0) return fail(FK_ERR_UNSUPPORTED, "linear map: record block >= 4 GiB, split the batch");
    if (N == 0) return FK_OK;
    const dim3 grid((unsigned)((N + BLOCK - 1) / BLOCK)), block(BLOCK);
#define CALL(NXV)                                                                                                \
    if (layout == FK_LAYOUT_SOA)                                                                                 \
        hipLaunchKernelGGL((linear_map_kernel<NXV, LAYOUT_SOA>), grid, block, 0, (hipStream_t)stream, n_in, n_out, k, \
                           (long)N, M, in, out);                                                                 \
    else                                                                                                         \
        hipLaunchKernelGGL((linear_map_kernel<NXV, LAYOUT_AOS>), grid, block, 0, (hipStream_t)stream, n_in, n_out, k, \
                           (long)N, M, in, out)
    if (n_in <= 4) { CALL(4); }
    else if (n_in <= 8) { CALL(8); }
    else { CALL(16); }
#undef CALL
    return check_launch("linear_map_kernel");
}

int fk_ukf_correct_f64(int32_t n, int32_t m, int64_t N, int32_t layout, const double *Pxz, const double *zp,
                       const double *S, const double *z, double *x, double *P, double *K, int32_t *status,
                       void *stream)
{
    if (n < 1 || n > 16 || m < 1 || m > 8) return fail(FK_ERR_UNSUPPORTED, "ukf correct: dim_x 1..16, dim_z 1..8");
    if (N < 0 || !Pxz || !S || !z || !x || !P) return fail(FK_ERR_BAD_ARG, "ukf correct: bad argument");
    if ((double)N * n * n * 8.0 >= 4294967296.0) return fail(FK_ERR_UNSUPPORTED, "ukf correct: record block >= 4 GiB, split the batch");
    if (N == 0) return FK_OK;
    const dim3 grid((unsigned)((N + BLOCK - 1) / BLOCK)), block(BLOCK);
    hipStream_t s = (hipStream_t)stream;
#define GOZ(NXV, NZV)                                                                                      \
    if (layout == FK_LAYOUT_SOA)                                                                           \
        hipLaunchKernelGGL((ukf_correct_kernel<NXV, NZV, LAYOUT_SOA>), grid, block, 0, s, n, m, N, Pxz, zp, \
                           S, z, x, P, K, status);                                                         \
    else                                                                                                   \
        hipLaunchKernelGGL((ukf_correct_kernel<NXV, NZV, LAYOUT_AOS>), grid, block, 0, s, n, m, N, Pxz, zp, \
                           S, z, x, P, K, status)
#define CALL(NXV)                  \
    if (m <= 4) { GOZ(NXV, 4); }   \
    else { GOZ(NXV, 8); }
    FK_BY_NX(n, CALL);
#undef CALL
#undef GOZ
    return check_launch("ukf_correct_kernel");
}

#endif   // FK_UT_PART
}  // extern "C"
