// fk_math.hpp -- per-track small-matrix arithmetic of the filter hot path.
//
// Everything here works on fully unrolled, compile-time-sized local arrays so
// that on gfx950 a track's state lives in VGPRs (one track per lane) for the
// whole time loop.  The functions are __host__ __device__ so the same source is
// also compiled by the test-only host harness (tests/hostcheck) to check the
// arithmetic against the oracle in the GPU-less build container; the product
// library only ever calls them from device code.
//
// Reference arithmetic (rlabbe/filterpy v1.4.5):
//   predict : filterpy/kalman/kalman_filter.py:472-478
//   update  : filterpy/kalman/kalman_filter.py:533-556 (Joseph form)
//   rts     : filterpy/kalman/kalman_filter.py:1066-1072
//   sigma   : filterpy/kalman/sigma_points.py:167-175
//   UT      : filterpy/kalman/unscented_transform.py:104,117-118,126
//   UKF     : filterpy/kalman/UKF.py:469-481, 493-504
// The reference applies S^-1 with numpy.linalg.inv (LU); here S is factored
// in-lane as L D L' (square-root-free Cholesky) and K is obtained by two
// triangular solves -- agreement is ~cond(S)*eps, far inside the 1e-10 bar.
#pragma once

#include <math.h>

#if defined(__HIPCC__)
#define FK_HD __host__ __device__ __forceinline__
#else
#define FK_HD inline
#endif

// FK_ROLLED builds (large dims, > 9) keep the loops rolled: the per-track arrays then live in
// scratch memory and the code stays small -- slow but compiles in seconds and covers every
// dim_x <= 16; the unrolled register-resident builds are the fast path.
#if defined(FK_ROLLED) && FK_ROLLED
#define FK_UNROLL _Pragma("nounroll")
#else
#define FK_UNROLL _Pragma("unroll")
#endif

// Scheduling fence between arithmetic stages: keeps hipcc's machine scheduler from
// interleaving every independent FMA chain of the (fully unrolled) step, which
// otherwise inflates live ranges past the VGPR budget of the occupancy target.
#if defined(__HIP_DEVICE_COMPILE__)
#define FK_STAGE() __builtin_amdgcn_sched_barrier(0)
#else
#define FK_STAGE() ((void)0)
#endif

namespace fk {

enum : int { ST_NOT_PD = 1, ST_NONFINITE = 2, ST_OVERRUN = 4, ST_INTERNAL = 8, ST_BAD_WEIGHTS = 16 };

// ln |S| from the reciprocal pivots of its L D L' factorisation: -ln prod_i (1 / d_i), ONE logarithm instead of a division and a
// logarithm per pivot (a double-precision log is ~80 VALU instructions).  The product is carried as mantissa x 2^exponent
// (frexp: two instructions per pivot), so it cannot leave the exponent range and needs no fall-back branch.  m: the run-time
// number of pivots that count (padded instantiations).
template <int NZ>
FK_HD double logdet_from_dinv(const double (&dinv)[NZ], int m)
{
    double pm = 1.0;
    int pe = 0;
    FK_UNROLL for (int i = 0; i < NZ; ++i)
        if (i < m) {
            int e;
            pm *= frexp(dinv[i], &e);
            pe += e;
        }
    return -fma((double)pe, 0.6931471805599453, log(pm));
}

// sqrt(prod_i 1 / d_i) = |S|^(-1/2) from the reciprocal pivots: the normalising factor of a Gaussian density WITHOUT a logarithm
// (mantissa x 2^exponent like logdet_from_dinv; the root of the mantissa -- in [2^-NZ, 2) after evening out the exponent --
// from the v_rsq_f64 seed, one Goldschmidt step and a residual correction: 1e-16, tools/experiments/rsq_seed_accuracy.hip;
// on the host the plain sqrt).
// (rsqrt_det_parts: the root as mantissa g in [1/2, 2) and binary exponent e2, |S|^(-1/2) = g * 2^e2 -- a caller that
//  multiplies by exp(.) folds e2 into that exponent and never leaves the range; rsqrt_det_from_dinv: the product itself)
template <int NZ>
FK_HD double rsqrt_det_parts(const double (&dinv)[NZ], int m, int &e2)
{
    double pm = 1.0;
    int pe = 0;
    FK_UNROLL for (int i = 0; i < NZ; ++i)
        if (i < m) {
            int e;
            pm *= frexp(dinv[i], &e);
            pe += e;
        }
    const int odd = pe & 1;
    pm = odd ? pm + pm : pm;
    pe -= odd;
    e2 = pe / 2;
#if defined(__HIP_DEVICE_COMPILE__)
    const double y = __builtin_amdgcn_rsq(pm);
    double g = pm * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    g = fma(fma(-g, g, pm), h, g);
#else
    const double g = sqrt(pm);
#endif
    return g;
}

template <int NZ>
FK_HD double rsqrt_det_from_dinv(const double (&dinv)[NZ], int m)
{
    int e2;
    const double g = rsqrt_det_parts<NZ>(dinv, m, e2);
    return ldexp(g, e2);
}

// 1 / d for the arithmetic-bound kernels: the v_rcp_f64 seed (2^-24, tools/experiments/rsq_seed_accuracy.hip) and two Newton
// steps -- 1.1e-16 relative, five instructions; the compiler's IEEE division is ~25 with its scaling and fix-up.  No range
// scaling: the callers divide by sums of probabilities and pivots of covariances, nowhere near the ends of the exponent range
// (a zero or non-finite divisor yields inf / NaN like the division would).  On the host: the plain division.
FK_HD double fk_rcp(double d)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(d);
    double e = fma(-d, r, 1.0);
    r = fma(r, e, r);
    e = fma(-d, r, 1.0);
    return fma(r, e, r);
#else
    return 1.0 / d;
#endif
}

// C[R x C] = A[R x K] * B[K x C]
template <int R, int K, int C>
FK_HD void matmul(const double (&A)[R * K], const double (&B)[K * C], double (&Cm)[R * C])
{
    FK_UNROLL for (int i = 0; i < R; ++i) {
        FK_UNROLL for (int j = 0; j < C; ++j) {
            double acc = A[i * K] * B[j];
            FK_UNROLL for (int k = 1; k < K; ++k) acc = fma(A[i * K + k], B[k * C + j], acc);
            Cm[i * C + j] = acc;
        }
    }
}

// C[R x C] = A[R x K] * B'[K x C]  with B given as [C x K]
template <int R, int K, int C>
FK_HD void matmul_nt(const double (&A)[R * K], const double (&B)[C * K], double (&Cm)[R * C])
{
    FK_UNROLL for (int i = 0; i < R; ++i) {
        FK_UNROLL for (int j = 0; j < C; ++j) {
            double acc = A[i * K] * B[j * K];
            FK_UNROLL for (int k = 1; k < K; ++k) acc = fma(A[i * K + k], B[j * K + k], acc);
            Cm[i * C + j] = acc;
        }
    }
}

template <int R, int K>
FK_HD void matvec(const double (&A)[R * K], const double (&v)[K], double (&out)[R])
{
    FK_UNROLL for (int i = 0; i < R; ++i) {
        double acc = A[i * K] * v[0];
        FK_UNROLL for (int k = 1; k < K; ++k) acc = fma(A[i * K + k], v[k], acc);
        out[i] = acc;
    }
}

// In-place L D L' (square-root-free Cholesky) of a symmetric M x M matrix; only the
// lower triangle is read.  On return the strict lower triangle holds L (unit
// diagonal implied), d[j] = D[j], dinv[j] = 1/D[j].  Returns true iff every
// pivot is > 0, i.e. the matrix is SPD.
// FAST: the pivots' reciprocals from fk_rcp (1e-16, five instructions) instead of the IEEE division (~25) -- for the
// arithmetic-bound callers (IMM); the memory-bound ones keep the division.
template <int M, bool FAST = false>
FK_HD bool ldlt2(double (&A)[M * M], double (&d)[M], double (&dinv)[M])
{
    bool pd = true;
    FK_UNROLL for (int j = 0; j < M; ++j) {
        double dj = A[j * M + j];
        FK_UNROLL for (int k = 0; k < j; ++k) {
            const double l = A[j * M + k];
            dj = fma(-l * l, d[k], dj);
        }
        pd = pd && (dj > 0.0);
        d[j] = dj;
        const double di = FAST ? fk_rcp(dj) : 1.0 / dj;
        dinv[j] = di;
        FK_UNROLL for (int i = j + 1; i < M; ++i) {
            double s = A[i * M + j];
            FK_UNROLL for (int k = 0; k < j; ++k)
                s = fma(-(A[i * M + k] * d[k]), A[j * M + k], s);
            A[i * M + j] = s * di;
        }
    }
    return pd;
}

// Solve  X * S = B  for X (R x M), S = L D L' given by ldlt2 (L strict-lower in Lm).
// i.e. each row r:  S x_r' = b_r'  (S symmetric).  In place on B.
template <int R, int M>
FK_HD void solve_rows_ldlt(const double (&Lm)[M * M], const double (&dinv)[M], double (&B)[R * M])
{
    FK_UNROLL for (int r = 0; r < R; ++r) {
        // forward: L w = b
        FK_UNROLL for (int i = 1; i < M; ++i) {
            double s = B[r * M + i];
            FK_UNROLL for (int k = 0; k < i; ++k) s = fma(-Lm[i * M + k], B[r * M + k], s);
            B[r * M + i] = s;
        }
        // diagonal
        FK_UNROLL for (int i = 0; i < M; ++i) B[r * M + i] *= dinv[i];
        // backward: L' x = w
        FK_UNROLL for (int i = M - 2; i >= 0; --i) {
            double s = B[r * M + i];
            FK_UNROLL for (int k = i + 1; k < M; ++k) s = fma(-Lm[k * M + i], B[r * M + k], s);
            B[r * M + i] = s;
        }
    }
}

// ------------------------------------------------------------------ model --
// The filter arithmetic below pulls the model matrices one ROW at a time through a
// `Model` policy (rowF / rowQ / rowH / rowR), so that only one row is live in VGPRs:
//   RegModel : per-track matrices held in registers (model_mode per-track);
//   LdsModel : matrices shared by all tracks, staged once in LDS and broadcast-read
//              (fk_device.hpp) -- at dim_x=4, dim_z=2 the shared model is 44 doubles
//              = 88 SGPRs, which does not fit the scalar file next to the addressing
//              state, and larger dims are hopeless; an LDS broadcast row costs one
//              ds_read_b128 per two doubles and no VALU work.
template <int NX, int NZ>
struct RegModel {
    double F[NX * NX], Q[NX * NX], H[NZ * NX], R[NZ * NZ];
    FK_HD void rowF(int i, double (&r)[NX]) const { FK_UNROLL for (int j = 0; j < NX; ++j) r[j] = F[i * NX + j]; }
    FK_HD void rowQ(int i, double (&r)[NX]) const { FK_UNROLL for (int j = 0; j < NX; ++j) r[j] = Q[i * NX + j]; }
    FK_HD void rowH(int i, double (&r)[NX]) const { FK_UNROLL for (int j = 0; j < NX; ++j) r[j] = H[i * NX + j]; }
    FK_HD void rowR(int i, double (&r)[NZ]) const { FK_UNROLL for (int j = 0; j < NZ; ++j) r[j] = R[i * NZ + j]; }
};

template <int K>
FK_HD double dot(const double (&a)[K], const double (&b)[K])
{
    double acc = a[0] * b[0];
    FK_UNROLL for (int k = 1; k < K; ++k) acc = fma(a[k], b[k], acc);
    return acc;
}

// ---------------------------------------------------------------- predict --
// x = F x (+ Bu added by the caller);  P = alpha_sq * ((F P) F') + Q
// (filterpy/kalman/kalman_filter.py:472-478)
template <int NX, class Model>
FK_HD void kf_predict(double (&x)[NX], double (&P)[NX * NX], const Model &M, double alpha_sq)
{
    double xn[NX];
    double FP[NX * NX];
    FK_UNROLL for (int i = 0; i < NX; ++i) {
        double f[NX];
        M.rowF(i, f);
        xn[i] = dot<NX>(f, x);
        FK_UNROLL for (int j = 0; j < NX; ++j) {
            double acc = f[0] * P[j];
            FK_UNROLL for (int k = 1; k < NX; ++k) acc = fma(f[k], P[k * NX + j], acc);
            FP[i * NX + j] = acc;
        }
        FK_STAGE();
    }
    FK_UNROLL for (int i = 0; i < NX; ++i) x[i] = xn[i];
    // (FP F'): column j of the product needs row j of F -- one model row live at a time;
    // Q is added in a third sweep, by rows.
    FK_UNROLL for (int j = 0; j < NX; ++j) {
        double f[NX];
        M.rowF(j, f);
        FK_UNROLL for (int i = 0; i < NX; ++i) {
            double acc = FP[i * NX] * f[0];
            FK_UNROLL for (int k = 1; k < NX; ++k) acc = fma(FP[i * NX + k], f[k], acc);
            P[i * NX + j] = acc;
        }
        FK_STAGE();
    }
    FK_UNROLL for (int i = 0; i < NX; ++i) {
        double q[NX];
        M.rowQ(i, q);
        FK_UNROLL for (int j = 0; j < NX; ++j) P[i * NX + j] = fma(alpha_sq, P[i * NX + j], q[j]);
        FK_STAGE();
    }
}

// ----------------------------------------------------------------- update --
// y = z - Hx; PHT = P H'; S = H PHT + R; K = PHT S^-1; x += K y;
// P = (I-KH) P (I-KH)' + K R K'     (filterpy/kalman/kalman_filter.py:533-556)
// Outputs K, y, S and the factorisation of S (Lf, dinv) for the optional SI output.
// Returns status bits.
template <int NX, int NZ, class Model>
FK_HD int kf_update(double (&x)[NX], double (&P)[NX * NX], const double (&z)[NZ], const Model &M,
                    double (&K)[NX * NZ], double (&y)[NZ], double (&S)[NZ * NZ],
                    double (&Lf)[NZ * NZ], double (&dinv)[NZ], bool rj_diag = false)
{
    // rj_diag: the reference with a scalar R attribute and dim_z > 1 (kalman_filter.py:540 adds r to every element of
    // S, :556 forms r K K'): R arrives as r * ones and the Joseph term below keeps only its diagonal
    int st = 0;
    double PHT[NX * NZ];
    FK_UNROLL for (int r = 0; r < NZ; ++r) {
        double h[NX];
        M.rowH(r, h);
        y[r] = z[r] - dot<NX>(h, x);
        FK_UNROLL for (int i = 0; i < NX; ++i) {
            double acc = P[i * NX] * h[0];
            FK_UNROLL for (int k = 1; k < NX; ++k) acc = fma(P[i * NX + k], h[k], acc);
            PHT[i * NZ + r] = acc;
        }
        FK_STAGE();
    }
    FK_UNROLL for (int r = 0; r < NZ; ++r) {
        double h[NX], rr[NZ];
        M.rowH(r, h);
        M.rowR(r, rr);
        FK_UNROLL for (int c = 0; c < NZ; ++c) {
            double acc = h[0] * PHT[c];
            FK_UNROLL for (int k = 1; k < NX; ++k) acc = fma(h[k], PHT[k * NZ + c], acc);
            S[r * NZ + c] = acc + rr[c];
        }
        FK_STAGE();
    }
    FK_UNROLL for (int i = 0; i < NX * NZ; ++i) K[i] = PHT[i];
    if constexpr (NZ == 1) {
        // 1x1: SI = 1/S exactly as numpy.linalg.inv returns it; K = PHT * SI
        const double si = 1.0 / S[0];
        if (!(S[0] != 0.0)) st |= ST_NOT_PD;
        dinv[0] = si;
        Lf[0] = S[0];
        FK_UNROLL for (int i = 0; i < NX; ++i) K[i] = PHT[i] * si;
    } else {
        double d[NZ];
        FK_UNROLL for (int i = 0; i < NZ * NZ; ++i) Lf[i] = S[i];
        if (!ldlt2<NZ>(Lf, d, dinv)) st |= ST_NOT_PD;
        solve_rows_ldlt<NX, NZ>(Lf, dinv, K);
    }
    FK_STAGE();

    FK_UNROLL for (int i = 0; i < NX; ++i) {
        double acc = x[i];
        FK_UNROLL for (int k = 0; k < NZ; ++k) acc = fma(K[i * NZ + k], y[k], acc);
        x[i] = acc;
    }

    // I_KH = I - K H
    double IKH[NX * NX];
    FK_UNROLL for (int i = 0; i < NX; ++i)
        FK_UNROLL for (int j = 0; j < NX; ++j) IKH[i * NX + j] = (i == j) ? 1.0 : 0.0;
    FK_UNROLL for (int r = 0; r < NZ; ++r) {
        double h[NX];
        M.rowH(r, h);
        FK_UNROLL for (int i = 0; i < NX; ++i)
            FK_UNROLL for (int j = 0; j < NX; ++j) IKH[i * NX + j] = fma(-K[i * NZ + r], h[j], IKH[i * NX + j]);
        FK_STAGE();
    }
    // T1 = I_KH P   (P is dead afterwards and is overwritten by the result below)
    double T1[NX * NX];
    matmul<NX, NX, NX>(IKH, P, T1);
    FK_STAGE();
    // KR = K R
    double KR[NX * NZ];
    FK_UNROLL for (int i = 0; i < NX * NZ; ++i) KR[i] = 0.0;
    FK_UNROLL for (int r = 0; r < NZ; ++r) {
        double rr[NZ];
        M.rowR(r, rr);
        FK_UNROLL for (int c = 0; c < NZ; ++c) rr[c] = (rj_diag && c != r) ? 0.0 : rr[c];
        FK_UNROLL for (int i = 0; i < NX; ++i)
            FK_UNROLL for (int c = 0; c < NZ; ++c)
                KR[i * NZ + c] = (r == 0) ? K[i * NZ] * rr[c] : fma(K[i * NZ + r], rr[c], KR[i * NZ + c]);
    }
    FK_STAGE();
    FK_UNROLL for (int i = 0; i < NX; ++i) {
        FK_UNROLL for (int j = 0; j < NX; ++j) {
            double acc = T1[i * NX] * IKH[j * NX];
            FK_UNROLL for (int k = 1; k < NX; ++k) acc = fma(T1[i * NX + k], IKH[j * NX + k], acc);
            FK_UNROLL for (int k = 0; k < NZ; ++k) acc = fma(KR[i * NZ + k], K[j * NZ + k], acc);
            P[i * NX + j] = acc;
        }
        FK_STAGE();
    }
    return st;
}

// ---- a caller-supplied inverse (KalmanFilter.inv = numpy.linalg.pinv, kalman_filter.py:363,434,541) ----------------------
// The reference calls whatever `self.inv` names on S; a drop-in cannot run a Python callable inside a kernel, so update() is cut
// at that call: kf_innovation forms y and S (:533-540), the host applies the callable, kf_update_given_si takes its result --
// K = PHT SI (:545), x += K y (:549), the Joseph form (:555-556) in kf_update's operation order.  No factorisation, no status.
template <int NX, int NZ, class Model>
FK_HD void kf_innovation(const double (&x)[NX], const double (&P)[NX * NX], const double (&z)[NZ], const Model &M,
                         double (&PHT)[NX * NZ], double (&y)[NZ], double (&S)[NZ * NZ])
{
    FK_UNROLL for (int r = 0; r < NZ; ++r) {
        double h[NX];
        M.rowH(r, h);
        y[r] = z[r] - dot<NX>(h, x);
        FK_UNROLL for (int i = 0; i < NX; ++i) {
            double acc = P[i * NX] * h[0];
            FK_UNROLL for (int k = 1; k < NX; ++k) acc = fma(P[i * NX + k], h[k], acc);
            PHT[i * NZ + r] = acc;
        }
        FK_STAGE();
    }
    FK_UNROLL for (int r = 0; r < NZ; ++r) {
        double h[NX], rr[NZ];
        M.rowH(r, h);
        M.rowR(r, rr);
        FK_UNROLL for (int c = 0; c < NZ; ++c) {
            double acc = h[0] * PHT[c];
            FK_UNROLL for (int k = 1; k < NX; ++k) acc = fma(h[k], PHT[k * NZ + c], acc);
            S[r * NZ + c] = acc + rr[c];
        }
        FK_STAGE();
    }
}

template <int NX, int NZ, class Model>
FK_HD void kf_update_given_si(double (&x)[NX], double (&P)[NX * NX], const double (&z)[NZ], const Model &M,
                              const double (&SI)[NZ * NZ], double (&K)[NX * NZ], double (&y)[NZ], double (&S)[NZ * NZ],
                              bool rj_diag = false)
{
    double PHT[NX * NZ];
    kf_innovation<NX, NZ>(x, P, z, M, PHT, y, S);
    FK_UNROLL for (int i = 0; i < NX; ++i) {
        FK_UNROLL for (int c = 0; c < NZ; ++c) {
            double acc = PHT[i * NZ] * SI[c];
            FK_UNROLL for (int k = 1; k < NZ; ++k) acc = fma(PHT[i * NZ + k], SI[k * NZ + c], acc);
            K[i * NZ + c] = acc;
        }
    }
    FK_STAGE();
    FK_UNROLL for (int i = 0; i < NX; ++i) {
        double acc = x[i];
        FK_UNROLL for (int k = 0; k < NZ; ++k) acc = fma(K[i * NZ + k], y[k], acc);
        x[i] = acc;
    }
    double IKH[NX * NX];
    FK_UNROLL for (int i = 0; i < NX; ++i)
        FK_UNROLL for (int j = 0; j < NX; ++j) IKH[i * NX + j] = (i == j) ? 1.0 : 0.0;
    FK_UNROLL for (int r = 0; r < NZ; ++r) {
        double h[NX];
        M.rowH(r, h);
        FK_UNROLL for (int i = 0; i < NX; ++i)
            FK_UNROLL for (int j = 0; j < NX; ++j) IKH[i * NX + j] = fma(-K[i * NZ + r], h[j], IKH[i * NX + j]);
        FK_STAGE();
    }
    double T1[NX * NX];
    matmul<NX, NX, NX>(IKH, P, T1);
    FK_STAGE();
    double KR[NX * NZ];
    FK_UNROLL for (int i = 0; i < NX * NZ; ++i) KR[i] = 0.0;
    FK_UNROLL for (int r = 0; r < NZ; ++r) {
        double rr[NZ];
        M.rowR(r, rr);
        FK_UNROLL for (int c = 0; c < NZ; ++c) rr[c] = (rj_diag && c != r) ? 0.0 : rr[c];
        FK_UNROLL for (int i = 0; i < NX; ++i)
            FK_UNROLL for (int c = 0; c < NZ; ++c)
                KR[i * NZ + c] = (r == 0) ? K[i * NZ] * rr[c] : fma(K[i * NZ + r], rr[c], KR[i * NZ + c]);
    }
    FK_STAGE();
    FK_UNROLL for (int i = 0; i < NX; ++i) {
        FK_UNROLL for (int j = 0; j < NX; ++j) {
            double acc = T1[i * NX] * IKH[j * NX];
            FK_UNROLL for (int k = 1; k < NX; ++k) acc = fma(T1[i * NX + k], IKH[j * NX + k], acc);
            FK_UNROLL for (int k = 0; k < NZ; ++k) acc = fma(KR[i * NZ + k], K[j * NZ + k], acc);
            P[i * NX + j] = acc;
        }
        FK_STAGE();
    }
}

// S^-1 from the factorisation produced inside kf_update (only for the optional SI output).
template <int NZ>
FK_HD void inv_from_ldlt(const double (&Lf)[NZ * NZ], const double (&dinv)[NZ], double (&SI)[NZ * NZ])
{
    if constexpr (NZ == 1) {
        SI[0] = dinv[0];
        return;
    }
    FK_UNROLL for (int i = 0; i < NZ; ++i)
        FK_UNROLL for (int j = 0; j < NZ; ++j) SI[i * NZ + j] = (i == j) ? 1.0 : 0.0;
    solve_rows_ldlt<NZ, NZ>(Lf, dinv, SI);
}

// -------------------------------------------------------------------- RTS --
// One backward step (kalman_filter.py:1067-1072):
//   Pp = F P F' + Q ;  K = (P F') Pp^-1 ;  x += K (xn - F x) ;  P += (K (Pn - Pp)) K'
// x,P: filtered at k (in) -> smoothed at k (out).  xn,Pn: smoothed at k+1.
template <int NX, class Model>
FK_HD int rts_step(double (&x)[NX], double (&P)[NX * NX], const double (&xn)[NX],
                   const double (&Pn)[NX * NX], const Model &M, double (&K)[NX * NX],
                   double (&Pp)[NX * NX])
{
    int st = 0;
    double FP[NX * NX];
    double Fx[NX];
    FK_UNROLL for (int i = 0; i < NX; ++i) {
        double f[NX];
        M.rowF(i, f);
        Fx[i] = dot<NX>(f, x);
        FK_UNROLL for (int j = 0; j < NX; ++j) {
            double acc = f[0] * P[j];
            FK_UNROLL for (int k = 1; k < NX; ++k) acc = fma(f[k], P[k * NX + j], acc);
            FP[i * NX + j] = acc;
        }
        FK_STAGE();
    }
    // Pp = FP F' + Q ;  K0 = P F'
    FK_UNROLL for (int j = 0; j < NX; ++j) {
        double f[NX];
        M.rowF(j, f);
        FK_UNROLL for (int i = 0; i < NX; ++i) {
            double acc = FP[i * NX] * f[0];
            double acc2 = P[i * NX] * f[0];
            FK_UNROLL for (int k = 1; k < NX; ++k) {
                acc = fma(FP[i * NX + k], f[k], acc);
                acc2 = fma(P[i * NX + k], f[k], acc2);
            }
            Pp[i * NX + j] = acc;
            K[i * NX + j] = acc2;
        }
        FK_STAGE();
    }
    FK_UNROLL for (int i = 0; i < NX; ++i) {
        double q[NX];
        M.rowQ(i, q);
        FK_UNROLL for (int j = 0; j < NX; ++j) Pp[i * NX + j] += q[j];
    }
    FK_STAGE();
    // K = (P F') Pp^-1
    {
        double Lf[NX * NX], d[NX], dinv[NX];
        FK_UNROLL for (int i = 0; i < NX * NX; ++i) Lf[i] = Pp[i];
        if (!ldlt2<NX>(Lf, d, dinv)) st |= ST_NOT_PD;
        solve_rows_ldlt<NX, NX>(Lf, dinv, K);
    }
    FK_STAGE();
    FK_UNROLL for (int i = 0; i < NX; ++i) {
        double acc = x[i];
        FK_UNROLL for (int k = 0; k < NX; ++k) acc = fma(K[i * NX + k], xn[k] - Fx[k], acc);
        x[i] = acc;
    }
    double KD[NX * NX];
    FK_UNROLL for (int i = 0; i < NX; ++i) {
        FK_UNROLL for (int j = 0; j < NX; ++j) {
            double acc = K[i * NX] * (Pn[j] - Pp[j]);
            FK_UNROLL for (int k = 1; k < NX; ++k) acc = fma(K[i * NX + k], Pn[k * NX + j] - Pp[k * NX + j], acc);
            KD[i * NX + j] = acc;
        }
    }
    FK_STAGE();
    FK_UNROLL for (int i = 0; i < NX; ++i) {
        FK_UNROLL for (int j = 0; j < NX; ++j) {
            double acc = P[i * NX + j];
            FK_UNROLL for (int k = 0; k < NX; ++k) acc = fma(KD[i * NX + k], K[j * NX + k], acc);
            P[i * NX + j] = acc;
        }
    }
    return st;
}

// rts_smoother(inv=...) (kalman_filter.py:995, 1069): the backward recursion cut at the callable like update() above.  Pp[k] =
// F P[k] F' + Q depends on the FILTERED covariance only, so one pass stores every Pp (rts_pp_only), the host inverts them, and
// the sweep takes the inverses (rts_step_given: K = (P F') PpInv, then :1071-1072 in rts_step's order).
template <int NX, class Model>
FK_HD void rts_pp_only(const double (&P)[NX * NX], const Model &M, double (&Pp)[NX * NX])
{
    double FP[NX * NX];
    FK_UNROLL for (int i = 0; i < NX; ++i) {
        double f[NX];
        M.rowF(i, f);
        FK_UNROLL for (int j = 0; j < NX; ++j) {
            double acc = f[0] * P[j];
            FK_UNROLL for (int k = 1; k < NX; ++k) acc = fma(f[k], P[k * NX + j], acc);
            FP[i * NX + j] = acc;
        }
        FK_STAGE();
    }
    FK_UNROLL for (int j = 0; j < NX; ++j) {
        double f[NX];
        M.rowF(j, f);
        FK_UNROLL for (int i = 0; i < NX; ++i) {
            double acc = FP[i * NX] * f[0];
            FK_UNROLL for (int k = 1; k < NX; ++k) acc = fma(FP[i * NX + k], f[k], acc);
            Pp[i * NX + j] = acc;
        }
        FK_STAGE();
    }
    FK_UNROLL for (int i = 0; i < NX; ++i) {
        double q[NX];
        M.rowQ(i, q);
        FK_UNROLL for (int j = 0; j < NX; ++j) Pp[i * NX + j] += q[j];
    }
}

// K: in = the caller's inv(Pp[k]), out = the gain
template <int NX, class Model>
FK_HD void rts_step_given(double (&x)[NX], double (&P)[NX * NX], const double (&xn)[NX], const double (&Pn)[NX * NX],
                          const Model &M, double (&K)[NX * NX], double (&Pp)[NX * NX])
{
    rts_pp_only<NX>(P, M, Pp);
    double Fx[NX], PFt[NX * NX];
    FK_UNROLL for (int i = 0; i < NX; ++i) {
        double f[NX];
        M.rowF(i, f);
        Fx[i] = dot<NX>(f, x);
        FK_UNROLL for (int r = 0; r < NX; ++r) {
            double acc = P[r * NX] * f[0];
            FK_UNROLL for (int k = 1; k < NX; ++k) acc = fma(P[r * NX + k], f[k], acc);
            PFt[r * NX + i] = acc;
        }
        FK_STAGE();
    }
    double G[NX * NX];
    FK_UNROLL for (int i = 0; i < NX; ++i)
        FK_UNROLL for (int j = 0; j < NX; ++j) {
            double acc = PFt[i * NX] * K[j];
            FK_UNROLL for (int k = 1; k < NX; ++k) acc = fma(PFt[i * NX + k], K[k * NX + j], acc);
            G[i * NX + j] = acc;
        }
    FK_UNROLL for (int i = 0; i < NX * NX; ++i) K[i] = G[i];
    FK_STAGE();
    FK_UNROLL for (int i = 0; i < NX; ++i) {
        double acc = x[i];
        FK_UNROLL for (int k = 0; k < NX; ++k) acc = fma(K[i * NX + k], xn[k] - Fx[k], acc);
        x[i] = acc;
    }
    double KD[NX * NX];
    FK_UNROLL for (int i = 0; i < NX; ++i) {
        FK_UNROLL for (int j = 0; j < NX; ++j) {
            double acc = K[i * NX] * (Pn[j] - Pp[j]);
            FK_UNROLL for (int k = 1; k < NX; ++k) acc = fma(K[i * NX + k], Pn[k * NX + j] - Pp[k * NX + j], acc);
            KD[i * NX + j] = acc;
        }
    }
    FK_STAGE();
    FK_UNROLL for (int i = 0; i < NX; ++i) {
        FK_UNROLL for (int j = 0; j < NX; ++j) {
            double acc = P[i * NX + j];
            FK_UNROLL for (int k = 0; k < NX; ++k) acc = fma(KD[i * NX + k], K[j * NX + k], acc);
            P[i * NX + j] = acc;
        }
    }
}

// ------------------------------------------------------------ sigma / UT --
// Upper Cholesky U (U'U = scale*P), rows U[k] as scipy.linalg.cholesky returns them.
// Stored as L = U' (lower): U[k][j] = L[j*NX+k].  Returns true iff SPD.
template <int NX>
FK_HD bool chol_lower(const double (&P)[NX * NX], double scale, double (&L)[NX * NX])
{
    bool pd = true;
    FK_UNROLL for (int j = 0; j < NX; ++j) {
        double d = scale * P[j * NX + j];
        FK_UNROLL for (int k = 0; k < j; ++k) d = fma(-L[j * NX + k], L[j * NX + k], d);
        pd = pd && (d > 0.0);
        const double ljj = sqrt(d);
        L[j * NX + j] = ljj;
        const double inv = 1.0 / ljj;
        FK_UNROLL for (int i = j + 1; i < NX; ++i) {
            // LAPACK dpotrf('U') reads the UPPER triangle of the input: element (j,i)
            double s = scale * P[j * NX + i];
            FK_UNROLL for (int k = 0; k < j; ++k) s = fma(-L[i * NX + k], L[j * NX + k], s);
            L[i * NX + j] = s * inv;
        }
        FK_UNROLL for (int i = 0; i < j; ++i) L[i * NX + j] = 0.0;
    }
    return pd;
}

// sigma point i (0..2n) component c, from x and L=U'   (sigma_points.py:170-175)
template <int NX>
FK_HD double sigma_elem(const double (&x)[NX], const double (&L)[NX * NX], int i, int c)
{
    if (i == 0) return x[c];
    if (i <= NX) return x[c] + L[c * NX + (i - 1)];
    return x[c] - L[c * NX + (i - 1 - NX)];
}

}  // namespace fk
