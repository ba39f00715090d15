// fk_math_sym.hpp -- Kalman predict/update on a PACKED SYMMETRIC covariance.
//
// Same equations as fk_math.hpp (filterpy/kalman/kalman_filter.py:472-478, 533-556) but P lives
// in registers as its upper triangle U (n(n+1)/2 doubles instead of n^2) and no n x n temporary
// is ever materialised: every product is formed one row at a time.  This is what lets the
// per-lane state of dim_x = 9 fit the register file without spilling and dim_x = 4 run at
// 3-4 waves per SIMD.  Only valid for symmetric P, Q, R (covariances): the fast kernel reads the
// upper triangle of P0 only; kf_kernels.hip keeps the fully general arithmetic.
//
// Update, with PHT = P H', S = H PHT + R, K = PHT S^-1 as in the reference:
//   (I-KH) P          = P - K PHT'                        =: T1   (row i: U(i,:) - K_i PHT')
//   T1 (I-KH)'        = T1 - (T1 H') K'                           (G_i = T1_i H', m values)
//   Joseph:  P+ = T1 (I-KH)' + K R K' = T1 + ((K R)_i - G_i) K'   (upper triangle only)
// i.e. the reference's (I-KH) P (I-KH)' + K R K' with the identity-minus-product factors applied
// implicitly; rounding differences vs the factored form are O(eps |P|), the same size as the
// reference's own.
#pragma once
#include <type_traits>

#include "fk_math.hpp"

namespace fk {

// packed index of element (i, j) of a symmetric NX x NX matrix stored by rows of its upper triangle
template <int NX>
FK_HD constexpr int sym_idx(int i, int j)
{
    return (i <= j) ? (i * NX - i * (i - 1) / 2 + (j - i)) : (j * NX - j * (j - 1) / 2 + (i - j));
}

template <int NX>
constexpr int SYM_LEN = NX * (NX + 1) / 2;

// A model staged in LDS (LdsModel: IN_LDS) hands out one row per call, and the arithmetic below consumes a row right after
// asking for it: every request exposes the full LDS latency.  With several waves per SIMD another wave covers it; the
// instantiations of dim_x >= 7 run ONE wave per SIMD and nothing does -- at (8,4) 240 of the step's 311 LDS waits sat directly
// behind their reads, about as many clocks as the step's 2500 VALU instructions (round 4, ISA of kf_fast 8_4).  There the
// whole block a half-step needs (F, then Q; H and R) is read into registers in ONE batch at its head -- the register file of a
// lone wave has the room (F at dim_x 9: 162 of 512 VGPRs) -- and the arithmetic runs on the copy.  Same values into the same
// operations: bit-identical results.
#ifndef FK_MODEL_CACHE_MIN_NX
#define FK_MODEL_CACHE_MIN_NX 7
#endif
#ifndef FK_MODEL_CACHE_UPDATE
#define FK_MODEL_CACHE_UPDATE 1
#endif
#ifndef FK_MODEL_CACHE_R
#define FK_MODEL_CACHE_R 1
#endif
template <class M, class = void>
struct model_in_lds { static constexpr bool value = false; };
template <class M>
struct model_in_lds<M, std::void_t<decltype(M::IN_LDS)>> { static constexpr bool value = M::IN_LDS; };
// (not in the rolled builds -- FK_ROLLED: loops kept as loops, arrays in scratch memory --, where a copy indexed by a loop
//  variable would be one more scratch array)
#if defined(FK_ROLLED) && FK_ROLLED
constexpr bool model_cache_build = false;
#else
constexpr bool model_cache_build = true;
#endif
template <class M, int NX>
constexpr bool model_cached = model_cache_build && model_in_lds<M>::value && NX >= FK_MODEL_CACHE_MIN_NX;

template <int NX, class Model>
FK_HD void kf_predict_sym(double (&x)[NX], double (&U)[NX * (NX + 1) / 2], const Model &M, double alpha_sq)
{
    constexpr bool PF = model_cached<Model, NX>;
    double xn[NX];
    double Un[NX * (NX + 1) / 2];
    double Fc[PF ? NX : 1][NX];
    if constexpr (PF) {
        FK_UNROLL for (int i = 0; i < NX; ++i) M.rowF(i, Fc[i]);
        FK_STAGE();
    }
    FK_UNROLL for (int i = 0; i < NX; ++i) {
        double f[NX];
        if constexpr (PF) { FK_UNROLL for (int k = 0; k < NX; ++k) f[k] = Fc[PF ? i : 0][k]; }
        else M.rowF(i, f);
        xn[i] = dot<NX>(f, x);
        // row i of F P
        double fp[NX];
        FK_UNROLL for (int j = 0; j < NX; ++j) {
            double acc = f[0] * U[sym_idx<NX>(0, j)];
            FK_UNROLL for (int k = 1; k < NX; ++k) acc = fma(f[k], U[sym_idx<NX>(k, j)], acc);
            fp[j] = acc;
        }
        // (F P F')(i, j >= i) = fp . F_j
        Un[sym_idx<NX>(i, i)] = dot<NX>(fp, f);
        FK_UNROLL for (int j = i + 1; j < NX; ++j) {
            double g[NX];
            if constexpr (PF) { FK_UNROLL for (int k = 0; k < NX; ++k) g[k] = Fc[PF ? j : 0][k]; }
            else M.rowF(j, g);
            Un[sym_idx<NX>(i, j)] = dot<NX>(fp, g);
        }
        FK_STAGE();
    }
    FK_UNROLL for (int i = 0; i < NX; ++i) x[i] = xn[i];
    if constexpr (PF) {
        double Qc[NX][NX];                       // (F's copy is dead: Q takes its registers)
        FK_UNROLL for (int i = 0; i < NX; ++i) M.rowQ(i, Qc[i]);
        FK_STAGE();
        FK_UNROLL for (int i = 0; i < NX; ++i)
            FK_UNROLL for (int j = i; j < NX; ++j) U[sym_idx<NX>(i, j)] = fma(alpha_sq, Un[sym_idx<NX>(i, j)], Qc[i][j]);
    } else {
        FK_UNROLL for (int i = 0; i < NX; ++i) {
            double q[NX];
            M.rowQ(i, q);
            FK_UNROLL for (int j = i; j < NX; ++j) U[sym_idx<NX>(i, j)] = fma(alpha_sq, Un[sym_idx<NX>(i, j)], q[j]);
        }
    }
    FK_STAGE();
}

// Returns status bits.  K, y, S (full m x m) and the factorisation are outputs like kf_update.
// UPD_CACHE: a caller whose own register peak sits in the update half switches the H / R copy off (model_cached)
template <int NX, int NZ, bool FAST_RCP = false, bool UPD_CACHE = true, class Model>
FK_HD int kf_update_sym(double (&x)[NX], double (&U)[NX * (NX + 1) / 2], const double (&z)[NZ], const Model &M,
                        double (&K)[NX * NZ], double (&y)[NZ], double (&S)[NZ * NZ],
                        double (&Lf)[NZ * NZ], double (&dinv)[NZ], bool rj_diag = false)
{
    int st = 0;
    constexpr bool PF = model_cached<Model, NX> && UPD_CACHE && FK_MODEL_CACHE_UPDATE;
    constexpr bool PFR = PF && FK_MODEL_CACHE_R;
    double Hc[PF ? NZ : 1][NX], Rc[PFR ? NZ : 1][NZ];
    if constexpr (PF) {
        FK_UNROLL for (int r = 0; r < NZ; ++r) {
            M.rowH(r, Hc[r]);
            if constexpr (PFR) M.rowR(r, Rc[PFR ? r : 0]);
        }
        FK_STAGE();
    }
#define FK_ROWH(r, h) do { if constexpr (PF) { FK_UNROLL for (int k_ = 0; k_ < NX; ++k_) h[k_] = Hc[PF ? (r) : 0][k_]; } else M.rowH((r), h); } while (0)
#define FK_ROWR(r, rr) do { if constexpr (PFR) { FK_UNROLL for (int k_ = 0; k_ < NZ; ++k_) rr[k_] = Rc[PFR ? (r) : 0][k_]; } else M.rowR((r), rr); } while (0)
    double PHT[NX * NZ];
    FK_UNROLL for (int r = 0; r < NZ; ++r) {
        double h[NX];
        FK_ROWH(r, h);
        y[r] = z[r] - dot<NX>(h, x);
        FK_UNROLL for (int i = 0; i < NX; ++i) {
            double acc = U[sym_idx<NX>(i, 0)] * h[0];
            FK_UNROLL for (int k = 1; k < NX; ++k) acc = fma(U[sym_idx<NX>(i, k)], h[k], acc);
            PHT[i * NZ + r] = acc;
        }
    }
    FK_STAGE();
    FK_UNROLL for (int r = 0; r < NZ; ++r) {
        double h[NX], rr[NZ];
        FK_ROWH(r, h);
        FK_ROWR(r, rr);
        FK_UNROLL for (int c = 0; c < NZ; ++c) {
            double acc = h[0] * PHT[c];
            FK_UNROLL for (int k = 1; k < NX; ++k) acc = fma(h[k], PHT[k * NZ + c], acc);
            S[r * NZ + c] = acc + rr[c];
        }
    }
    FK_STAGE();
    FK_UNROLL for (int i = 0; i < NX * NZ; ++i) K[i] = PHT[i];
    if constexpr (NZ == 1) {
        const double si = FAST_RCP ? fk_rcp(S[0]) : 1.0 / S[0];
        if (!(S[0] != 0.0)) st |= ST_NOT_PD;
        dinv[0] = si;
        Lf[0] = S[0];
        FK_UNROLL for (int i = 0; i < NX; ++i) K[i] = PHT[i] * si;
    } else {
        double d[NZ];
        FK_UNROLL for (int i = 0; i < NZ * NZ; ++i) Lf[i] = S[i];
        if (!ldlt2<NZ, FAST_RCP>(Lf, d, dinv)) st |= ST_NOT_PD;
        solve_rows_ldlt<NX, NZ>(Lf, dinv, K);
    }
    FK_STAGE();
    FK_UNROLL for (int i = 0; i < NX; ++i) {
        double acc = x[i];
        FK_UNROLL for (int k = 0; k < NZ; ++k) acc = fma(K[i * NZ + k], y[k], acc);
        x[i] = acc;
    }
    // Joseph form, one row of the result at a time
    double Un[NX * (NX + 1) / 2];
    FK_UNROLL for (int i = 0; i < NX; ++i) {
        // T1_i = row i of (I - K H) P = U(i,:) - K_i PHT'
        double t1[NX];
        FK_UNROLL for (int k = 0; k < NX; ++k) {
            double acc = U[sym_idx<NX>(i, k)];
            FK_UNROLL for (int r = 0; r < NZ; ++r) acc = fma(-K[i * NZ + r], PHT[k * NZ + r], acc);
            t1[k] = acc;
        }
        // D_i = (K R)_i - T1_i H'
        double D[NZ];
        FK_UNROLL for (int c = 0; c < NZ; ++c) D[c] = 0.0;
        FK_UNROLL for (int r = 0; r < NZ; ++r) {
            double rr[NZ];
            FK_ROWR(r, rr);
            FK_UNROLL for (int c = 0; c < NZ; ++c) rr[c] = (rj_diag && c != r) ? 0.0 : rr[c];   // fk_math.hpp, kf_update
            FK_UNROLL for (int c = 0; c < NZ; ++c) D[c] = (r == 0) ? K[i * NZ] * rr[c] : fma(K[i * NZ + r], rr[c], D[c]);
        }
        FK_UNROLL for (int r = 0; r < NZ; ++r) {
            double h[NX];
            FK_ROWH(r, h);
            D[r] -= dot<NX>(t1, h);
        }
        FK_UNROLL for (int j = i; j < NX; ++j) {
            double acc = t1[j];
            FK_UNROLL for (int r = 0; r < NZ; ++r) acc = fma(D[r], K[j * NZ + r], acc);
            Un[sym_idx<NX>(i, j)] = acc;
        }
        FK_STAGE();
    }
    FK_UNROLL for (int e = 0; e < NX * (NX + 1) / 2; ++e) U[e] = Un[e];
#undef FK_ROWH
#undef FK_ROWR
    return st;
}

// ------------------------------------------------------------------------------------ RTS --
// In-place L D L' of a packed symmetric matrix (upper-triangle storage A(i,j), i <= j, read as the
// lower triangle A(j,i)).  On return A(i,j), i < j, holds L(j,i) and A(j,j) is untouched garbage-free:
// d[j], dinv[j] carry the pivots.  Returns true iff SPD.
template <int N>
FK_HD bool ldlt_packed(double (&A)[N * (N + 1) / 2], double (&d)[N], double (&dinv)[N])
{
    bool pd = true;
    FK_UNROLL for (int j = 0; j < N; ++j) {
        double dj = A[sym_idx<N>(j, j)];
        FK_UNROLL for (int k = 0; k < N; ++k)
            if (k < j) {
                const double l = A[sym_idx<N>(j, k)];
                dj = fma(-l * l, d[k], dj);
            }
        pd = pd && (dj > 0.0);
        d[j] = dj;
        const double di = 1.0 / dj;
        dinv[j] = di;
        FK_UNROLL for (int i = 0; i < N; ++i)
            if (i > j) {
                double t = A[sym_idx<N>(i, j)];
                FK_UNROLL for (int k = 0; k < N; ++k)
                    if (k < j) t = fma(-(A[sym_idx<N>(i, k)] * d[k]), A[sym_idx<N>(j, k)], t);
                A[sym_idx<N>(i, j)] = t * di;
            }
    }
    return pd;
}

// solve x S = b for one row (S = L D L' from ldlt_packed), in place
template <int N>
FK_HD void solve_row_packed(const double (&Lp)[N * (N + 1) / 2], const double (&dinv)[N], double (&b)[N])
{
    FK_UNROLL for (int i = 1; i < N; ++i) {
        double t = b[i];
        FK_UNROLL for (int k = 0; k < N; ++k)
            if (k < i) t = fma(-Lp[sym_idx<N>(i, k)], b[k], t);
        b[i] = t;
    }
    FK_UNROLL for (int i = 0; i < N; ++i) b[i] *= dinv[i];
    FK_UNROLL for (int i = N - 2; i >= 0; --i) {
        double t = b[i];
        FK_UNROLL for (int k = 0; k < N; ++k)
            if (k > i) t = fma(-Lp[sym_idx<N>(k, i)], b[k], t);
        b[i] = t;
    }
}

// One backward RTS step on packed symmetric covariances (kalman_filter.py:1067-1072):
//   Pp = F P F' + Q ; K = (P F') Pp^-1 ; x += K (xn - F x) ; P += K (Pn - Pp) K'
// U: filtered P_k in, smoothed P_k out.  Un: smoothed P_{k+1} in, DESTROYED (holds Pn - Pp).
// K: full n x n gain (row-major).  The packed predicted covariance Pp is handed to `pp_sink` the
// moment it is complete and its registers are then reused for the factorisation (at dim_x = 9 the
// live set would otherwise exceed the 512-register file).
template <int NX, class Model, class PpSink>
FK_HD int rts_step_sym(double (&x)[NX], double (&U)[NX * (NX + 1) / 2], const double (&xn)[NX],
                       double (&Un)[NX * (NX + 1) / 2], const Model &M, double (&K)[NX * NX],
                       PpSink &&pp_sink)
{
    double Pp[NX * (NX + 1) / 2];
    int st = 0;
    double dx[NX];
    // K0 = P F' = (F P)' : row i of F P is column i of K0
    // (no model_cached copy here: the one-lane smoother of dim_x 7, 8 already spills, and the copy only adds to it)
    FK_UNROLL for (int i = 0; i < NX; ++i) {
        double f[NX];
        M.rowF(i, f);
        dx[i] = xn[i] - dot<NX>(f, x);
        FK_UNROLL for (int j = 0; j < NX; ++j) {
            double acc = f[0] * U[sym_idx<NX>(0, j)];
            FK_UNROLL for (int k = 1; k < NX; ++k) acc = fma(f[k], U[sym_idx<NX>(k, j)], acc);
            K[j * NX + i] = acc;
        }
        FK_STAGE();
    }
    // Pp(i, j >= i) = (F P)_i . F_j + Q(i,j) ;  (F P)_i = column i of K0
    FK_UNROLL for (int j = 0; j < NX; ++j) {
        double f[NX], q[NX];
        M.rowF(j, f);
        M.rowQ(j, q);
        FK_UNROLL for (int i = 0; i < NX; ++i)
            if (i <= j) {
                double acc = K[i] * f[0];
                FK_UNROLL for (int k = 1; k < NX; ++k) acc = fma(K[k * NX + i], f[k], acc);
                Pp[sym_idx<NX>(i, j)] = acc + q[i];
            }
        FK_STAGE();
    }
    pp_sink(Pp);
    // D = Pn - Pp (in place in Un), then factor Pp in place
    FK_UNROLL for (int e = 0; e < NX * (NX + 1) / 2; ++e) Un[e] -= Pp[e];
    double (&Lp)[NX * (NX + 1) / 2] = Pp;
    double d[NX], dinv[NX];
    if (!ldlt_packed<NX>(Lp, d, dinv)) st |= ST_NOT_PD;
    FK_STAGE();
    // K rows: solve, then the state update
    FK_UNROLL for (int r = 0; r < NX; ++r) {
        double b[NX];
        FK_UNROLL for (int c = 0; c < NX; ++c) b[c] = K[r * NX + c];
        solve_row_packed<NX>(Lp, dinv, b);
        double acc = x[r];
        FK_UNROLL for (int c = 0; c < NX; ++c) {
            K[r * NX + c] = b[c];
            acc = fma(b[c], dx[c], acc);
        }
        x[r] = acc;
        FK_STAGE();
    }
    // P += (K D) K', upper triangle
    FK_UNROLL for (int i = 0; i < NX; ++i) {
        double kd[NX];
        FK_UNROLL for (int b2 = 0; b2 < NX; ++b2) {
            double acc = K[i * NX] * Un[sym_idx<NX>(0, b2)];
            FK_UNROLL for (int a2 = 1; a2 < NX; ++a2) acc = fma(K[i * NX + a2], Un[sym_idx<NX>(a2, b2)], acc);
            kd[b2] = acc;
        }
        FK_UNROLL for (int j = 0; j < NX; ++j)
            if (j >= i) {
                double acc = U[sym_idx<NX>(i, j)];
                FK_UNROLL for (int k = 0; k < NX; ++k) acc = fma(kd[k], K[j * NX + k], acc);
                U[sym_idx<NX>(i, j)] = acc;
            }
        FK_STAGE();
    }
    return st;
}

// Cholesky factor of scale * P for a packed symmetric P: L (lower, L L' = scale P) packed with the
// same index map (L(i,j), i >= j, lives at sym_idx(j,i)).  U(k,j) = L(j,k) are the rows that
// scipy.linalg.cholesky (upper) returns (sigma_points.py:167-168).  Returns true iff SPD.
template <int NX>
FK_HD bool chol_packed(const double (&P)[NX * (NX + 1) / 2], double scale, double (&L)[NX * (NX + 1) / 2])
{
    bool pd = true;
    FK_UNROLL for (int j = 0; j < NX; ++j) {
        double d = scale * P[sym_idx<NX>(j, j)];
        FK_UNROLL for (int k = 0; k < NX; ++k)
            if (k < j) d = fma(-L[sym_idx<NX>(j, k)], L[sym_idx<NX>(j, k)], d);
        pd = pd && (d > 0.0);
        const double ljj = sqrt(d);
        L[sym_idx<NX>(j, j)] = ljj;
        const double inv = 1.0 / ljj;
        FK_UNROLL for (int i = 0; i < NX; ++i)
            if (i > j) {
                double t = scale * P[sym_idx<NX>(j, i)];
                FK_UNROLL for (int k = 0; k < NX; ++k)
                    if (k < j) t = fma(-L[sym_idx<NX>(i, k)], L[sym_idx<NX>(j, k)], t);
                L[sym_idx<NX>(i, j)] = t * inv;
            }
    }
    return pd;
}

// component c of column k of L (0 above the diagonal)
template <int NX>
FK_HD double lcol(const double (&L)[NX * (NX + 1) / 2], int c, int k)
{
    return c >= k ? L[sym_idx<NX>(c, k)] : 0.0;
}

}  // namespace fk
