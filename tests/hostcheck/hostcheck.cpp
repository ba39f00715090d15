// hostcheck.cpp -- TEST-ONLY harness: compiles the per-track arithmetic of the HIP kernels
// (filterpy_amd/csrc/fk_math.hpp, the same templates the gfx950 kernels instantiate) for the
// HOST with g++, so the build container (no GPU) can check the arithmetic against the oracle
// and the goldens before any GPU time is spent.  It is never loaded by filterpy_amd/ and is not
// part of libfilterhip.so: the product has no CPU path.
#include <stdint.h>
#include <string.h>

#include "../../filterpy_amd/csrc/fk_math.hpp"
#include "../../filterpy_amd/csrc/fk_math_sym.hpp"
#include "../../filterpy_amd/csrc/fk_imm.hpp"

using namespace fk;

// pad an r x c matrix into ROWS x COLS (diag_pad on the padded diagonal)
template <int ROWS, int COLS>
static void pad(double (&M)[ROWS * COLS], const double *src, int r, int c, double diag_pad)
{
    for (int a = 0; a < ROWS; ++a)
        for (int b = 0; b < COLS; ++b)
            M[a * COLS + b] = (a < r && b < c) ? src[a * c + b] : (a == b ? diag_pad : 0.0);
}
template <int ROWS, int COLS>
static void unpad(const double (&M)[ROWS * COLS], double *dst, int r, int c)
{
    for (int a = 0; a < r; ++a)
        for (int b = 0; b < c; ++b) dst[a * c + b] = M[a * COLS + b];
}

// One track, T steps, shared model; mirrors kf_kernel's control flow.
template <int NX, int NZ>
static int kf_batch(int n, int m, long T, const double *F, const double *Q, const double *H, const double *R,
                    const double *z, const uint8_t *mask, double *x0, double *P0, double *means, double *covs,
                    double *means_p, double *covs_p, double alpha_sq, int update_first)
{
    RegModel<NX, NZ> M;
    pad<NX, NX>(M.F, F, n, n, 1.0);
    pad<NX, NX>(M.Q, Q, n, n, 0.0);
    pad<NZ, NX>(M.H, H, m, n, 0.0);
    pad<NZ, NZ>(M.R, R, m, m, 1.0);
    double x[NX], P[NX * NX];
    pad<NX, 1>(x, x0, n, 1, 0.0);
    pad<NX, NX>(P, P0, n, n, 1.0);
    int st = 0;
    for (long t = 0; t < T; ++t) {
        double zz[NZ];
        pad<NZ, 1>(zz, z + t * m, m, 1, 0.0);
        const bool has_z = !mask || mask[t];
        for (int phase = 0; phase < 2; ++phase) {
            const bool is_predict = (phase == 0) != (update_first != 0);
            if (is_predict) {
                kf_predict<NX>(x, P, M, alpha_sq);
                unpad<NX, 1>(x, means_p + t * n, n, 1);
                unpad<NX, NX>(P, covs_p + t * n * n, n, n);
            } else {
                if (has_z) {
                    double K[NX * NZ], y[NZ], S[NZ * NZ], Lf[NZ * NZ], dinv[NZ];
                    st |= kf_update<NX, NZ>(x, P, zz, M, K, y, S, Lf, dinv);
                }
                unpad<NX, 1>(x, means + t * n, n, 1);
                unpad<NX, NX>(P, covs + t * n * n, n, n);
            }
        }
    }
    unpad<NX, 1>(x, x0, n, 1);
    unpad<NX, NX>(P, P0, n, n);
    return st;
}

template <int NX>
static int rts(int n, long T, const double *F, const double *Q, const double *Xs, const double *Ps, double *xs,
               double *Pso, double *Ko, double *Ppo)
{
    RegModel<NX, 1> M;
    pad<NX, NX>(M.F, F, n, n, 1.0);
    pad<NX, NX>(M.Q, Q, n, n, 0.0);
    double xn[NX], Pn[NX * NX];
    pad<NX, 1>(xn, Xs + (T - 1) * n, n, 1, 0.0);
    pad<NX, NX>(Pn, Ps + (T - 1) * n * n, n, n, 1.0);
    unpad<NX, 1>(xn, xs + (T - 1) * n, n, 1);
    unpad<NX, NX>(Pn, Pso + (T - 1) * n * n, n, n);
    unpad<NX, NX>(Pn, Ppo + (T - 1) * n * n, n, n);
    memset(Ko + (T - 1) * n * n, 0, sizeof(double) * n * n);
    int st = 0;
    for (long k = T - 2; k >= 0; --k) {
        double x[NX], P[NX * NX], K[NX * NX], Pp[NX * NX];
        pad<NX, 1>(x, Xs + k * n, n, 1, 0.0);
        pad<NX, NX>(P, Ps + k * n * n, n, n, 1.0);
        st |= rts_step<NX>(x, P, xn, Pn, M, K, Pp);
        unpad<NX, 1>(x, xs + k * n, n, 1);
        unpad<NX, NX>(P, Pso + k * n * n, n, n);
        unpad<NX, NX>(K, Ko + k * n * n, n, n);
        unpad<NX, NX>(Pp, Ppo + k * n * n, n, n);
        for (int i = 0; i < NX; ++i) xn[i] = x[i];
        for (int i = 0; i < NX * NX; ++i) Pn[i] = P[i];
    }
    return st;
}

template <int NX>
static int sigma(int n, double scale, const double *x0, const double *P0, double *sig)
{
    double x[NX], P[NX * NX], L[NX * NX];
    pad<NX, 1>(x, x0, n, 1, 0.0);
    pad<NX, NX>(P, P0, n, n, 1.0 / scale);
    const bool pd = chol_lower<NX>(P, scale, L);
    for (int c = 0; c < n; ++c) sig[c] = x[c];
    for (int k = 0; k < n; ++k)
        for (int c = 0; c < n; ++c) {
            sig[(k + 1) * n + c] = x[c] - (-L[c * NX + k]);
            sig[(n + k + 1) * n + c] = x[c] - L[c * NX + k];
        }
    return pd ? 0 : ST_NOT_PD;
}


// Packed-symmetric variant (fk_math_sym.hpp), exact dims only -- what kf_fast_kernel runs.
template <int NX, int NZ>
static int kf_batch_sym(long T, const double *F, const double *Q, const double *H, const double *R,
                        const double *z, const uint8_t *mask, double *x0, double *P0, double *means, double *covs,
                        double *means_p, double *covs_p, double alpha_sq)
{
    RegModel<NX, NZ> M;
    pad<NX, NX>(M.F, F, NX, NX, 1.0);
    pad<NX, NX>(M.Q, Q, NX, NX, 0.0);
    pad<NZ, NX>(M.H, H, NZ, NX, 0.0);
    pad<NZ, NZ>(M.R, R, NZ, NZ, 1.0);
    double x[NX], U[NX * (NX + 1) / 2];
    for (int i = 0; i < NX; ++i) x[i] = x0[i];
    for (int i = 0; i < NX; ++i)
        for (int j = i; j < NX; ++j) U[sym_idx<NX>(i, j)] = P0[i * NX + j];
    int st = 0;
    auto unpackP = [&](double *dst) {
        for (int i = 0; i < NX; ++i)
            for (int j = 0; j < NX; ++j) dst[i * NX + j] = U[sym_idx<NX>(i, j)];
    };
    for (long t = 0; t < T; ++t) {
        double zz[NZ];
        for (int i = 0; i < NZ; ++i) zz[i] = z[t * NZ + i];
        kf_predict_sym<NX>(x, U, M, alpha_sq);
        for (int i = 0; i < NX; ++i) means_p[t * NX + i] = x[i];
        unpackP(covs_p + t * NX * NX);
        if (!mask || mask[t]) {
            double K[NX * NZ], y[NZ], S[NZ * NZ], Lf[NZ * NZ], dinv[NZ];
            st |= kf_update_sym<NX, NZ>(x, U, zz, M, K, y, S, Lf, dinv);
        }
        for (int i = 0; i < NX; ++i) means[t * NX + i] = x[i];
        unpackP(covs + t * NX * NX);
    }
    for (int i = 0; i < NX; ++i) x0[i] = x[i];
    unpackP(P0);
    return st;
}

extern "C" int hc_kf_batch_sym(int n, int m, long T, const double *F, const double *Q, const double *H,
                               const double *R, const double *z, const uint8_t *mask, double *x0, double *P0,
                               double *means, double *covs, double *means_p, double *covs_p, double alpha_sq)
{
#define SYMCALL(NX, NZ) \
    if (n == NX && m == NZ) return kf_batch_sym<NX, NZ>(T, F, Q, H, R, z, mask, x0, P0, means, covs, means_p, covs_p, alpha_sq);
    SYMCALL(1, 1) SYMCALL(2, 1) SYMCALL(4, 2) SYMCALL(6, 3) SYMCALL(9, 3)
#undef SYMCALL
    return -1;
}

template <int NX>
static int rts_sym(long T, const double *F, const double *Q, const double *Xs, const double *Ps, double *xs,
                   double *Pso, double *Ko, double *Ppo)
{
    constexpr int PL = NX * (NX + 1) / 2;
    RegModel<NX, 1> M;
    pad<NX, NX>(M.F, F, NX, NX, 1.0);
    pad<NX, NX>(M.Q, Q, NX, NX, 0.0);
    double xn[NX], Un[PL];
    auto pack = [](const double *src, double (&dst)[PL]) {
        for (int i = 0; i < NX; ++i)
            for (int j = i; j < NX; ++j) dst[sym_idx<NX>(i, j)] = src[i * NX + j];
    };
    auto unpack = [](const double (&src)[PL], double *dst) {
        for (int i = 0; i < NX; ++i)
            for (int j = 0; j < NX; ++j) dst[i * NX + j] = src[sym_idx<NX>(i, j)];
    };
    for (int i = 0; i < NX; ++i) xn[i] = Xs[(T - 1) * NX + i];
    pack(Ps + (T - 1) * NX * NX, Un);
    for (int i = 0; i < NX; ++i) xs[(T - 1) * NX + i] = xn[i];
    unpack(Un, Pso + (T - 1) * NX * NX);
    unpack(Un, Ppo + (T - 1) * NX * NX);
    memset(Ko + (T - 1) * NX * NX, 0, sizeof(double) * NX * NX);
    int st = 0;
    for (long k = T - 2; k >= 0; --k) {
        double x[NX], U[PL], K[NX * NX], Pp[PL];
        for (int i = 0; i < NX; ++i) x[i] = Xs[k * NX + i];
        pack(Ps + k * NX * NX, U);
        st |= rts_step_sym<NX>(x, U, xn, Un, M, K, [&](const double (&Ppk)[PL]) {
            for (int e = 0; e < PL; ++e) Pp[e] = Ppk[e];
        });
        for (int i = 0; i < NX; ++i) xs[k * NX + i] = x[i];
        unpack(U, Pso + k * NX * NX);
        unpack(Pp, Ppo + k * NX * NX);
        for (int i = 0; i < NX * NX; ++i) Ko[k * NX * NX + i] = K[i];
        for (int i = 0; i < NX; ++i) xn[i] = x[i];
        for (int i = 0; i < PL; ++i) Un[i] = U[i];
    }
    return st;
}

extern "C" int hc_rts_sym(int n, long T, const double *F, const double *Q, const double *Xs, const double *Ps,
                          double *xs, double *Pso, double *Ko, double *Ppo)
{
    if (n == 1) return rts_sym<1>(T, F, Q, Xs, Ps, xs, Pso, Ko, Ppo);
    if (n == 2) return rts_sym<2>(T, F, Q, Xs, Ps, xs, Pso, Ko, Ppo);
    if (n == 4) return rts_sym<4>(T, F, Q, Xs, Ps, xs, Pso, Ko, Ppo);
    if (n == 6) return rts_sym<6>(T, F, Q, Xs, Ps, xs, Pso, Ko, Ppo);
    if (n == 9) return rts_sym<9>(T, F, Q, Xs, Ps, xs, Pso, Ko, Ppo);
    return -1;
}

#define BY_DIMS(n, m, CALL)                                   \
    if ((n) == 1 && (m) == 1) return CALL(1, 1);              \
    if ((n) == 2 && (m) == 1) return CALL(2, 1);              \
    if ((n) == 4 && (m) == 2) return CALL(4, 2);              \
    if ((n) == 6 && (m) == 3) return CALL(6, 3);              \
    if ((n) == 9 && (m) == 3) return CALL(9, 3);              \
    if ((n) <= 2 && (m) <= 2) return CALL(2, 2);              \
    if ((n) <= 4 && (m) <= 4) return CALL(4, 4);              \
    if ((n) <= 6 && (m) <= 6) return CALL(6, 6);              \
    if ((n) <= 8 && (m) <= 4) return CALL(8, 4);              \
    return CALL(16, 8);

extern "C" {

int hc_kf_batch(int n, int m, long T, const double *F, const double *Q, const double *H, const double *R,
                const double *z, const uint8_t *mask, double *x0, double *P0, double *means, double *covs,
                double *means_p, double *covs_p, double alpha_sq, int update_first)
{
#define CALL(NX, NZ) kf_batch<NX, NZ>(n, m, T, F, Q, H, R, z, mask, x0, P0, means, covs, means_p, covs_p, alpha_sq, update_first)
    BY_DIMS(n, m, CALL)
#undef CALL
}

int hc_rts(int n, long T, const double *F, const double *Q, const double *Xs, const double *Ps, double *xs,
           double *Pso, double *Ko, double *Ppo)
{
    if (n == 1) return rts<1>(n, T, F, Q, Xs, Ps, xs, Pso, Ko, Ppo);
    if (n == 2) return rts<2>(n, T, F, Q, Xs, Ps, xs, Pso, Ko, Ppo);
    if (n <= 3) return rts<3>(n, T, F, Q, Xs, Ps, xs, Pso, Ko, Ppo);
    if (n == 4) return rts<4>(n, T, F, Q, Xs, Ps, xs, Pso, Ko, Ppo);
    if (n == 6) return rts<6>(n, T, F, Q, Xs, Ps, xs, Pso, Ko, Ppo);
    if (n <= 8) return rts<8>(n, T, F, Q, Xs, Ps, xs, Pso, Ko, Ppo);
    if (n == 9) return rts<9>(n, T, F, Q, Xs, Ps, xs, Pso, Ko, Ppo);
    return rts<16>(n, T, F, Q, Xs, Ps, xs, Pso, Ko, Ppo);
}

int hc_sigma(int n, double scale, const double *x, const double *P, double *sig)
{
    if (n <= 2) return sigma<2>(n, scale, x, P, sig);
    if (n <= 4) return sigma<4>(n, scale, x, P, sig);
    if (n <= 6) return sigma<6>(n, scale, x, P, sig);
    if (n <= 8) return sigma<8>(n, scale, x, P, sig);
    return sigma<16>(n, scale, x, P, sig);
}
}

// ---------------------------------------------------------------------------------------------
// Exact parallel cumsum (filterpy_amd/csrc/fk_exact_scan.hpp): host emulation of the block-wide
// scan of resample_kernels.hip::tile_cumsum_exact with the same tile / thread / wave
// decomposition (RS_ITEMS consecutive elements per thread, Hillis-Steele over 64 lanes, serial
// wave prefix), so the monoid's associativity and the segment logic are exercised exactly as
// on the GPU.
#include "../../filterpy_amd/csrc/fk_exact_scan.hpp"
#include <limits>
#include <vector>

namespace {
constexpr int T_THREADS = 256, T_ITEMS = 8, T_TILE = T_THREADS * T_ITEMS;

double tile_cumsum_emul(double *w, int len, double carry, bool &started, long *n_segments, int &prelude)
{
    int pos = 0;
    while (pos < len) {
        const bool finite_pos = started && carry > 0.0 && carry <= 1.79769313486231570815e+308;
        if (!finite_pos) {
            if (!started || carry == 0.0) {
                int j0 = len;
                for (int j = pos; j < len; ++j)
                    if (w[j] != 0.0) { j0 = j; break; }
                if (j0 < len) {
                    carry = started ? carry + w[j0] : w[j0];
                    started = true;
                    pos = j0 + 1;
                } else {
                    started = started || len > pos;
                    pos = len;
                }
                continue;
            }
            carry = carry + w[pos];
            w[pos] = carry;
            ++pos;
            continue;
        }
        if (prelude > 0) {
            const int stop = (pos + prelude < len) ? pos + prelude : len;
            for (int j = pos; j < stop; ++j) {
                carry = carry + w[j];
                w[j] = carry;
            }
            prelude -= stop - pos;
            pos = stop;
            continue;
        }
        ++*n_segments;
        const double u = ulp_of(carry);
        const int eu = ulp_exp(carry);
        if (scale2(1.0, eu) != u) return carry / 0.0 * 0.0;   // self-check: poisons the result
        // fast path of the kernel (fast_inc, fk_exact_scan.hpp): no half-ulp tie in [pos, len) -> the
        // increments are summed as doubles, thread by thread and then across threads in a different
        // association order than the Mono scan (all exact below 2^53)
        {
            const double C0d = scale2(carry, -eu);
            std::vector<double> incl(T_TILE), run_t(T_THREADS);
            bool any_tie = false;
            for (int t = 0; t < T_THREADS; ++t) {
                double run = 0.0;
                for (int k = 0; k < T_ITEMS; ++k) {
                    const int j = t * T_ITEMS + k;
                    bool tk = false;
                    const double e = (j >= pos && j < len) ? fast_inc(w[j], eu, tk) : 0.0;
                    any_tie = any_tie || tk;
                    run += e;
                    incl[j] = run;
                }
                run_t[t] = run;
            }
            if (!any_tie) {
                int cross = len;
                std::vector<double> Cd(T_TILE);
                double excl = 0.0;
                for (int t = 0; t < T_THREADS; ++t) {
                    for (int k = 0; k < T_ITEMS; ++k) {
                        const int j = t * T_ITEMS + k;
                        Cd[j] = C0d + (excl + incl[j]);
                        if (j >= pos && j < len && !(Cd[j] < 0x1p53) && j < cross) cross = j;
                    }
                    excl += run_t[t];
                }
                for (int j = pos; j < cross; ++j) w[j] = Cd[j] * u;
                if (cross > pos) carry = w[cross - 1];
                if (cross < len) {
                    carry = carry + w[cross];
                    w[cross] = carry;
                    pos = cross + 1;
                } else {
                    pos = len;
                }
                continue;
            }
        }
        const long long C0 = (long long)scale2(carry, -eu);
        std::vector<Mono> loc(T_TILE), tot(T_THREADS), inc(T_THREADS), excl(T_THREADS);
        for (int t = 0; t < T_THREADS; ++t) {
            Mono run = mono_identity();
            for (int k = 0; k < T_ITEMS; ++k) {
                const int j = t * T_ITEMS + k;
                const Mono e = (j >= pos && j < len) ? mono_elem(w[j], u, eu) : mono_identity();
                run = mono_compose(run, e);
                loc[j] = run;
            }
            tot[t] = run;
        }
        // Hillis-Steele inside each wave of 64
        inc = tot;
        for (int d = 1; d < 64; d <<= 1) {
            std::vector<Mono> nxt = inc;
            for (int t = 0; t < T_THREADS; ++t)
                if ((t & 63) >= d) nxt[t] = mono_compose(inc[t - d], inc[t]);
            inc = nxt;
        }
        for (int t = 0; t < T_THREADS; ++t) {
            Mono e = (t & 63) ? inc[t - 1] : mono_identity();
            Mono wp = mono_identity();
            for (int wv = 0; wv < (t >> 6); ++wv) wp = mono_compose(wp, inc[wv * 64 + 63]);
            excl[t] = mono_compose(wp, e);
        }
        int cross = len;
        std::vector<long long> Cj(T_TILE);
        for (int j = 0; j < T_TILE; ++j) {
            Cj[j] = mono_apply(C0, mono_compose(excl[j / T_ITEMS], loc[j]));
            if (j >= pos && j < len && Cj[j] >= MONO_LIMIT && j < cross) cross = j;
        }
        for (int j = pos; j < cross; ++j) w[j] = (double)Cj[j] * u;
        if (cross > pos) carry = w[cross - 1];
        if (cross < len) {
            carry = carry + w[cross];
            w[cross] = carry;
            pos = cross + 1;
        } else {
            pos = len;
        }
    }
    return carry;
}
}  // namespace

extern "C" long hc_cumsum_exact(long N, const double *w, double *cs)
{
    double carry = 0.0;
    bool started = false;
    long segs = 0;
    int prelude = 128;
    std::vector<double> tile(T_TILE);
    for (long base = 0; base < N; base += T_TILE) {
        const int len = (int)((N - base) < T_TILE ? (N - base) : T_TILE);
        for (int j = 0; j < T_TILE; ++j) tile[j] = j < len ? w[base + j] : 0.0;
        carry = tile_cumsum_emul(tile.data(), len, carry, started, &segs, prelude);
        for (int j = 0; j < len; ++j) cs[base + j] = tile[j];
    }
    return segs;
}


// The output loop of resample_kernel / resample_chunk_kernel on one tile: the tile's cumulative sums
// (cs, from carry-in c_in), the slot positions ps, and fk::tile_upper_bound exactly as the kernels call it.
extern "C" void hc_tile_search(int len, const double *cs, double c_in, long n_pos, const double *ps, int *out)
{
    const double inf = std::numeric_limits<double>::infinity();
    std::vector<double> g(1 + len + fk::TILE_GUARD, inf);      // guarded like ScanShared::tile
    g[0] = -inf;
    std::copy(cs, cs + len, g.begin() + 1);
    const double inv_span = (double)len / (cs[len - 1] - c_in);
    for (long i = 0; i < n_pos; ++i) out[i] = fk::tile_upper_bound(g.data() + 1, len, ps[i], c_in, inv_span);
}



// --------------------------------------------------------------------------------- IMM --
// One track's bank of nm filters, T x { predict; update }, padded like imm_kernel.
template <int NX, int NZ, int NM>
static int imm_batch(int n, int m, long T, const double *F, const double *Q, const double *H, const double *R,
                     const double *Mt, const double *z, double *xs0, double *Ps0, double *mu0, double *x_out,
                     double *P_out, double *mu_out, double *xp_out, double *Pp_out, double *L_out, int mmae)
{
    constexpr int PL = NX * (NX + 1) / 2;
    RegModel<NX, NZ> mods[NM];
    double xs[NM][NX], Ps[NM][PL], mu[NM];
    for (int j = 0; j < NM; ++j) {
        pad<NX, NX>(mods[j].F, F + j * n * n, n, n, 1.0);
        pad<NX, NX>(mods[j].Q, Q + j * n * n, n, n, 0.0);
        pad<NZ, NX>(mods[j].H, H + j * m * n, m, n, 0.0);
        pad<NZ, NZ>(mods[j].R, R + j * m * m, m, m, 1.0);
        double Pf[NX * NX];
        pad<NX, 1>(xs[j], xs0 + j * n, n, 1, 0.0);
        pad<NX, NX>(Pf, Ps0 + j * n * n, n, n, 1.0);
        for (int r = 0; r < NX; ++r)
            for (int c = r; c < NX; ++c) Ps[j][sym_idx<NX>(r, c)] = Pf[r * NX + c];
        mu[j] = mu0[j];
    }
    int st = 0;
    for (long t = 0; t < T; ++t) {
        double zz[NZ], cbar[NM], L[NM], x[NX], P[NX * NX];
        pad<NZ, 1>(zz, z + t * m, m, 1, 0.0);
        if (mmae) {   // mirrors imm_kernel's MMAE branch
            for (int j = 0; j < NM; ++j) cbar[j] = mu[j];
            for (int j = 0; j < NM; ++j) kf_predict_sym<NX>(xs[j], Ps[j], mods[j], 1.0);
        } else {
            imm_mixing_cbar<NM>(mu, Mt, cbar);
            imm_predict<NX, NM>(xs, Ps, mu, cbar, Mt, mods);
            imm_estimate<NX, NM>(xs, Ps, mu, x, P);
            unpad<NX, 1>(x, xp_out + t * n, n, 1);
            unpad<NX, NX>(P, Pp_out + t * n * n, n, n);
        }
        st |= imm_update<NX, NZ, NM>(xs, Ps, mu, cbar, zz, m, mods, L);
        if (mmae) mmae_estimate<NX, NM>(xs, Ps, mu, n, x, P);
        else imm_estimate<NX, NM>(xs, Ps, mu, x, P);
        unpad<NX, 1>(x, x_out + t * n, n, 1);
        unpad<NX, NX>(P, P_out + t * n * n, n, n);
        for (int j = 0; j < NM; ++j) { mu_out[t * NM + j] = mu[j]; L_out[t * NM + j] = L[j]; }
    }
    for (int j = 0; j < NM; ++j) {
        double Pf[NX * NX];
        for (int r = 0; r < NX; ++r)
            for (int c = 0; c < NX; ++c) Pf[r * NX + c] = Ps[j][sym_idx<NX>(r, c)];
        unpad<NX, 1>(xs[j], xs0 + j * n, n, 1);
        unpad<NX, NX>(Pf, Ps0 + j * n * n, n, n);
        mu0[j] = mu[j];
    }
    return st;
}

extern "C" int hc_imm_batch(int n, int m, int nm, long T, const double *F, const double *Q, const double *H,
                            const double *R, const double *Mt, const double *z, double *xs0, double *Ps0,
                            double *mu0, double *x_out, double *P_out, double *mu_out, double *xp_out,
                            double *Pp_out, double *L_out, int mmae)
{
#define GO(NXV, NZV, NMV) \
    return imm_batch<NXV, NZV, NMV>(n, m, T, F, Q, H, R, Mt, z, xs0, Ps0, mu0, x_out, P_out, mu_out, xp_out, Pp_out, L_out, mmae)
    const int cls = (n <= 2 && m <= 1) ? 0 : (n <= 4 && m <= 2) ? 1 : 2;
    if (nm == 2) {
        if (cls == 0) GO(2, 1, 2);
        if (cls == 1) GO(4, 2, 2);
        GO(6, 3, 2);
    }
    if (nm == 3) {
        if (cls == 0) GO(2, 1, 3);
        if (cls == 1) GO(4, 2, 3);
        GO(6, 3, 3);
    }
#undef GO
    return -1;
}


// ---------------------------------------------------------------------------------------------------------
// The fused linear UKF step (filterpy_amd/csrc/fk_ukf.hpp, the arithmetic of ukf_kernels.hip) on the host: T x { predict; update } for one track with exact dims, so that its arithmetic
// can be held against the oracle without a GPU.
#include "../../filterpy_amd/csrc/fk_ukf.hpp"

namespace {
template <int NX, int NZ, bool V4 = false>
int ukf_v3_batch(long T, const double *F, const double *H, const double *Q, const double *R, const double *Wm,
                 const double *Wc, double scale, const double *zs, const unsigned char *mask, double *x0,
                 double *P0, double *means, double *covs)
{
    constexpr int PL = NX * (NX + 1) / 2, KS = 2 * NX + 1;
    struct View {
        fk::RegModel<NX, NZ> sm;
        const double *Wm, *Wc, *Wp;
    };
    View v;
    std::copy(F, F + NX * NX, v.sm.F);
    std::copy(Q, Q + NX * NX, v.sm.Q);
    std::copy(H, H + NZ * NX, v.sm.H);
    std::copy(R, R + NZ * NZ, v.sm.R);
    double wm[KS], wc[KS], wp[2 + NX];
    std::copy(Wm, Wm + KS, wm);
    std::copy(Wc, Wc + KS, wc);
    fk::make_pair_table<NX>(wm, wc, wp);
    if (V4 && !fk::pair_weights_symmetric<NX>(wm, wc)) return -2;
    v.Wm = wm;
    v.Wc = wc;
    v.Wp = wp;
    auto fresh = [&](double = 0.0) -> const View & { return v; };
    double x[NX], P[PL];
    for (int i = 0; i < NX; ++i) {
        x[i] = x0[i];
        for (int j = i; j < NX; ++j) P[fk::sym_idx<NX>(i, j)] = P0[i * NX + j];
    }
    int st = 0;
    for (long t = 0; t < T; ++t) {
        auto load_z = [&](double (&z)[NZ]) {
            for (int r = 0; r < NZ; ++r) z[r] = zs[t * NZ + r];
        };
        if (V4) st |= fk::ukf_linear_step_v4<NX, NZ>(x, P, load_z, mask ? mask[t] != 0 : true, scale, fresh);
        else st |= fk::ukf_linear_step_v3<NX, NZ>(x, P, load_z, mask ? mask[t] != 0 : true, scale, fresh);
        for (int i = 0; i < NX; ++i) {
            means[t * NX + i] = x[i];
            for (int j = 0; j < NX; ++j) covs[(t * NX + i) * NX + j] = P[fk::sym_idx<NX>(i, j)];
        }
    }
    for (int i = 0; i < NX; ++i) {
        x0[i] = x[i];
        for (int j = 0; j < NX; ++j) P0[i * NX + j] = P[fk::sym_idx<NX>(i, j)];
    }
    return st;
}
}  // namespace

namespace {
// the fused linear-model UKF smoother's step (fk_ukf.hpp: ukf_linear_rts_gain_v3 / _correct) over a whole backward pass
template <int NX, bool V4 = false>
int ukf_rts_batch(long T, const double *F, const double *Q, const double *Wm, const double *Wc, double scale,
                  const double *Xs, const double *Ps, double *xs, double *ps, double *Ks)
{
    constexpr int PL = NX * (NX + 1) / 2, KS = 2 * NX + 1;
    struct View {
        fk::RegModel<NX, 1> sm;
        const double *Wm, *Wc, *Wp;
    };
    View v;
    std::copy(F, F + NX * NX, v.sm.F);
    std::copy(Q, Q + NX * NX, v.sm.Q);
    double wm[KS], wc[KS], wp[2 + NX];
    std::copy(Wm, Wm + KS, wm);
    std::copy(Wc, Wc + KS, wc);
    fk::make_pair_table<NX>(wm, wc, wp);
    if (V4 && !fk::pair_weights_symmetric<NX>(wm, wc)) return -2;
    v.Wm = wm;
    v.Wc = wc;
    v.Wp = wp;
    auto fresh = [&](double = 0.0) -> const View & { return v; };
    auto load = [&](long t, double (&x)[NX], double (&P)[PL]) {
        for (int i = 0; i < NX; ++i) {
            x[i] = Xs[t * NX + i];
            for (int j = i; j < NX; ++j) P[fk::sym_idx<NX>(i, j)] = Ps[(t * NX + i) * NX + j];
        }
    };
    auto store = [&](long t, const double (&x)[NX], const double (&P)[PL]) {
        for (int i = 0; i < NX; ++i) {
            xs[t * NX + i] = x[i];
            for (int j = 0; j < NX; ++j) ps[(t * NX + i) * NX + j] = P[fk::sym_idx<NX>(i, j)];
        }
    };
    double xn[NX], Pn[PL];
    load(T - 1, xn, Pn);
    for (int i = 0; i < NX; ++i) xs[(T - 1) * NX + i] = Xs[(T - 1) * NX + i];
    for (int e = 0; e < NX * NX; ++e) ps[(T - 1) * NX * NX + e] = Ps[(T - 1) * NX * NX + e];     // copied as it is
    for (int e = 0; e < NX * NX; ++e) Ks[(T - 1) * NX * NX + e] = 0.0;
    int st = 0;
    for (long t = T - 2; t >= 0; --t) {
        double x[NX], P[PL], K[NX * NX], xb[NX], Pb[PL];
        load(t, x, P);
        if (V4) st |= fk::ukf_linear_rts_gain_v4<NX>(x, P, scale, xb, Pb, K, fresh);
        else st |= fk::ukf_linear_rts_gain_v3<NX>(x, P, scale, xb, Pb, K, fresh);
        fk::ukf_linear_rts_correct<NX>(x, P, xn, Pn, xb, Pb, K);
        store(t, x, P);
        for (int e = 0; e < NX * NX; ++e) Ks[t * NX * NX + e] = K[e];
        for (int i = 0; i < NX; ++i) xn[i] = x[i];
        for (int e = 0; e < PL; ++e) Pn[e] = P[e];
    }
    return st;
}
}  // namespace

// the factor-image organisation of the step (fk_ukf.hpp, ukf_linear_step_v3: what the kernels run since round 3)
extern "C" int hc_ukf_linear_v3(int n, int m, long T, const double *F, const double *H, const double *Q, const double *R,
                                const double *Wm, const double *Wc, double scale, const double *zs,
                                const unsigned char *mask, double *x0, double *P0, double *means, double *covs)
{
#define GO(NXV, NZV) if (n == NXV && m == NZV) return ukf_v3_batch<NXV, NZV>(T, F, H, Q, R, Wm, Wc, scale, zs, mask, x0, P0, means, covs)
    GO(2, 2); GO(4, 2); GO(6, 3); GO(8, 4); GO(9, 3); GO(9, 4); GO(3, 1); GO(5, 2); GO(7, 3);
#undef GO
    return -1;
}

extern "C" int hc_ukf_linear_rts_v3(int n, long T, const double *F, const double *Q, const double *Wm, const double *Wc,
                                    double scale, const double *Xs, const double *Ps, double *xs, double *ps, double *Ks)
{
#define GO(NXV) if (n == NXV) return ukf_rts_batch<NXV>(T, F, Q, Wm, Wc, scale, Xs, Ps, xs, ps, Ks)
    GO(2); GO(3); GO(4); GO(5); GO(6); GO(7); GO(8); GO(9);
#undef GO
    return -1;
}

// the pair-regrouped steps (fk_ukf.hpp, ukf_linear_step_v4 / ukf_linear_rts_gain_v4: what the kernels run since round 4 for
// weights equal within every +- pair)
extern "C" int hc_ukf_linear_v4(int n, int m, long T, const double *F, const double *H, const double *Q, const double *R,
                                const double *Wm, const double *Wc, double scale, const double *zs,
                                const unsigned char *mask, double *x0, double *P0, double *means, double *covs)
{
#define GO(NXV, NZV) if (n == NXV && m == NZV) return ukf_v3_batch<NXV, NZV, true>(T, F, H, Q, R, Wm, Wc, scale, zs, mask, x0, P0, means, covs)
    GO(2, 2); GO(4, 2); GO(6, 3); GO(8, 4); GO(9, 3); GO(9, 4); GO(3, 1); GO(5, 2); GO(7, 3);
#undef GO
    return -1;
}

extern "C" int hc_ukf_linear_rts_v4(int n, long T, const double *F, const double *Q, const double *Wm, const double *Wc,
                                    double scale, const double *Xs, const double *Ps, double *xs, double *ps, double *Ks)
{
#define GO(NXV) if (n == NXV) return ukf_rts_batch<NXV, true>(T, F, Q, Wm, Wc, scale, Xs, Ps, xs, ps, Ks)
    GO(2); GO(3); GO(4); GO(5); GO(6); GO(7); GO(8); GO(9);
#undef GO
    return -1;
}

// ln |S| and |S|^(-1/2) from the reciprocal pivots (fk_math.hpp: logdet_from_dinv, rsqrt_det_from_dinv -- the mantissa /
// exponent bookkeeping is the same code on the host; only the root's refinement differs)
extern "C" void hc_det_from_dinv(int m, const double *dinv, double *logdet, double *rsqrt_det)
{
    double d[4] = {1.0, 1.0, 1.0, 1.0};
    for (int i = 0; i < m && i < 4; ++i) d[i] = dinv[i];
    *logdet = fk::logdet_from_dinv<4>(d, m);
    *rsqrt_det = fk::rsqrt_det_from_dinv<4>(d, m);
}
