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

#ifndef FK_NX
#error "compile with -DFK_NX= -DFK_NZ= -DFK_NM= -DFK_IMM_WAVES= (see Makefile, fk_dims_imm.def)"
#endif

namespace fk {


// OUTS: which per-step outputs this instantiation writes -- bit 0: x_out, P_out, mu_out; bit 1: the
// prior estimate; bit 2: likelihoods -- unconditionally (a store under a run-time pointer test costs
// the whole kernel its register allocation); OUTS < 0: tested at run time (any subset).
template <int NX, int NZ, int NM, int LAYOUT, bool EXACT, int OUTS>
__global__ void __launch_bounds__(BLOCK, (EXACT && OUTS >= 0) ? FK_IMM_WAVES : 1)
imm_kernel(const ImmArgs a)
{
    using LM = LdsModel<NX, NZ>;
    __shared__ double smem[NM * LM::SIZE + NM * NM];
    const int n = EXACT ? NX : a.n, m = EXACT ? NZ : a.m;   // EXACT: no padding guards, no branches
    const long N = a.N;
    FK_UNROLL for (int j = 0; j < NM; ++j) {
        double *s = smem + j * LM::SIZE;
        lds_fill<NX, NX>(s + LM::OFF_F, a.F + (long)j * n * n, n, n, 1.0, threadIdx.x);
        lds_fill<NX, NX>(s + LM::OFF_Q, a.Q + (long)j * n * n, n, n, 0.0, threadIdx.x);
        lds_fill<NZ, NX>(s + LM::OFF_H, a.H + (long)j * m * n, m, n, 0.0, threadIdx.x);
        lds_fill<NZ, NZ>(s + LM::OFF_R, a.R + (long)j * m * m, m, m, 1.0, threadIdx.x);
    }
    if (threadIdx.x < NM * NM) smem[NM * LM::SIZE + threadIdx.x] = a.Mt ? a.Mt[threadIdx.x] : 0.0;
    __syncthreads();
    const double *sM = smem + NM * LM::SIZE;

    Lane ln{a.i0 + (long)blockIdx.x * BLOCK, threadIdx.x, N};
    const bool live = ln.blk0 + ln.tid < a.i0 + a.cnt;
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
    // missing measurements (general kernel only): the log-density of a zero residual under each filter's last S
    constexpr bool MASKED = !(EXACT && OUTS >= 0);
    double ll0[NM];
    // before any update S = 0: the reference's density of y = 0 under it evaluates to 0 and is floored at DBL_MIN
    // (kalman_filter.py:1221-1225; frozen in tests/golden/imm_missing.npz, first step)
    FK_UNROLL for (int j = 0; j < NM; ++j) ll0[j] = -__builtin_inf();
    if (MASKED && a.ll0) {
        const RecView<LAYOUT> vl(a.ll0, ln, NM);
        FK_UNROLL for (int j = 0; j < NM; ++j) ll0[j] = vl.load(j);
    }

    // z is carried from step to step (round 4): z[t+1] is requested at the TOP of step t, in front of that step's stores.
    // vmcnt retires in order: a load issued at the top of step t+1 queues behind every store of step t and the wave sits
    // there until they have drained -- once per step (the fused UKF lost 40 % of its wave time to exactly this wait).
    // Requested a step ahead, it is waited for with vmcnt(#stores of the step) in the instantiations whose outputs are
    // compile-time (the general kernel's stores are conditional: its wait stays conservative).
    // ... which is what the ISA of the compiled-output instantiations did NOT show: the carried registers were waited for with
    // vmcnt(2) / (1) / (0) at the END of every step, behind all of its stores (88 at (6,3) x 2).  Those instantiations now
    // fetch the measurement by LDS-DMA (LaneRecordDma, fk_device.hpp): requested at the top of step t into one of two LDS
    // images, read at the top of step t + 1 behind s_waitcnt vmcnt(k), k = the step's store instructions (at most 63) --
    // hipcc does not see these loads, so the wait is exactly what the in-order counter needs and no more.
    constexpr bool ZDMA = EXACT && OUTS >= 0;
    // store instructions of one step -- a LOWER bound (the wait must never name more operations than a step issues behind the
    // request): NumPy-order records leave as 16-byte pairs, and the NM mode probabilities / likelihoods of a NumPy-order bank
    // are adjacent 8-byte stores the compiler is free to merge into pairs too (at (2,1) x 3 the count was exact,
    // margin zero).  tests/test_host_logic.py reads K and the stores behind the request off every built instantiation.
    constexpr int ST_REC = LAYOUT == LAYOUT_AOS ? NX / 2 + NX * NX / 2 : NX + NX * NX;
    constexpr int ST_NM = LAYOUT == LAYOUT_AOS ? (NM + 1) / 2 : NM;
    constexpr int ST_STEP = ((OUTS & 1) ? ST_REC + ST_NM : 0) + ((OUTS & 2) ? ST_REC : 0) + ((OUTS & 4) ? ST_NM : 0);
    constexpr int ZWAIT = ST_STEP < 63 ? ST_STEP : 63;
    __shared__ double s_z[ZDMA ? (BLOCK / 64) * 2 * LaneRecordDma<NZ, LAYOUT>::IMG_DOUBLES : 1];
    LaneRecordDma<NZ, LAYOUT> zdma;
    double zc[NZ];
    if constexpr (ZDMA) {
        // (the bank's state lands HERE, visibly to the compiler: left pending across the loop header, every first use inside
        //  the loop gets a counted wait that also has to hold on the back edge -- i.e. drains the step's stores)
        FK_UNROLL for (int j = 0; j < NM; ++j) {
            asm volatile("" ::"v"(mu[j]));
            FK_UNROLL for (int r = 0; r < NX; ++r) asm volatile("" ::"v"(xs[j][r]));
            FK_UNROLL for (int e = 0; e < PL; ++e) asm volatile("" ::"v"(Ps[j][e]));
        }
        zdma.init(s_z + wave_index() * (2 * LaneRecordDma<NZ, LAYOUT>::IMG_DOUBLES), (unsigned)(ln.blk0 + ln.tid), (unsigned)N, threadIdx.x & 63u);
        zdma.request(a.z, (unsigned)N * (unsigned)NZ * 8u, 0u);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        const RecView<LAYOUT> vz(a.z, ln, m);
        FK_UNROLL for (int r = 0; r < NZ; ++r) zc[r] = (r < m) ? vz.load(r) : 0.0;
        FK_UNROLL for (int r = 0; r < NZ; ++r) asm volatile("" ::"v"(zc[r]));       // landed before the loop
    }
    for (long t = 0; t < a.T; ++t) {
        double z[NZ];
        if constexpr (ZDMA) {
            if (t > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ZWAIT) : "memory");
            zdma.read((unsigned)(t & 1), z);
            const long tn = t + 1 < a.T ? t + 1 : t;
            zdma.request(a.z + tn * N * NZ, (unsigned)N * (unsigned)NZ * 8u, (unsigned)((t + 1) & 1));
        } else {
            FK_UNROLL for (int r = 0; r < NZ; ++r) z[r] = zc[r];
            long tn = t + 1 < a.T ? t + 1 : t;
            asm volatile("" : "+s"(tn));      // opaque: keeps the compiler from re-deriving this load one iteration later
            const RecView<LAYOUT> vz(a.z + tn * N * m, ln, m);
            FK_UNROLL for (int r = 0; r < NZ; ++r) zc[r] = (r < m) ? vz.load(r) : 0.0;
        }
        constexpr bool GENERAL = !(EXACT && OUTS >= 0);     // only the general kernel carries MMAE
        const bool mmae = GENERAL && a.mmae;
        double cbar[NM];
        if (mmae) {
            FK_UNROLL for (int j = 0; j < NM; ++j) cbar[j] = mu[j];      // p_i *= likelihood_i (mmae.py:186-187)
        } else {
            imm_mixing_cbar<NM>(mu, sM, cbar);
        }
        if (a.phase != FK_IMM_UPDATE) {
        if (mmae) {
            FK_UNROLL for (int j = 0; j < NM; ++j) kf_predict_sym<NX>(xs[j], Ps[j], mods[j], 1.0);   // mmae.py:153-154
        } else {
            imm_predict<NX, NM>(xs, Ps, mu, cbar, sM, mods);
        }
        if (MASKED && a.nu > 0) {
            // every filter's predict(u): x = F x + B u (kalman_filter.py:472-475), B u formed on its own like dot(B, u)
            const RecView<LAYOUT> vu(a.u + t * N * a.nu, ln, a.nu);
            double uu[4];
            FK_UNROLL for (int c = 0; c < 4; ++c) uu[c] = c < a.nu ? vu.load(c) : 0.0;
            FK_UNROLL for (int j = 0; j < NM; ++j)
                FK_UNROLL for (int r = 0; r < NX; ++r) {
                    if (r < n) {
                        double bu = a.B[(j * n + r) * a.nu] * uu[0];
                        FK_UNROLL for (int c = 1; c < 4; ++c)
                            if (c < a.nu) bu = fma(a.B[(j * n + r) * a.nu + c], uu[c], bu);
                        xs[j][r] += bu;
                    }
                }
        }
        if (OUTS < 0 ? (a.xp_out || a.Pp_out) : (OUTS & 2) != 0) {
            double x[NX], P[NX * NX];
            imm_estimate<NX, NM>(xs, Ps, mu, x, P);
            if (OUTS >= 0 || a.xp_out) store_rec<NX, 1, LAYOUT, EXACT>(x, a.xp_out + t * N * n, ln, n, 1);
            if (OUTS >= 0 || a.Pp_out) store_rec<NX, NX, LAYOUT, EXACT>(P, a.Pp_out + t * N * n * n, ln, n, n);
        }
        }
        if (a.phase == FK_IMM_PREDICT) break;
        double L[NM];
        bool has_z = true;
        if (MASKED && a.mask) has_z = a.mask[t * N + ln.blk0 + ln.tid] != 0;
        if (has_z) {
            st |= imm_update<NX, NZ, NM>(xs, Ps, mu, cbar, z, m, mods, L, MASKED ? ll0 : nullptr);
        } else {
            // IMMEstimator.update(None) / MMAEFilterBank.update(None): the filters keep x, P; each likelihood is the
            // density of a zero residual under the S of that filter's last real update (IMM.py:171-179 reads
            // f.likelihood after f.update(None): kalman_filter.py:511-520 set y = 0 and cleared the cache)
            double sum = 0.0;
            FK_UNROLL for (int j = 0; j < NM; ++j) {
                double lj = exp(ll0[j]);
                if (lj == 0.0) lj = 2.2250738585072014e-308;
                L[j] = lj;
                mu[j] = cbar[j] * lj;
                sum += mu[j];
            }
            FK_UNROLL for (int j = 0; j < NM; ++j) mu[j] /= sum;
        }
        if (OUTS < 0 ? (a.x_out || a.P_out) : (OUTS & 1) != 0) {
            double x[NX], P[NX * NX];
            if (mmae) mmae_estimate<NX, NM>(xs, Ps, mu, n, x, P);
            else imm_estimate<NX, NM>(xs, Ps, mu, x, P);
            if (OUTS >= 0 || a.x_out) store_rec<NX, 1, LAYOUT, EXACT>(x, a.x_out + t * N * n, ln, n, 1);
            if (OUTS >= 0 || a.P_out) store_rec<NX, NX, LAYOUT, EXACT>(P, a.P_out + t * N * n * n, ln, n, n);
        }
        if (OUTS < 0 ? a.mu_out != nullptr : (OUTS & 1) != 0) {
            const RecView<LAYOUT> v(a.mu_out + t * N * NM, ln, NM);
            FK_UNROLL for (int j = 0; j < NM; ++j) v.store(j, mu[j]);
        }
        if (OUTS < 0 ? a.L_out != nullptr : (OUTS & 4) != 0) {
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
        if (MASKED && a.ll0) {
            const RecView<LAYOUT> vl(a.ll0, ln, NM);
            FK_UNROLL for (int j = 0; j < NM; ++j) vl.store(j, ll0[j]);
        }
        if (a.status) {
            const int sv = st | (fin ? 0 : ST_NONFINITE);
            a.status[ln.blk0 + ln.tid] = a.status_or ? (a.status[ln.blk0 + ln.tid] | sv) : sv;
        }
    }
}

}  // namespace fk

using namespace fk;

// One object per (layout, compiled output set): FK_IMM_PART = 4 * layout + set (set 0..2: the exact dims with OUTS bits 0 / 1 /
// 7, set 3: the padded any-dims kernel) -- the unrolled (9,4) x 4 instantiation alone compiled for six of the library's 8.5
// minutes as one object; its eight kernels now build side by side.  Part 0 also holds the launcher.
#ifndef FK_IMM_PART
#error "compile with -DFK_IMM_PART=0..7"
#endif
#define FK_CAT5_(a, b, c, d, e) a##b##_##c##_##d##_p##e
#define FK_CAT5(a, b, c, d, e) FK_CAT5_(a, b, c, d, e)
#define FK_CAT_(a, b, c, d) a##b##_##c##_##d
#define FK_CAT(a, b, c, d) FK_CAT_(a, b, c, d)

#if defined(FK_IMM_GENERAL_ONLY) && (FK_IMM_PART % 4) != 3
// Classes built UNROLLED beyond (9,4) x 4 -- (9,4) x 5..8, (16,8) x 2: static scratch offsets instead of dynamically indexed
// scratch arrays, 7-9 x faster (profiles/r05/imm/), at 4-9 minutes of compile per kernel -- carry only the general kernel (any
// dims of the class, any output set: parts 3 and 7); the compiled-output-set parts forward to it.
void FK_CAT5(launch_imm_, FK_NX, FK_NZ, FK_NM, 3)(const ImmArgs &, hipStream_t);
void FK_CAT5(launch_imm_, FK_NX, FK_NZ, FK_NM, 7)(const ImmArgs &, hipStream_t);
void FK_CAT5(launch_imm_, FK_NX, FK_NZ, FK_NM, FK_IMM_PART)(const ImmArgs &a, hipStream_t s)
{
    if (FK_IMM_PART / 4) FK_CAT5(launch_imm_, FK_NX, FK_NZ, FK_NM, 7)(a, s);
    else FK_CAT5(launch_imm_, FK_NX, FK_NZ, FK_NM, 3)(a, s);
}
#else
void FK_CAT5(launch_imm_, FK_NX, FK_NZ, FK_NM, FK_IMM_PART)(const ImmArgs &a, hipStream_t s)
{
    constexpr int NX = FK_NX, NZ = FK_NZ, NM = FK_NM;
    constexpr int LAYOUT = (FK_IMM_PART / 4) ? LAYOUT_AOS : LAYOUT_SOA;
    constexpr int SET = FK_IMM_PART % 4;
    const dim3 grid((unsigned)((a.cnt + BLOCK - 1) / BLOCK)), block(BLOCK);
    if constexpr (SET == 0) hipLaunchKernelGGL((imm_kernel<NX, NZ, NM, LAYOUT, true, 0>), grid, block, 0, s, a);
    else if constexpr (SET == 1) hipLaunchKernelGGL((imm_kernel<NX, NZ, NM, LAYOUT, true, 1>), grid, block, 0, s, a);
    else if constexpr (SET == 2) hipLaunchKernelGGL((imm_kernel<NX, NZ, NM, LAYOUT, true, 7>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((imm_kernel<NX, NZ, NM, LAYOUT, false, -1>), grid, block, 0, s, a);
}
#endif

#if FK_IMM_PART == 0
#define FK_IMM_DECL(k) void FK_CAT5(launch_imm_, FK_NX, FK_NZ, FK_NM, k)(const ImmArgs &, hipStream_t);
FK_IMM_DECL(1) FK_IMM_DECL(2) FK_IMM_DECL(3) FK_IMM_DECL(4) FK_IMM_DECL(5) FK_IMM_DECL(6) FK_IMM_DECL(7)
// launch_imm_<NX>_<NZ>_<NM>: mask = OUTS bits when the outputs form one of the compiled sets, else -1
void FK_CAT(launch_imm_, FK_NX, FK_NZ, FK_NM)(const ImmArgs &a, int layout, int mask, hipStream_t s)
{
    const bool exact = a.n == FK_NX && a.m == FK_NZ;
    const int set = (exact && mask == 0) ? 0 : (exact && mask == 1) ? 1 : (exact && mask == 7) ? 2 : 3;
    switch ((layout == FK_LAYOUT_SOA ? 0 : 4) + set) {
        case 0: FK_CAT5(launch_imm_, FK_NX, FK_NZ, FK_NM, 0)(a, s); break;
        case 1: FK_CAT5(launch_imm_, FK_NX, FK_NZ, FK_NM, 1)(a, s); break;
        case 2: FK_CAT5(launch_imm_, FK_NX, FK_NZ, FK_NM, 2)(a, s); break;
        case 3: FK_CAT5(launch_imm_, FK_NX, FK_NZ, FK_NM, 3)(a, s); break;
        case 4: FK_CAT5(launch_imm_, FK_NX, FK_NZ, FK_NM, 4)(a, s); break;
        case 5: FK_CAT5(launch_imm_, FK_NX, FK_NZ, FK_NM, 5)(a, s); break;
        case 6: FK_CAT5(launch_imm_, FK_NX, FK_NZ, FK_NM, 6)(a, s); break;
        default: FK_CAT5(launch_imm_, FK_NX, FK_NZ, FK_NM, 7)(a, s); break;
    }
}
#endif
