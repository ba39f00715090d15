// kf_fast.hip -- the headline kernel: KalmanFilter.batch_filter for a bank of tracks with ONE
// model shared by every track and step (BASELINE.json configs[1] / SURVEY.md §8 row K3).
//
// Same arithmetic as kf_kernels.hip (fk_math.hpp: filterpy/kalman/kalman_filter.py:472-478,
// 533-556, loop :980-991), specialised for the common call -- predict then update, no control
// input, constant shared F/Q/H/R -- so that the per-step instruction stream is nothing but
//   1 measurement load (software-pipelined one step ahead, counted vmcnt: the wave never
//     drains its own stores), ~480 fp64 FMAs, 22 LDS broadcast reads of the model,
//   the four output records (x-, P-, x+, P+) streamed straight to HBM.
// Layouts:
//   SOA  a[t][e][i]: each store instruction covers 64 consecutive tracks of one element
//        (512 B contiguous per wave);
//   AOS  a[t][i][e] (NumPy C order): records are transposed through a wave-private LDS tile so
//        every store instruction still writes 1 KiB of contiguous memory (16 B per lane).
// HBM-bound: 8*(m + 2n + 2n^2) algorithmic bytes per track-step (336 B at n=4, m=2).
#include "fk_device.hpp"
#include "fk_kernel_args.hpp"
#include "fk_math_sym.hpp"

#ifndef FK_NX
#error "compile with -DFK_NX=<dim_x> -DFK_NZ=<dim_z> -DFK_VARIANT=<v> -DFK_FAST_WAVES=<w>"
#endif
#ifndef FK_FAST_WAVES
#define FK_FAST_WAVES 0
#endif
// FK_FAST_SYM=1: covariance kept as a packed upper triangle (fk_math_sym.hpp) -- fewer registers
// and flops; reads only the upper triangle of P0 and writes a mirrored P.
#ifndef FK_FAST_SYM
#define FK_FAST_SYM 0
#endif
// FK_FAST_ZDEPTH: steps of measurement prefetch (0 = default: 2 for AOS with small records, else 1)
// FK_FAST_ZDMA=0 (build time): the one-wave-per-SIMD instantiations prefetch their measurements into registers again (A/B)
#ifndef FK_FAST_ZDMA
#define FK_FAST_ZDMA 1
#endif
#ifndef FK_FAST_ZDEPTH
#define FK_FAST_ZDEPTH 0
#endif

#ifndef FK_VARIANT
#define FK_VARIANT 0
#endif
#define FK_CAT_(a, b, c, d) a##b##_##c##_v##d
#define FK_CAT(a, b, c, d) FK_CAT_(a, b, c, d)

namespace fk {
namespace FK_CAT(fastv_, FK_NX, FK_NZ, FK_VARIANT) {

// Occupancy target per SIMD.  The AOS kernels carry the transpose addressing on top of the filter
// state: at dim_x = 4 they need ~190 VGPRs, so they run at 2 waves per SIMD (and make up for it with a
// two-step-deep store pipeline, ZDEPTH) instead of spilling at 3.
#ifndef FK_FAST_WAVES_AOS
#define FK_FAST_WAVES_AOS 0
#endif
constexpr int fast_min_waves(int nx, int layout)
{
    if (layout == LAYOUT_AOS && FK_FAST_WAVES_AOS) return FK_FAST_WAVES_AOS;
    if (layout == LAYOUT_AOS && nx > 2 && nx <= 4) return 2;
    return FK_FAST_WAVES ? FK_FAST_WAVES : (nx <= 2 ? 8 : nx <= 4 ? 3 : nx <= 6 ? 2 : 1);
}

// expand the covariance state (full or packed upper triangle) to a row-major NX x NX array;
// pure register renaming after optimisation
template <int NX, bool SYM, int PLEN>
__device__ __forceinline__ void cov_full(const double (&P)[PLEN], double (&M)[NX * NX])
{
    FK_UNROLL for (int a = 0; a < NX; ++a)
        FK_UNROLL for (int b = 0; b < NX; ++b) M[a * NX + b] = SYM ? P[sym_idx<NX>(a, b)] : P[a * NX + b];
}

// MMODE (= FK_MODEL_* of include/filterhip.h): 0 one shared constant model (LDS), 1 one model per
// track (registers, loaded once: coalesced in SOA), 2 one model per track and step (registers,
// reloaded every step), 3 one model per step shared by all tracks (LDS, double-buffered, fetched a
// step ahead by the first SIZE threads).
// UF: update_first (kalman_filter.py:966-978): every step is update(z) -> store posterior -> predict -> store prior
// CTRL: control input x = F x + B u (kalman_filter.py:472-475) with one shared B (dim_u <= 4); u[t] travels with z[t]
// EX: the update's by-products as per-step histories (fk_kf_batch_filter_ex_f64: y, K, S, SI, log-likelihood, mahalanobis;
// KalmanFilter.batch_filter with a Saver, kalman_filter.py:533-563 and the lazy properties :1180-1225) stored by this kernel
// instead of the generic one -- shared constant model, predict -> update, all four outputs.
// IL (round 4; NumPy order, dim_x <= 4, FK_KF_FLAG_COV_INTERLEAVED): the prior covariance of a step waits in registers for the
// posterior and the two leave TOGETHER as one 2 n^2-double record per track -- 1 KiB contiguous per store instruction, one
// write front for both histories (written apart, as two n^2 islands per track at different moments of the step, the
// interleaved array is slower than two arrays: profiles/r04/placement).
template <int NX, int NZ, int LAYOUT, bool HAS_MASK, bool OUTS, bool SYM, int MMODE, bool UF, bool CTRL, bool EX = false, bool IL = false>
__global__ void __launch_bounds__(BLOCK, fast_min_waves(NX, LAYOUT) > 1 && (MMODE == 1 || MMODE == 2 || EX) ? fast_min_waves(NX, LAYOUT) - 1 : fast_min_waves(NX, LAYOUT))
kf_fast_kernel(const KfArgs a, const double *__restrict__ pF, const double *__restrict__ pQ,
               const double *__restrict__ pH, const double *__restrict__ pR,
               const double *__restrict__ pz, const uint8_t *__restrict__ pmask)
{
    using SharedModel = LdsModel<NX, NZ>;
    // AOS records go through the LDS transpose while the wave tiles fit: dim_x <= 6 at any occupancy,
    // dim_x <= 8 for the one-wave-per-SIMD instantiations (4 tiles of 64 x 65 doubles = 133 KB);
    // dim_x = 9 falls back to per-lane 16-byte stores (the three-lane kernel takes the common call)
    constexpr bool COOP = (LAYOUT == LAYOUT_AOS) && (NX * NX <= 36 || (NX <= 8 && fast_min_waves(NX, LAYOUT) == 1));
    static_assert(!IL || (COOP && OUTS && !UF && !EX && NX * NX <= 16), "IL: the plain predict -> update call in NumPy order, dim_x <= 4");
    constexpr int TILE = COOP ? 64 * ((IL ? 2 * NX * NX : NX * NX) | 1) : 0;   // doubles per wave
    constexpr int MSIZE = SharedModel::SIZE * (MMODE == 3 ? 2 : 1);   // per-step models: double buffer
    __shared__ double s_mem[MSIZE + (BLOCK / 64) * TILE + 1];
    constexpr int NUC = 4;                                   // padded dim_u
    constexpr int ZW = NZ + (CTRL ? NUC : 0);                // a measurement buffer carries [z | u]
    __shared__ double s_B[CTRL ? NX * NUC : 1];
    // ZDMA (dim_x >= 7: one wave per SIMD, the plain and extras calls without a mask or a control input): the measurement of
    // step t + 1 travels HBM -> LDS by LDS-DMA while step t computes.  These instantiations have no VGPR to spare: a register
    // prefetch was "spilled" to an AGPR the moment it was issued, i.e. waited for on the spot with vmcnt(0) -- dim_z pipeline
    // drains per step behind the previous step's 144 stores (round 4, ISA of kf_fast 8_4).  LaneRecordDma (fk_device.hpp): 2 dim_z
    // dword-DMA instructions per step into one of two images; the image of step t is read at the top of step t behind
    // s_waitcnt vmcnt(63) -- the DMA is older than the >= 72 stores step t - 1 issued after it (vmcnt retires in order) -- or
    // vmcnt(0) where a step has no stores to drain.
    constexpr bool ZDMA = NX > 6 && !CTRL && !HAS_MASK && FK_FAST_ZDMA;
    constexpr int ZIMG = 64 * NZ;                             // doubles per image
    __shared__ double s_z[ZDMA ? (BLOCK / 64) * 2 * ZIMG : 1];
    if constexpr (CTRL) lds_fill<NX, NUC>(s_B, a.B, NX, a.nu, 0.0, threadIdx.x);   // synchronised with the model fill below

    const long N = a.N, T = a.T;
    // Workgroup -> track-block mapping.  Workgroups are dispatched round-robin over the 8 XCDs
    // (block b on XCD b % 8, MI355X_MICROARCH.md); with xcd_swizzle each XCD works on ONE contiguous
    // eighth of the tracks, so the lines (and translations) it touches per step are contiguous too.
    unsigned bid = blockIdx.x;
    if (a.xcd_swizzle) {
        const unsigned nb = gridDim.x, per = nb / 8u, rem = nb % 8u, xcd = bid % 8u;
        bid = xcd * per + (xcd < rem ? xcd : rem) + bid / 8u;
    }
    const long blk0 = a.i0 + (long)bid * BLOCK;
    const unsigned tid = threadIdx.x;
    const long left = a.i0 + a.cnt - blk0;                       // >= 1
    const unsigned last_row = (unsigned)(left < BLOCK ? left : BLOCK) - 1u;
    const Lane ln{blk0, min(threadIdx.x, last_row), N};          // tail lanes duplicate the last track
    const Lane lr = ln;
    const unsigned lane = tid & 63u, wave = tid >> 6;
    double *tile = s_mem + MSIZE + wave * TILE;
    // ZDMA: every lane fetches its own (clamped) track's measurement (LaneRecordDma, fk_device.hpp)
    LaneRecordDma<NZ, LAYOUT> zdma;
    if constexpr (ZDMA) zdma.init(s_z + wave_index() * (2 * ZIMG), (unsigned)blk0 + ln.tid, (unsigned)N, lane);
    [[maybe_unused]] auto dma_z = [&](long tt, unsigned buf) {
        const long tq = tt < T ? tt : T - 1;
        zdma.request(pz + tq * N * NZ, (unsigned)N * (unsigned)NZ * 8u, buf);
    };

    RegModel<NX, NZ> tm;                       // per-track models (MMODE 1, 2)
    constexpr long FSZ = NX * NX, HSZ = NZ * NX, RSZ = NZ * NZ;
    // element `tid` of the concatenated shared model [F | Q | H | R] of step tt (MMODE 3)
    auto shared_elem = [&](long tt) -> double {
        const int k = (int)tid;
        if (k < SharedModel::OFF_Q) return pF[tt * FSZ + k];
        if (k < SharedModel::OFF_H) return pQ[tt * FSZ + (k - SharedModel::OFF_Q)];
        if (k < SharedModel::OFF_R) return pH[tt * HSZ + (k - SharedModel::OFF_H)];
        if (k < SharedModel::SIZE) return pR[tt * RSZ + (k - SharedModel::OFF_R)];
        return 0.0;
    };
    auto load_track_model = [&](long tt) {
        load_rec<NX, NX, LAYOUT, true>(tm.F, pF + tt * N * FSZ, lr, NX, NX, 1.0);
        load_rec<NX, NX, LAYOUT, true>(tm.Q, pQ + tt * N * FSZ, lr, NX, NX, 0.0);
        load_rec<NZ, NX, LAYOUT, true>(tm.H, pH + tt * N * HSZ, lr, NZ, NX, 0.0);
        load_rec<NZ, NZ, LAYOUT, true>(tm.R, pR + tt * N * RSZ, lr, NZ, NZ, 1.0);
    };
    double mnext = 0.0;                        // MMODE 3: this thread's element of the NEXT step's model
    if constexpr (MMODE == 0 || MMODE == 3) {
        lds_fill<NX, NX>(s_mem + SharedModel::OFF_F, pF, NX, NX, 1.0, tid);
        lds_fill<NX, NX>(s_mem + SharedModel::OFF_Q, pQ, NX, NX, 0.0, tid);
        lds_fill<NZ, NX>(s_mem + SharedModel::OFF_H, pH, NZ, NX, 0.0, tid);
        lds_fill<NZ, NZ>(s_mem + SharedModel::OFF_R, pR, NZ, NZ, 1.0, tid);
        if constexpr (MMODE == 3) {
            mnext = shared_elem(T > 1 ? 1 : 0);     // model[1]
            asm volatile("" ::"v"(mnext));
        }
        __syncthreads();
    } else {
        load_track_model(0);
        FK_UNROLL for (int i = 0; i < NX * NX; ++i) { asm volatile("" ::"v"(tm.F[i])); asm volatile("" ::"v"(tm.Q[i])); }
        FK_UNROLL for (int i = 0; i < NZ * NX; ++i) asm volatile("" ::"v"(tm.H[i]));
        FK_UNROLL for (int i = 0; i < NZ * NZ; ++i) asm volatile("" ::"v"(tm.R[i]));
    }
    SharedModel sm{s_mem};

    constexpr int PLEN = SYM ? NX * (NX + 1) / 2 : NX * NX;
    double x[NX], P[PLEN];      // SYM: P is the packed upper triangle
    load_rec<NX, 1, LAYOUT, true>(x, a.x, lr, NX, 1, 0.0);
    {
        const RecView<LAYOUT> pv(a.P, lr, NX * NX);
        FK_UNROLL for (int i = 0; i < NX; ++i)
            FK_UNROLL for (int j = (SYM ? i : 0); j < NX; ++j)
                P[SYM ? sym_idx<NX>(i, j) : i * NX + j] = pv.load(i * NX + j);
    }

    // Measurement pipeline: z[t + ZDEPTH] is requested at the top of step t and consumed ZDEPTH steps
    // later; the time loop is unrolled by three with statically rotating buffers so that a pending
    // load is never copied (a copy would wait for it).  The wait for a measurement only needs the
    // stores issued BEFORE its load to have drained, so a wave may have ZDEPTH steps of stores in
    // flight: ZDEPTH = 2 for AOS (20 x 1 KiB stores per step), 1 for SOA (40 x 512 B per step; the
    // vmcnt counter tops out at 63 outstanding operations).
    constexpr int ZDEPTH = (FK_FAST_ZDEPTH > 0) ? FK_FAST_ZDEPTH : ((LAYOUT == LAYOUT_AOS && NX * NX + NX <= 24) ? 2 : 1);
    double zb[3][ZW];
    bool hb[3] = {true, true, true};
    auto load_z = [&](long tt, double (&zd)[ZW], bool &hd) {
        long tq = tt < T ? tt : T - 1;                // clamp: always issue the same number of loads
        // the rolled loop of dim_x >= 7 (below): opaque, or the compiler re-derives this load one iteration LATER -- right in
        // front of its use, each element followed by its own s_waitcnt vmcnt(0): no prefetch at all and four pipeline drains
        // per step (round 4, ISA of kf_fast 8_4; kf_ml.hip does the same for the same reason)
        if constexpr (NX > 6) asm volatile("" : "+s"(tq));
        {
            const RecView<LAYOUT> zv(pz + tq * N * NZ, lr, NZ);
            FK_UNROLL for (int c = 0; c < NZ; ++c) zd[c] = zv.load(c);
        }
        if constexpr (CTRL) {
            const RecView<LAYOUT> uv(a.u + tq * N * a.nu, lr, a.nu);
            FK_UNROLL for (int c = 0; c < NUC; ++c) zd[NZ + c] = uv.load(c < a.nu ? c : a.nu - 1);   // clamped: no branch
        }
        if constexpr (HAS_MASK && EX) {
            // (the extras twins also serve calls WITHOUT a mask -- see the launcher: pmask == NULL reads a valid dummy byte of z
            //  and selects "present"; no branch in the time loop)
            const uint8_t *pm = pmask ? pmask : reinterpret_cast<const uint8_t *>(pz);
            hd = (pm[tq * N + lr.blk0 + lr.tid] != 0) | (pmask == nullptr);
        } else if (HAS_MASK) hd = pmask[tq * N + lr.blk0 + lr.tid] != 0;
    };
    if constexpr (ZDMA) {
        dma_z(0, 0u);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        load_z(0, zb[0], hb[0]);
    }
    if (ZDEPTH == 2) load_z(1, zb[1], hb[1]);
    // Land every prologue load before the loop: a load still pending at the loop header would
    // make the compiler wait vmcnt(0) inside the loop on every iteration (draining the stores).
    FK_UNROLL for (int i = 0; i < NX; ++i) asm volatile("" ::"v"(x[i]));
    FK_UNROLL for (int i = 0; i < PLEN; ++i) asm volatile("" ::"v"(P[i]));
    FK_UNROLL for (int i = 0; i < ZW; ++i) asm volatile("" ::"v"(zb[0][i]));
    if (ZDEPTH == 2) { FK_UNROLL for (int i = 0; i < ZW; ++i) asm volatile("" ::"v"(zb[1][i])); }

    int st = 0;
    // EX with a mask: a missing measurement stores y = 0 and the LAST K / S / SI / log det S (kalman_filter.py:515-520
    // leaves those attributes alone)
    constexpr bool CARRY = EX && HAS_MASK;
    double cK[CARRY ? NX * NZ : 1], cS[CARRY ? NZ * NZ : 1], cSI[CARRY ? NZ * NZ : 1], c_logdet = 0.0;
    if constexpr (CARRY) {
        FK_UNROLL for (int i = 0; i < NX * NZ; ++i) cK[i] = 0.0;
        FK_UNROLL for (int i = 0; i < NZ * NZ; ++i) { cS[i] = 0.0; cSI[i] = 0.0; }
    }
    // one time step: consumes (zu, hu), requests the measurement of step t + ZDEPTH into (zl, hl)
    auto step = [&](long t, const double (&zu)[ZW], bool hu, double (&zl)[ZW], bool &hl) {
        double zdm[ZDMA ? NZ : 1];
        if constexpr (ZDMA) {
            // this step's image (requested a step ago), then the request for the next one into the other image
            if constexpr (OUTS) asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            zdma.read((unsigned)(t & 1), zdm);
            dma_z(t + 1, (unsigned)((t + 1) & 1));          // (the other image: nothing in flight reads or writes it)
        } else {
            load_z(t + ZDEPTH, zl, hl);
        }
        if constexpr (MMODE == 2) {
            if (t > 0) load_track_model(t);            // this step's per-track model
        }
        // MMODE 3: mnext holds this thread's element of model[t+1]; keep it for the hand-over at the end
        // of the step and request model[t+2] (consumed one step later: a counted wait)
        const double mpub = mnext;
        if constexpr (MMODE == 3) mnext = shared_elem(t + 2 < T ? t + 2 : T - 1);
        double Pf[NX * NX];
        double Pprior[IL ? NX * NX : 1];
        auto do_predict = [&]() {
            if constexpr (MMODE == 1 || MMODE == 2) {
                if constexpr (SYM) kf_predict_sym<NX>(x, P, tm, a.alpha_sq);
                else kf_predict<NX>(x, P, tm, a.alpha_sq);
            } else {
                if constexpr (SYM) kf_predict_sym<NX>(x, P, sm, a.alpha_sq);
                else kf_predict<NX>(x, P, sm, a.alpha_sq);
            }
            if constexpr (CTRL) {
                // x = (F x) + (B u): the product first, like dot(F, x) + dot(B, u)
                FK_UNROLL for (int i = 0; i < NX; ++i) {
                    double bu = s_B[i * NUC] * zu[NZ];
                    FK_UNROLL for (int k = 1; k < NUC; ++k) bu = fma(s_B[i * NUC + k], (k < a.nu ? zu[NZ + k] : 0.0), bu);
                    x[i] += bu;
                }
            }
            cov_full<NX, SYM, PLEN>(P, Pf);
            if (!OUTS) {
            } else if (!COOP) {
                store_rec<NX, 1, LAYOUT, true>(x, a.means_p + t * N * NX, ln, NX, 1);
                store_rec<NX, NX, LAYOUT, true>(Pf, a.covs_p + t * a.cov_step, ln, NX, NX);
            } else if constexpr (IL) {
                wave_store_aos<NX>(x, a.means_p + (t * N + blk0) * NX, wave * 64u, tile, lane, last_row);
                FK_UNROLL for (int e = 0; e < NX * NX; ++e) Pprior[e] = Pf[e];        // leaves with the posterior
            } else {
                wave_store_aos<NX>(x, a.means_p + (t * N + blk0) * NX, wave * 64u, tile, lane, last_row);
                wave_store_aos_pitch<NX * NX>(Pf, a.covs_p + t * a.cov_step + blk0 * a.cov_pitch, wave * 64u, tile, lane, last_row, (unsigned)a.cov_pitch);
            }
        };
        // (the NumPy-order plain kernel at dim_x 8 has its register peak in the update half -- the full covariance staged for the
        //  slab store sits next to it --: the H / R copy of model_cached spilled 184 B there and is switched off)
        constexpr bool UPDC = NX <= 8 && !(COOP && !EX && NX == 8);      // (dim_x 9: no room for it at all)
        auto do_update = [&]() {
            // EX: the update's by-products leave the branch below in registers; the histories are formed from them in
            // straight-line code with per-lane selects (a lane without a measurement keeps y = 0 and the LAST K / S / SI /
            // log det S): the stores that follow are wave-cooperative, nothing between here and them may diverge
            double K[NX * NZ], y[NZ], S[NZ * NZ], Lf[NZ * NZ], dinv[NZ];
            if constexpr (EX && HAS_MASK) {
                FK_UNROLL for (int i = 0; i < NX * NZ; ++i) K[i] = 0.0;
                FK_UNROLL for (int i = 0; i < NZ * NZ; ++i) { S[i] = 0.0; Lf[i] = 0.0; }
                FK_UNROLL for (int i = 0; i < NZ; ++i) { y[i] = 0.0; dinv[i] = 1.0; }
            }
            if (hu) {
                double zq[NZ];
                FK_UNROLL for (int c = 0; c < NZ; ++c) zq[c] = ZDMA ? zdm[ZDMA ? c : 0] : zu[c];
                if constexpr (MMODE == 1 || MMODE == 2) {
                    if constexpr (SYM) st |= kf_update_sym<NX, NZ>(x, P, zq, tm, K, y, S, Lf, dinv);
                    else st |= kf_update<NX, NZ>(x, P, zq, tm, K, y, S, Lf, dinv);
                } else {
                    if constexpr (SYM) st |= kf_update_sym<NX, NZ, false, UPDC>(x, P, zq, sm, K, y, S, Lf, dinv);
                    else st |= kf_update<NX, NZ>(x, P, zq, sm, K, y, S, Lf, dinv);
                }
            }
            double eSI[EX ? NZ * NZ : 1], e_ll = 0.0, e_maha = 0.0;
            if constexpr (EX) {
                // S^-1, log det S and y' S^-1 y from the factorisation, exactly as kf_kernel does (kf_kernels.hip)
                inv_from_ldlt<NZ>(Lf, dinv, eSI);
                double logdet = 0.0, q = 0.0;
                if constexpr (NZ == 1) {
                    logdet = log(S[0]);                                 // (a lane without a measurement: selected away below)
                    q = y[0] * y[0] * dinv[0];
                } else {
                    double w[NZ];
                    FK_UNROLL for (int i = 0; i < NZ; ++i) {
                        double acc = y[i];
                        FK_UNROLL for (int k2 = 0; k2 < NZ; ++k2)
                            if (k2 < i) acc = fma(-Lf[i * NZ + k2], w[k2], acc);
                        w[i] = acc;
                        q = fma(acc * acc, dinv[i], q);
                    }
                    logdet = logdet_from_dinv<NZ>(dinv, NZ);            // one logarithm, not a division and a logarithm per pivot
                }
                if constexpr (CARRY) {
                    FK_UNROLL for (int i = 0; i < NX * NZ; ++i) { cK[i] = hu ? K[i] : cK[i]; K[i] = cK[i]; }
                    FK_UNROLL for (int i = 0; i < NZ * NZ; ++i) {
                        cS[i] = hu ? S[i] : cS[i];
                        cSI[i] = hu ? eSI[i] : cSI[i];
                        S[i] = cS[i];
                        eSI[i] = cSI[i];
                    }
                    c_logdet = hu ? logdet : c_logdet;
                    logdet = c_logdet;
                    q = hu ? q : 0.0;
                }
                e_ll = -0.5 * (NZ * 1.8378770664093453 + logdet + q);
                e_maha = sqrt(q);
            }
            if constexpr (EX) {
                // uniform branches on the pointers: a caller asks for any subset
                if constexpr (!COOP) {
                    if (a.y_out) store_rec<NZ, 1, LAYOUT, true>(y, a.y_out + t * N * NZ, ln, NZ, 1);
                    if (a.K_out) store_rec<NX, NZ, LAYOUT, true>(K, a.K_out + t * N * NX * NZ, ln, NX, NZ);
                    if (a.S_out) store_rec<NZ, NZ, LAYOUT, true>(S, a.S_out + t * N * NZ * NZ, ln, NZ, NZ);
                    if (a.SI_out) store_rec<NZ, NZ, LAYOUT, true>(eSI, a.SI_out + t * N * NZ * NZ, ln, NZ, NZ);
                } else {
                    if (a.y_out) wave_store_aos<NZ>(y, a.y_out + (t * N + blk0) * NZ, wave * 64u, tile, lane, last_row);
                    if (a.K_out) wave_store_aos<NX * NZ>(K, a.K_out + (t * N + blk0) * NX * NZ, wave * 64u, tile, lane, last_row);
                    if (a.S_out) wave_store_aos<NZ * NZ>(S, a.S_out + (t * N + blk0) * NZ * NZ, wave * 64u, tile, lane, last_row);
                    if (a.SI_out) wave_store_aos<NZ * NZ>(eSI, a.SI_out + (t * N + blk0) * NZ * NZ, wave * 64u, tile, lane, last_row);
                }
                if (a.ll_out) a.ll_out[t * N + blk0 + ln.tid] = e_ll;          // tail lanes rewrite the last track's value
                if (a.maha_out) a.maha_out[t * N + blk0 + ln.tid] = e_maha;
            }
            cov_full<NX, SYM, PLEN>(P, Pf);
            if (!OUTS) {
            } else if (!COOP) {
                store_rec<NX, 1, LAYOUT, true>(x, a.means + t * N * NX, ln, NX, 1);
                store_rec<NX, NX, LAYOUT, true>(Pf, a.covs + t * a.cov_step, ln, NX, NX);
            } else if constexpr (IL) {
                wave_store_aos<NX>(x, a.means + (t * N + blk0) * NX, wave * 64u, tile, lane, last_row);
                double rec[2 * NX * NX];                                            // [posterior | prior]: cov2[t][track][2][n*n]
                FK_UNROLL for (int e = 0; e < NX * NX; ++e) { rec[e] = Pf[e]; rec[NX * NX + e] = Pprior[e]; }
                wave_store_aos<2 * NX * NX>(rec, a.covs + t * a.cov_step + blk0 * (2 * NX * NX), wave * 64u, tile, lane, last_row);
            } else {
                wave_store_aos<NX>(x, a.means + (t * N + blk0) * NX, wave * 64u, tile, lane, last_row);
                wave_store_aos_pitch<NX * NX>(Pf, a.covs + t * a.cov_step + blk0 * a.cov_pitch, wave * 64u, tile, lane, last_row, (unsigned)a.cov_pitch);
            }
        };
        if constexpr (UF) {
            do_update();
            do_predict();
        } else {
            do_predict();
            do_update();
        }
        if constexpr (MMODE == 3) {
            // publish model[t+1] into the other LDS buffer: nobody reads that buffer during step t, and
            // one barrier makes it visible for step t+1
            double *nb = s_mem + ((t + 1) & 1) * SharedModel::SIZE;
            if (tid < (unsigned)SharedModel::SIZE) nb[tid] = mpub;
            __syncthreads();
            sm.s = nb;
        }
    };
    if constexpr (NX <= 6) {
        // buffer roles rotate statically: step t uses zb[t % 3] and loads into zb[(t + ZDEPTH) % 3]
        for (long t = 0; t < T; t += 3) {
            step(t, zb[0], hb[0], zb[ZDEPTH % 3], hb[ZDEPTH % 3]);
            if (t + 1 < T) step(t + 1, zb[1], hb[1], zb[(1 + ZDEPTH) % 3], hb[(1 + ZDEPTH) % 3]);
            if (t + 2 < T) step(t + 2, zb[2], hb[2], zb[(2 + ZDEPTH) % 3], hb[(2 + ZDEPTH) % 3]);
        }
    } else {
        // large dim_x: one step body is already several thousand instructions -- unrolling it three
        // times overflows the instruction cache (measured 3x slower at dim_x = 9); depth-1 pipeline
        // with a register copy of the prefetched measurement instead
        static_assert(NX <= 6 || ZDEPTH == 1, "the rolled loop implements ZDEPTH == 1");
        for (long t = 0; t < T; ++t) {
            step(t, zb[0], hb[0], zb[1], hb[1]);
            if constexpr (!ZDMA) {
                FK_UNROLL for (int i = 0; i < ZW; ++i) zb[0][i] = zb[1][i];
                hb[0] = hb[1];
            }
        }
    }

    // The final state goes back IN PLACE (a.x, a.P are inputs too), so only the owner of a track may write it: a tail
    // lane that duplicates the last track and loaded its x0 / P0 late would otherwise find the final state there and
    // filter it a second time.  These stores are outside the time loop: predication costs nothing here (the
    // per-step output stores stay unpredicated -- they only ever rewrite the same values).
    // ... and no lane of the workgroup may still be about to LOAD x0 / P0 when an owner overwrites them: every wave has
    // consumed its initial state once it arrives here (one barrier per launch, outside the time loop)
    __syncthreads();
    if (tid <= last_row) {
        store_rec<NX, 1, LAYOUT, true>(x, a.x, ln, NX, 1);
        double Pl[NX * NX];
        cov_full<NX, SYM, PLEN>(P, Pl);
        store_rec<NX, NX, LAYOUT, true>(Pl, a.P, ln, NX, NX);
        if (a.status) {
            if (!all_finite<NX>(x) || !all_finite<PLEN>(P)) st |= ST_NONFINITE;
            a.status[blk0 + ln.tid] = a.status_or ? (a.status[blk0 + ln.tid] | st) : st;
        }
    }
}



}  // namespace (variant)
using namespace FK_CAT(fastv_, FK_NX, FK_NZ, FK_VARIANT);

// Handles tracks [a.i0, a.i0 + a.cnt).  outs: all four outputs non-NULL (true) or all NULL (false).
// mmode: FK_MODEL_* of the descriptor.  Returns 1 if this instantiation does not carry that model
// mode (only variant 0 at dim_x <= 6 compiles the per-track / per-step modes): the caller then falls
// back to the generic kernel.
#define FK_FAST_ALL_MODES (FK_VARIANT == 0 && FK_NX <= 6)
// the extras instantiations are compiled once per (dim_x, dim_z): in variant 0 where it exists, else in the lean variant
#define FK_FAST_EX (FK_VARIANT == 0 || !((FK_NX == 4 && FK_NZ == 2) || (FK_NX == 6 && FK_NZ == 3)))
int FK_CAT(launch_kf_fast_, FK_NX, FK_NZ, FK_VARIANT)(const KfArgs &a, int layout, bool outs, int mmode, hipStream_t stream)
{
    const dim3 grid((unsigned)((a.cnt + BLOCK - 1) / BLOCK)), block(BLOCK);
    // FK_KF_FLAG_COV_INTERLEAVED in NumPy order needs the wave-cooperative store path (the record pitch is its argument)
    constexpr bool coop_aos = FK_NX * FK_NX <= 36 || (FK_NX <= 8 && fast_min_waves(FK_NX, LAYOUT_AOS) == 1);
    if (layout == LAYOUT_AOS && a.cov_pitch != FK_NX * FK_NX && !coop_aos) return 1;
    if (a.y_out || a.K_out || a.S_out || a.SI_out || a.ll_out || a.maha_out) {
#if FK_FAST_EX
        if (mmode != 0 || !outs || a.update_first || a.nu > 0 || !a.extras_per_step) return 1;
#define FK_GOEX(LAY, MSK)                                                                                                \
    hipLaunchKernelGGL((kf_fast_kernel<FK_NX, FK_NZ, LAY, MSK, true, (FK_FAST_SYM != 0), 0, false, false, true>), grid, block, 0, \
                       stream, a, a.F, a.Q, a.H, a.R, a.z, a.mask)
        // Round 5: a call WITHOUT a mask runs on the masked twin where the no-mask instantiation is the worse kernel.  Without the
        // `if (hu)` branches its three unrolled steps are one basic block, and at some shapes the scheduler interleaves them past the
        // register file: (6,3) 512 VGPRs + 568 / 168 B of scratch against 414 / 328 and none.  Measured A/B/A/B over every shape
        // whose twins differ statically (profiles/r05/extras/, ms NumPy order / element-major, N x 100 steps):
        //   (6,3) 8.53 -> 4.85 / 5.85 -> 4.61    (5,4) 15.2 -> 5.96 / 11.2 -> 5.88    (6,4) 16.9 -> 5.77 / 14.4 -> 5.31
        //   (9,4) 18.7 -> 11.4 / 11.9 -> 7.58;   within 2 % at (4,2) (4,3) (4,4) (5,2) (5,3) (6,2) (7,3); the masked twin is the
        //   slower one at (8,4) (x 2) and (9,3) (x 1.14): the rule is per instantiation.  FK_FAST_EX_MASKED=0 / 1 forces either.
        constexpr bool prefer_masked = (FK_NX == 5 && FK_NZ == 4) || (FK_NX == 6 && FK_NZ >= 3) || (FK_NX == 9 && FK_NZ == 4);
        static const int ex_masked = [] { const char *v = getenv("FK_FAST_EX_MASKED"); return v ? (v[0] == '1' ? 1 : 0) : -1; }();
        const bool msk = a.mask != nullptr || (ex_masked < 0 ? prefer_masked : ex_masked == 1);
        if (layout == LAYOUT_SOA) {
            if (msk) FK_GOEX(LAYOUT_SOA, true);
            else FK_GOEX(LAYOUT_SOA, false);
        } else {
            if (msk) FK_GOEX(LAYOUT_AOS, true);
            else FK_GOEX(LAYOUT_AOS, false);
        }
#undef FK_GOEX
        return check_launch("kf_fast_kernel (extras)");
#else
        return 1;
#endif
    }
    if (mmode != 0 && !FK_FAST_ALL_MODES) return 1;
    if (!outs && FK_VARIANT != 0) return 1;
    if (a.update_first && !(FK_FAST_ALL_MODES && mmode == 0)) return 1;   // update_first: shared model, variant 0, dim_x <= 6
    if (a.nu > 0 && !(FK_FAST_ALL_MODES && mmode == 0 && !a.update_first && a.nu <= 4)) return 1;   // control input: same, dim_u <= 4
    // FK_KF_FLAG_COV_INTERLEAVED, NumPy order, dim_x <= 4, the plain call on a shared model: prior and posterior leave together
#if FK_NX * FK_NX <= 16 && FK_VARIANT == 0
    if (layout == LAYOUT_AOS && a.cov_pitch == 2 * FK_NX * FK_NX && outs && mmode == 0 && !a.update_first && a.nu == 0 && !getenv("FK_FAST_NO_IL")) {
        if (a.mask) hipLaunchKernelGGL((kf_fast_kernel<FK_NX, FK_NZ, LAYOUT_AOS, true, true, (FK_FAST_SYM != 0), 0, false, false, false, true>), grid, block, 0, stream, a, a.F, a.Q, a.H, a.R, a.z, a.mask);
        else hipLaunchKernelGGL((kf_fast_kernel<FK_NX, FK_NZ, LAYOUT_AOS, false, true, (FK_FAST_SYM != 0), 0, false, false, false, true>), grid, block, 0, stream, a, a.F, a.Q, a.H, a.R, a.z, a.mask);
        return check_launch("kf_fast_kernel (interleaved)");
    }
#endif
#define FK_GO(LAY, MSK, OUT, MM)                                                                          \
    hipLaunchKernelGGL((kf_fast_kernel<FK_NX, FK_NZ, LAY, MSK, OUT, (FK_FAST_SYM != 0), MM, false, false>), grid, block, 0, \
                       stream, a, a.F, a.Q, a.H, a.R, a.z, a.mask)
#define FK_GOUF(LAY, MSK, OUT)                                                                                \
    hipLaunchKernelGGL((kf_fast_kernel<FK_NX, FK_NZ, LAY, MSK, OUT, (FK_FAST_SYM != 0), 0, true, false>), grid, block, 0, \
                       stream, a, a.F, a.Q, a.H, a.R, a.z, a.mask)
#if FK_VARIANT == 0
#define FK_GO3(LAY, MM)                          \
    do {                                         \
        if (a.mask) {                            \
            if (outs) FK_GO(LAY, true, true, MM); \
            else FK_GO(LAY, true, false, MM);    \
        } else {                                 \
            if (outs) FK_GO(LAY, false, true, MM); \
            else FK_GO(LAY, false, false, MM);   \
        }                                        \
    } while (0)
#else       // lean / tuning variants: only the four-outputs kernels (a final-state-only call runs variant 0 or the generic kernel)
#define FK_GO3(LAY, MM)                          \
    do {                                         \
        if (a.mask) FK_GO(LAY, true, true, MM);  \
        else FK_GO(LAY, false, true, MM);        \
    } while (0)
#endif
#if FK_FAST_ALL_MODES
#define FK_GO2(LAY)                      \
    do {                                 \
        if (mmode == 0) FK_GO3(LAY, 0);  \
        else if (mmode == 1) FK_GO3(LAY, 1); \
        else if (mmode == 2) FK_GO3(LAY, 2); \
        else FK_GO3(LAY, 3);             \
    } while (0)
#else
#define FK_GO2(LAY) FK_GO3(LAY, 0)
#endif
#if FK_FAST_ALL_MODES
#define FK_GOUF2(LAY)                              \
    do {                                           \
        if (a.mask) {                              \
            if (outs) FK_GOUF(LAY, true, true);    \
            else FK_GOUF(LAY, true, false);        \
        } else {                                   \
            if (outs) FK_GOUF(LAY, false, true);   \
            else FK_GOUF(LAY, false, false);       \
        }                                          \
    } while (0)
    if (a.update_first) {
        if (layout == LAYOUT_SOA) FK_GOUF2(LAYOUT_SOA);
        else FK_GOUF2(LAYOUT_AOS);
        return check_launch("kf_fast_kernel");
    }
#undef FK_GOUF2
#define FK_GOC(LAY, MSK, OUT)                                                                                     \
    hipLaunchKernelGGL((kf_fast_kernel<FK_NX, FK_NZ, LAY, MSK, OUT, (FK_FAST_SYM != 0), 0, false, true>), grid, block, 0, \
                       stream, a, a.F, a.Q, a.H, a.R, a.z, a.mask)
#define FK_GOC2(LAY)                              \
    do {                                          \
        if (a.mask) {                             \
            if (outs) FK_GOC(LAY, true, true);    \
            else FK_GOC(LAY, true, false);        \
        } else {                                  \
            if (outs) FK_GOC(LAY, false, true);   \
            else FK_GOC(LAY, false, false);       \
        }                                         \
    } while (0)
    if (a.nu > 0) {
        if (layout == LAYOUT_SOA) FK_GOC2(LAYOUT_SOA);
        else FK_GOC2(LAYOUT_AOS);
        return check_launch("kf_fast_kernel");
    }
#undef FK_GOC2
#undef FK_GOC
#endif
    if (layout == LAYOUT_SOA) FK_GO2(LAYOUT_SOA);
    else FK_GO2(LAYOUT_AOS);
#undef FK_GO2
#undef FK_GOUF
#undef FK_GO3
#undef FK_GO
    return check_launch("kf_fast_kernel");
}

}  // namespace fk
