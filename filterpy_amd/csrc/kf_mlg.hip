// kf_mlg.hip -- KalmanFilter.batch_filter for dim_x = 10..16 (dim_z = 1..4) with FOUR LANES PER TRACK (gfx950).
//
// One lane per track ends at dim_x = 9: P alone is 2 n^2 VGPRs.  Above that the library used to run the padded /
// rolled one-lane instantiations (scratch-resident arrays, guards around every access inside the time loop): 0.002-0.004
// of the HBM peak (profiles/r02/dims_10_16.jsonl).  Here the scheme of kf_ml.hip (dim_x = 9 on three lanes) is generalised: a QUAD of lanes owns a
// track, lane L holds rows L*R .. L*R+R-1 of P (R = ceil(n/4)); rows past n-1 -- the tail of lane 3 for n = 10, 11,
// 13, 14, 15 -- are CLAMPED to row n-1: the lane recomputes and rewrites that row (same values, same addresses), so no
// lane is ever predicated, and it enters the one cross-lane SUM of the step (H P) with coefficient zero.  x, y, S
// and its L D L' are replicated in the quad; rows travel by quad-permute DPP moves; every product keeps the
// reference's k = 0..n-1 order (filterpy/kalman/kalman_filter.py:472-478 predict, :533-556 Joseph-form update,
// :980-991 the batch loop):
//
//   predict  T = P F' (rows local) ; P' = a2 (F T) + Q : row k of T is broadcast by its owner
//   update   PHT = P H' rows local, broadcast -> S (+R), L D L', y replicated ; K rows local ;
//            H P = butterfly sum of the lanes' partial products ; T1 = P - K (H P) ; D = K R - T1 H' ;
//            P+ = T1 + D K' with K's rows broadcast (the same row updates x)
//
// Exact dims (one instantiation per (dim_x, dim_z)), all four outputs, optional mask (branch-free, see kf_ml.hip), SOA
// and AOS (outputs staged through a wave-private LDS tile and written as 1 KiB stores); the plain call (one constant
// model, predict -> update) and, as VAR instantiations, per-step model lists, a control input and update_first.  One
// wave per SIMD: at dim_x = 16 a lane holds P (64 doubles) and T (64) at once.  Per-track models, the update's
// by-products and dim_z > 4 at these sizes stay on the padded kernels.
#include <stdlib.h>
#include <type_traits>

#include "fk_device.hpp"
#include "fk_math_sym.hpp"
#include "fk_kernel_args.hpp"
#include "fk_ml.hpp"
#include "fk_chunks.hpp"
#include "../../include/filterhip.h"

#ifndef FK_NX
#error "compile with -DFK_NX=<dim_x> -DFK_NZ=<dim_z>"
#endif
// FK_MLG_ZDMA (build time): 0 = measurement and mask prefetched into registers again (A/B), 1 = LDS-DMA in the element-major
// kernels (measured +2..3 % at (10,2), (14,4), (16,4)), 2 = in the NumPy-order kernels too (measured -2..3 % at (12,3), (16,4))
#ifndef FK_MLG_ZDMA
#define FK_MLG_ZDMA 1
#endif

#define FK_MLG_CAT_(a, b, c) a##b##_##c
#define FK_MLG_CAT(a, b, c) FK_MLG_CAT_(a, b, c)

namespace fk {
namespace FK_MLG_CAT(mlg_, FK_NX, FK_NZ) {

// value v of row K's owner (lane K / R of the quad); K is a compile-time constant wherever this is used
template <int OWNER>
__device__ __forceinline__ double from_owner(double v)
{
    static_assert(OWNER >= 0 && OWNER < 4, "four lanes per track");
    return quad_bcast<OWNER>(v);
}
#define FK_OWNER_ROW(dst, M, k, LEN)                                              \
    FK_UNROLL for (int j_ = 0; j_ < (LEN); ++j_) {                                \
        const double v_ = M[(k) % R][j_];                                         \
        dst[j_] = ((k) / R == 0) ? from_owner<0>(v_) : ((k) / R == 1) ? from_owner<1>(v_) \
                : ((k) / R == 2) ? from_owner<2>(v_) : from_owner<3>(v_);         \
    }

// sum over the four lanes of a quad, the same bits in every lane: (a0 + a1) + (a2 + a3) in lanes 0, 1 and
// (a2 + a3) + (a0 + a1) in lanes 2, 3 (fp addition commutes)
__device__ __forceinline__ double quad_sum(double v)
{
    v += quad_rot<0xB1>(v);      // quad_perm:[1,0,3,2]
    v += quad_rot<0x4E>(v);      // quad_perm:[2,3,0,1]
    return v;
}

// AOS ([track][element], NumPy order) output of one (x, P) set: a wave's 16 tracks are one contiguous slab; the quads
// write their rows into a wave-private LDS tile laid out like the slab, then the 64 lanes copy consecutive 16-byte
// units (1 KiB per store instruction).  The descriptor is sized to the wave's valid tracks: the range check drops the tail.
template <int R, int NX>
__device__ __forceinline__ void mlg_store_aos(const double (&x)[NX], const double (&P)[R][NX], const unsigned (&row)[R],
                                              double *xdst, double *Pdst, double *tile, unsigned lane, unsigned valid)
{
    constexpr int EP = NX * NX, UP = 16 * EP / 2, UX = 16 * NX / 2;      // 16-byte units per wave
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
    const unsigned q = lane >> 2;
    double *tx = tile, *tP = tile + 16 * NX;
    ml_wave_fence();
    FK_UNROLL for (int k = 0; k < NX; ++k) tx[q * NX + k] = x[k];                         // the quad writes the same value
    FK_UNROLL for (int r = 0; r < R; ++r)
        FK_UNROLL for (int c = 0; c < NX; ++c) tP[q * EP + row[r] * NX + c] = P[r][c];
    ml_wave_fence();
    const rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(xdst, 0, (int)(valid * (unsigned)NX * 8u), 0x00020000);
    const rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc(Pdst, 0, (int)(valid * (unsigned)EP * 8u), 0x00020000);
    // (reads in batches ahead of their stores: ml_copy_units)
    ml_copy_units<UX, 4>(lane, [&](unsigned unit) { return tx + 2u * unit; },
                         [&](unsigned unit, bool ok, const u32x4 &v) {
                             __builtin_amdgcn_raw_buffer_store_b128(v, rx, ok ? unit * 16u : ML_OFF_DROP, 0, 0);
                         });
    ml_copy_units<UP, 4>(lane, [&](unsigned unit) { return tP + 2u * unit; },
                         [&](unsigned unit, bool ok, const u32x4 &v) {
                             __builtin_amdgcn_raw_buffer_store_b128(v, rP, ok ? unit * 16u : ML_OFF_DROP, 0, 0);
                         });
}

// VAR: batch_filter's other arguments (kalman_filter.py:941-991) exactly as kf_ml.hip's VAR family serves them at (9, 3):
// one model per step shared by the bank ([F | Q | H | R | B] double-buffered in LDS, fetched a step ahead), a control
// input x = F x + B u (u[t] travels with z[t]), both as BRANCH-FREE run-time switches; UF (update_first) swaps the
// halves of the step at compile time (the predict half is kf_mlg_predict.inc, included before or after the update half).
// EX: the update's by-products as per-step histories (fk_kf_batch_filter_ex_f64: y, K, S, SI, log-likelihood, mahalanobis --
// what a Saver or batch_filter(saver=...) asks for, common/helpers.py:121-152) from the plain call without a mask: S^-1, log det S
// and y' S^-1 y from the factorisation the gain was solved with, exactly as kf_fast / kf_kernel form them; every array
// leaves through the wave's LDS tile in 16-byte units.  (With a mask the histories carry the LAST K / S / SI across missing
// measurements: those calls stay on kf_fast's extras instantiations / the generic kernel.)
template <int NX, int NZ, int LAYOUT, bool VAR = false, bool UF = false, bool EX = false>
__global__ void __launch_bounds__(BLOCK, (EX ? 1 : NX <= 8 ? 3 : NX <= 9 ? 2 : 1))
kf_mlg_kernel(const KfArgs a)
{
    constexpr int R = (NX + 3) / 4;
    using LM = LdsModel<NX, NZ>;
    constexpr bool AOS = LAYOUT == LAYOUT_AOS;
    constexpr int TILE = 16 * NX + 16 * NX * NX;                 // AOS: one (x, P) output set of a wave
    constexpr int NUC = 4;                                       // padded dim_u (VAR)
    constexpr int MLEN = LM::SIZE + NX * NUC;                    // VAR: [F | Q | H | R | B padded to NX x NUC]
    constexpr int MSTR = (MLEN + 2) & ~1;                        // ... + the pad slot idle fetch slots publish into
    constexpr int MPT = (MLEN + BLOCK - 1) / BLOCK;              // model elements a thread fetches per step
    constexpr int MSZ = VAR ? 2 * MSTR : LM::SIZE;
    static_assert(!UF || VAR, "update_first is a VAR instantiation");
    static_assert(!EX || !VAR, "the by-product histories come with the plain call");
    // SOA covariances leave through an LDS slab as 16-byte units (ml_store_rows_soa_slab) up to dim_x = 12: measured
    // 0.37 -> 0.48 of HBM at (10,2), 0.34 -> 0.35 at (12,3), but 0.39 -> 0.32 at (14,4) -- the larger kernels are bound by
    // their arithmetic and code size, not by store slots (profiles/r02/dims_10_16.jsonl vs dims_10_16_slab.jsonl)
    constexpr bool SOA_SLAB = !AOS && NX <= FK_SOA_SLAB_MAX;
    __shared__ double smem[MSZ + (AOS || SOA_SLAB || EX ? (BLOCK / 64) * TILE : 0)];
    double *tile = smem + MSZ + (threadIdx.x >> 6) * TILE;
    lds_fill<NX, NX>(smem + LM::OFF_F, a.F, NX, NX, 1.0, threadIdx.x);
    lds_fill<NX, NX>(smem + LM::OFF_Q, a.Q, NX, NX, 0.0, threadIdx.x);
    lds_fill<NZ, NX>(smem + LM::OFF_H, a.H, NZ, NX, 0.0, threadIdx.x);
    lds_fill<NZ, NZ>(smem + LM::OFF_R, a.R, NZ, NZ, 1.0, threadIdx.x);
    if constexpr (VAR) lds_fill<NX, NUC>(smem + LM::SIZE, a.nu > 0 ? a.B : nullptr, NX, a.nu, 0.0, threadIdx.x);
    // element k of the concatenated model of step tt (B goes with the model mode like the others; no control input:
    // a valid dummy address, the value selected to 0)
    const double *B_or_dummy = VAR && a.nu > 0 ? a.B : a.F;
    auto model_elem = [&](long tt, int k) -> double {
        if (k < LM::OFF_Q) return a.F[tt * (NX * NX) + k];
        if (k < LM::OFF_H) return a.Q[tt * (NX * NX) + (k - LM::OFF_Q)];
        if (k < LM::OFF_R) return a.H[tt * (NZ * NX) + (k - LM::OFF_H)];
        if (k < LM::SIZE) return a.R[tt * (NZ * NZ) + (k - LM::OFF_R)];
        const int i = (k - LM::SIZE) / NUC, j = (k - LM::SIZE) % NUC;
        const bool live = k < MLEN && j < a.nu;
        const double v = B_or_dummy[live ? tt * (long)(NX * a.nu) + i * a.nu + j : 0];
        return live ? v : 0.0;
    };
    double mnext[VAR ? MPT : 1];               // this thread's elements of the NEXT step's model
    if constexpr (VAR) {
        FK_UNROLL for (int e = 0; e < MPT; ++e) {
            mnext[e] = model_elem(a.model_t && a.T > 1 ? 1 : 0, (int)threadIdx.x + e * BLOCK);
            asm volatile("" ::"v"(mnext[e]));
        }
    }
    __syncthreads();
    const double *sF = smem + LM::OFF_F, *sQ = smem + LM::OFF_Q, *sH = smem + LM::OFF_H, *sR = smem + LM::OFF_R;
    const double *sB = smem + LM::SIZE;

    const long N = a.N;
    const unsigned L = threadIdx.x & 3u;
    const long iend = a.i0 + a.cnt;                            // this launch's track group (chunked calls, fk_chunks.hpp)
    long trk = a.i0 + (long)blockIdx.x * (BLOCK / 4) + (threadIdx.x >> 2);
    const bool owner = trk < iend;                             // tail quads only duplicate: they never write the final state
    if (trk >= iend) trk = iend - 1;
    unsigned row[R];                                           // the rows this lane holds (clamped: see the header)
    double live[R];                                            // 1.0 for a row of its own, 0.0 for a clamped duplicate
    FK_UNROLL for (int r = 0; r < R; ++r) {
        const unsigned g = L * (unsigned)R + (unsigned)r;
        row[r] = g < (unsigned)NX ? g : (unsigned)NX - 1u;
        live[r] = g < (unsigned)NX ? 1.0 : 0.0;
    }
    // element e of this lane's track sits at  lane offset + e * estride:  SOA: track*8 + e*N*8 ; AOS: track*E*8 + e*8
    unsigned estride = AOS ? 8u : (unsigned)N * 8u;
    asm volatile("" : "+s"(estride));
    const unsigned t8 = (unsigned)trk * (AOS ? (unsigned)NX * 8u : 8u);                       // x-like arrays
    const unsigned tz8 = (unsigned)trk * (AOS ? (unsigned)NZ * 8u : 8u);                      // z
    unsigned off_row[R];                                                                      // element row[r] * NX of a covariance record
    FK_UNROLL for (int r = 0; r < R; ++r)
        off_row[r] = (AOS ? (unsigned)trk * (unsigned)(NX * NX) * 8u : (unsigned)trk * 8u) + row[r] * (unsigned)NX * estride;
    const long w0 = a.i0 + (long)blockIdx.x * (BLOCK / 4) + (long)wave_index() * 16;        // scalar: see wave_index()
    const unsigned valid = (unsigned)(iend - w0 >= 16 ? 16 : (iend - w0 > 0 ? iend - w0 : 0));
    const unsigned lane = threadIdx.x & 63u;
    const uint8_t *mask_or_dummy = a.mask ? a.mask : reinterpret_cast<const uint8_t *>(a.z);
    const unsigned nu = VAR ? (unsigned)a.nu : 0u;
    const unsigned nu_idx = nu ? nu - 1u : 0u;
    const unsigned tu8 = (unsigned)trk * (AOS ? nu * 8u : 8u);                                // u
    const double *u_or_dummy = VAR && nu ? a.u : a.z;

    double P[R][NX], x[NX];
    {
        const MlView vx(a.x, t8, estride);
        FK_UNROLL for (int r = 0; r < R; ++r) {
            const MlView vP(a.P, off_row[r], estride);
            FK_UNROLL for (int c = 0; c < NX; ++c) P[r][c] = vP.load(c);
        }
        FK_UNROLL for (int k = 0; k < NX; ++k) x[k] = vx.load(k);
        FK_UNROLL for (int r = 0; r < R; ++r)
            FK_UNROLL for (int c = 0; c < NX; ++c) asm volatile("" ::"v"(P[r][c]));     // landed before the loop (kf_ml.hip)
        FK_UNROLL for (int k = 0; k < NX; ++k) asm volatile("" ::"v"(x[k]));
    }
    int st = 0;
    // ZDMA (the plain call and its EX twin): the measurement and the mask byte of step t + 1 travel HBM -> LDS by LDS-DMA while
    // step t computes (LaneRecordDma, fk_device.hpp; the byte as the aligned dword around it).  As register prefetches they
    // were waited for at the end of the step with vmcnt(12 .. 24): a partial drain of the step's stores, with one wave per SIMD
    // and nothing to cover it.  Read behind s_waitcnt vmcnt(k), k = the store instructions of a step (a lower bound, <= 63).
    constexpr int ZIMGD = LaneRecordDma<NZ, LAYOUT>::IMG_DOUBLES + 32;        // + 64 dwords: the mask bytes' dwords
    // (where the images fit next to the staging tiles: at dim_x 16 with dim_z >= 5 they do not)
    constexpr long LDS_OTHER = (long)(MSZ + (AOS || SOA_SLAB || EX ? (BLOCK / 64) * TILE : 0)) * 8 + 64;
    constexpr bool ZDMA = !VAR && (AOS ? FK_MLG_ZDMA >= 2 : FK_MLG_ZDMA >= 1) && LDS_OTHER + (long)(BLOCK / 64) * 2 * ZIMGD * 8 <= 160 * 1024;
    constexpr int ZST = AOS ? 2 * ((16 * NX / 2 + 63) / 64 + (16 * NX * NX / 2 + 63) / 64)
                            : (SOA_SLAB ? 2 * (NX + (NX * NX * 8 + 63) / 64) : 2 * (NX + R * NX));
    constexpr int ZWAIT = ZST < 63 ? ZST : 63;
    __shared__ double s_zd[ZDMA ? (BLOCK / 64) * 2 * ZIMGD : 1];
    LaneRecordDma<NZ, LAYOUT> zdma;
    [[maybe_unused]] auto zreq = [&](long tt, unsigned buf) {
        zdma.request(a.z + tt * N * NZ, (unsigned)N * (unsigned)NZ * 8u, buf);
        // the mask byte of (tt, trk): the aligned dword around it (base and its misalignment are wave-uniform)
        const unsigned long long mb = reinterpret_cast<unsigned long long>(mask_or_dummy) + (unsigned long long)(tt * N);
        const unsigned delta = (unsigned)(mb & 3ull);
        const dma_rsrc_t rm = make_dma_rsrc(reinterpret_cast<const void *>(mb & ~3ull), (unsigned)N + 8u);
        lds_dma4(rm, ((unsigned)trk + delta) & ~3u, 0u, zdma.lds + buf * (unsigned)(ZIMGD * 8) + (unsigned)(LaneRecordDma<NZ, LAYOUT>::IMG_DOUBLES * 8));
    };
    [[maybe_unused]] auto zread = [&](long tt, unsigned buf, double (&zd)[NZ]) -> unsigned {
        zdma.read(buf, zd);                                    // (its images are ZIMGD apart: see init below)
        const unsigned long long mb = reinterpret_cast<unsigned long long>(mask_or_dummy) + (unsigned long long)(tt * N);
        const unsigned sh = (((unsigned)trk + (unsigned)(mb & 3ull)) & 3u) * 8u;
        const unsigned dw = zdma.img0[buf * (unsigned)(ZIMGD * 2) + (unsigned)(LaneRecordDma<NZ, LAYOUT>::IMG_DOUBLES * 2) + (threadIdx.x & 63u)];
        return (dw >> sh) & 0xffu;
    };
    double zn[NZ];
    double un[VAR ? NUC : 1];
    unsigned hn = 1u;
    if constexpr (ZDMA) {
        zdma.init(s_zd + wave_index() * (2 * ZIMGD), (unsigned)trk, (unsigned)N, threadIdx.x & 63u);
        zdma.stride_doubles = ZIMGD;
        zreq(0, 0u);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        const MlView vz(a.z, tz8, estride);
        FK_UNROLL for (int c = 0; c < NZ; ++c) zn[c] = vz.load(c);
        const unsigned hb = mask_or_dummy[trk];
        hn = a.mask ? hb : 1u;
        if constexpr (VAR) {
            const MlView vu(u_or_dummy, tu8, estride);
            FK_UNROLL for (int c = 0; c < NUC; ++c) {
                const double v = vu.load(uniform_int((unsigned)c < nu ? c : (int)nu_idx));   // clamped element index: no branch
                un[c] = nu ? v : 0.0;
            }
            FK_UNROLL for (int c = 0; c < NUC; ++c) asm volatile("" ::"v"(un[c]));
        }
        FK_UNROLL for (int c = 0; c < NZ; ++c) asm volatile("" ::"v"(zn[c]));
        asm volatile("" ::"v"(hn));
    }
    _Pragma("nounroll") for (long t = 0; t < a.T; ++t) {
        double z[NZ];
        double u[VAR ? NUC : 1];
        if constexpr (ZDMA) {
            if (t > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ZWAIT) : "memory");
            const unsigned hb = zread(t, (unsigned)(t & 1), zn);
            hn = a.mask ? hb : 1u;
            zreq(t + 1 < a.T ? t + 1 : t, (unsigned)((t + 1) & 1));
        }
        const bool has_z = hn != 0u;
        FK_UNROLL for (int c = 0; c < NZ; ++c) z[c] = has_z ? zn[c] : 0.0;
        if constexpr (VAR) {
            FK_UNROLL for (int c = 0; c < NUC; ++c) u[c] = un[c];
        }
        if constexpr (!ZDMA) {
            long tn = t + 1 < a.T ? t + 1 : t;
            asm volatile("" : "+s"(tn));
            const MlView vz(a.z + tn * N * NZ, tz8, estride);
            FK_UNROLL for (int c = 0; c < NZ; ++c) zn[c] = vz.load(c);
            const unsigned hb = mask_or_dummy[tn * N + trk];
            hn = a.mask ? hb : 1u;
            if constexpr (VAR) {
                const MlView vu(u_or_dummy + tn * N * (long)nu, tu8, estride);
                FK_UNROLL for (int c = 0; c < NUC; ++c) {
                    const double v = vu.load(uniform_int((unsigned)c < nu ? c : (int)nu_idx));
                    un[c] = nu ? v : 0.0;
                }
            }
        }
        // VAR: mnext holds this thread's elements of model[t+1]; keep them for the hand-over at the end of the step and
        // request model[t+2] (consumed one step later: a counted wait)
        double mpub[VAR ? MPT : 1];
        if constexpr (VAR) {
            long t2 = a.model_t ? (t + 2 < a.T ? t + 2 : a.T - 1) : 0;
            asm volatile("" : "+s"(t2));
            FK_UNROLL for (int e = 0; e < MPT; ++e) {
                mpub[e] = mnext[e];
                mnext[e] = model_elem(t2, (int)threadIdx.x + e * BLOCK);
            }
        }
        if constexpr (!UF) {
#include "kf_mlg_predict.inc"
        }
        {
        // ----------------------------------------------------------------- update --
        // Joseph form with the identity-minus-product factors applied implicitly, WITHOUT assuming P symmetric
        // (kf_ml.hip explains why H P is not replaced by (P H')'):
        //   T1 = (I-KH) P = P - K (H P) ;  G = T1 H' ;  P+ = T1 (I-KH)' + K R K' = T1 + (K R - G) K'
        double y[NZ], K[R][NZ];
        double exLf[EX ? NZ * NZ : 1], exdinv[EX ? NZ : 1];     // EX: the factor of S, kept for S^-1 at the end of the step
        [[maybe_unused]] const unsigned g = lane >> 2;
        // BRANCH-FREE like every other run-time switch of these kernels (a branch inside the time loop splits its one basic
        // block and the register allocation falls apart: 0.2 -> 1.6 KB of scratch per lane measured here): a history the caller
        // did not ask for is written through a descriptor of zero tracks -- the stores are issued and dropped
        [[maybe_unused]] auto ex_out = [&](double *hist, auto e_tag) {
            constexpr int E = decltype(e_tag)::value;
            double *dst = hist ? hist : a.means;
            const unsigned vv = hist ? valid : 0u;
            if constexpr (AOS) ml_tile_out_aos<E, 16>(dst + (t * N + w0) * E, tile, lane, vv);
            else ml_tile_out_soa<E, 16>(dst + t * N * E, N, w0, tile, lane, vv);
        };
        [[maybe_unused]] auto ex_scalar = [&](double *hist, double v) {
            double *dst = hist ? hist : a.means;
            const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(dst + t * N, 0, hist ? (int)((unsigned)N * 8u) : 0, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), rs, (unsigned)trk * 8u, 0, 0);
        };
#define FK_TILE_AT(E, e) tile[AOS ? g * (unsigned)(E) + (unsigned)(e) : (unsigned)(e) * 16u + g]
        {
            // Row c of H serves y[c] and column c of P H' in one pass and is requested one row ahead (kf_mlg_predict.inc);
            // column k of H for S likewise, R in front of the sums it closes
            double PHT[R][NZ], S[NZ * NZ];
            {
                double Hr[2][NX];
                FK_UNROLL for (int k = 0; k < NX; ++k) Hr[0][k] = sH[k];
                FK_UNROLL for (int c = 0; c < NZ; ++c) {
                    if (c + 1 < NZ) {
                        FK_UNROLL for (int k = 0; k < NX; ++k) Hr[(c + 1) & 1][k] = sH[(c + 1) * NX + k];
                    }
                    const double (&Hc)[NX] = Hr[c & 1];
                    double acc = Hc[0] * x[0];
                    FK_UNROLL for (int k = 1; k < NX; ++k) acc = fma(Hc[k], x[k], acc);
                    y[c] = z[c] - acc;
                    FK_UNROLL for (int r = 0; r < R; ++r) {
                        double pa = P[r][0] * Hc[0];
                        FK_UNROLL for (int k = 1; k < NX; ++k) pa = fma(P[r][k], Hc[k], pa);
                        PHT[r][c] = pa;
                    }
                    FK_STAGE();
                }
            }
            // S = H PHT + R, replicated in every lane: PHT's row k comes from its owner
            {
                double hk[2][NZ], Rs[NZ * NZ];
                FK_UNROLL for (int r = 0; r < NZ; ++r) hk[0][r] = sH[r * NX];
                FK_UNROLL for (int e = 0; e < NZ * NZ; ++e) Rs[e] = sR[e];
                FK_UNROLL for (int k = 0; k < NX; ++k) {
                    if (k + 1 < NX) {
                        FK_UNROLL for (int r = 0; r < NZ; ++r) hk[(k + 1) & 1][r] = sH[r * NX + k + 1];
                    }
                    double pk[NZ];
                    FK_OWNER_ROW(pk, PHT, k, NZ);
                    FK_UNROLL for (int r = 0; r < NZ; ++r)
                        FK_UNROLL for (int c = 0; c < NZ; ++c)
                            S[r * NZ + c] = (k == 0) ? hk[k & 1][r] * pk[c] : fma(hk[k & 1][r], pk[c], S[r * NZ + c]);
                    if (k % 4 == 3) FK_STAGE();
                }
                FK_UNROLL for (int e = 0; e < NZ * NZ; ++e) S[e] += Rs[e];
            }
            FK_STAGE();
            double Lf[NZ * NZ], d[NZ], dinv[NZ], Kr[R * NZ];
            FK_UNROLL for (int e = 0; e < NZ * NZ; ++e) Lf[e] = S[e];
            if (!ldlt2<NZ>(Lf, d, dinv) && has_z) st |= ST_NOT_PD;
            FK_UNROLL for (int r = 0; r < R; ++r)
                FK_UNROLL for (int c = 0; c < NZ; ++c) Kr[r * NZ + c] = PHT[r][c];
            solve_rows_ldlt<R, NZ>(Lf, dinv, Kr);
            FK_UNROLL for (int r = 0; r < R; ++r)
                FK_UNROLL for (int c = 0; c < NZ; ++c) K[r][c] = has_z ? Kr[r * NZ + c] : 0.0;
            if constexpr (EX) {
                // early: y, S and the two scalars (the residual is spent by the state update, S by nothing else); the factor
                // (Lf, dinv) stays live to the end of the step, where K and S^-1 leave -- the step's register peak lies between
                double logdet, q = 0.0;
                if constexpr (NZ == 1) {
                    logdet = log(S[0]);
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
                    logdet = logdet_from_dinv<NZ>(dinv, NZ);
                }
                {
                    ml_wave_fence();
                    FK_UNROLL for (int c = 0; c < NZ; ++c) FK_TILE_AT(NZ, c) = y[c];
                    ml_wave_fence();
                    ex_out(a.y_out, std::integral_constant<int, NZ>{});
                }
                {
                    ml_wave_fence();
                    FK_UNROLL for (int e = 0; e < NZ * NZ; ++e) FK_TILE_AT(NZ * NZ, e) = S[e];
                    ml_wave_fence();
                    ex_out(a.S_out, std::integral_constant<int, NZ * NZ>{});
                }
                ml_wave_fence();
                ex_scalar(a.ll_out, -0.5 * (NZ * 1.8378770664093453 + logdet + q));   // the quad writes the same value
                ex_scalar(a.maha_out, sqrt(q));
                FK_UNROLL for (int e = 0; e < NZ * NZ; ++e) exLf[e] = Lf[e];
                FK_UNROLL for (int e = 0; e < NZ; ++e) exdinv[e] = dinv[e];
            }
        }
        FK_STAGE();
        {
            // H P: this lane's OWN rows contribute sum_r H[c][row r] P[r][:] (a clamped duplicate: coefficient 0);
            // the quad adds the four parts
            double HP[NZ][NX], hcn[2][R];
            FK_UNROLL for (int r = 0; r < R; ++r) hcn[0][r] = sH[row[r]];
            FK_UNROLL for (int c = 0; c < NZ; ++c) {
                if (c + 1 < NZ) {
                    FK_UNROLL for (int r = 0; r < R; ++r) hcn[(c + 1) & 1][r] = sH[(c + 1) * NX + row[r]];
                }
                double hc[R];
                FK_UNROLL for (int r = 0; r < R; ++r) hc[r] = live[r] * hcn[c & 1][r];
                FK_UNROLL for (int j = 0; j < NX; ++j) {
                    double acc = hc[0] * P[0][j];
                    FK_UNROLL for (int r = 1; r < R; ++r) acc = fma(hc[r], P[r][j], acc);
                    HP[c][j] = quad_sum(acc);
                }
                FK_STAGE();
            }
            // T1 = P - K (H P) (own rows, in place)
            FK_UNROLL for (int r = 0; r < R; ++r) {
                FK_UNROLL for (int j = 0; j < NX; ++j) {
                    double acc = P[r][j];
                    FK_UNROLL for (int c = 0; c < NZ; ++c) acc = fma(-K[r][c], HP[c][j], acc);
                    P[r][j] = acc;
                }
                FK_STAGE();
            }
            // D = K R - T1 H' (own rows)
            double D[R][NZ];
            {
                double Hr[2][NX], Rc[2][NZ];             // row c of H and column c of R one iteration ahead
                FK_UNROLL for (int k = 0; k < NX; ++k) Hr[0][k] = sH[k];
                FK_UNROLL for (int q = 0; q < NZ; ++q) Rc[0][q] = sR[q * NZ];
                FK_UNROLL for (int c = 0; c < NZ; ++c) {
                    if (c + 1 < NZ) {
                        FK_UNROLL for (int k = 0; k < NX; ++k) Hr[(c + 1) & 1][k] = sH[(c + 1) * NX + k];
                        FK_UNROLL for (int q = 0; q < NZ; ++q) Rc[(c + 1) & 1][q] = sR[q * NZ + c + 1];
                    }
                    const double (&Hc)[NX] = Hr[c & 1];
                    FK_UNROLL for (int r = 0; r < R; ++r) {
                        double kr = K[r][0] * Rc[c & 1][0];
                        FK_UNROLL for (int q = 1; q < NZ; ++q) kr = fma(K[r][q], Rc[c & 1][q], kr);
                        double g = P[r][0] * Hc[0];
                        FK_UNROLL for (int k = 1; k < NX; ++k) g = fma(P[r][k], Hc[k], g);
                        D[r][c] = kr - g;
                    }
                    FK_STAGE();
                }
            }
            // P+ = T1 + D K' : column j needs K's row j from its owner; the same row updates x[j]
            FK_UNROLL for (int j = 0; j < NX; ++j) {
                double Kj[NZ];
                FK_OWNER_ROW(Kj, K, j, NZ);
                double xa = x[j];
                FK_UNROLL for (int c = 0; c < NZ; ++c) xa = fma(Kj[c], y[c], xa);
                x[j] = xa;
                FK_UNROLL for (int r = 0; r < R; ++r) {
                    double acc = P[r][j];
                    FK_UNROLL for (int c = 0; c < NZ; ++c) acc = fma(D[r][c], Kj[c], acc);
                    P[r][j] = acc;
                }
            }
        }
        if constexpr (EX) {
            // late: the gain (own rows) and S^-1 from the kept factor, as kf_fast / kf_kernel form it (inv_from_ldlt)
            {
                ml_wave_fence();
                FK_UNROLL for (int r = 0; r < R; ++r)
                    FK_UNROLL for (int c = 0; c < NZ; ++c) FK_TILE_AT(NX * NZ, row[r] * (unsigned)NZ + (unsigned)c) = K[r][c];
                ml_wave_fence();
                ex_out(a.K_out, std::integral_constant<int, NX * NZ>{});
            }
            {
                double SI[NZ * NZ];
                inv_from_ldlt<NZ>(exLf, exdinv, SI);
                ml_wave_fence();
                FK_UNROLL for (int e = 0; e < NZ * NZ; ++e) FK_TILE_AT(NZ * NZ, e) = SI[e];
                ml_wave_fence();
                ex_out(a.SI_out, std::integral_constant<int, NZ * NZ>{});
            }
            ml_wave_fence();
        }
#undef FK_TILE_AT
        if constexpr (AOS) {
            mlg_store_aos<R, NX>(x, P, row, a.means + (t * N + w0) * NX, a.covs + (t * N + w0) * NX * NX, tile, lane, valid);
        } else {
            const MlView vx(a.means + t * N * NX, t8, estride);
            FK_UNROLL for (int k = 0; k < NX; ++k) vx.store(k, x[k]);
            if constexpr (SOA_SLAB) {
                ml_store_rows_soa_slab<R, NX, 16>(P, row, a.covs + t * N * NX * NX, N, w0, tile, lane, lane >> 2, valid);
            } else {
                FK_UNROLL for (int r = 0; r < R; ++r) {
                    const MlView vP(a.covs + t * N * NX * NX, off_row[r], estride);
                    FK_UNROLL for (int c = 0; c < NX; ++c) vP.store(c, P[r][c]);
                }
            }
        }
        }      // update half
        if constexpr (UF) {
#include "kf_mlg_predict.inc"
        }
        if constexpr (VAR) {
            // publish model[t+1] into the other LDS buffer: nobody reads that buffer during step t, and one barrier makes
            // it visible for step t+1 (every thread of the workgroup runs all T steps)
            double *nb = smem + ((t + 1) & 1) * MSTR;
            FK_UNROLL for (int e = 0; e < MPT; ++e) {
                const unsigned k = threadIdx.x + (unsigned)e * BLOCK;
                nb[k < (unsigned)MLEN ? k : (unsigned)MLEN] = mpub[e];       // slots past the model: one pad slot
            }
            __syncthreads();
            sF = nb + LM::OFF_F;
            sQ = nb + LM::OFF_Q;
            sH = nb + LM::OFF_H;
            sR = nb + LM::OFF_R;
            sB = nb + LM::SIZE;
        }
    }
    // the final state goes back in place: only a track's own quad writes it (a duplicating tail quad that loaded
    // x0 / P0 late must not find the final state there)
    // ... and no lane of the workgroup may still be about to LOAD x0 / P0 when an owner overwrites them: every wave has
    // consumed its initial state once it arrives here (one barrier per launch, outside the time loop)
    __syncthreads();
    if (owner) {
        const MlView vx(a.x, t8, estride);
        bool fin = all_finite<NX>(x);
        FK_UNROLL for (int k = 0; k < NX; ++k) vx.store(k, x[k]);
        FK_UNROLL for (int r = 0; r < R; ++r) {
            const MlView vP(a.P, off_row[r], estride);
            FK_UNROLL for (int c = 0; c < NX; ++c) {
                vP.store(c, P[r][c]);
                fin = fin && (fabs(P[r][c]) <= 1.79769313486231570815e+308);
            }
        }
        if (a.status) {
            int s = st | (fin ? 0 : ST_NONFINITE);
            s |= __builtin_amdgcn_mov_dpp(s, 0xB1, 0xf, 0xf, true);
            s |= __builtin_amdgcn_mov_dpp(s, 0x4E, 0xf, 0xf, true);
            if (L == 0) a.status[trk] = a.status_or ? (a.status[trk] | s) : s;
        }
    }
}

}  // namespace (instantiation)

// returns 1 when this call is not one the four-lane kernel serves
int FK_MLG_CAT(launch_kf_mlg_, FK_NX, FK_NZ)(const KfArgs &a, int layout, bool outs, int model_mode, hipStream_t s)
{
    using namespace FK_MLG_CAT(mlg_, FK_NX, FK_NZ);
    if ((model_mode != FK_MODEL_SHARED && model_mode != FK_MODEL_PER_STEP) || a.n != FK_NX || a.m != FK_NZ || !outs) return 1;
    if (a.y_out || a.K_out || a.S_out || a.SI_out || a.ll_out || a.maha_out) {
        // the by-product histories: the plain call without a mask (EX instantiations)
        if (model_mode != FK_MODEL_SHARED || a.nu > 0 || a.update_first || a.mask || !a.extras_per_step) return 1;
        auto onex = [layout](const KfArgs &b, hipStream_t sb) -> int {
            const dim3 gb((unsigned)((b.cnt + BLOCK / 4 - 1) / (BLOCK / 4))), bb(BLOCK);
            if (layout == FK_LAYOUT_AOS) hipLaunchKernelGGL((kf_mlg_kernel<FK_NX, FK_NZ, LAYOUT_AOS, false, false, true>), gb, bb, 0, sb, b);
            else hipLaunchKernelGGL((kf_mlg_kernel<FK_NX, FK_NZ, LAYOUT_SOA, false, false, true>), gb, bb, 0, sb, b);
            return check_launch("kf_mlg_kernel<ex>");
        };
        return kf_chunked_call(a, FK_NX, FK_NZ, 1024, onex, s);      // one wave per SIMD at every size
    }
    if (model_mode == FK_MODEL_PER_STEP || a.nu > 0 || a.update_first) {
        // the VAR instantiations (FK_ML_VAR=0 sends these calls back to the padded kernel)
        const char *vv = getenv("FK_ML_VAR");
        if (a.nu > 4 || (vv && atoi(vv) == 0)) return 1;
        auto onev = [layout](const KfArgs &b, hipStream_t sb) -> int {
            const dim3 gb((unsigned)((b.cnt + BLOCK / 4 - 1) / (BLOCK / 4))), bb(BLOCK);
#define GOV(UFV)                                                                                                    \
    if (layout == FK_LAYOUT_AOS) hipLaunchKernelGGL((kf_mlg_kernel<FK_NX, FK_NZ, LAYOUT_AOS, true, UFV>), gb, bb, 0, sb, b); \
    else hipLaunchKernelGGL((kf_mlg_kernel<FK_NX, FK_NZ, LAYOUT_SOA, true, UFV>), gb, bb, 0, sb, b)
            if (b.update_first) { GOV(true); } else { GOV(false); }
#undef GOV
            return check_launch("kf_mlg_kernel<var>");
        };
        return kf_chunked_call(a, FK_NX, FK_NZ, FK_NX <= 8 ? 3072 : FK_NX <= 9 ? 2048 : 1024, onev, s);
    }
    // the plain call, with tail filling where the last round of waves would be mostly idle (fk_chunks.hpp)
    auto one = [layout](const KfArgs &b, hipStream_t sb) -> int {
        const dim3 gb((unsigned)((b.cnt + BLOCK / 4 - 1) / (BLOCK / 4))), bb(BLOCK);
        if (layout == FK_LAYOUT_AOS) hipLaunchKernelGGL((kf_mlg_kernel<FK_NX, FK_NZ, LAYOUT_AOS>), gb, bb, 0, sb, b);
        else hipLaunchKernelGGL((kf_mlg_kernel<FK_NX, FK_NZ, LAYOUT_SOA>), gb, bb, 0, sb, b);
        return check_launch("kf_mlg_kernel");
    };
    return kf_chunked_call(a, FK_NX, FK_NZ, FK_NX <= 8 ? 3072 : FK_NX <= 9 ? 2048 : 1024, one, s);
}

}  // namespace fk
