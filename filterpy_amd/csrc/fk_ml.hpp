// fk_ml.hpp -- cross-lane and buffer-access helpers shared by the several-lanes-per-track kernels
// (kf_ml.hip: dim_x = 9 on three lanes; kf_mlg.hip: dim_x 10..16 on four lanes).
#pragma once
#include <type_traits>
#include "fk_device.hpp"

// element-major covariances of the four-lane kernels leave through the LDS slab (ml_store_rows_soa_slab) up to this dim_x;
// above it: 8-byte stores per lane.  (Round 2 measured the slab as a loss above 12 -- with a copy-out that cost a division,
// two address computations, two exec regions and a waterfall loop per unit; round 4's ml_slab_out_soa has none of those.)
#ifndef FK_SOA_SLAB_MAX
#define FK_SOA_SLAB_MAX 16
#endif

namespace fk {

// FK_QUAD_SWIZZLE (build-time A/B): the broadcast through ds_swizzle_b32 (quad-permute mode: the LDS crossbar, no memory
// access) instead of v_mov_b32 DPP -- two instructions per double either way, but on the LDS pipe: the multi-lane kernels are
// bound by VALU issue and a fifth of their VALU instructions are these moves.
template <int SRC>
__device__ __forceinline__ double quad_bcast(double v)
{
    constexpr int ctrl = SRC * 0x55;   // quad_perm:[SRC,SRC,SRC,SRC]
    int lo = __double2loint(v), hi = __double2hiint(v);
#if defined(FK_QUAD_SWIZZLE)
    lo = __builtin_amdgcn_ds_swizzle(lo, 0x8000 | ctrl);
    hi = __builtin_amdgcn_ds_swizzle(hi, 0x8000 | ctrl);
#else
    lo = __builtin_amdgcn_mov_dpp(lo, ctrl, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, ctrl, 0xf, 0xf, true);
#endif
    return __hiloint2double(hi, lo);
}

// value of the lane selected by an arbitrary quad permutation (CTRL = quad_perm encoding)
template <int CTRL>
__device__ __forceinline__ double quad_rot(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// A buffer store of more than 8 bytes reads its data registers over several cycles; the VALU must not overwrite
// them in the next cycle.  hipcc keeps that distance itself -- except when the store's soffset is an SGPR, which the
// GCN3 manual exempts and LLVM's hazard recogniser therefore skips.  gfx950 does not honour the exemption: with the
// very next instruction a v_mov into dword 0 of the data (the next pair's DPP exchange), that dword reached memory
// with the NEXT value -- the low half of one covariance element per pair, 3e-7 relative, on every fourth track of
// every workgroup that was not the first on its CU (round 2, tests/test_gpu_baseline_configs.py found it at
// N = 1e5; tools/dbg_r02.py pinned the dword).  Two wait states after every such store.
// (the asm takes the four data dwords as inputs and clobbers memory: the registers stay live up to it and it cannot
// move above the store -- a bare `s_nop` was scheduled BEFORE the store it was meant to follow)
using hz_u32x4 = __attribute__((ext_vector_type(4))) unsigned;
__device__ __forceinline__ void store_data_hazard(const hz_u32x4 &v)
{
    asm volatile("s_nop 1" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w) : "memory");
}

// raw buffer access with a per-lane byte offset and a wave-uniform element offset
struct MlView {
    rsrc_t rs;
    unsigned voff, voff2, estride;
    // lane_off: this lane's byte offset for 8-byte accesses; pair_off: for the 16-byte pair accesses
    // (even quads: lane_off; odd quads: one track back and one element plane up)
    __device__ __forceinline__ MlView(const double *blk, unsigned lane_off, unsigned es, unsigned pair_off = 0)
        : rs(make_rsrc(blk)), voff(lane_off), voff2(pair_off), estride(es)
    {
        // re-laundered per view (i.e. per time step): otherwise every e * estride is loop-invariant,
        // gets hoisted out of the time loop (60 SGPRs) and the scalar file spills into VGPRs
        asm volatile("" : "+s"(estride));
    }
    // AUX: the instruction's cache-policy bits (gfx950: 16 = sc1, agent scope -- the access is coherent across the XCDs' L2s
    // by itself; the persistent grid hands a track's state from one workgroup to another this way, without L2 write-backs)
    template <int AUX = 0>
    __device__ __forceinline__ double load(int e) const
    {
        return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, (unsigned)e * estride, AUX));
    }
    template <int AUX = 0>
    __device__ __forceinline__ void store(int e, double x) const
    {
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, x), rs, voff, (unsigned)e * estride, AUX);
    }
    // Two elements per lane as ONE 16-byte store.  In the element-major layout the 16 bytes next to a
    // lane's value belong to the next track, i.e. to the next quad: even quads write element e of tracks
    // (q, q+1), odd quads element e+1 of tracks (q-1, q), each taking the partner's value over a row
    // shift by 4 lanes with a bank mask (DPP banks are the quads).  A wave may have 63 vector-memory
    // operations in flight whatever their size; with 8-byte stores that, not HBM, bounded this kernel
    // (~4 TB/s; the same bytes in half as many stores: +25 %).
    //   a = element e, b = element e + 1 of this lane's track.
    __device__ __forceinline__ void store_pair(int e, double a, double b) const
    {
        using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
        const int ax = __double2loint(a), ay = __double2hiint(a), bx = __double2loint(b), by = __double2hiint(b);
        u32x4 v;
        v.x = (unsigned)__builtin_amdgcn_update_dpp(ax, bx, 0x114, 0xf, 0xa, false);   // odd quads: b of quad - 1 (row_shr:4)
        v.y = (unsigned)__builtin_amdgcn_update_dpp(ay, by, 0x114, 0xf, 0xa, false);
        v.z = (unsigned)__builtin_amdgcn_update_dpp(bx, ax, 0x104, 0xf, 0x5, false);   // even quads: a of quad + 1 (row_shl:4)
        v.w = (unsigned)__builtin_amdgcn_update_dpp(by, ay, 0x104, 0xf, 0x5, false);
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff2, (unsigned)e * estride, 0);
        store_data_hazard(v);
    }
    // AOS: elements e and e + 1 of a track are adjacent in memory -- one 16-byte access, no exchange
    __device__ __forceinline__ void store2(int e, double a, double b) const
    {
        using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
        const u32x2 lo = __builtin_bit_cast(u32x2, a), hi = __builtin_bit_cast(u32x2, b);
        const u32x4 v = {lo.x, lo.y, hi.x, hi.y};
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff, (unsigned)e * estride, 0);
        store_data_hazard(v);
    }
    __device__ __forceinline__ void load2(int e, double &a, double &b) const
    {
        using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (unsigned)e * estride, 0);
        a = __hiloint2double((int)v.y, (int)v.x);
        b = __hiloint2double((int)v.w, (int)v.z);
    }
    // the mirror image for loads: one 16-byte load per lane, then the quads swap halves
    __device__ __forceinline__ void load_pair(int e, double &a, double &b) const
    {
        using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff2, (unsigned)e * estride, 0);
        // even quad holds [a(q), a(q+1)], odd quad [b(q-1), b(q)]
        const int ax = __builtin_amdgcn_update_dpp((int)v.x, (int)v.z, 0x114, 0xf, 0xa, false);   // odd: a(q) = even's high half
        const int ay = __builtin_amdgcn_update_dpp((int)v.y, (int)v.w, 0x114, 0xf, 0xa, false);
        const int bx = __builtin_amdgcn_update_dpp((int)v.z, (int)v.x, 0x104, 0xf, 0x5, false);   // even: b(q) = odd's low half
        const int by = __builtin_amdgcn_update_dpp((int)v.w, (int)v.y, 0x104, 0xf, 0x5, false);
        a = __hiloint2double(ay, ax);
        b = __hiloint2double(by, bx);
    }
};

__device__ __forceinline__ void ml_wave_fence()
{
    // LDS operations of one wave execute in order; this only keeps the compiler from moving the tile
    // writes and the transposed reads across each other
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}


// Copy-out of U 16-byte units staged in a wave-private LDS tile: the reads of a batch of B units FIRST, then their stores.
// (read -> s_waitcnt -> store per unit, which is what a plain loop compiles to, exposes the LDS latency once per unit:
// 13 .. 32 times per output set, with one wave per SIMD nothing covers it.)  No lane is predicated on the unit count: a
// lane past the last unit reads a clamped unit, and `store(unit, in_range, v)` is expected to drop it (an offset outside
// the descriptor).  Up to 2 B units: one straight-line batch pair; above: a rolled loop over batches.
//   addr(unit): tile address of the unit;  store(unit, in_range, value)
template <int U, int B, class Addr, class Store>
__device__ __forceinline__ void ml_copy_units(unsigned lane, Addr &&addr, Store &&store)
{
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
    constexpr int IT = (U + 63) / 64;
    auto batch = [&](int it0, auto nb_tag) {
        constexpr int NB = decltype(nb_tag)::value;
        u32x4 v[NB];
        FK_UNROLL for (int b = 0; b < NB; ++b) {
            const unsigned unit = (unsigned)(it0 + b) * 64u + lane, cu = unit < (unsigned)U ? unit : (unsigned)U - 1u;
            v[b] = *reinterpret_cast<const u32x4 *>(addr(cu));
        }
        __builtin_amdgcn_sched_barrier(0);
        FK_UNROLL for (int b = 0; b < NB; ++b) {
            const unsigned unit = (unsigned)(it0 + b) * 64u + lane;
            store(unit, unit < (unsigned)U, v[b]);
        }
    };
    if constexpr (IT <= 2 * B) {
        constexpr int H1 = IT < B ? IT : B;
        batch(0, std::integral_constant<int, H1>{});
        if constexpr (IT > B) batch(B, std::integral_constant<int, IT - B>{});
    } else {
        _Pragma("nounroll") for (int it0 = 0; it0 < IT; it0 += B) batch(it0, std::integral_constant<int, B>{});
    }
}
// the offset of a dropped store: outside every descriptor SIZED TO ITS SLAB (the NumPy-order copy-outs; a descriptor made by
// make_rsrc spans 4 GiB and drops nothing -- the element-major copy-outs predicate their stores instead)
constexpr unsigned ML_OFF_DROP = 0xfffffff0u;

// Element-major copy-out of a wave's staged block: tile[e * TPW + g] (element e of the wave's track g), E elements, TPW tracks
// (HP = TPW / 2 track pairs).  Read as 16-byte units the tile is LINEAR (unit u = element u / HP, pair u % HP sits at byte 16 u),
// and in memory unit u lands at (u / HP) * n8 + (u % HP) * 16 from the wave's first track of element 0: with u = 64 it + lane
// that is one per-lane offset (lane / HP) * n8 + (lane % HP) * 16 plus a wave-uniform (64 / HP) it n8, and whether a lane's track pair exists
// (2 p + 1 < valid; == valid: its first track only) does not depend on `it` at all -- ONE predicate per copy-out, hoisted
// around each batch of stores, instead of a division, two address computations and two exec regions per unit (the unit
// addresses of a plain loop were held in ~20 VGPRs across the time loop or spilled and reloaded behind a vmcnt(0)).
// NOBRANCH: no exec region at all -- the descriptor ends 32 bytes short of 4 GiB (the host refuses banks whose step block
// reaches that far), a lane without a track pair stores to ML_OFF_DROP, beyond it, and the odd tail's 8-byte store is issued
// by every lane with the same select.  Twice the store instructions (half of them dropped by the range check), but straight-
// line code: the by-product histories of the EX instantiations, where four more predicated regions per step cost 1.5 KB of
// scratch per lane at dim_x 16.
template <int E, int TPW = 16, int B = 4, bool NOBRANCH = false>
__device__ __forceinline__ void ml_slab_out_soa(const double *tile, const double *first, unsigned n8, unsigned lane, unsigned valid)
{
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
    constexpr int HP = TPW / 2, EPI = 64 / HP;                    // track pairs per element; elements per store instruction
    static_assert(TPW % 2 == 0 && 64 % HP == 0, "whole elements per store instruction");
    constexpr int U = E * HP, IT = (U + 63) / 64;
    const rsrc_t rs = NOBRANCH ? __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(first), 0, (int)0xffffffe0u, 0x00020000) : make_rsrc(first);
    const unsigned p = lane % (unsigned)HP, voff = (lane / (unsigned)HP) * n8 + p * 16u;
    const bool full = 2u * p + 1u < valid, half = 2u * p + 1u == valid;
    auto batch = [&](int it0, auto nb_tag) {
        constexpr int NB = decltype(nb_tag)::value;
        u32x4 v[NB];
        FK_UNROLL for (int b = 0; b < NB; ++b) {
            const unsigned unit = (unsigned)(it0 + b) * 64u + lane, cu = unit < (unsigned)U ? unit : (unsigned)U - 1u;
            v[b] = *reinterpret_cast<const u32x4 *>(tile + 2u * cu);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (NOBRANCH) {
            FK_UNROLL for (int b = 0; b < NB; ++b) {
                const unsigned unit = (unsigned)(it0 + b) * 64u + lane, off = voff + (unsigned)(it0 + b) * (unsigned)EPI * n8;
                const bool ok = U % 64 == 0 || unit < (unsigned)U;
                __builtin_amdgcn_raw_buffer_store_b128(v[b], rs, ok && full ? off : ML_OFF_DROP, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b64(u32x2{v[b].x, v[b].y}, rs, ok && half ? off : ML_OFF_DROP, 0, 0);
            }
        } else if (full) {
            FK_UNROLL for (int b = 0; b < NB; ++b) {
                const unsigned unit = (unsigned)(it0 + b) * 64u + lane;
                if (U % 64 == 0 || unit < (unsigned)U)
                    __builtin_amdgcn_raw_buffer_store_b128(v[b], rs, voff + (unsigned)(it0 + b) * (unsigned)EPI * n8, 0, 0);
            }
        } else if (half) {                                         // an odd tail: the last wave of a launch only
            FK_UNROLL for (int b = 0; b < NB; ++b) {
                const unsigned unit = (unsigned)(it0 + b) * 64u + lane;
                if (U % 64 == 0 || unit < (unsigned)U)
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2{v[b].x, v[b].y}, rs, voff + (unsigned)(it0 + b) * (unsigned)EPI * n8, 0, 0);
            }
        }
    };
    if constexpr (IT <= 2 * B) {
        constexpr int H1 = IT < B ? IT : B;
        batch(0, std::integral_constant<int, H1>{});
        if constexpr (IT > B) batch(B, std::integral_constant<int, IT - B>{});
    } else {
        constexpr int ITF = IT / B * B;
        _Pragma("nounroll") for (int it0 = 0; it0 < ITF; it0 += B) batch(it0, std::integral_constant<int, B>{});
        if constexpr (IT > ITF) batch(ITF, std::integral_constant<int, IT - ITF>{});
    }
}

// SOA (element-major, a[e][track]) output of one row-block matrix of a wave's TPW consecutive tracks: the lanes write
// their rows into a wave-private LDS tile laid out [element][track], then the 64 lanes copy 16-byte units -- two
// adjacent tracks of one element -- so that a store instruction moves 1 KiB instead of 512 B.  (A wave may have 63
// vector-memory operations in flight whatever their size: with 8-byte stores that, not HBM, bounds the write stream
// of the several-lanes-per-track kernels -- their AOS twins, which always left through a slab, ran 1.3-1.4x faster.)
//   plane0: address of element 0, track 0 of the destination array (this time step); N: tracks per element plane;
//   w0: first track of the wave; g: this lane's track within the wave; valid: how many of the wave's tracks exist.
// Units may be 8-byte aligned only (odd N): buffer stores need dword alignment.  A unit whose second track does not
// exist (odd `valid`, last wave only) is written as 8 bytes.
template <int R, int NX, int TPW>
__device__ __forceinline__ void ml_store_rows_soa_slab(const double (&M)[R][NX], const unsigned (&row)[R], double *plane0, long N,
                                                       long w0, double *tile, unsigned lane, unsigned g, unsigned valid)
{
    constexpr int EP = NX * NX;
    ml_wave_fence();
    FK_UNROLL for (int r = 0; r < R; ++r)
        FK_UNROLL for (int c = 0; c < NX; ++c) tile[(row[r] * NX + c) * TPW + g] = M[r][c];
    ml_wave_fence();
    ml_slab_out_soa<EP, TPW>(tile, plane0 + w0, (unsigned)N * 8u, lane, valid);
    ml_wave_fence();
}

// Copy-out of a per-track record array of E doubles (the update's by-products: y, K, S, SI) that the lanes of a wave have
// just staged in the wave-private tile -- 16-byte units, 1 KiB per store instruction, like the (x, P) sets above.
//   AOS: the tile is laid out like the wave's slab ([track][e], tile[g * E + e]); dst = record of the wave's first track.
//   SOA: the tile is [e][TPW] (tile[e * TPW + g]); plane0 = element 0 of the array at this time step, track 0.
// The caller brackets its tile writes with ml_wave_fence(); `valid`: how many of the wave's TPW tracks exist.
template <int E, int TPW>
__device__ __forceinline__ void ml_tile_out_aos(double *dst, const double *tile, unsigned lane, unsigned valid)
{
    constexpr int U = TPW * E / 2;
    static_assert((TPW * E) % 2 == 0, "whole 16-byte units");
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(dst, 0, (int)(valid * (unsigned)E * 8u), 0x00020000);
    ml_copy_units<U, 4>(lane, [&](unsigned unit) { return tile + 2u * unit; },
                        [&](unsigned unit, bool ok, const u32x4 &v) {
                            __builtin_amdgcn_raw_buffer_store_b128(v, rs, ok ? unit * 16u : ML_OFF_DROP, 0, 0);
                        });
}

template <int E, int TPW>
__device__ __forceinline__ void ml_tile_out_soa(double *plane0, long N, long w0, const double *tile, unsigned lane, unsigned valid)
{
    ml_slab_out_soa<E, TPW, 4, true>(tile, plane0 + w0, (unsigned)N * 8u, lane, valid);
}

}  // namespace fk
