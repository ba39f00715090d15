// fk_device.hpp -- record addressing, padded register loads/stores and host-side
// launch helpers shared by the gfx950 kernels of libfilterhip.
//
// Record addressing (see include/filterhip.h):
//   AOS  a[i][e]   element e of track i at  i*E + e     (NumPy C order)
//   SOA  a[e][i]   element e of track i at  e*N + i     (lane-coalesced)
// A kernel instantiated for <NX,NZ> also serves any runtime n <= NX, m <= NZ by
// padding in registers (P,F with identity, Q,H with zeros, R with identity,
// vectors with zeros): the padded block never couples to the real block and
// every extra FMA term is an exact zero, so results are bit-identical to the
// unpadded arithmetic.  EXACT instantiations (n == NX, m == NZ) drop the guards.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fk_math.hpp"

namespace fk {

constexpr int LAYOUT_AOS = 0;
constexpr int LAYOUT_SOA = 1;
constexpr int BLOCK = 256;   // 4 wavefronts of 64 tracks

// Position of a lane inside the record arrays.  All record traffic goes through
// raw buffer loads/stores: per (array, time step) ONE wave-uniform 128-bit descriptor
// (SGPRs; base = record block advanced to the workgroup) plus ONE 32-bit per-lane byte
// offset; the element index goes into the instruction's immediate offset (AOS) or the
// scalar soffset e*N*8 (SOA).  (Written as 64-bit per-lane pointer arithmetic instead,
// hipcc hoists one loop-invariant 64-bit VGPR pointer per output element out of the
// time loop -- 80+ VGPRs at dim_x=4 -- and spills.)
// Limit that follows: one step's record block must stay below 4 GiB (N*E*8 < 2^32);
// the host launcher checks it.
struct Lane {
    long blk0;      // first track of this workgroup (uniform)
    unsigned tid;   // lane's track offset inside the workgroup
    long N;         // tracks (uniform)
};

using rsrc_t = __amdgpu_buffer_rsrc_t;

// The wave's index in its workgroup as a SCALAR.  threadIdx.x >> 6 is the same in every lane of a wave, but the compiler
// does not know it: everything derived from it (the wave's first track, the base address and the size of the slab it
// writes) sits in VGPRs, a buffer descriptor built from those is "divergent", and every access through it is wrapped in
// a waterfall loop (4 v_readfirstlane + 2 v_cmp + s_and_saveexec + a branch per store: 26 of them per step in the
// NumPy-order three-lane kernel, each splitting the time loop's basic block).  One v_readfirstlane here makes the whole
// chain scalar (round 4; found in the ISA of the LDS-exchange build).
__device__ __forceinline__ unsigned wave_index()
{
    return (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
}

// ... and for a pointer / a count that IS the same in every lane but reaches a descriptor through registers the compiler
// treats as divergent: three v_readfirstlane per descriptor instead of a waterfall loop per access
template <class T>
__device__ __forceinline__ T *uniform_ptr(T *p)
{
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return reinterpret_cast<T *>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ int uniform_int(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ rsrc_t make_rsrc(const void *p)
{
    // raw buffer (stride 0), no clipping: tail lanes exit before touching memory.  The base is wave-uniform by construction
    // at every call site; uniform_ptr makes the compiler know it (where it already does, the two v_readfirstlane fold away)
    return __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(const_cast<void *>(p)), /*stride*/ 0, /*num_records*/ -1, 0x00020000);
}

using u32x2 = __attribute__((ext_vector_type(2))) unsigned;


// Per-lane view of one record block `blk` ([N][E] AOS / [E][N] SOA) for the lane's workgroup.
template <int LAYOUT>
struct RecView {
    rsrc_t rs;
    unsigned voff;        // per-lane byte offset
    unsigned estride;     // SOA: N*8 (bytes between consecutive elements), uniform
    // present = false: a descriptor of ZERO records -- every access through it is out of range, i.e. loads return 0 and stores
    // are dropped by the hardware, but they are still ISSUED: an optional output costs no branch around its stores, the
    // number of stores of a time step is a constant and the compiler can count them in its s_waitcnt (a load requested in
    // front of them is then waited for with vmcnt(#stores) instead of vmcnt(0))
    __device__ __forceinline__ RecView(const double *blk, const Lane &ln, int E, bool present)
        : RecView(blk, ln, E)
    {
        if (!present) rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(blk), 0, 0, 0x00020000);
    }
    __device__ __forceinline__ RecView(const double *blk, const Lane &ln, int E)
    {
        if (LAYOUT == LAYOUT_AOS) {
            rs = make_rsrc(blk + ln.blk0 * (long)E);
            voff = ln.tid * (unsigned)E * 8u;
            estride = 0;
        } else {
            rs = make_rsrc(blk + ln.blk0);
            voff = ln.tid * 8u;
            estride = (unsigned)ln.N * 8u;
            // Opaque to the optimiser: otherwise every e*estride (one SGPR per record element, 100+
            // at dim_x = 9) is hoisted out of the time loop and the scalar file spills; recomputing
            // the product is one s_mul_i32 next to each store.
            asm volatile("" : "+s"(estride));
        }
    }
    __device__ __forceinline__ double load(int e) const
    {
        const u32x2 v = (LAYOUT == LAYOUT_AOS)
            ? __builtin_amdgcn_raw_buffer_load_b64(rs, voff + (unsigned)e * 8u, 0, 0)
            : __builtin_amdgcn_raw_buffer_load_b64(rs, voff, (unsigned)e * estride, 0);
        return __builtin_bit_cast(double, v);
    }
    __device__ __forceinline__ void store(int e, double x) const
    {
        const u32x2 v = __builtin_bit_cast(u32x2, x);
        if (LAYOUT == LAYOUT_AOS) __builtin_amdgcn_raw_buffer_store_b64(v, rs, voff + (unsigned)e * 8u, 0, 0);
        else __builtin_amdgcn_raw_buffer_store_b64(v, rs, voff, (unsigned)e * estride, 0);
    }
    // AOS only: elements e and e + 1 of a record are adjacent in memory -> one 16-byte access.  A wave
    // may have 63 vector-memory operations in flight whatever their size, so on the lane-strided AOS
    // paths twice the bytes per operation is close to twice the bandwidth.
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
    __device__ __forceinline__ void load2(int e, double &a, double &b) const
    {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + (unsigned)e * 8u, 0, 0);
        a = __builtin_bit_cast(double, u32x2{v.x, v.y});
        b = __builtin_bit_cast(double, u32x2{v.z, v.w});
    }
    __device__ __forceinline__ void store2(int e, double a, double b) const
    {
        const u32x2 lo = __builtin_bit_cast(u32x2, a), hi = __builtin_bit_cast(u32x2, b);
        const u32x4 v = {lo.x, lo.y, hi.x, hi.y};
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff + (unsigned)e * 8u, 0, 0);
    }
};

// Load an r x c matrix record of the lane's track (C order, row stride c) into a ROWS x COLS
// register tile, padding with diag_pad on the diagonal and 0 elsewhere.
template <int ROWS, int COLS, int LAYOUT, bool EXACT>
__device__ __forceinline__ void load_rec(double (&M)[ROWS * COLS], const double *__restrict__ blk,
                                         const Lane &ln, int r, int c, double diag_pad)
{
    const int E = EXACT ? ROWS * COLS : r * c;
    const RecView<LAYOUT> v(blk, ln, E);
    if constexpr (EXACT && LAYOUT == LAYOUT_AOS && ROWS * COLS >= 2) {
        FK_UNROLL for (int e = 0; e + 1 < ROWS * COLS; e += 2) v.load2(e, M[e], M[e + 1]);
        if ((ROWS * COLS) % 2) M[ROWS * COLS - 1] = v.load(ROWS * COLS - 1);
        return;
    }
    FK_UNROLL for (int a = 0; a < ROWS; ++a) {
        FK_UNROLL for (int b = 0; b < COLS; ++b) {
            if (EXACT || (a < r && b < c))
                M[a * COLS + b] = v.load(EXACT ? a * COLS + b : a * c + b);
            else
                M[a * COLS + b] = (a == b) ? diag_pad : 0.0;
        }
    }
}

template <int ROWS, int COLS, int LAYOUT, bool EXACT>
__device__ __forceinline__ void store_rec(const double (&M)[ROWS * COLS], double *__restrict__ blk,
                                          const Lane &ln, int r, int c, bool present = true)
{
    const int E = EXACT ? ROWS * COLS : r * c;
    const RecView<LAYOUT> v(blk, ln, E, present);
    if constexpr (EXACT && LAYOUT == LAYOUT_AOS && ROWS * COLS >= 2) {
        FK_UNROLL for (int e = 0; e + 1 < ROWS * COLS; e += 2) v.store2(e, M[e], M[e + 1]);
        if ((ROWS * COLS) % 2) v.store(ROWS * COLS - 1, M[ROWS * COLS - 1]);
        return;
    }
    FK_UNROLL for (int a = 0; a < ROWS; ++a) {
        FK_UNROLL for (int b = 0; b < COLS; ++b) {
            if (EXACT || (a < r && b < c))
                v.store(EXACT ? a * COLS + b : a * c + b, M[a * COLS + b]);
        }
    }
}

// Model shared by all tracks, staged in LDS (padded to NX/NZ) and broadcast-read one row
// at a time (all lanes read the same address: conflict-free, one ds_read_b128 per 2 doubles).
// Layout in LDS: F[NX*NX] | Q[NX*NX] | H[NZ*NX] | R[NZ*NZ].
template <int NX, int NZ>
struct LdsModel {
    static constexpr int OFF_F = 0, OFF_Q = NX * NX, OFF_H = 2 * NX * NX, OFF_R = 2 * NX * NX + NZ * NX;
    static constexpr int SIZE = OFF_R + NZ * NZ;
    static constexpr bool IN_LDS = true;          // fk_math_sym.hpp: model_cached
    const double *s;
    template <int LEN>
    __device__ __forceinline__ void row(int off, double (&r)[LEN]) const
    {
        FK_UNROLL for (int j = 0; j < LEN; ++j) r[j] = s[off + j];
    }
    __device__ __forceinline__ void rowF(int i, double (&r)[NX]) const { row<NX>(OFF_F + i * NX, r); }
    __device__ __forceinline__ void rowQ(int i, double (&r)[NX]) const { row<NX>(OFF_Q + i * NX, r); }
    __device__ __forceinline__ void rowH(int i, double (&r)[NX]) const { row<NX>(OFF_H + i * NX, r); }
    __device__ __forceinline__ void rowR(int i, double (&r)[NZ]) const { row<NZ>(OFF_R + i * NZ, r); }
};

// Cooperative fill of one padded ROWS x COLS matrix in LDS from an r x c matrix in global
// memory (src may be NULL: pure padding).  Caller synchronises.
template <int ROWS, int COLS>
__device__ __forceinline__ void lds_fill(double *dst, const double *__restrict__ src, int r, int c,
                                         double diag_pad, unsigned tid)
{
    for (unsigned k = tid; k < (unsigned)(ROWS * COLS); k += BLOCK) {
        const int a = (int)k / COLS, b = (int)k % COLS;
        dst[k] = (src != nullptr && a < r && b < c) ? src[a * c + b] : ((a == b) ? diag_pad : 0.0);
    }
}

template <int LEN>
__device__ __forceinline__ bool all_finite(const double (&v)[LEN])
{
    bool ok = true;
    FK_UNROLL for (int k = 0; k < LEN; ++k) ok = ok && (fabs(v[k]) <= 1.79769313486231570815e+308);
    return ok;
}

using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

__device__ __forceinline__ void wave_lds_fence()
{
    // LDS operations of one wave execute in order; this only stops the compiler from
    // reordering the tile writes and the transposed reads around each other.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Wave-cooperative store of one LEN-double record per lane into an AOS block:
// lane l owns the record of track (block_first + wave_row0 + l); `slab` is the workgroup's slab
// of the output array.  The descriptor is sized to the block's valid rows, so in the tail workgroup
// the hardware range check drops the stores of rows past the last track (AOS offsets are
// voffset + immediate: no scalar offset takes part in the check) -- no store is predicated.
// The tile is written row-per-lane (row stride LEN|1 doubles: conflict-free ds_write_b64) and read
// back in memory order, two consecutive doubles per lane per pass -> buffer_store_dwordx4,
// 1 KiB contiguous per instruction.  When LEN divides 128 every pass uses the same lane-dependent
// base addresses plus compile-time immediates (2 VGPRs of addressing for the whole record).
template <int LEN>
__device__ __forceinline__ void wave_store_aos(const double (&v)[LEN], const double *slab, unsigned wave_row0,
                                               double *tile, unsigned lane, unsigned last_row)
{
    constexpr int LENP = LEN | 1;
    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(slab), 0,
                                                        (int)((last_row + 1u) * (unsigned)LEN * 8u), 0x00020000);
    FK_UNROLL for (int e = 0; e < LEN; ++e) tile[lane * LENP + e] = v[e];
    wave_lds_fence();
    if constexpr (LEN % 2 == 0 && 128 % LEN == 0) {
        constexpr int PASSES = LEN / 2;          // 64*LEN doubles, 128 per pass
        constexpr int RPP = 128 / LEN;           // rows per pass
        const unsigned r0 = (lane * 2u) / LEN, col = (lane * 2u) % LEN;
        const double *tb = tile + r0 * LENP + col;
        const unsigned gb = ((wave_row0 + r0) * LEN + col) * 8u;
        FK_UNROLL for (int it = 0; it < PASSES; ++it) {
            const double a = tb[it * RPP * LENP], b = tb[it * RPP * LENP + 1];
            const u32x2 ua = __builtin_bit_cast(u32x2, a), ub = __builtin_bit_cast(u32x2, b);
            const u32x4 w = {ua.x, ua.y, ub.x, ub.y};
            __builtin_amdgcn_raw_buffer_store_b128(w, rs, gb + (unsigned)(it * RPP * LEN * 8), 0, 0);
        }
    } else if constexpr (LEN % 2 == 0) {
        constexpr int PASSES = LEN / 2;
        FK_UNROLL for (int it = 0; it < PASSES; ++it) {
            const unsigned q = it * 128u + lane * 2u;
            const unsigned row = q / LEN, col = q % LEN;
            const double a = tile[row * LENP + col], b = tile[row * LENP + col + 1];
            const u32x2 ua = __builtin_bit_cast(u32x2, a), ub = __builtin_bit_cast(u32x2, b);
            const u32x4 w = {ua.x, ua.y, ub.x, ub.y};
            __builtin_amdgcn_raw_buffer_store_b128(w, rs, ((wave_row0 + row) * LEN + col) * 8u, 0, 0);
        }
    } else {
        FK_UNROLL for (int it = 0; it < LEN; ++it) {
            const unsigned q = it * 64u + lane;
            const unsigned row = q / LEN, col = q % LEN;
            const double a = tile[row * LENP + col];
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, a), rs, ((wave_row0 + row) * LEN + col) * 8u, 0, 0);
        }
    }
    wave_lds_fence();
}

// wave_store_aos with a run-time record pitch (doubles between the records of consecutive tracks, >= LEN): the wave's 64
// records are LEN-double islands `pitch` apart (FK_KF_FLAG_COV_INTERLEAVED: the other half of each period is the partner
// array's record).  pitch == LEN is wave_store_aos.  Every store instruction still writes whole records' worth of
// contiguous 16-byte units (LEN = 16: eight 128-byte records per instruction, each a full line).
template <int LEN>
__device__ __forceinline__ void wave_store_aos_pitch(const double (&v)[LEN], const double *slab, unsigned wave_row0,
                                                     double *tile, unsigned lane, unsigned last_row, unsigned pitch)
{
    constexpr int LENP = LEN | 1;
    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(slab), 0,
                                                        (int)((last_row * pitch + (unsigned)LEN) * 8u), 0x00020000);
    FK_UNROLL for (int e = 0; e < LEN; ++e) tile[lane * LENP + e] = v[e];
    wave_lds_fence();
    if constexpr (LEN % 2 == 0 && 128 % LEN == 0) {
        constexpr int PASSES = LEN / 2;          // 64*LEN doubles, 128 per pass
        constexpr int RPP = 128 / LEN;           // rows per pass
        const unsigned r0 = (lane * 2u) / LEN, col = (lane * 2u) % LEN;
        const double *tb = tile + r0 * LENP + col;
        const unsigned gb = ((wave_row0 + r0) * pitch + col) * 8u;
        const unsigned pass = (unsigned)RPP * pitch * 8u;                  // uniform
        FK_UNROLL for (int it = 0; it < PASSES; ++it) {
            const double a = tb[it * RPP * LENP], b = tb[it * RPP * LENP + 1];
            const u32x2 ua = __builtin_bit_cast(u32x2, a), ub = __builtin_bit_cast(u32x2, b);
            const u32x4 w = {ua.x, ua.y, ub.x, ub.y};
            __builtin_amdgcn_raw_buffer_store_b128(w, rs, gb + (unsigned)it * pass, 0, 0);
        }
    } else if constexpr (LEN % 2 == 0) {
        constexpr int PASSES = LEN / 2;
        FK_UNROLL for (int it = 0; it < PASSES; ++it) {
            const unsigned q = it * 128u + lane * 2u;
            const unsigned row = q / LEN, col = q % LEN;
            const double a = tile[row * LENP + col], b = tile[row * LENP + col + 1];
            const u32x2 ua = __builtin_bit_cast(u32x2, a), ub = __builtin_bit_cast(u32x2, b);
            const u32x4 w = {ua.x, ua.y, ub.x, ub.y};
            __builtin_amdgcn_raw_buffer_store_b128(w, rs, ((wave_row0 + row) * pitch + col) * 8u, 0, 0);
        }
    } else {
        FK_UNROLL for (int it = 0; it < LEN; ++it) {
            const unsigned q = it * 64u + lane;
            const unsigned row = q / LEN, col = q % LEN;
            const double a = tile[row * LENP + col];
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, a), rs, ((wave_row0 + row) * pitch + col) * 8u, 0, 0);
        }
    }
    wave_lds_fence();
}

// wave_store_aos for the inside of a time loop, even LEN: the tile is FLAT (row stride LEN doubles, i.e. laid out exactly
// like the wave's slab of the array), so the copy-out needs no row / column arithmetic at all -- one lane-dependent LDS
// address and one lane-dependent byte offset for the whole record, pass `it` adds the constant it * 1024 to both.  The
// padded tile of wave_store_aos costs a division per pass where LEN does not divide 128; hoisted out of a time loop
// those are two VGPRs of addressing per pass (36 at LEN = 36 -- the fused UKF spilled).  Price: the row-per-lane tile
// writes are bank-conflicted (row stride 2 LEN dwords); at one record per time step that is noise.
template <int LEN>
__device__ __forceinline__ void wave_store_aos_flat(const double (&v)[LEN], const double *slab, unsigned wave_row0,
                                                    double *tile, unsigned lane, unsigned last_row, bool present = true)
{
    static_assert(LEN % 2 == 0, "16-byte units");
    // (present = false: zero records -- the stores are issued and dropped, see RecView)
    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(slab), 0,
                                                        present ? (int)((last_row + 1u) * (unsigned)LEN * 8u) : 0, 0x00020000);
    wave_lds_fence();
    FK_UNROLL for (int e = 0; e < LEN; ++e) tile[lane * LEN + e] = v[e];
    wave_lds_fence();
    const double *tb = tile + lane * 2u;
    const unsigned gb = (wave_row0 * LEN + lane * 2u) * 8u;
    FK_UNROLL for (int it = 0; it < LEN / 2; ++it) {
        const u32x4 w = *reinterpret_cast<const u32x4 *>(tb + it * 128);
        __builtin_amdgcn_raw_buffer_store_b128(w, rs, gb + (unsigned)(it * 1024), 0, 0);
        // (a >8-byte store whose offset ends up in an SGPR is not covered by the compiler's hazard recogniser on
        //  gfx950: see store_data_hazard in fk_ml.hpp)
        asm volatile("s_nop 1" ::"v"(w.x), "v"(w.y), "v"(w.z), "v"(w.w) : "memory");
    }
    wave_lds_fence();
}

// Element-major ([element][track]) output of one LEN-double record per lane as 16-BYTE stores (round 4).  A lane-per-track
// 8-byte store writes 512 contiguous bytes of ONE element row per instruction; here an instruction writes two rows: lanes 0..31
// store tracks (2l, 2l+1) of row 2p, lanes 32..63 of row 2p+1 -- the same bytes in half as many vector-memory operations
// (a wave may have 63 in flight: a step of the fused UKF issues 42 + 4 of them, two steps' worth does not fit, and the stores of
// step t are what step t+1's issue then waits for).  The exchange goes through a wave-private LDS tile [LEN][64].
// Full waves only (no clipping), 16-byte aligned rows: `rows` = the array's first row at this wave's first track, N even.
template <int LEN>
__device__ __forceinline__ void wave_store_soa_pairs(const double (&v)[LEN], double *rows, unsigned n8 /* N * 8 */,
                                                     double *tile, unsigned lane, bool present = true, unsigned valid = 64u)
{
    static_assert(LEN % 2 == 0, "pairs of element rows");
    // `valid` (even): how many of the wave's 64 tracks exist.  A lane whose track pair does not exist stores to an offset
    // just below 4 GiB, outside the descriptor (32 bytes short of it; the host keeps every step block below that): no exec
    // region, and the last partial workgroup of a bank runs in the same launch as the full ones.  The whole offset sits in
    // the VGPR (no soffset: a dropped offset plus a scalar offset would wrap around into range).
    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(rows), 0, uniform_int(present ? (int)0xffffffe0u : 0), 0x00020000);
    wave_lds_fence();
    FK_UNROLL for (int e = 0; e < LEN; ++e) tile[e * 64 + lane] = v[e];
    wave_lds_fence();
    const unsigned half = lane >> 5, l2 = (lane & 31u) * 2u;
    const double *tb = tile + half * 64u + l2;
    const unsigned voff = l2 * 8u + half * n8;
    const bool ok = l2 + 1u < valid;
    FK_UNROLL for (int p = 0; p < LEN / 2; ++p) {
        const u32x4 w = *reinterpret_cast<const u32x4 *>(tb + p * 128);
        __builtin_amdgcn_raw_buffer_store_b128(w, rs, ok ? voff + (unsigned)(2 * p) * n8 : 0xfffffff0u, 0, 0);
    }
    wave_lds_fence();
}

// The mirror image: wave-cooperative LOAD of one LEN-double record per lane from an AOS block -- memory order into the
// tile (two consecutive doubles per lane per pass: buffer_load_dwordx4, 1 KiB contiguous per instruction; rows past the
// block's last track read as 0 through the descriptor's range check), then every lane reads its own row.  A lane-strided
// read of a 78-double record touches 64 different lines per instruction and thrashes the 16 KiB vector L1.
template <int LEN>
__device__ __forceinline__ void wave_load_aos(double (&v)[LEN], const double *slab, unsigned wave_row0, double *tile,
                                              unsigned lane, unsigned last_row)
{
    constexpr int LENP = LEN | 1;
    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(slab), 0,
                                                        (int)((last_row + 1u) * (unsigned)LEN * 8u), 0x00020000);
    if constexpr (LEN % 2 != 0) {
        // odd records: 8-byte units (512 B contiguous per instruction)
        u32x2 w1[LEN];
        FK_UNROLL for (int it = 0; it < LEN; ++it)
            w1[it] = __builtin_amdgcn_raw_buffer_load_b64(rs, (wave_row0 * LEN + it * 64u + lane) * 8u, 0, 0);
        FK_UNROLL for (int it = 0; it < LEN; ++it) {
            const unsigned q = it * 64u + lane;
            tile[(q / LEN) * LENP + q % LEN] = __builtin_bit_cast(double, w1[it]);
        }
        wave_lds_fence();
        FK_UNROLL for (int e = 0; e < LEN; ++e) v[e] = tile[lane * LENP + e];
        wave_lds_fence();
    } else {
        constexpr int PASSES = LEN / 2;
        u32x4 w[PASSES];
        FK_UNROLL for (int it = 0; it < PASSES; ++it) {
            const unsigned q = it * 128u + lane * 2u;
            w[it] = __builtin_amdgcn_raw_buffer_load_b128(rs, (wave_row0 * LEN + q) * 8u, 0, 0);
    }
    FK_UNROLL for (int it = 0; it < PASSES; ++it) {
        const unsigned q = it * 128u + lane * 2u;
        const unsigned row = q / LEN, col = q % LEN;
        tile[row * LENP + col] = __builtin_bit_cast(double, u32x2{w[it].x, w[it].y});
        tile[row * LENP + col + 1] = __builtin_bit_cast(double, u32x2{w[it].z, w[it].w});
    }
    wave_lds_fence();
    FK_UNROLL for (int e = 0; e < LEN; ++e) v[e] = tile[lane * LENP + e];
    wave_lds_fence();
    }
}

// wave_load_aos split in two for a fetch that is issued one time step ahead (even LEN, flat tile like wave_store_aos_flat):
// issue() puts the wave's slab of the array in flight, memory order, 1 KiB per instruction, into LEN / 2 16-byte register
// quads; row() -- a step later -- passes them through the tile and returns the lane's own record.  Rows past the block's
// last track read as zeros (the descriptor's range check).
template <int LEN>
struct WaveAosFetch {
    static_assert(LEN % 2 == 0, "16-byte units");
    u32x4 w[LEN / 2];
    __device__ __forceinline__ void issue(const double *slab, unsigned wave_row0, unsigned lane, unsigned last_row)
    {
        const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(slab), 0,
                                                            (int)((last_row + 1u) * (unsigned)LEN * 8u), 0x00020000);
        const unsigned gb = (wave_row0 * LEN + lane * 2u) * 8u;
        FK_UNROLL for (int it = 0; it < LEN / 2; ++it) w[it] = __builtin_amdgcn_raw_buffer_load_b128(rs, gb + (unsigned)(it * 1024), 0, 0);
    }
    // the fetched quads -> tile (flat); afterwards tile[lane * LEN + e] is element e of the lane's record.  The caller
    // fences (wave_lds_fence) after its reads, before the tile is written again.
    __device__ __forceinline__ void to_tile(double *tile, unsigned lane)
    {
        wave_lds_fence();
        FK_UNROLL for (int it = 0; it < LEN / 2; ++it) *reinterpret_cast<u32x4 *>(tile + lane * 2u + it * 128) = w[it];
        wave_lds_fence();
    }
};

// LDS-DMA (buffer_load_dwordx4 ... lds): 16 bytes per lane from global memory straight into LDS at (M0 base) + lane * 16 --
// no VGPR holds the data, so a fetch can stay in flight for a whole time step of a kernel that has no register to spare.
// hipcc does not count these (inline asm): the caller waits with its own s_waitcnt vmcnt before reading the LDS image, and
// with lgkmcnt(0) before re-targeting a region it has just read.  M0 is written in the statement that uses it and restored.
using dma_rsrc_t = __attribute__((ext_vector_type(4))) int;
__device__ __forceinline__ dma_rsrc_t make_dma_rsrc(const void *p, unsigned bytes)
{
    const unsigned long long b = reinterpret_cast<unsigned long long>(p);
    dma_rsrc_t r = {(int)(unsigned)b, (int)((unsigned)(b >> 32) & 0xffffu), (int)bytes, 0x00020000};
    r.x = __builtin_amdgcn_readfirstlane(r.x);
    r.y = __builtin_amdgcn_readfirstlane(r.y);
    r.z = __builtin_amdgcn_readfirstlane(r.z);
    return r;
}
__device__ __forceinline__ unsigned lds_address(const double *p)      // wave-uniform LDS byte address
{
    return __builtin_amdgcn_readfirstlane((unsigned)reinterpret_cast<unsigned long long>(
        (const __attribute__((address_space(3))) double *)p));
}
__device__ __forceinline__ void lds_dma16(dma_rsrc_t rs, unsigned voff, unsigned soff, unsigned lds_addr)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rs), "s"(soff), "s"(lds_addr) : "memory");
}

// the 4-byte form (dword alignment is all it asks of the global address): lane l's dword lands at (M0 base) + 4 l
__device__ __forceinline__ void lds_dma4(dma_rsrc_t rs, unsigned voff, unsigned soff, unsigned lds_addr)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rs), "s"(soff), "s"(lds_addr) : "memory");
}

// One NZ-double record per lane (a measurement), fetched a time step ahead by LDS-DMA: for kernels that run one wave per
// SIMD with no VGPR to spare -- a register prefetch there is parked in an AGPR the moment it is issued, i.e. waited for on the
// spot with vmcnt(0), behind every store of the previous step.  Lane l requests the two dwords of each of its OWN record's
// elements (2 NZ dword-DMA instructions per step; dword alignment is all they need; a lane that left the kernel or duplicates
// another track needs no special case); they land in planes [element][dword half][lane] of one of two images per wave.
// The caller waits (hipcc does not count these loads) with s_waitcnt vmcnt(k), k = min(63, vector-memory operations it has
// issued since the request) -- vmcnt retires in order -- before read().
template <int NZ, int LAYOUT>
struct LaneRecordDma {
    static constexpr int IMG_DOUBLES = NZ * 64;                  // per image
    unsigned lds, voff, soff_e;                                  // wave's LDS byte address; lane's byte offset; element stride (SOA) in bytes
    const unsigned *img0;
    unsigned lane;
    unsigned stride_doubles = IMG_DOUBLES;                       // doubles between the two images (a caller may keep more per image)
    // s_wave: this wave's 2 * IMG_DOUBLES doubles of LDS; rec: the lane's record index inside a step's block; N: records per block
    __device__ __forceinline__ void init(const double *s_wave, unsigned rec, unsigned N, unsigned lane_)
    {
        lds = lds_address(s_wave);
        img0 = reinterpret_cast<const unsigned *>(s_wave);
        lane = lane_;
        voff = LAYOUT == LAYOUT_AOS ? rec * (unsigned)NZ * 8u : rec * 8u;
        soff_e = LAYOUT == LAYOUT_AOS ? 0u : N * 8u;
    }
    // block: the step's [N][NZ] (AOS) / [NZ][N] (SOA) array; bytes: its size (the descriptor's range)
    __device__ __forceinline__ void request(const double *block, unsigned bytes, unsigned buf) const
    {
        const dma_rsrc_t rs = make_dma_rsrc(block, bytes);
        FK_UNROLL for (int c = 0; c < NZ; ++c)
            FK_UNROLL for (int h = 0; h < 2; ++h)
                lds_dma4(rs, voff + (LAYOUT == LAYOUT_AOS ? (unsigned)(c * 8) : 0u) + (unsigned)(h * 4), (unsigned)c * soff_e,
                         lds + buf * (stride_doubles * 8u) + (unsigned)((c * 2 + h) * 256));
    }
    __device__ __forceinline__ void read(unsigned buf, double (&z)[NZ]) const
    {
        const unsigned *img = img0 + buf * (stride_doubles * 2u);
        FK_UNROLL for (int c = 0; c < NZ; ++c)
            z[c] = __hiloint2double((int)img[(c * 2 + 1) * 64 + lane], (int)img[(c * 2) * 64 + lane]);
    }
};

// ---- host side ------------------------------------------------------------
void set_last_error(const char *msg);
int check_launch(const char *what);   // hipGetLastError -> FK_ERR_LAUNCH + message

}  // namespace fk
