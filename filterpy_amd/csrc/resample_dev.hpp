// resample_dev.hpp -- device helpers shared by the resampling kernels (resample_onepass.hip, resample_whole.hip): wave
// scans on DPP moves (no LDS traffic), lane broadcasts, and the reference's merge loop run literally by one thread.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fk_device.hpp"
#include "fk_exact_scan.hpp"

namespace fk {

using u64 = unsigned long long;
using i32x4 = __attribute__((ext_vector_type(4))) int;
using f64x2 = __attribute__((ext_vector_type(2))) double;

// ---- wave primitives (DPP: no LDS traffic) ---------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(double acc)
{
    const int lo = __double2loint(acc), hi = __double2hiint(acc);
    const int slo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);   // no source / masked row: +0.0
    const int shi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
    return acc + __hiloint2double(shi, slo);
}
// inclusive prefix sum over the 64 lanes (values whose partial sums are exact, or whose order is free)
__device__ __forceinline__ double wave_incl_sum(double v)
{
    v = dpp_add<0x111, 0xf>(v);   // row_shr:1
    v = dpp_add<0x112, 0xf>(v);   // row_shr:2
    v = dpp_add<0x114, 0xf>(v);   // row_shr:4
    v = dpp_add<0x118, 0xf>(v);   // row_shr:8 -> inclusive within rows of 16
    v = dpp_add<0x142, 0xa>(v);   // row_bcast:15 into rows 1, 3
    v = dpp_add<0x143, 0xc>(v);   // row_bcast:31 into rows 2, 3
    return v;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_max(int acc)
{
    const int s = __builtin_amdgcn_update_dpp(acc, acc, CTRL, ROW_MASK, 0xf, false);   // no source: itself
    return s > acc ? s : acc;
}
__device__ __forceinline__ int wave_incl_max(int v)
{
    v = dpp_max<0x111, 0xf>(v);
    v = dpp_max<0x112, 0xf>(v);
    v = dpp_max<0x114, 0xf>(v);
    v = dpp_max<0x118, 0xf>(v);
    v = dpp_max<0x142, 0xa>(v);
    v = dpp_max<0x143, 0xc>(v);
    return v;
}
// inclusive prefix sums over the 64 lanes of 64-bit (wrapping) and 32-bit integers
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ u64 dpp_add_u64(u64 acc)
{
    const int slo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)acc, CTRL, ROW_MASK, 0xf, false);
    const int shi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(acc >> 32), CTRL, ROW_MASK, 0xf, false);
    return acc + (((u64)(unsigned)shi << 32) | (u64)(unsigned)slo);
}
__device__ __forceinline__ u64 wave_incl_sum_u64(u64 v)
{
    v = dpp_add_u64<0x111, 0xf>(v);
    v = dpp_add_u64<0x112, 0xf>(v);
    v = dpp_add_u64<0x114, 0xf>(v);
    v = dpp_add_u64<0x118, 0xf>(v);
    v = dpp_add_u64<0x142, 0xa>(v);
    v = dpp_add_u64<0x143, 0xc>(v);
    return v;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_add_i32(int acc)
{
    return acc + __builtin_amdgcn_update_dpp(0, acc, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ int wave_incl_sum_i32(int v)
{
    v = dpp_add_i32<0x111, 0xf>(v);
    v = dpp_add_i32<0x112, 0xf>(v);
    v = dpp_add_i32<0x114, 0xf>(v);
    v = dpp_add_i32<0x118, 0xf>(v);
    v = dpp_add_i32<0x142, 0xa>(v);
    v = dpp_add_i32<0x143, 0xc>(v);
    return v;
}
__device__ __forceinline__ double lane_bcast(double v, int src)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}
__device__ __forceinline__ u64 lane_bcast_u64(u64 v, int src)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, src);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), src);
    return ((u64)hi << 32) | lo;
}

// the reference's merge loop, literally, by one thread (a filter holding a negative / NaN / huge weight)
template <bool STRATIFIED>
__device__ __forceinline__ int literal_merge(const double *wf, const double *u, long Np, int32_t *of)
{
    const double Nd = (double)Np;
    long i = 0, j = 0;
    double c = wf[0];
    while (i < Np) {
        const double ui = STRATIFIED ? u[i] : u[0];
        const double p = (ui + (double)i) / Nd;
        if (p < c) {
            of[i] = (int32_t)j;
            ++i;
        } else {
            ++j;
            if (j == Np) break;
            c = c + wf[j];
        }
    }
    int st = 0;
    for (; i < Np; ++i) {
        of[i] = (int32_t)(Np - 1);
        st = ST_OVERRUN;
    }
    return st;
}

}  // namespace fk
