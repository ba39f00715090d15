// resample_onepass.hpp -- entry of the one-pass resampling path (resample_onepass.hip), called by the C-ABI
// functions in resample_kernels.hip.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace fk {

size_t onepass_workspace_bytes(int64_t Fn, int64_t Np);

// systematic (u [Fn]) / stratified (u [Fn][Np]) resampling of Fn weight vectors of Np weights; `ws` must hold
// onepass_workspace_bytes(Fn, Np) bytes (it is zeroed here, on `s`).  Returns FK_OK / FK_ERR_*.
int onepass_launch(bool stratified, int64_t Fn, int64_t Np, const double *w, const double *u, int32_t *idx,
                   int32_t *status, void *ws, size_t ws_bytes, hipStream_t s);

// short vectors (Np below the one-pass threshold): one workgroup per filter, chunks in sequence, no workspace; writes
// status[f] itself (0 / ST_OVERRUN) and redoes garbage-weight filters with the reference's literal loop
int local_launch(bool stratified, int64_t Fn, int64_t Np, const double *w, const double *u, int32_t *idx,
                 int32_t *status, hipStream_t s);

// Np <= 8192 (resample_whole.hip): one workgroup takes the filter's whole vector in one round, no workspace; writes
// status[f] itself and hands vectors it cannot take to the reference's literal loop
bool whole_supported(int64_t Np);
int whole_launch(bool stratified, int64_t Fn, int64_t Np, const double *w, const double *u, int32_t *idx,
                 int32_t *status, hipStream_t s);

// after resample_kernel (short vectors): filters holding a negative / NaN / huge weight are redone by the reference's
// merge loop, literally (one thread each)
int literal_fixup_launch(bool stratified, int64_t Fn, int64_t Np, const double *w, const double *u, int32_t *idx,
                         int32_t *status, hipStream_t s);

}  // namespace fk
