// hostcheck_rs.cpp -- TEST ONLY: the resampler's position / boundary arithmetic (filterpy_amd/csrc/fk_resample_math.hpp,
// the same __host__ __device__ source the kernels use) compiled for the host with -ffp-contract=off, so that it can
// be held against brute-force counts with real IEEE divisions (numpy) in the GPU-less container.
#include <stdint.h>

#include "../../filterpy_amd/csrc/fk_resample_math.hpp"

extern "C" {

// out[k] = n(c[k]) for the systematic positions fl(fl(u + i) / Np)
void hc_n_boundary_sys(int Np, double u, long K, const double *c, int *out)
{
    const double Nd = (double)Np, h = 0.5 * Nd;
    for (long k = 0; k < K; ++k) out[k] = fk::n_boundary<false>(c[k], Np, Nd, h, u, nullptr);
}

// stratified positions fl(fl(u[i] + i) / Np)
void hc_n_boundary_strat(int Np, const double *u, long K, const double *c, int *out)
{
    const double Nd = (double)Np, h = 0.5 * Nd;
    for (long k = 0; k < K; ++k) out[k] = fk::n_boundary<true>(c[k], Np, Nd, h, 0.0, u);
}

void hc_n_boundary_fast_sys(int Np, double u, long K, const double *c, int *out)
{
    const double Nd = (double)Np, h = 0.5 * Nd;
    for (long k = 0; k < K; ++k) out[k] = fk::n_boundary_fast<false>(c[k], Np, Nd, h, u, nullptr);
}

void hc_n_boundary_fast_strat(int Np, const double *u, long K, const double *c, int *out)
{
    const double Nd = (double)Np, h = 0.5 * Nd;
    for (long k = 0; k < K; ++k) out[k] = fk::n_boundary_fast<true>(c[k], Np, Nd, h, 0.0, u);
}

// single comparison: fl(a / N) >= c
int hc_pos_ge(double a, double c, double Nd) { return fk::pos_ge(a, c, Nd, 0.5 * Nd) ? 1 : 0; }

}  // extern "C"
