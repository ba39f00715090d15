// imm_dispatch.cpp -- C ABI entry of the batched IMM estimator: argument checks and the choice of
// the (dim_x, dim_z, n_models) instantiation of imm_kernels.hip.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "fk_chunks.hpp"
#include "fk_device.hpp"
#include "fk_kernel_args.hpp"
#include "../../include/filterhip.h"

using namespace fk;

#define FK_IMM_INST(NX, NZ, NM, W) void launch_imm_##NX##_##NZ##_##NM(const ImmArgs &, int, int, hipStream_t);
#include "fk_dims_imm.def"
#undef FK_IMM_INST
// imm_lanes.hip: one lane per filter of a bank (round 6) -- the class (9, 4), every bank size 2..16
#define FK_IL_DECL1(NX, NZ, X)                                                                                         \
    int launch_imm_lanes_##NX##_##NZ##_g2_x##X(const ImmArgs &, int, int, hipStream_t);                                 \
    int launch_imm_lanes_##NX##_##NZ##_g4_x##X(const ImmArgs &, int, int, hipStream_t);                                 \
    int launch_imm_lanes_##NX##_##NZ##_g8_x##X(const ImmArgs &, int, int, hipStream_t);                                 \
    int launch_imm_lanes_##NX##_##NZ##_g16_x##X(const ImmArgs &, int, int, hipStream_t);                                \
    static int launch_imm_lanes_##NX##_##NZ##_x##X(const ImmArgs &a, int nm, int layout, hipStream_t s)                 \
    {                                                                                                                  \
        return nm <= 2 ? launch_imm_lanes_##NX##_##NZ##_g2_x##X(a, nm, layout, s)                                      \
             : nm <= 4 ? launch_imm_lanes_##NX##_##NZ##_g4_x##X(a, nm, layout, s)                                      \
             : nm <= 8 ? launch_imm_lanes_##NX##_##NZ##_g8_x##X(a, nm, layout, s)                                      \
                       : launch_imm_lanes_##NX##_##NZ##_g16_x##X(a, nm, layout, s);                                    \
    }
// (x0: the plain call; x1: the instantiation that also carries MMAE, missing measurements and the control input)
#define FK_IL_DECL(NX, NZ)                                                                                             \
    FK_IL_DECL1(NX, NZ, 0)                                                                                             \
    FK_IL_DECL1(NX, NZ, 1)                                                                                             \
    static int launch_imm_lanes_##NX##_##NZ(const ImmArgs &a, int nm, int layout, hipStream_t s)                       \
    {                                                                                                                  \
        const bool ext = a.mmae || a.mask || a.ll0 || a.nu > 0 || a.phase != FK_IMM_STEP;                              \
        return ext ? launch_imm_lanes_##NX##_##NZ##_x1(a, nm, layout, s) : launch_imm_lanes_##NX##_##NZ##_x0(a, nm, layout, s); \
    }
FK_IL_DECL(4, 2)
FK_IL_DECL(6, 3)
FK_IL_DECL(9, 4)
#undef FK_IL_DECL
#undef FK_IL_DECL1
// imm_quad.hip: four lanes per filter -- the classes (12, 4) and (16, 8), every bank size 2..16
#define FK_IQ_DECL1(NX, NZ, X)                                                                                         \
    int launch_imm_quad_##NX##_##NZ##_g2_x##X(const ImmArgs &, int, int, hipStream_t);                                  \
    int launch_imm_quad_##NX##_##NZ##_g4_x##X(const ImmArgs &, int, int, hipStream_t);                                  \
    int launch_imm_quad_##NX##_##NZ##_g8_x##X(const ImmArgs &, int, int, hipStream_t);                                  \
    int launch_imm_quad_##NX##_##NZ##_g16_x##X(const ImmArgs &, int, int, hipStream_t);                                 \
    static int launch_imm_quad_##NX##_##NZ##_x##X(const ImmArgs &a, int nm, int layout, hipStream_t s)                  \
    {                                                                                                                  \
        return nm <= 2 ? launch_imm_quad_##NX##_##NZ##_g2_x##X(a, nm, layout, s)                                       \
             : nm <= 4 ? launch_imm_quad_##NX##_##NZ##_g4_x##X(a, nm, layout, s)                                       \
             : nm <= 8 ? launch_imm_quad_##NX##_##NZ##_g8_x##X(a, nm, layout, s)                                       \
                       : launch_imm_quad_##NX##_##NZ##_g16_x##X(a, nm, layout, s);                                     \
    }
#define FK_IQ_DECL(NX, NZ)                                                                                             \
    FK_IQ_DECL1(NX, NZ, 0)                                                                                             \
    FK_IQ_DECL1(NX, NZ, 1)                                                                                             \
    static int launch_imm_quad_##NX##_##NZ(const ImmArgs &a, int nm, int layout, hipStream_t s)                        \
    {                                                                                                                  \
        const bool ext = a.mmae || a.mask || a.ll0 || a.nu > 0 || a.phase != FK_IMM_STEP;                              \
        return ext ? launch_imm_quad_##NX##_##NZ##_x1(a, nm, layout, s) : launch_imm_quad_##NX##_##NZ##_x0(a, nm, layout, s); \
    }
FK_IQ_DECL(12, 4)
FK_IQ_DECL(16, 8)
// the same unit built with EIGHT lanes per filter (FK_IQ_LPF=8): the class (16, 8), banks of two filters -- where it is the faster one
// (x, P, mu out: 7.2 -> 6.4 ms per 1e6 bank-steps; banks of four: the same, of eight: slower; docs/KERNEL_NOTES.md)
int launch_imm_oct_16_8_g2_x0(const ImmArgs &, int, int, hipStream_t);
int launch_imm_oct_16_8_g2_x1(const ImmArgs &, int, int, hipStream_t);
static int launch_imm_oct_16_8(const ImmArgs &a, int nm, int layout, hipStream_t s)
{
    const bool ext = a.mmae || a.mask || a.ll0 || a.nu > 0 || a.phase != FK_IMM_STEP;
    return ext ? launch_imm_oct_16_8_g2_x1(a, nm, layout, s) : launch_imm_oct_16_8_g2_x0(a, nm, layout, s);
}
#undef FK_IQ_DECL
#undef FK_IQ_DECL1

static int fail(int code, const char *msg)
{
    set_last_error(msg);
    return code;
}

extern "C" int fk_imm_batch_f64(const fk_imm_desc *d, const double *F, const double *Q, const double *H,
                                const double *R, const double *M, const double *z, double *xs, double *Ps,
                                double *mu, double *x_out, double *P_out, double *mu_out, double *x_prior_out,
                                double *P_prior_out, double *likelihood_out, int32_t *status, void *stream)
{
    return fk_imm_batch_ex_f64(d, F, Q, H, R, M, z, nullptr, nullptr, 0, nullptr, nullptr, xs, Ps, mu, x_out, P_out, mu_out, x_prior_out,
                                   P_prior_out, likelihood_out, status, stream);
}

extern "C" int fk_imm_batch_ex_f64(const fk_imm_desc *d, const double *F, const double *Q, const double *H,
                                       const double *R, const double *M, const double *z, const uint8_t *zmask,
                                       double *ll0, int32_t nu, const double *B, const double *u, double *xs, double *Ps,
                                       double *mu, double *x_out, double *P_out,
                                       double *mu_out, double *x_prior_out, double *P_prior_out, double *likelihood_out,
                                       int32_t *status, void *stream)
{
    if (!d) return fail(FK_ERR_BAD_ARG, "desc is NULL");
    if (d->n < 1 || d->n > 16 || d->m < 1 || d->m > 8 || d->n_models < 2 || d->n_models > 16)
        return fail(FK_ERR_UNSUPPORTED, "IMM: dim_x 1..16, dim_z 1..8, 2..16 models");
    if (d->layout != FK_LAYOUT_AOS && d->layout != FK_LAYOUT_SOA) return fail(FK_ERR_BAD_ARG, "IMM: bad layout");
    if (d->phase < FK_IMM_STEP || d->phase > FK_IMM_UPDATE) return fail(FK_ERR_BAD_ARG, "IMM: bad phase");
    const bool needs_z = (d->phase == FK_IMM_STEP && d->T > 0) || d->phase == FK_IMM_UPDATE;
    if (d->N < 0 || d->T < 0 || !F || !Q || !H || !R || (!M && !(d->flags & FK_IMM_FLAG_MMAE)) || !xs || !Ps || !mu || (needs_z && !z))
        return fail(FK_ERR_BAD_ARG, "IMM: bad argument");
    if ((double)d->N * d->n_models * d->n * d->n * 8.0 >= 4294967296.0)
        return fail(FK_ERR_UNSUPPORTED, "IMM: record block >= 4 GiB, split the batch");
    if (d->N == 0) return FK_OK;
    ImmArgs a{};
    a.F = F; a.Q = Q; a.H = H; a.R = R; a.Mt = M;
    a.z = z ? z : xs;   // predict-only: the measurement is read but never used
    a.xs = xs; a.Ps = Ps; a.mu = mu;
    a.x_out = x_out; a.P_out = P_out; a.mu_out = mu_out; a.xp_out = x_prior_out; a.Pp_out = P_prior_out;
    a.L_out = likelihood_out; a.status = status; a.N = d->N; a.T = d->phase == FK_IMM_STEP ? d->T : 1;
    a.n = d->n; a.m = d->m; a.phase = d->phase; a.mmae = (d->flags & FK_IMM_FLAG_MMAE) ? 1 : 0;
    a.mask = zmask; a.ll0 = ll0;
    if (nu < 0 || nu > 4 || (nu > 0 && (!B || !u))) return fail(FK_ERR_BAD_ARG, "IMM: control input needs 1 <= dim_u <= 4, B and u");
    a.nu = nu; a.B = B; a.u = u;
    if (a.mmae && (x_prior_out || P_prior_out)) return fail(FK_ERR_BAD_ARG, "MMAE: prior outputs are not defined");
    // which compiled output set (if any) the given pointers form; the single-phase calls use the
    // run-time-tested kernel (one step, launch-bound anyway)
    const int post = (x_out && P_out && mu_out) ? 1 : ((x_out || P_out || mu_out) ? -1 : 0);
    const int prior = (x_prior_out && P_prior_out) ? 2 : ((x_prior_out || P_prior_out) ? -1 : 0);
    int mask = (post < 0 || prior < 0) ? -1 : (post | prior | (likelihood_out ? 4 : 0));
    if (mask != 0 && mask != 1 && mask != 7) mask = -1;
    if (d->phase != FK_IMM_STEP || a.mmae) mask = -1;   // the general kernel also carries the MMAE arithmetic
    if (zmask || ll0 || nu > 0) mask = -1;               // ... the missing-measurement bookkeeping and the control input
    a.i0 = 0; a.cnt = d->N; a.status_or = 0;
    const int layout = d->layout, n_models = d->n_models;
    // register-resident instantiations for the small banks (2 / 3 filters, dim_x <= 6, dim_z <= 3: imm_kernels.hip, fk_dims_imm.def)
    const bool small = d->n <= 6 && d->m <= 3 && n_models <= 3;
    const int cls = (d->n <= 2 && d->m <= 1) ? 0 : (d->n <= 4 && d->m <= 2) ? 1 : 2;
    // Every other bank, 2..16 filters, IMM or MMAE, missing measurements, control input and the single-phase calls included: one lane per
    // FILTER up to dim_x 9 / dim_z 4 (imm_lanes.hip: classes (4,2), (6,3), (9,4)), FOUR lanes per filter above (imm_quad.hip: classes
    // (12,4), (16,8)).  FK_IMM_LANES=2: the small banks on imm_lanes.hip too (the A/B of tests/test_gpu_imm.py).
    static const int lanes_mode = [] { const char *v = getenv("FK_IMM_LANES"); return v ? atoi(v) : 1; }();
    const bool quad = !(d->n <= 9 && d->m <= 4);
    const bool lanes = (lanes_mode > 1 || !small) && !quad;
    auto one = [&](const ImmArgs &b, hipStream_t s) -> int {
        if (lanes) {
            // the smallest class that holds the filters: (4, 2), (6, 3), (9, 4)
            const int rc = (b.n <= 4 && b.m <= 2) ? launch_imm_lanes_4_2(b, n_models, layout, s)
                         : (b.n <= 6 && b.m <= 3) ? launch_imm_lanes_6_3(b, n_models, layout, s) : launch_imm_lanes_9_4(b, n_models, layout, s);
            if (rc == 0) return check_launch("imm_lanes_kernel");
        }
        if (quad) {
            // (the class (16, 8) with two filters: eight lanes per filter; FK_IMM_OCT=0: four, the A/B)
            static const int oct_mode = [] { const char *v = getenv("FK_IMM_OCT"); return v ? atoi(v) : 1; }();
            const int rc = (b.n <= 12 && b.m <= 4) ? launch_imm_quad_12_4(b, n_models, layout, s)
                         : (oct_mode > 0 && n_models <= 2) ? launch_imm_oct_16_8(b, n_models, layout, s) : launch_imm_quad_16_8(b, n_models, layout, s);
            if (rc == 0) return check_launch("imm_quad_kernel");
        }
        if (small) {
            if (n_models == 2) {
                if (cls == 0) launch_imm_2_1_2(b, layout, mask, s);
                else if (cls == 1) launch_imm_4_2_2(b, layout, mask, s);
                else launch_imm_6_3_2(b, layout, mask, s);
            } else {
                if (cls == 0) launch_imm_2_1_3(b, layout, mask, s);
                else if (cls == 1) launch_imm_4_2_3(b, layout, mask, s);
                else launch_imm_6_3_3(b, layout, mask, s);
            }
        } else {
            return fail(FK_ERR_UNSUPPORTED, "IMM: no kernel holds this bank");      // (not reached: the three families cover dim_x <= 16, dim_z <= 8, 2..16 filters)
        }
        return check_launch("imm_kernel");
    };
    hipStream_t s = (hipStream_t)stream;
    if (d->phase != FK_IMM_STEP || a.T < 2) return one(a, s);
    // tail filling (fk_chunks.hpp, imm_chunked_call): wave slots of the instantiation the call runs on -- the compiled output
    // sets of the small banks at their FK_IMM_WAVES per SIMD (fk_dims_imm.def), everything else at one
    static const int small_waves[3][2] = {{4, 3}, {2, 2}, {1, 1}};            // [class][n_models - 2]
    // (the lanes kernel: a wave holds 64 / G banks, G = the bank size rounded up to a power of two)
    const int lanes_g = n_models <= 2 ? 2 : n_models <= 4 ? 4 : n_models <= 8 ? 8 : 16;
    const long slots = quad ? 256L / lanes_g : lanes ? 1024L / lanes_g : 1024L * ((small && mask >= 0) ? small_waves[cls][n_models - 2] : 1);
    return imm_chunked_call(a, d->n, d->m, n_models, slots, one, s);
}
