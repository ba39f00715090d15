// hostcheck_rs.cpp -- TEST ONLY: the resampler's position / boundary arithmetic (filterpy_amd/csrc/fk_resample_math.hpp,
// the same __host__ __device__ source the kernels use) compiled for the host with -ffp-contract=off, so that it can
// be held against brute-force counts with real IEEE divisions (numpy) in the GPU-less container.
#include <stdint.h>

#include <vector>

#include "../../filterpy_amd/csrc/fk_resample_math.hpp"
#include "../../filterpy_amd/csrc/fk_resample_whole.hpp"

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


// resample_whole_kernel's arithmetic, thread by thread in the kernel's phase order (NT threads x 8 consecutive weights,
// the per-thread pieces are the kernel's own functions, fk_resample_whole.hpp; the chain runs in its serial form and the
// emission is a plain fill): cumulative sums, slot boundaries, indices.  info[0] = dirty elements, info[1] = 1 when the
// round declined (the kernel then runs the reference's loop literally), info[2] = last boundary (slots filled),
// info[3] = uniform threads, info[4] = 1 when the plain prefix sums alone answered.
template <bool STRATIFIED>
static int whole_resample(int NT, int Np, const double *wts, const double *u, double *cs_out, int *idx_out, int *info, int mode)
{
    using namespace fk;
    if (Np > NT * WH_ITEMS) return -1;
    std::vector<double> w((size_t)NT * WH_ITEMS, 0.0), before(NT);
    for (int j = 0; j < Np; ++j) w[j] = wts[j];
    // plain prefix sums in the kernel's shape: serial inside a thread, then over the 64 lanes of a wave, then over waves
    std::vector<double> tsum(NT);
    for (int t = 0; t < NT; ++t) {
        double s = 0.0;
        for (int q = 0; q < WH_ITEMS; ++q) s += w[t * WH_ITEMS + q];
        tsum[t] = s;
    }
    const int NW = (NT + 63) / 64;
    std::vector<double> wtot(NW, 0.0), lane_excl(NT);
    for (int wv = 0; wv < NW; ++wv) {
        double run = 0.0;
        for (int l = 0; l < 64 && wv * 64 + l < NT; ++l) {
            lane_excl[wv * 64 + l] = run;
            run += tsum[wv * 64 + l];
        }
        wtot[wv] = run;
    }
    for (int t = 0; t < NT; ++t) {
        double b = lane_excl[t];
        for (int wv = 0; wv < t / 64; ++wv) b += wtot[wv];
        before[t] = b;
    }
    WhPos<STRATIFIED> px;
    px.Np = Np;
    px.Nd = (double)Np;
    px.halfNd = 0.5 * px.Nd;
    px.u_sys = STRATIFIED ? 0.0 : u[0];
    px.u_str = STRATIFIED ? u : nullptr;
    std::vector<int> nb((size_t)NT * WH_ITEMS);
    // step 0: boundaries from the plain prefix sums; mode 1 = stop here when no element is inside the band (the kernel's
    // common path: cs_out is not produced), mode 0 = always run the exact round
    info[4] = 0;
    if (mode == 1) {
        unsigned any = 0;
        for (int t = 0; t < NT; ++t) {
            double w8[WH_ITEMS];
            int n8[WH_ITEMS];
            for (int q = 0; q < WH_ITEMS; ++q) w8[q] = w[t * WH_ITEMS + q];
            any |= wh_approx_boundaries<STRATIFIED>(w8, before[t], px, n8);
            for (int q = 0; q < WH_ITEMS; ++q) nb[t * WH_ITEMS + q] = n8[q];
        }
        if (!any) {
            info[0] = info[1] = info[3] = 0;
            info[4] = 1;
            int prev = 0;
            for (int j = 0; j < NT * WH_ITEMS; ++j) {
                for (int i = prev; i < nb[j] && i < Np; ++i) idx_out[i] = j;
                if (nb[j] > prev) prev = nb[j];
            }
            for (int i = prev; i < Np; ++i) idx_out[i] = Np - 1;
            info[2] = prev;
            return 0;
        }
    }
    std::vector<WhThread> th(NT);
    std::vector<int> dbase(NT);
    std::vector<wh_u64> pbase(NT);
    int D = 0;
    wh_u64 ptotal = 0;
    info[3] = 0;
    for (int t = 0; t < NT; ++t) {
        double w8[WH_ITEMS];
        for (int q = 0; q < WH_ITEMS; ++q) w8[q] = w[t * WH_ITEMS + q];
        wh_classify(w8, before[t], t * WH_ITEMS, Np, th[t]);
        dbase[t] = D;
        pbase[t] = ptotal;
        if (th[t].overflow) { info[0] = -1; info[1] = 1; return 0; }
        D += th[t].ndirty;
        ptotal += th[t].psum;
        info[3] += th[t].uniform ? 1 : 0;
    }
    info[0] = D;
    info[1] = 0;
    info[2] = 0;
    if (D > WH_DMAX) { info[1] = 1; return 0; }
    std::vector<int> seg_e(WH_DMAX + 1, WH_NONE);
    std::vector<double> d_w(WH_DMAX), d_cs(WH_DMAX), seg_c(WH_DMAX + 1);
    std::vector<wh_u64> d_ps(WH_DMAX), seg_ps0(WH_DMAX + 1);
    for (int t = 0; t < NT; ++t) {
        double w8[WH_ITEMS];
        for (int q = 0; q < WH_ITEMS; ++q) w8[q] = w[t * WH_ITEMS + q];
        wh_lists(w8, th[t], dbase[t], pbase[t], seg_e.data(), d_w.data(), d_ps.data());
    }
    bool bad = false;
    for (int t = 0; t < NT; ++t) bad = bad || wh_claims_bad(th[t], dbase[t], seg_e.data());
    double carry_out = 0.0;
    const bool ok = wh_chain_serial(D, ptotal, seg_e.data(), d_ps.data(), d_w.data(), seg_c.data(), seg_ps0.data(), d_cs.data(), &carry_out);
    if (bad || !ok) { info[1] = 1; return 0; }
    for (int t = 0; t < NT; ++t) {
        double w8[WH_ITEMS], cs[WH_ITEMS];
        int n8[WH_ITEMS];
        for (int q = 0; q < WH_ITEMS; ++q) w8[q] = w[t * WH_ITEMS + q];
        wh_cumsums(th[t], t * WH_ITEMS, Np, dbase[t], pbase[t], seg_e.data(), seg_c.data(), seg_ps0.data(), d_cs.data(), carry_out, cs);
        wh_boundaries<STRATIFIED>(th[t], dbase[t], pbase[t], seg_e.data(), seg_c.data(), seg_ps0.data(), d_cs.data(), px, n8);
        for (int q = 0; q < WH_ITEMS; ++q) {
            const int j = t * WH_ITEMS + q;
            if (j < Np) cs_out[j] = cs[q];
            nb[j] = n8[q];
        }
    }
    // weight j owns the slots [n_{j-1}, n_j)
    int prev = 0;
    for (int j = 0; j < NT * WH_ITEMS; ++j) {
        for (int i = prev; i < nb[j] && i < Np; ++i) idx_out[i] = j;
        if (nb[j] > prev) prev = nb[j];
    }
    for (int i = prev; i < Np; ++i) idx_out[i] = Np - 1;                 // positions >= cumsum[-1]: IndexError in the reference
    info[2] = prev;
    return 0;
}

// mode 0: the exact round; mode 1: the kernel's order -- plain-prefix boundaries first, the exact round only for a vector with
// an element inside the band (info[4] = 1: answered by the plain prefix sums alone)
extern "C" int hc_whole_resample(int NT, int Np, const double *wts, int stratified, const double *u, double *cs_out, int *idx_out, int *info,
                                 int mode)
{
    return stratified ? whole_resample<true>(NT, Np, wts, u, cs_out, idx_out, info, mode) : whole_resample<false>(NT, Np, wts, u, cs_out, idx_out, info, mode);
}
