// hostcheck_quad.cpp -- TEST-ONLY harness: the four-lanes-per-track UKF step (filterpy_amd/csrc/fk_ukf_quad.hpp, the arithmetic
// ukf_mlg.hip runs with DPP quad exchanges) compiled for the HOST, the four lanes of a quad running as four fibers in lockstep:
// an exchange stores the lane's value, hands control round the quad and reads the owner's slot.  Never loaded by filterpy_amd/.
#include <stdint.h>
#include <string.h>
#include <ucontext.h>

#include <algorithm>
#include <vector>

#include "../../filterpy_amd/csrc/fk_ukf_quad.hpp"

namespace {

struct FiberQuad {
    int lanes = 4;                               // lanes per track: 4 (a quad) or 8
    ucontext_t main_ctx, ctx[8];
    std::vector<char> stack[8];
    double slot[2][8];
    long exchanges[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    void (*body)(void *, int) = nullptr;
    void *arg = nullptr;
};
thread_local FiberQuad *g_fq = nullptr;

void fiber_entry(int lane) { g_fq->body(g_fq->arg, lane); }

// lane `lane` of the quad: bcast<O>(v) = the value lane O passed to ITS call of the same exchange
struct HostQuad {
    FiberQuad *fq;
    int lane;
    unsigned calls = 0;
    template <int O>
    double bcast(double v)
    {
        const unsigned p = calls++ & 1u;
        fq->slot[p][lane] = v;
        fq->exchanges[lane]++;
        swapcontext(&fq->ctx[lane], &fq->ctx[(lane + 1) % fq->lanes]);   // round the group; back here once every lane has stored
        return fq->slot[p][O];
    }
};

void run_quad(void (*body)(void *, int), void *arg, int lanes = 4)
{
    FiberQuad fq;
    fq.lanes = lanes;
    fq.body = body;
    fq.arg = arg;
    g_fq = &fq;
    for (int l = 0; l < lanes; ++l) {
        fq.stack[l].resize(1 << 20);
        getcontext(&fq.ctx[l]);
        fq.ctx[l].uc_stack.ss_sp = fq.stack[l].data();
        fq.ctx[l].uc_stack.ss_size = fq.stack[l].size();
        fq.ctx[l].uc_link = l < lanes - 1 ? &fq.ctx[l + 1] : &fq.main_ctx;   // a lane that returns hands over to the next one
        makecontext(&fq.ctx[l], (void (*)())fiber_entry, 1, l);
    }
    swapcontext(&fq.main_ctx, &fq.ctx[0]);
    g_fq = nullptr;
}

template <int NX, int NZ>
struct QuadJob {
    long T;
    const double *F, *H, *Q, *R, *Wp, *zs;
    const unsigned char *mask;
    double scale;
    double *x0, *P0, *means, *covs;
    int st[8];
};

template <int NX, int NZ, int LN = 4>
void quad_lane(void *vp, int lane)
{
    constexpr int R = (NX + LN - 1) / LN;
    auto &job = *static_cast<QuadJob<NX, NZ> *>(vp);
    HostQuad quad{g_fq, lane};
    unsigned g[R];
    double x[NX], P[R][NX];
    for (int r = 0; r < R; ++r) {
        const unsigned row = (unsigned)LN * (unsigned)r + (unsigned)lane;
        g[r] = row < (unsigned)NX ? row : (unsigned)NX - 1u;
        for (int c = 0; c < NX; ++c) P[r][c] = job.P0[g[r] * NX + c];
    }
    for (int i = 0; i < NX; ++i) x[i] = job.x0[i];
    const fk::UkfQuadModel mv{job.F, job.Q, job.H, job.R, job.Wp};
    int st = 0;
    for (long t = 0; t < job.T; ++t) {
        const bool has_z = job.mask ? job.mask[t] != 0 : true;
        double z[NZ];
        for (int c = 0; c < NZ; ++c) z[c] = has_z ? job.zs[t * NZ + c] : 0.0;
        st |= fk::ukf_quad_step_v4<NX, NZ, LN>(x, P, g, z, [&] { return has_z; }, job.scale, mv, quad);
        // every lane writes what it holds: x (replicated) must agree, rows of P go where their slot says
        for (int i = 0; i < NX; ++i) {
            if (lane == 0) job.means[t * NX + i] = x[i];
            else if (memcmp(&job.means[t * NX + i], &x[i], 8) != 0) st |= 1 << 20;        // replicated values differ between lanes
        }
        for (int r = 0; r < R; ++r) {
            const bool dup = (unsigned)LN * (unsigned)r + (unsigned)lane >= (unsigned)NX;
            for (int c = 0; c < NX; ++c) {
                double &dst = job.covs[(t * NX + g[r]) * NX + c];
                if (!dup) dst = P[r][c];
            }
        }
    }
    // duplicates (slots past row n-1) must hold row n-1's values bit for bit: checked against what its owner wrote last
    for (int r = 0; r < R; ++r)
        if ((unsigned)LN * (unsigned)r + (unsigned)lane >= (unsigned)NX && job.T > 0 && ((NX - 1) % LN) < lane) {
            for (int c = 0; c < NX; ++c)
                if (memcmp(&job.covs[((job.T - 1) * NX + (NX - 1)) * NX + c], &P[r][c], 8) != 0) st |= 1 << 21;
        }
    if (lane == 0)
        for (int i = 0; i < NX; ++i) job.x0[i] = x[i];
    for (int r = 0; r < R; ++r)
        if ((unsigned)LN * (unsigned)r + (unsigned)lane < (unsigned)NX)
            for (int c = 0; c < NX; ++c) job.P0[g[r] * NX + c] = P[r][c];
    job.st[lane] = st;
}

template <int NX, int NZ, int LN = 4>
int ukf_quad_batch(long T, const double *F, const double *H, const double *Q, const double *R, const double *Wm,
                   const double *Wc, double scale, const double *zs, const unsigned char *mask, double *x0, double *P0,
                   double *means, double *covs)
{
    constexpr int KS = 2 * NX + 1;
    double wm[KS], wc[KS], wp[2 + NX];
    std::copy(Wm, Wm + KS, wm);
    std::copy(Wc, Wc + KS, wc);
    fk::make_pair_table<NX>(wm, wc, wp);
    if (!fk::pair_weights_symmetric<NX>(wm, wc)) return -2;
    QuadJob<NX, NZ> job{T, F, H, Q, R, wp, zs, mask, scale, x0, P0, means, covs, {0, 0, 0, 0, 0, 0, 0, 0}};
    run_quad(&quad_lane<NX, NZ, LN>, &job, LN);
    for (int l = 1; l < LN; ++l)
        if (job.st[l] != job.st[0]) return 1 << 22;                                              // the status is replicated
    return job.st[0];
}

}  // namespace

extern "C" int hc_ukf_quad_v4(int n, int m, long T, const double *F, const double *H, const double *Q, const double *R,
                              const double *Wm, const double *Wc, double scale, const double *zs,
                              const unsigned char *mask, double *x0, double *P0, double *means, double *covs)
{
#define GO(NXV, NZV) if (n == NXV && m == NZV) return ukf_quad_batch<NXV, NZV>(T, F, H, Q, R, Wm, Wc, scale, zs, mask, x0, P0, means, covs)
#define GOM(NXV) GO(NXV, 1); GO(NXV, 2); GO(NXV, 3); GO(NXV, 4); GO(NXV, 5); GO(NXV, 6); GO(NXV, 7); GO(NXV, 8)
    GO(4, 2); GO(5, 2); GO(7, 1); GO(7, 3); GO(8, 2); GO(8, 4); GO(9, 3); GO(9, 4);
    GOM(10); GOM(11); GOM(12); GOM(13); GOM(14); GOM(15); GOM(16);
#undef GOM
#undef GO
    return -1;
}

// ---- the smoother's backward pass (ukf_quad_rts_step_v4) over one track
namespace {

template <int NX>
struct RtsJob {
    long T;
    const double *F, *Q, *Wp, *Xs, *Ps;
    double scale;
    double *xs, *ps, *Ks;
    int st[8];
};

template <int NX, bool PARKV, int LN = 4>
struct HostRtsIo {
    static constexpr int R = (NX + LN - 1) / LN;
    static constexpr bool PARK = PARKV;
    const RtsJob<NX> &job;
    const unsigned (&g)[R];
    const double (&xn)[NX];
    const double (&Pn)[R][NX];
    long t;
    double lot[R][NX];                                   // the track's parking lot, this lane's rows
    void next_x(double (&out)[NX]) { for (int c = 0; c < NX; ++c) out[c] = xn[c]; }
    void next_row(int r, double (&out)[NX]) { for (int c = 0; c < NX; ++c) out[c] = Pn[r][c]; }
    void own_x(double (&out)[NX]) { for (int c = 0; c < NX; ++c) out[c] = job.Xs[t * NX + c]; }
    void own_row(int r, double (&out)[NX]) { for (int c = 0; c < NX; ++c) out[c] = job.Ps[(t * NX + g[r]) * NX + c]; }
    void park_k(double (&Kv)[R][NX]) { for (int r = 0; r < R; ++r) for (int c = 0; c < NX; ++c) { lot[r][c] = Kv[r][c]; Kv[r][c] = -7e300; } }
    void unpark_k(double (&Kv)[R][NX]) { for (int r = 0; r < R; ++r) for (int c = 0; c < NX; ++c) Kv[r][c] = lot[r][c]; }
    void park_pb(const double (&Pbv)[R][NX]) { for (int r = 0; r < R; ++r) for (int c = 0; c < NX; ++c) lot[r][c] = Pbv[r][c]; }
    void pb_row(int r, double (&out)[NX]) { for (int c = 0; c < NX; ++c) out[c] = lot[r][c]; }
};

template <int NX, bool PARKV, int LN = 4>
void quad_rts_lane(void *vp, int lane)
{
    constexpr int R = (NX + LN - 1) / LN;
    auto &job = *static_cast<RtsJob<NX> *>(vp);
    HostQuad quad{g_fq, lane};
    unsigned g[R];
    bool dup[R];
    for (int r = 0; r < R; ++r) {
        const unsigned row = (unsigned)LN * (unsigned)r + (unsigned)lane;
        dup[r] = row >= (unsigned)NX;
        g[r] = dup[r] ? (unsigned)NX - 1u : row;
    }
    const long T = job.T;
    const fk::UkfQuadModel mv{job.F, job.Q, nullptr, nullptr, job.Wp};
    // the last step is the filter's own output (xs, ps = Xs.copy(), Ps.copy(); K[T-1] = 0)
    double xn[NX], Pn[R][NX];
    for (int i = 0; i < NX; ++i) xn[i] = job.Xs[(T - 1) * NX + i];
    for (int r = 0; r < R; ++r)
        for (int c = 0; c < NX; ++c) Pn[r][c] = job.Ps[((T - 1) * NX + g[r]) * NX + c];
    if (lane == 0)
        for (int i = 0; i < NX; ++i) job.xs[(T - 1) * NX + i] = xn[i];
    for (int r = 0; r < R; ++r)
        if (!dup[r])
            for (int c = 0; c < NX; ++c) {
                job.ps[((T - 1) * NX + g[r]) * NX + c] = Pn[r][c];
                job.Ks[((T - 1) * NX + g[r]) * NX + c] = 0.0;
            }
    int st = 0;
    for (long t = T - 2; t >= 0; --t) {
        double x[NX], P[R][NX], K[R][NX];
        for (int i = 0; i < NX; ++i) x[i] = job.Xs[t * NX + i];
        for (int r = 0; r < R; ++r)
            for (int c = 0; c < NX; ++c) P[r][c] = c <= LN * r + LN - 1 ? job.Ps[(t * NX + g[r]) * NX + c] : -1e300;   // only the lower part is handed over
        HostRtsIo<NX, PARKV, LN> io{job, g, xn, Pn, t, {}};
        st |= fk::ukf_quad_rts_step_v4<NX, LN>(x, P, g, job.scale, mv, quad, K, io);
        for (int i = 0; i < NX; ++i) {
            if (lane == 0) job.xs[t * NX + i] = x[i];
            else if (memcmp(&job.xs[t * NX + i], &x[i], 8) != 0) st |= 1 << 20;
            xn[i] = x[i];
        }
        for (int r = 0; r < R; ++r)
            for (int c = 0; c < NX; ++c) {
                if (!dup[r]) {
                    job.ps[(t * NX + g[r]) * NX + c] = P[r][c];
                    job.Ks[(t * NX + g[r]) * NX + c] = K[r][c];
                } else if (memcmp(&job.ps[(t * NX + g[r]) * NX + c], &P[r][c], 8) != 0 || memcmp(&job.Ks[(t * NX + g[r]) * NX + c], &K[r][c], 8) != 0)
                    st |= 1 << 21;                                   // a duplicate differs from the row it duplicates
                Pn[r][c] = P[r][c];
            }
    }
    job.st[lane] = st;
}

template <int NX, bool PARKV, int LN = 4>
int ukf_quad_rts_batch(long T, const double *F, const double *Q, const double *Wm, const double *Wc, double scale,
                       const double *Xs, const double *Ps, double *xs, double *ps, double *Ks)
{
    constexpr int KS = 2 * NX + 1;
    double wm[KS], wc[KS], wp[2 + NX];
    std::copy(Wm, Wm + KS, wm);
    std::copy(Wc, Wc + KS, wc);
    fk::make_pair_table<NX>(wm, wc, wp);
    if (!fk::pair_weights_symmetric<NX>(wm, wc)) return -2;
    RtsJob<NX> job{T, F, Q, wp, Xs, Ps, scale, xs, ps, Ks, {0, 0, 0, 0, 0, 0, 0, 0}};
    run_quad(&quad_rts_lane<NX, PARKV, LN>, &job, LN);
    for (int l = 1; l < LN; ++l)
        if (job.st[l] != job.st[0]) return 1 << 22;
    return job.st[0];
}

}  // namespace

extern "C" int hc_ukf_quad_rts_v4(int n, long T, const double *F, const double *Q, const double *Wm, const double *Wc,
                                  double scale, const double *Xs, const double *Ps, double *xs, double *ps, double *Ks)
{
#define GO(NXV) if (n == NXV) return ukf_quad_rts_batch<NXV, false>(T, F, Q, Wm, Wc, scale, Xs, Ps, xs, ps, Ks)
    GO(4); GO(5); GO(7); GO(8); GO(9); GO(10); GO(11); GO(12); GO(13); GO(14); GO(15); GO(16);
#undef GO
    return -1;
}

// ... with the parking lot (Io::PARK: Pxb parked through the Pb pass, Pb's rows read back for the correction)
extern "C" int hc_ukf_quad_rts_park_v4(int n, long T, const double *F, const double *Q, const double *Wm, const double *Wc,
                                       double scale, const double *Xs, const double *Ps, double *xs, double *ps, double *Ks)
{
#define GO(NXV) if (n == NXV) return ukf_quad_rts_batch<NXV, true>(T, F, Q, Wm, Wc, scale, Xs, Ps, xs, ps, Ks)
    GO(4); GO(5); GO(7); GO(8); GO(9); GO(10); GO(11); GO(12); GO(13); GO(14); GO(15); GO(16);
#undef GO
    return -1;
}

// ... on EIGHT lanes per track (LN = 8: what the smoother at dim_x >= 13 is headed for; dim_x >= 8)
extern "C" int hc_ukf_oct_rts_v4(int n, long T, const double *F, const double *Q, const double *Wm, const double *Wc,
                                 double scale, const double *Xs, const double *Ps, double *xs, double *ps, double *Ks)
{
#define GO(NXV) if (n == NXV) return ukf_quad_rts_batch<NXV, false, 8>(T, F, Q, Wm, Wc, scale, Xs, Ps, xs, ps, Ks)
    GO(8); GO(9); GO(10); GO(11); GO(12); GO(13); GO(14); GO(15); GO(16);
#undef GO
    return -1;
}

// the filter step on eight lanes per track (where dim_z >= 5 at dim_x >= 13 is headed: one row of H L per lane)
extern "C" int hc_ukf_oct_v4(int n, int m, long T, const double *F, const double *H, const double *Q, const double *R,
                             const double *Wm, const double *Wc, double scale, const double *zs,
                             const unsigned char *mask, double *x0, double *P0, double *means, double *covs)
{
#define GO(NXV, NZV) if (n == NXV && m == NZV) return ukf_quad_batch<NXV, NZV, 8>(T, F, H, Q, R, Wm, Wc, scale, zs, mask, x0, P0, means, covs)
#define GOM(NXV) GO(NXV, 1); GO(NXV, 3); GO(NXV, 4); GO(NXV, 5); GO(NXV, 6); GO(NXV, 7); GO(NXV, 8)
    GO(8, 4); GO(9, 3); GOM(10); GOM(13); GOM(14); GOM(15); GOM(16);
#undef GOM
#undef GO
    return -1;
}
