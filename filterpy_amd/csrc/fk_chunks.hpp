// fk_chunks.hpp -- tail filling for the several-lanes-per-track kernels (host side).
//
// Their step is bound by arithmetic and latency, every wave runs the same T steps, so a bank of W waves on S wave slots
// takes ceil(W / S) rounds: BASELINE config 3 (1e5 tracks = 6250 waves on 2048 slots) pays 4 rounds for 3.05 rounds of
// work.  A chunked call cuts the bank into G track groups (multiples of 64 tracks) and the T steps into H time chunks and
// launches the pieces on G streams -- group g's chunks in order on stream g, the state handed from chunk to chunk through
// memory (kernel boundaries of one stream: no protocol), different groups concurrently: while one group's piece tails
// off, the other groups' pieces fill the slots, and what is left at the very end is the tail of a piece 1/H as long.
// Group g's chunk boundaries are shifted by g / G of a chunk, else all groups would tail off at the same moments.
// Same arithmetic per track: results are bit-identical to the single launch (tests/test_gpu_kf.py).  The helper streams
// fork from and join the caller's stream with events; they and the events are created once per device.
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <mutex>

// default decomposition of a chunked call (launch_kf_ml_chunked): track groups x time chunks
#ifndef FK_ML_CHUNK_G
#define FK_ML_CHUNK_G 3
#endif
#ifndef FK_ML_CHUNK_H
#define FK_ML_CHUNK_H 4
#endif
#ifndef FK_ML_CHUNK_STAGGER
#define FK_ML_CHUNK_STAGGER 1
#endif

namespace fk {

struct MlStreams {
    static constexpr int MAXG = 4;
    hipStream_t st[MAXG] = {};
    hipEvent_t fork = nullptr, done[MAXG] = {};
    bool ok = false;
    std::mutex mu;      // one chunked call enqueues at a time: a stream wait binds to the event's LATEST record
    void create()       // on the device that is current: streams and events belong to a device
    {
        ok = hipEventCreateWithFlags(&fork, hipEventDisableTiming) == hipSuccess;
        for (int g = 1; g < MAXG && ok; ++g)
            ok = hipStreamCreateWithFlags(&st[g], hipStreamNonBlocking) == hipSuccess &&
                 hipEventCreateWithFlags(&done[g], hipEventDisableTiming) == hipSuccess;
    }
};

// one set of helper streams per device, created on first use with that device current (a process that drives several
// GPUs must not launch one device's pieces on another device's streams: ADVICE r2); nullptr: no chunking
inline MlStreams *ml_streams()
{
    static constexpr int MAXDEV = 16;
    static MlStreams sets[MAXDEV];
    static bool made[MAXDEV] = {};
    static std::mutex mk;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return nullptr;
    std::lock_guard<std::mutex> lock(mk);
    if (!made[dev]) {
        sets[dev].create();
        made[dev] = true;
    }
    return sets[dev].ok ? &sets[dev] : nullptr;
}

// every piece that was forked onto helper stream g is joined back into the caller's stream -- also when a later piece
// failed to launch: the caller may free or reuse the buffers as soon as its own stream is done
inline int ml_join(MlStreams &ms, const bool (&forked)[MlStreams::MAXG], hipStream_t s, int rc)
{
    for (int g = 1; g < MlStreams::MAXG; ++g)
        if (forked[g] && (hipEventRecord(ms.done[g], ms.st[g]) != hipSuccess || hipStreamWaitEvent(s, ms.done[g], 0) != hipSuccess)) {
            (void)hipStreamSynchronize(ms.st[g]);          // last resort: the join must not be skipped
            if (rc == 0) rc = -1;
        }
    return rc;
}

// optional output / input arrays: a NULL stays NULL in every piece
template <class Ptr>
inline Ptr ml_off(Ptr p, long d)
{
    return p ? p + d : nullptr;
}

// G x H decomposition of a call over `waves` waves and T steps ("G,H" from FK_ML_CHUNKS, else the default where the
// last round would be less than 40 % full); false: one launch
inline bool ml_chunk_policy(long waves, long T, int &G, int &H, long slots = 2048)
{
    G = H = 1;
    if (const char *cv = getenv("FK_ML_CHUNKS")) {
        if (sscanf(cv, "%d,%d", &G, &H) != 2) G = H = 1;
    } else if (waves > 2 * slots && T >= 16) {
        const long rem = waves % slots;
        if (rem != 0 && rem * 10 < slots * 4) { G = FK_ML_CHUNK_G; H = FK_ML_CHUNK_H; }
    }
    if (G > MlStreams::MAXG) G = MlStreams::MAXG;
    if (H > 64) H = 64;
    if (H > T) H = (int)T;
    return G >= 1 && H >= 1 && !(G == 1 && H == 1);
}


// Window h (0..H; there is one more window than chunks because of the stagger) of track group g over L steps: [w0, w1).
// Group g's boundaries are shifted down by g / G of a chunk; false: empty window.  The H + 1 windows of a group tile
// [0, L) in order (tests/test_host_logic.py checks this through fk_chunk_plan for every L <= 128, G <= 4, H <= L).
inline bool chunk_window(long L, int G, int H, int g, int h, long &w0, long &w1)
{
    const long shift = (FK_ML_CHUNK_STAGGER && !getenv("FK_ML_NO_STAGGER")) ? (L * g) / ((long)H * G) : 0;
    w0 = L * h / H - shift;
    w1 = L * (h + 1) / H - shift;
    if (w0 < 0) w0 = 0;
    if (h == H) w1 = L;
    if (w1 > L) w1 = L;
    return w1 > w0;
}

// Forward filter: `one(args, stream)` launches one piece (tracks [i0, i0 + cnt), T steps from the pointers in args); the
// state is handed from chunk to chunk through x / P in place.  (KfArgs as a template parameter only keeps this header
// free of the kernel headers.)  FK_ML_CHUNKS="G,H" forces a decomposition; default: ml_chunk_policy.
template <class Args, class One>
int kf_chunked_call(const Args &a, int n, int m, long slots, One &&one, hipStream_t s, int tracks_per_wave = 16, long group_quantum = 64)
{
    int G, H;
    if (!ml_chunk_policy((a.cnt + tracks_per_wave - 1) / tracks_per_wave, a.T, G, H, slots) || a.cnt < group_quantum * G) return one(a, s);
    MlStreams *msp = ml_streams();
    if (!msp) return one(a, s);
    MlStreams &ms = *msp;
    std::lock_guard<std::mutex> lock(ms.mu);
    if (hipEventRecord(ms.fork, s) != hipSuccess) return one(a, s);
    // track groups: multiples of a workgroup's tracks, the last one takes the remainder
    const long blocks = (a.cnt + group_quantum - 1) / group_quantum, per = (blocks + G - 1) / G * group_quantum, nn = (long)n * n;
    int rc = 0;
    bool forked[MlStreams::MAXG] = {};
    for (int g = 0; g < G && rc == 0; ++g) {
        const long g0 = a.i0 + (long)g * per;
        const long gcnt = (g0 + per <= a.i0 + a.cnt) ? per : (a.i0 + a.cnt - g0);
        if (gcnt <= 0) break;
        hipStream_t sg = g == 0 ? s : ms.st[g];
        if (g > 0) {
            if (hipStreamWaitEvent(sg, ms.fork, 0) != hipSuccess) { rc = -1; break; }
            forked[g] = true;
        }
        for (int h = 0; h <= H && rc == 0; ++h) {
            long t0, t1;
            if (!chunk_window(a.T, G, H, g, h, t0, t1)) continue;
            Args b = a;
            b.i0 = g0;
            b.cnt = gcnt;
            b.T = t1 - t0;
            b.status_or = t0 > 0 ? 1 : a.status_or;
            b.z = a.z + t0 * a.N * m;
            b.mask = ml_off(a.mask, t0 * a.N);
            b.means = ml_off(a.means, t0 * a.N * n);
            b.means_p = ml_off(a.means_p, t0 * a.N * n);
            b.covs = ml_off(a.covs, t0 * a.cov_step);              // (cov_step = N n^2, or 2 N n^2: FK_KF_FLAG_COV_INTERLEAVED)
            b.covs_p = ml_off(a.covs_p, t0 * a.cov_step);
            if (a.model_t) {                                   // one model per step, shared by the bank (VAR instantiations)
                b.F = a.F + t0 * nn;
                b.Q = a.Q + t0 * nn;
                b.H = a.H + t0 * (long)m * n;
                b.R = a.R + t0 * (long)m * m;
                if (a.nu > 0) b.B = ml_off(a.B, t0 * (long)n * a.nu);
            }
            if (a.nu > 0) b.u = ml_off(a.u, t0 * a.N * a.nu);
            if (a.extras_per_step) {                           // the by-product histories advance with the time window too
                b.y_out = ml_off(a.y_out, t0 * a.N * m);
                b.K_out = ml_off(a.K_out, t0 * a.N * (long)n * m);
                b.S_out = ml_off(a.S_out, t0 * a.N * (long)m * m);
                b.SI_out = ml_off(a.SI_out, t0 * a.N * (long)m * m);
                b.ll_out = ml_off(a.ll_out, t0 * a.N);
                b.maha_out = ml_off(a.maha_out, t0 * a.N);
            }
            rc = one(b, sg);
        }
    }
    return ml_join(ms, forked, s, rc);
}

// The smoother runs backwards: group g's chunks go from the last time window to the first on stream g; a chunk's window
// [k0, k1] shares its top step k1 with the chunk before it (which smoothed it): RtsArgs::cont.  `one(args, stream)`
// launches one piece.  (RtsArgs is a template parameter only to keep this header free of the kernel headers.)
template <class Args, class One>
int rts_chunked_call(const Args &a, int n, long slots, One &&one, hipStream_t s)
{
    int G, H;
    const long steps = a.T - 1;                                   // backward steps T-2 .. 0
    // (the call may itself be a track window of a larger bank: cnt != 0 -- kf_dispatch.cpp; N stays the array stride)
    const long w0 = a.cnt ? a.i0 : 0, wn = a.cnt ? a.cnt : a.N;
    if (!ml_chunk_policy((wn + 15) / 16, steps, G, H, slots) || wn < 64L * G) return one(a, s);
    MlStreams *msp = ml_streams();
    if (!msp) return one(a, s);
    MlStreams &ms = *msp;
    std::lock_guard<std::mutex> lock(ms.mu);
    if (hipEventRecord(ms.fork, s) != hipSuccess) return one(a, s);
    const long blocks = (wn + 63) / 64, per = (blocks + G - 1) / G * 64, nn = (long)n * n;
    int rc = 0;
    bool forked[MlStreams::MAXG] = {};
    for (int g = 0; g < G && rc == 0; ++g) {
        const long g0 = w0 + (long)g * per, gcnt = (g0 + per <= w0 + wn) ? per : (w0 + wn - g0);
        if (gcnt <= 0) break;
        hipStream_t sg = g == 0 ? s : ms.st[g];
        if (g > 0) {
            if (hipStreamWaitEvent(sg, ms.fork, 0) != hipSuccess) { rc = -1; break; }
            forked[g] = true;
        }
        bool first = true;
        for (int h = H; h >= 0 && rc == 0; --h) {                 // windows of backward steps [k0, k1), last first
            long k0, k1;
            if (!chunk_window(steps, G, H, g, h, k0, k1)) continue;
            Args b = a;
            b.i0 = g0;
            b.cnt = gcnt;
            b.T = k1 - k0 + 1;                                     // steps k0 .. k1 of the arrays; k1 is the window's "T-1"
            b.cont = first ? 0 : 1;
            b.status_or = first ? a.status_or : 1;
            b.Xs = a.Xs + k0 * a.N * n;
            b.Ps = a.Ps + k0 * a.N * nn;
            b.xs = a.xs + k0 * a.N * n;
            b.Ps_out = a.Ps_out + k0 * a.N * nn;
            b.K = ml_off(a.K, k0 * a.N * nn);
            b.Pp = ml_off(a.Pp, k0 * a.N * nn);
            rc = one(b, sg);
            first = false;
        }
    }
    return ml_join(ms, forked, s, rc);
}

// The IMM / MMAE banks (imm_kernels.hip, ImmArgs; whole steps only): forward time chunks like kf_chunked_call, the state handed
// from chunk to chunk through xs / Ps / mu (and ll0) in place.  Default policy: where the waves of the call are one to four
// rounds and the last round is less than 60 % full (the one-wave-per-SIMD classes lose up to a quarter to it: 2e5 banks of
// (6,3) x 2 took 4.25 ms where 196 608 -- three full rounds -- took 3.28); FK_IMM_CHUNKS="G,H" forces a decomposition ("1,1":
// one launch).
template <class Args, class One>
int imm_chunked_call(const Args &a, int n, int m, int nm, long slots, One &&one, hipStream_t s)
{
    int G = 1, H = 1;
    const long waves = (a.cnt + 63) / 64;
    if (const char *cv = getenv("FK_IMM_CHUNKS")) {
        if (sscanf(cv, "%d,%d", &G, &H) != 2) G = H = 1;
    } else if (waves > slots && waves <= 4 * slots && a.T >= 16) {
        const long rem = waves % slots;
        if (rem != 0 && rem * 10 < slots * 6) { G = FK_ML_CHUNK_G; H = FK_ML_CHUNK_H; }
    }
    if (G > MlStreams::MAXG) G = MlStreams::MAXG;
    if (H > 64) H = 64;
    if (H > a.T) H = (int)a.T;
    if (G < 1 || H < 1 || (G == 1 && H == 1) || a.cnt < 256L * G) return one(a, s);
    // A masked call carries, per filter, the log-density of a zero residual under the LAST S (what update(None) leaves,
    // kalman_filter.py:515-520 + IMM.py:176-177) from step to step: in registers inside one launch, through ll0 between
    // launches.  Without ll0 a later time chunk would restart them at -inf: such a call is one launch.
    if (a.mask && !a.ll0) return one(a, s);
    MlStreams *msp = ml_streams();
    if (!msp) return one(a, s);
    MlStreams &ms = *msp;
    std::lock_guard<std::mutex> lock(ms.mu);
    if (hipEventRecord(ms.fork, s) != hipSuccess) return one(a, s);
    const long blocks = (a.cnt + 255) / 256, per = (blocks + G - 1) / G * 256, nn = (long)n * n;
    int rc = 0;
    bool forked[MlStreams::MAXG] = {};
    for (int g = 0; g < G && rc == 0; ++g) {
        const long g0 = a.i0 + (long)g * per;
        const long gcnt = (g0 + per <= a.i0 + a.cnt) ? per : (a.i0 + a.cnt - g0);
        if (gcnt <= 0) break;
        hipStream_t sg = g == 0 ? s : ms.st[g];
        if (g > 0) {
            if (hipStreamWaitEvent(sg, ms.fork, 0) != hipSuccess) { rc = -1; break; }
            forked[g] = true;
        }
        for (int h = 0; h <= H && rc == 0; ++h) {
            long t0, t1;
            if (!chunk_window(a.T, G, H, g, h, t0, t1)) continue;
            Args b = a;
            b.i0 = g0;
            b.cnt = gcnt;
            b.T = t1 - t0;
            b.status_or = t0 > 0 ? 1 : a.status_or;
            b.z = a.z + t0 * a.N * m;
            b.mask = ml_off(a.mask, t0 * a.N);
            if (a.nu > 0) b.u = ml_off(a.u, t0 * a.N * a.nu);
            b.x_out = ml_off(a.x_out, t0 * a.N * n);
            b.P_out = ml_off(a.P_out, t0 * a.N * nn);
            b.mu_out = ml_off(a.mu_out, t0 * a.N * nm);
            b.xp_out = ml_off(a.xp_out, t0 * a.N * n);
            b.Pp_out = ml_off(a.Pp_out, t0 * a.N * nn);
            b.L_out = ml_off(a.L_out, t0 * a.N * nm);
            rc = one(b, sg);
        }
    }
    return ml_join(ms, forked, s, rc);
}

// The fused linear UKF smoother (ukf_kernels.hip, UkfRtsArgs): backward windows like rts_chunked_call.  Default policy: where
// the waves of the call are more than one round, at most three, and the last round is less than 60 % full (one wave per SIMD makes that
// BASELINE configs[3]: 1563 waves on 1024 slots); FK_UKF_RTS_CHUNKS="G,H" forces a decomposition ("1,1": one launch).
template <class Args, class One>
int ukf_rts_chunked_call(const Args &a, int n, long slots, One &&one, hipStream_t s)
{
    int G = 1, H = 1;
    const long steps = a.T - 1, waves = (a.cnt + 63) / 64;        // backward steps T-2 .. 0
    if (const char *cv = getenv("FK_UKF_RTS_CHUNKS")) {
        if (sscanf(cv, "%d,%d", &G, &H) != 2) G = H = 1;
    } else if (waves > slots && waves <= 3 * slots && steps >= 16) {
        // (only for a handful of rounds: at 15 rounds the partial last one is 2 % of the call and the pieces' overheads are
        //  not -- 1e6 tracks x 20 steps measured 3.06 ms in one launch, 3.63 ms cut up)
        const long rem = waves % slots;
        if (rem != 0 && rem * 10 < slots * 6) { G = FK_ML_CHUNK_G; H = FK_ML_CHUNK_H; }
    }
    if (G > MlStreams::MAXG) G = MlStreams::MAXG;
    if (H > 64) H = 64;
    if (H > steps) H = (int)steps;
    if (G < 1 || H < 1 || (G == 1 && H == 1) || a.cnt < 256L * G) return one(a, s);
    MlStreams *msp = ml_streams();
    if (!msp) return one(a, s);
    MlStreams &ms = *msp;
    std::lock_guard<std::mutex> lock(ms.mu);
    if (hipEventRecord(ms.fork, s) != hipSuccess) return one(a, s);
    const long blocks = (a.cnt + 255) / 256, per = (blocks + G - 1) / G * 256, nn = (long)n * n;
    int rc = 0;
    bool forked[MlStreams::MAXG] = {};
    for (int g = 0; g < G && rc == 0; ++g) {
        const long g0 = a.i0 + (long)g * per;
        const long gcnt = (g0 + per <= a.i0 + a.cnt) ? per : (a.i0 + a.cnt - g0);
        if (gcnt <= 0) break;
        hipStream_t sg = g == 0 ? s : ms.st[g];
        if (g > 0) {
            if (hipStreamWaitEvent(sg, ms.fork, 0) != hipSuccess) { rc = -1; break; }
            forked[g] = true;
        }
        bool first = true;
        for (int h = H; h >= 0 && rc == 0; --h) {                 // windows of backward steps [k0, k1), last first
            long k0, k1;
            if (!chunk_window(steps, G, H, g, h, k0, k1)) continue;
            Args b = a;
            b.i0 = g0;
            b.cnt = gcnt;
            b.T = k1 - k0 + 1;                                     // steps k0 .. k1 of the arrays; k1 is the window's "T-1"
            b.cont = first ? a.cont : 1;
            b.status_or = first ? a.status_or : 1;
            b.Xs = a.Xs + k0 * a.N * n;
            b.Ps = a.Ps + k0 * a.N * nn;
            b.xs = a.xs + k0 * a.N * n;
            b.ps = a.ps + k0 * a.N * nn;
            b.Ks = ml_off(a.Ks, k0 * a.N * nn);
            rc = one(b, sg);
            first = false;
        }
    }
    return ml_join(ms, forked, s, rc);
}

// The fused linear UKF (ukf_kernels.hip, UkfArgs): only on request -- FK_UKF_CHUNKS="G,H" --, same hand-over through x / P.
template <class Args, class One>
int ukf_chunked_call(const Args &a, int n, int m, One &&one, hipStream_t s)
{
    int G = 1, H = 1;
    const char *cv = getenv("FK_UKF_CHUNKS");
    if (!cv || sscanf(cv, "%d,%d", &G, &H) != 2) return one(a, s);
    if (G > MlStreams::MAXG) G = MlStreams::MAXG;
    if (H > 64) H = 64;
    if (H > a.T) H = (int)a.T;
    if (G < 1 || H < 1 || (G == 1 && H == 1) || a.cnt < 256L * G) return one(a, s);
    MlStreams *msp = ml_streams();
    if (!msp) return one(a, s);
    MlStreams &ms = *msp;
    std::lock_guard<std::mutex> lock(ms.mu);
    if (hipEventRecord(ms.fork, s) != hipSuccess) return one(a, s);
    const long blocks = (a.cnt + 255) / 256, per = (blocks + G - 1) / G * 256, nn = (long)n * n;
    int rc = 0;
    bool forked[MlStreams::MAXG] = {};
    for (int g = 0; g < G && rc == 0; ++g) {
        const long g0 = a.i0 + (long)g * per;
        const long gcnt = (g0 + per <= a.i0 + a.cnt) ? per : (a.i0 + a.cnt - g0);
        if (gcnt <= 0) break;
        hipStream_t sg = g == 0 ? s : ms.st[g];
        if (g > 0) {
            if (hipStreamWaitEvent(sg, ms.fork, 0) != hipSuccess) { rc = -1; break; }
            forked[g] = true;
        }
        for (int h = 0; h <= H && rc == 0; ++h) {
            long t0, t1;
            if (!chunk_window(a.T, G, H, g, h, t0, t1)) continue;
            Args b = a;
            b.i0 = g0;
            b.cnt = gcnt;
            b.T = t1 - t0;
            b.status_or = t0 > 0 ? 1 : a.status_or;
            b.z = a.z + t0 * a.N * m;
            b.mask = ml_off(a.mask, t0 * a.N);
            b.means = ml_off(a.means, t0 * a.N * n);
            b.covs = ml_off(a.covs, t0 * a.N * nn);
            rc = one(b, sg);
        }
    }
    return ml_join(ms, forked, s, rc);
}

}  // namespace fk
