// fk_host.cpp -- host-side plumbing of libfilterhip: error capture and ABI utilities.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/filterhip.h"
#include "fk_device.hpp"
#include "fk_chunks.hpp"

namespace fk {

static thread_local char g_err[512] = "";

void set_last_error(const char *msg)
{
    strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}

int check_launch(const char *what)
{
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return FK_OK;
    char buf[512];
    snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
    set_last_error(buf);
    return FK_ERR_LAUNCH;
}

// Where the several-lane fused UKF (ukf_mlg.hip) takes over from the one-lane kernels (ukf_kernels.hip: FK_UKF_MLG=0 turns it off,
// FK_UKF_MLG_MIN_NX / FK_UKF_MLG_RTS_MIN_NX = 7..10 move the filter's / the smoother's lower bound).  Read ONCE per process, here:
// every part of ukf_kernels.hip and fk_ukf_linear_supported share this snapshot.
struct UkfMlgRoute { int fwd_min, rts_min; };
UkfMlgRoute ukf_mlg_route()
{
    static const UkfMlgRoute v = [] {
        const char *on = getenv("FK_UKF_MLG");
        if (on && on[0] == '0') return UkfMlgRoute{99, 99};
        auto knob = [](const char *name, int dflt) {
            const char *mn = getenv(name);
            const int m = mn ? atoi(mn) : dflt;
            return m >= 7 && m <= 10 ? m : dflt;
        };
        return UkfMlgRoute{knob("FK_UKF_MLG_MIN_NX", 10), knob("FK_UKF_MLG_RTS_MIN_NX", 7)};
    }();
    return v;
}

}  // namespace fk

extern "C" {

int fk_abi_version(void) { return FK_ABI_VERSION; }
const char *fk_build_arch(void) { return "gfx950"; }
const char *fk_last_error(void) { return fk::g_err; }

// what a chunked call would do (fk_chunks.hpp): host arithmetic only, no GPU work
int fk_chunk_plan(int64_t n_tracks, int64_t n_steps, int32_t tracks_per_wave, int64_t wave_slots, int32_t group,
                  int64_t *windows, int32_t *n_groups, int32_t *n_chunks)
{
    if (n_tracks < 0 || n_steps < 1 || tracks_per_wave < 1 || wave_slots < 1 || !windows) return -1;
    int G = 1, H = 1;
    const bool chunked = fk::ml_chunk_policy((n_tracks + tracks_per_wave - 1) / tracks_per_wave, n_steps, G, H, wave_slots);
    if (!chunked) G = H = 1;
    if (n_groups) *n_groups = G;
    if (n_chunks) *n_chunks = H;
    if (group < 0 || group >= G) return -1;
    int nw = 0;
    for (int h = 0; h <= H; ++h) {
        long w0, w1;
        if (!chunked) {
            if (h > 0) break;
            w0 = 0;
            w1 = n_steps;
        } else if (!fk::chunk_window(n_steps, G, H, group, h, w0, w1)) continue;
        windows[2 * nw] = w0;
        windows[2 * nw + 1] = w1;
        ++nw;
    }
    return nw;
}

}  // extern "C"
