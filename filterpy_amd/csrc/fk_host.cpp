// fk_host.cpp -- host-side plumbing of libfilterhip: error capture and ABI utilities.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include "../../include/filterhip.h"
#include "fk_device.hpp"

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

}  // namespace fk

extern "C" {

int fk_abi_version(void) { return FK_ABI_VERSION; }
const char *fk_build_arch(void) { return "gfx950"; }
const char *fk_last_error(void) { return fk::g_err; }

}  // extern "C"
