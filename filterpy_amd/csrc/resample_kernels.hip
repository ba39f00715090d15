// resample_kernels.hip -- particle-filter resampling kernels for gfx950 (placeholder until
// the exact-scan kernels land; entry points fail loudly).
#include "../../include/filterhip.h"
#include "fk_device.hpp"

extern "C" {

size_t fk_resample_workspace_bytes(int64_t, int64_t) { return 0; }

int fk_resample_systematic_f64(int64_t, int64_t, const double *, const double *, int32_t *, int32_t *,
                               void *, size_t, void *)
{
    fk::set_last_error("fk_resample_systematic_f64: not implemented yet");
    return FK_ERR_UNSUPPORTED;
}
int fk_resample_stratified_f64(int64_t, int64_t, const double *, const double *, int32_t *, int32_t *,
                               void *, size_t, void *)
{
    fk::set_last_error("fk_resample_stratified_f64: not implemented yet");
    return FK_ERR_UNSUPPORTED;
}
int fk_resample_multinomial_f64(int64_t, int64_t, int64_t, const double *, const double *, int64_t *,
                                void *, size_t, void *)
{
    fk::set_last_error("fk_resample_multinomial_f64: not implemented yet");
    return FK_ERR_UNSUPPORTED;
}
}
