/* Plain-C restatement of filterpy.monte_carlo.resampling's merge loops --
 * TEST INFRASTRUCTURE ONLY (checker + timed CPU baseline), never linked into
 * the product library.
 *
 * Follows rlabbe/filterpy v1.4.5 filterpy/monte_carlo/resampling.py:
 *   systematic_resample :139-149, stratified_resample :103-113.
 * numpy.cumsum is restated as a strictly sequential fp64 add chain (that is what
 * NumPy 2.2.6 does for a contiguous float64 vector; pinned bitwise against
 * np.cumsum in tests/test_oracle_resample.py).  positions are
 * fl(fl(u + (double)i) / (double)N) -- one IEEE add, one IEEE divide; build with
 * -ffp-contract=off and no fast-math.
 *
 * Where the reference would raise IndexError (a position >= cumsum[-1],
 * resampling.py:109,145), the index is set to N and the call returns the
 * number of such positions.
 */
#include <stdint.h>
#include <stdlib.h>

static int64_t merge(int64_t N, const double *w, const double *u, int per_particle_u, int32_t *idx)
{
    /* cumulative_sum = np.cumsum(weights) : sequential */
    double *cs = (double *)malloc((size_t)(N > 0 ? N : 1) * sizeof(double));
    double c = 0.0;
    for (int64_t k = 0; k < N; ++k) {
        c = (k == 0) ? w[0] : c + w[k];
        cs[k] = c;
    }
    int64_t i = 0, j = 0, overrun = 0;
    const double dN = (double)N;
    while (i < N) {
        const double ui = per_particle_u ? u[i] : u[0];
        const double pos = (ui + (double)i) / dN;
        if (j >= N) {               /* reference: IndexError */
            idx[i] = (int32_t)N;
            ++overrun;
            ++i;
        } else if (pos < cs[j]) {
            idx[i] = (int32_t)j;
            ++i;
        } else {
            ++j;
        }
    }
    free(cs);
    return overrun;
}

int64_t oracle_systematic(int64_t N, const double *w, const double *u, int32_t *idx)
{
    return merge(N, w, u, 0, idx);
}

int64_t oracle_stratified(int64_t N, const double *w, const double *u, int32_t *idx)
{
    return merge(N, w, u, 1, idx);
}

/* sequential cumsum exposed for pinning against np.cumsum */
void oracle_cumsum(int64_t N, const double *w, double *cs)
{
    double c = 0.0;
    for (int64_t k = 0; k < N; ++k) {
        c = (k == 0) ? w[0] : c + w[k];
        cs[k] = c;
    }
}
