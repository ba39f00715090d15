// rsq_seed_accuracy.hip -- how accurate are the v_rsq_f64 / v_rcp_f64 seeds of gfx950, and what do the refinement ladders of
// fk_ukf.hpp (sqrt_rsqrt, rcp_refined) reach after each rung?  Maximum relative error over 2^22 arguments spread over eight
// decades, against the correctly rounded host results.
//   hipcc -O3 --offload-arch=gfx950 tools/experiments/rsq_seed_accuracy.hip -o tools/experiments/build/rsq_seed_accuracy
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

__global__ void probe(const double *d, double *out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = d[i];
    const double y = __builtin_amdgcn_rsq(x);
    out[i] = y;                                            // 0: rsq seed
    double g = x * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    out[n + i] = g;                                        // 1: sqrt after one Goldschmidt step
    out[2 * n + i] = h + h;                                // 2: 1/sqrt after one Goldschmidt step
    const double e = fma(-g, g, x);
    g = fma(e, h, g);
    const double r2 = fma(-h, g, 0.5);
    h = fma(h, r2, h);
    out[3 * n + i] = g;                                    // 3: sqrt, with the residual correction (what the kernels use)
    out[4 * n + i] = h + h;                                // 4: 1/sqrt, likewise
    double q = __builtin_amdgcn_rcp(x);
    out[5 * n + i] = q;                                    // 5: rcp seed
    double e1 = fma(-x, q, 1.0);
    q = fma(q, e1, q);
    out[6 * n + i] = q;                                    // 6: one Newton step
    e1 = fma(-x, q, 1.0);
    q = fma(q, e1, q);
    out[7 * n + i] = q;                                    // 7: two (what the kernels use)
}

int main()
{
    const int n = 1 << 22;
    std::vector<double> h(n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const double u = (double)(s >> 11) / 9007199254740992.0;
        h[i] = std::pow(10.0, -4.0 + 8.0 * u);
    }
    double *d, *o;
    hipMalloc(&d, sizeof(double) * n);
    hipMalloc(&o, sizeof(double) * n * 8);
    hipMemcpy(d, h.data(), sizeof(double) * n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(n / 256), dim3(256), 0, 0, d, o, n);
    std::vector<double> r((size_t)n * 8);
    hipMemcpy(r.data(), o, sizeof(double) * n * 8, hipMemcpyDeviceToHost);
    const char *names[8] = {"v_rsq_f64 seed", "sqrt, one Goldschmidt step", "1/sqrt, one Goldschmidt step", "sqrt, + residual correction",
                            "1/sqrt, + second step", "v_rcp_f64 seed", "rcp, one Newton step", "rcp, two Newton steps"};
    for (int k = 0; k < 8; ++k) {
        double worst = 0;
        for (int i = 0; i < n; ++i) {
            const double x = h[i];
            const long double ref = (k == 0 || k == 2 || k == 4) ? 1.0L / sqrtl((long double)x) : (k == 1 || k == 3) ? sqrtl((long double)x) : 1.0L / (long double)x;
            const double rel = (double)fabsl(((long double)r[(size_t)k * n + i] - ref) / ref);
            if (rel > worst) worst = rel;
        }
        printf("{\"quantity\": \"%s\", \"max_rel_err\": %.3e, \"log2\": %.1f}\n", names[k], worst, std::log2(worst));
    }
    return 0;
}
