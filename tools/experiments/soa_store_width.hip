// Does the width of a wave's contiguous store matter for element-major histories?  (docs/KERNEL_NOTES.md: element-major (4,x)
// banks above ~5e5 tracks run 0.54-0.59 of HBM where NumPy order -- 1 KiB per store instruction -- runs 0.70-0.74.)
// A wave writes E element planes of a [T][E][N] array, step after step, like kf_fast's element-major outputs:
//   W = 8 : 64 tracks per wave, 8 bytes per lane   -> 512 contiguous bytes per store instruction and plane
//   W = 16: 128 tracks per wave, 16 bytes per lane -> 1 KiB per store instruction and plane (half the waves)
// and reads nothing.  Build: hipcc --offload-arch=gfx950 -O3 soa_store_width.hip -o soa_store_width ; run: ./soa_store_width [N] [T] [E]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int W>
__global__ void __launch_bounds__(256) planes_kernel(double *out, long N, int T, int E, int waves_per_simd_pad)
{
    const long lane_tracks = W / 8;
    const long trk = ((long)blockIdx.x * 256 + threadIdx.x) * lane_tracks;
    if (trk >= N) return;
    double v = (double)trk;
    for (int t = 0; t < T; ++t) {
        double *base = out + (long)t * E * N + trk;
        for (int e = 0; e < E; ++e) {
            v = v * 1.0000001 + 1.0;
            if (W == 8) __builtin_nontemporal_store(v, base + (long)e * N);
            else {
                typedef double d2 __attribute__((ext_vector_type(2)));
                d2 w = {v, v + 0.5};
                __builtin_nontemporal_store(w, reinterpret_cast<d2 *>(base + (long)e * N));
            }
        }
    }
}

template <int W, bool NT>
__global__ void __launch_bounds__(256) planes_kernel2(double *out, long N, int T, int E)
{
    const long lane_tracks = W / 8;
    const long trk = ((long)blockIdx.x * 256 + threadIdx.x) * lane_tracks;
    if (trk >= N) return;
    double v = (double)trk;
    for (int t = 0; t < T; ++t) {
        double *base = out + (long)t * E * N + trk;
        for (int e = 0; e < E; ++e) {
            v = v * 1.0000001 + 1.0;
            if (W == 8) base[(long)e * N] = v;
            else {
                typedef double d2 __attribute__((ext_vector_type(2)));
                d2 w = {v, v + 0.5};
                *reinterpret_cast<d2 *>(base + (long)e * N) = w;
            }
        }
    }
}

int main(int argc, char **argv)
{
    const long N = argc > 1 ? atol(argv[1]) : 1000000;
    const int T = argc > 2 ? atoi(argv[2]) : 20, E = argc > 3 ? atoi(argv[3]) : 40;
    double *out;
    const size_t bytes = (size_t)T * E * N * 8;
    if (hipMalloc(&out, bytes) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep)
        for (int mode = 0; mode < 4; ++mode) {
            const int W = (mode & 1) ? 16 : 8;
            const long lanes = (N + W / 8 - 1) / (W / 8);
            const dim3 grid((unsigned)((lanes + 255) / 256)), block(256);
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL((planes_kernel2<8, false>), grid, block, 0, 0, out, N, T, E);
            else if (mode == 1) hipLaunchKernelGGL((planes_kernel2<16, false>), grid, block, 0, 0, out, N, T, E);
            else if (mode == 2) hipLaunchKernelGGL((planes_kernel<8>), grid, block, 0, 0, out, N, T, E, 0);
            else hipLaunchKernelGGL((planes_kernel<16>), grid, block, 0, 0, out, N, T, E, 0);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            printf("{\"N\": %ld, \"T\": %d, \"planes\": %d, \"bytes_per_lane\": %d, \"nontemporal\": %d, \"ms\": %.3f, \"GBs\": %.1f, \"frac_of_8TBs\": %.3f}\n", N, T, E, W, mode >= 2,
                   ms, bytes / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 8e12);
        }
    return 0;
}
