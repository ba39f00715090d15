// valu_latency.hip -- what a dependent fp64 VALU chain costs on gfx950, per instruction, with 1 / 2 / 4 independent chains per
// wave and 1 / 2 waves per SIMD.  The arithmetic-bound kernels of this package (fused UKF, IMM, the multi-lane KF) are chains
// of dependent v_fma_f64 at one or two waves per SIMD; this is the number their "clocks per instruction" is read against.
//   hipcc -O3 --offload-arch=gfx950 tools/experiments/valu_latency.hip -o tools/experiments/build/valu_latency
//   (run on the GPU box; prints one JSON line per variant)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

constexpr int ITER = 16384;

template <int CHAINS, int OP>
__global__ void __launch_bounds__(256) chain_kernel(double *out, long long *clk, double a, double b)
{
    double v[CHAINS];
    for (int c = 0; c < CHAINS; ++c) v[c] = a + threadIdx.x * 1e-9 + c;
    // warm the instruction cache with one untimed pass
    for (int pass = 0; pass < 2; ++pass) {
        __builtin_amdgcn_s_waitcnt(0);
        const long long t0 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
        for (int i = 0; i < ITER / 16; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
#pragma unroll
                for (int c = 0; c < CHAINS; ++c) {
                    if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[c]) : "v"(b), "v"(a));
                    else if (OP == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(v[c]) : "v"(b));
                    else if (OP == 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v[c]) : "v"(b));
                    else if (OP == 3) asm volatile("v_rsq_f64 %0, %0" : "+v"(v[c]));
                    else if (OP == 4) asm volatile("v_rcp_f64 %0, %0" : "+v"(v[c]));
                    else if (OP == 5) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(reinterpret_cast<float &>(v[c])) : "v"((float)b), "v"((float)a));
                    else if (OP == 6) asm volatile("v_floor_f64 %0, %0" : "+v"(v[c]));
                    else if (OP == 7) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(v[c]) : "v"(0));
                    else if (OP == 8) asm volatile("v_trunc_f64 %0, %0" : "+v"(v[c]));
                    else if (OP == 9) asm volatile("v_fract_f64 %0, %0" : "+v"(v[c]));
                    else if (OP == 10) asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(reinterpret_cast<unsigned &>(v[c])) : "v"(v[c]));          // f64 -> u32 (result reused as bits)
                    else if (OP == 11) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(v[c]) : "v"(reinterpret_cast<unsigned &>(v[c])));
                    else if (OP == 12) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(reinterpret_cast<unsigned &>(v[c])) : "v"(3u));
                    else if (OP == 13) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(v[c]) : "v"(3u), "v"(5u) : "vcc");
                    else if (OP == 14) asm volatile("v_rndne_f64 %0, %0" : "+v"(v[c]));
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0);
        const long long t1 = __builtin_amdgcn_s_memtime();
        if (pass == 1 && (threadIdx.x & 63) == 0) clk[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    }
    double s = 0;
    for (int c = 0; c < CHAINS; ++c) s += v[c];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int CHAINS, int OP>
static void run(const char *op, int blocks_per_cu)
{
    const int blocks = 256 * blocks_per_cu;
    double *out;
    long long *clk;
    hipMalloc(&out, sizeof(double) * blocks * 256);
    hipMalloc(&clk, sizeof(long long) * blocks * 4);
    hipLaunchKernelGGL((chain_kernel<CHAINS, OP>), dim3(blocks), dim3(256), 0, 0, out, clk, 1.0000001, 0.9999999);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((chain_kernel<CHAINS, OP>), dim3(blocks), dim3(256), 0, 0, out, clk, 1.0000001, 0.9999999);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks * 4);
    hipMemcpy(h.data(), clk, sizeof(long long) * blocks * 4, hipMemcpyDeviceToHost);
    double sum = 0;
    for (long long v : h) sum += (double)v;
    const double memtime_per_instr = sum / h.size() / (double)(ITER * CHAINS);
    // s_memtime counts a constant 100 MHz clock on gfx9: convert with the kernel's wall time instead
    const double ns_per_instr = (double)ms * 1e6 / 2.0 / (double)(ITER * CHAINS);   // two passes
    printf("{\"op\": \"%s\", \"chains_per_wave\": %d, \"waves_per_simd\": %d, \"memtime_ticks_per_instr\": %.4f, "
           "\"kernel_ms\": %.4f, \"ns_per_instr_per_wave\": %.3f}\n",
           op, CHAINS, blocks_per_cu, memtime_per_instr, ms, ns_per_instr);
    hipFree(out);
    hipFree(clk);
}

int main()
{
    for (int w = 1; w <= 4; w *= 2) {
        run<1, 0>("v_fma_f64", w);
        run<2, 0>("v_fma_f64", w);
        run<4, 0>("v_fma_f64", w);
        run<8, 0>("v_fma_f64", w);
    }
    for (int w = 1; w <= 2; ++w) {
        run<1, 1>("v_add_f64", w);
        run<4, 1>("v_add_f64", w);
        run<1, 2>("v_mul_f64", w);
        run<4, 2>("v_mul_f64", w);
        run<1, 3>("v_rsq_f64", w);
        run<4, 3>("v_rsq_f64", w);
        run<1, 4>("v_rcp_f64", w);
        run<4, 4>("v_rcp_f64", w);
        run<1, 5>("v_fma_f32", w);
        run<4, 5>("v_fma_f32", w);
        run<4, 6>("v_floor_f64", w);
        run<4, 7>("v_ldexp_f64", w);
        run<4, 8>("v_trunc_f64", w);
        run<4, 9>("v_fract_f64", w);
        run<4, 10>("v_cvt_u32_f64", w);
        run<4, 11>("v_cvt_f64_u32", w);
        run<4, 12>("v_mul_lo_u32", w);
        run<4, 13>("v_mad_u64_u32", w);
        run<4, 14>("v_rndne_f64", w);
    }
    return 0;
}
