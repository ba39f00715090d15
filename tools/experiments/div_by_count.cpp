// Design check for DESIGN.md §8 item 2: can the resamplers' slot position fl(fl(i + u) / N) be produced without
// the IEEE division sequence (v_div_scale x2, v_rcp_f64, 8 FMAs, v_div_fmas, v_div_fixup per output)?
// Markstein's correction with y = RN(1/N):  q = x*y;  twice { r = fma(-q, N, x); q = fma(r, y, q) }  -- the first
// round makes q faithful, the second is then correctly rounded (no scaling needed: 1 <= N < 2^31, 0 <= x <= N,
// N's significand is never all ones).  This program compares it with x / N on random and adversarial inputs.
//     g++ -O2 -mfma -o /tmp/div_by_count tools/experiments/div_by_count.cpp && /tmp/div_by_count 400000000
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

static inline double div_by_count(double x, double N, double y)
{
    double q = x * y;
    double r = std::fma(-q, N, x);
    q = std::fma(r, y, q);
    r = std::fma(-q, N, x);
    return std::fma(r, y, q);
}

static uint64_t s[2] = {0x9E3779B97F4A7C15ull, 0xD1B54A32D192ED03ull};
static inline uint64_t rnd()
{   // xorshift128+
    uint64_t a = s[0], b = s[1];
    s[0] = b;
    a ^= a << 23;
    s[1] = a ^ b ^ (a >> 17) ^ (b >> 26);
    return s[1] + b;
}

int main(int argc, char **argv)
{
    const long long cases = argc > 1 ? atoll(argv[1]) : 100000000LL;
    long long bad = 0, tiny_bad = 0, tested = 0;
    for (long long c = 0; c < cases; ++c) {
        uint64_t a = rnd(), b = rnd(), d = rnd();
        // N: all magnitudes up to 2^31 - 2; sometimes a power of two, 2^k +- 1, or a small count
        uint32_t Ni;
        switch (a & 7) {
        case 0: Ni = 1u << ((a >> 8) % 31); break;
        case 1: Ni = (1u << (1 + (a >> 8) % 30)) + 1; break;
        case 2: Ni = (1u << (1 + (a >> 8) % 30)) - 1; break;
        case 3: Ni = 1 + (a >> 8) % 1000; break;
        default: Ni = (uint32_t)((a >> 8) % 2147483646u) + 1; break;
        }
        const double N = (double)Ni;
        const double y = 1.0 / N;
        const uint32_t i = (uint32_t)(b % Ni);
        // u in [0, 1): uniform mantissa, or tiny, or just below 1, or exactly 0
        double u;
        switch (d & 7) {
        case 0: u = 0.0; break;
        case 1: u = std::ldexp((double)(d >> 11), -53 - (int)((d >> 3) % 200)); break;   // tiny but normal
        case 2: u = 1.0 - std::ldexp((double)(1 + (d >> 40)), -53); break;
        default: u = std::ldexp((double)(d >> 11), -53); break;
        }
        const double x = u + (double)i;          // fl(u + i): resampling.py:103,139
        const double want = x / N;
        const double got = div_by_count(x, N, y);
        ++tested;
        if (std::memcmp(&want, &got, 8) != 0) {
            if (want < 0x1p-1000) ++tiny_bad;      // subnormal-range quotients need the real division
            else if (++bad < 10) std::printf("MISMATCH x=%a N=%a want=%a got=%a\n", x, N, want, got);
        }
    }
    std::printf("%lld cases, %lld mismatches (+ %lld with a quotient below 2^-1000)\n", tested, bad, tiny_bad);
    return bad != 0;
}
