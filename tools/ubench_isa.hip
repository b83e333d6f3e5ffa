// Development microbenchmark: per-instruction issue cost of the FP64 / integer ops the MCMC
// kernels are made of, one wave per SIMD and eight waves per SIMD (MI355X, gfx950).
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_isa.hip -o tools/ubench_isa
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITERS = 4096;

template <int MODE>
__global__ __launch_bounds__(256) void k(double *out, double seed, long long *cyc)
{
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const double b = 1.0000001, c = 1e-9;
    uint32_t u0 = threadIdx.x * 2654435761u, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3;
    long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < ITERS; ++i) {
        if (MODE == 0) { // 8 independent fma chains
            a0 = fma(a0, b, c); a1 = fma(a1, b, c); a2 = fma(a2, b, c); a3 = fma(a3, b, c);
            a4 = fma(a4, b, c); a5 = fma(a5, b, c); a6 = fma(a6, b, c); a7 = fma(a7, b, c);
        } else if (MODE == 1) { // one dependent chain, 8 per iter
#pragma unroll
            for (int j = 0; j < 8; ++j) a0 = fma(a0, b, c);
        } else if (MODE == 2) { // 2 chains
#pragma unroll
            for (int j = 0; j < 4; ++j) { a0 = fma(a0, b, c); a1 = fma(a1, b, c); }
        } else if (MODE == 3) { // mad_u64_u32 x8 independent-ish (4 chains)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                uint64_t p0 = (uint64_t)u0 * 0xD2511F53u + u1; uint64_t p1 = (uint64_t)u2 * 0xCD9E8D57u + u3;
                u0 = (uint32_t)(p0 >> 32) ^ u1; u1 = (uint32_t)p0; u2 = (uint32_t)(p1 >> 32) ^ u3; u3 = (uint32_t)p1;
                uint64_t p2 = (uint64_t)u1 * 0xD2511F53u + u0; uint64_t p3 = (uint64_t)u3 * 0xCD9E8D57u + u2;
                u1 = (uint32_t)(p2 >> 32) ^ u0; u0 = (uint32_t)p2; u3 = (uint32_t)(p3 >> 32) ^ u2; u2 = (uint32_t)p3;
            }
        } else if (MODE == 4) { // rcp f64 x8 independent
            a0 = __builtin_amdgcn_rcp(a0); a1 = __builtin_amdgcn_rcp(a1); a2 = __builtin_amdgcn_rcp(a2); a3 = __builtin_amdgcn_rcp(a3);
            a4 = __builtin_amdgcn_rcp(a4); a5 = __builtin_amdgcn_rcp(a5); a6 = __builtin_amdgcn_rcp(a6); a7 = __builtin_amdgcn_rcp(a7);
        } else if (MODE == 5) { // rsq f64 x8
            a0 = __builtin_amdgcn_rsq(a0); a1 = __builtin_amdgcn_rsq(a1); a2 = __builtin_amdgcn_rsq(a2); a3 = __builtin_amdgcn_rsq(a3);
            a4 = __builtin_amdgcn_rsq(a4); a5 = __builtin_amdgcn_rsq(a5); a6 = __builtin_amdgcn_rsq(a6); a7 = __builtin_amdgcn_rsq(a7);
        } else if (MODE == 6) { // full IEEE divide x8
            a0 = b / a0; a1 = b / a1; a2 = b / a2; a3 = b / a3; a4 = b / a4; a5 = b / a5; a6 = b / a6; a7 = b / a7;
        } else if (MODE == 7) { // exp x4 (ocml)
            a0 = exp(-a0 * 1e-3); a1 = exp(-a1 * 1e-3); a2 = exp(-a2 * 1e-3); a3 = exp(-a3 * 1e-3);
        } else if (MODE == 8) { // log x4 (ocml)
            a0 = log(a0 + 2); a1 = log(a1 + 2); a2 = log(a2 + 2); a3 = log(a3 + 2);
        } else if (MODE == 9) { // sqrt x4 (ocml, correctly rounded)
            a0 = sqrt(a0 + 2); a1 = sqrt(a1 + 2); a2 = sqrt(a2 + 2); a3 = sqrt(a3 + 2);
        } else if (MODE == 10) { // sincospi x2
            double s, cc; sincospi(a0 * 1e-3, &s, &cc); a0 = s + cc; sincospi(a1 * 1e-3, &s, &cc); a1 = s + cc;
        } else if (MODE == 11) { // f64 add x8 independent
            a0 += b; a1 += b; a2 += b; a3 += b; a4 += b; a5 += b; a6 += b; a7 += b;
        } else if (MODE == 12) { // f64 mul x8 independent
            a0 *= b; a1 *= b; a2 *= b; a3 *= b; a4 *= b; a5 *= b; a6 *= b; a7 *= b;
        } else if (MODE == 13) { // 32-bit int xor/add x8 (full-rate reference)
            u0 = (u0 ^ u1) + 0x9E3779B9u; u1 = (u1 ^ u2) + 0xBB67AE85u; u2 = (u2 ^ u3) + 0x9E3779B9u; u3 = (u3 ^ u0) + 0xBB67AE85u;
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (double)(u0 ^ u1 ^ u2 ^ u3);
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE>
int run(const char *name, int per_iter, int waves_per_simd, double *d_out, long long *d_cyc)
{
    // 256 CUs x 4 SIMDs; block = 256 threads = 4 waves = 1 wave per SIMD of one CU
    const int blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, 1.0, d_cyc);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, 1.0, d_cyc);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    long long cyc; CHECK(hipMemcpy(&cyc, d_cyc, sizeof cyc, hipMemcpyDeviceToHost));
    // s_memtime ticks at 100 MHz-ish constant clock on some parts; report both
    double ops_per_wave = (double)ITERS * per_iter;
    printf("%-28s waves/SIMD=%d  %8.3f ms  memtime/op=%7.2f  ns/op/wave=%7.3f  (=%6.1f cyc @2.4GHz)  per-SIMD op interval %6.2f cyc\n", name,
           waves_per_simd, ms, cyc / ops_per_wave, ms * 1e6 / ops_per_wave, ms * 1e6 / ops_per_wave * 2.4,
           ms * 1e6 / ops_per_wave * 2.4 / waves_per_simd);
    return 0;
}

int main()
{
    double *d_out; long long *d_cyc;
    CHECK(hipMalloc(&d_out, sizeof(double) * 256 * 8 * 256));
    CHECK(hipMalloc(&d_cyc, sizeof(long long)));
    for (int w : {1, 2, 8}) {
        run<0>("fma_f64 x8 indep", 8, w, d_out, d_cyc);
        run<1>("fma_f64 dependent", 8, w, d_out, d_cyc);
        run<2>("fma_f64 2 chains", 8, w, d_out, d_cyc);
        run<11>("add_f64 x8 indep", 8, w, d_out, d_cyc);
        run<12>("mul_f64 x8 indep", 8, w, d_out, d_cyc);
        run<3>("mad_u64_u32 (8/iter)", 8, w, d_out, d_cyc);
        run<13>("int xor+add (8/iter)", 8, w, d_out, d_cyc);
        run<4>("rcp_f64 x8", 8, w, d_out, d_cyc);
        run<5>("rsq_f64 x8", 8, w, d_out, d_cyc);
        run<6>("div f64 x8 (IEEE)", 8, w, d_out, d_cyc);
        run<7>("exp f64 x4 (ocml)", 4, w, d_out, d_cyc);
        run<8>("log f64 x4 (ocml)", 4, w, d_out, d_cyc);
        run<9>("sqrt f64 x4 (ocml)", 4, w, d_out, d_cyc);
        run<10>("sincospi f64 x2 (ocml)", 2, w, d_out, d_cyc);
    }
    return 0;
}
