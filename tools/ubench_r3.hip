// Round-3 microbenchmark that closes north_star's MFMA experiment (BASELINE config 4; replaces the fresh chol of
// adaptMCMC_source.cpp:153 at d = 100): the issue interval of v_mfma_f64_16x16x4_f64 on gfx950, measured INSIDE the
// kernel over >= 1.3e5 MFMAs per wave (round 2 timed 0.1 ms launches with HIP events and reported
// "independent" MFMAs slower than dependent ones).
//   * time: s_memtime (shader clock) around the loop, per wave; s_memrealtime (100 MHz) beside it gives the clock
//   * the loop bodies are inline asm on fixed registers, so there is nothing for the compiler to insert between the
//     MFMAs; tools/ubench_r3.sh also dumps the ISA of the loops into profiles/r3_ubench_mfma_isa.txt
//   * NACC independent accumulators (1, 2, 4, 8), 1 / 2 / 4 waves per SIMD; VALU FP64 FMA streams beside them
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_r3.hip -o tools/ubench_r3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <algorithm>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITERS = 4096; // x 32 MFMAs (or 32 FMAs) per iteration = 131 072 per wave
typedef double d4 __attribute__((ext_vector_type(4)));

// MODE 0..3: 32 MFMAs per iteration over 1 / 2 / 4 / 8 independent accumulators (round-robin)
// MODE 4: 32 independent v_fma_f64 per iteration (8 accumulators); MODE 5: 32 dependent v_fma_f64 (1 accumulator)
// MODE 6: 8 MFMAs (8 accumulators) interleaved with 24 v_fma_f64: do the two pipes overlap?
template <int MODE>
__global__ __launch_bounds__(256) void k(double *out, double seed, long long *cyc, long long *real)
{
    const double x = 1e-7 * (threadIdx.x + 1), y = 1e-3 + seed;
    d4 m[8];
    double a[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        m[q] = d4{0, 0, 0, 0};
        a[q] = seed + q + threadIdx.x;
    }
    const double b = 1.0000001, c = 1e-9;
    const long long r0 = __builtin_amdgcn_s_memrealtime();
    const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < ITERS; ++i) {
        if constexpr (MODE <= 3) {
            constexpr int NACC = 1 << MODE;
#pragma unroll
            for (int u = 0; u < 32; ++u)
                asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(m[u % NACC]) : "v"(x), "v"(y));
        } else if constexpr (MODE == 4) {
#pragma unroll
            for (int u = 0; u < 32; ++u) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[u % 8]) : "v"(b), "v"(c));
        } else if constexpr (MODE == 5) {
#pragma unroll
            for (int u = 0; u < 32; ++u) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c));
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(m[u]) : "v"(x), "v"(y));
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[(3 * u) % 8]) : "v"(b), "v"(c));
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[(3 * u + 1) % 8]) : "v"(b), "v"(c));
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[(3 * u + 2) % 8]) : "v"(b), "v"(c));
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    const long long r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) s += a[q] + m[q][0] + m[q][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) { // every wave reports; the host takes the median
        const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
        cyc[w] = t1 - t0;
        real[w] = r1 - r0;
    }
}

template <int MODE>
int run(const char *name, int mfma_per_iter, int fma_per_iter, int waves_per_simd, double *d_out, long long *d_cyc, long long *d_real)
{
    const int blocks = 256 * waves_per_simd, waves = blocks * 4;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, 1.0, d_cyc, d_real); // warm-up: clocks up
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, 1.0, d_cyc, d_real);
    CHECK(hipDeviceSynchronize());
    std::vector<long long> c(waves), r(waves);
    CHECK(hipMemcpy(c.data(), d_cyc, sizeof(long long) * waves, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(r.data(), d_real, sizeof(long long) * waves, hipMemcpyDeviceToHost));
    std::sort(c.begin(), c.end());
    std::sort(r.begin(), r.end());
    const double cm = (double)c[waves / 2], rm = (double)r[waves / 2];
    const double mhz = cm / (rm / 100.0); // s_memrealtime ticks at 100 MHz
    const double per_wave_iter = cm / ITERS;
    printf("%-44s waves/SIMD=%d  s_memtime/iter/wave %8.1f (min %8.1f max %8.1f)  s_memtime clock %6.0f MHz", name, waves_per_simd,
           per_wave_iter, (double)c[0] / ITERS, (double)c[waves - 1] / ITERS, mhz);
    if (mfma_per_iter) printf("  per-SIMD interval per MFMA %6.1f ticks", per_wave_iter / mfma_per_iter / waves_per_simd);
    if (fma_per_iter && !mfma_per_iter) printf("  per-SIMD interval per FMA %5.2f ticks", per_wave_iter / fma_per_iter / waves_per_simd);
    printf("\n");
    return 0;
}

int main()
{
    double *d_out;
    long long *d_cyc, *d_real;
    CHECK(hipMalloc(&d_out, sizeof(double) * 256 * 4 * 256));
    CHECK(hipMalloc(&d_cyc, sizeof(long long) * 4096));
    CHECK(hipMalloc(&d_real, sizeof(long long) * 4096));
    for (int w : {1, 2, 4}) {
        run<0>("mfma_f64_16x16x4 x32, 1 accumulator (dependent)", 32, 0, w, d_out, d_cyc, d_real);
        run<1>("mfma_f64_16x16x4 x32, 2 accumulators", 32, 0, w, d_out, d_cyc, d_real);
        run<2>("mfma_f64_16x16x4 x32, 4 accumulators", 32, 0, w, d_out, d_cyc, d_real);
        run<3>("mfma_f64_16x16x4 x32, 8 accumulators", 32, 0, w, d_out, d_cyc, d_real);
        run<4>("v_fma_f64 x32, 8 accumulators", 0, 32, w, d_out, d_cyc, d_real);
        run<5>("v_fma_f64 x32, 1 accumulator (dependent)", 0, 32, w, d_out, d_cyc, d_real);
        run<6>("8 mfma (8 acc) + 24 v_fma_f64 interleaved", 8, 24, w, d_out, d_cyc, d_real);
    }
    return 0;
}
