// Development microbenchmark (round 2): issue cost and correctness of the cross-lane / matrix primitives the
// row kernel (rwm_row.hip) and the MFMA Cholesky experiment are built on (MI355X, gfx950).
//   v_fmac_f64_dpp row_newbcast:k   acc += (lane k of this 16-lane row).x * y   -- one instruction
//   v_permlane16_swap_b32           both rows of a 32-lane half get row A's (resp. row B's) values
//   v_mfma_f64_16x16x4_f64          16x16 (+)= 16x4 * 4x16 in FP64
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_r2.hip -o tools/ubench_r2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cmath>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITERS = 2048;
typedef double d4 __attribute__((ext_vector_type(4)));

template <int K>
__device__ __forceinline__ void fmac_bcast(double &acc, double x, double y)
{
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(y), "n"(K));
}

template <int MODE>
__global__ __launch_bounds__(256) void k(double *out, double seed, long long *cyc)
{
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const double b = 1.0000001, c = 1e-9;
    double x = 1e-7 * threadIdx.x, y = 1e-3;
    d4 m0 = {0, 0, 0, 0}, m1 = m0, m2 = m0, m3 = m0;
    double T[28];
    d4 M[7];
#pragma unroll
    for (int q = 0; q < 28; ++q) T[q] = seed * q;
#pragma unroll
    for (int t = 0; t < 7; ++t) M[t] = m0;
    long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < ITERS; ++i) {
        if (MODE == 0) { // 8 independent fma
            a0 = fma(a0, b, c); a1 = fma(a1, b, c); a2 = fma(a2, b, c); a3 = fma(a3, b, c);
            a4 = fma(a4, b, c); a5 = fma(a5, b, c); a6 = fma(a6, b, c); a7 = fma(a7, b, c);
        } else if (MODE == 1) { // 8 independent fmac dpp row_newbcast
            fmac_bcast<0>(a0, x, y); fmac_bcast<1>(a1, x, y); fmac_bcast<2>(a2, x, y); fmac_bcast<3>(a3, x, y);
            fmac_bcast<4>(a4, x, y); fmac_bcast<5>(a5, x, y); fmac_bcast<6>(a6, x, y); fmac_bcast<7>(a7, x, y);
        } else if (MODE == 2) { // one dependent chain of fmac dpp (the triangular matvec: one accumulator)
            fmac_bcast<0>(a0, x, y); fmac_bcast<1>(a0, x, y); fmac_bcast<2>(a0, x, y); fmac_bcast<3>(a0, x, y);
            fmac_bcast<4>(a0, x, y); fmac_bcast<5>(a0, x, y); fmac_bcast<6>(a0, x, y); fmac_bcast<7>(a0, x, y);
        } else if (MODE == 3) { // mfma f64 16x16x4, 4 independent accumulators
            m0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, m0, 0, 0, 0);
            m1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, m1, 0, 0, 0);
            m2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, m2, 0, 0, 0);
            m3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, m3, 0, 0, 0);
        } else if (MODE == 4) { // mfma f64, one dependent accumulator
            m0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, m0, 0, 0, 0);
            m0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, m0, 0, 0, 0);
            m0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, m0, 0, 0, 0);
            m0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, m0, 0, 0, 0);
        } else if (MODE == 5) { // mfma f64 (1) + 8 independent fma: do they overlap?
            m0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, m0, 0, 0, 0);
            a0 = fma(a0, b, c); a1 = fma(a1, b, c); a2 = fma(a2, b, c); a3 = fma(a3, b, c);
            a4 = fma(a4, b, c); a5 = fma(a5, b, c); a6 = fma(a6, b, c); a7 = fma(a7, b, c);
        } else if (MODE == 6) { // permlane16_swap on both halves of a double, 4 per iteration
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint64_t u = (uint64_t)__double_as_longlong(a0);
                auto lo = __builtin_amdgcn_permlane16_swap((uint32_t)u, (uint32_t)u, false, false);
                auto hi = __builtin_amdgcn_permlane16_swap((uint32_t)(u >> 32), (uint32_t)(u >> 32), false, false);
                a0 = __longlong_as_double((long long)(((uint64_t)hi[0] << 32) | lo[0])) + __longlong_as_double((long long)(((uint64_t)hi[1] << 32) | lo[1]));
            }
        } else if (MODE == 7) {
            // rank-16 trailing update of the d = 100 Cholesky as rwm_block.hip does it: per pivot every thread updates its
            // 28 tile elements, a_ik -= (a_ji / a_jj) a_jk -- 28 FMAs with 7 + 7 operands (here from registers; the kernel
            // reads them from LDS and scales one side), 16 pivots = one 16-row panel
#pragma unroll
            for (int piv = 0; piv < 16; ++piv) {
                const double r0 = a0 * (piv + 1), r1 = a1 + piv, r2 = a2 - piv, r3 = a3 * 0.5, r4 = a4, r5 = a5, r6 = a6;
                int q = 0;
#pragma unroll
                for (int bb = 0; bb < 7; ++bb) {
                    const double rk = (bb == 0 ? r0 : bb == 1 ? r1 : bb == 2 ? r2 : bb == 3 ? r3 : bb == 4 ? r4 : bb == 5 ? r5 : r6);
#pragma unroll
                    for (int aa = 0; aa <= bb; ++aa, ++q) {
                        const double ri = (aa == 0 ? r6 : aa == 1 ? r5 : aa == 2 ? r4 : aa == 3 ? r3 : aa == 4 ? r2 : aa == 5 ? r1 : r0);
                        T[q] = fma(-ri, rk, T[q]);
                    }
                }
            }
        } else if (MODE == 8) {
            // the same update with the matrix cores: a wave owns 7 of the 28 tiles (16 x 16 accumulators, 8 VGPRs each) and
            // applies the 16-row panel as four 16x16x4 steps per tile: 28 v_mfma_f64_16x16x4_f64 per wave and panel
#pragma unroll
            for (int t = 0; t < 7; ++t) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) M[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(x + ks, y - t, M[t], 0, 0, 0);
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double extra = 0.0;
    if (MODE == 7) {
#pragma unroll
        for (int q = 0; q < 28; ++q) extra += T[q];
    }
    if (MODE == 8) {
#pragma unroll
        for (int t = 0; t < 7; ++t) extra += M[t][0] + M[t][3];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = extra + a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + m0[0] + m1[1] + m2[2] + m3[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

// correctness: what each primitive really does, with exact small integers
__global__ void check(double *out)
{
    const int l = threadIdx.x; // one wave
    double acc = 0.0;
    fmac_bcast<5>(acc, (double)(l + 1), 2.0); // expect 2 * (16 * (l / 16) + 5 + 1)
    out[l] = acc;
    const uint32_t v = 100 + l;
    auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    out[64 + l] = r[0];
    out[128 + l] = r[1];
    // mfma: A[i][k] = 10 i + k (lane l: i = l & 15, k = l >> 4), B[k][j] = (k == 0) * (j + 1) + (k == 2) * 100  (asymmetric)
    const int i = l & 15, kk = l >> 4;
    const double A = 10.0 * i + kk, B = (kk == 0 ? (double)(i + 1) : 0.0) + (kk == 2 ? 100.0 : 0.0);
    d4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(A, B, c, 0, 0, 0);
    for (int r4 = 0; r4 < 4; ++r4) out[192 + 4 * l + r4] = c[r4];
}

template <int MODE>
int run(const char *name, int per_iter, int waves_per_simd, double *d_out, long long *d_cyc)
{
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
    double ops_per_wave = (double)ITERS * per_iter;
    printf("%-34s waves/SIMD=%d  %8.3f ms  memtime/op=%7.2f  cyc/op/wave@2.4GHz=%7.1f  per-SIMD op interval %6.2f cyc\n", name, waves_per_simd, ms,
           cyc / ops_per_wave, ms * 1e6 / ops_per_wave * 2.4, ms * 1e6 / ops_per_wave * 2.4 / waves_per_simd);
    return 0;
}

int main()
{
    double *d_out; long long *d_cyc;
    CHECK(hipMalloc(&d_out, sizeof(double) * 256 * 8 * 256));
    CHECK(hipMalloc(&d_cyc, sizeof(long long)));
    hipLaunchKernelGGL(check, dim3(1), dim3(64), 0, 0, d_out);
    std::vector<double> h(192 + 256);
    CHECK(hipMemcpy(h.data(), d_out, h.size() * sizeof(double), hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; ++l) bad += h[l] != 2.0 * (16 * (l / 16) + 6);
    printf("fmac_dpp row_newbcast:5  %s (lane 0: %g, lane 17: %g, lane 63: %g)\n", bad ? "UNEXPECTED" : "ok", h[0], h[17], h[63]);
    printf("permlane16_swap(v,v): first  result lanes 0,15,16,31,32,48: %g %g %g %g %g %g\n", h[64], h[64 + 15], h[64 + 16], h[64 + 31], h[64 + 32], h[64 + 48]);
    printf("permlane16_swap(v,v): second result lanes 0,15,16,31,32,48: %g %g %g %g %g %g\n", h[128], h[128 + 15], h[128 + 16], h[128 + 31], h[128 + 32], h[128 + 48]);
    // mfma check: C[i][j] = sum_k A[i][k] B[k][j] = (10 i)(j + 1) + (10 i + 2) 100; C/D: col = lane & 15, row = (lane >> 4) + 4 reg
    bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int r4 = 0; r4 < 4; ++r4) {
            const int col = l & 15, row = (l >> 4) + 4 * r4;
            bad += h[192 + 4 * l + r4] != (10.0 * row) * (col + 1) + (10.0 * row + 2) * 100.0;
        }
    printf("mfma_f64_16x16x4 layout (A: i = l&15, k = l>>4; B: k = l>>4, j = l&15; C: col = l&15, row = (l>>4) + 4 reg): %s\n", bad ? "MISMATCH" : "ok");
    for (int w : {1, 2, 3}) {
        run<0>("fma_f64 x8 indep", 8, w, d_out, d_cyc);
        run<1>("fmac_f64_dpp newbcast x8 indep", 8, w, d_out, d_cyc);
        run<2>("fmac_f64_dpp newbcast dependent", 8, w, d_out, d_cyc);
        run<3>("mfma_f64_16x16x4 x4 indep", 4, w, d_out, d_cyc);
        run<4>("mfma_f64_16x16x4 dependent", 4, w, d_out, d_cyc);
        run<5>("1 mfma_f64 + 8 fma_f64 (per group)", 1, w, d_out, d_cyc);
        run<6>("permlane16_swap pair + add x4", 4, w, d_out, d_cyc);
        run<7>("16-row panel update, VALU (448 fma)", 1, w, d_out, d_cyc);
        run<8>("16-row panel update, MFMA (28 mfma)", 1, w, d_out, d_cyc);
    }
    return 0;
}
