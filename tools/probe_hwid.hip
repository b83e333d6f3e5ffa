// Development probe: which SIMD does wave w of a 512-thread workgroup land on (MI355X, gfx950)?
// Build: hipcc -O3 --offload-arch=gfx950 tools/probe_hwid.hip -o tools/probe_hwid
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ __launch_bounds__(512) void probe(uint32_t *out, int spin)
{
    extern __shared__ double lds[];
    const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4); // HW_REG_HW_ID
    double x = threadIdx.x;
    for (int i = 0; i < spin; ++i) x = fma(x, 1.0000001, 1e-9);
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = hw;
    if (x == 12345.678) lds[threadIdx.x] = x;
}

int main()
{
    const int blocks = 512, waves = 8;
    uint32_t *d;
    hipMalloc(&d, blocks * waves * 4);
    for (int lds_kb : {150, 36}) {
        hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(probe, dim3(blocks), dim3(64 * waves), lds_kb * 1024, 0, d, 20000);
        hipDeviceSynchronize();
        std::vector<uint32_t> h(blocks * waves);
        hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
        int hist[8][4] = {};
        int balanced = 0;
        for (int b = 0; b < blocks; ++b) {
            int cnt[4] = {};
            for (int w = 0; w < waves; ++w) {
                int simd = (h[b * waves + w] >> 4) & 3;
                hist[w][simd]++;
                cnt[simd]++;
            }
            balanced += (cnt[0] == 2 && cnt[1] == 2 && cnt[2] == 2 && cnt[3] == 2);
        }
        printf("LDS %d KB per WG: %d of %d workgroups have 2 waves on every SIMD\n", lds_kb, balanced, blocks);
        for (int w = 0; w < waves; ++w)
            printf("  wave %d: simd0 %d simd1 %d simd2 %d simd3 %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
        for (int b = 0; b < 3; ++b) {
            printf("  wg %d:", b);
            for (int w = 0; w < waves; ++w)
                printf(" [simd %u slot %u cu %u]", (h[b * waves + w] >> 4) & 3, h[b * waves + w] & 15, (h[b * waves + w] >> 8) & 15);
            printf("\n");
        }
    }
    return 0;
}
