// Development probe: which XCD (XCC_ID) does workgroup b of a 256-thread, 51 KB-LDS launch run on (MI355X, gfx950)?
// The per-workgroup kernel maps chains to workgroups assuming XCD = blockIdx % 8 (rwm_block.hip).
// Build: hipcc -O3 --offload-arch=gfx950 tools/probe_xcc.hip -o tools/probe_xcc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ __launch_bounds__(256) void probe(uint32_t *out, int spin)
{
    extern __shared__ double lds[];
    const uint32_t xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20); // HW_REG_XCC_ID
    double x = threadIdx.x;
    for (int i = 0; i < spin + (int)(blockIdx.x % 7) * 50; ++i) x = fma(x, 1.0000001, 1e-9);
    if (threadIdx.x == 0) out[blockIdx.x] = xcc;
    if (x == 12345.678) lds[threadIdx.x] = x;
}

int main()
{
    const int blocks = 4096;
    uint32_t *d;
    hipMalloc(&d, blocks * 4);
    hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 51 * 1024, 0, d, 20000);
    hipDeviceSynchronize();
    std::vector<uint32_t> h(blocks);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    int match = 0, match_first = 0;
    for (int b = 0; b < blocks; ++b) {
        match += (int)(h[b] & 15) == b % 8;
        if (b < 512) match_first += (int)(h[b] & 15) == b % 8;
    }
    printf("XCC_ID == blockIdx %% 8: %d of %d workgroups (first 512: %d)\n", match, blocks, match_first);
    printf("first 24:");
    for (int b = 0; b < 24; ++b) printf(" %u", h[b] & 15);
    printf("\nblocks 512..535:");
    for (int b = 512; b < 536; ++b) printf(" %u", h[b] & 15);
    printf("\nraw[0] = 0x%x\n", h[0]);
    return 0;
}
