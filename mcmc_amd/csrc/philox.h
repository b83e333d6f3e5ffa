// Counter-based RNG for the gfx950 kernels: Philox4x32-10 (Salmon et al., SC'11).
//
// Replaces the reference's sequential generators -- std::mt19937 + normal/uniform
// distributions (adaptMCMC_source.cpp:78-80,149-152,174) and R's runif
// (tmvrnormGibbs.cpp:52,97) -- with a stateless generator so that every
// (chain, iteration, draw) has its own 128-bit counter:
//
//   key     = (seed_lo, seed_hi)
//   counter = (chain_lo, chain_hi, iteration, (tag << 16) | block)
//
// Results therefore do not depend on how chains are spread over lanes, launches or GPUs.
#pragma once
#include <hip/hip_runtime.h>
#ifndef __HIPCC_RTC__
#include <stdint.h>
#endif

#include "fastmath.h"

namespace mcmc {

enum PhiloxTag : uint32_t {
    TAG_RWM_NORMAL = 0,   // block b -> proposal normals Y(2b), Y(2b+1) of that iteration
    TAG_RWM_UNIFORM = 1,  // block 0 -> the acceptance uniform A
    TAG_GIBBS = 2,        // block b -> uniforms of coordinates 2b, 2b+1 of that sweep
    TAG_INIT = 3,         // block b -> start-point normals 2b, 2b+1 (iteration 0)
    TAG_RTNORM = 4        // chain = sample index / 2, both uniforms of block 0
};

struct u32x4 {
    uint32_t x, y, z, w;
};

__host__ __device__ inline u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                               uint32_t k0, uint32_t k1)
{
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    constexpr uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)M0 * c0;
        uint64_t p1 = (uint64_t)M1 * c2;
#if defined(__HIP_DEVICE_COMPILE__)
        // three-input xor in one instruction (v_bitop3_b32, truth table 0x96); halves the bit-op count of a round
        uint32_t n0 = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c1, k0, 0x96);
        uint32_t n2 = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c3, k1, 0x96);
#else
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
#endif
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += W0;
        k1 += W1;
    }
    return u32x4{c0, c1, c2, c3};
}

// 52 random bits -> double in the OPEN interval (0,1): (m + 0.5) * 2^-52, m in [0, 2^52).
// Never 0 or 1, so log(u) and qnorm(u) stay finite; symmetric about 0.5.
__host__ __device__ inline double u01_from_bits(uint32_t hi, uint32_t lo)
{
    uint64_t w = ((uint64_t)hi << 32) | lo;
    uint64_t bits = 0x3FF0000000000000ull | (w >> 12);
    double f;
#if defined(__HIP_DEVICE_COMPILE__)
    f = __longlong_as_double((long long)bits);
#else
    __builtin_memcpy(&f, &bits, sizeof f);
#endif
#if defined(__HIP_DEVICE_COMPILE__)
    return f - sconst(0.99999999999999988897769753748, f); // f - (1 - 2^-53), exact
#else
    return f - 0.99999999999999988897769753748; // f - (1 - 2^-53), exact
#endif
}

struct PhiloxKey {
    uint32_t k0, k1;
    __host__ __device__ explicit PhiloxKey(uint64_t seed) : k0((uint32_t)seed), k1((uint32_t)(seed >> 32)) {}
};

// two uniforms of block `block` for (chain, iteration, tag)
__device__ inline void philox_uniform2(const PhiloxKey &key, uint64_t chain, uint32_t iteration,
                                       uint32_t tag, uint32_t block, double &u0, double &u1)
{
    u32x4 r = philox4x32_10((uint32_t)chain, (uint32_t)(chain >> 32), iteration, (tag << 16) | block,
                            key.k0, key.k1);
    u0 = u01_from_bits(r.x, r.y);
    u1 = u01_from_bits(r.z, r.w);
}

// Box-Muller: two independent N(0,1) from the 128 bits of one Philox block.
//   radius: R = sqrt(-2 log u0), u0 in (0,1) from words x,y (52 bits)
//   angle : uniform on the circle, built as pi*r + k*pi/2 with the quadrant k from the top two bits
//           of word z and r in (-1/4, 1/4) from the next 52 bits of (z,w): no range reduction needed.
__device__ inline void box_muller_bits(const u32x4 &b, double &z0, double &z1)
{
    const double u0 = u01_from_bits(b.x, b.y);
    const double R = fm_sqrt(-2.0 * fm_log<false>(u0));
    const uint32_t k = b.z >> 30;
    const double f = u01_from_bits(b.z << 2 | b.w >> 30, b.w << 2); // 52 of the remaining 62 bits
    double s, c;
    fm_sincospi_quarter(0.5 * f - 0.25, s, c);
    // rotate by k quarter turns
    const double s1 = (k & 1) ? c : s;
    const double c1 = (k & 1) ? -s : c;
    const double sg = (k & 2) ? -1.0 : 1.0;
    z0 = R * (sg * c1);
    z1 = R * (sg * s1);
}

__device__ inline void philox_normal2(const PhiloxKey &key, uint64_t chain, uint32_t iteration,
                                      uint32_t tag, uint32_t block, double &z0, double &z1)
{
    const u32x4 r = philox4x32_10((uint32_t)chain, (uint32_t)(chain >> 32), iteration, (tag << 16) | block,
                                  key.k0, key.k1);
    box_muller_bits(r, z0, z1);
}

// The same two normals with the log and sin/cos coefficients already in scalar registers (fastmath.h: fm_tab)
__device__ inline void philox_normal2_t(const PhiloxKey &key, uint64_t chain, uint32_t iteration, uint32_t tag,
                                        uint32_t block, double &z0, double &z1, const ScalarCoef<16> &tl,
                                        const ScalarCoef<16> &ts)
{
    const u32x4 b = philox4x32_10((uint32_t)chain, (uint32_t)(chain >> 32), iteration, (tag << 16) | block,
                                  key.k0, key.k1);
    const double u0 = u01_from_bits(b.x, b.y);
    const double R = fm_sqrt(-2.0 * fm_log_t(u0, tl));
    const uint32_t k = b.z >> 30;
    const double f = u01_from_bits(b.z << 2 | b.w >> 30, b.w << 2);
    double s, c;
    fm_sincospi_quarter_t(0.5 * f - 0.25, s, c, ts);
    const double s1 = (k & 1) ? c : s;
    const double c1 = (k & 1) ? -s : c;
    const double sg = (k & 2) ? -1.0 : 1.0;
    z0 = R * (sg * c1);
    z1 = R * (sg * s1);
}

// One Philox block per lane, either kind: z0, z1 are the two normals of philox_normal2_t for (tag, block) and lg is
// log(u0) of the block's first uniform -- for (TAG_*_UNIFORM, 0) exactly the log(A) that philox_uniform2 + fm_log_t
// give.  Lets one wavefront draw the d normals and the uniform of an iteration in ONE pass (no divergent second branch).
__device__ inline void philox_normal2_log_t(const PhiloxKey &key, uint64_t chain, uint32_t iteration, uint32_t tag,
                                            uint32_t block, double &z0, double &z1, double &lg, const ScalarCoef<16> &tl,
                                            const ScalarCoef<16> &ts)
{
    const u32x4 b = philox4x32_10((uint32_t)chain, (uint32_t)(chain >> 32), iteration, (tag << 16) | block,
                                  key.k0, key.k1);
    const double u0 = u01_from_bits(b.x, b.y);
    lg = fm_log_t(u0, tl);
    const double R = fm_sqrt(-2.0 * lg);
    const uint32_t k = b.z >> 30;
    const double f = u01_from_bits(b.z << 2 | b.w >> 30, b.w << 2);
    double s, c;
    fm_sincospi_quarter_t(0.5 * f - 0.25, s, c, ts);
    const double s1 = (k & 1) ? c : s;
    const double c1 = (k & 1) ? -s : c;
    const double sg = (k & 2) ? -1.0 : 1.0;
    z0 = R * (sg * c1);
    z1 = R * (sg * s1);
}

} // namespace mcmc
