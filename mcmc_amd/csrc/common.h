// Shared host-side declarations of the engine (error handling, launch plumbing).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <vector>

#include "../../include/mcmc_amd.h"

namespace mcmc {

// largest d handled by the register-resident one-chain-per-lane kernels
constexpr int kMaxLaneDimRwm = 12;
// largest d of the paired-wave variant (sampler + generator wavefront per 64 chains, <= 256 registers each)
constexpr int kMaxPairDimRwm = 10;
// largest d of the LDS-resident kernels (one 64-lane workgroup must fit 160 KiB of LDS)
constexpr int kMaxDimGibbs = 256;
// above this d the Gibbs regression rows are staged through LDS by the workgroup (wide variant)
constexpr int kGibbsLaneMaxDim = 32;
constexpr int kMaxDimRwm = 128;

int set_error(int code, const char *fmt, ...);

#define MCMC_HIP_CHECK(expr)                                                                       \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return ::mcmc::set_error(MCMC_E_HIP, "%s failed: %s (%s:%d)", #expr,                   \
                                     hipGetErrorString(_e), __FILE__, __LINE__);                   \
    } while (0)

// ---- arguments common to every Gibbs kernel (plain data, passed by value) -------------
struct GibbsArgs {
    int64_t n_chains;
    int64_t chain_offset;
    int32_t d;
    int32_t n_sweeps;
    int32_t thin;
    int32_t burn;
    uint64_t seed;
    const double *d_init;   // [d][n] or nullptr (then the shared start is used)
    const double *d_U;      // injected uniforms or nullptr (Philox)
    double *d_trace;
    double *d_final;
    double *d_mean;
    double *d_m2;
    int32_t *d_status;
};

// shared (per-call) Gibbs parameters on the host: mu, sd, lower, upper, init, P (row i has d-1 entries)
struct GibbsParams {
    int d;
    std::vector<double> mu, sd, lower, upper, init, P;
};

int launch_gibbs(const GibbsArgs &a, const GibbsParams &p, bool tail_safe, hipStream_t s);
int launch_gibbs_streams(uint64_t seed, int64_t chain_offset, int64_t n_chains, int32_t n_sweeps,
                         int32_t d, double *d_U, hipStream_t s);
int launch_rtnorm(int64_t n, double mu, double sigma, double a, double b, const double *d_F,
                  uint64_t seed, int64_t offset, bool tail_safe, double *d_out, hipStream_t s);

// ---- arguments common to every random-walk Metropolis kernel ---------------------------
struct RwmArgs {
    int64_t n_chains;
    int64_t chain_offset;
    int32_t d;
    int32_t n_iterations;
    int32_t adapt_size_start;
    int32_t adapt_shape_start;
    int32_t acceptance_window;
    int32_t thin;
    double acceptance_rate_weight;
    double max_scaling_sd;
    double shape_scaling_sd;   // 2.38 / sqrt(d), evaluated on the host (adaptMCMC_source.cpp:137)
    double twopi_pow_d;        // pow(2*pi, d) (adaptMCMC_source.cpp:56)
    uint64_t seed;
    const double *d_cool_pow;  // [n_iterations+1]: pow(adapt_size_cooling, i - adapt_size_start) (:123), or nullptr
    const double *d_init;
    const double *d_Z;
    const double *d_U;
    double *d_trace;
    double *d_sigma_trace;
    double *d_final_theta;
    double *d_final_mean;
    double *d_final_cov;
    double *d_acceptance_rate;
    int32_t *d_status;
    int32_t *d_shape_iterations;
    uint8_t *d_accepted;
    double *d_log_acceptance;
    double *d_log_A;
    double *d_scaling_sd;
    double *d_scaling_multiplier;
    double *d_rate_trace;
};

struct RwmParams {
    int d;
    int target_kind;
    std::vector<double> init;       // [d]
    std::vector<double> cov_packed; // upper triangle of covmat, (r,c) r<=c at c*(c+1)/2 + r
    std::vector<double> R0_packed;  // chol(covmat), same packing
    double det0;                    // det(covmat) = prod(diag R0)^2
    std::vector<double> tparams;    // target parameter block
};

int launch_rwm_lane(const RwmArgs &a, const RwmParams &p, hipStream_t s);
int launch_rwm_block(const RwmArgs &a, const RwmParams &p, hipStream_t s);
bool rwm_pair_serves(const RwmArgs &a);
int launch_rwm_pair(const RwmArgs &a, const RwmParams &p, hipStream_t s);
int launch_rwm_streams(uint64_t seed, int64_t chain_offset, int64_t n_chains, int32_t n_iterations,
                       int32_t d, double *d_Z, double *d_U, hipStream_t s);
int launch_init_normal(uint64_t seed, int64_t chain_offset, int64_t n_chains, int32_t d,
                       const double *h_mean, double sd, double *d_out, hipStream_t s);

// packed upper-triangular index, r <= c
__host__ __device__ constexpr int tri(int r, int c) { return c * (c + 1) / 2 + r; }

} // namespace mcmc
