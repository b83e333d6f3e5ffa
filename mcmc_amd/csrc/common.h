// Shared host-side declarations of the engine (error handling, launch plumbing).
#pragma once
// Under hipRTC (user targets, rtc.hip) only the device-side declarations are seen: the run-time
// compiler has no host headers.
#include <hip/hip_runtime.h>
#ifndef __HIPCC_RTC__
#include <stdint.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <functional>
#include <vector>
#endif

#include "mcmc_amd.h"

namespace mcmc {

// largest d handled by the register-resident one-chain-per-lane kernels
constexpr int kMaxLaneDimRwm = 12;
// largest d of the paired-wave variant (sampler + generator wavefront per 64 chains, <= 256 registers each)
constexpr int kMaxPairDimRwm = 10;
// largest d of the LDS-resident kernels (one 64-lane workgroup must fit 160 KiB of LDS)
constexpr int kMaxDimGibbs = 256;
// above this d the Gibbs sampler runs draw / dot wavefront groups (measured crossover: d = 24 one chain per lane
// 1.79e9 against 1.66e9 chain-sweeps/s, d = 28 1.40e9 against 1.43e9, d = 32 1.10e9 against 1.26e9)
constexpr int kGibbsLaneMaxDim = 27;
constexpr int kMaxDimRwm = 128;
// largest d of the one-chain-per-row kernel (a chain on 16 or 32 lanes of a wavefront, rwm_row.hip)
constexpr int kMaxRowDimRwm = 64;

#ifndef __HIPCC_RTC__
int set_error(int code, const char *fmt, ...);

#define MCMC_HIP_CHECK(expr)                                                                       \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return ::mcmc::set_error(MCMC_E_HIP, "%s failed: %s (%s:%d)", #expr,                   \
                                     hipGetErrorString(_e), __FILE__, __LINE__);                   \
    } while (0)
#endif

// ---- arguments common to every Gibbs kernel (plain data, passed by value) -------------
struct GibbsArgs {
    int64_t n_chains;
    int64_t chain_offset;
    int32_t d;
    int32_t n_sweeps;
    int32_t thin;
    int32_t burn;
    int32_t sweep_offset;   // sweeps done before this launch (resume): Philox counter and thinning count from it
    uint64_t seed;
    const double *d_init;   // [d][n] or nullptr (then the shared start is used)
    const double *d_U;      // injected uniforms or nullptr (Philox)
    double *d_trace;
    double *d_final;
    double *d_mean;
    double *d_m2;
    int32_t *d_status;
};

#ifndef __HIPCC_RTC__
// shared (per-call) Gibbs parameters on the host: mu, sd, lower, upper, init, P (row i has d-1 entries)
struct GibbsParams {
    int d;
    std::vector<double> mu, sd, lower, upper, init, P;
};

int launch_gibbs(const GibbsArgs &a, const GibbsParams &p, bool tail_safe, hipStream_t s);
int launch_gibbs_noalloc(const GibbsArgs &a, const GibbsParams &p, bool tail_safe, hipStream_t s); // static device scratch
int gibbs_noalloc_max_dim();
int launch_gibbs_streams(uint64_t seed, int64_t chain_offset, int64_t n_chains, int32_t n_sweeps,
                         int32_t d, double *d_U, hipStream_t s);
int launch_rtnorm(int64_t n, double mu, double sigma, double a, double b, const double *d_F,
                  uint64_t seed, int64_t offset, bool tail_safe, double *d_out, hipStream_t s);
#endif

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
    const double *d_cool_pow;  // [cool_last+1]: pow(adapt_size_cooling, i - adapt_size_start) (:123) for the iterations of this
                               // launch, cut where the power has become 0 / Inf / 1 for good: index min(li, cool_last); or nullptr
    int32_t cool_last;
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
    // checkpoint / resume (rwm_state_layout below): iterations iter_offset+1 .. iter_offset+n_iterations run
    int32_t iter_offset;
    const double *d_state_in;
    double *d_state_out;
    // acceptance_window > 64 (accept_window.h): the outcome ring, [win_words][n_chains] 64-bit words, zeroed by the host
    unsigned long long *d_win_ring;
    int32_t win_words;
    // MCMC_FACTOR_RANK1 (per-workgroup kernel): the factor follows covmat_empirical by closed-form rank-1 updates and is
    // refactored when the iteration is a multiple of refresh_every
    int32_t factor_rank1;
    int32_t refresh_every;
    // MCMC_TRACE_CHAIN_MAJOR: d_trace is [n_chains][n_kept][d]; n_kept = rows this launch keeps
    int32_t trace_chain_major;
    int64_t n_kept;
};

#ifndef __HIPCC_RTC__
struct RwmParams {
    int d;
    int target_kind;
    std::vector<double> init;       // [d]
    std::vector<double> cov_packed; // upper triangle of covmat, (r,c) r<=c at c*(c+1)/2 + r
    std::vector<double> R0_packed;  // chol(covmat), same packing
    double det0;                    // det(covmat) = prod(diag R0)^2
    std::vector<double> tparams;    // target parameter block
    const void *user = nullptr;     // MCMC_TARGET_USER: the handle mcmc_target_compile returned
};

// How a Metropolis kernel gets onto the device: the kernels compiled into the library go through
// hipLaunchKernelGGL, a run-time compiled user-target module through hipModuleLaunchKernel (rtc.hip).
// `variant` picks the instantiation: lane kernel 0 / 1 = FULL; pair kernel 0 = no moments, 1 = MOMENTS,
// 2 = MOMENTS + STATE (checkpoint / resume).
using RwmLauncher = std::function<hipError_t(int variant, unsigned grid, unsigned block, size_t lds_bytes,
                                             hipStream_t s, const RwmArgs &a, const double *d_prm)>;

int launch_rwm_lane(const RwmArgs &a, const RwmParams &p, hipStream_t s);
int launch_rwm_block(const RwmArgs &a, const RwmParams &p, hipStream_t s);
bool rwm_pair_serves(const RwmArgs &a);
bool rwm_row_serves(const RwmArgs &a, int target_kind);
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per kernel, device and size: the call waits for the device to go
// idle when made while work is in flight (measured: 14 ms behind a 14 ms launch), so it must not precede every launch
hipError_t ensure_dynamic_lds(const void *kern, size_t bytes);
// host doubles -> device, ordered on the stream without waiting for it (kernel-argument fills; gibbs.hip)
int upload_block(double *d_dst, const double *h_src, size_t count, hipStream_t s);
int launch_rwm_row(const RwmArgs &a, const RwmParams &p, hipStream_t s);
int launch_rwm_pair(const RwmArgs &a, const RwmParams &p, hipStream_t s);
int launch_rwm_streams(uint64_t seed, int64_t chain_offset, int64_t n_chains, int32_t n_iterations,
                       int32_t d, double *d_Z, double *d_U, hipStream_t s);
int launch_init_normal(uint64_t seed, int64_t chain_offset, int64_t n_chains, int32_t d,
                       const double *h_mean, double sd, double *d_out, hipStream_t s);
// rtc.hip: the same two launches for a MCMC_TARGET_USER target
int launch_rwm_lane_user(const RwmArgs &a, const RwmParams &p, hipStream_t s);
int launch_rwm_pair_user(const RwmArgs &a, const RwmParams &p, hipStream_t s);
int launch_rwm_block_user(const RwmArgs &a, const RwmParams &p, hipStream_t s);
int launch_rwm_block_with(const RwmArgs &a, const RwmParams &p, hipStream_t s, bool user, const RwmLauncher &launch);
int user_target_param_count(const void *handle, int d);
#endif

// packed upper-triangular index, r <= c
__host__ __device__ constexpr int tri(int r, int c) { return c * (c + 1) / 2 + r; }
// column of packed element e (the inverse of tri): folds to a constant where e is one
__host__ __device__ constexpr int tri_col(int e)
{
    int c = 0;
    while ((c + 1) * (c + 2) / 2 <= e) ++c;
    return c;
}

// ---- chain state as a checkpoint holds it: [slot][chain] doubles ------------------------------
// theta[d] mean[d] covmat_empirical[T] (packed upper) R[T] (packed upper factor as the kernel keeps it),
// then kRwmStateScalars scalars.  theta / mean / R are in the WRITING kernel's own representation
// (the pair kernel keeps the Gaussian state centred, the lane kernel keeps R scaled), so a state is
// resumed by the kernel that wrote it; ST_TAG says which and is checked on the device.
enum RwmStateScalar {
    ST_TAG = 0,   // MCMC_KERNEL_* of the writer
    ST_ITER,      // iterations done
    ST_TCUR,      // target(theta_current)
    ST_RATE, ST_SD, ST_MULT, // acceptance_rate, scaling_sd, scaling_multiplier
    ST_SEFF, ST_PRODR, ST_ROOT, // scale kept apart from R, prod diag(R), 1/sqrt((2 pi)^d det)
    ST_WLO, ST_WHI, ST_WLEN,    // acceptance window: 64 outcomes (window > 64: the ring word in use) as two 32-bit halves, and its length
    ST_BASE,      // 1 when R factors covmat_empirical
    ST_STATUS, ST_NSHAPE,
    ST_WCOUNT,    // acceptance window > 64: accepted outcomes held
    ST_FAST,      // MCMC_FACTOR_RANK1: 1 when R tracks chol(covmat_empirical) and Q holds the whitened residual
    kRwmStateScalars
};
// With acceptance_window > 64 the state continues behind TOTAL with the outcome ring: word w as two 32-bit halves at
// TOTAL + 2 w, TOTAL + 2 w + 1 (mcmc_rwm_state_doubles_window).
// Q (per-workgroup kernel): the last residual theta - theta_mean [d], then its whitened form R^-T r [d] -- what a
// MCMC_FACTOR_RANK1 run needs to continue its factor updates bit for bit; the other kernels leave it untouched.
struct RwmStateLayout {
    int THETA, MEAN, C, R, S, Q, TOTAL;
};
__host__ __device__ constexpr RwmStateLayout rwm_state_layout(int d)
{
    const int T = d * (d + 1) / 2, S = 2 * d + 2 * T, Q = S + (int)kRwmStateScalars;
    return RwmStateLayout{0, d, 2 * d, 2 * d + T, S, Q, Q + 2 * d};
}

// ---- parameter-block / LDS layouts of the one-chain-per-lane kernels ---------------------------
// constexpr functions of (d, number of target parameters), so that the kernels use them as
// compile-time constants and the host can lay out the block for a run-time compiled kernel.
//
// lane kernel, block in HBM (doubles): init[D] cov[T] R0[T] root0 tprm[NP]; T = D(D+1)/2.
// cov = covmat_proposal_init and R0 = chol(covmat), both packed upper: (r,c), r<=c at c(c+1)/2+r.
struct LaneLayout {
    int T, NP, INIT, COV, R0, ROOT0, TPRM, TOTAL;
    int SH;          // shared parameters staged in LDS (16-B aligned size)
    int LDS_DOUBLES; // + covmat_empirical and theta_mean as [e][lane]
};
__host__ __device__ constexpr LaneLayout lane_layout(int D, int NP)
{
    const int T = D * (D + 1) / 2, TPRM = D + 2 * T + 1, TOTAL = TPRM + NP, SH = (TOTAL + 1) & ~1;
    return LaneLayout{T, NP, 0, D, D + T, D + 2 * T, TPRM, TOTAL, SH, SH + (T + D) * 64};
}

constexpr int kPairGroups = 4; // chain groups (of 64) per workgroup = sampler waves = generator waves

// pair kernel, block in HBM (doubles): min diag(covmat), -, tprm[NP] | init[D] R0[T] prodR0
// the first SH doubles are staged in LDS; the rest is read once (or on a rare phase change)
struct PairLayoutRT {
    int T, NP, COVD, TPRM, SH, INIT, R0, PROD0, TOTAL;
    // per chain group, all as [element][lane]:
    int DRAW;  // generator -> sampler: Y[D], exp(-q/2), log A
    int BACK;  // sampler -> generator: theta of the iteration just finished
    int CX;    // covmat_empirical: updated by the generator, read by the sampler when it factors
    int GROUP; // + 8 doubles of flags
    int LDS_DOUBLES;
};
__host__ __device__ constexpr PairLayoutRT pair_layout(int D, int NP, bool MOMENTS)
{
    const int T = D * (D + 1) / 2, SH = (2 + NP + 1) & ~1;
    const int DRAW = D + 2, BACK = MOMENTS ? D : 0, CX = MOMENTS ? T : 0, GROUP = (DRAW + BACK + CX) * 64 + 8;
    return PairLayoutRT{T, NP, 0, 2, SH, SH, SH + D, SH + D + T, SH + D + T + 1, DRAW, BACK, CX, GROUP, SH + kPairGroups * GROUP};
}

#ifndef __HIPCC_RTC__
// the launches proper, shared by the built-in and the run-time compiled kernels; `packed_tparams`
// is the target parameter block as the pair kernel wants it (PackedTarget::pack)
int launch_rwm_lane_with(const RwmArgs &a, const RwmParams &p, hipStream_t s, const LaneLayout &L,
                         const RwmLauncher &launch);
int launch_rwm_pair_with(const RwmArgs &a, const RwmParams &p, hipStream_t s, int n_packed,
                         const std::vector<double> &packed_tparams, const RwmLauncher &launch);
#endif

} // namespace mcmc
