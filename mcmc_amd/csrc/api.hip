// extern "C" entry points of libmcmc_amd.so (declared in include/mcmc_amd.h).
//
// Host side of the engine: argument validation (the reference has none, SURVEY.md
// section 8b), the once-per-call shared precompute, kernel selection, and the host-pointer
// single-chain wrappers that give the Rcpp shim the reference's own argument / return layout.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <thread>
#include <vector>

#include "accept_window.h"
#include "common.h"
#include "host_linalg.h"
#include "targets.h"

namespace mcmc {

static thread_local std::string g_last_error;

int set_error(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

// RAII device buffer for the host-pointer wrappers
struct DevBuf {
    void *p = nullptr;
    ~DevBuf()
    {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8); }
    template <class T>
    T *as() const { return (T *)p; }
};

static int build_rwm_params(const mcmc_target *target, int d, const double *h_init, const double *h_covmat,
                            RwmParams &p)
{
    p.d = d;
    p.target_kind = target->kind;
    if (target->kind < 0 || target->kind > MCMC_TARGET_USER)
        return set_error(MCMC_E_BAD_ARG, "unknown target kind %d", target->kind);
    int np = target_param_count(target->kind, d);
    if (target->kind == MCMC_TARGET_USER) {
        np = user_target_param_count(target->user, d);
        if (np < 0) return np;
        p.user = target->user;
    }
    if (target->n_params != np || (np > 0 && !target->h_params))
        return set_error(MCMC_E_BAD_ARG, "target kind %d with d=%d needs %d parameters, got %d", target->kind, d, np,
                         target->n_params);
    p.tparams.assign(target->h_params, target->h_params + np);
    p.init.assign(d, 0.0);
    if (h_init) p.init.assign(h_init, h_init + d);
    // covmat: arma::chol reads the upper triangle only; det/inv in mvrnorm_pdf read all of it, so a
    // non-symmetric covmat has no single meaning here and is rejected (the reference has no check).
    double scale = 0.0;
    for (int e = 0; e < d * d; ++e) {
        if (!std::isfinite(h_covmat[e])) return set_error(MCMC_E_BAD_ARG, "covmat has a non-finite entry");
        scale = std::fmax(scale, std::fabs(h_covmat[e]));
    }
    for (int r = 0; r < d; ++r)
        for (int c = r + 1; c < d; ++c)
            if (std::fabs(cmr(h_covmat, r, c, d) - cmr(h_covmat, c, r, d)) > 1e-12 * scale)
                return set_error(MCMC_E_BAD_ARG, "covmat is not symmetric at (%d,%d)", r, c);
    p.cov_packed.resize((size_t)d * (d + 1) / 2);
    for (int c = 0; c < d; ++c)
        for (int r = 0; r <= c; ++r) p.cov_packed[tri(r, c)] = cmr(h_covmat, r, c, d);
    if (!chol_upper_packed(d, h_covmat, p.R0_packed, p.det0))
        return set_error(MCMC_E_NOT_PD, "covmat is not positive definite (arma::chol would throw at iteration 1)");
    return MCMC_OK;
}

static int build_gibbs_params(int d, const double *h_mu, const double *h_sigma, const double *h_lower,
                              const double *h_upper, const double *h_init, GibbsParams &p)
{
    p.d = d;
    p.mu.assign(h_mu, h_mu + d);
    p.lower.assign(h_lower, h_lower + d);
    p.upper.assign(h_upper, h_upper + d);
    p.init.assign(d, 0.0);
    if (h_init) p.init.assign(h_init, h_init + d);
    p.sd.resize(d);
    p.P.assign((size_t)d * (d > 1 ? d - 1 : 1), 0.0);
    for (int k = 0; k < d; ++k) {
        if (std::isnan(p.lower[k]) || std::isnan(p.upper[k]) || !(p.lower[k] < p.upper[k]))
            return set_error(MCMC_E_BAD_ARG, "need lower[%d] < upper[%d]", k, k);
        if (!std::isfinite(p.mu[k])) return set_error(MCMC_E_BAD_ARG, "mu[%d] is not finite", k);
    }
    for (int e = 0; e < d * d; ++e)
        if (!std::isfinite(h_sigma[e])) return set_error(MCMC_E_BAD_ARG, "sigma has a non-finite entry");
    // precision matrix H = sigma^-1; P_i = -H[i,-i] / H_ii, sd_i = 1 / sqrt(H_ii)
    std::vector<double> H;
    if (!spd_inverse(d, h_sigma, H)) return set_error(MCMC_E_NOT_PD, "sigma is not positive definite");
    for (int i = 0; i < d; ++i) {
        const double hii = H[(size_t)i * d + i];
        p.sd[i] = 1.0 / std::sqrt(hii);
        for (int k = 0, c = 0; k < d; ++k) {
            if (k == i) continue;
            p.P[(size_t)i * (d - 1) + c++] = -H[(size_t)k * d + i] / hii;
        }
    }
    return MCMC_OK;
}

} // namespace mcmc

using namespace mcmc;

extern "C" int mcmc_amd_abi_version(void) { return MCMC_AMD_ABI_VERSION; }
extern "C" const char *mcmc_amd_last_error(void) { return g_last_error.c_str(); }
extern "C" int mcmc_amd_max_lane_dim(void) { return kMaxLaneDimRwm; }
extern "C" int mcmc_amd_max_dim(void) { return kMaxDimRwm; }

// ============================================================================================
extern "C" int mcmc_adapt_rwm_batch(const mcmc_target *target, const mcmc_rwm_config *cfg, const mcmc_rwm_io *io,
                                    void *stream)
{
    if (!target || !cfg || !io) return set_error(MCMC_E_BAD_ARG, "mcmc_adapt_rwm_batch: NULL argument");
    const int d = cfg->d;
    if (d < 1) return set_error(MCMC_E_BAD_ARG, "d=%d must be >= 1", d);
    if (cfg->n_chains < 1) return set_error(MCMC_E_BAD_ARG, "n_chains=%lld must be >= 1", (long long)cfg->n_chains);
    if (cfg->n_chains > (int64_t)64 * 0x7fffffff) return set_error(MCMC_E_BAD_ARG, "n_chains too large for one launch");
    if (cfg->n_iterations < 1) return set_error(MCMC_E_BAD_ARG, "n_iterations=%d must be >= 1", cfg->n_iterations);
    if (cfg->thin < 1) return set_error(MCMC_E_BAD_ARG, "thin=%d must be >= 1", cfg->thin);
    if (cfg->adapt_size_start < 0 || cfg->adapt_shape_start < 0)
        return set_error(MCMC_E_BAD_ARG, "adapt_size_start / adapt_shape_start must be >= 0");
    if (cfg->acceptance_window < 0) return set_error(MCMC_E_BAD_ARG, "acceptance_window must be >= 0");
    if (cfg->acceptance_window > kMaxAcceptanceWindow)
        return set_error(MCMC_E_UNSUPPORTED, "acceptance_window=%d: the device keeps at most %d outcomes per chain", cfg->acceptance_window,
                         kMaxAcceptanceWindow);
    if (std::isnan(cfg->acceptance_rate_weight) || std::isnan(cfg->adapt_size_cooling) || std::isnan(cfg->max_scaling_sd))
        return set_error(MCMC_E_BAD_ARG, "NaN in the adaptation parameters");
    if (!io->h_covmat) return set_error(MCMC_E_BAD_ARG, "h_covmat is NULL");
    if (!io->h_init_theta && !io->d_init) return set_error(MCMC_E_BAD_ARG, "need h_init_theta or d_init");
    if (cfg->rng_mode == MCMC_RNG_INJECTED) {
        if (!io->d_Z || !io->d_U) return set_error(MCMC_E_BAD_ARG, "MCMC_RNG_INJECTED needs d_Z and d_U");
    } else if (cfg->rng_mode != MCMC_RNG_PHILOX) {
        return set_error(MCMC_E_BAD_ARG, "unknown rng_mode %d", cfg->rng_mode);
    }
    if (d > kMaxDimRwm) return set_error(MCMC_E_UNSUPPORTED, "adaptive RWM: d=%d above the limit %d", d, kMaxDimRwm);
    if (cfg->kernel < MCMC_KERNEL_AUTO || cfg->kernel > MCMC_KERNEL_ROW)
        return set_error(MCMC_E_BAD_ARG, "unknown kernel choice %d", cfg->kernel);
    // the status word is (iteration << 4) | code in 32 bits
    if (cfg->iter_offset < 0 || (int64_t)cfg->iter_offset + cfg->n_iterations >= (1 << 27))
        return set_error(MCMC_E_BAD_ARG, "iter_offset + n_iterations = %lld: must stay below 2^27 (the per-chain status word holds the iteration in 27 bits)",
                         (long long)cfg->iter_offset + cfg->n_iterations);
    if ((cfg->iter_offset > 0) != (io->d_state_in != nullptr))
        return set_error(MCMC_E_BAD_ARG, "resume needs both iter_offset > 0 and d_state_in (a fresh start neither)");
    if (cfg->factor_update != MCMC_FACTOR_EXACT && cfg->factor_update != MCMC_FACTOR_RANK1)
        return set_error(MCMC_E_BAD_ARG, "unknown factor_update %d", cfg->factor_update);
    if (cfg->trace_layout != MCMC_TRACE_SOA && cfg->trace_layout != MCMC_TRACE_CHAIN_MAJOR)
        return set_error(MCMC_E_BAD_ARG, "unknown trace_layout %d", cfg->trace_layout);
    const bool rank1 = cfg->factor_update == MCMC_FACTOR_RANK1, chain_major = cfg->trace_layout == MCMC_TRACE_CHAIN_MAJOR;
    if (rank1) {
        if (cfg->refresh_every < 1) return set_error(MCMC_E_BAD_ARG, "MCMC_FACTOR_RANK1 needs refresh_every >= 1, got %d", cfg->refresh_every);
        if (d <= kMaxLaneDimRwm || (cfg->kernel != MCMC_KERNEL_AUTO && cfg->kernel != MCMC_KERNEL_BLOCK))
            return set_error(MCMC_E_UNSUPPORTED, "MCMC_FACTOR_RANK1 is served by the per-workgroup kernel only (d > %d, kernel AUTO or BLOCK)", kMaxLaneDimRwm);
        if (target->kind != MCMC_TARGET_GAUSSIAN_PREC)
            return set_error(MCMC_E_UNSUPPORTED, "MCMC_FACTOR_RANK1 serves the Gaussian target only");
        if (cfg->rng_mode == MCMC_RNG_INJECTED || io->d_sigma_trace || io->d_accepted || io->d_log_acceptance || io->d_log_A || io->d_scaling_sd ||
            io->d_scaling_multiplier || io->d_rate_trace)
            return set_error(MCMC_E_UNSUPPORTED, "MCMC_FACTOR_RANK1 is not a parity mode: no injected streams, sigma_empirical trace or per-iteration diagnostics");
    }
    if (chain_major && (d <= kMaxLaneDimRwm || cfg->kernel == MCMC_KERNEL_LANE || cfg->kernel == MCMC_KERNEL_PAIR))
        return set_error(MCMC_E_UNSUPPORTED, "MCMC_TRACE_CHAIN_MAJOR is served by the per-row and per-workgroup kernels only (d > %d)", kMaxLaneDimRwm);
    if (cfg->kernel == MCMC_KERNEL_LANE && d > kMaxLaneDimRwm)
        return set_error(MCMC_E_UNSUPPORTED, "one-chain-per-lane kernel: d=%d above its limit %d", d, kMaxLaneDimRwm);
    if (cfg->kernel == MCMC_KERNEL_ROW && d > kMaxRowDimRwm)
        return set_error(MCMC_E_UNSUPPORTED, "one-chain-per-row kernel: d=%d above its limit %d", d, kMaxRowDimRwm);
    bool use_block = cfg->kernel == MCMC_KERNEL_BLOCK || (cfg->kernel == MCMC_KERNEL_AUTO && d > kMaxLaneDimRwm);
    if (use_block && target->kind != MCMC_TARGET_GAUSSIAN_PREC && target->kind != MCMC_TARGET_USER)
        return set_error(MCMC_E_UNSUPPORTED, "the per-workgroup kernel (d > %d) serves the Gaussian target and user targets only", kMaxLaneDimRwm);

    RwmParams p;
    int rc = build_rwm_params(target, d, io->h_init_theta, io->h_covmat, p);
    if (rc != MCMC_OK) return rc;

    hipStream_t s = (hipStream_t)stream;
    RwmArgs a{};
    a.n_chains = cfg->n_chains;
    a.chain_offset = cfg->chain_offset;
    a.d = d;
    a.n_iterations = cfg->n_iterations;
    a.adapt_size_start = cfg->adapt_size_start;
    a.adapt_shape_start = cfg->adapt_shape_start;
    a.acceptance_window = cfg->acceptance_window;
    a.thin = cfg->thin;
    a.acceptance_rate_weight = cfg->acceptance_rate_weight;
    a.max_scaling_sd = cfg->max_scaling_sd;
    a.shape_scaling_sd = 2.38 / std::sqrt((double)d);
    a.twopi_pow_d = std::pow(2 * M_PI, d);
    a.seed = cfg->seed;
    a.d_init = io->d_init;
    const bool inject = cfg->rng_mode == MCMC_RNG_INJECTED;
    a.d_Z = inject ? io->d_Z : nullptr;
    a.d_U = inject ? io->d_U : nullptr;
    a.d_trace = io->d_trace;
    a.d_sigma_trace = io->d_sigma_trace;
    a.d_final_theta = io->d_final_theta;
    a.d_final_mean = io->d_final_mean;
    a.d_final_cov = io->d_final_cov;
    a.d_acceptance_rate = io->d_acceptance_rate;
    a.d_status = io->d_status;
    a.d_shape_iterations = io->d_shape_iterations;
    a.d_accepted = io->d_accepted;
    a.d_log_acceptance = io->d_log_acceptance;
    a.d_log_A = io->d_log_A;
    a.d_scaling_sd = io->d_scaling_sd;
    a.d_scaling_multiplier = io->d_scaling_multiplier;
    a.d_rate_trace = io->d_rate_trace;
    a.iter_offset = cfg->iter_offset;
    a.d_state_in = io->d_state_in;
    a.d_state_out = io->d_state_out;
    a.factor_rank1 = rank1 ? 1 : 0;
    a.refresh_every = rank1 ? cfg->refresh_every : 1;
    a.trace_chain_major = chain_major ? 1 : 0;
    a.n_kept = (int64_t)(cfg->iter_offset + cfg->n_iterations) / cfg->thin - (int64_t)cfg->iter_offset / cfg->thin;

    // per-call device scratch: the cooling table and, for acceptance_window > 64, the outcome ring; both are released
    // on every path out of this function
    double *d_cool = nullptr;
    unsigned long long *d_ring = nullptr;
    auto release = [&](int code) {
        for (void *ptr : {(void *)d_cool, (void *)d_ring}) {
            if (!ptr) continue;
            hipError_t e = hipFreeAsync(ptr, s);
            if (code == MCMC_OK && e != hipSuccess) code = set_error(MCMC_E_HIP, "hipFreeAsync failed: %s", hipGetErrorString(e));
        }
        return code;
    };
    // pow(adapt_size_cooling, i - adapt_size_start) for i = iter_offset + (0..n) (adaptMCMC_source.cpp:123), host libm
    if (cfg->adapt_size_start != 0) {
        // The table stops where the power has reached its limit for good (0.99^k is exactly 0 from k = 74 100 on; a
        // cooling of 1 never moves): the kernels clamp the index, so a call of 2^27 iterations no longer uploads 1 GB.
        std::vector<double> cool;
        cool.reserve(std::min<size_t>((size_t)cfg->n_iterations + 1, 1 << 17));
        for (int i = 0; i <= cfg->n_iterations; ++i) {
            const int k = cfg->iter_offset + i - cfg->adapt_size_start;
            const double v = std::pow(cfg->adapt_size_cooling, k);
            const bool settled = !cool.empty() && k > 0 && v == cool.back() &&
                                 (v == 0.0 || std::isinf(v) || cfg->adapt_size_cooling == 1.0 || cfg->adapt_size_cooling == 0.0);
            if (settled) break; // every later entry equals this one
            cool.push_back(v);
        }
        a.cool_last = (int32_t)cool.size() - 1;
        hipError_t e = hipMallocAsync((void **)&d_cool, cool.size() * sizeof(double), s);
        if (e != hipSuccess) return release(set_error(MCMC_E_HIP, "cooling table allocation failed: %s", hipGetErrorString(e)));
        if (const int ru = upload_block(d_cool, cool.data(), cool.size(), s); ru != MCMC_OK) return release(ru);
        a.d_cool_pow = d_cool;
    }
    a.win_words = cfg->acceptance_rate_weight == 0 ? accept_window_words(cfg->acceptance_window) : 0;
    if (a.win_words > 0) {
        const size_t bytes = (size_t)a.win_words * (size_t)cfg->n_chains * sizeof(unsigned long long);
        hipError_t e = hipMallocAsync((void **)&d_ring, bytes, s);
        if (e == hipSuccess) e = hipMemsetAsync(d_ring, 0, bytes, s);
        if (e != hipSuccess)
            return release(set_error(MCMC_E_HIP, "acceptance_window=%d: no room for the outcome rings (%zu bytes): %s", cfg->acceptance_window,
                                     bytes, hipGetErrorString(e)));
        a.d_win_ring = d_ring;
    }
    // AUTO picks between the two one-chain-per-lane kernels by what the call asks for (diagnostics, injected streams and
    // windows above 64 outcomes need the one-wave kernel).  A checkpoint is in its writer's representation, so a resumed
    // AUTO run goes back to the kernel named in the state, or is refused here if that kernel cannot serve the request.
    // 12 < d <= 64: one chain per 16 / 32 / 64 lanes where that kernel serves the call (Philox, no diagnostics), else per workgroup
    bool use_row = cfg->kernel == MCMC_KERNEL_ROW || (cfg->kernel == MCMC_KERNEL_AUTO && use_block && rwm_row_serves(a, target->kind));
    if (cfg->kernel == MCMC_KERNEL_AUTO && use_block && a.d_state_in) { // a checkpoint goes back to its writer
        double tag = 0.0;
        const RwmStateLayout SL = rwm_state_layout(d);
        hipError_t e = hipMemcpyAsync(&tag, a.d_state_in + (int64_t)(SL.S + ST_TAG) * cfg->n_chains, sizeof(double), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return release(set_error(MCMC_E_HIP, "reading the checkpoint's kernel tag failed: %s", hipGetErrorString(e)));
        if (tag == (double)MCMC_KERNEL_ROW && !use_row)
            return release(set_error(MCMC_E_UNSUPPORTED,
                                     "this checkpoint was written by the one-chain-per-row kernel, which cannot serve this call (per-iteration "
                                     "diagnostics, injected streams or sigma_empirical trace): resume with the outputs of the run that wrote "
                                     "it, or run both parts with kernel = MCMC_KERNEL_BLOCK"));
        if (tag == (double)MCMC_KERNEL_BLOCK) use_row = false;
    }
    if (use_row) use_block = false;
    bool use_pair = !use_row && (cfg->kernel == MCMC_KERNEL_PAIR || (cfg->kernel == MCMC_KERNEL_AUTO && rwm_pair_serves(a)));
    if (cfg->kernel == MCMC_KERNEL_AUTO && !use_block && !use_row && a.d_state_in) {
        double tag = 0.0;
        const RwmStateLayout SL = rwm_state_layout(d);
        hipError_t e = hipMemcpyAsync(&tag, a.d_state_in + (int64_t)(SL.S + ST_TAG) * cfg->n_chains, sizeof(double), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return release(set_error(MCMC_E_HIP, "reading the checkpoint's kernel tag failed: %s", hipGetErrorString(e)));
        if (tag == (double)MCMC_KERNEL_PAIR && !use_pair)
            return release(set_error(MCMC_E_UNSUPPORTED,
                                     "this checkpoint was written by the paired-wave kernel, which cannot serve this call (per-iteration "
                                     "diagnostics, injected streams, sigma_empirical trace or acceptance_window > 64): resume with the outputs "
                                     "of the run that wrote it, or run both parts with kernel = MCMC_KERNEL_LANE"));
        if (tag == (double)MCMC_KERNEL_LANE) use_pair = false;
    }
    if (use_row)
        rc = launch_rwm_row(a, p, s);
    else if (use_block)
        rc = launch_rwm_block(a, p, s);
    else if (use_pair)
        rc = launch_rwm_pair(a, p, s);
    else
        rc = launch_rwm_lane(a, p, s);
    return release(rc);
}

extern "C" int64_t mcmc_rwm_state_doubles(int32_t d)
{
    if (d < 1 || d > kMaxDimRwm) return set_error(MCMC_E_BAD_ARG, "mcmc_rwm_state_doubles: d=%d outside 1..%d", d, kMaxDimRwm);
    return rwm_state_layout(d).TOTAL;
}

extern "C" int64_t mcmc_rwm_state_doubles_window(int32_t d, int32_t acceptance_window)
{
    if (d < 1 || d > kMaxDimRwm) return set_error(MCMC_E_BAD_ARG, "mcmc_rwm_state_doubles_window: d=%d outside 1..%d", d, kMaxDimRwm);
    if (acceptance_window < 0 || acceptance_window > kMaxAcceptanceWindow)
        return set_error(MCMC_E_BAD_ARG, "mcmc_rwm_state_doubles_window: acceptance_window=%d outside 0..%d", acceptance_window, kMaxAcceptanceWindow);
    return rwm_state_layout(d).TOTAL + 2 * accept_window_words(acceptance_window);
}

extern "C" int mcmc_rwm_streams(uint64_t seed, int64_t chain_offset, int64_t n_chains, int32_t n_iterations,
                                int32_t d, double *d_Z, double *d_U, void *stream)
{
    if (n_chains < 1 || n_iterations < 1 || d < 1 || (!d_Z && !d_U))
        return set_error(MCMC_E_BAD_ARG, "mcmc_rwm_streams: bad argument");
    return launch_rwm_streams(seed, chain_offset, n_chains, n_iterations, d, d_Z, d_U, (hipStream_t)stream);
}

extern "C" int mcmc_init_normal(uint64_t seed, int64_t chain_offset, int64_t n_chains, int32_t d, const double *h_mean,
                                double sd, double *d_out, void *stream)
{
    if (n_chains < 1 || d < 1 || !h_mean || !d_out) return set_error(MCMC_E_BAD_ARG, "mcmc_init_normal: bad argument");
    return launch_init_normal(seed, chain_offset, n_chains, d, h_mean, sd, d_out, (hipStream_t)stream);
}

// --------------------------------------------------------------------------------------------
static int run_single_rwm(const mcmc_target *target, int32_t d, const double *h_init_theta, const double *h_covmat,
                          int32_t n, int32_t adapt_size_start, double weight, int32_t window, int32_t adapt_shape_start,
                          int32_t info, double seedMH, double cooling, double max_sd, const double *h_Z,
                          const double *h_U, double *h_theta_trace, double *h_sigma_empirical, double *h_acc,
                          double *h_info_log, int32_t *fail_iter)
{
    if (fail_iter) *fail_iter = 0;
    if (!target || !h_init_theta || !h_covmat || !h_theta_trace)
        return set_error(MCMC_E_BAD_ARG, "NULL argument");
    if (d < 1 || n < 1) return set_error(MCMC_E_BAD_ARG, "d and n_iterations must be >= 1");
    if (info <= 0) return set_error(MCMC_E_BAD_ARG, "info=%d must be > 0 (the reference computes i %% info)", info);
    if ((h_Z == nullptr) != (h_U == nullptr)) return set_error(MCMC_E_BAD_ARG, "give both h_Z and h_U, or neither");
    if (!std::isfinite(seedMH) || seedMH < 0) return set_error(MCMC_E_BAD_ARG, "seedMH must be a finite non-negative number");

    const size_t nd = (size_t)n * d, ndd = (size_t)n * d * d;
    DevBuf bZ, bU, bT, bS, bAcc, bSt, bSd, bMul, bRate;
    MCMC_HIP_CHECK(bT.alloc(nd * sizeof(double)));
    MCMC_HIP_CHECK(bAcc.alloc(sizeof(double)));
    MCMC_HIP_CHECK(bSt.alloc(sizeof(int32_t)));
    if (h_sigma_empirical) MCMC_HIP_CHECK(bS.alloc(ndd * sizeof(double)));
    if (h_Z) {
        MCMC_HIP_CHECK(bZ.alloc(nd * sizeof(double)));
        MCMC_HIP_CHECK(bU.alloc((size_t)n * sizeof(double)));
        MCMC_HIP_CHECK(hipMemcpy(bZ.p, h_Z, nd * sizeof(double), hipMemcpyHostToDevice));
        MCMC_HIP_CHECK(hipMemcpy(bU.p, h_U, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    }
    if (h_info_log) {
        MCMC_HIP_CHECK(bSd.alloc((size_t)n * sizeof(double)));
        MCMC_HIP_CHECK(bMul.alloc((size_t)n * sizeof(double)));
        MCMC_HIP_CHECK(bRate.alloc((size_t)n * sizeof(double)));
    }
    mcmc_rwm_config cfg{};
    cfg.d = d;
    cfg.n_iterations = n;
    cfg.n_chains = 1;
    cfg.chain_offset = 0;
    cfg.adapt_size_start = adapt_size_start;
    cfg.adapt_shape_start = adapt_shape_start;
    cfg.acceptance_window = window;
    cfg.thin = 1;
    cfg.acceptance_rate_weight = weight;
    cfg.adapt_size_cooling = cooling;
    cfg.max_scaling_sd = max_sd;
    cfg.seed = (uint64_t)seedMH; // std::mt19937 engine(seedMH): double -> unsigned (adaptMCMC_source.cpp:78)
    cfg.rng_mode = h_Z ? MCMC_RNG_INJECTED : MCMC_RNG_PHILOX;
    mcmc_rwm_io io{};
    io.h_init_theta = h_init_theta;
    io.h_covmat = h_covmat;
    io.d_Z = bZ.as<double>();
    io.d_U = bU.as<double>();
    io.d_trace = bT.as<double>();
    io.d_sigma_trace = h_sigma_empirical ? bS.as<double>() : nullptr;
    io.d_acceptance_rate = bAcc.as<double>();
    io.d_status = bSt.as<int32_t>();
    if (h_info_log) {
        io.d_scaling_sd = bSd.as<double>();
        io.d_scaling_multiplier = bMul.as<double>();
        io.d_rate_trace = bRate.as<double>();
    }
    int rc = mcmc_adapt_rwm_batch(target, &cfg, &io, nullptr);
    if (rc != MCMC_OK) return rc;
    MCMC_HIP_CHECK(hipDeviceSynchronize());

    // [n][d][1] -> n x d column-major (arma::mat theta_trace, :92,185)
    std::vector<double> tmp(nd);
    MCMC_HIP_CHECK(hipMemcpy(tmp.data(), bT.p, nd * sizeof(double), hipMemcpyDeviceToHost));
    for (int t = 0; t < n; ++t)
        for (int k = 0; k < d; ++k) h_theta_trace[(size_t)k * n + t] = tmp[(size_t)t * d + k];
    // [n][d*d][1] is already the arma::cube layout d x d x n (:93,214)
    if (h_sigma_empirical) MCMC_HIP_CHECK(hipMemcpy(h_sigma_empirical, bS.p, ndd * sizeof(double), hipMemcpyDeviceToHost));
    if (h_acc) MCMC_HIP_CHECK(hipMemcpy(h_acc, bAcc.p, sizeof(double), hipMemcpyDeviceToHost));
    if (h_info_log) {
        // what the reference prints at :142-146: the rate BEFORE iteration i's update, scaling_sd and
        // scaling_multiplier after iteration i's adaptation block
        std::vector<double> sd(n), mul(n), rate(n);
        MCMC_HIP_CHECK(hipMemcpy(sd.data(), bSd.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
        MCMC_HIP_CHECK(hipMemcpy(mul.data(), bMul.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
        MCMC_HIP_CHECK(hipMemcpy(rate.data(), bRate.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
        for (int i = info; i <= n; i += info) {
            double *row = h_info_log + (size_t)3 * (i / info - 1);
            row[0] = i >= 2 ? rate[i - 2] : 0.0;
            row[1] = sd[i - 1];
            row[2] = mul[i - 1];
        }
    }
    int32_t st = 0;
    MCMC_HIP_CHECK(hipMemcpy(&st, bSt.p, sizeof st, hipMemcpyDeviceToHost));
    if (st != MCMC_CHAIN_OK) {
        if (fail_iter) *fail_iter = MCMC_STATUS_ITER(st);
        return set_error(MCMC_E_CHAIN_FAILED, "chol(covmat_proposal) failed at iteration %d", MCMC_STATUS_ITER(st));
    }
    return MCMC_OK;
}

extern "C" int mcmc_adaptMCMC(const mcmc_target *target, int32_t d, const double *h_init_theta, const double *h_covmat,
                              int32_t n_iterations, int32_t adapt_size_start, double acceptance_rate_weight,
                              int32_t acceptance_window, int32_t adapt_shape_start, int32_t info, double seedMH,
                              double adapt_size_cooling, double max_scaling_sd, const double *h_Z, const double *h_U,
                              double *h_theta_trace, double *h_sigma_empirical, double *h_acceptance_rate,
                              double *h_info_log, int32_t *fail_iter)
{
    return run_single_rwm(target, d, h_init_theta, h_covmat, n_iterations, adapt_size_start, acceptance_rate_weight,
                          acceptance_window, adapt_shape_start, info, seedMH, adapt_size_cooling, max_scaling_sd, h_Z,
                          h_U, h_theta_trace, h_sigma_empirical, h_acceptance_rate, h_info_log, fail_iter);
}

extern "C" int mcmc_rwMCMC(const mcmc_target *target, int32_t d, const double *h_init_theta, const double *h_covmat,
                           int32_t n_iterations, int32_t info, double seedMH, const double *h_Z, const double *h_U,
                           double *h_theta_trace, double *h_acceptance_rate, int32_t *fail_iter)
{
    return run_single_rwm(target, d, h_init_theta, h_covmat, n_iterations, 0, 0.0, 0, 0, info, seedMH, 0.99, 50.0, h_Z,
                          h_U, h_theta_trace, nullptr, h_acceptance_rate, nullptr, fail_iter);
}

// --------------------------------------------------------------------------------------------
// Many chains, host pointers: what an R caller reaches the batch kernels through.  Replaces the reference's only
// multi-chain mechanism, one adaptMCMC call per worker process under mclapply_RNG (MCMCutilty.R:103-149), and hands
// back per-chain matrices in the layout melt_trace (MCMCutilty.R:19-35) reads.  Chains are processed in slabs whose
// device trace stays below kChainsSlabBytes, so the host-side result may exceed the GPU's memory.
static constexpr size_t kChainsSlabBytes = (size_t)4 << 30; // per device buffer; there are two (sampler layout, caller's layout)

// device [n_kept][d][nc] (chain fastest) -> per chain an n_kept x d column-major matrix, chain c at out + c*n_kept*d:
// for every coordinate k a transpose of the (t, c) plane, 32 x 32 tiles through LDS (both sides whole 256-byte runs).
// Done on the DEVICE, so that one copy lands the result in the caller's buffer: the first version copied the slab to a
// host staging vector and untangled it there on one core, which made the call 500 x the kernel time (2.2 s for the 5.2 GB
// of 65 536 chains x 1 000 iterations at d = 10, 4 ms of them sampling; measured).
__global__ __launch_bounds__(256) void untangle_trace_kernel(const double *__restrict__ in, double *__restrict__ out, int64_t n_kept,
                                                             int d, int64_t nc)
{
    __shared__ double tile[32][33];
    const int k = blockIdx.z;
    const int64_t c0 = (int64_t)blockIdx.x * 32, t0 = (int64_t)blockIdx.y * 32;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5; // 32 x 8 threads, four rows each
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t t = t0 + ly + 8 * r, c = c0 + lx;
        if (t < n_kept && c < nc) tile[ly + 8 * r][lx] = in[((size_t)t * d + k) * nc + c];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t c = c0 + ly + 8 * r, t = t0 + lx;
        if (t < n_kept && c < nc) out[(size_t)c * n_kept * d + (size_t)k * n_kept + t] = tile[lx][ly + 8 * r];
    }
}
// Device -> pageable host memory.  hipMemcpy stages such a copy through the runtime's own pinned buffer and empties that
// with one thread: 9.5 GB/s measured, which made the copy 98 % of a list-of-chains call.  Here: two pinned chunks; while
// the copy engine fills one, up to eight host threads empty the other into the caller's buffer (the copy engine then waits
// for the host: 30-40 GB/s; 65 536 chains x 1 000 iterations at d = 10 end to end 550 -> ~200 ms).  Copies below four
// chunks go the plain way (pinning the chunks costs more than it saves).
static int download_pipelined(void *h_dst, const void *d_src, size_t bytes)
{
    constexpr size_t kChunk = (size_t)64 << 20;
    if (bytes < 4 * kChunk) {
        MCMC_HIP_CHECK(hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost));
        return MCMC_OK;
    }
    struct Stage { // released on every path out
        char *buf[2] = {nullptr, nullptr};
        hipStream_t stream = nullptr;
        hipEvent_t done[2] = {nullptr, nullptr};
        ~Stage()
        {
            for (int q = 0; q < 2; ++q) {
                if (done[q]) (void)hipEventDestroy(done[q]);
                if (buf[q]) (void)hipHostFree(buf[q]);
            }
            if (stream) (void)hipStreamDestroy(stream);
        }
    } st;
    MCMC_HIP_CHECK(hipDeviceSynchronize()); // the producer ran on the null stream; the copies run on a stream of their own
    MCMC_HIP_CHECK(hipStreamCreateWithFlags(&st.stream, hipStreamNonBlocking));
    for (int q = 0; q < 2; ++q) {
        MCMC_HIP_CHECK(hipHostMalloc((void **)&st.buf[q], kChunk, hipHostMallocDefault));
        MCMC_HIP_CHECK(hipEventCreateWithFlags(&st.done[q], hipEventDisableTiming));
    }
    const unsigned hw = std::thread::hardware_concurrency();
    int nthreads = (int)std::min(8u, std::max(1u, hw)); // eight threads empty a chunk at 30-40 GB/s (measured; sixteen are no faster)
    if (const char *e = std::getenv("OMP_NUM_THREADS")) { // the job's CPU share, where the scheduler states one
        const int lim = std::atoi(e);
        if (lim >= 1 && lim < nthreads) nthreads = lim;
    }
    auto drain = [&](const char *src, char *dst, size_t n) { // n bytes out of a pinned chunk, split over the threads
        const size_t per = ((n + nthreads - 1) / nthreads + 4095) & ~(size_t)4095;
        std::vector<std::thread> pool;
        for (int t = 1; t < nthreads; ++t) {
            const size_t lo = std::min(n, (size_t)t * per), hi = std::min(n, lo + per);
            if (lo < hi) pool.emplace_back([=] { std::memcpy(dst + lo, src + lo, hi - lo); });
        }
        std::memcpy(dst, src, std::min(n, per));
        for (auto &th : pool) th.join();
    };
    const size_t nchunks = (bytes + kChunk - 1) / kChunk;
    for (size_t c = 0; c <= nchunks; ++c) {
        if (c < nchunks) { // chunk c into buffer c & 1 (emptied two rounds ago)
            const size_t off = c * kChunk, n = std::min(kChunk, bytes - off);
            MCMC_HIP_CHECK(hipMemcpyAsync(st.buf[c & 1], (const char *)d_src + off, n, hipMemcpyDeviceToHost, st.stream));
            MCMC_HIP_CHECK(hipEventRecord(st.done[c & 1], st.stream));
        }
        if (c > 0) { // chunk c - 1 out of its buffer while chunk c is on its way
            const size_t off = (c - 1) * kChunk, n = std::min(kChunk, bytes - off);
            MCMC_HIP_CHECK(hipEventSynchronize(st.done[(c - 1) & 1]));
            drain(st.buf[(c - 1) & 1], (char *)h_dst + off, n);
        }
    }
    return MCMC_OK;
}

static int untangle_trace(const double *d_slab, double *d_out, int64_t n_kept, int d, int64_t nc, double *h_out)
{
    if (n_kept <= 0 || nc <= 0) return MCMC_OK;
    const dim3 grid((unsigned)((nc + 31) / 32), (unsigned)((n_kept + 31) / 32), (unsigned)d);
    if (grid.y > 65535u) return set_error(MCMC_E_UNSUPPORTED, "list-of-chains entry: more than 2 097 120 kept rows per chain");
    hipLaunchKernelGGL(untangle_trace_kernel, grid, dim3(256), 0, nullptr, d_slab, d_out, n_kept, d, nc);
    MCMC_HIP_CHECK(hipGetLastError());
    return download_pipelined(h_out, d_out, (size_t)nc * n_kept * d * sizeof(double));
}

static int slab_chains(int64_t n_chains, int64_t n_kept, int d)
{
    const size_t per_chain = (size_t)(n_kept > 0 ? n_kept : 1) * d * sizeof(double);
    int64_t m = (int64_t)(kChainsSlabBytes / per_chain);
    m = m < 64 ? 64 : (m / 64) * 64; // whole wavefronts of chains
    return (int)(m < n_chains ? m : n_chains);
}

extern "C" int mcmc_adaptMCMC_chains(const mcmc_target *target, int32_t d, int64_t n_chains, const double *h_init_theta,
                                     int32_t init_per_chain, const double *h_covmat, int32_t n_iterations,
                                     int32_t adapt_size_start, double acceptance_rate_weight, int32_t acceptance_window,
                                     int32_t adapt_shape_start, int32_t info, double seedMH, double adapt_size_cooling,
                                     double max_scaling_sd, int32_t thin, double *h_theta_trace, double *h_acceptance_rate,
                                     double *h_covmat_empirical, int32_t *h_status)
{
    if (!target || !h_init_theta || !h_covmat) return set_error(MCMC_E_BAD_ARG, "mcmc_adaptMCMC_chains: NULL argument");
    if (d < 1 || n_iterations < 1 || n_chains < 1 || thin < 1)
        return set_error(MCMC_E_BAD_ARG, "mcmc_adaptMCMC_chains: d, n_chains, n_iterations and thin must be >= 1");
    if (n_chains > 0x7fffffff) return set_error(MCMC_E_BAD_ARG, "mcmc_adaptMCMC_chains: n_chains too large");
    if (info <= 0) return set_error(MCMC_E_BAD_ARG, "info=%d must be > 0 (the reference computes i %% info)", info);
    if (!std::isfinite(seedMH) || seedMH < 0) return set_error(MCMC_E_BAD_ARG, "seedMH must be a finite non-negative number");
    const int64_t n_kept = n_iterations / thin;
    const int slab = slab_chains(n_chains, h_theta_trace ? n_kept : 0, d);
    DevBuf bInit, bT, bT2, bAcc, bSt, bCov;
    std::vector<double> hInit, hCov;
    if (init_per_chain) {
        MCMC_HIP_CHECK(bInit.alloc((size_t)slab * d * sizeof(double)));
        hInit.resize((size_t)slab * d);
    }
    if (h_theta_trace) {
        MCMC_HIP_CHECK(bT.alloc((size_t)slab * n_kept * d * sizeof(double)));
        MCMC_HIP_CHECK(bT2.alloc((size_t)slab * n_kept * d * sizeof(double)));
    }
    if (h_acceptance_rate) MCMC_HIP_CHECK(bAcc.alloc((size_t)slab * sizeof(double)));
    if (h_covmat_empirical) {
        MCMC_HIP_CHECK(bCov.alloc((size_t)slab * d * d * sizeof(double)));
        hCov.resize((size_t)slab * d * d);
    }
    MCMC_HIP_CHECK(bSt.alloc((size_t)slab * sizeof(int32_t)));
    std::vector<int32_t> hSt(slab);
    for (int64_t c0 = 0; c0 < n_chains; c0 += slab) {
        const int64_t nc = n_chains - c0 < slab ? n_chains - c0 : slab;
        mcmc_rwm_config cfg{};
        cfg.d = d;
        cfg.n_iterations = n_iterations;
        cfg.n_chains = nc;
        cfg.chain_offset = c0; // Philox counters carry the global chain id: results do not depend on the slab size
        cfg.adapt_size_start = adapt_size_start;
        cfg.adapt_shape_start = adapt_shape_start;
        cfg.acceptance_window = acceptance_window;
        cfg.thin = thin;
        cfg.acceptance_rate_weight = acceptance_rate_weight;
        cfg.adapt_size_cooling = adapt_size_cooling;
        cfg.max_scaling_sd = max_scaling_sd;
        cfg.seed = (uint64_t)seedMH;
        cfg.rng_mode = MCMC_RNG_PHILOX;
        mcmc_rwm_io io{};
        io.h_covmat = h_covmat;
        if (init_per_chain) { // d x n_chains column-major (one column per chain) -> [d][nc]
            for (int k = 0; k < d; ++k)
                for (int64_t c = 0; c < nc; ++c) hInit[(size_t)k * nc + c] = h_init_theta[(size_t)(c0 + c) * d + k];
            MCMC_HIP_CHECK(hipMemcpy(bInit.p, hInit.data(), (size_t)nc * d * sizeof(double), hipMemcpyHostToDevice));
            io.d_init = bInit.as<double>();
        } else {
            io.h_init_theta = h_init_theta;
        }
        io.d_trace = h_theta_trace ? bT.as<double>() : nullptr;
        io.d_acceptance_rate = h_acceptance_rate ? bAcc.as<double>() : nullptr;
        io.d_final_cov = h_covmat_empirical ? bCov.as<double>() : nullptr;
        io.d_status = bSt.as<int32_t>();
        int rc = mcmc_adapt_rwm_batch(target, &cfg, &io, nullptr);
        if (rc != MCMC_OK) return rc;
        MCMC_HIP_CHECK(hipDeviceSynchronize());
        if (h_theta_trace) {
            rc = untangle_trace(bT.as<double>(), bT2.as<double>(), n_kept, d, nc, h_theta_trace + (size_t)c0 * n_kept * d);
            if (rc != MCMC_OK) return rc;
        }
        if (h_acceptance_rate)
            MCMC_HIP_CHECK(hipMemcpy(h_acceptance_rate + c0, bAcc.p, (size_t)nc * sizeof(double), hipMemcpyDeviceToHost));
        if (h_covmat_empirical) { // [d*d][nc] -> per chain a d x d column-major matrix
            MCMC_HIP_CHECK(hipMemcpy(hCov.data(), bCov.p, (size_t)nc * d * d * sizeof(double), hipMemcpyDeviceToHost));
            for (int e = 0; e < d * d; ++e)
                for (int64_t c = 0; c < nc; ++c) h_covmat_empirical[(size_t)(c0 + c) * d * d + e] = hCov[(size_t)e * nc + c];
        }
        MCMC_HIP_CHECK(hipMemcpy(hSt.data(), bSt.p, (size_t)nc * sizeof(int32_t), hipMemcpyDeviceToHost));
        if (h_status) std::memcpy(h_status + c0, hSt.data(), (size_t)nc * sizeof(int32_t));
    }
    return MCMC_OK;
}

extern "C" int mcmc_rtmvnorm_gibbs_chains(int32_t n, int32_t d, int64_t n_chains, const double *h_mu, const double *h_sigma,
                                          const double *h_lower, const double *h_upper, const double *h_init_state,
                                          int32_t init_per_chain, uint64_t seed, int32_t thin, int32_t tail_safe,
                                          double *h_trace, int32_t *h_status)
{
    if (n < 1 || d < 1 || n_chains < 1 || thin < 1 || !h_mu || !h_sigma || !h_lower || !h_upper || !h_init_state || !h_trace)
        return set_error(MCMC_E_BAD_ARG, "mcmc_rtmvnorm_gibbs_chains: bad argument");
    if (n_chains > 0x7fffffff) return set_error(MCMC_E_BAD_ARG, "mcmc_rtmvnorm_gibbs_chains: n_chains too large");
    const int64_t n_kept = n / thin;
    const int slab = slab_chains(n_chains, n_kept, d);
    DevBuf bInit, bT, bT2, bSt;
    std::vector<double> hInit;
    std::vector<int32_t> hSt(slab);
    if (init_per_chain) {
        MCMC_HIP_CHECK(bInit.alloc((size_t)slab * d * sizeof(double)));
        hInit.resize((size_t)slab * d);
    }
    MCMC_HIP_CHECK(bT.alloc((size_t)slab * n_kept * d * sizeof(double)));
    MCMC_HIP_CHECK(bT2.alloc((size_t)slab * n_kept * d * sizeof(double)));
    MCMC_HIP_CHECK(bSt.alloc((size_t)slab * sizeof(int32_t)));
    for (int64_t c0 = 0; c0 < n_chains; c0 += slab) {
        const int64_t nc = n_chains - c0 < slab ? n_chains - c0 : slab;
        mcmc_gibbs_config cfg{};
        cfg.d = d;
        cfg.n_sweeps = n;
        cfg.n_chains = nc;
        cfg.chain_offset = c0;
        cfg.thin = thin;
        cfg.seed = seed;
        cfg.rng_mode = MCMC_RNG_PHILOX;
        cfg.tail_safe = tail_safe;
        mcmc_gibbs_io io{};
        io.h_mu = h_mu;
        io.h_sigma = h_sigma;
        io.h_lower = h_lower;
        io.h_upper = h_upper;
        if (init_per_chain) {
            for (int k = 0; k < d; ++k)
                for (int64_t c = 0; c < nc; ++c) hInit[(size_t)k * nc + c] = h_init_state[(size_t)(c0 + c) * d + k];
            MCMC_HIP_CHECK(hipMemcpy(bInit.p, hInit.data(), (size_t)nc * d * sizeof(double), hipMemcpyHostToDevice));
            io.d_init = bInit.as<double>();
        } else {
            io.h_init_state = h_init_state;
        }
        io.d_trace = bT.as<double>();
        io.d_status = bSt.as<int32_t>();
        int rc = mcmc_tmvn_gibbs_batch(&cfg, &io, nullptr);
        if (rc != MCMC_OK) return rc;
        MCMC_HIP_CHECK(hipDeviceSynchronize());
        rc = untangle_trace(bT.as<double>(), bT2.as<double>(), n_kept, d, nc, h_trace + (size_t)c0 * n_kept * d);
        if (rc != MCMC_OK) return rc;
        if (h_status) {
            MCMC_HIP_CHECK(hipMemcpy(hSt.data(), bSt.p, (size_t)nc * sizeof(int32_t), hipMemcpyDeviceToHost));
            std::memcpy(h_status + c0, hSt.data(), (size_t)nc * sizeof(int32_t));
        }
    }
    return MCMC_OK;
}

// ============================================================================================
extern "C" int mcmc_tmvn_precompute(int32_t d, const double *h_sigma, double *h_P, double *h_sd)
{
    if (d < 1 || !h_sigma || !h_P || !h_sd) return set_error(MCMC_E_BAD_ARG, "mcmc_tmvn_precompute: bad argument");
    std::vector<double> zeros(d, 0.0), lo(d, -1.0), hi(d, 1.0);
    GibbsParams p;
    int rc = build_gibbs_params(d, zeros.data(), h_sigma, lo.data(), hi.data(), nullptr, p);
    if (rc != MCMC_OK) return rc;
    std::memcpy(h_sd, p.sd.data(), sizeof(double) * d);
    if (d > 1) std::memcpy(h_P, p.P.data(), sizeof(double) * (size_t)d * (d - 1));
    return MCMC_OK;
}

static int tmvn_gibbs_batch_impl(const mcmc_gibbs_config *cfg, const mcmc_gibbs_io *io, void *stream, bool noalloc)
{
    if (!cfg || !io) return set_error(MCMC_E_BAD_ARG, "mcmc_tmvn_gibbs_batch: NULL argument");
    const int d = cfg->d;
    if (d < 1 || d > kMaxDimGibbs) return set_error(d < 1 ? MCMC_E_BAD_ARG : MCMC_E_UNSUPPORTED, "d=%d outside 1..%d", d, kMaxDimGibbs);
    if (cfg->n_chains < 1) return set_error(MCMC_E_BAD_ARG, "n_chains must be >= 1");
    if (cfg->n_chains > (int64_t)64 * 0x7fffffff) return set_error(MCMC_E_BAD_ARG, "n_chains too large for one launch");
    if (cfg->n_sweeps < 1) return set_error(MCMC_E_BAD_ARG, "n_sweeps must be >= 1");
    if (cfg->thin < 1) return set_error(MCMC_E_BAD_ARG, "thin must be >= 1");
    if (cfg->burn < 0) return set_error(MCMC_E_BAD_ARG, "burn must be >= 0");
    if (cfg->sweep_offset < 0 || (int64_t)cfg->sweep_offset + cfg->n_sweeps >= (1 << 27))
        return set_error(MCMC_E_BAD_ARG, "sweep_offset + n_sweeps = %lld: must stay below 2^27 (the per-chain status word holds the sweep in 27 bits)",
                         (long long)cfg->sweep_offset + cfg->n_sweeps);
    if (!io->h_mu || !io->h_sigma || !io->h_lower || !io->h_upper) return set_error(MCMC_E_BAD_ARG, "mu/sigma/lower/upper missing");
    if (!io->h_init_state && !io->d_init) return set_error(MCMC_E_BAD_ARG, "need h_init_state or d_init");
    if ((io->d_mean == nullptr) != (io->d_m2 == nullptr)) return set_error(MCMC_E_BAD_ARG, "d_mean and d_m2 come as a pair");
    if (cfg->rng_mode == MCMC_RNG_INJECTED) {
        if (!io->d_U) return set_error(MCMC_E_BAD_ARG, "MCMC_RNG_INJECTED needs d_U");
    } else if (cfg->rng_mode != MCMC_RNG_PHILOX) {
        return set_error(MCMC_E_BAD_ARG, "unknown rng_mode %d", cfg->rng_mode);
    }
    GibbsParams p;
    int rc = build_gibbs_params(d, io->h_mu, io->h_sigma, io->h_lower, io->h_upper, io->h_init_state, p);
    if (rc != MCMC_OK) return rc;
    GibbsArgs a{};
    a.n_chains = cfg->n_chains;
    a.chain_offset = cfg->chain_offset;
    a.d = d;
    a.n_sweeps = cfg->n_sweeps;
    a.thin = cfg->thin;
    a.burn = cfg->burn;
    a.sweep_offset = cfg->sweep_offset;
    a.seed = cfg->seed;
    a.d_init = io->d_init;
    a.d_U = cfg->rng_mode == MCMC_RNG_INJECTED ? io->d_U : nullptr;
    a.d_trace = io->d_trace;
    a.d_final = io->d_final_state;
    a.d_mean = io->d_mean;
    a.d_m2 = io->d_m2;
    a.d_status = io->d_status;
    if (noalloc) return launch_gibbs_noalloc(a, p, cfg->tail_safe != 0, (hipStream_t)stream);
    return launch_gibbs(a, p, cfg->tail_safe != 0, (hipStream_t)stream);
}

extern "C" int mcmc_tmvn_gibbs_batch(const mcmc_gibbs_config *cfg, const mcmc_gibbs_io *io, void *stream)
{
    return tmvn_gibbs_batch_impl(cfg, io, stream, false);
}

extern "C" int mcmc_tmvn_gibbs_batch_noalloc(const mcmc_gibbs_config *cfg, const mcmc_gibbs_io *io, void *stream)
{
    return tmvn_gibbs_batch_impl(cfg, io, stream, true);
}

extern "C" int mcmc_tmvn_gibbs_noalloc_max_dim(void) { return gibbs_noalloc_max_dim(); }

extern "C" int mcmc_gibbs_streams(uint64_t seed, int64_t chain_offset, int64_t n_chains, int32_t n_sweeps, int32_t d,
                                  double *d_U, void *stream)
{
    if (n_chains < 1 || n_sweeps < 1 || d < 1 || !d_U) return set_error(MCMC_E_BAD_ARG, "mcmc_gibbs_streams: bad argument");
    return launch_gibbs_streams(seed, chain_offset, n_chains, n_sweeps, d, d_U, (hipStream_t)stream);
}

extern "C" int mcmc_rtmvnorm_gibbs(int32_t n, int32_t d, const double *h_mu, const double *h_sigma,
                                   const double *h_lower, const double *h_upper, const double *h_init_state,
                                   const double *h_U, uint64_t seed, double *h_trace)
{
    if (n < 1 || d < 1 || !h_mu || !h_sigma || !h_lower || !h_upper || !h_init_state || !h_trace)
        return set_error(MCMC_E_BAD_ARG, "mcmc_rtmvnorm_gibbs: bad argument");
    const size_t nd = (size_t)n * d;
    DevBuf bU, bT;
    MCMC_HIP_CHECK(bT.alloc(nd * sizeof(double)));
    if (h_U) {
        MCMC_HIP_CHECK(bU.alloc(nd * sizeof(double)));
        MCMC_HIP_CHECK(hipMemcpy(bU.p, h_U, nd * sizeof(double), hipMemcpyHostToDevice));
    }
    mcmc_gibbs_config cfg{};
    cfg.d = d;
    cfg.n_sweeps = n;
    cfg.n_chains = 1;
    cfg.thin = 1;
    cfg.seed = seed;
    cfg.rng_mode = h_U ? MCMC_RNG_INJECTED : MCMC_RNG_PHILOX;
    mcmc_gibbs_io io{};
    io.h_mu = h_mu;
    io.h_sigma = h_sigma;
    io.h_lower = h_lower;
    io.h_upper = h_upper;
    io.h_init_state = h_init_state;
    io.d_U = bU.as<double>();
    io.d_trace = bT.as<double>();
    int rc = mcmc_tmvn_gibbs_batch(&cfg, &io, nullptr);
    if (rc != MCMC_OK) return rc;
    MCMC_HIP_CHECK(hipDeviceSynchronize());
    std::vector<double> tmp(nd);
    MCMC_HIP_CHECK(hipMemcpy(tmp.data(), bT.p, nd * sizeof(double), hipMemcpyDeviceToHost));
    for (int t = 0; t < n; ++t)
        for (int k = 0; k < d; ++k) h_trace[(size_t)k * n + t] = tmp[(size_t)t * d + k]; // n x d column-major (:94,137)
    return MCMC_OK;
}

extern "C" int mcmc_rtnorm_batch(int64_t n, double mu, double sigma, double a, double b, const double *d_F, uint64_t seed,
                                 int64_t offset, int32_t tail_safe, double *d_out, void *stream)
{
    if (n < 0 || !d_out) return set_error(MCMC_E_BAD_ARG, "mcmc_rtnorm_batch: bad argument");
    return launch_rtnorm(n, mu, sigma, a, b, d_F, seed, offset, tail_safe != 0, d_out, (hipStream_t)stream);
}

extern "C" int mcmc_rtnorm_gibbs(int32_t n, double mu, double sigma, double a, double b, const double *h_F, uint64_t seed,
                                 double *h_out)
{
    if (n < 0 || (n > 0 && !h_out)) return set_error(MCMC_E_BAD_ARG, "mcmc_rtnorm_gibbs: bad argument");
    if (n == 0) return MCMC_OK;
    DevBuf bF, bO;
    MCMC_HIP_CHECK(bO.alloc((size_t)n * sizeof(double)));
    if (h_F) {
        MCMC_HIP_CHECK(bF.alloc((size_t)n * sizeof(double)));
        MCMC_HIP_CHECK(hipMemcpy(bF.p, h_F, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    }
    int rc = launch_rtnorm(n, mu, sigma, a, b, bF.as<double>(), seed, 0, false, bO.as<double>(), nullptr);
    if (rc != MCMC_OK) return rc;
    MCMC_HIP_CHECK(hipDeviceSynchronize());
    MCMC_HIP_CHECK(hipMemcpy(h_out, bO.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    return MCMC_OK;
}

// ============================================================================================
// index-shedding helpers: pure host copies (on the device they are folded into indexing)
extern "C" int mcmc_sub1(int32_t d, const double *h_x, int32_t i, double *h_out)
{
    if (d < 1 || !h_x || !h_out || i < 0 || i >= d) return set_error(MCMC_E_BAD_ARG, "mcmc_sub1: index %d outside 0..%d", i, d - 1);
    const int m = d - 1;
    for (int c = 0, oc = 0; c < d; ++c) {
        if (c == i) continue;
        for (int r = 0, orow = 0; r < d; ++r) {
            if (r == i) continue;
            h_out[(size_t)oc * m + orow++] = h_x[(size_t)c * d + r];
        }
        ++oc;
    }
    return MCMC_OK;
}

extern "C" int mcmc_sub2(int32_t nr, int32_t nc, const double *h_x, int32_t a, int32_t b, double *h_out)
{
    if (nr < 1 || nc < 1 || !h_x || !h_out || a < 0 || a >= nr || b < 0 || b >= nc)
        return set_error(MCMC_E_BAD_ARG, "mcmc_sub2: index outside the matrix");
    for (int c = 0, oc = 0; c < nc; ++c) {
        if (c == b) continue;
        h_out[oc++] = h_x[(size_t)c * nr + a];
    }
    return MCMC_OK;
}

extern "C" int mcmc_negSubCol(int32_t n, const double *h_x, int32_t i, double *h_out)
{
    if (n < 1 || !h_x || !h_out || i < 0 || i >= n) return set_error(MCMC_E_BAD_ARG, "negSub: index %d outside 0..%d", i, n - 1);
    for (int k = 0, o = 0; k < n; ++k)
        if (k != i) h_out[o++] = h_x[k];
    return MCMC_OK;
}

extern "C" int mcmc_negSubRow(int32_t n, const double *h_x, int32_t i, double *h_out)
{
    return mcmc_negSubCol(n, h_x, i, h_out);
}
