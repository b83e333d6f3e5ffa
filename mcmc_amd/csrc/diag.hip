// Cross-chain convergence diagnostics on gfx950 (no counterpart in the reference: coda is
// loaded at MCMCutilty.R:11 but never called).  These reduce a [n_kept][d][n_chains] trace to
// the few numbers an RCCL all-reduce carries between GPUs.
#include "common.h"

namespace mcmc {

// one thread per (chain, coordinate): Welford over kept draws t >= burn
__global__ __launch_bounds__(256) void chain_moments_kernel(const double *__restrict__ trace, int32_t n_kept,
                                                            int32_t burn, int32_t d, int64_t n, double *mean_out,
                                                            double *var_out)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int k = blockIdx.y;
    if (c >= n) return;
    double mean = 0.0, m2 = 0.0;
    int cnt = 0;
    for (int t = burn; t < n_kept; ++t) {
        const double x = trace[((int64_t)t * d + k) * n + c];
        ++cnt;
        const double delta = x - mean;
        mean += delta / cnt;
        m2 += delta * (x - mean);
    }
    mean_out[(int64_t)k * n + c] = mean;
    var_out[(int64_t)k * n + c] = cnt > 1 ? m2 / (cnt - 1) : 0.0;
}

// one 1024-thread block per coordinate, fixed-order tree: deterministic sums over chains
__global__ __launch_bounds__(1024) void rhat_partial_kernel(const double *__restrict__ mean, const double *__restrict__ var,
                                                            int32_t d, int64_t n, double *partial)
{
    __shared__ double sm[3][1024];
    const int k = blockIdx.x;
    const int tid = threadIdx.x;
    double a = 0.0, b = 0.0, s = 0.0;
    for (int64_t c = tid; c < n; c += 1024) {
        const double m = mean[(int64_t)k * n + c];
        a += m;
        b += m * m;
        s += var[(int64_t)k * n + c];
    }
    sm[0][tid] = a;
    sm[1][tid] = b;
    sm[2][tid] = s;
    __syncthreads();
    for (int off = 512; off > 0; off >>= 1) {
        if (tid < off) {
            sm[0][tid] += sm[0][tid + off];
            sm[1][tid] += sm[1][tid + off];
            sm[2][tid] += sm[2][tid + off];
        }
        __syncthreads();
    }
    if (tid == 0) {
        partial[k] = sm[0][0];
        partial[d + k] = sm[1][0];
        partial[2 * d + k] = sm[2][0];
        if (k == 0) partial[3 * d] = (double)n;
    }
}

// pooled covariance over chains: one 1024-thread block per matrix element (r, cl), fixed-order tree.
// partial = [sum_c C_c (d*d, column-major), sum_c m_c m_c^T (d*d), sum_c m_c (d), chain count]
__global__ __launch_bounds__(1024) void pooled_cov_partial_kernel(const double *__restrict__ mean, const double *__restrict__ cov,
                                                                  int32_t d, int64_t n, double *partial)
{
    __shared__ double sm[3][1024];
    const int e = blockIdx.x, r = e % d, cl = e / d; // element (r, cl) of the column-major d x d matrices
    const int tid = threadIdx.x;
    double sc = 0.0, smm = 0.0, s1 = 0.0;
    for (int64_t c = tid; c < n; c += 1024) {
        const double mr = mean[(int64_t)r * n + c];
        sc += cov[(int64_t)e * n + c];
        smm += mr * mean[(int64_t)cl * n + c];
        s1 += mr;
    }
    sm[0][tid] = sc;
    sm[1][tid] = smm;
    sm[2][tid] = s1;
    __syncthreads();
    for (int off = 512; off > 0; off >>= 1) {
        if (tid < off) {
            sm[0][tid] += sm[0][tid + off];
            sm[1][tid] += sm[1][tid + off];
            sm[2][tid] += sm[2][tid + off];
        }
        __syncthreads();
    }
    if (tid == 0) {
        partial[e] = sm[0][0];
        partial[d * d + e] = sm[1][0];
        if (cl == 0) partial[2 * d * d + r] = sm[2][0];
        if (e == 0) partial[2 * d * d + d] = (double)n;
    }
}

// Geyer initial-positive-sequence ESS, one thread per (chain, coordinate).  `capped_out` (may be NULL) receives 1
// where the sum of autocorrelation pairs was still positive at max_lag, i.e. where the truncation -- not Geyer's
// stopping rule -- ended it and the ESS is therefore OVER-stated (tau under-stated); 0 otherwise.
__global__ __launch_bounds__(256) void chain_ess_kernel(const double *__restrict__ trace, int32_t n_kept, int32_t burn,
                                                        int32_t d, int64_t n, int32_t max_lag, double *ess_out,
                                                        int32_t *capped_out)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int k = blockIdx.y;
    if (c >= n) return;
    const int N = n_kept - burn;
    const double *x = trace + ((int64_t)burn * d + k) * n + c;
    const int64_t stride = (int64_t)d * n;
    double mean = 0.0, lo = x[0], hi = x[0];
    for (int t = 0; t < N; ++t) {
        const double xt = x[t * stride];
        mean += xt;
        lo = fmin(lo, xt);
        hi = fmax(hi, xt);
    }
    mean /= N;
    double g0 = 0.0;
    for (int t = 0; t < N; ++t) {
        const double v = x[t * stride] - mean;
        g0 += v * v;
    }
    // a chain that never moved carries no information: ESS 0.  Decided on the values themselves, because the rounded
    // mean of N equal numbers need not equal them, which leaves a g0 of rounding noise (and an ESS of ~1) behind.
    if (!(hi > lo)) g0 = 0.0;
    double ess = 0.0;
    int32_t capped = 0;
    if (g0 > 0.0 && N > 1) {
        double tau = -1.0;
        const int kmax = max_lag < N - 1 ? max_lag : N - 1;
        capped = kmax < N - 1; // cleared when Geyer's rule stops the sum before the cap does
        // Eight lags per pass over the series: w[] is a sliding window of the eight values v[t+lag0 .. t+lag0+7] (zero
        // past the end, so a term beyond the series adds nothing), rotated statically by unrolling t by eight, which
        // makes it two loads per t for eight autocovariances.  Each sum runs over t in ascending order.
        auto val = [&](int idx) { return idx < N ? x[idx * stride] - mean : 0.0; };
        bool done = false;
        for (int lag0 = 0; lag0 <= kmax && !done; lag0 += 8) {
            double g[8], w[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                g[j] = 0.0;
                w[j] = val(lag0 + j);
            }
            for (int t0 = 0; t0 + lag0 < N; t0 += 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const double a = val(t0 + u);
#pragma unroll
                    for (int j = 0; j < 8; ++j) g[j] += a * w[(u + j) & 7];
                    w[u] = val(t0 + u + lag0 + 8);
                }
            }
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                const int lag = lag0 + j;
                if (done || lag > kmax) continue;
                const double gb = lag + 1 > kmax ? 0.0 : g[j + 1];
                const double pair = (g[j] + gb) / g0;
                if (pair <= 0.0) {
                    capped = 0;
                    done = true;
                } else {
                    tau += 2.0 * pair;
                }
            }
        }
        ess = tau > 0.0 ? N / tau : (double)N;
    }
    ess_out[(int64_t)k * n + c] = ess;
    if (capped_out) capped_out[(int64_t)k * n + c] = capped;
}

} // namespace mcmc

using namespace mcmc;

extern "C" int mcmc_chain_moments(const double *d_trace, int32_t n_kept, int32_t burn_kept, int32_t d, int64_t n_chains,
                                  double *d_mean, double *d_var, void *stream)
{
    if (!d_trace || !d_mean || !d_var || d < 1 || n_chains < 1 || burn_kept < 0 || n_kept <= burn_kept)
        return set_error(MCMC_E_BAD_ARG, "mcmc_chain_moments: bad argument");
    dim3 grid((unsigned)((n_chains + 255) / 256), (unsigned)d);
    hipLaunchKernelGGL(chain_moments_kernel, grid, dim3(256), 0, (hipStream_t)stream, d_trace, n_kept, burn_kept, d,
                       n_chains, d_mean, d_var);
    MCMC_HIP_CHECK(hipGetLastError());
    return MCMC_OK;
}

extern "C" int mcmc_rhat_partial(const double *d_mean, const double *d_var, int32_t d, int64_t n_chains,
                                 double *d_partial, void *stream)
{
    if (!d_mean || !d_var || !d_partial || d < 1 || n_chains < 1)
        return set_error(MCMC_E_BAD_ARG, "mcmc_rhat_partial: bad argument");
    hipLaunchKernelGGL(rhat_partial_kernel, dim3((unsigned)d), dim3(1024), 0, (hipStream_t)stream, d_mean, d_var, d,
                       n_chains, d_partial);
    MCMC_HIP_CHECK(hipGetLastError());
    return MCMC_OK;
}

extern "C" int mcmc_pooled_cov_partial(const double *d_mean, const double *d_cov, int32_t d, int64_t n_chains,
                                       double *d_partial, void *stream)
{
    if (!d_mean || !d_cov || !d_partial || d < 1 || n_chains < 1)
        return set_error(MCMC_E_BAD_ARG, "mcmc_pooled_cov_partial: bad argument");
    hipLaunchKernelGGL(pooled_cov_partial_kernel, dim3((unsigned)(d * d)), dim3(1024), 0, (hipStream_t)stream, d_mean, d_cov,
                       d, n_chains, d_partial);
    MCMC_HIP_CHECK(hipGetLastError());
    return MCMC_OK;
}

extern "C" int mcmc_chain_ess_capped(const double *d_trace, int32_t n_kept, int32_t burn_kept, int32_t d, int64_t n_chains,
                                     int32_t max_lag, double *d_ess, int32_t *d_capped, void *stream)
{
    if (!d_trace || !d_ess || d < 1 || n_chains < 1 || burn_kept < 0 || n_kept <= burn_kept || max_lag < 1)
        return set_error(MCMC_E_BAD_ARG, "mcmc_chain_ess: bad argument");
    dim3 grid((unsigned)((n_chains + 63) / 64), (unsigned)d);
    hipLaunchKernelGGL(chain_ess_kernel, grid, dim3(64), 0, (hipStream_t)stream, d_trace, n_kept, burn_kept, d,
                       n_chains, max_lag, d_ess, d_capped);
    MCMC_HIP_CHECK(hipGetLastError());
    return MCMC_OK;
}

extern "C" int mcmc_chain_ess(const double *d_trace, int32_t n_kept, int32_t burn_kept, int32_t d, int64_t n_chains,
                              int32_t max_lag, double *d_ess, void *stream)
{
    return mcmc_chain_ess_capped(d_trace, n_kept, burn_kept, d, n_chains, max_lag, d_ess, nullptr, stream);
}
