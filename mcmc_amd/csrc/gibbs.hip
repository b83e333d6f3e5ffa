// Truncated multivariate normal Gibbs sweep on gfx950.
//
// Replaces the sweep loop of rtmvnorm_gibbs (tmvrnormGibbs.cpp:115-139) and the draw loop of
// rtnorm_gibbs (:58-67).  One chain per lane: a wavefront advances 64 independent chains in
// lockstep, so the 2 x pnorm + qnorm work of every coordinate (about 70 % of a sweep) runs on
// all 64 lanes; a wavefront-per-chain layout would leave 63 of 64 lanes idle there.
// Coordinates inside a sweep are visited in order (Gauss-Seidel), as the reference does.
//
// Data layout
//   * chain state: the centred vector x - mu in LDS as [k][lane] (d * 512 B per wavefront;
//     ds_read_b64 / ds_write_b64 with bank = lane, conflict free).  d = 10 -> 5 KiB per wave, so
//     eight waves per SIMD fit in the 160 KiB of a CU; d = 128 -> 64 KiB, two waves per CU.
//   * shared parameters (mu, sd, bounds, regression rows P_i): one block in HBM, read with
//     wave-uniform addresses, i.e. through the scalar cache into SGPRs (s_load_dwordx*), and
//     used directly as the scalar operand of v_fma_f64.  The offset `z` below is an opaque zero
//     refreshed every sweep: it keeps the compiler from hoisting ~200 parameter SGPRs out of
//     the loop and then spilling them to VGPR lanes (v_readlane in the hot loop).
//   * samples: one coalesced store per coordinate and kept sweep, trace[(t*d+k)*n + c] --
//     512 contiguous bytes per wavefront.
#include "common.h"
#include "normal_cdf.h"
#include "philox.h"

namespace mcmc {

// parameter block: mu[d] sd[d] lower[d] upper[d] init[d] inv_sd[d] P[d*(d-1)]
template <bool TAIL_SAFE, bool MOMENTS>
__global__ __launch_bounds__(64, 3) void gibbs_kernel(const GibbsArgs a, const double *__restrict__ prm)
{
    extern __shared__ double xc[]; // centred state, [k][lane]
    const int lane = threadIdx.x;
    const int64_t c = (int64_t)blockIdx.x * 64 + lane;
    const int64_t n = a.n_chains;
    const bool live = c < n;
    const int64_t cc = live ? c : n - 1; // tail lanes shadow the last chain and store nothing
    const int d = a.d;
    const uint64_t gchain = (uint64_t)(a.chain_offset + cc);
    const PhiloxKey key(a.seed);
    const bool inject = a.d_U != nullptr;

    for (int k = 0; k < d; ++k) {
        const double x0 = a.d_init ? a.d_init[(int64_t)k * n + cc] : prm[4 * d + k];
        xc[k * 64 + lane] = x0 - prm[k];
        if (MOMENTS && live) {
            a.d_mean[(int64_t)k * n + c] = 0.0;
            a.d_m2[(int64_t)k * n + c] = 0.0;
        }
    }
    int32_t status = MCMC_CHAIN_OK;

    for (int j = 1; j <= a.n_sweeps; ++j) {
        int z = 0;
        asm volatile("" : "+s"(z)); // opaque zero: parameter loads stay inside the sweep loop
        const double *__restrict__ mu = prm + z;
        const double *__restrict__ sd = mu + d;
        const double *__restrict__ lower = mu + 2 * d;
        const double *__restrict__ upper = mu + 3 * d;
        const double *__restrict__ inv_sd = mu + 5 * d;
        const double *__restrict__ P = mu + 6 * d;
        const bool keep = a.d_trace && (j % a.thin) == 0;
        const int64_t t = j / a.thin - 1;
        const bool mom = MOMENTS && j > a.burn;
        const double inv = mom ? 1.0 / (double)(j - a.burn) : 0.0;
        double ua = 0.0, ub = 0.0;
        for (int i = 0; i < d; ++i) {
            double u;
            if (inject) {
                u = a.d_U[((int64_t)(j - 1) * d + i) * n + cc];
            } else {
                if ((i & 1) == 0) philox_uniform2(key, gchain, (uint32_t)j, TAG_GIBBS, (uint32_t)(i >> 1), ua, ub);
                u = (i & 1) ? ub : ua;
            }
            // conditional mean mu_i + P_i (x[-i] - mu[-i])   (tmvrnormGibbs.cpp:124-127)
            const double *__restrict__ Pi = P + (int64_t)i * (d - 1);
            double s = 0.0;
            for (int k = 0; k < i; ++k) s += Pi[k] * xc[k * 64 + lane];
            for (int k = i + 1; k < d; ++k) s += Pi[k - 1] * xc[k * 64 + lane];
            const double mu_i = mu[i] + s;
            const double xi = rtnorm_draw_std<TAIL_SAFE>(u, mu_i, sd[i], inv_sd[i], lower[i], upper[i]);
            xc[i * 64 + lane] = xi - mu[i];
            if (!isfinite(xi) && status == MCMC_CHAIN_OK) status = (j << 4) | MCMC_CHAIN_NONFINITE_STATE;
            if (live) {
                if (keep) a.d_trace[(t * d + i) * n + c] = xi;
                if (j == a.n_sweeps && a.d_final) a.d_final[(int64_t)i * n + c] = xi;
                if (mom) {
                    const int64_t o = (int64_t)i * n + c;
                    double mean = a.d_mean[o];
                    const double delta = xi - mean;
                    mean += delta * inv;
                    a.d_mean[o] = mean;
                    a.d_m2[o] += delta * (xi - mean);
                }
            }
        }
    }
    if (live && a.d_status) a.d_status[c] = status;
}

// uniforms a Philox Gibbs run draws, [n_sweeps][d][n_chains]
__global__ __launch_bounds__(256) void gibbs_streams_kernel(uint64_t seed, int64_t chain_offset, int64_t n,
                                                            int32_t n_sweeps, int32_t d, double *U)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    const PhiloxKey key(seed);
    const uint64_t gchain = (uint64_t)(chain_offset + c);
    for (int j = 1; j <= n_sweeps; ++j) {
        for (int b = 0; 2 * b < d; ++b) {
            double u0, u1;
            philox_uniform2(key, gchain, (uint32_t)j, TAG_GIBBS, (uint32_t)b, u0, u1);
            U[((int64_t)(j - 1) * d + 2 * b) * n + c] = u0;
            if (2 * b + 1 < d) U[((int64_t)(j - 1) * d + 2 * b + 1) * n + c] = u1;
        }
    }
}

// rtnorm_gibbs (tmvrnormGibbs.cpp:49-70): Fa, Fb once (:55-56), then one quantile per draw.
template <bool TAIL_SAFE>
__global__ __launch_bounds__(256) void rtnorm_kernel(int64_t n, double mu, double sigma, double lo, double hi,
                                                     const double *F, uint64_t seed, int64_t offset, double *out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double u;
    if (F) {
        u = F[i];
    } else {
        const uint64_t g = (uint64_t)(offset + i);
        double u0, u1;
        philox_uniform2(PhiloxKey(seed), g >> 1, 0u, TAG_RTNORM, 0u, u0, u1);
        u = (g & 1) ? u1 : u0;
    }
    if (TAIL_SAFE) {
        out[i] = rtnorm_draw<true>(u, mu, sigma, lo, hi);
    } else {
        const double Fa = pnorm(lo, mu, sigma);
        const double Fb = pnorm(hi, mu, sigma);
        const double p_i = u * (Fb - Fa) + Fa;
        out[i] = mu + sigma * qnorm_std(p_i);
    }
}

// ------------------------------------------------------------------------------------------
int launch_gibbs(const GibbsArgs &a, const GibbsParams &p, bool tail_safe, hipStream_t s)
{
    const int d = a.d;
    if (d < 1 || d > kMaxDimGibbs) return set_error(MCMC_E_UNSUPPORTED, "gibbs: d=%d outside 1..%d", d, kMaxDimGibbs);
    std::vector<double> blob;
    blob.reserve((size_t)6 * d + (size_t)d * (d - 1));
    blob.insert(blob.end(), p.mu.begin(), p.mu.end());
    blob.insert(blob.end(), p.sd.begin(), p.sd.end());
    blob.insert(blob.end(), p.lower.begin(), p.lower.end());
    blob.insert(blob.end(), p.upper.begin(), p.upper.end());
    blob.insert(blob.end(), p.init.begin(), p.init.end());
    for (int k = 0; k < d; ++k) blob.push_back(1.0 / p.sd[k]);
    blob.insert(blob.end(), p.P.begin(), p.P.begin() + (size_t)d * (d - 1));
    double *d_prm = nullptr;
    MCMC_HIP_CHECK(hipMallocAsync((void **)&d_prm, blob.size() * sizeof(double), s));
    MCMC_HIP_CHECK(hipMemcpyAsync(d_prm, blob.data(), blob.size() * sizeof(double), hipMemcpyHostToDevice, s));
    MCMC_HIP_CHECK(hipStreamSynchronize(s)); // blob is a local: do not let the copy outlive it
    const size_t lds = (size_t)d * 64 * sizeof(double);
    const int64_t grid = (a.n_chains + 63) / 64;
    const bool moments = a.d_mean != nullptr;
    auto go = [&](auto kern) -> int {
        MCMC_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64), lds, s, a, (const double *)d_prm);
        MCMC_HIP_CHECK(hipGetLastError());
        return MCMC_OK;
    };
    int rc;
    if (tail_safe)
        rc = moments ? go(gibbs_kernel<true, true>) : go(gibbs_kernel<true, false>);
    else
        rc = moments ? go(gibbs_kernel<false, true>) : go(gibbs_kernel<false, false>);
    hipError_t e = hipFreeAsync(d_prm, s);
    if (rc == MCMC_OK && e != hipSuccess) rc = set_error(MCMC_E_HIP, "hipFreeAsync failed: %s", hipGetErrorString(e));
    return rc;
}

int launch_gibbs_streams(uint64_t seed, int64_t chain_offset, int64_t n_chains, int32_t n_sweeps, int32_t d,
                         double *d_U, hipStream_t s)
{
    const int block = 256;
    const int64_t grid = (n_chains + block - 1) / block;
    hipLaunchKernelGGL(gibbs_streams_kernel, dim3((unsigned)grid), dim3(block), 0, s, seed, chain_offset, n_chains,
                       n_sweeps, d, d_U);
    MCMC_HIP_CHECK(hipGetLastError());
    return MCMC_OK;
}

int launch_rtnorm(int64_t n, double mu, double sigma, double a, double b, const double *d_F, uint64_t seed,
                  int64_t offset, bool tail_safe, double *d_out, hipStream_t s)
{
    if (n == 0) return MCMC_OK;
    const int block = 256;
    const int64_t grid = (n + block - 1) / block;
    if (tail_safe)
        hipLaunchKernelGGL((rtnorm_kernel<true>), dim3((unsigned)grid), dim3(block), 0, s, n, mu, sigma, a, b, d_F, seed,
                           offset, d_out);
    else
        hipLaunchKernelGGL((rtnorm_kernel<false>), dim3((unsigned)grid), dim3(block), 0, s, n, mu, sigma, a, b, d_F, seed,
                           offset, d_out);
    MCMC_HIP_CHECK(hipGetLastError());
    return MCMC_OK;
}

} // namespace mcmc
