// the last two pieces of the reference outside its two samplers.
//
//   * mvrGaussian_pdf / tmvrGaussian_pdf (deprecated/truncGaussian.cpp:11-33, :38-78): the multivariate normal density
//     and its truncated form, evaluated at many points per launch (one point per lane, coalesced [d][n] input);
//     the normaliser of the truncated form is mvtnorm::pmvnorm in the reference (:68-75), a third-party randomized
//     integral -- here the caller passes it, or obtains it from mcmc_pmvnorm (Genz' separation-of-variables estimator
//     on the device, Philox points; same estimand, NOT the same random numbers: parity unpinned);
//   * random_walk2d / random_walk3d / random_walkGraph (randomWalkers.cpp:9-100): integer lattice and graph walks, many
//     walkers per launch, uniforms injected in the reference's draw order or from Philox.
#include <algorithm>
#include <numeric>

#include "common.h"
#include "host_linalg.h"
#include "normal_cdf.h"
#include "philox.h"

namespace mcmc {

constexpr uint32_t TAG_WALK = 5, TAG_PMVNORM = 6;
constexpr int kMaxDimPmvnorm = 32;

// prm: mu[d], precision H[d*d] column-major, rootTerm, then (TRUNC) lower[d], upper[d], cdf
template <bool TRUNC>
__global__ __launch_bounds__(256) void mvn_pdf_kernel(int64_t n, int d, const double *__restrict__ x, const double *__restrict__ prm,
                                                       int do_log, double *__restrict__ out)
{
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= n) return;
    const double *mu = prm, *H = prm + d, *tail = prm + d + (size_t)d * d;
    // (-0.5 v^T) Sigma^-1 v as the reference forms it (:21): row vector times matrix, then times v
    double q = 0.0;
    bool inside = true;
    for (int j = 0; j < d; ++j) {
        double s = 0.0;
        for (int i = 0; i < d; ++i) s += (x[(int64_t)i * n + c] - mu[i]) * H[(size_t)j * d + i];
        const double xj = x[(int64_t)j * n + c];
        q += (-0.5 * s) * (xj - mu[j]);
        if (TRUNC) inside = inside && (xj >= tail[1 + j]) && (xj <= tail[1 + d + j]); // :45-51
    }
    const double dens = tail[0] * exp(q); // rootTerm * expTerm (:28)
    if (TRUNC)
        out[c] = inside ? log(dens) / tail[1 + 2 * d] : 0.0; // :61-63, :73-76 (the LOG density over cdf: do_log defaults to true at :73)
    else
        out[c] = do_log ? log(dens) : dens;
}

// lattice walk: out[i][k][w], row 0 the origin; step i >= 1 draws one uniform (randomWalkers.cpp:13,45)
__global__ __launch_bounds__(256) void lattice_walk_kernel(int dim, int n, int64_t nw, int64_t offset, const double *__restrict__ U,
                                                            uint64_t seed, double *__restrict__ out)
{
    const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (w >= nw) return;
    const PhiloxKey key(seed);
    double pos[3] = {0.0, 0.0, 0.0};
    for (int k = 0; k < dim; ++k) out[(int64_t)k * nw + w] = 0.0;
    double ua = 0.0, ub = 0.0;
    for (int i = 1; i < n; ++i) {
        double u;
        if (U) {
            u = U[(int64_t)(i - 1) * nw + w];
        } else {
            if ((i - 1) % 2 == 0) philox_uniform2(key, (uint64_t)(offset + w), (uint32_t)((i - 1) >> 1), TAG_WALK, 0u, ua, ub);
            u = ((i - 1) & 1) ? ub : ua;
        }
        const double r_num = 1 + floor(u * (2 * dim));
        for (int k = 0; k < dim; ++k) {
            if (r_num == 2 * k + 1) pos[k] += 1;
            if (r_num == 2 * k + 2) pos[k] -= 1;
            out[((int64_t)i * dim + k) * nw + w] = pos[k];
        }
    }
}

// graph walk: cum[pos][j] = running sum of row pos' normalised probabilities sorted in descending order, perm[pos][j] the
// node at sorted position j (randomWalkers.cpp:86-93 through RcppArmadillo::sample's ProbSampleNoReplace)
__global__ __launch_bounds__(256) void graph_walk_kernel(int nn, const double *__restrict__ cum, const int *__restrict__ perm, int init_pos,
                                                          int n, int64_t nw, int64_t offset, const double *__restrict__ U, uint64_t seed,
                                                          double *__restrict__ out)
{
    const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (w >= nw) return;
    const PhiloxKey key(seed);
    int pos = init_pos;
    if (n > 0) out[w] = pos;
    double ua = 0.0, ub = 0.0;
    for (int i = 1; i < n; ++i) {
        double rT;
        if (U) {
            rT = U[(int64_t)(i - 1) * nw + w];
        } else {
            if ((i - 1) % 2 == 0) philox_uniform2(key, (uint64_t)(offset + w), (uint32_t)((i - 1) >> 1), TAG_WALK, 1u, ua, ub);
            rT = ((i - 1) & 1) ? ub : ua;
        }
        const double *row = cum + (size_t)pos * nn;
        int j = 0;
        while (j < nn - 1 && !(rT <= row[j])) ++j; // first sorted position whose running sum reaches rT, else the last
        pos = perm[(size_t)pos * nn + j];
        out[(int64_t)i * nw + w] = pos;
    }
}

// Genz (1992) separation of variables for P(lower <= X <= upper), X ~ N(mu, Sigma): Lc = lower Cholesky factor (row-major,
// d x d), a = lower - mu, b = upper - mu.  sums[0] += f, sums[1] += f^2 over all points.
__global__ __launch_bounds__(256) void pmvnorm_kernel(int d, const double *__restrict__ prm, int64_t n_points, uint64_t seed,
                                                       double *__restrict__ sums)
{
    __shared__ double red[2][256];
    const double *Lc = prm, *a = prm + (size_t)d * d, *b = a + d;
    const PhiloxKey key(seed);
    double s1 = 0.0, s2 = 0.0;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < n_points; p += (int64_t)gridDim.x * 256) {
        double y[kMaxDimPmvnorm];
        double f = 1.0, ua = 0.0, ub = 0.0;
        for (int i = 0; i < d; ++i) {
            double s = 0.0;
            for (int j = 0; j < i; ++j) s += Lc[(size_t)i * d + j] * y[j];
            const double lii = Lc[(size_t)i * d + i];
            // An interval that lies in the upper tail is mirrored: Phi(b) - Phi(a) with both near 1 cancels, and beyond
            // ~8.3 sd both ARE 1 -- the width would be 0 / the draw qnorm(1) = Inf and the next coordinate Inf - Inf
            // (ADVICE r2).  In the lower tail the probabilities are small numbers with full relative precision.
            const double za = (a[i] - s) / lii, zb = (b[i] - s) / lii;
            const bool up = za > 0.0;
            const double di = up ? pnorm_std(-zb) : pnorm_std(za), ei = up ? pnorm_std(-za) : pnorm_std(zb);
            const double width = ei - di;
            if (!(width > 0.0)) { // a box the arithmetic cannot tell from empty: this point contributes nothing
                f = 0.0;
                break;
            }
            f *= width;
            if (i + 1 < d) {
                if ((i & 1) == 0) philox_uniform2(key, (uint64_t)p, 0u, TAG_PMVNORM, (uint32_t)(i >> 1), ua, ub);
                const double wv = (i & 1) ? ub : ua;
                const double yy = qnorm_std(di + wv * width); // argument in (0, 1): di < ei <= 1, 0 < wv < 1
                y[i] = up ? -yy : yy;
            }
        }
        s1 += f;
        s2 += f * f;
    }
    red[0][threadIdx.x] = s1;
    red[1][threadIdx.x] = s2;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) {
            red[0][threadIdx.x] += red[0][threadIdx.x + st];
            red[1][threadIdx.x] += red[1][threadIdx.x + st];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        unsafeAtomicAdd(sums, red[0][0]);
        unsafeAtomicAdd(sums + 1, red[1][0]);
    }
}

// RAII device buffer (hipMalloc: these are host-synchronous convenience entry points, not hot-path launches)
struct XBuf {
    void *p = nullptr;
    ~XBuf()
    {
        if (p) (void)hipFree(p);
    }
    hipError_t put(const void *h, size_t bytes)
    {
        hipError_t e = hipMalloc(&p, bytes ? bytes : 8);
        if (e == hipSuccess && h && bytes) e = hipMemcpy(p, h, bytes, hipMemcpyHostToDevice);
        return e;
    }
};

static int mvn_block(int d, const double *h_mu, const double *h_sigma, std::vector<double> &blk)
{
    for (int e = 0; e < d * d; ++e)
        if (!std::isfinite(h_sigma[e])) return set_error(MCMC_E_BAD_ARG, "sigma has a non-finite entry");
    std::vector<double> H, R;
    double det = 0.0;
    if (!spd_inverse(d, h_sigma, H) || !chol_upper_packed(d, h_sigma, R, det)) return set_error(MCMC_E_NOT_PD, "sigma is not positive definite");
    blk.assign(h_mu, h_mu + d);
    blk.insert(blk.end(), H.begin(), H.end());
    blk.push_back(1.0 / std::sqrt(std::pow(2 * M_PI, d) * det)); // rootTerm (:18)
    return MCMC_OK;
}

static int pdf_launch(bool trunc, int d, int64_t n, const double *d_x, const std::vector<double> &blk, int do_log, double *d_out, hipStream_t s)
{
    XBuf prm;
    MCMC_HIP_CHECK(prm.put(blk.data(), blk.size() * sizeof(double)));
    const unsigned grid = (unsigned)((n + 255) / 256);
    if (trunc)
        hipLaunchKernelGGL(mvn_pdf_kernel<true>, dim3(grid), dim3(256), 0, s, n, d, d_x, (const double *)prm.p, do_log, d_out);
    else
        hipLaunchKernelGGL(mvn_pdf_kernel<false>, dim3(grid), dim3(256), 0, s, n, d, d_x, (const double *)prm.p, do_log, d_out);
    MCMC_HIP_CHECK(hipGetLastError());
    MCMC_HIP_CHECK(hipStreamSynchronize(s)); // prm is freed on return
    return MCMC_OK;
}

} // namespace mcmc

using namespace mcmc;

extern "C" int mcmc_mvrGaussian_pdf_batch(int32_t d, int64_t n, const double *d_x, const double *h_mu, const double *h_sigma, int32_t do_log,
                                          double *d_out, void *stream)
{
    if (d < 1 || n < 0 || !h_mu || !h_sigma || (n > 0 && (!d_x || !d_out))) return set_error(MCMC_E_BAD_ARG, "mcmc_mvrGaussian_pdf_batch: bad argument");
    if (n == 0) return MCMC_OK;
    std::vector<double> blk;
    int rc = mvn_block(d, h_mu, h_sigma, blk);
    if (rc != MCMC_OK) return rc;
    return pdf_launch(false, d, n, d_x, blk, do_log, d_out, (hipStream_t)stream);
}

extern "C" int mcmc_tmvrGaussian_pdf_batch(int32_t d, int64_t n, const double *d_x, const double *h_mu, const double *h_sigma,
                                           const double *h_lower, const double *h_upper, double cdf, double *d_out, void *stream)
{
    if (d < 1 || n < 0 || !h_mu || !h_sigma || !h_lower || !h_upper || (n > 0 && (!d_x || !d_out)))
        return set_error(MCMC_E_BAD_ARG, "mcmc_tmvrGaussian_pdf_batch: bad argument");
    if (n == 0) return MCMC_OK;
    std::vector<double> blk;
    int rc = mvn_block(d, h_mu, h_sigma, blk);
    if (rc != MCMC_OK) return rc;
    blk.insert(blk.end(), h_lower, h_lower + d);
    blk.insert(blk.end(), h_upper, h_upper + d);
    blk.push_back(cdf);
    return pdf_launch(true, d, n, d_x, blk, 1, d_out, (hipStream_t)stream);
}

static int pdf_single(bool trunc, int32_t d, const double *h_x, const double *h_mu, const double *h_sigma, const double *h_lower,
                      const double *h_upper, double cdf, int32_t do_log, double *out)
{
    if (d < 1 || !h_x || !out) return set_error(MCMC_E_BAD_ARG, "pdf: bad argument");
    XBuf x, o;
    MCMC_HIP_CHECK(x.put(h_x, sizeof(double) * d)); // [d][1]
    MCMC_HIP_CHECK(o.put(nullptr, sizeof(double)));
    int rc = trunc ? mcmc_tmvrGaussian_pdf_batch(d, 1, (const double *)x.p, h_mu, h_sigma, h_lower, h_upper, cdf, (double *)o.p, nullptr)
                   : mcmc_mvrGaussian_pdf_batch(d, 1, (const double *)x.p, h_mu, h_sigma, do_log, (double *)o.p, nullptr);
    if (rc != MCMC_OK) return rc;
    MCMC_HIP_CHECK(hipMemcpy(out, o.p, sizeof(double), hipMemcpyDeviceToHost));
    return MCMC_OK;
}

extern "C" int mcmc_mvrGaussian_pdf(int32_t d, const double *h_x, const double *h_mu, const double *h_sigma, int32_t do_log, double *out)
{
    return pdf_single(false, d, h_x, h_mu, h_sigma, nullptr, nullptr, 0.0, do_log, out);
}

extern "C" int mcmc_tmvrGaussian_pdf(int32_t d, const double *h_x, const double *h_mu, const double *h_sigma, const double *h_lower,
                                     const double *h_upper, double cdf, double *out)
{
    return pdf_single(true, d, h_x, h_mu, h_sigma, h_lower, h_upper, cdf, 1, out);
}

extern "C" int mcmc_pmvnorm(int32_t d, const double *h_lower, const double *h_upper, const double *h_mu, const double *h_sigma,
                            int64_t n_points, uint64_t seed, double *prob, double *std_err)
{
    if (d < 1 || d > kMaxDimPmvnorm) return set_error(d < 1 ? MCMC_E_BAD_ARG : MCMC_E_UNSUPPORTED, "mcmc_pmvnorm: d=%d outside 1..%d", d, kMaxDimPmvnorm);
    if (!h_lower || !h_upper || !h_mu || !h_sigma || !prob || n_points < 2) return set_error(MCMC_E_BAD_ARG, "mcmc_pmvnorm: bad argument");
    std::vector<double> R;
    double det = 0.0;
    if (!chol_upper_packed(d, h_sigma, R, det)) return set_error(MCMC_E_NOT_PD, "sigma is not positive definite");
    std::vector<double> blk((size_t)d * d + 2 * d, 0.0);
    for (int i = 0; i < d; ++i) {
        for (int j = 0; j <= i; ++j) blk[(size_t)i * d + j] = R[tri(j, i)]; // lower factor L = R^T, row-major
        if (!(h_lower[i] < h_upper[i])) return set_error(MCMC_E_BAD_ARG, "need lower[%d] < upper[%d]", i, i); // also refuses NaN bounds
        if (!std::isfinite(h_mu[i])) return set_error(MCMC_E_BAD_ARG, "mu[%d] is not finite", i);
        blk[(size_t)d * d + i] = h_lower[i] - h_mu[i];
        blk[(size_t)d * d + d + i] = h_upper[i] - h_mu[i];
    }
    XBuf prm, sums;
    const double zero[2] = {0.0, 0.0};
    MCMC_HIP_CHECK(prm.put(blk.data(), blk.size() * sizeof(double)));
    MCMC_HIP_CHECK(sums.put(zero, sizeof zero));
    const unsigned grid = (unsigned)std::min<int64_t>((n_points + 255) / 256, 4096);
    hipLaunchKernelGGL(pmvnorm_kernel, dim3(grid), dim3(256), 0, nullptr, d, (const double *)prm.p, n_points, seed, (double *)sums.p);
    MCMC_HIP_CHECK(hipGetLastError());
    double h[2];
    MCMC_HIP_CHECK(hipMemcpy(h, sums.p, sizeof h, hipMemcpyDeviceToHost));
    const double mean = h[0] / (double)n_points;
    const double var = std::fmax(0.0, h[1] / (double)n_points - mean * mean);
    *prob = mean;
    if (std_err) *std_err = std::sqrt(var / (double)n_points);
    return MCMC_OK;
}

extern "C" int mcmc_random_walk_batch(int32_t dim, int32_t n, int64_t n_walkers, int64_t walker_offset, const double *d_U, uint64_t seed,
                                      double *d_out, void *stream)
{
    if ((dim != 2 && dim != 3) || n < 1 || n_walkers < 1 || !d_out) return set_error(MCMC_E_BAD_ARG, "mcmc_random_walk_batch: dim must be 2 or 3, n and n_walkers >= 1");
    hipLaunchKernelGGL(lattice_walk_kernel, dim3((unsigned)((n_walkers + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dim, n, n_walkers,
                       walker_offset, d_U, seed, d_out);
    MCMC_HIP_CHECK(hipGetLastError());
    return MCMC_OK;
}

static int walk_single(int dim, int32_t n, const double *h_U, uint64_t seed, double *h_out)
{
    if (n < 1 || !h_out) return set_error(MCMC_E_BAD_ARG, "random_walk: n must be >= 1");
    XBuf U, out;
    if (h_U && n > 1) MCMC_HIP_CHECK(U.put(h_U, sizeof(double) * (n - 1)));
    MCMC_HIP_CHECK(out.put(nullptr, sizeof(double) * (size_t)n * dim));
    int rc = mcmc_random_walk_batch(dim, n, 1, 0, h_U ? (const double *)U.p : nullptr, seed, (double *)out.p, nullptr);
    if (rc != MCMC_OK) return rc;
    std::vector<double> tmp((size_t)n * dim);
    MCMC_HIP_CHECK(hipMemcpy(tmp.data(), out.p, tmp.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < dim; ++k) h_out[(size_t)k * n + i] = tmp[(size_t)i * dim + k]; // n x dim column-major (NumericMatrix)
    return MCMC_OK;
}

extern "C" int mcmc_random_walk2d(int32_t n, const double *h_U, uint64_t seed, double *h_out) { return walk_single(2, n, h_U, seed, h_out); }
extern "C" int mcmc_random_walk3d(int32_t n, const double *h_U, uint64_t seed, double *h_out) { return walk_single(3, n, h_U, seed, h_out); }

extern "C" int mcmc_random_walkGraph_batch(int32_t n_nodes, const double *h_graph, int32_t init_pos, int32_t n, int64_t n_walkers,
                                           int64_t walker_offset, const double *d_U, uint64_t seed, double *d_out, void *stream)
{
    if (n_nodes < 1 || !h_graph || init_pos < 0 || init_pos >= n_nodes || n < 1 || n_walkers < 1 || !d_out)
        return set_error(MCMC_E_BAD_ARG, "mcmc_random_walkGraph_batch: bad argument");
    // per row: FixProb (normalise) + descending sort carrying the node indices + running sums, as ProbSampleNoReplace walks them
    std::vector<double> cum((size_t)n_nodes * n_nodes);
    std::vector<int> perm((size_t)n_nodes * n_nodes);
    std::vector<double> p(n_nodes);
    std::vector<int> idx(n_nodes);
    for (int r = 0; r < n_nodes; ++r) {
        double sum = 0.0;
        for (int k = 0; k < n_nodes; ++k) {
            p[k] = h_graph[(size_t)k * n_nodes + r]; // graph(r, _), column-major
            if (!(p[k] >= 0.0) || !std::isfinite(p[k])) return set_error(MCMC_E_BAD_ARG, "graph(%d,%d) is not a finite non-negative weight", r, k);
            sum += p[k];
        }
        if (!(sum > 0.0)) return set_error(MCMC_E_BAD_ARG, "row %d of graph has no positive weight", r);
        for (int k = 0; k < n_nodes; ++k) p[k] = p[k] / sum;
        std::iota(idx.begin(), idx.end(), 0);
        std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return p[a] > p[b]; });
        double mass = 0.0;
        for (int j = 0; j < n_nodes; ++j) {
            mass += p[idx[j]];
            cum[(size_t)r * n_nodes + j] = mass;
            perm[(size_t)r * n_nodes + j] = idx[j];
        }
    }
    XBuf dc, dp;
    MCMC_HIP_CHECK(dc.put(cum.data(), cum.size() * sizeof(double)));
    MCMC_HIP_CHECK(dp.put(perm.data(), perm.size() * sizeof(int)));
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(graph_walk_kernel, dim3((unsigned)((n_walkers + 255) / 256)), dim3(256), 0, s, n_nodes, (const double *)dc.p,
                       (const int *)dp.p, init_pos, n, n_walkers, walker_offset, d_U, seed, d_out);
    MCMC_HIP_CHECK(hipGetLastError());
    MCMC_HIP_CHECK(hipStreamSynchronize(s)); // the tables are freed on return
    return MCMC_OK;
}

extern "C" int mcmc_random_walkGraph(int32_t n_nodes, const double *h_graph, int32_t init_pos, int32_t n, const double *h_U, uint64_t seed,
                                     double *h_out)
{
    if (n < 1 || !h_out) return set_error(MCMC_E_BAD_ARG, "mcmc_random_walkGraph: n must be >= 1");
    XBuf U, out;
    if (h_U && n > 1) MCMC_HIP_CHECK(U.put(h_U, sizeof(double) * (n - 1)));
    MCMC_HIP_CHECK(out.put(nullptr, sizeof(double) * (size_t)n));
    int rc = mcmc_random_walkGraph_batch(n_nodes, h_graph, init_pos, n, 1, 0, h_U ? (const double *)U.p : nullptr, seed, (double *)out.p, nullptr);
    if (rc != MCMC_OK) return rc;
    MCMC_HIP_CHECK(hipMemcpy(h_out, out.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
    return MCMC_OK;
}
