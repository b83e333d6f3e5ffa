// Adaptive random-walk Metropolis, one chain per lane, d <= 12 (gfx950).
//
// Replaces the main loop of adaptMCMC (adaptMCMC_source.cpp:108-216) and of rwMCMC
// (:252-313; the same loop with both adapt_*_start = 0).  A wavefront advances 64 chains in
// lockstep.  Per chain the kernel keeps theta and the packed upper Cholesky factor R of
// covmat_proposal (R^T R = covmat_proposal, as arma::chol returns it) in VGPRs, and theta_mean
// plus the packed upper triangle of covmat_empirical in LDS as [element][lane] (touched once per
// iteration; bank = lane, conflict free).  Everything shared by the chains (covmat, chol(covmat),
// target parameters) is one small block staged into LDS at kernel start and read with
// same-address (broadcast) LDS loads.
//
// What is done differently from the reference, and why it does not change results beyond
// rounding:
//   * chol(covmat_proposal) is recomputed only in iterations that assign covmat_proposal
//     (:129,138); the reference refactors the unchanged matrix every iteration (:153).
//   * the Hastings term mvrnorm_pdf (:48-64,169-170) needs det(Sigma) and v^T Sigma^-1 v with
//     v = Y R: det = prod(diag R)^2 and v^T Sigma^-1 v = |Y|^2, so no LU det/inv is formed.  The
//     term is still evaluated as log(rootTerm * exp(-q/2)), added and subtracted in that order,
//     so the (la + p) - p rounding and the NaN-reject path when p is +-Inf are kept.
//   * divisions by the iteration count in update_sigma (:43) and the mean update (:212) are
//     multiplications by 1/i; sqrt / reciprocal pivots / log use the lean sequences of fastmath.h
//     (<= 1-2 ulp).
// The acceptance-rate recurrences (:188-207) and the phase tests (:117,131) are evaluated
// with the reference's exact operation sequence (no FMA contraction), because they feed
// discrete decisions.
#include "accept_window.h"
#include "common.h"
#include "philox.h"
#include "targets.h"

namespace mcmc {

// parameter block and LDS layout: lane_layout() in common.h
template <int D, int KIND>
struct RwmLayout {
    static constexpr LaneLayout L = lane_layout(D, target_param_count(KIND, D));
    static constexpr int T = L.T, NP = L.NP;
    static constexpr int INIT = L.INIT, COV = L.COV, R0 = L.R0, ROOT0 = L.ROOT0, TPRM = L.TPRM;
    static constexpr int TOTAL = L.TOTAL, SH = L.SH, LDS_DOUBLES = L.LDS_DOUBLES;
};

// view of the target parameters inside the block, indexable like an array
struct PrmView {
    const double *p;
    __device__ __forceinline__ double operator[](int i) const { return p[i]; }
};

// FULL = false is the production variant (Philox streams; theta trace, final state and per-chain
// summaries only).  FULL = true adds injected streams, the sigma_empirical trace and the per-iteration
// parity diagnostics; keeping those ~250 extra store addresses out of the production kernel is what
// keeps its scalar registers from spilling.
template <int D, int KIND, bool FULL>
__global__ __launch_bounds__(64) void rwm_lane_kernel(const RwmArgs a, const double *__restrict__ prm)
{
    using Lay = RwmLayout<D, KIND>;
    constexpr int T = Lay::T;
    extern __shared__ double lds[]; // shared parameters, then C[e][lane] for e < T, then mean[k][lane]
    const int lane = threadIdx.x;
    const int64_t c = (int64_t)blockIdx.x * 64 + lane;
    const int64_t n = a.n_chains;
    const bool live = c < n;
    const int64_t cc = live ? c : n - 1; // tail lanes shadow the last chain and store nothing
    const uint64_t gchain = (uint64_t)(a.chain_offset + cc);
    const PhiloxKey key(a.seed);
    const bool inject = FULL && a.d_Z != nullptr;
    const int ass = a.adapt_size_start, ashs = a.adapt_shape_start;
    const int window = a.acceptance_window;
    const double w = a.acceptance_rate_weight;
    const bool need_moments = ashs != 0 || (FULL && a.d_sigma_trace) || a.d_final_mean || a.d_final_cov || a.d_state_in || a.d_state_out;
    const int i0 = a.iter_offset; // iterations done before this launch (checkpoint / resume)
    // Stage the shared parameter block in LDS: the target alone needs d + d^2 values per iteration,
    // more than the ~100 SGPRs of a wave, so scalar loads would be spilled to VGPR lanes; from LDS
    // they arrive as same-address (broadcast) reads.
    for (int e = lane; e < Lay::TOTAL; e += 64) lds[e] = prm[e];
    __syncthreads();
    const double *const sh = lds;
    double *const Cl = lds + Lay::SH + lane;          // C element e at Cl[e * 64]
    double *const Ml = lds + Lay::SH + T * 64 + lane; // mean element k at Ml[k * 64]

    double theta[D], R[T];
#pragma unroll
    for (int k = 0; k < D; ++k) {
        theta[k] = a.d_init ? a.d_init[(int64_t)k * n + cc] : sh[Lay::INIT + k];
        Ml[k * 64] = theta[k];
    }
#pragma unroll
    for (int e = 0; e < T; ++e) {
        Cl[e * 64] = 0.0;
        R[e] = sh[Lay::R0 + e];
    }
    double rootTerm = sh[Lay::ROOT0];
    double t_cur = Target<KIND, D>::eval(theta, PrmView{sh + Lay::TPRM});
    double rate = 0.0, scaling_sd = 1.0, scaling_multiplier = 1.0;
    AcceptWindow win;
    int win_pos = accept_window_pos(i0, window); // wave-uniform: every chain is at the same iteration
    unsigned long long *const ring = a.d_win_ring ? a.d_win_ring + cc : nullptr;
    int32_t status = MCMC_CHAIN_OK, n_shape = 0;
    bool dead = false;
    constexpr RwmStateLayout SL = rwm_state_layout(D);
    if (a.d_state_in) { // resume: everything the loop carries from one iteration to the next
        const double *const sp = a.d_state_in + cc;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            theta[k] = sp[(SL.THETA + k) * n];
            Ml[k * 64] = sp[(SL.MEAN + k) * n];
        }
#pragma unroll
        for (int e = 0; e < T; ++e) {
            Cl[e * 64] = sp[(SL.C + e) * n];
            R[e] = sp[(SL.R + e) * n];
        }
        const double *const ss = sp + (int64_t)SL.S * n;
        t_cur = ss[ST_TCUR * n];
        rate = ss[ST_RATE * n];
        scaling_sd = ss[ST_SD * n];
        scaling_multiplier = ss[ST_MULT * n];
        rootTerm = ss[ST_ROOT * n];
        win.bits = ((uint64_t)(uint32_t)ss[ST_WHI * n] << 32) | (uint64_t)(uint32_t)ss[ST_WLO * n];
        win.len = (int)ss[ST_WLEN * n];
        win.count = (int)ss[ST_WCOUNT * n];
        if (live)
            for (int wd = 0; wd < a.win_words; ++wd) // the outcome ring (acceptance_window > 64) follows the scalars
                ring[(int64_t)wd * n] = ((uint64_t)(uint32_t)sp[(int64_t)(SL.TOTAL + 2 * wd + 1) * n] << 32) |
                                        (uint64_t)(uint32_t)sp[(int64_t)(SL.TOTAL + 2 * wd) * n];
        status = (int32_t)ss[ST_STATUS * n];
        n_shape = (int32_t)ss[ST_NSHAPE * n];
        dead = status != MCMC_CHAIN_OK;
        if (ss[ST_TAG * n] != (double)MCMC_KERNEL_LANE || ss[ST_ITER * n] != (double)i0) { // not a state this launch can continue
            dead = true;
            status = ((i0 + 1) << 4) | MCMC_CHAIN_BAD_STATE;
#pragma unroll
            for (int k = 0; k < D; ++k) theta[k] = __builtin_nan("");
        }
    }

    for (int i = i0 + 1; i <= i0 + a.n_iterations; ++i) {
        const int li = i - i0; // position inside this launch: streams, diagnostics and the cooling table are per launch
        const double *const pz = sh;
        const double di = (double)i;
        // ---- adaptation of covmat_proposal (adaptMCMC_source.cpp:117-139) ------------------
        bool dirty = false, use_emp = false;
        double s2 = 0.0;
        if (ass != 0 && i >= ass && (ashs == 0 || mul_rn(rate, di) < (double)ashs)) {
            scaling_multiplier = exp(a.d_cool_pow[min(li, a.cool_last)] * (rate - 0.234));
            scaling_sd = mul_rn(scaling_sd, scaling_multiplier);
            scaling_sd = (a.max_scaling_sd < scaling_sd) ? a.max_scaling_sd : scaling_sd; // std::min
            s2 = mul_rn(scaling_sd, scaling_sd);
            bool any_small = false;
#pragma unroll
            for (int k = 0; k < D; ++k) any_small |= (mul_rn(s2, pz[Lay::COV + tri(k, k)]) < 2E-16);
            dirty = !any_small;
        } else if (ashs != 0 && mul_rn(rate, di) >= (double)ashs) {
            scaling_sd = a.shape_scaling_sd;
            s2 = mul_rn(scaling_sd, scaling_sd);
            dirty = true;
            use_emp = true;
            ++n_shape;
        }
        if (FULL && a.d_scaling_sd && live) a.d_scaling_sd[(int64_t)(li - 1) * n + c] = scaling_sd;
        if (FULL && a.d_scaling_multiplier && live) a.d_scaling_multiplier[(int64_t)(li - 1) * n + c] = scaling_multiplier;

        // ---- R = chol(covmat_proposal) when it was assigned (:153) -------------------------
        if (dirty && !dead) {
            bool ok = true;
            double prod = 1.0;
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const double cjj = Cl[tri(j, j) * 64], pjj = pz[Lay::COV + tri(j, j)];
                double ajj = s2 * (use_emp ? cjj : pjj);
#pragma unroll
                for (int m = 0; m < j; ++m) ajj -= R[tri(m, j)] * R[tri(m, j)];
                ok = ok && (ajj > 0.0);
                double rjj, rinv;
                fm_sqrt_rsq(ajj, rjj, rinv);
                R[tri(j, j)] = rjj;
                prod *= rjj;
#pragma unroll
                for (int k = j + 1; k < D; ++k) {
                    const double cjk = Cl[tri(j, k) * 64], pjk = pz[Lay::COV + tri(j, k)];
                    double s = s2 * (use_emp ? cjk : pjk);
#pragma unroll
                    for (int m = 0; m < j; ++m) s -= R[tri(m, j)] * R[tri(m, k)];
                    R[tri(j, k)] = s * rinv;
                }
            }
            rootTerm = 1.0 / sqrt(a.twopi_pow_d * (prod * prod));
            if (!ok) { // arma::chol throws here; the chain stops and reports where
                dead = true;
                status = (i << 4) | MCMC_CHAIN_CHOL_FAILED;
#pragma unroll
                for (int k = 0; k < D; ++k) theta[k] = __builtin_nan("");
            }
        }

        // ---- proposal theta + Y * R (:149-154) ----------------------------------------------
        double Y[D], prop[D];
        if (inject) {
#pragma unroll
            for (int k = 0; k < D; ++k) Y[k] = a.d_Z[((int64_t)(li - 1) * D + k) * n + cc];
        } else {
#pragma unroll
            for (int b = 0; 2 * b < D; ++b) {
                double z0, z1;
                philox_normal2(key, gchain, (uint32_t)i, TAG_RWM_NORMAL, (uint32_t)b, z0, z1);
                Y[2 * b] = z0;
                if (2 * b + 1 < D) Y[2 * b + 1] = z1;
            }
        }
        double q = 0.0;
#pragma unroll
        for (int j = 0; j < D; ++j) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k <= j; ++k) s += Y[k] * R[tri(k, j)];
            prop[j] = theta[j] + s;
            q += Y[j] * Y[j];
        }

        // ---- target and Metropolis-Hastings log ratio (:157-171) ----------------------------
        const double t_prop = Target<KIND, D>::eval(prop, PrmView{pz + Lay::TPRM});
        double la;
        if (!isfinite(t_prop)) {
            la = -__builtin_inf();
        } else {
            la = t_prop - t_cur;
            const double p = fm_log<true>(rootTerm * exp(-0.5 * q)); // mvrnorm_pdf, both directions (:48-64)
            la = add_rn(la, p);
            la = add_rn(la, -p);
        }

        // ---- accept / reject (:174-182) ------------------------------------------------------
        double A;
        if (inject) {
            A = a.d_U[(int64_t)(li - 1) * n + cc];
        } else {
            double u1;
            philox_uniform2(key, gchain, (uint32_t)i, TAG_RWM_UNIFORM, 0u, A, u1);
        }
        const double logA = fm_log<true>(A);
        const bool accept = logA < la;
        if (accept) {
#pragma unroll
            for (int k = 0; k < D; ++k) theta[k] = prop[k];
            t_cur = t_prop;
        }
        if (FULL && live) {
            if (a.d_accepted) a.d_accepted[(int64_t)(li - 1) * n + c] = accept ? 1 : 0;
            if (a.d_log_acceptance) a.d_log_acceptance[(int64_t)(li - 1) * n + c] = la;
            if (a.d_log_A) a.d_log_A[(int64_t)(li - 1) * n + c] = logA;
        }

        // ---- trace (:185) -------------------------------------------------------------------
        const bool keep = (i % a.thin) == 0;
        const int64_t t = i / a.thin - i0 / a.thin - 1; // row inside this launch's trace
        if (a.d_trace && keep && live) {
#pragma unroll
            for (int k = 0; k < D; ++k) a.d_trace[(t * D + k) * n + c] = theta[k];
        }

        // ---- acceptance rate (:188-207): the reference's exact operation sequence ------------
        const double acc = accept ? 1.0 : 0.0;
        if (i == 1) {
            rate = acc;
        } else if (w == 0) {
            if (window == 0) {
                rate = add_rn(rate, div_rn(add_rn(acc, -rate), di));
            } else {
                rate = accept_window_push(win, accept, window, win_pos, ring, n, live);
            }
        } else {
            rate = add_rn(mul_rn(rate, 1 - w), mul_rn(acc, w));
        }
        if (FULL && a.d_rate_trace && live) a.d_rate_trace[(int64_t)(li - 1) * n + c] = rate;

        // ---- running mean / covariance (:210-214, update_sigma :42-45) ----------------------
        if (need_moments) {
            const double im1 = di - 1.0;
            const double sfac = im1 / di;
            const double inv_i = 1.0 / di;
            double res[D];
#pragma unroll
            for (int k = 0; k < D; ++k) {
                const double mk = Ml[k * 64];
                res[k] = theta[k] - mk;
                Ml[k * 64] = mk + res[k] * inv_i;
            }
            const bool sig = FULL && a.d_sigma_trace && keep && live;
#pragma unroll
            for (int cc2 = 0; cc2 < D; ++cc2) {
#pragma unroll
                for (int r = 0; r <= cc2; ++r) {
                    const double v = (Cl[tri(r, cc2) * 64] * im1 + (sfac * res[r]) * res[cc2]) * inv_i;
                    Cl[tri(r, cc2) * 64] = v;
                    if (sig) {
                        a.d_sigma_trace[(t * (D * D) + cc2 * D + r) * n + c] = v;
                        if (r != cc2) a.d_sigma_trace[(t * (D * D) + r * D + cc2) * n + c] = v;
                    }
                }
            }
        }
    }

    if (live) {
#pragma unroll
        for (int k = 0; k < D; ++k) {
            if (a.d_final_theta) a.d_final_theta[(int64_t)k * n + c] = theta[k];
            if (a.d_final_mean) a.d_final_mean[(int64_t)k * n + c] = Ml[k * 64];
        }
        if (a.d_final_cov) {
#pragma unroll
            for (int cc2 = 0; cc2 < D; ++cc2) {
#pragma unroll
                for (int r = 0; r < D; ++r)
                    a.d_final_cov[(int64_t)(cc2 * D + r) * n + c] = Cl[(r <= cc2 ? tri(r, cc2) : tri(cc2, r)) * 64];
            }
        }
        if (a.d_acceptance_rate) a.d_acceptance_rate[c] = rate;
        if (a.d_status) a.d_status[c] = status;
        if (a.d_shape_iterations) a.d_shape_iterations[c] = n_shape;
        if (a.d_state_out) {
            double *const sp = a.d_state_out + c;
#pragma unroll
            for (int k = 0; k < D; ++k) {
                sp[(SL.THETA + k) * n] = theta[k];
                sp[(SL.MEAN + k) * n] = Ml[k * 64];
            }
#pragma unroll
            for (int e = 0; e < T; ++e) {
                sp[(SL.C + e) * n] = Cl[e * 64];
                sp[(SL.R + e) * n] = R[e];
            }
            double *const ss = sp + (int64_t)SL.S * n;
            ss[ST_TAG * n] = (double)MCMC_KERNEL_LANE;
            ss[ST_ITER * n] = (double)(i0 + a.n_iterations);
            ss[ST_TCUR * n] = t_cur;
            ss[ST_RATE * n] = rate;
            ss[ST_SD * n] = scaling_sd;
            ss[ST_MULT * n] = scaling_multiplier;
            ss[ST_SEFF * n] = 1.0;
            ss[ST_PRODR * n] = 1.0;
            ss[ST_ROOT * n] = rootTerm;
            ss[ST_WLO * n] = (double)(uint32_t)win.bits;
            ss[ST_WHI * n] = (double)(uint32_t)(win.bits >> 32);
            ss[ST_WLEN * n] = (double)win.len;
            ss[ST_BASE * n] = 0.0;
            ss[ST_STATUS * n] = (double)status;
            ss[ST_NSHAPE * n] = (double)n_shape;
            ss[ST_WCOUNT * n] = (double)win.count;
            for (int wd = 0; wd < a.win_words; ++wd) {
                const uint64_t v = (wd == (win_pos >> 6)) ? win.bits
                                                          : (uint64_t)__hip_atomic_load(ring + (int64_t)wd * n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                sp[(int64_t)(SL.TOTAL + 2 * wd) * n] = (double)(uint32_t)v;
                sp[(int64_t)(SL.TOTAL + 2 * wd + 1) * n] = (double)(uint32_t)(v >> 32);
            }
        }
    }
}

#ifndef __HIPCC_RTC__
// The normals / uniforms a Philox run draws: Z [n_iterations][d][n], U [n_iterations][n].
__global__ __launch_bounds__(256) void rwm_streams_kernel(uint64_t seed, int64_t chain_offset, int64_t n,
                                                          int32_t n_iterations, int32_t d, double *Z, double *U)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    const PhiloxKey key(seed);
    const uint64_t gchain = (uint64_t)(chain_offset + c);
    for (int i = 1; i <= n_iterations; ++i) {
        if (Z) {
            for (int b = 0; 2 * b < d; ++b) {
                double z0, z1;
                philox_normal2(key, gchain, (uint32_t)i, TAG_RWM_NORMAL, (uint32_t)b, z0, z1);
                Z[((int64_t)(i - 1) * d + 2 * b) * n + c] = z0;
                if (2 * b + 1 < d) Z[((int64_t)(i - 1) * d + 2 * b + 1) * n + c] = z1;
            }
        }
        if (U) {
            double u0, u1;
            philox_uniform2(key, gchain, (uint32_t)i, TAG_RWM_UNIFORM, 0u, u0, u1);
            U[(int64_t)(i - 1) * n + c] = u0;
        }
    }
}

__global__ __launch_bounds__(256) void init_normal_kernel(uint64_t seed, int64_t chain_offset, int64_t n, int32_t d,
                                                          const double *__restrict__ mean, double sd, double *out)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    const PhiloxKey key(seed);
    const uint64_t gchain = (uint64_t)(chain_offset + c);
    for (int b = 0; 2 * b < d; ++b) {
        double z0, z1;
        philox_normal2(key, gchain, 0u, TAG_INIT, (uint32_t)b, z0, z1);
        out[(int64_t)(2 * b) * n + c] = mean[2 * b] + sd * z0;
        if (2 * b + 1 < d) out[(int64_t)(2 * b + 1) * n + c] = mean[2 * b + 1] + sd * z1;
    }
}

// ------------------------------------------------------------------------------------------
int launch_rwm_lane_with(const RwmArgs &a, const RwmParams &p, hipStream_t s, const LaneLayout &L, const RwmLauncher &launch)
{
    const int D = a.d;
    if ((int)p.tparams.size() != L.NP)
        return set_error(MCMC_E_BAD_ARG, "target kind %d with d=%d needs %d parameters, got %d", p.target_kind, D, L.NP,
                         (int)p.tparams.size());
    std::vector<double> blob(L.TOTAL, 0.0);
    for (int k = 0; k < D; ++k) blob[L.INIT + k] = p.init[k];
    for (int e = 0; e < L.T; ++e) {
        blob[L.COV + e] = p.cov_packed[e];
        blob[L.R0 + e] = p.R0_packed[e];
    }
    blob[L.ROOT0] = 1.0 / std::sqrt(a.twopi_pow_d * p.det0);
    for (int e = 0; e < L.NP; ++e) blob[L.TPRM + e] = p.tparams[e];
    double *d_prm = nullptr;
    MCMC_HIP_CHECK(hipMallocAsync((void **)&d_prm, blob.size() * sizeof(double), s));
    if (const int ru = upload_block(d_prm, blob.data(), blob.size(), s); ru != MCMC_OK) {
        (void)hipFreeAsync(d_prm, s);
        return ru;
    }
    const int64_t grid = (a.n_chains + 63) / 64;
    const size_t ldsb = (size_t)L.LDS_DOUBLES * sizeof(double);
    const bool full = a.d_Z || a.d_sigma_trace || a.d_accepted || a.d_log_acceptance || a.d_log_A || a.d_scaling_sd ||
                      a.d_scaling_multiplier || a.d_rate_trace;
    hipError_t e1 = launch(full ? 1 : 0, (unsigned)grid, 64u, ldsb, s, a, d_prm);
    hipError_t e2 = hipFreeAsync(d_prm, s);
    if (e1 != hipSuccess) return set_error(MCMC_E_HIP, "rwm_lane_kernel launch failed: %s", hipGetErrorString(e1));
    if (e2 != hipSuccess) return set_error(MCMC_E_HIP, "hipFreeAsync failed: %s", hipGetErrorString(e2));
    return MCMC_OK;
}

template <int D, int KIND>
static int launch_rwm_lane_dk(const RwmArgs &a, const RwmParams &p, hipStream_t s)
{
    return launch_rwm_lane_with(a, p, s, RwmLayout<D, KIND>::L,
                                [](int full, unsigned grid, unsigned block, size_t ldsb, hipStream_t st, const RwmArgs &aa,
                                   const double *d_prm) {
                                    if (full)
                                        hipLaunchKernelGGL((rwm_lane_kernel<D, KIND, true>), dim3(grid), dim3(block), ldsb, st, aa, d_prm);
                                    else
                                        hipLaunchKernelGGL((rwm_lane_kernel<D, KIND, false>), dim3(grid), dim3(block), ldsb, st, aa, d_prm);
                                    return hipGetLastError();
                                });
}

int launch_rwm_lane(const RwmArgs &a, const RwmParams &p, hipStream_t s)
{
    if (p.target_kind == MCMC_TARGET_USER) return launch_rwm_lane_user(a, p, s);
    if (p.target_kind == MCMC_TARGET_GOLDSTEIN_PRICE) {
        if (a.d != 2) return set_error(MCMC_E_BAD_ARG, "Goldstein-Price target is 2-dimensional (d=%d)", a.d);
        return launch_rwm_lane_dk<2, MCMC_TARGET_GOLDSTEIN_PRICE>(a, p, s);
    }
    if (p.target_kind == MCMC_TARGET_BANANA) {
        if (a.d != 2) return set_error(MCMC_E_BAD_ARG, "banana target is 2-dimensional (d=%d)", a.d);
        return launch_rwm_lane_dk<2, MCMC_TARGET_BANANA>(a, p, s);
    }
    if (p.target_kind != MCMC_TARGET_GAUSSIAN_PREC)
        return set_error(MCMC_E_BAD_ARG, "unknown target kind %d", p.target_kind);
    switch (a.d) {
#define MCMC_CASE(D) \
    case D: return launch_rwm_lane_dk<D, MCMC_TARGET_GAUSSIAN_PREC>(a, p, s);
        MCMC_CASE(1) MCMC_CASE(2) MCMC_CASE(3) MCMC_CASE(4) MCMC_CASE(5) MCMC_CASE(6)
        MCMC_CASE(7) MCMC_CASE(8) MCMC_CASE(9) MCMC_CASE(10) MCMC_CASE(11) MCMC_CASE(12)
#undef MCMC_CASE
    default: return set_error(MCMC_E_UNSUPPORTED, "rwm lane kernel: d=%d outside 1..%d", a.d, kMaxLaneDimRwm);
    }
}

int launch_rwm_streams(uint64_t seed, int64_t chain_offset, int64_t n_chains, int32_t n_iterations, int32_t d,
                       double *d_Z, double *d_U, hipStream_t s)
{
    const int block = 256;
    const int64_t grid = (n_chains + block - 1) / block;
    hipLaunchKernelGGL(rwm_streams_kernel, dim3((unsigned)grid), dim3(block), 0, s, seed, chain_offset, n_chains,
                       n_iterations, d, d_Z, d_U);
    MCMC_HIP_CHECK(hipGetLastError());
    return MCMC_OK;
}

int launch_init_normal(uint64_t seed, int64_t chain_offset, int64_t n_chains, int32_t d, const double *h_mean,
                       double sd, double *d_out, hipStream_t s)
{
    double *d_mean = nullptr;
    MCMC_HIP_CHECK(hipMallocAsync((void **)&d_mean, sizeof(double) * d, s));
    MCMC_HIP_CHECK(hipMemcpyAsync(d_mean, h_mean, sizeof(double) * d, hipMemcpyHostToDevice, s));
    MCMC_HIP_CHECK(hipStreamSynchronize(s)); // h_mean belongs to the caller
    const int block = 256;
    const int64_t grid = (n_chains + block - 1) / block;
    hipLaunchKernelGGL(init_normal_kernel, dim3((unsigned)grid), dim3(block), 0, s, seed, chain_offset, n_chains, d,
                       d_mean, sd, d_out);
    MCMC_HIP_CHECK(hipGetLastError());
    MCMC_HIP_CHECK(hipFreeAsync(d_mean, s));
    return MCMC_OK;
}

#endif // !__HIPCC_RTC__

} // namespace mcmc
