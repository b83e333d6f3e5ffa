// Device-side target log-densities: the replacement for the R closure `target` that
// adaptMCMC / rwMCMC call twice per iteration (adaptMCMC_source.cpp:97,158; :246,277).
#pragma once
#include <hip/hip_runtime.h>

#include "mcmc_amd.h"
#include "fastmath.h"

// A user target exists only inside a run-time compiled module (rtc.hip defines MCMC_USER_D and
// MCMC_USER_NP ahead of the kernel sources and appends the user's definition of this function).
#ifdef MCMC_USER_D
__device__ double mcmc_user_target(const double (&x)[MCMC_USER_D], const double *params);
#else
#define MCMC_USER_NP 0
#endif

namespace mcmc {

// parameter block sizes (doubles) per kind, for a given d
__host__ __device__ constexpr int target_param_count(int kind, int d)
{
    return kind == MCMC_TARGET_GAUSSIAN_PREC ? d + d * d
                                             : (kind == MCMC_TARGET_BANANA ? 1 : (kind == MCMC_TARGET_USER ? MCMC_USER_NP : 0));
}

template <int KIND, int D>
struct Target;

// -0.5 (x-m)^T H (x-m); prm = m[D], H[D*D] column-major.  Loop order matches the oracle:
// q = sum_i v_i * (sum_j H_ij v_j).
template <int D>
struct Target<MCMC_TARGET_GAUSSIAN_PREC, D> {
    template <class Prm>
    __device__ static inline double eval(const double (&x)[D], const Prm &prm)
    {
        double v[D];
#pragma unroll
        for (int k = 0; k < D; ++k) v[k] = x[k] - prm[k];
        double q = 0.0;
#pragma unroll
        for (int i = 0; i < D; ++i) {
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < D; ++j) s += prm[D + j * D + i] * v[j];
            q += v[i] * s;
        }
        return -0.5 * q;
    }
};

// adaptMCMC_analysis.R:115-129
template <>
struct Target<MCMC_TARGET_GOLDSTEIN_PRICE, 2> {
    template <class Prm>
    __device__ static inline double eval(const double (&xx)[2], const Prm &)
    {
        const double x1 = xx[0], x2 = xx[1];
        const double s = x1 + x2 + 1;
        const double fact1a = s * s;
        const double fact1b = 19 - 14 * x1 + 3 * (x1 * x1) - 14 * x2 + 6 * x1 * x2 + 3 * (x2 * x2);
        const double fact1 = 1 + fact1a * fact1b;
        const double t = 2 * x1 - 3 * x2;
        const double fact2a = t * t;
        const double fact2b = 18 - 32 * x1 + 12 * (x1 * x1) + 48 * x2 - 36 * x1 * x2 + 27 * (x2 * x2);
        const double fact2 = 30 + fact2a * fact2b;
        return -fm_log<true>(fact1 * fact2);
    }
};

// adaptMCMC_analysis.R:19-22; prm[0] = B
template <>
struct Target<MCMC_TARGET_BANANA, 2> {
    template <class Prm>
    __device__ static inline double eval(const double (&x)[2], const Prm &prm)
    {
        const double B = prm[0];
        const double u = x[1] + B * (x[0] * x[0]) - 100 * B;
        return -(x[0] * x[0]) / 200 - 1.0 / 2 * (u * u);
    }
};

#ifdef MCMC_USER_D
// prm is a view of the parameter block in LDS (PrmView / PackedPrmView: a pointer named p)
template <>
struct Target<MCMC_TARGET_USER, MCMC_USER_D> {
    template <class Prm>
    __device__ static inline double eval(const double (&x)[MCMC_USER_D], const Prm &prm)
    {
        return ::mcmc_user_target(x, prm.p);
    }
};
#endif

// log-density at n points, x as [D][n]: mcmc_target_eval_batch
template <int D, int KIND>
__global__ __launch_bounds__(256) void target_eval_kernel(int64_t n, const double *__restrict__ x,
                                                          const double *__restrict__ prm, double *__restrict__ out)
{
    struct View {
        const double *p;
        __device__ double operator[](int i) const { return p[i]; }
    };
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= n) return;
    double v[D];
#pragma unroll
    for (int k = 0; k < D; ++k) v[k] = x[(int64_t)k * n + c];
    out[c] = Target<KIND, D>::eval(v, View{prm});
}

} // namespace mcmc
