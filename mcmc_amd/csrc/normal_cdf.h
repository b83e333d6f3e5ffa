// Device-side normal CDF / quantile for the truncated-normal inverse-CDF draw.
//
// Replaces R::pnorm(q, mean, sd, 1, 0) and R::qnorm(p, 0, 1, 1, 0) as called at
// tmvrnormGibbs.cpp:55-56,61,130-132.  Both follow the published algorithms R's nmath uses
// (W. J. Cody 1969 rational Chebyshev approximations; Wichura 1988 AS241 PPND16) with the
// same branch points, so a draw matches the reference to rounding wherever the reference
// formula is well conditioned.
#pragma once
#include <hip/hip_runtime.h>

#include "fastmath.h"

namespace mcmc {

// Horner step with the coefficient delivered through a scalar register pair at its use (fastmath.h: sconst)
__device__ __forceinline__ double hstep(double acc, double x, double c) { return fma(acc, x, sconst(c, acc)); }

// The rational-function coefficients of pnorm / qnorm live in constant memory and reach the Horner chains
// through wide scalar loads (one s_load_dwordx8/x16 brings four / eight of them) instead of two s_mov_b32
// literals each: in the one-chain-per-lane Gibbs kernel the scalar unit, not the vector unit, was the busier
// one (ISA count: 909 scalar against 568 vector instructions, most of them coefficient moves).  Every range
// starts on a 64-byte boundary.  The pointer goes through an empty asm so that the loads stay inside their
// branch (hoisted out of the sweep loop they would fill the scalar register file and spill to VGPR lanes).
enum CdfTab : int { CT_P1 = 0, CT_P2 = 16, CT_P3 = 40, CT_Q1 = 56, CT_Q2 = 72, CT_Q3 = 88, CT_EXP = 104, CT_LOG = 120, CT_SIZE = 136 };
static __constant__ __attribute__((aligned(64))) double kCdfTab[CT_SIZE] = {
    // CT_P1: Cody |x| <= 0.67448975, numerator a (5, leading first) then denominator b (4)
    0.065682337918207449113, 2.2352520354606839287, 161.02823106855587881, 1067.6894854603709582,
    18154.981253343561249, 47.20258190468824187, 976.09855173777669322, 10260.932208618978205,
    45507.789335026729956, 0, 0, 0, 0, 0, 0, 0,
    // CT_P2: |x| <= sqrt(32), numerator c (9) then denominator d (8)
    1.0765576773720192317e-8, 0.39894151208813466764, 8.8831497943883759412, 93.506656132177855979,
    597.27027639480026226, 2494.5375852903726711, 6848.1904505362823326, 11602.651437647350124,
    9842.7148383839780218, 22.266688044328115691, 235.38790178262499861, 1519.377599407554805,
    6485.558298266760755, 18615.571640885098091, 34900.952721145977266, 38912.003286093271411,
    19685.429676859990727, 0, 0, 0, 0, 0, 0, 0,
    // CT_P3: tails, numerator p (6), denominator q (5), 1/sqrt(2 pi)
    0.02307344176494017303, 0.21589853405795699, 0.1274011611602473639, 0.022235277870649807,
    0.001421619193227893466, 2.9112874951168792e-5, 1.28426009614491121, 0.468238212480865118,
    0.0659881378689285515, 0.00378239633202758244, 7.29751555083966205e-5, 0.398942280401432677939946059934,
    0, 0, 0, 0,
    // CT_Q1: AS241 |q| <= 0.425, numerator (8) then denominator (7)
    2509.0809287301226727, 33430.575583588128105, 67265.770927008700853, 45921.953931549871457,
    13731.693765509461125, 1971.5909503065514427, 133.14166789178437745, 3.387132872796366608,
    5226.495278852854561, 28729.085735721942674, 39307.89580009271061, 21213.794301586595867,
    5394.1960214247511077, 687.1870074920579083, 42.313330701600911252, 0,
    // CT_Q2: r <= 5
    7.7454501427834140764e-4, .0227238449892691845833, .24178072517745061177, 1.27045825245236838258,
    3.64784832476320460504, 5.7694972214606914055, 4.6303378461565452959, 1.42343711074968357734,
    1.05075007164441684324e-9, 5.475938084995344946e-4, .0151986665636164571966, .14810397642748007459,
    .68976733498510000455, 1.6763848301838038494, 2.05319162663775882187, 0,
    // CT_Q3: r > 5
    2.01033439929228813265e-7, 2.71155556874348757815e-5, .0012426609473880784386, .026532189526576123093,
    .29656057182850489123, 1.7848265399172913358, 5.4637849111641143699, 6.6579046435011037772,
    2.04426310338993978564e-15, 1.4215117583164458887e-7, 1.8463183175100546818e-5, 7.868691311456132591e-4,
    .0148753612908506148525, .13692988092273580531, .59983220655588793769, 0,
    // CT_EXP (fastmath.h fm_exp): log2(e), ln2 high / low, 1/13! .. 1/3!
    1.4426950408889634074, 6.93147180369123816490e-01, 1.90821492927058770002e-10, 1.6059043836821613e-10,
    2.08767569878681e-09, 2.505210838544172e-08, 2.755731922398589e-07, 2.7557319223985893e-06,
    2.48015873015873e-05, 0.0001984126984126984, 0.001388888888888889, 0.008333333333333333,
    0.041666666666666664, 0.16666666666666666, 0, 0,
    // CT_LOG (fastmath.h fm_log): sqrt(1/2), the two interleaved polynomials, ln2 high / low
    0.70710678118654752440, 1.531383769920937332e-01, 2.222219843214978396e-01, 3.999999999940941908e-01,
    1.479819860511658591e-01, 1.818357216161805012e-01, 2.857142874366239149e-01, 6.666666666666735130e-01,
    6.93147180369123816490e-01, 1.90821492927058770002e-10, 0, 0, 0, 0, 0, 0};

// N (a multiple of 8) coefficients of the range starting at `first`, not before `after` is known
template <int N>
using CdfCoef = ScalarCoef<N>;
template <int N>
__device__ __forceinline__ CdfCoef<N> cdf_tab(int first, double after)
{
    const MCMC_CONST_AS double *t = (const MCMC_CONST_AS double *)kCdfTab + first;
    asm volatile("" : "+s"(t) : "v"(after));
    return load_coef<N>(t);
}

// fm_exp / fm_log<false> of fastmath.h, operation for operation, with the coefficients from the table
__device__ __forceinline__ double cdf_exp(double x)
{
    const double xc = fmin(fmax(x, -1100.0), 1100.0);
    const CdfCoef<16> t = cdf_tab<16>(CT_EXP, xc);
    const double kd = rint(xc * t[0]);
    double r = fma(kd, -t[1], xc);
    r = fma(kd, -t[2], r);
    double p = vreg(t[3]);
#pragma unroll
    for (int k = 4; k < 14; ++k) p = fma_sc(p, r, t[k]);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    double res = ldexp(p, (int)kd);
    if (x != x) res = x;
    return res;
}
__device__ __forceinline__ double cdf_log(double x)
{
    int e;
    double m = frexp(x, &e);
    const CdfCoef<16> t = cdf_tab<16>(CT_LOG, m);
    const bool lo = m < t[0];
    m = lo ? m + m : m;
    e = lo ? e - 1 : e;
    const double f = m - 1.0;
    const double s = f * fm_rcp(2.0 + f);
    const double z = s * s;
    const double w = z * z;
    double t1 = fma_sc(w, vreg(t[1]), t[2]);
    t1 = fma_sc(w, t1, t[3]);
    t1 *= w;
    double t2 = fma_sc(w, vreg(t[4]), t[5]);
    t2 = fma_sc(w, t2, t[6]);
    t2 = fma_sc(w, t2, t[7]);
    t2 *= z;
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)e;
    return dk * t[8] - ((hfsq - (s * (hfsq + R) + dk * t[9])) - f);
}

// lower-tail standard normal CDF, Cody's three ranges (|x| <= 0.67448975, <= sqrt(32), tails
// cut at -37.5193 / +8.2924 exactly as R does for the lower tail).  Cody writes the rational functions as
// repeated (acc + c) * x; here they are the same polynomials in Horner form, one FMA per coefficient.
__device__ __forceinline__ double pnorm_std(double x)
{
    const double y = fabs(x);
    if (y <= 0.67448975) { // for |x| < 1.1e-16 (where R sets both sums to 0) x^2 underflows harmlessly: same value
        const double xsq = x * x;
        const CdfCoef<16> t = cdf_tab<16>(CT_P1, xsq);
        double xnum = fma_sc(vreg(t[0]), xsq, t[1]);
        xnum = fma_sc(xnum, xsq, t[2]);
        xnum = fma_sc(xnum, xsq, t[3]);
        xnum = fma_sc(xnum, xsq, t[4]);
        double xden = xsq + t[5];
        xden = fma_sc(xden, xsq, t[6]);
        xden = fma_sc(xden, xsq, t[7]);
        xden = fma_sc(xden, xsq, t[8]);
        return 0.5 + x * fm_div(xnum, xden);
    }
    double temp;
    if (y <= 5.656854249492380195206754896838) {
        const CdfCoef<24> t = cdf_tab<24>(CT_P2, y);
        double xnum = fma_sc(vreg(t[0]), y, t[1]);
#pragma unroll
        for (int k = 2; k < 9; ++k) xnum = fma_sc(xnum, y, t[k]);
        double xden = y + t[9];
#pragma unroll
        for (int k = 10; k < 17; ++k) xden = fma_sc(xden, y, t[k]);
        temp = fm_div(xnum, xden);
    } else if (-37.5193 < x && x < 8.2924) {
        const double xsq = fm_rcp(x * x);
        const CdfCoef<16> t = cdf_tab<16>(CT_P3, xsq);
        double xnum = fma_sc(vreg(t[0]), xsq, t[1]);
#pragma unroll
        for (int k = 2; k < 6; ++k) xnum = fma_sc(xnum, xsq, t[k]);
        double xden = xsq + t[6];
#pragma unroll
        for (int k = 7; k < 11; ++k) xden = fma_sc(xden, xsq, t[k]);
        temp = xsq * fm_div(xnum, xden);
        temp = fm_div(t[11] - temp, y);
    } else {
        return (x > 0) ? 1.0 : 0.0;
    }
    // exp(-y^2/2): Cody/R split y^2 as trunc(16y)/16 squared plus a remainder and multiply two exp's so
    // that the rounding of y*y is not amplified; here y*y is formed exactly as hi + lo with one FMA and
    // the lo part applied to first order (|lo| < 2^-43), which is as accurate and needs one exp.
    const double hi = y * y;
    const double lo = fma(y, y, -hi);
    const double E = cdf_exp(-0.5 * hi);
    const double cum = fma(-0.5 * lo, E, E) * temp;
    return (x > 0.) ? 1.0 - cum : cum;
}

// R::pnorm(x, mu, sigma, lower = 1, log = 0)
__device__ inline double pnorm(double x, double mu, double sigma)
{
    if (isnan(x) || isnan(mu) || isnan(sigma)) return x + mu + sigma;
    if (!isfinite(x) && mu == x) return __builtin_nan("");
    if (sigma <= 0) {
        if (sigma < 0) return __builtin_nan("");
        return (x < mu) ? 0.0 : 1.0;
    }
    const double p = (x - mu) / sigma;
    if (!isfinite(p)) return (x < mu) ? 0.0 : 1.0;
    return pnorm_std(p);
}

// R::qnorm(p, 0, 1, lower = 1, log = 0): AS241
__device__ __forceinline__ double qnorm_std(double p)
{
    if (isnan(p)) return p;
    if (p < 0 || p > 1) return __builtin_nan("");
    if (p == 0) return -__builtin_inf();
    if (p == 1) return __builtin_inf();
    const double q = p - 0.5;
    if (fabs(q) <= 0.425) {
        const double r = .180625 - q * q;
        const CdfCoef<16> t = cdf_tab<16>(CT_Q1, r);
        double num = fma_sc(vreg(t[0]), r, t[1]);
#pragma unroll
        for (int k = 2; k < 8; ++k) num = fma_sc(num, r, t[k]);
        double den = fma_sc(vreg(t[8]), r, t[9]);
#pragma unroll
        for (int k = 10; k < 15; ++k) den = fma_sc(den, r, t[k]);
        den = fma(den, r, 1.0);
        return q * num * fm_rcp(den);
    }
    double r = (q < 0) ? p : 1.0 - p;
    r = fm_sqrt(-cdf_log(r));
    double val;
    if (r <= 5.) {
        r += -1.6;
        const CdfCoef<16> t = cdf_tab<16>(CT_Q2, r);
        double num = fma_sc(vreg(t[0]), r, t[1]);
#pragma unroll
        for (int k = 2; k < 8; ++k) num = fma_sc(num, r, t[k]);
        double den = fma_sc(vreg(t[8]), r, t[9]);
#pragma unroll
        for (int k = 10; k < 15; ++k) den = fma_sc(den, r, t[k]);
        den = fma(den, r, 1.0);
        val = num * fm_rcp(den);
    } else {
        r += -5.;
        const CdfCoef<16> t = cdf_tab<16>(CT_Q3, r);
        double num = fma_sc(vreg(t[0]), r, t[1]);
#pragma unroll
        for (int k = 2; k < 8; ++k) num = fma_sc(num, r, t[k]);
        double den = fma_sc(vreg(t[8]), r, t[9]);
#pragma unroll
        for (int k = 10; k < 15; ++k) den = fma_sc(den, r, t[k]);
        den = fma(den, r, 1.0);
        val = num * fm_rcp(den);
    }
    return (q < 0.0) ? -val : val;
}

// One truncated-normal inverse-CDF draw on [lower, upper] for N(mu_i, sd^2), uniform u.
//   TAIL_SAFE = false: the reference statement verbatim (tmvrnormGibbs.cpp:130-132)
//       Fa = pnorm(lower), Fb = pnorm(upper), x = mu_i + sd * qnorm(u*(Fb-Fa)+Fa)
//   TAIL_SAFE = true: when the whole interval lies above mu_i, the same draw is taken in
//       the mirrored (lower) tail, x = mu_i - sd * qnorm(Qa - u*(Qa-Qb)), Qa = Phi((mu_i-lower)/sd),
//       Qb = Phi((mu_i-upper)/sd); algebraically identical, but Qa, Qb do not round to 1, so bounds
//       more than ~8.3 sd out (where the reference returns +Inf) still give finite draws.
template <bool TAIL_SAFE>
__device__ __forceinline__ double rtnorm_draw(double u, double mu_i, double sd, double lower, double upper)
{
    if (TAIL_SAFE && lower > mu_i) {
        const double Qa = pnorm_std((mu_i - lower) / sd); // P(X > lower)
        const double Qb = pnorm_std((mu_i - upper) / sd); // P(X > upper); upper = +Inf -> 0
        return mu_i - sd * qnorm_std(Qa - u * (Qa - Qb));
    }
    const double Fa = pnorm(lower, mu_i, sd);
    const double Fb = pnorm(upper, mu_i, sd);
    return mu_i + sd * qnorm_std(u * (Fb - Fa) + Fa);
}

// Same draw with the standardisation done by a reciprocal: a = (lower - mu_i) * (1/sd).  Used by
// the Gibbs kernels, where sd > 0 is guaranteed by the host and 1/sd is a shared parameter;
// -Inf / +Inf bounds fall through pnorm_std to exactly 0 / 1 as in R's pnorm.
template <bool TAIL_SAFE>
__device__ __forceinline__ double rtnorm_draw_std(double u, double mu_i, double sd, double inv_sd, double lower, double upper)
{
    const double a = (lower - mu_i) * inv_sd;
    const double b = (upper - mu_i) * inv_sd;
    if (TAIL_SAFE && a > 0.0) {
        const double Qa = pnorm_std(-a);
        const double Qb = pnorm_std(-b);
        return mu_i - sd * qnorm_std(Qa - u * (Qa - Qb));
    }
    const double Fa = pnorm_std(a);
    const double Fb = pnorm_std(b);
    return mu_i + sd * qnorm_std(u * (Fb - Fa) + Fa);
}

} // namespace mcmc
