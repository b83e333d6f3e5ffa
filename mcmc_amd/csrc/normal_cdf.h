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

// lower-tail standard normal CDF, Cody's three ranges (|x| <= 0.67448975, <= sqrt(32), tails
// cut at -37.5193 / +8.2924 exactly as R does for the lower tail).
__device__ inline double pnorm_std(double x)
{
    const double y = fabs(x);
    if (y <= 0.67448975) {
        double xnum = 0.0, xden = 0.0;
        if (y > 1.1102230246251565e-16) {
            const double xsq = x * x;
            xnum = 0.065682337918207449113 * xsq;
            xden = xsq;
            xnum = (xnum + 2.2352520354606839287) * xsq;
            xden = (xden + 47.20258190468824187) * xsq;
            xnum = (xnum + 161.02823106855587881) * xsq;
            xden = (xden + 976.09855173777669322) * xsq;
            xnum = (xnum + 1067.6894854603709582) * xsq;
            xden = (xden + 10260.932208618978205) * xsq;
        }
        const double temp = x * fm_div(xnum + 18154.981253343561249, xden + 45507.789335026729956);
        return 0.5 + temp;
    }
    double temp;
    if (y <= 5.656854249492380195206754896838) {
        double xnum = 1.0765576773720192317e-8 * y;
        double xden = y;
        xnum = (xnum + 0.39894151208813466764) * y;
        xden = (xden + 22.266688044328115691) * y;
        xnum = (xnum + 8.8831497943883759412) * y;
        xden = (xden + 235.38790178262499861) * y;
        xnum = (xnum + 93.506656132177855979) * y;
        xden = (xden + 1519.377599407554805) * y;
        xnum = (xnum + 597.27027639480026226) * y;
        xden = (xden + 6485.558298266760755) * y;
        xnum = (xnum + 2494.5375852903726711) * y;
        xden = (xden + 18615.571640885098091) * y;
        xnum = (xnum + 6848.1904505362823326) * y;
        xden = (xden + 34900.952721145977266) * y;
        xnum = (xnum + 11602.651437647350124) * y;
        xden = (xden + 38912.003286093271411) * y;
        temp = fm_div(xnum + 9842.7148383839780218, xden + 19685.429676859990727);
    } else if (-37.5193 < x && x < 8.2924) {
        const double xsq = fm_rcp(x * x);
        double xnum = 0.02307344176494017303 * xsq;
        double xden = xsq;
        xnum = (xnum + 0.21589853405795699) * xsq;
        xden = (xden + 1.28426009614491121) * xsq;
        xnum = (xnum + 0.1274011611602473639) * xsq;
        xden = (xden + 0.468238212480865118) * xsq;
        xnum = (xnum + 0.022235277870649807) * xsq;
        xden = (xden + 0.0659881378689285515) * xsq;
        xnum = (xnum + 0.001421619193227893466) * xsq;
        xden = (xden + 0.00378239633202758244) * xsq;
        temp = xsq * fm_div(xnum + 2.9112874951168792e-5, xden + 7.29751555083966205e-5);
        temp = fm_div(0.398942280401432677939946059934 - temp, y);
    } else {
        return (x > 0) ? 1.0 : 0.0;
    }
    // exp(-y^2/2): Cody/R split y^2 as trunc(16y)/16 squared plus a remainder and multiply two exp's so
    // that the rounding of y*y is not amplified; here y*y is formed exactly as hi + lo with one FMA and
    // the lo part applied to first order (|lo| < 2^-43), which is as accurate and needs one exp.
    const double hi = y * y;
    const double lo = fma(y, y, -hi);
    const double E = exp(-0.5 * hi);
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
__device__ inline double qnorm_std(double p)
{
    if (isnan(p)) return p;
    if (p < 0 || p > 1) return __builtin_nan("");
    if (p == 0) return -__builtin_inf();
    if (p == 1) return __builtin_inf();
    const double q = p - 0.5;
    if (fabs(q) <= 0.425) {
        const double r = .180625 - q * q;
        return q * (((((((r * 2509.0809287301226727 + 33430.575583588128105) * r + 67265.770927008700853) * r +
                        45921.953931549871457) * r + 13731.693765509461125) * r + 1971.5909503065514427) * r +
                     133.14166789178437745) * r + 3.387132872796366608) *
               fm_rcp(((((((r * 5226.495278852854561 + 28729.085735721942674) * r + 39307.89580009271061) * r +
                    21213.794301586595867) * r + 5394.1960214247511077) * r + 687.1870074920579083) * r +
                 42.313330701600911252) * r + 1.);
    }
    double r = (q < 0) ? p : 1.0 - p;
    r = fm_sqrt(-fm_log<false>(r));
    double val;
    if (r <= 5.) {
        r += -1.6;
        val = (((((((r * 7.7454501427834140764e-4 + .0227238449892691845833) * r + .24178072517745061177) * r +
                   1.27045825245236838258) * r + 3.64784832476320460504) * r + 5.7694972214606914055) * r +
                4.6303378461565452959) * r + 1.42343711074968357734) *
              fm_rcp(((((((r * 1.05075007164441684324e-9 + 5.475938084995344946e-4) * r + .0151986665636164571966) * r +
                   .14810397642748007459) * r + .68976733498510000455) * r + 1.6763848301838038494) * r +
                2.05319162663775882187) * r + 1.);
    } else {
        r += -5.;
        val = (((((((r * 2.01033439929228813265e-7 + 2.71155556874348757815e-5) * r + .0012426609473880784386) * r +
                   .026532189526576123093) * r + .29656057182850489123) * r + 1.7848265399172913358) * r +
                5.4637849111641143699) * r + 6.6579046435011037772) *
              fm_rcp(((((((r * 2.04426310338993978564e-15 + 1.4215117583164458887e-7) * r + 1.8463183175100546818e-5) * r +
                   7.868691311456132591e-4) * r + .0148753612908506148525) * r + .13692988092273580531) * r +
                .59983220655588793769) * r + 1.);
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
__device__ inline double rtnorm_draw(double u, double mu_i, double sd, double lower, double upper)
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
__device__ inline double rtnorm_draw_std(double u, double mu_i, double sd, double inv_sd, double lower, double upper)
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
