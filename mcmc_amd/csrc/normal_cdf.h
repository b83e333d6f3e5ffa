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

// lower-tail standard normal CDF, Cody's three ranges (|x| <= 0.67448975, <= sqrt(32), tails
// cut at -37.5193 / +8.2924 exactly as R does for the lower tail).  Cody writes the rational functions as
// repeated (acc + c) * x; here they are the same polynomials in Horner form, one FMA per coefficient.
__device__ __forceinline__ double pnorm_std(double x)
{
    const double y = fabs(x);
    if (y <= 0.67448975) { // for |x| < 1.1e-16 (where R sets both sums to 0) x^2 underflows harmlessly: same value
        const double xsq = x * x;
        double xnum = fma(sconst(0.065682337918207449113, xsq), xsq, sconst(2.2352520354606839287, xsq));
        xnum = hstep(xnum, xsq, 161.02823106855587881);
        xnum = hstep(xnum, xsq, 1067.6894854603709582);
        xnum = hstep(xnum, xsq, 18154.981253343561249);
        double xden = xsq + sconst(47.20258190468824187, xsq);
        xden = hstep(xden, xsq, 976.09855173777669322);
        xden = hstep(xden, xsq, 10260.932208618978205);
        xden = hstep(xden, xsq, 45507.789335026729956);
        return 0.5 + x * fm_div(xnum, xden);
    }
    double temp;
    if (y <= 5.656854249492380195206754896838) {
        double xnum = fma(sconst(1.0765576773720192317e-8, y), y, sconst(0.39894151208813466764, y));
        xnum = hstep(xnum, y, 8.8831497943883759412);
        xnum = hstep(xnum, y, 93.506656132177855979);
        xnum = hstep(xnum, y, 597.27027639480026226);
        xnum = hstep(xnum, y, 2494.5375852903726711);
        xnum = hstep(xnum, y, 6848.1904505362823326);
        xnum = hstep(xnum, y, 11602.651437647350124);
        xnum = hstep(xnum, y, 9842.7148383839780218);
        double xden = y + sconst(22.266688044328115691, y);
        xden = hstep(xden, y, 235.38790178262499861);
        xden = hstep(xden, y, 1519.377599407554805);
        xden = hstep(xden, y, 6485.558298266760755);
        xden = hstep(xden, y, 18615.571640885098091);
        xden = hstep(xden, y, 34900.952721145977266);
        xden = hstep(xden, y, 38912.003286093271411);
        xden = hstep(xden, y, 19685.429676859990727);
        temp = fm_div(xnum, xden);
    } else if (-37.5193 < x && x < 8.2924) {
        const double xsq = fm_rcp(x * x);
        double xnum = fma(sconst(0.02307344176494017303, xsq), xsq, sconst(0.21589853405795699, xsq));
        xnum = hstep(xnum, xsq, 0.1274011611602473639);
        xnum = hstep(xnum, xsq, 0.022235277870649807);
        xnum = hstep(xnum, xsq, 0.001421619193227893466);
        xnum = hstep(xnum, xsq, 2.9112874951168792e-5);
        double xden = xsq + sconst(1.28426009614491121, xsq);
        xden = hstep(xden, xsq, 0.468238212480865118);
        xden = hstep(xden, xsq, 0.0659881378689285515);
        xden = hstep(xden, xsq, 0.00378239633202758244);
        xden = hstep(xden, xsq, 7.29751555083966205e-5);
        temp = xsq * fm_div(xnum, xden);
        temp = fm_div(sconst(0.398942280401432677939946059934, temp) - temp, y);
    } else {
        return (x > 0) ? 1.0 : 0.0;
    }
    // exp(-y^2/2): Cody/R split y^2 as trunc(16y)/16 squared plus a remainder and multiply two exp's so
    // that the rounding of y*y is not amplified; here y*y is formed exactly as hi + lo with one FMA and
    // the lo part applied to first order (|lo| < 2^-43), which is as accurate and needs one exp.
    const double hi = y * y;
    const double lo = fma(y, y, -hi);
    const double E = fm_exp(-0.5 * hi);
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
        double num = fma(sconst(2509.0809287301226727, r), r, sconst(33430.575583588128105, r));
        num = hstep(num, r, 67265.770927008700853);
        num = hstep(num, r, 45921.953931549871457);
        num = hstep(num, r, 13731.693765509461125);
        num = hstep(num, r, 1971.5909503065514427);
        num = hstep(num, r, 133.14166789178437745);
        num = hstep(num, r, 3.387132872796366608);
        double den = fma(sconst(5226.495278852854561, r), r, sconst(28729.085735721942674, r));
        den = hstep(den, r, 39307.89580009271061);
        den = hstep(den, r, 21213.794301586595867);
        den = hstep(den, r, 5394.1960214247511077);
        den = hstep(den, r, 687.1870074920579083);
        den = hstep(den, r, 42.313330701600911252);
        den = fma(den, r, 1.0);
        return q * num * fm_rcp(den);
    }
    double r = (q < 0) ? p : 1.0 - p;
    r = fm_sqrt(-fm_log<false>(r));
    double val;
    if (r <= 5.) {
        r += -1.6;
        double num = fma(sconst(7.7454501427834140764e-4, r), r, sconst(.0227238449892691845833, r));
        num = hstep(num, r, .24178072517745061177);
        num = hstep(num, r, 1.27045825245236838258);
        num = hstep(num, r, 3.64784832476320460504);
        num = hstep(num, r, 5.7694972214606914055);
        num = hstep(num, r, 4.6303378461565452959);
        num = hstep(num, r, 1.42343711074968357734);
        double den = fma(sconst(1.05075007164441684324e-9, r), r, sconst(5.475938084995344946e-4, r));
        den = hstep(den, r, .0151986665636164571966);
        den = hstep(den, r, .14810397642748007459);
        den = hstep(den, r, .68976733498510000455);
        den = hstep(den, r, 1.6763848301838038494);
        den = hstep(den, r, 2.05319162663775882187);
        den = fma(den, r, 1.0);
        val = num * fm_rcp(den);
    } else {
        r += -5.;
        double num = fma(sconst(2.01033439929228813265e-7, r), r, sconst(2.71155556874348757815e-5, r));
        num = hstep(num, r, .0012426609473880784386);
        num = hstep(num, r, .026532189526576123093);
        num = hstep(num, r, .29656057182850489123);
        num = hstep(num, r, 1.7848265399172913358);
        num = hstep(num, r, 5.4637849111641143699);
        num = hstep(num, r, 6.6579046435011037772);
        double den = fma(sconst(2.04426310338993978564e-15, r), r, sconst(1.4215117583164458887e-7, r));
        den = hstep(den, r, 1.8463183175100546818e-5);
        den = hstep(den, r, 7.868691311456132591e-4);
        den = hstep(den, r, .0148753612908506148525);
        den = hstep(den, r, .13692988092273580531);
        den = hstep(den, r, .59983220655588793769);
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
