/*
 * mcmc_oracle.c -- CPU restatement of the slwu89/MCMC hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the *checker*: only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  Nothing under mcmc_amd/ (the product)
 * imports, links or calls it, and the product fails loudly without its HIP library.
 *
 * It follows the reference statement by statement (citations are into /root/reference):
 *   update_sigma      adaptMCMC_source.cpp:42-45
 *   mvrnorm_pdf       adaptMCMC_source.cpp:48-64   (un-logged density, then log)
 *   adaptMCMC         adaptMCMC_source.cpp:75-219
 *   rwMCMC            adaptMCMC_source.cpp:230-316
 *   sub1/sub2/negSub* tmvrnormGibbs.cpp:12-37
 *   rtnorm_gibbs      tmvrnormGibbs.cpp:49-70
 *   rtmvnorm_gibbs    tmvrnormGibbs.cpp:91-142
 *
 * Third-party arithmetic that is NOT under /root/reference is restated from the
 * published algorithms (all version-unpinned upstream):
 *   arma::chol  -> LAPACK dpotf2 'U' (row-by-row upper factor, scale by reciprocal pivot)
 *   arma::det   -> LU with partial pivoting (dgetf2), product of the diagonal
 *   arma::inv   -> LU with partial pivoting + column solves (dgetri equivalent)
 *   R::pnorm    -> W. J. Cody (1969) rational Chebyshev, as arranged in R nmath pnorm_both
 *   R::qnorm    -> Wichura (1988) AS241 PPND16; coefficients cross-checked against the
 *                  copy the reference carries at deprecated/gibbs_tmvtgaussian.cpp:125-156
 *   std::mt19937 + libstdc++ normal_distribution (Marsaglia polar, cached second value)
 *                  and uniform_real_distribution (generate_canonical<double,53>)
 *
 * PARITY UNPINNED: the reference stores no expected numeric output for any of these
 * functions and cannot be built here (no R / Rcpp / Armadillo).
 * The only hand-derivable known answer is the update_sigma case of
 * deprecated/adaptMCMC.R:146-155 ([[4.01,0.06],[0.06,0.44]]), which tests/ checks.
 * pnorm/qnorm/chol/det/inv are validated against scipy/numpy in tests/test_oracle.py.
 *
 * Random numbers are INJECTED: the samplers take the N(0,1) and U(0,1) streams as
 * arrays, in the order the reference draws them (d normals then one uniform per
 * Metropolis iteration, adaptMCMC_source.cpp:149-152,174; U[j*d+i] for Gibbs,
 * tmvrnormGibbs.cpp:97,132-133).  orc_stdlib_streams() reproduces what a libstdc++
 * build of the reference would draw for a given seedMH.
 *
 * Matrices are column-major (Armadillo / R convention) unless stated otherwise.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_OK 0
#define ORC_CHOL_FAILED 1   /* arma::chol would throw (adaptMCMC_source.cpp:153) */
#define ORC_BAD_ARG 2
#define ORC_SINGULAR 3

/* ------------------------------------------------------------------------- */
/* R::pnorm(x, mu, sigma, lower=1, log=0): Cody's algorithm                   */
/* ------------------------------------------------------------------------- */
static const double CODY_A[5] = {
    2.2352520354606839287, 161.02823106855587881, 1067.6894854603709582,
    18154.981253343561249, 0.065682337918207449113};
static const double CODY_B[4] = {
    47.20258190468824187, 976.09855173777669322, 10260.932208618978205,
    45507.789335026729956};
static const double CODY_C[9] = {
    0.39894151208813466764, 8.8831497943883759412, 93.506656132177855979,
    597.27027639480026226, 2494.5375852903726711, 6848.1904505362823326,
    11602.651437647350124, 9842.7148383839780218, 1.0765576773720192317e-8};
static const double CODY_D[8] = {
    22.266688044328115691, 235.38790178262499861, 1519.377599407554805,
    6485.558298266760755, 18615.571640885098091, 34900.952721145977266,
    38912.003286093271411, 19685.429676859990727};
static const double CODY_P[6] = {
    0.21589853405795699, 0.1274011611602473639, 0.022235277870649807,
    0.001421619193227893466, 2.9112874951168792e-5, 0.02307344176494017303};
static const double CODY_Q[5] = {
    1.28426009614491121, 0.468238212480865118, 0.0659881378689285515,
    0.00378239633202758244, 7.29751555083966205e-5};

#define ORC_SQRT32 5.656854249492380195206754896838
#define ORC_1_SQRT_2PI 0.398942280401432677939946059934

/* lower-tail standard normal CDF of z (the i_tail = 0, log_p = 0 arm of pnorm_both) */
static double cody_lower(double x)
{
    const double eps = 2.220446049250313e-16 * 0.5;
    double y = fabs(x), xnum, xden, temp, xsq, del, cum, ccum;
    int i;
    if (y <= 0.67448975) {
        if (y > eps) {
            xsq = x * x;
            xnum = CODY_A[4] * xsq;
            xden = xsq;
            for (i = 0; i < 3; ++i) {
                xnum = (xnum + CODY_A[i]) * xsq;
                xden = (xden + CODY_B[i]) * xsq;
            }
        } else {
            xnum = xden = 0.0;
        }
        temp = x * (xnum + CODY_A[3]) / (xden + CODY_B[3]);
        return 0.5 + temp;
    }
    if (y <= ORC_SQRT32) {
        xnum = CODY_C[8] * y;
        xden = y;
        for (i = 0; i < 7; ++i) {
            xnum = (xnum + CODY_C[i]) * y;
            xden = (xden + CODY_D[i]) * y;
        }
        temp = (xnum + CODY_C[7]) / (xden + CODY_D[7]);
        xsq = trunc(y * 16) / 16;
        del = (y - xsq) * (y + xsq);
        cum = exp(-xsq * xsq * 0.5) * exp(-del * 0.5) * temp;
        ccum = 1.0 - cum;
        return (x > 0.) ? ccum : cum;
    }
    if (-37.5193 < x && x < 8.2924) {
        xsq = 1.0 / (x * x);
        xnum = CODY_P[5] * xsq;
        xden = xsq;
        for (i = 0; i < 4; ++i) {
            xnum = (xnum + CODY_P[i]) * xsq;
            xden = (xden + CODY_Q[i]) * xsq;
        }
        temp = xsq * (xnum + CODY_P[4]) / (xden + CODY_Q[4]);
        temp = (ORC_1_SQRT_2PI - temp) / y;
        xsq = trunc(x * 16) / 16;
        del = (x - xsq) * (x + xsq);
        cum = exp(-xsq * xsq * 0.5) * exp(-del * 0.5) * temp;
        ccum = 1.0 - cum;
        return (x > 0.) ? ccum : cum;
    }
    return (x > 0) ? 1.0 : 0.0;
}

/* R::pnorm(x, mu, sigma, 1, 0) -- used at tmvrnormGibbs.cpp:55-56,130-131 */
double orc_pnorm(double x, double mu, double sigma)
{
    double p;
    if (isnan(x) || isnan(mu) || isnan(sigma)) return x + mu + sigma;
    if (!isfinite(x) && mu == x) return NAN;
    if (sigma <= 0) {
        if (sigma < 0) return NAN;
        return (x < mu) ? 0.0 : 1.0;
    }
    p = (x - mu) / sigma;
    if (!isfinite(p)) return (x < mu) ? 0.0 : 1.0;
    return cody_lower(p);
}

/* ------------------------------------------------------------------------- */
/* R::qnorm(p, 0, 1, lower=1, log=0): Wichura AS241 (PPND16)                   */
/* ------------------------------------------------------------------------- */
double orc_qnorm(double p)
{
    double q, r, val;
    if (isnan(p)) return p;
    if (p < 0 || p > 1) return NAN;
    if (p == 0) return -INFINITY;
    if (p == 1) return INFINITY;
    q = p - 0.5;
    if (fabs(q) <= 0.425) {
        r = .180625 - q * q;
        val = q * (((((((r * 2509.0809287301226727 +
                         33430.575583588128105) * r + 67265.770927008700853) * r +
                       45921.953931549871457) * r + 13731.693765509461125) * r +
                     1971.5909503065514427) * r + 133.14166789178437745) * r +
                   3.387132872796366608) /
              (((((((r * 5226.495278852854561 +
                     28729.085735721942674) * r + 39307.89580009271061) * r +
                   21213.794301586595867) * r + 5394.1960214247511077) * r +
                 687.1870074920579083) * r + 42.313330701600911252) * r + 1.);
        return val;
    }
    r = (q < 0) ? p : 1.0 - p;
    r = sqrt(-log(r));
    if (r <= 5.) {
        r += -1.6;
        val = (((((((r * 7.7454501427834140764e-4 +
                     .0227238449892691845833) * r + .24178072517745061177) * r +
                   1.27045825245236838258) * r + 3.64784832476320460504) * r +
                 5.7694972214606914055) * r + 4.6303378461565452959) * r +
               1.42343711074968357734) /
              (((((((r * 1.05075007164441684324e-9 +
                     5.475938084995344946e-4) * r + .0151986665636164571966) * r +
                   .14810397642748007459) * r + .68976733498510000455) * r +
                 1.6763848301838038494) * r + 2.05319162663775882187) * r + 1.);
    } else {
        r += -5.;
        val = (((((((r * 2.01033439929228813265e-7 +
                     2.71155556874348757815e-5) * r + .0012426609473880784386) * r +
                   .026532189526576123093) * r + .29656057182850489123) * r +
                 1.7848265399172913358) * r + 5.4637849111641143699) * r +
               6.6579046435011037772) /
              (((((((r * 2.04426310338993978564e-15 +
                     1.4215117583164458887e-7) * r + 1.8463183175100546818e-5) * r +
                   7.868691311456132591e-4) * r + .0148753612908506148525) * r +
                 .13692988092273580531) * r + .59983220655588793769) * r + 1.);
    }
    if (q < 0.0) val = -val;
    return val;
}

void orc_pnorm_vec(int n, const double *x, double mu, double sigma, double *out)
{
    for (int i = 0; i < n; ++i) out[i] = orc_pnorm(x[i], mu, sigma);
}
void orc_qnorm_vec(int n, const double *p, double *out)
{
    for (int i = 0; i < n; ++i) out[i] = orc_qnorm(p[i]);
}

/* ------------------------------------------------------------------------- */
/* dense helpers (column-major, leading dimension = d)                         */
/* ------------------------------------------------------------------------- */
#define AT(M, r, c, d) ((M)[(size_t)(c) * (d) + (r)])

/* arma::chol(A): upper R with R^T R = A, reading only the upper triangle
 * (LAPACK dpotf2 'U': row j = (A[j,j+1:] - R[0:j,j]^T R[0:j,j+1:]) * (1/R[j,j])). */
int orc_chol_upper(int d, const double *A, double *R)
{
    int j, k, m;
    for (j = 0; j < d * d; ++j) R[j] = 0.0;
    for (j = 0; j < d; ++j) {
        double ajj = AT(A, j, j, d);
        for (m = 0; m < j; ++m) ajj -= AT(R, m, j, d) * AT(R, m, j, d);
        if (!(ajj > 0.0) || isnan(ajj)) return ORC_CHOL_FAILED;
        ajj = sqrt(ajj);
        AT(R, j, j, d) = ajj;
        double rinv = 1.0 / ajj;
        for (k = j + 1; k < d; ++k) {
            double s = AT(A, j, k, d);
            for (m = 0; m < j; ++m) s -= AT(R, m, j, d) * AT(R, m, k, d);
            AT(R, j, k, d) = s * rinv;
        }
    }
    return ORC_OK;
}

/* LU with partial pivoting in place (dgetf2); returns sign of the permutation, 0 if singular */
static int lu_factor(int d, double *A, int *piv)
{
    int sign = 1, i, j, k;
    for (k = 0; k < d; ++k) {
        int p = k;
        double best = fabs(AT(A, k, k, d));
        for (i = k + 1; i < d; ++i) {
            double v = fabs(AT(A, i, k, d));
            if (v > best) { best = v; p = i; }
        }
        piv[k] = p;
        if (AT(A, p, k, d) == 0.0) { sign = 0; continue; }
        if (p != k) {
            for (j = 0; j < d; ++j) {
                double t = AT(A, k, j, d);
                AT(A, k, j, d) = AT(A, p, j, d);
                AT(A, p, j, d) = t;
            }
            sign = -sign;
        }
        {
            double rinv = 1.0 / AT(A, k, k, d);
            for (i = k + 1; i < d; ++i) AT(A, i, k, d) *= rinv;
        }
        for (j = k + 1; j < d; ++j) {
            double akj = AT(A, k, j, d);
            for (i = k + 1; i < d; ++i) AT(A, i, j, d) -= AT(A, i, k, d) * akj;
        }
    }
    return sign;
}

/* arma::det */
double orc_det(int d, const double *A)
{
    double *W = (double *)malloc(sizeof(double) * d * d);
    int *piv = (int *)malloc(sizeof(int) * d);
    double det;
    int sign, i;
    memcpy(W, A, sizeof(double) * d * d);
    sign = lu_factor(d, W, piv);
    det = 1.0;
    for (i = 0; i < d; ++i) det *= AT(W, i, i, d);
    if (sign < 0) det = -det;
    if (sign == 0) det = 0.0;
    free(W);
    free(piv);
    return det;
}

/* arma::inv: LU, then solve A X = I column by column */
int orc_inv(int d, const double *A, double *Ainv)
{
    double *W = (double *)malloc(sizeof(double) * d * d);
    int *piv = (int *)malloc(sizeof(int) * d);
    int sign, i, j, k, status = ORC_OK;
    memcpy(W, A, sizeof(double) * d * d);
    sign = lu_factor(d, W, piv);
    if (sign == 0) status = ORC_SINGULAR;
    for (j = 0; j < d && status == ORC_OK; ++j) {
        double *x = Ainv + (size_t)j * d;
        for (i = 0; i < d; ++i) x[i] = (i == j) ? 1.0 : 0.0;
        for (k = 0; k < d; ++k) {           /* apply the row interchanges */
            if (piv[k] != k) { double t = x[k]; x[k] = x[piv[k]]; x[piv[k]] = t; }
        }
        for (k = 0; k < d; ++k)             /* forward: unit lower */
            for (i = k + 1; i < d; ++i) x[i] -= AT(W, i, k, d) * x[k];
        for (k = d - 1; k >= 0; --k) {      /* backward: upper */
            x[k] /= AT(W, k, k, d);
            for (i = 0; i < k; ++i) x[i] -= AT(W, i, k, d) * x[k];
        }
    }
    free(W);
    free(piv);
    return status;
}

/* ------------------------------------------------------------------------- */
/* update_sigma (adaptMCMC_source.cpp:42-45); i is a double on purpose          */
/* ------------------------------------------------------------------------- */
void orc_update_sigma(int d, const double *sigma, const double *residual, double i, double *out)
{
    double s = (i - 1) / i;
    for (int c = 0; c < d; ++c)
        for (int r = 0; r < d; ++r)
            AT(out, r, c, d) = (AT(sigma, r, c, d) * (i - 1) + (s * residual[r]) * residual[c]) / i;
}

/* mvrnorm_pdf (adaptMCMC_source.cpp:48-64) */
double orc_mvrnorm_pdf(int d, const double *x, const double *mu, const double *sigma)
{
    double twoPi = 2 * M_PI;
    double rootTerm = 1 / sqrt(pow(twoPi, d) * orc_det(d, sigma));
    double *Sinv = (double *)malloc(sizeof(double) * d * d);
    double *v = (double *)malloc(sizeof(double) * d);
    double *t = (double *)malloc(sizeof(double) * d);
    double q = 0.0, expTerm;
    int i, j;
    if (orc_inv(d, sigma, Sinv) != ORC_OK) { /* arma::inv would throw; callers never get here after a successful chol */
        for (i = 0; i < d * d; ++i) Sinv[i] = NAN;
    }
    for (i = 0; i < d; ++i) v[i] = x[i] - mu[i];
    /* (-0.5 * v^T) * Sinv, then * v : Armadillo folds the scalar into the first product */
    for (j = 0; j < d; ++j) {
        double s = 0.0;
        for (i = 0; i < d; ++i) s += v[i] * AT(Sinv, i, j, d);
        t[j] = -0.5 * s;
    }
    for (j = 0; j < d; ++j) q += t[j] * v[j];
    expTerm = exp(q);
    free(Sinv);
    free(v);
    free(t);
    return log(rootTerm * expTerm);
}

/* ------------------------------------------------------------------------- */
/* targets: the R closure `target` (adaptMCMC_source.cpp:97,158) as a C callback */
/* ------------------------------------------------------------------------- */
typedef double (*orc_target_fn)(const double *theta, int d, void *ctx);

/* adaptMCMC_analysis.R:115-129 */
double orc_target_goldstein_price(const double *xx, int d, void *ctx)
{
    (void)d; (void)ctx;
    double x1 = xx[0], x2 = xx[1];
    double s = x1 + x2 + 1;
    double fact1a = s * s;
    double fact1b = 19 - 14 * x1 + 3 * (x1 * x1) - 14 * x2 + 6 * x1 * x2 + 3 * (x2 * x2);
    double fact1 = 1 + fact1a * fact1b;
    double t = 2 * x1 - 3 * x2;
    double fact2a = t * t;
    double fact2b = 18 - 32 * x1 + 12 * (x1 * x1) + 48 * x2 - 36 * x1 * x2 + 27 * (x2 * x2);
    double fact2 = 30 + fact2a * fact2b;
    return -log(fact1 * fact2);
}

/* adaptMCMC_analysis.R:19-22; ctx -> double B (bananacity) */
double orc_target_banana(const double *x, int d, void *ctx)
{
    (void)d;
    double B = ctx ? *(const double *)ctx : 0.03;
    double u = x[1] + B * (x[0] * x[0]) - 100 * B;
    return -(x[0] * x[0]) / 200 - 1.0 / 2 * (u * u);
}

/* log N(mean, H^-1) up to a constant: -0.5 (x-m)^T H (x-m); ctx -> [mean(d), H(d*d col-major)] */
double orc_target_gaussian_prec(const double *x, int d, void *ctx)
{
    const double *mean = (const double *)ctx;
    const double *H = mean + d;
    double q = 0.0;
    for (int i = 0; i < d; ++i) {
        double s = 0.0;
        for (int j = 0; j < d; ++j) s += AT(H, i, j, d) * (x[j] - mean[j]);
        q += (x[i] - mean[i]) * s;
    }
    return -0.5 * q;
}

/* ------------------------------------------------------------------------- */
/* adaptMCMC (adaptMCMC_source.cpp:75-219)                                     */
/* ------------------------------------------------------------------------- */
typedef struct {
    /* optional per-iteration diagnostics (any may be NULL), each of length n_iterations */
    int *accepted;            /* is_accepted                                  */
    double *log_acceptance;   /* log_acceptance after :168-170                */
    double *log_A;            /* log(A)                                       */
    double *pdf_term;         /* mvrnorm_pdf(cur; prop) (== the p of SURVEY 8 a4) */
    double *scaling_sd;       /* after the adaptation block                   */
    double *acceptance_rate;  /* after the update at :188-207                 */
    double *target_propose;
} orc_diag;

/*
 * Z: n_iterations*d normals, Z[(i-1)*d + k] is Y(k) of iteration i (:149-152)
 * U: n_iterations uniforms, U[i-1] is A of iteration i (:174)
 * theta_trace: n x d column-major (arma::mat / R matrix), :92,185
 * sigma_empirical: d x d x n (arma::cube) or NULL to skip the store (:93,214)
 * fail_iter: iteration (1-based) at which chol failed, 0 if none.
 */
int orc_adapt_mcmc(orc_target_fn target, void *ctx, int d, const double *init_theta,
                   const double *covmat, int n_iterations, int adapt_size_start,
                   double acceptance_rate_weight, int acceptance_window, int adapt_shape_start,
                   double adapt_size_cooling, double max_scaling_sd, const double *Z,
                   const double *U, double *theta_trace, double *sigma_empirical,
                   double *acceptance_rate_out, int *fail_iter, orc_diag *diag)
{
    size_t dd = (size_t)d * d;
    double *theta_current = (double *)malloc(sizeof(double) * d);
    double *theta_propose = (double *)malloc(sizeof(double) * d);
    double *theta_mean = (double *)malloc(sizeof(double) * d);
    double *residual = (double *)malloc(sizeof(double) * d);
    double *covmat_proposal = (double *)malloc(sizeof(double) * dd);
    double *covmat_proposal_init = (double *)malloc(sizeof(double) * dd);
    double *covmat_proposal_new = (double *)malloc(sizeof(double) * dd);
    double *covmat_empirical = (double *)calloc(dd, sizeof(double));
    double *tmp = (double *)malloc(sizeof(double) * dd);
    double *R = (double *)malloc(sizeof(double) * dd);
    int *series = (int *)malloc(sizeof(int) * (acceptance_window > 0 ? acceptance_window + 1 : 1));
    int series_len = 0, series_head = 0; /* ring: newest at head */
    int status = ORC_OK, i, k, j;
    double target_theta_current, acceptance_rate = 0.0, scaling_sd = 1.0, scaling_multiplier = 1.0;

    if (fail_iter) *fail_iter = 0;
    memcpy(theta_current, init_theta, sizeof(double) * d);
    memcpy(theta_propose, init_theta, sizeof(double) * d);
    memcpy(theta_mean, init_theta, sizeof(double) * d);
    memcpy(covmat_proposal, covmat, sizeof(double) * dd);
    memcpy(covmat_proposal_init, covmat, sizeof(double) * dd);
    target_theta_current = target(theta_current, d, ctx);
    (void)scaling_multiplier;

    for (i = 1; i <= n_iterations; ++i) {
        /* adaptive routine, :117-139 */
        if (adapt_size_start != 0 && i >= adapt_size_start &&
            (adapt_shape_start == 0 || acceptance_rate * i < adapt_shape_start)) {
            scaling_multiplier = exp(pow(adapt_size_cooling, i - adapt_size_start) * (acceptance_rate - 0.234));
            scaling_sd = scaling_sd * scaling_multiplier;
            scaling_sd = scaling_sd < max_scaling_sd ? scaling_sd : max_scaling_sd; /* std::min */
            {
                double s2 = scaling_sd * scaling_sd; /* pow(scaling_sd,2) */
                int any_small = 0;
                for (size_t e = 0; e < dd; ++e) covmat_proposal_new[e] = s2 * covmat_proposal_init[e];
                for (k = 0; k < d; ++k) if (AT(covmat_proposal_new, k, k, d) < 2E-16) any_small = 1;
                if (!any_small) memcpy(covmat_proposal, covmat_proposal_new, sizeof(double) * dd);
            }
        } else if (adapt_shape_start != 0 && acceptance_rate * i >= adapt_shape_start) {
            scaling_sd = 2.38 / sqrt((double)d);
            double s2 = scaling_sd * scaling_sd;
            for (size_t e = 0; e < dd; ++e) covmat_proposal[e] = s2 * covmat_empirical[e];
        }
        if (diag && diag->scaling_sd) diag->scaling_sd[i - 1] = scaling_sd;

        /* propose, :149-154 */
        if (orc_chol_upper(d, covmat_proposal, R) != ORC_OK) {
            status = ORC_CHOL_FAILED;
            if (fail_iter) *fail_iter = i;
            break;
        }
        {
            const double *Y = Z + (size_t)(i - 1) * d;
            for (j = 0; j < d; ++j) {
                double s = 0.0;
                for (k = 0; k <= j; ++k) s += Y[k] * AT(R, k, j, d);
                theta_propose[j] = theta_current[j] + s;
            }
        }

        /* target + MH ratio, :157-171 */
        double target_theta_propose = target(theta_propose, d, ctx);
        double log_acceptance, pterm = NAN;
        int is_accepted;
        if (!isfinite(target_theta_propose)) {
            log_acceptance = -INFINITY;
        } else {
            log_acceptance = target_theta_propose - target_theta_current;
            pterm = orc_mvrnorm_pdf(d, theta_current, theta_propose, covmat_proposal);
            log_acceptance = log_acceptance + pterm;
            log_acceptance = log_acceptance - orc_mvrnorm_pdf(d, theta_propose, theta_current, covmat_proposal);
        }
        /* accept / reject, :174-182 */
        double A = U[i - 1];
        double logA = log(A);
        if (logA < log_acceptance) {
            is_accepted = 1;
            memcpy(theta_current, theta_propose, sizeof(double) * d);
            target_theta_current = target_theta_propose;
        } else {
            is_accepted = 0;
        }
        if (diag) {
            if (diag->accepted) diag->accepted[i - 1] = is_accepted;
            if (diag->log_acceptance) diag->log_acceptance[i - 1] = log_acceptance;
            if (diag->log_A) diag->log_A[i - 1] = logA;
            if (diag->pdf_term) diag->pdf_term[i - 1] = pterm;
            if (diag->target_propose) diag->target_propose[i - 1] = target_theta_propose;
        }
        /* trace, :185 */
        for (k = 0; k < d; ++k) theta_trace[(size_t)k * n_iterations + (i - 1)] = theta_current[k];

        /* acceptance rate, :188-207 */
        if (i == 1) {
            acceptance_rate = is_accepted;
        } else if (acceptance_rate_weight == 0) {
            if (acceptance_window == 0) {
                acceptance_rate = acceptance_rate + (is_accepted - acceptance_rate) / i;
            } else {
                /* prepend; drop the oldest once longer than the window; mean */
                int cap = acceptance_window + 1;
                series_head = (series_head + cap - 1) % cap;
                series[series_head] = is_accepted;
                series_len++;
                if (series_len > acceptance_window) series_len--;
                double sum = 0.0;
                for (k = 0; k < series_len; ++k) sum += series[(series_head + k) % cap];
                acceptance_rate = sum / series_len;
            }
        } else {
            acceptance_rate = acceptance_rate * (1 - acceptance_rate_weight) + is_accepted * acceptance_rate_weight;
        }
        if (diag && diag->acceptance_rate) diag->acceptance_rate[i - 1] = acceptance_rate;

        /* empirical covariance, :210-214 */
        for (k = 0; k < d; ++k) residual[k] = theta_current[k] - theta_mean[k];
        orc_update_sigma(d, covmat_empirical, residual, (double)i, tmp);
        memcpy(covmat_empirical, tmp, sizeof(double) * dd);
        for (k = 0; k < d; ++k) theta_mean[k] = theta_mean[k] + residual[k] / i;
        if (sigma_empirical) memcpy(sigma_empirical + (size_t)(i - 1) * dd, covmat_empirical, sizeof(double) * dd);
    }
    if (acceptance_rate_out) *acceptance_rate_out = acceptance_rate;
    free(theta_current); free(theta_propose); free(theta_mean); free(residual);
    free(covmat_proposal); free(covmat_proposal_init); free(covmat_proposal_new);
    free(covmat_empirical); free(tmp); free(R); free(series);
    return status;
}

/* rwMCMC (adaptMCMC_source.cpp:230-316) */
int orc_rw_mcmc(orc_target_fn target, void *ctx, int d, const double *init_theta,
                const double *covmat, int n_iterations, const double *Z, const double *U,
                double *theta_trace, double *acceptance_rate_out, orc_diag *diag)
{
    size_t dd = (size_t)d * d;
    double *theta_current = (double *)malloc(sizeof(double) * d);
    double *theta_propose = (double *)malloc(sizeof(double) * d);
    double *R = (double *)malloc(sizeof(double) * dd);
    double target_theta_current, acceptance_rate = 0.0;
    int i, j, k, status = ORC_OK;
    memcpy(theta_current, init_theta, sizeof(double) * d);
    target_theta_current = target(theta_current, d, ctx);
    for (i = 1; i <= n_iterations; ++i) {
        if (orc_chol_upper(d, covmat, R) != ORC_OK) { status = ORC_CHOL_FAILED; break; }
        const double *Y = Z + (size_t)(i - 1) * d;
        for (j = 0; j < d; ++j) {
            double s = 0.0;
            for (k = 0; k <= j; ++k) s += Y[k] * AT(R, k, j, d);
            theta_propose[j] = theta_current[j] + s;
        }
        double target_theta_propose = target(theta_propose, d, ctx);
        double log_acceptance, pterm = NAN;
        int is_accepted;
        if (!isfinite(target_theta_propose)) {
            log_acceptance = -INFINITY;
        } else {
            log_acceptance = target_theta_propose - target_theta_current;
            pterm = orc_mvrnorm_pdf(d, theta_current, theta_propose, covmat);
            log_acceptance = log_acceptance + pterm;
            log_acceptance = log_acceptance - orc_mvrnorm_pdf(d, theta_propose, theta_current, covmat);
        }
        double logA = log(U[i - 1]);
        if (logA < log_acceptance) {
            is_accepted = 1;
            memcpy(theta_current, theta_propose, sizeof(double) * d);
            target_theta_current = target_theta_propose;
        } else {
            is_accepted = 0;
        }
        if (diag) {
            if (diag->accepted) diag->accepted[i - 1] = is_accepted;
            if (diag->log_acceptance) diag->log_acceptance[i - 1] = log_acceptance;
            if (diag->log_A) diag->log_A[i - 1] = logA;
            if (diag->pdf_term) diag->pdf_term[i - 1] = pterm;
            if (diag->target_propose) diag->target_propose[i - 1] = target_theta_propose;
        }
        for (k = 0; k < d; ++k) theta_trace[(size_t)k * n_iterations + (i - 1)] = theta_current[k];
        if (i == 1) acceptance_rate = is_accepted;
        else acceptance_rate = acceptance_rate + (is_accepted - acceptance_rate) / i;
        if (diag && diag->acceptance_rate) diag->acceptance_rate[i - 1] = acceptance_rate;
    }
    if (acceptance_rate_out) *acceptance_rate_out = acceptance_rate;
    free(theta_current); free(theta_propose); free(R);
    return status;
}

/* convenience wrappers selecting a built-in target by id (ctypes-friendly) */
static orc_target_fn builtin_target(int id)
{
    switch (id) {
    case 0: return orc_target_gaussian_prec;
    case 1: return orc_target_goldstein_price;
    case 2: return orc_target_banana;
    default: return NULL;
    }
}
double orc_target_eval(int id, const double *theta, int d, void *ctx)
{
    orc_target_fn f = builtin_target(id);
    return f ? f(theta, d, ctx) : NAN;
}
int orc_adapt_mcmc_builtin(int target_id, void *ctx, int d, const double *init_theta,
                           const double *covmat, int n_iterations, int adapt_size_start,
                           double acceptance_rate_weight, int acceptance_window,
                           int adapt_shape_start, double adapt_size_cooling, double max_scaling_sd,
                           const double *Z, const double *U, double *theta_trace,
                           double *sigma_empirical, double *acceptance_rate_out, int *fail_iter,
                           orc_diag *diag)
{
    orc_target_fn f = builtin_target(target_id);
    if (!f) return ORC_BAD_ARG;
    return orc_adapt_mcmc(f, ctx, d, init_theta, covmat, n_iterations, adapt_size_start,
                          acceptance_rate_weight, acceptance_window, adapt_shape_start,
                          adapt_size_cooling, max_scaling_sd, Z, U, theta_trace, sigma_empirical,
                          acceptance_rate_out, fail_iter, diag);
}
int orc_rw_mcmc_builtin(int target_id, void *ctx, int d, const double *init_theta,
                        const double *covmat, int n_iterations, const double *Z, const double *U,
                        double *theta_trace, double *acceptance_rate_out, orc_diag *diag)
{
    orc_target_fn f = builtin_target(target_id);
    if (!f) return ORC_BAD_ARG;
    return orc_rw_mcmc(f, ctx, d, init_theta, covmat, n_iterations, Z, U, theta_trace,
                       acceptance_rate_out, diag);
}

/* ------------------------------------------------------------------------- */
/* tmvrnormGibbs.cpp                                                           */
/* ------------------------------------------------------------------------- */
/* sub1: x[-i,-i] (:12-16); x is d x d, out is (d-1) x (d-1) */
void orc_sub1(int d, const double *x, int i, double *out)
{
    int m = d - 1;
    for (int c = 0, oc = 0; c < d; ++c) {
        if (c == i) continue;
        for (int r = 0, orow = 0; r < d; ++r) {
            if (r == i) continue;
            out[(size_t)oc * m + orow] = AT(x, r, c, d);
            ++orow;
        }
        ++oc;
    }
}
/* sub2: x[a,-b] (:20-23); x is nr x nc; out has nc-1 entries */
void orc_sub2(int nr, int nc, const double *x, int a, int b, double *out)
{
    for (int c = 0, oc = 0; c < nc; ++c) {
        if (c == b) continue;
        out[oc++] = x[(size_t)c * nr + a];
    }
}
/* negSubCol / negSubRow: x[-i] (:27-37) */
void orc_neg_sub(int n, const double *x, int i, double *out)
{
    for (int k = 0, o = 0; k < n; ++k) {
        if (k == i) continue;
        out[o++] = x[k];
    }
}

/* rtnorm_gibbs (:49-70); F are the runif(n) draws */
void orc_rtnorm_gibbs(int n, double mu, double sigma, double a, double b, const double *F, double *out)
{
    double Fa = orc_pnorm(a, mu, sigma);
    double Fb = orc_pnorm(b, mu, sigma);
    for (int i = 0; i < n; ++i) {
        double p_i = F[i] * (Fb - Fa) + Fa;
        double z = orc_qnorm(p_i);
        out[i] = mu + sigma * z;
    }
}

/* precompute of :100-113: P is d slices of 1 x (d-1) (row i = regression of x_i on x_-i), sd(d) */
int orc_tmvn_precompute(int d, const double *sigma, double *P, double *sd)
{
    int m = d - 1, status = ORC_OK;
    double *Sigma = (double *)malloc(sizeof(double) * (m > 0 ? m * m : 1));
    double *Sinv = (double *)malloc(sizeof(double) * (m > 0 ? m * m : 1));
    double *Sigma_i = (double *)malloc(sizeof(double) * (m > 0 ? m : 1));
    for (int i = 0; i < d; ++i) {
        orc_sub1(d, sigma, i, Sigma);
        double sigma_ii = AT(sigma, i, i, d);
        orc_sub2(d, d, sigma, i, i, Sigma_i);
        if (m > 0 && orc_inv(m, Sigma, Sinv) != ORC_OK) { status = ORC_SINGULAR; break; }
        double p_i = 0.0;
        for (int c = 0; c < m; ++c) {
            double s = 0.0;
            for (int r = 0; r < m; ++r) s += Sigma_i[r] * Sinv[(size_t)c * m + r];
            P[(size_t)i * m + c] = s;
        }
        for (int c = 0; c < m; ++c) p_i += P[(size_t)i * m + c] * Sigma_i[c];
        sd[i] = sqrt(sigma_ii - p_i);
    }
    free(Sigma); free(Sinv); free(Sigma_i);
    return status;
}

/* rtmvnorm_gibbs (:91-142); U has n*d entries consumed as U[l], l = j*d + i; trace is n x d column-major */
int orc_rtmvnorm_gibbs(int n, int d, const double *mu, const double *sigma, const double *lower,
                       const double *upper, const double *init_state, const double *U, double *trace)
{
    int m = d - 1;
    double *P = (double *)malloc(sizeof(double) * (size_t)d * (m > 0 ? m : 1));
    double *sd = (double *)malloc(sizeof(double) * d);
    double *x = (double *)malloc(sizeof(double) * d);
    int status = orc_tmvn_precompute(d, sigma, P, sd);
    size_t l = 0;
    if (status != ORC_OK) { free(P); free(sd); free(x); return status; }
    memcpy(x, init_state, sizeof(double) * d);
    for (int j = 0; j < n; ++j) {
        for (int i = 0; i < d; ++i) {
            const double *slice_i = P + (size_t)i * m;
            double s = 0.0;
            for (int k = 0, c = 0; k < d; ++k) {
                if (k == i) continue;
                s += slice_i[c++] * (x[k] - mu[k]);
            }
            double mu_i = mu[i] + s;
            double Fa = orc_pnorm(lower[i], mu_i, sd[i]);
            double Fb = orc_pnorm(upper[i], mu_i, sd[i]);
            x[i] = mu_i + sd[i] * orc_qnorm(U[l] * (Fb - Fa) + Fa);
            l = l + 1;
        }
        for (int k = 0; k < d; ++k) trace[(size_t)k * n + j] = x[k];
    }
    free(P); free(sd); free(x);
    return ORC_OK;
}

/* ------------------------------------------------------------------------- */
/* libstdc++-compatible streams: what `std::mt19937 engine(seedMH)` +            */
/* normal_distribution<> + uniform_real_distribution<> (adaptMCMC_source.cpp:78-80) */
/* draw under GNU libstdc++ (bits/random.tcc: polar method with a saved value;   */
/* generate_canonical<double,53> = (lo + hi*2^32) / 2^64 from two 32-bit words). */
/* ------------------------------------------------------------------------- */
typedef struct { uint32_t mt[624]; int idx; } orc_mt19937;

static void mt_seed(orc_mt19937 *g, uint32_t seed)
{
    g->mt[0] = seed;
    for (int i = 1; i < 624; ++i)
        g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
    g->idx = 624;
}
static uint32_t mt_next(orc_mt19937 *g)
{
    if (g->idx >= 624) {
        for (int i = 0; i < 624; ++i) {
            uint32_t y = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7fffffffu);
            g->mt[i] = g->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        g->idx = 0;
    }
    uint32_t y = g->mt[g->idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}
static double mt_canonical(orc_mt19937 *g)
{
    /* generate_canonical<double, 53>: k = 2 calls, sum += (x - min) * tmp; tmp *= 2^32 */
    double lo = (double)mt_next(g);
    double hi = (double)mt_next(g);
    double r = (lo + hi * 4294967296.0) / 18446744073709551616.0;
    if (r >= 1.0) r = nextafter(1.0, 0.0);
    return r;
}

uint32_t orc_mt19937_u32(uint32_t seed, int skip)
{
    orc_mt19937 g;
    mt_seed(&g, seed);
    uint32_t v = 0;
    for (int i = 0; i <= skip; ++i) v = mt_next(&g);
    return v;
}

/* Fill Z (n_iterations*d) and U (n_iterations) in reference draw order. seedMH is a double
 * converted to the engine's result_type (uint_fast32_t = 64-bit here) then reduced mod 2^32. */
void orc_stdlib_streams(double seedMH, int n_iterations, int d, double *Z, double *U)
{
    orc_mt19937 g;
    int saved_available = 0;
    double saved = 0.0;
    mt_seed(&g, (uint32_t)((uint64_t)seedMH & 0xffffffffu));
    for (int i = 0; i < n_iterations; ++i) {
        for (int k = 0; k < d; ++k) {
            double ret;
            if (saved_available) {
                saved_available = 0;
                ret = saved;
            } else {
                double x, y, r2;
                do {
                    x = 2.0 * mt_canonical(&g) - 1.0;
                    y = 2.0 * mt_canonical(&g) - 1.0;
                    r2 = x * x + y * y;
                } while (r2 > 1.0 || r2 == 0.0);
                double mult = sqrt(-2 * log(r2) / r2);
                saved = x * mult;
                saved_available = 1;
                ret = y * mult;
            }
            Z[(size_t)i * d + k] = ret * 1.0 + 0.0; /* stddev 1, mean 0 */
        }
        U[i] = mt_canonical(&g) * (1.0 - 0.0) + 0.0;
    }
}

/* ------------------------------------------------------------------------- */
/* CPU baseline driver: one adaptMCMC chain with inline libstdc++-style RNG,     */
/* all five factorizations per iteration, as the reference does them.  Used by  */
/* bench.py's cpu_baseline leg (one call per host thread; ctypes drops the GIL). */
/* ------------------------------------------------------------------------- */
int orc_adapt_mcmc_seeded(int target_id, void *ctx, int d, const double *init_theta,
                          const double *covmat, int n_iterations, int adapt_size_start,
                          double acceptance_rate_weight, int acceptance_window,
                          int adapt_shape_start, double seedMH, double adapt_size_cooling,
                          double max_scaling_sd, double *theta_trace, double *sigma_empirical,
                          double *acceptance_rate_out)
{
    double *Z = (double *)malloc(sizeof(double) * (size_t)n_iterations * d);
    double *U = (double *)malloc(sizeof(double) * (size_t)n_iterations);
    int fail = 0, st;
    orc_stdlib_streams(seedMH, n_iterations, d, Z, U);
    st = orc_adapt_mcmc_builtin(target_id, ctx, d, init_theta, covmat, n_iterations,
                                adapt_size_start, acceptance_rate_weight, acceptance_window,
                                adapt_shape_start, adapt_size_cooling, max_scaling_sd, Z, U,
                                theta_trace, sigma_empirical, acceptance_rate_out, &fail, NULL);
    free(Z);
    free(U);
    return st;
}

/* ------------------------------------------------------------------------- */
/* truncated-MVN density and the toy walkers    */
/* ------------------------------------------------------------------------- */

/* mvrGaussian_pdf, deprecated/truncGaussian.cpp:11-33: the un-logged density first, then log if do_log */
double orc_mvrGaussian_pdf(int d, const double *x, const double *mu, const double *sigma, int do_log)
{
    double lg = orc_mvrnorm_pdf(d, x, mu, sigma); /* log(rootTerm * expTerm): the same statements as :19-26 */
    if (do_log) return lg;
    {
        /* out = rootTerm * expTerm itself (:28,32): redo the product rather than exp(log(.)) */
        double twoPi = 2 * M_PI;
        double rootTerm = 1 / sqrt(pow(twoPi, d) * orc_det(d, sigma));
        double *Sinv = (double *)malloc(sizeof(double) * d * d);
        double *v = (double *)malloc(sizeof(double) * d);
        double q = 0.0, out;
        int i, j;
        if (orc_inv(d, sigma, Sinv) != ORC_OK)
            for (i = 0; i < d * d; ++i) Sinv[i] = NAN;
        for (i = 0; i < d; ++i) v[i] = x[i] - mu[i];
        for (j = 0; j < d; ++j) {
            double s = 0.0;
            for (i = 0; i < d; ++i) s += v[i] * AT(Sinv, i, j, d);
            q += (-0.5 * s) * v[j];
        }
        out = rootTerm * exp(q);
        free(Sinv);
        free(v);
        (void)lg;
        return out;
    }
}

/* tmvrGaussian_pdf, deprecated/truncGaussian.cpp:38-78.  `cdf` is the value the reference obtains from
 * mvtnorm::pmvnorm(lower, upper, mu, sigma) (:74-75; a third-party randomized integral, not restated here).
 * Quirk kept: mvrGaussian_pdf is called with its default do_log = true (:73), so the LOG density is divided by cdf. */
double orc_tmvrGaussian_pdf(int d, const double *x, const double *mu, const double *sigma, const double *lower,
                            const double *upper, double cdf)
{
    int i;
    for (i = 0; i < d; ++i)
        if (!((x[i] >= lower[i]) && (x[i] <= upper[i]))) return 0.0; /* outside the support (:45-63) */
    return orc_mvrGaussian_pdf(d, x, mu, sigma, 1) / cdf;
}

/* random_walk2d / random_walk3d, randomWalkers.cpp:9-74: out is n x dim column-major, row 0 the origin;
 * step i uses U[i-1] = R::runif(0,1): r_num = 1 + floor(U * 2 dim); odd r_num: +1, even: -1 on axis (r_num-1)/2 */
void orc_random_walk(int dim, int n, const double *U, double *out)
{
    int i, k;
    for (k = 0; k < dim; ++k) out[(size_t)k * n] = 0.0;
    for (i = 1; i < n; ++i) {
        double r_num = 1 + floor(U[i - 1] * (2 * dim));
        for (k = 0; k < dim; ++k) out[(size_t)k * n + i] = out[(size_t)k * n + i - 1];
        for (k = 0; k < dim; ++k) {
            if (r_num == 2 * k + 1) out[(size_t)k * n + i] = out[(size_t)k * n + i - 1] + 1;
            if (r_num == 2 * k + 2) out[(size_t)k * n + i] = out[(size_t)k * n + i - 1] - 1;
        }
    }
}

/* random_walkGraph, randomWalkers.cpp:80-100: out(i) = RcppArmadillo::sample(0..nn-1, 1, false, graph(out(i-1), _)).
 * RcppArmadillo's sample.h is third-party and not under /root/reference; restated from its published algorithm
 * (FixProb + ProbSampleNoReplace, the C++ transcription of R's do_sample): the probabilities are normalised,
 * sorted in DESCENDING order carrying their indices, and the first position whose running sum reaches
 * rT = unif_rand() is taken (the last position if none does).  Ties in the sort keep index order here; Armadillo's
 * sort_index does not promise that -- PARITY UNPINNED for rows with equal probabilities. */
int orc_random_walkGraph(int nn, const double *graph, int init_pos, int n, const double *U, double *out)
{
    double *p = (double *)malloc(sizeof(double) * nn);
    int *perm = (int *)malloc(sizeof(int) * nn);
    int i, j, k, status = ORC_OK;
    if (init_pos < 0 || init_pos >= nn) status = ORC_BAD_ARG;
    if (n > 0) out[0] = init_pos;
    for (i = 1; i < n && status == ORC_OK; ++i) {
        int pos = (int)out[i - 1];
        double sum = 0.0, mass = 0.0, rT;
        for (k = 0; k < nn; ++k) {
            p[k] = AT(graph, pos, k, nn); /* graph(pos_begin, _) */
            perm[k] = k;
            if (!(p[k] >= 0.0) || !isfinite(p[k])) status = ORC_BAD_ARG;
            sum += p[k];
        }
        if (!(sum > 0.0)) status = ORC_BAD_ARG;
        if (status != ORC_OK) break;
        for (k = 0; k < nn; ++k) p[k] = p[k] / sum;
        for (k = 1; k < nn; ++k) { /* stable insertion sort, descending */
            double pk = p[k];
            int ik = perm[k];
            for (j = k - 1; j >= 0 && p[j] < pk; --j) {
                p[j + 1] = p[j];
                perm[j + 1] = perm[j];
            }
            p[j + 1] = pk;
            perm[j + 1] = ik;
        }
        rT = U[i - 1];
        for (j = 0; j < nn - 1; ++j) {
            mass += p[j];
            if (rT <= mass) break;
        }
        out[i] = perm[j];
    }
    free(p);
    free(perm);
    return status;
}
