// Small dense host-side helpers for the once-per-call work that is shared by all chains
// (validating / factoring covmat, inverting sigma for the Gibbs regression rows).
#pragma once
#include <cmath>
#include <vector>

namespace mcmc {

// column-major accessor
inline double &cm(std::vector<double> &M, int r, int c, int d) { return M[(size_t)c * d + r]; }
inline double cmr(const double *M, int r, int c, int d) { return M[(size_t)c * d + r]; }

// Upper Cholesky factor (R^T R = A) from the upper triangle of column-major A, packed
// (r,c), r<=c at c*(c+1)/2 + r.  Same row-by-row order as the device kernels.  false if not PD.
inline bool chol_upper_packed(int d, const double *A, std::vector<double> &R, double &det)
{
    auto tri = [](int r, int c) { return c * (c + 1) / 2 + r; };
    R.assign((size_t)d * (d + 1) / 2, 0.0);
    double prod = 1.0;
    for (int j = 0; j < d; ++j) {
        double ajj = cmr(A, j, j, d);
        for (int m = 0; m < j; ++m) ajj -= R[tri(m, j)] * R[tri(m, j)];
        if (!(ajj > 0.0)) return false;
        const double rjj = std::sqrt(ajj);
        const double rinv = 1.0 / rjj;
        R[tri(j, j)] = rjj;
        prod *= rjj;
        for (int k = j + 1; k < d; ++k) {
            double s = cmr(A, j, k, d);
            for (int m = 0; m < j; ++m) s -= R[tri(m, j)] * R[tri(m, k)];
            R[tri(j, k)] = s * rinv;
        }
    }
    det = prod * prod;
    return true;
}

// Inverse of a symmetric positive definite column-major matrix through its Cholesky factor,
// accumulated in long double (the result is shared by every chain, so the extra care is free).
inline bool spd_inverse(int d, const double *A, std::vector<double> &Ainv)
{
    std::vector<long double> L((size_t)d * d, 0.0L); // lower factor, L L^T = A
    for (int j = 0; j < d; ++j) {
        long double s = cmr(A, j, j, d);
        for (int m = 0; m < j; ++m) s -= L[(size_t)m * d + j] * L[(size_t)m * d + j];
        if (!(s > 0.0L)) return false;
        const long double ljj = sqrtl(s);
        L[(size_t)j * d + j] = ljj;
        for (int i = j + 1; i < d; ++i) {
            long double t = 0.5L * ((long double)cmr(A, i, j, d) + (long double)cmr(A, j, i, d));
            for (int m = 0; m < j; ++m) t -= L[(size_t)m * d + i] * L[(size_t)m * d + j];
            L[(size_t)j * d + i] = t / ljj;
        }
    }
    // Linv (lower) by forward substitution, column by column
    std::vector<long double> Li((size_t)d * d, 0.0L);
    for (int c = 0; c < d; ++c) {
        Li[(size_t)c * d + c] = 1.0L / L[(size_t)c * d + c];
        for (int i = c + 1; i < d; ++i) {
            long double t = 0.0L;
            for (int m = c; m < i; ++m) t -= L[(size_t)m * d + i] * Li[(size_t)c * d + m];
            Li[(size_t)c * d + i] = t / L[(size_t)i * d + i];
        }
    }
    // A^-1 = Linv^T Linv
    Ainv.assign((size_t)d * d, 0.0);
    for (int r = 0; r < d; ++r)
        for (int c = r; c < d; ++c) {
            long double t = 0.0L;
            for (int m = c; m < d; ++m) t += Li[(size_t)r * d + m] * Li[(size_t)c * d + m];
            Ainv[(size_t)c * d + r] = (double)t;
            Ainv[(size_t)r * d + c] = (double)t;
        }
    return true;
}

} // namespace mcmc
