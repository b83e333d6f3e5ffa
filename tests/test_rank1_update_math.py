"""CPU suite: the closed-form rank-1 update of the proposal factor that MCMC_FACTOR_RANK1 runs on the device
(rwm_block.hip: fast_step), restated in numpy and checked against a fresh Cholesky factorization.

update_sigma (adaptMCMC_source.cpp:42-45) gives C_i = a C_{i-1} + b r r^T with a = (i-1)/i, b = a/i; the reference then
refactors C_i from scratch (:153).  With C_{i-1} = R^T R and q = R^-T r, p = sqrt(b/a) q:  C_i = a R^T (I + p p^T) R, and the
upper factor of I + p p^T is known in closed form, so  R_i = G R  needs a prefix sum, d square roots and one suffix sum
per column -- no recurrence over the pivots."""
import numpy as np


def rank1_step(R, q, i):
    """R = chol(C_{i-1}) (upper), q = R^-T r_i  ->  (chol(C_i), R_i^-T r_i)"""
    a = (i - 1.0) / i
    p = q / np.sqrt(i)  # sqrt(b / a) = 1 / sqrt(i)
    sg = 1.0 + np.concatenate([[0.0], np.cumsum(p * p)])  # sg[k] = 1 + sum_{l<k} p_l^2
    rs = 1.0 / np.sqrt(sg[:-1] * sg[1:])
    g, u = np.sqrt(a) * sg[1:] * rs, np.sqrt(a) * p * rs
    Rn = np.zeros_like(R)
    for j in range(R.shape[0]):
        sfx = 0.0
        for k in range(j, -1, -1):
            Rn[k, j] = g[k] * R[k, j] + u[k] * sfx
            sfx += p[k] * R[k, j]
    return Rn, q * rs / np.sqrt(a)


def test_closed_form_factor_update_tracks_the_fresh_cholesky():
    rng = np.random.default_rng(0)
    for d in (3, 17, 64):
        A = rng.standard_normal((d, d))
        C = A @ A.T / d + 0.1 * np.eye(d)
        mean = rng.standard_normal(d)
        theta = mean + 0.3 * rng.standard_normal(d)
        i = 3 * d
        r = theta - mean
        C = (C * (i - 1) + ((i - 1) / i) * np.outer(r, r)) / i  # update_sigma, :43
        mean = mean + r / i
        R = np.linalg.cholesky(C).T
        qw = np.linalg.solve(R.T, r)
        s = 2.38 / np.sqrt(d)
        worst = 0.0
        for i in range(3 * d + 1, 3 * d + 400):
            Y = rng.standard_normal(d)
            acc = rng.random() < 0.4
            if acc:
                theta = theta + s * (Y @ R)  # the proposal step, :153-154
            r = theta - mean
            # the whitened residual without a triangular solve: r_i = (1 - 1/(i-1)) r_{i-1} + step, step whitened = s Y
            q = (1.0 - 1.0 / (i - 1)) * qw + (s * Y if acc else 0.0)
            np.testing.assert_allclose(q, np.linalg.solve(R.T, r), rtol=1e-8, atol=1e-11)
            C = (C * (i - 1) + ((i - 1) / i) * np.outer(r, r)) / i
            mean = mean + r / i
            prod_before = np.prod(np.diag(R))
            R, qw = rank1_step(R, q, i)
            exact = np.linalg.cholesky(C).T
            worst = max(worst, np.abs(R - exact).max() / np.abs(exact).max())
            assert np.all(np.triu(R) == R)
            # the determinant term of mvrnorm_pdf (:56) follows without a product over the diagonal
            p2 = np.sum((q / np.sqrt(i)) ** 2)
            np.testing.assert_allclose(np.prod(np.diag(R)), prod_before * ((i - 1) / i) ** (d / 2) * np.sqrt(1 + p2), rtol=1e-12)
        assert worst < 1e-12, (d, worst)  # 400 updates in a row, no refresh: rounding drift only
