"""CPU suite: pins the oracle (oracle/mcmc_oracle.c) -- the checker the GPU parity tests rely on.

The reference stores no expected outputs ("parity unpinned"), so the oracle is
pinned by (i) the one hand-derivable known answer the reference's scripts contain
(deprecated/adaptMCMC.R:146-155), (ii) independent implementations present in this image
(scipy / numpy / mpmath / this toolchain's libstdc++ <random>), (iii) a pure-Python restatement
of the acceptance-rate logic, and (iv) the committed vectors under tests/golden/.
"""
import os

import numpy as np
import pytest
import scipy.special as sp
from scipy import stats

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_update_sigma_known_answer(oracle):
    """deprecated/adaptMCMC.R:146-155: cov=[[5,0],[0,.1]], residual=(.25,1.5), i=5."""
    got = oracle.update_sigma([[5, 0], [0, 0.1]], [0.25, 1.5], 5)
    np.testing.assert_allclose(got, [[4.01, 0.06], [0.06, 0.44]], rtol=2e-16 * 4)
    g = np.load(os.path.join(GOLD, "g1_update_sigma.npz"))
    np.testing.assert_array_equal(got, g["expected"])
    # i is a double on purpose (adaptMCMC_source.cpp:42): (i-1)/i must not be integer division
    assert oracle.update_sigma([[1.0]], [1.0], 2)[0, 0] == pytest.approx((1.0 * 1 + 0.5) / 2)
    # at i = 1 the covariance stays 0
    assert np.all(oracle.update_sigma(np.zeros((2, 2)), [3.0, -1.0], 1) == 0.0)


def test_pnorm_matches_scipy_and_mpmath(oracle):
    import mpmath as mp

    x = np.concatenate([np.linspace(-37.5, 8.29, 50001), np.random.default_rng(0).normal(size=20000) * 3])
    p = oracle.pnorm(x)
    ref = sp.ndtr(x)
    body = x > -5  # scipy's ndtr itself loses relative accuracy further out (2e-13 at -37); mpmath covers the tail
    assert np.max(np.abs(p[body] - ref[body]) / ref[body]) < 5e-15
    assert np.max(np.abs(p - ref) / ref) < 1e-12
    mp.mp.dps = 40
    for xv in (-37.4, -35.0, -20.0, -8.0, -0.3, 0.5, 3.0):
        assert oracle.pnorm(xv)[0] == pytest.approx(float(mp.ncdf(mp.mpf(xv))), rel=4e-16)
    # R's cut-offs for the lower tail (pnorm_both): 0 below -37.5193, 1 above 8.2924
    assert oracle.pnorm(-37.52)[0] == 0.0 and oracle.pnorm(-37.51)[0] > 0.0
    assert oracle.pnorm(8.30)[0] == 1.0
    assert oracle.pnorm(np.inf)[0] == 1.0 and oracle.pnorm(-np.inf)[0] == 0.0
    # location / scale form used at tmvrnormGibbs.cpp:130-131
    np.testing.assert_allclose(oracle.pnorm([1.0, 4.0], 2.0, 3.0), sp.ndtr([-1 / 3, 2 / 3]), rtol=1e-15)


def test_qnorm_matches_scipy_and_in_tree_coefficients(oracle):
    pp = np.concatenate([np.linspace(1e-300, 1, 20001)[:-1], 10.0 ** -np.arange(1, 300), 1 - 10.0 ** -np.arange(1, 16)])
    q = oracle.qnorm(pp)
    qr = sp.ndtri(pp)
    assert np.max(np.abs(q - qr) / np.maximum(1, np.abs(qr))) < 4e-15
    out = oracle.qnorm([0.0, 1.0, 0.5, -0.1, 1.1])
    assert out[0] == -np.inf and out[1] == np.inf and out[2] == 0.0 and np.isnan(out[3]) and np.isnan(out[4])
    # spot values of AS241 at the branch points 0.075 / 0.925 (|q| = 0.425) and r = 5
    for p in (0.075, 0.925, np.exp(-25.0), 1 - 0.0749999):
        assert oracle.qnorm(p)[0] == pytest.approx(sp.ndtri(p), rel=1e-15, abs=1e-15)


def test_dense_helpers_match_numpy(oracle):
    rng = np.random.default_rng(1)
    for d in (1, 2, 3, 7, 20):
        A = rng.standard_normal((d, d))
        S = A @ A.T + np.eye(d)
        st, R = oracle.chol(S)
        assert st == 0
        np.testing.assert_allclose(R, np.linalg.cholesky(S).T, atol=1e-13)  # arma::chol is upper: R^T R = S
        assert np.allclose(np.tril(R, -1), 0.0)
        assert oracle.det(S) == pytest.approx(np.linalg.det(S), rel=1e-12)
        st, Si = oracle.inv(S)
        np.testing.assert_allclose(Si, np.linalg.inv(S), atol=1e-12)
        G = rng.standard_normal((d, d)) + 3 * np.eye(d)  # general (pivoting) case
        assert oracle.det(G) == pytest.approx(np.linalg.det(G), rel=1e-11)
    st, _ = oracle.chol(np.array([[1.0, 2.0], [2.0, 1.0]]))
    assert st == oracle.ORC_CHOL_FAILED


def test_mvrnorm_pdf_is_the_unlogged_density_then_log(oracle):
    """adaptMCMC_source.cpp:48-64: log(rootTerm * exp(...)), so it underflows to -Inf for far points."""
    rng = np.random.default_rng(2)
    d = 4
    A = rng.standard_normal((d, d))
    S = A @ A.T + np.eye(d)
    x, mu = rng.standard_normal(d), rng.standard_normal(d)
    assert oracle.mvrnorm_pdf(x, mu, S) == pytest.approx(stats.multivariate_normal(mu, S).logpdf(x), rel=1e-12)
    # symmetric in (x, mu): the two Hastings terms at :169-170 cancel in exact arithmetic
    assert oracle.mvrnorm_pdf(x, mu, S) == oracle.mvrnorm_pdf(mu, x, S)
    assert oracle.mvrnorm_pdf(np.full(d, 100.0), np.zeros(d), np.eye(d)) == -np.inf  # exp underflow (q > 1490)


def test_stdlib_streams_equal_this_toolchains_libstdcxx(oracle):
    """std::mt19937 + normal_distribution (polar, cached) + uniform_real_distribution, drawn in the
    reference's order (adaptMCMC_source.cpp:78-80,149-152,174)."""
    assert oracle.lib().orc_mt19937_u32(5489, 9999) == 4123659995  # the C++ standard's mt19937 check value
    for seed, n, d in ((123.0, 500, 3), (1.0, 200, 10), (4294967296.0 + 7, 50, 1)):
        Z, U = oracle.stdlib_streams(seed, n, d)
        Z2, U2 = np.empty(n * d), np.empty(n)
        oracle.stdlib_check_lib().stdcheck_streams(seed, n, d, Z2.ctypes.data_as(oracle._dp), U2.ctypes.data_as(oracle._dp))
        np.testing.assert_array_equal(Z.ravel(), Z2)
        np.testing.assert_array_equal(U, U2)


def _py_adapt(target, init, cov, n, ass, w, win, ashs, Z, U, cooling=0.99, max_sd=50.0):
    """Independent pure-numpy restatement of adaptMCMC_source.cpp:108-216 (numpy LAPACK for chol/det/inv)."""
    th = np.array(init, float)
    d = th.size
    covp, cov0 = np.array(cov, float), np.array(cov, float)
    t_cur = target(th)
    rate, series, sd = 0.0, [], 1.0
    C, mean = np.zeros((d, d)), th.copy()
    trace, accs, rates = np.zeros((n, d)), np.zeros(n, int), np.zeros(n)

    def pdf(x, mu, S):
        with np.errstate(all="ignore"):
            v = x - mu
            return np.log(1 / np.sqrt((2 * np.pi) ** d * np.linalg.det(S)) * np.exp(-0.5 * v @ np.linalg.inv(S) @ v))

    for i in range(1, n + 1):
        if ass != 0 and i >= ass and (ashs == 0 or rate * i < ashs):
            sd = min(sd * np.exp(cooling ** (i - ass) * (rate - 0.234)), max_sd)
            new = sd ** 2 * cov0
            if not np.any(np.diag(new) < 2e-16):
                covp = new
        elif ashs != 0 and rate * i >= ashs:
            sd = 2.38 / np.sqrt(d)
            covp = sd ** 2 * C
        prop = th + Z[i - 1] @ np.linalg.cholesky(covp).T
        t_prop = target(prop)
        if not np.isfinite(t_prop):
            la = -np.inf
        else:
            la = t_prop - t_cur
            la = la + pdf(th, prop, covp)
            la = la - pdf(prop, th, covp)
        acc = bool(np.log(U[i - 1]) < la)
        if acc:
            th, t_cur = prop, t_prop
        trace[i - 1], accs[i - 1] = th, acc
        if i == 1:
            rate = float(acc)
        elif w == 0:
            if win == 0:
                rate = rate + (acc - rate) / i
            else:
                series.insert(0, float(acc))
                if len(series) > win:
                    series.pop()
                rate = float(np.mean(series))
        else:
            rate = rate * (1 - w) + acc * w
        rates[i - 1] = rate
        r = th - mean
        C = (C * (i - 1) + (i - 1) / i * np.outer(r, r)) / i
        mean = mean + r / i
    return trace, accs, rates, C


@pytest.mark.parametrize("w,win", [(0.0, 0), (0.05, 0), (0.0, 7)])
def test_adaptMCMC_against_independent_numpy_restatement(oracle, w, win):
    rng = np.random.default_rng(3)
    d, n = 3, 400
    A = rng.standard_normal((d, d))
    H = np.linalg.inv(A @ A.T / d + np.eye(d))
    H = 0.5 * (H + H.T)
    Z, U = rng.standard_normal((n, d)), rng.random(n)
    init = [0.5, -0.5, 1.0]
    r = oracle.adaptMCMC(oracle.gaussian_target(np.zeros(d), H), init, np.eye(d), n, 20, w, win, 60, Z, U, diagnostics=True)
    tr, accs, rates, C = _py_adapt(lambda x: -0.5 * x @ H @ x, init, np.eye(d), n, 20, w, win, 60, Z, U)
    np.testing.assert_array_equal(r["diag"]["accepted"], accs)
    np.testing.assert_allclose(r["theta_trace"], tr, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(r["diag"]["acceptance_rate"], rates, rtol=1e-13)
    np.testing.assert_allclose(r["sigma_empirical"][:, :, -1], C, rtol=1e-9, atol=1e-12)
    assert r["acceptance_rate"] == r["diag"]["acceptance_rate"][-1]


def test_rwMCMC_is_adaptMCMC_without_adaptation(oracle):
    """adaptMCMC_source.cpp:31: both adapt_*_start = 0 runs the plain random walk of :230-316."""
    rng = np.random.default_rng(4)
    n = 300
    Z, U = rng.standard_normal((n, 2)), rng.random(n)
    a = oracle.adaptMCMC(oracle.banana_target(), [1.0, 2.0], 2 * np.eye(2), n, 0, 0.0, 0, 0, Z, U, store_sigma=False)
    b = oracle.rwMCMC(oracle.banana_target(), [1.0, 2.0], 2 * np.eye(2), n, Z, U)
    np.testing.assert_array_equal(a["theta_trace"], b["theta_trace"])
    assert a["acceptance_rate"] == b["acceptance_rate"]
    # a Python callable works as the `target` closure and gives the same chain as the built-in
    B = 0.03
    c = oracle.rwMCMC(oracle.callable_target(lambda x: -(x[0] * x[0]) / 200 - 0.5 * (x[1] + B * (x[0] * x[0]) - 100 * B) ** 2),
                      [1.0, 2.0], 2 * np.eye(2), n, Z, U)
    np.testing.assert_allclose(c["theta_trace"], b["theta_trace"], rtol=1e-13)


def test_goldstein_price_and_banana_targets(oracle):
    # Goldstein-Price global minimum f(0,-1) = 3 -> log target = -log 3 (adaptMCMC_analysis.R:115-129)
    assert oracle.goldstein_price_target()([0.0, -1.0]) == pytest.approx(-np.log(3.0))
    assert oracle.goldstein_price_target()([1.0, 1.0]) == pytest.approx(-np.log((1 + 9 * 3) * (30 + 1 * 37)))
    assert oracle.banana_target(0.03)([10.0, 10.0]) == pytest.approx(-100 / 200 - 0.5 * (10 + 3 - 3) ** 2)


def test_index_helpers(oracle):
    """tmvrnormGibbs.cpp:12-37, 0-based."""
    x = np.arange(16.0).reshape(4, 4)
    np.testing.assert_array_equal(oracle.sub1(x, 1), np.delete(np.delete(x, 1, 0), 1, 1))
    np.testing.assert_array_equal(oracle.sub2(x, 2, 0), np.delete(x[2:3], 0, 1))
    np.testing.assert_array_equal(oracle.negSubCol(np.arange(5.0), 4), np.arange(4.0))


def test_gibbs_precompute_and_sweep_semantics(oracle):
    rng = np.random.default_rng(5)
    d = 5
    A = rng.standard_normal((d, d))
    S = A @ A.T / d + np.eye(d)
    st, P, sd = oracle.tmvn_precompute(S)
    H = np.linalg.inv(S)
    for i in range(d):
        others = [k for k in range(d) if k != i]
        np.testing.assert_allclose(P[i], -H[i, others] / H[i, i], rtol=1e-11)  # SURVEY 8 a11 equivalence
        assert sd[i] == pytest.approx(1 / np.sqrt(H[i, i]), rel=1e-12)
    # one sweep by hand: Gauss-Seidel order, U consumed as U[j*d+i] (tmvrnormGibbs.cpp:121-133)
    mu, lo, hi = rng.standard_normal(d), np.full(d, -1.0), np.full(d, 1.5)
    x = np.zeros(d)
    U = rng.random(2 * d)
    tr = oracle.rtmvnorm_gibbs(2, mu, S, lo, hi, x, U)
    y = x.copy()
    for j in range(2):
        for i in range(d):
            others = [k for k in range(d) if k != i]
            m = mu[i] + P[i] @ (y[others] - mu[others])
            Fa, Fb = sp.ndtr((lo[i] - m) / sd[i]), sp.ndtr((hi[i] - m) / sd[i])
            y[i] = m + sd[i] * sp.ndtri(U[j * d + i] * (Fb - Fa) + Fa)
        np.testing.assert_allclose(tr[j], y, rtol=1e-12)
    assert np.all(tr >= -1.0) and np.all(tr <= 1.5)


def test_rtnorm_reference_case_and_degenerate_tails(oracle):
    """tmvrnormGibbs.cpp:72-78 runs (-Inf, Inf): the draw is then qnorm(u); far tails give +-Inf."""
    u = np.random.default_rng(6).random(1000)
    np.testing.assert_allclose(oracle.rtnorm_gibbs(u.size, 0, 1, -np.inf, np.inf, u), sp.ndtri(u), rtol=1e-14, atol=1e-15)
    assert np.all(np.isinf(oracle.rtnorm_gibbs(8, 0, 1, 9.0, np.inf, u[:8])))        # lower bound > 8.3 sd
    s = oracle.rtnorm_gibbs(u.size, 2.0, 0.5, 1.0, 2.5, u)
    assert stats.kstest(s, stats.truncnorm(-2, 1, loc=2.0, scale=0.5).cdf).pvalue > 1e-3


def test_committed_golden_vectors_are_reproduced_bit_for_bit(oracle):
    g = np.load(os.path.join(GOLD, "g3_gibbs_diag4.npz"))
    np.testing.assert_array_equal(
        oracle.rtmvnorm_gibbs(int(g["n"]), g["mu"], g["sigma"], g["lower"], g["upper"], g["init"], g["U"]), g["trace"])
    g = np.load(os.path.join(GOLD, "g4_gibbs_corr2.npz"))
    np.testing.assert_array_equal(
        oracle.rtmvnorm_gibbs(int(g["n"]), g["mu"], g["sigma"], g["lower"], g["upper"], g["init"], g["U"]), g["trace"])
    for name, tgt in (("banana", oracle.banana_target(0.03)), ("goldpr", oracle.goldstein_price_target())):
        g = np.load(os.path.join(GOLD, f"g5_adapt_{name}.npz"))
        r = oracle.adaptMCMC(tgt, g["init"], np.eye(2), int(g["n"]), int(g["adapt_size_start"]), 0.0, 0,
                             int(g["adapt_shape_start"]), g["Z"], g["U"], diagnostics=True)
        np.testing.assert_array_equal(r["theta_trace"], g["theta_trace"])
        np.testing.assert_array_equal(r["diag"]["accepted"], g["accepted"])
        assert r["acceptance_rate"] == float(g["acceptance_rate"])
    g = np.load(os.path.join(GOLD, "g8_chol_fail.npz"))
    r = oracle.adaptMCMC(oracle.gaussian_target(np.zeros(3), np.eye(3)), np.zeros(3), np.eye(3), int(g["n"]), 0, 0.0, 0, 1,
                         g["Z"], g["U"], store_sigma=False)
    assert r["status"] == oracle.ORC_CHOL_FAILED and r["fail_iter"] == int(g["fail_iter"])
    g = np.load(os.path.join(GOLD, "g7_nan_paths.npz"))
    r = oracle.rwMCMC(oracle.gaussian_target(np.zeros(2), np.eye(2)), g["init"], np.eye(2), int(g["n"]), g["Z"], g["U"],
                      diagnostics=True)
    assert np.isnan(r["diag"]["log_acceptance"][10]) and r["diag"]["log_acceptance"][50] == -np.inf
    np.testing.assert_array_equal(r["theta_trace"], g["theta_trace"])


def test_density_helpers_and_walkers_restatement(oracle):
    """mvrGaussian_pdf / tmvrGaussian_pdf (deprecated/truncGaussian.cpp:11-78) against scipy,
    the lattice walkers (randomWalkers.cpp:9-74) against a numpy cumulative sum, the graph walker (:80-100) against a
    direct transcription of R's ProbSampleNoReplace for one draw."""
    from scipy.stats import multivariate_normal as mvn

    rng = np.random.default_rng(0)
    d = 4
    A = rng.standard_normal((d, d))
    S, mu, x = A @ A.T + np.eye(d), rng.standard_normal(d), rng.standard_normal(d)
    np.testing.assert_allclose(oracle.mvrGaussian_pdf(x, mu, S), mvn(mu, S).logpdf(x), rtol=1e-13)
    np.testing.assert_allclose(oracle.mvrGaussian_pdf(x, mu, S, do_log=False), mvn(mu, S).pdf(x), rtol=1e-13)
    lo, hi = x - 1, x + 1
    assert oracle.tmvrGaussian_pdf(x, mu, S, lo, hi, 0.5) == oracle.mvrGaussian_pdf(x, mu, S) / 0.5  # LOG density over cdf (:73-76)
    assert oracle.tmvrGaussian_pdf(x, mu, S, x + 0.1, hi, 0.5) == 0.0                                 # outside the support (:61-63)
    for dim in (2, 3):
        U = rng.random(99)
        w = oracle.random_walk(dim, 100, U)
        r = 1 + np.floor(U * 2 * dim)
        steps = np.zeros((99, dim))
        for k in range(dim):
            steps[:, k] = (r == 2 * k + 1).astype(float) - (r == 2 * k + 2)
        np.testing.assert_array_equal(w, np.vstack([np.zeros(dim), np.cumsum(steps, axis=0)]))
    nn = 6
    G = rng.random((nn, nn))
    U = rng.random(199)
    got = oracle.random_walkGraph(G, 1, 200, U)
    pos = 1
    for i in range(1, 200):
        p = G[pos] / G[pos].sum()
        order = np.argsort(-p, kind="stable")
        cum, j = 0.0, 0
        for j in range(nn - 1):
            cum += p[order[j]]
            if U[i - 1] <= cum:
                break
        else:
            j = nn - 1
        pos = int(order[j])
        assert got[i] == pos
