"""GPU suite: the device kernels behind `rhat_max` and `ess_per_sec` of the bench line (diag.hip), driven through the
C ABI (mcmc_chain_moments, mcmc_rhat_partial, mcmc_pooled_cov_partial, mcmc_chain_ess[_capped]).

The reference defines neither diagnostic (coda is loaded at MCMCutilty.R:11 and never called), so the pins are the
textbook definitions: AR(1) series x_t = phi x_{t-1} + eps have ESS/N = (1 - phi)/(1 + phi); Gelman-Rubin from
within / between variances; numpy for the plain moments.  SURVEY.md section 8(d) "Metric"."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PHIS = (0.0, 0.5, 0.9, 0.99)


def _ar1(n, phis, n_chains, seed):
    """[n][len(phis)][n_chains], stationary start, unit innovation variance."""
    rng = np.random.default_rng(seed)
    phi = np.asarray(phis)[:, None]
    x = np.empty((n, len(phis), n_chains))
    x[0] = rng.standard_normal((len(phis), n_chains)) / np.sqrt(1.0 - phi * phi)
    for t in range(1, n):
        x[t] = phi * x[t - 1] + rng.standard_normal((len(phis), n_chains))
    return x


def _geyer_ess(x, max_lag):
    """numpy restatement of chain_ess_kernel's definition for ONE series: (ess, capped)."""
    N = len(x)
    v = x - x.sum() / N
    g0 = float(np.dot(v, v))
    if g0 <= 0.0 or N < 2 or not x.max() > x.min():  # a series that never moved: ESS 0 by definition
        return 0.0, 0
    tau, kmax = -1.0, min(max_lag, N - 1)
    capped = int(kmax < N - 1)
    for lag in range(0, kmax + 1, 2):
        ga = np.dot(v[:N - lag], v[lag:])
        gb = np.dot(v[:N - lag - 1], v[lag + 1:]) if lag + 1 <= kmax else 0.0
        pair = (ga + gb) / g0
        if pair <= 0.0:
            capped = 0
            break
        tau += 2.0 * pair
    return (N / tau if tau > 0 else float(N)), capped


@pytest.fixture(scope="module")
def ar1_trace(cuda):
    x = _ar1(20000, PHIS, 512, seed=11)
    return x, cuda.tensor(x, device="cuda")


def test_ess_of_ar1_series_matches_theory(cuda, ar1_trace):
    from mcmc_amd import diagnostics as diag

    x, xd = ar1_trace
    N = x.shape[0]
    ess, capped = diag.chain_ess(xd, 0, 5000, return_capped=True)
    ess, capped = ess.cpu().numpy(), capped.cpu().numpy()
    assert ess.shape == (len(PHIS), 512)
    for k, phi in enumerate(PHIS):
        want = (1.0 - phi) / (1.0 + phi)
        got = ess[k].mean() / N
        assert abs(got - want) < 0.10 * want, (phi, got, want)  # mean ESS/N within 10 % of theory
    assert capped.sum() == 0  # max_lag 5000 covers the decay of every series: Geyer's rule ended every sum
    # the same numbers as the numpy restatement of the definition, chain by chain (summation order differs: 1e-9)
    for k in range(len(PHIS)):
        for c in (0, 17, 511):
            e, cp = _geyer_ess(x[:, k, c], 5000)
            assert cp == 0
            np.testing.assert_allclose(ess[k, c], e, rtol=1e-9)
    # the plain entry point (no flag array) gives the same ESS
    assert np.array_equal(diag.chain_ess(xd, 0, 5000).cpu().numpy(), ess)


def test_ess_is_overstated_when_max_lag_cuts_the_sum(cuda, ar1_trace):
    """phi = 0.99 has an autocorrelation time of 199 draws: a sum cut at lag 50 sees a quarter of it, the ESS comes out
    ~3x too large, and the flag array says so for every chain -- what `ess_lag_capped_fraction` of the bench line reports."""
    from mcmc_amd import diagnostics as diag

    x, xd = ar1_trace
    N = x.shape[0]
    ess, capped = diag.chain_ess(xd, 0, 50, return_capped=True)
    ess, capped = ess.cpu().numpy(), capped.cpu().numpy()
    k = PHIS.index(0.99)
    want = (1.0 - 0.99) / (1.0 + 0.99)
    assert ess[k].mean() / N > 2.0 * want
    assert capped[k].mean() == 1.0
    assert capped[PHIS.index(0.0)].mean() == 0.0 and capped[PHIS.index(0.5)].mean() < 0.01
    e, cp = _geyer_ess(x[:, k, 5], 50)
    assert cp == 1
    np.testing.assert_allclose(ess[k, 5], e, rtol=1e-9)


def test_ess_edge_cases(cuda):
    from mcmc_amd import diagnostics as diag

    rng = np.random.default_rng(3)
    n, d, nc = 400, 3, 70  # ragged chain count (not a multiple of the wavefront)
    x = rng.standard_normal((n, d, nc))
    x[:, 1, 5] = 2.5            # a frozen coordinate: no information
    x[100:, :, 9] = x[100, :, 9]  # a chain frozen after the burn-in (C4's permanent NaN-reject)
    xd = cuda.tensor(x, device="cuda")
    ess, capped = diag.chain_ess(xd, 100, 150, return_capped=True)
    ess, capped = ess.cpu().numpy(), capped.cpu().numpy()
    assert ess[1, 5] == 0.0 and capped[1, 5] == 0
    assert np.all(ess[:, 9] == 0.0)
    for k, c in ((0, 0), (2, 69), (1, 4)):
        e, cp = _geyer_ess(x[100:, k, c], 150)  # burn_kept rows are skipped
        np.testing.assert_allclose(ess[k, c], e, rtol=1e-10)
        assert capped[k, c] == cp
    # max_lag beyond the series: cut at N - 1, never flagged
    ess2, capped2 = diag.chain_ess(xd, 0, 10 ** 6, return_capped=True)
    assert capped2.sum().item() == 0
    np.testing.assert_allclose(ess2.cpu().numpy()[0, 3], _geyer_ess(x[:, 0, 3], 10 ** 6)[0], rtol=1e-10)
    # an anti-correlated series has ESS > N (tau < 1): the estimator does not clip it
    y = np.empty((2000, 1, 64))
    y[0] = rng.standard_normal((1, 64))
    for t in range(1, 2000):
        y[t] = -0.5 * y[t - 1] + rng.standard_normal((1, 64))
    e = diag.chain_ess(cuda.tensor(y, device="cuda"), 0, 500).cpu().numpy()
    assert 2.0 < e.mean() / 2000 < 4.0  # (1 + 0.5)/(1 - 0.5) = 3


def test_chain_moments_match_numpy(cuda):
    from mcmc_amd import diagnostics as diag

    rng = np.random.default_rng(5)
    n, d, nc = 501, 7, 333
    x = 1e3 + rng.standard_normal((n, d, nc)) * rng.uniform(0.1, 10.0, (1, d, nc))  # large offset: Welford, not sum of squares
    xd = cuda.tensor(x, device="cuda")
    for burn in (0, 1, 250, 499):
        mean, var = diag.chain_moments(xd, burn)
        np.testing.assert_allclose(mean.cpu().numpy(), x[burn:].mean(0), rtol=1e-12)
        np.testing.assert_allclose(var.cpu().numpy(), x[burn:].var(0, ddof=1), rtol=1e-9)
    mean, var = diag.chain_moments(xd, 500)  # one draw left: its value, variance 0
    assert np.array_equal(mean.cpu().numpy(), x[500]) and np.all(var.cpu().numpy() == 0.0)
    # centred data: the variance itself to 1e-12
    z = rng.standard_normal((n, d, nc))
    mean, var = diag.chain_moments(cuda.tensor(z, device="cuda"), 100)
    np.testing.assert_allclose(var.cpu().numpy(), z[100:].var(0, ddof=1), rtol=1e-12)
    np.testing.assert_allclose(mean.cpu().numpy(), z[100:].mean(0), rtol=1e-12, atol=1e-15)


def _partial(mean, var):
    """the contract tests/test_distributed_cpu.py reproduces in numpy: sum m, sum m^2, sum s^2, count"""
    d = mean.shape[0]
    out = np.empty(3 * d + 1)
    out[:d], out[d:2 * d], out[2 * d:3 * d], out[3 * d] = mean.sum(1), (mean ** 2).sum(1), var.sum(1), mean.shape[1]
    return out


@pytest.mark.parametrize("nc", [1, 37, 1024, 5000])
def test_rhat_partial_kernel_and_rhat(cuda, nc):
    from mcmc_amd import diagnostics as diag

    rng = np.random.default_rng(nc)
    n, d = 300, 6
    offs = np.array([0.0, 0.0, 0.3, 0.3, 1.0, 3.0])[None, :, None]  # injected between-chain offsets per coordinate
    x = rng.standard_normal((n, d, nc)) + offs * rng.standard_normal((1, d, nc))
    xd = cuda.tensor(x, device="cuda")
    mean, var = diag.chain_moments(xd, 0)
    part = diag.rhat_partial(mean, var).cpu().numpy()
    np.testing.assert_allclose(part, _partial(mean.cpu().numpy(), var.cpu().numpy()), rtol=1e-12, atol=1e-12)
    assert part[3 * d] == nc
    if nc == 1:
        return  # one chain: no between-chain variance to speak of
    rh = diag.rhat(mean, var, n)
    m, s2 = x.mean(0), x.var(0, ddof=1)
    W, B_over_n = s2.mean(1), m.var(1, ddof=1)
    np.testing.assert_allclose(rh, np.sqrt(((n - 1) / n * W + B_over_n) / W), rtol=1e-10)
    if nc >= 1024:
        assert np.all(np.abs(rh[:2] - 1.0) < 0.01)
        assert rh[2] > 1.03 and rh[4] > 1.3 and rh[5] > 2.5


def test_rhat_partial_is_deterministic_and_shard_additive(cuda):
    """the partial sums of two shards add up to those of the whole (what the all-reduce relies on)"""
    from mcmc_amd import diagnostics as diag

    rng = np.random.default_rng(9)
    d, nc = 10, 4099
    mean = cuda.tensor(rng.standard_normal((d, nc)), device="cuda")
    var = cuda.tensor(rng.uniform(0.5, 2.0, (d, nc)), device="cuda")
    whole = diag.rhat_partial(mean, var)
    assert cuda.equal(whole, diag.rhat_partial(mean, var))
    a = diag.rhat_partial(mean[:, :2000].contiguous(), var[:, :2000].contiguous())
    b = diag.rhat_partial(mean[:, 2000:].contiguous(), var[:, 2000:].contiguous())
    np.testing.assert_allclose((a + b).cpu().numpy(), whole.cpu().numpy(), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(diag.rhat_from_partial(a + b, 100), diag.rhat_from_partial(whole, 100), rtol=1e-11)


@pytest.mark.parametrize("d,nc", [(3, 11), (100, 700)])
def test_pooled_cov_partial_kernel(cuda, d, nc):
    """mcmc_pooled_cov_partial at C4's d = 100: sum_c C_c, sum_c m_c m_c^T, sum_c m_c, count -> pooled covariance of all draws"""
    from mcmc_amd import diagnostics as diag

    rng = np.random.default_rng(d)
    n = 40
    A = rng.standard_normal((d, d)) / np.sqrt(d)
    draws = np.einsum("kl,tlc->tkc", A, rng.standard_normal((n, d, nc))) + rng.standard_normal((1, d, nc))
    m = draws.mean(0)
    Cc = np.einsum("tkc,tlc->klc", draws - m, draws - m) / n  # population covariance per chain (update_sigma's)
    cov_flat = Cc.transpose(1, 0, 2).reshape(d * d, nc)       # column-major d x d per chain: element (r, cl) at cl*d + r
    part = diag.pooled_cov_partial(cuda.tensor(m, device="cuda"), cuda.tensor(np.ascontiguousarray(cov_flat), device="cuda"))
    p = part.cpu().numpy()
    assert p.shape == (2 * d * d + d + 1,) and p[-1] == nc
    np.testing.assert_allclose(p[:d * d].reshape(d, d, order="F"), Cc.sum(2), rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(p[d * d:2 * d * d].reshape(d, d, order="F"), np.einsum("kc,lc->kl", m, m), rtol=1e-11, atol=1e-10)
    np.testing.assert_allclose(p[2 * d * d:2 * d * d + d], m.sum(1), rtol=1e-11, atol=1e-11)
    mbar, S = diag.pooled_covariance(cuda.tensor(m, device="cuda"), cuda.tensor(np.ascontiguousarray(cov_flat), device="cuda"))
    flat = draws.transpose(1, 0, 2).reshape(d, -1)
    np.testing.assert_allclose(mbar, flat.mean(1), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(S, np.cov(flat, bias=True), rtol=1e-9, atol=1e-11)


def test_diagnostics_reject_bad_arguments(cuda):
    from mcmc_amd import diagnostics as diag
    from mcmc_amd._lib import McmcError

    x = cuda.zeros((10, 2, 8), dtype=cuda.float64, device="cuda")
    with pytest.raises(McmcError):
        diag.chain_moments(x, 10)  # nothing left after the burn-in
    with pytest.raises(McmcError):
        diag.chain_ess(x, 0, 0)    # max_lag < 1
    with pytest.raises(ValueError):
        diag.pooled_cov_partial(x[0], x[1])
