"""GPU parity: SURVEY.md section 8(f) row 4 -- the multivariate normal density helpers of deprecated/truncGaussian.cpp
and the walkers of randomWalkers.cpp (through the C ABI) against the oracle's restatement.

Integer paths (walkers) with injected uniforms: bit-exact.  Densities: FP64 within 1e-12 relative (the kernel gets
Sigma^-1 and det from a Cholesky factor, the reference / oracle from LU).  The normaliser of tmvrGaussian_pdf is
mvtnorm::pmvnorm in the reference -- third-party, randomized, absent here: PARITY UNPINNED; mcmc_pmvnorm is checked
against exact box probabilities (independent coordinates) and scipy's Genz integrator."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _spd(rng, d):
    A = rng.standard_normal((d, d))
    return A @ A.T / d + np.eye(d)


@pytest.mark.parametrize("d", [1, 2, 5, 17])
def test_mvrGaussian_pdf_matches_oracle(cuda, oracle, d):
    import mcmc_amd

    rng = np.random.default_rng(d)
    S, mu = _spd(rng, d), rng.standard_normal(d)
    X = mu[:, None] + 1.5 * rng.standard_normal((d, 300))
    lg = mcmc_amd.mvrGaussian_pdf(cuda.tensor(X, device="cuda"), mu, S).cpu().numpy()
    un = mcmc_amd.mvrGaussian_pdf(cuda.tensor(X, device="cuda"), mu, S, do_log=False).cpu().numpy()
    for c in range(0, 300, 37):
        np.testing.assert_allclose(lg[c], oracle.mvrGaussian_pdf(X[:, c], mu, S), rtol=1e-12)
        np.testing.assert_allclose(un[c], oracle.mvrGaussian_pdf(X[:, c], mu, S, do_log=False), rtol=1e-12)
    np.testing.assert_allclose(mcmc_amd.mvrGaussian_pdf(X[:, 3], mu, S), lg[3], rtol=0, atol=0)  # single point = batch of one
    # the reference's route: the un-logged density underflows far out, and its log is -Inf (:28-32), not a finite log-density
    far = mu + 40.0 * np.sqrt(np.diag(S))
    assert mcmc_amd.mvrGaussian_pdf(far, mu, S) == -np.inf == oracle.mvrGaussian_pdf(far, mu, S)


def test_tmvrGaussian_pdf_support_and_quirk(cuda, oracle):
    import mcmc_amd

    rng = np.random.default_rng(7)
    d = 3
    S, mu = _spd(rng, d), np.array([0.2, -0.1, 0.4])
    lower, upper = np.array([-1.0, -np.inf, 0.0]), np.array([1.0, 0.5, np.inf])
    X = rng.uniform(-1.5, 1.5, size=(d, 400))
    cdf = 0.1234
    got = mcmc_amd.tmvrGaussian_pdf(cuda.tensor(X, device="cuda"), mu, S, lower, upper, cdf).cpu().numpy()
    inside = np.all((X >= lower[:, None]) & (X <= upper[:, None]), axis=0)
    assert 20 < inside.sum() < 380
    assert np.all(got[~inside] == 0.0)                                    # :61-63
    for c in np.nonzero(inside)[0][:25]:
        want = oracle.tmvrGaussian_pdf(X[:, c], mu, S, lower, upper, cdf)
        np.testing.assert_allclose(got[c], want, rtol=1e-12)
        np.testing.assert_allclose(want, oracle.mvrGaussian_pdf(X[:, c], mu, S) / cdf, rtol=1e-15)  # LOG density over cdf (:73-76)
    assert mcmc_amd.tmvrGaussian_pdf(X[:, np.nonzero(~inside)[0][0]], mu, S, lower, upper, cdf) == 0.0


def test_pmvnorm_against_exact_and_scipy(cuda):
    import mcmc_amd
    from scipy import stats

    # independent coordinates: the box probability is a product of univariate differences
    mu, sd = np.array([0.0, 1.0, -0.5, 0.3]), np.array([1.0, 2.0, 0.5, 1.5])
    lower, upper = np.array([-1.0, 0.0, -np.inf, -2.0]), np.array([2.0, np.inf, 0.0, 2.5])
    exact = np.prod(stats.norm.cdf((upper - mu) / sd) - stats.norm.cdf((lower - mu) / sd))
    p, se = mcmc_amd.pmvnorm(lower, upper, mu, np.diag(sd ** 2), n_points=1 << 16, seed=3)
    assert abs(p - exact) < 1e-12 and se < 1e-8  # no randomness is left when Sigma is diagonal (se: cancellation in E f^2 - (E f)^2)
    # correlated: against scipy's integrator (also Genz), within a few standard errors of both
    rng = np.random.default_rng(11)
    d = 6
    S, m = _spd(rng, d), 0.2 * rng.standard_normal(d)
    lo, hi = -1.2 * np.ones(d), 1.7 * np.ones(d)
    p, se = mcmc_amd.pmvnorm(lo, hi, m, S, n_points=1 << 22, seed=5)
    ref = stats.multivariate_normal(m, S).cdf(hi, lower_limit=lo)
    assert 0 < se < 2e-4 and abs(p - ref) < 5 * se + 5e-4, (p, ref, se)
    p2, _ = mcmc_amd.pmvnorm(lo, hi, m, S, n_points=1 << 22, seed=6)
    assert p2 != p and abs(p2 - p) < 6 * se
    # and as the normaliser of the truncated density
    x = np.zeros(d)
    np.testing.assert_allclose(mcmc_amd.tmvrGaussian_pdf(x, m, S, lo, hi, n_points=1 << 22, seed=5),
                               mcmc_amd.mvrGaussian_pdf(x, m, S) / p, rtol=1e-13)


def test_pmvnorm_far_tail_boxes_are_finite(cuda):
    """a box more than ~8.3 sd above the mean made Phi(upper) - Phi(lower) = 1 - 1 and qnorm(1) = Inf, and
    the sum NaN.  Upper-tail intervals are mirrored now: small probabilities with full relative precision."""
    import mcmc_amd
    from scipy import stats

    mu, sd = np.zeros(3), np.array([1.0, 2.0, 0.5])
    lower, upper = np.array([9.0, 19.0, -np.inf]), np.array([np.inf, 30.0, 0.1])
    exact = np.prod(stats.norm.sf((lower - mu) / sd) - stats.norm.sf((upper - mu) / sd))
    p, se = mcmc_amd.pmvnorm(lower, upper, mu, np.diag(sd ** 2), n_points=1 << 14, seed=1)
    assert 0 < exact < 1e-35 and abs(p - exact) < 1e-10 * exact, (p, exact)
    S = np.array([[1.0, 0.6, 0.2], [0.6, 1.0, 0.3], [0.2, 0.3, 1.0]])
    p, se = mcmc_amd.pmvnorm(np.array([8.5, 8.5, 8.5]), np.full(3, np.inf), np.zeros(3), S, n_points=1 << 18, seed=2)
    assert np.isfinite(p) and np.isfinite(se) and 0 < p < stats.norm.sf(8.5)
    with pytest.raises(mcmc_amd.McmcError):
        mcmc_amd.pmvnorm(np.array([0.0, np.nan]), np.array([1.0, 1.0]), np.zeros(2), np.eye(2))
    with pytest.raises(mcmc_amd.McmcError):
        mcmc_amd.pmvnorm(np.zeros(2), np.ones(2), np.array([0.0, np.nan]), np.eye(2))


@pytest.mark.parametrize("dim", [2, 3])
def test_lattice_walkers_bit_exact(cuda, oracle, dim):
    import mcmc_amd

    rng = np.random.default_rng(dim)
    n = 500
    U = rng.random(n - 1)
    U[:4] = [0.0, 0.25, 0.5, 1.0 - 2 ** -53]  # the edges of the four / six bins
    fn = mcmc_amd.random_walk2d if dim == 2 else mcmc_amd.random_walk3d
    got = fn(n, U=U)
    want = oracle.random_walk(dim, n, U)
    assert got.shape == (n, dim)
    np.testing.assert_array_equal(got, want)
    assert np.all(got[0] == 0) and np.all(np.abs(np.diff(got, axis=0)).sum(axis=1) == 1)
    # many walkers, injected and Philox; a walker is keyed by its global id
    nw = 300
    Ub = rng.random((n - 1, nw))
    b = mcmc_amd.random_walk_batch(dim, n, nw, U=cuda.tensor(Ub, device="cuda")).cpu().numpy()
    for w in (0, 63, 64, 299):
        np.testing.assert_array_equal(b[:, :, w], oracle.random_walk(dim, n, Ub[:, w]))
    whole = mcmc_amd.random_walk_batch(dim, n, nw, seed=9)
    part = mcmc_amd.random_walk_batch(dim, n, 100, seed=9, walker_offset=200)
    assert cuda.equal(whole[:, :, 200:], part)
    end = whole[-1].cpu().numpy()  # E = 0, Var = (n - 1) / dim per axis
    assert np.all(np.abs(end.mean(axis=1)) < 5 * np.sqrt((n - 1) / dim / nw))
    assert fn(1).shape == (1, dim)


def test_graph_walker_bit_exact(cuda, oracle):
    import mcmc_amd

    rng = np.random.default_rng(5)
    nn, n = 7, 400
    G = rng.random((nn, nn)) * (rng.random((nn, nn)) < 0.7)
    G[np.arange(nn), (np.arange(nn) + 1) % nn] += 0.1  # every node has a way out; weights need not be normalised (FixProb does)
    U = rng.random(n - 1)
    got = mcmc_amd.random_walkGraph(G, 2, n, U=U)
    want = oracle.random_walkGraph(G, 2, n, U)
    np.testing.assert_array_equal(got, want)
    assert got[0] == 2 and np.all((got >= 0) & (got < nn))
    assert np.all(G[got[:-1].astype(int), got[1:].astype(int)] > 0)  # only edges with positive weight are walked
    nw = 200
    Ub = rng.random((n - 1, nw))
    b = mcmc_amd.random_walkGraph_batch(G, 0, n, nw, U=cuda.tensor(Ub, device="cuda")).cpu().numpy()
    for w in (0, 64, 199):
        np.testing.assert_array_equal(b[:, w], oracle.random_walkGraph(G, 0, n, Ub[:, w]))
    # Philox: long-run visit frequencies follow the stationary law of the normalised transition matrix
    P = G / G.sum(axis=1, keepdims=True)
    ev, evec = np.linalg.eig(P.T)
    pi = np.real(evec[:, np.argmax(np.real(ev))])
    pi = pi / pi.sum()
    v = mcmc_amd.random_walkGraph_batch(G, 0, 600, 4096, seed=4)[300:].cpu().numpy().astype(int)
    freq = np.bincount(v.reshape(-1), minlength=nn) / v.size
    assert np.max(np.abs(freq - pi)) < 0.01
    with pytest.raises(mcmc_amd.McmcError):
        mcmc_amd.random_walkGraph(-G, 0, 10)
