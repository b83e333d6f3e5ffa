"""GPU parity: truncated-normal / TMVN Gibbs kernels (through the C ABI) vs the CPU oracle.

Bar (BASELINE.json north_star): with identical injected uniforms, trajectories agree within
1e-12 relative in FP64.  The inverse-CDF formula the reference uses (tmvrnormGibbs.cpp:130-132)
is ill conditioned when u*(Fb-Fa)+Fa rounds towards 1 (dz = dp / phi(z)), so where that happens
the comparison allows the forward error of a 4-ulp change in p on top of the 1e-12 -- the
tolerance is written out in `_tol` below.
"""
import os

import numpy as np
import pytest
from scipy.stats import norm as _norm

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
RTOL = 1e-12
EPS = np.finfo(np.float64).eps


def _tol(want, mu, sd):
    """1e-12 relative + sd * (4 eps) / phi(z): forward error of the reference formula itself."""
    z = (want - mu) / sd
    return RTOL * np.maximum(1.0, np.abs(want)) + sd * 4 * EPS / np.maximum(_norm.pdf(z), 1e-300)


def _cond_tol(want, init, mu, P, sd):
    """`_tol` with the conditional mean each draw was made around (tmvrnormGibbs.cpp:124-127), rebuilt from the
    oracle's own trajectory: coordinate i of sweep j sees x_0..x_(i-1) of sweep j and x_(i+1).. of sweep j-1."""
    n, d = want.shape
    prev = np.vstack([np.asarray(init, dtype=np.float64)[None, :], want[:-1]])
    tol = np.empty_like(want)
    for i in range(d):
        x = np.concatenate([want[:, :i], prev[:, i + 1:]], axis=1) - np.delete(mu, i)[None, :]
        tol[:, i] = _tol(want[:, i], mu[i] + x @ P[i], sd[i])
    return tol


def _c3_params(d=10, rho=0.5, lo=-1.0, hi=2.0):
    idx = np.arange(d)
    sigma = rho ** np.abs(idx[:, None] - idx[None, :])
    return np.zeros(d), sigma, np.full(d, lo), np.full(d, hi), np.zeros(d)


def test_rtnorm_grid_matches_oracle_and_golden(cuda, oracle):
    import mcmc_amd

    g = np.load(os.path.join(GOLD, "g2_rtnorm.npz"))
    u = g["u"]
    for (mu, sd, key) in ((0.0, 1.0, "out_std"), (1.5, 2.5, "out_mu1p5_sd2p5")):
        for bi, (a, b) in enumerate(g["bounds"]):
            got = mcmc_amd.rtnorm_gibbs(u.size, mu, sd, a, b, F=u)
            want = oracle.rtnorm_gibbs(u.size, mu, sd, a, b, u)
            np.testing.assert_array_equal(want, g[key][bi])  # oracle still equals the committed fixture
            fin = np.isfinite(want)
            # where the reference formula degenerates to +-Inf the kernel must degenerate too
            np.testing.assert_array_equal(np.isfinite(got), fin)
            np.testing.assert_array_equal(got[~fin], want[~fin])
            assert np.all(np.abs(got[fin] - want[fin]) <= _tol(want[fin], mu, sd)), (mu, sd, a, b)


def test_rtnorm_empty_and_device_variant(cuda, oracle):
    import mcmc_amd

    assert mcmc_amd.rtnorm_gibbs(0, 0.0, 1.0, -1.0, 1.0).shape == (0,)
    rng = np.random.default_rng(3)
    F = rng.random(100_003)
    out = mcmc_amd.rtnorm_batch(F.size, 0.3, 1.7, -2.0, 0.5, F=cuda.tensor(F, device="cuda")).cpu().numpy()
    want = oracle.rtnorm_gibbs(F.size, 0.3, 1.7, -2.0, 0.5, F)
    assert np.all(np.abs(out - want) <= _tol(want, 0.3, 1.7))
    assert out.min() >= -2.0 and out.max() <= 0.5


@pytest.mark.parametrize("name", ["g3_gibbs_diag4.npz", "g4_gibbs_corr2.npz"])
def test_reference_fixtures_single_chain(cuda, oracle, name):
    """The two inputs the reference author ran (tmvrnormGibbs.cpp:146,153-159)."""
    import mcmc_amd

    g = np.load(os.path.join(GOLD, name))
    n = int(g["n"])
    got = mcmc_amd.rtmvnorm_gibbs(n, g["mu"], g["sigma"], g["lower"], g["upper"], g["init"], U=g["U"])
    assert got.shape == (n, g["mu"].size)
    want = g["trace"]
    _, P, sd = oracle.tmvn_precompute(g["sigma"])
    tol = _cond_tol(want, g["init"], g["mu"], P, sd)  # 1e-12 relative + the formula's own conditioning, per draw
    assert np.all(np.abs(got - want) <= tol), np.max(np.abs(got - want) / tol)


def test_precompute_matches_reference_partitioning(oracle):
    """P_i, sd_i from the precision matrix (the engine) vs from d sub-block inverses
    (tmvrnormGibbs.cpp:104-113, the oracle).  Host-side, but kept with the Gibbs parity tests."""
    import mcmc_amd

    rng = np.random.default_rng(5)
    for d in (2, 5, 10, 33):
        A = rng.standard_normal((d, d))
        sigma = A @ A.T / d + np.eye(d)
        P, sd = mcmc_amd.tmvn_precompute(sigma)
        st, Po, sdo = oracle.tmvn_precompute(sigma)
        assert st == 0
        np.testing.assert_allclose(P, Po, rtol=0, atol=1e-13)
        np.testing.assert_allclose(sd, sdo, rtol=1e-13)


@pytest.mark.parametrize("d,n_chains,n_sweeps", [(10, 200, 300), (3, 70, 500), (16, 64, 100), (1, 65, 50)])
def test_batch_lane_kernel_injected(cuda, oracle, d, n_chains, n_sweeps):
    import mcmc_amd

    rng = np.random.default_rng(100 + d)
    mu, sigma, lower, upper, _ = _c3_params(d)
    mu = mu + 0.1 * np.arange(d)
    init = rng.uniform(-0.5, 1.5, size=(d, n_chains))
    U = rng.random((n_sweeps, d, n_chains))
    out = mcmc_amd.tmvn_gibbs_batch(n_chains, n_sweeps, mu, sigma, lower, upper, cuda.tensor(init, device="cuda"),
                                    U=cuda.tensor(U, device="cuda"), moments=True)
    cuda.cuda.synchronize()
    trace = out["trace"].cpu().numpy()
    assert trace.shape == (n_sweeps, d, n_chains)
    assert np.all(out["status"].cpu().numpy() == 0)
    for c in sorted({0, 1, n_chains // 2, n_chains - 1}):
        want = oracle.rtmvnorm_gibbs(n_sweeps, mu, sigma, lower, upper, init[:, c], U[:, :, c].reshape(-1))
        err = np.max(np.abs(trace[:, :, c] - want) / np.maximum(1.0, np.abs(want)))
        assert err < RTOL, (c, err)
        np.testing.assert_array_equal(out["final_state"].cpu().numpy()[:, c], trace[-1, :, c])
        np.testing.assert_allclose(out["mean"].cpu().numpy()[:, c], want.mean(axis=0), rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(out["m2"].cpu().numpy()[:, c], ((want - want.mean(0)) ** 2).sum(0), rtol=1e-10)


@pytest.mark.parametrize("d,n_chains,n_sweeps", [(24, 100, 60), (33, 300, 20), (64, 130, 15), (65, 70, 12), (100, 130, 10), (128, 70, 12), (200, 65, 5)])
def test_batch_lds_kernel_injected(cuda, oracle, d, n_chains, n_sweeps):
    import mcmc_amd

    rng = np.random.default_rng(200 + d)
    mu, sigma, lower, upper, _ = _c3_params(d, rho=0.3)
    init = rng.uniform(-0.5, 1.5, size=(d, n_chains))
    U = rng.random((n_sweeps, d, n_chains))
    out = mcmc_amd.tmvn_gibbs_batch(n_chains, n_sweeps, mu, sigma, lower, upper, cuda.tensor(init, device="cuda"),
                                    U=cuda.tensor(U, device="cuda"), moments=True, burn=2)
    cuda.cuda.synchronize()
    trace = out["trace"].cpu().numpy()
    for c in (0, 63, 64, n_chains - 1):
        want = oracle.rtmvnorm_gibbs(n_sweeps, mu, sigma, lower, upper, init[:, c], U[:, :, c].reshape(-1))
        err = np.max(np.abs(trace[:, :, c] - want) / np.maximum(1.0, np.abs(want)))
        assert err < RTOL, (c, err)
        np.testing.assert_array_equal(out["final_state"].cpu().numpy()[:, c], trace[-1, :, c])
        np.testing.assert_allclose(out["mean"].cpu().numpy()[:, c], want[2:].mean(axis=0), rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(out["m2"].cpu().numpy()[:, c], ((want[2:] - want[2:].mean(0)) ** 2).sum(0), rtol=1e-10)


@pytest.mark.parametrize("tail_safe", [False, True])
@pytest.mark.parametrize("d,n_chains,n_sweeps", [(28, 70, 12), (48, 130, 12), (100, 70, 8), (129, 65, 5), (200, 65, 4)])
def test_wide_kernel_infinite_bounds(cuda, oracle, d, n_chains, n_sweeps, tail_safe):
    """One-sided truncation (the reference's own example is c(-Inf,-Inf,0,0), tmvrnormGibbs.cpp:146) in the d > 27
    kernel: a dot wave's 64-wide window must never meet the +-Inf bounds (Inf * 0 = NaN).  d is chosen so that the
    window of the last dot wave runs past coordinate d (28, 48 < 64; 100 < 128; 129, 200 < 256)."""
    import mcmc_amd

    rng = np.random.default_rng(300 + d)
    mu, sigma, _, _, _ = _c3_params(d, rho=0.3)
    mu = mu + 0.02 * np.arange(d)
    k = np.arange(d)
    lower = np.where(k % 3 == 0, -np.inf, -1.0)
    upper = np.where(k % 3 == 1, np.inf, 2.0)
    lower[-1], upper[-1] = -np.inf, np.inf  # and an untruncated coordinate, both bounds infinite
    init = rng.uniform(-0.5, 1.5, size=(d, n_chains))
    U = rng.random((n_sweeps, d, n_chains))
    out = mcmc_amd.tmvn_gibbs_batch(n_chains, n_sweeps, mu, sigma, lower, upper, cuda.tensor(init, device="cuda"),
                                    U=cuda.tensor(U, device="cuda"), moments=True, tail_safe=tail_safe)
    cuda.cuda.synchronize()
    assert np.all(out["status"].cpu().numpy() == 0)
    trace = out["trace"].cpu().numpy()
    assert np.all(np.isfinite(trace))
    _, P, sd = oracle.tmvn_precompute(sigma)
    for c in (0, 63, n_chains - 1):
        want = oracle.rtmvnorm_gibbs(n_sweeps, mu, sigma, lower, upper, init[:, c], U[:, :, c].reshape(-1))
        tol = _cond_tol(want, init[:, c], mu, P, sd)
        assert np.all(np.abs(trace[:, :, c] - want) <= tol), (c, np.max(np.abs(trace[:, :, c] - want) / tol))
        np.testing.assert_allclose(out["mean"].cpu().numpy()[:, c], trace[:, :, c].mean(axis=0), rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(out["m2"].cpu().numpy()[:, c], ((trace[:, :, c] - trace[:, :, c].mean(0)) ** 2).sum(0),
                                   rtol=1e-10)


def test_thinning_keeps_every_kth_sweep(cuda):
    import mcmc_amd

    mu, sigma, lower, upper, init = _c3_params(10)
    full = mcmc_amd.tmvn_gibbs_batch(128, 40, mu, sigma, lower, upper, init, seed=7)["trace"]
    thin = mcmc_amd.tmvn_gibbs_batch(128, 40, mu, sigma, lower, upper, init, seed=7, thin=4)["trace"]
    assert thin.shape[0] == 10
    assert cuda.equal(thin, full[3::4])


def test_philox_run_replays_through_oracle(cuda, oracle):
    """A Philox-mode run equals the oracle fed with the uniforms mcmc_gibbs_streams reports."""
    import mcmc_amd

    d, n_chains, n_sweeps = 10, 130, 100
    mu, sigma, lower, upper, init = _c3_params(d)
    out = mcmc_amd.tmvn_gibbs_batch(n_chains, n_sweeps, mu, sigma, lower, upper, init, seed=2, chain_offset=5)
    U = mcmc_amd.gibbs_streams(2, n_chains, n_sweeps, d, chain_offset=5).cpu().numpy()
    assert U.min() > 0.0 and U.max() < 1.0
    trace = out["trace"].cpu().numpy()
    for c in (0, 64, 129):
        want = oracle.rtmvnorm_gibbs(n_sweeps, mu, sigma, lower, upper, init, U[:, :, c].reshape(-1))
        assert np.max(np.abs(trace[:, :, c] - want) / np.maximum(1.0, np.abs(want))) < RTOL


def test_results_do_not_depend_on_sharding(cuda):
    """Chains are keyed by global id: [0,192) in one launch == [0,64) + [64,192) in two."""
    import mcmc_amd

    mu, sigma, lower, upper, init = _c3_params(10)
    whole = mcmc_amd.tmvn_gibbs_batch(192, 50, mu, sigma, lower, upper, init, seed=11)["trace"]
    a = mcmc_amd.tmvn_gibbs_batch(64, 50, mu, sigma, lower, upper, init, seed=11, chain_offset=0)["trace"]
    b = mcmc_amd.tmvn_gibbs_batch(128, 50, mu, sigma, lower, upper, init, seed=11, chain_offset=64)["trace"]
    assert cuda.equal(whole[:, :, :64], a) and cuda.equal(whole[:, :, 64:], b)
    # and the LDS kernel (d > 16)
    mu, sigma, lower, upper, init = _c3_params(20)
    whole = mcmc_amd.tmvn_gibbs_batch(100, 20, mu, sigma, lower, upper, init, seed=11)["trace"]
    b = mcmc_amd.tmvn_gibbs_batch(36, 20, mu, sigma, lower, upper, init, seed=11, chain_offset=64)["trace"]
    assert cuda.equal(whole[:, :, 64:], b)


def test_philox_moments_match_truncated_normal(cuda):
    """Philox run, independent coordinates: per-coordinate KS test and mean/variance against the
    exact truncated normal (scipy), pooled over chains after burn-in."""
    import mcmc_amd
    from scipy import stats

    d, n_chains, n_sweeps = 4, 4096, 60
    mu = np.array([0.0, 1.0, -0.5, 2.0])
    sds = np.array([1.0, 2.0, 0.5, 1.5])
    lower = np.array([-1.0, 0.0, -np.inf, 2.5])
    upper = np.array([2.0, np.inf, 0.0, 6.0])
    out = mcmc_amd.tmvn_gibbs_batch(n_chains, n_sweeps, mu, np.diag(sds ** 2), lower, upper, mu.clip(lower + 0.1, upper - 0.1),
                                    seed=99)
    x = out["trace"][10:].cpu().numpy()  # [t, d, chains]
    for k in range(d):
        a, b = (lower[k] - mu[k]) / sds[k], (upper[k] - mu[k]) / sds[k]
        dist = stats.truncnorm(a, b, loc=mu[k], scale=sds[k])
        s = x[:, k, :].reshape(-1)[:: 7][:20000]
        assert stats.kstest(s, dist.cdf).pvalue > 1e-3
        full = x[:, k, :].reshape(-1)
        assert abs(full.mean() - dist.mean()) < 5 * dist.std() / np.sqrt(full.size)
        assert abs(full.var() - dist.var()) < 0.02 * dist.var()


def test_tail_safe_path_where_reference_is_finite_and_beyond(cuda, oracle):
    """tail_safe=1 is the same draw taken in the mirrored tail: equal to the reference formula where
    that is well conditioned, finite where the reference returns +Inf (bound > 8.3 sd out)."""
    import mcmc_amd

    rng = np.random.default_rng(8)
    F = rng.random(20000)
    Fd = cuda.tensor(F, device="cuda")
    # well conditioned: both paths agree
    ref = mcmc_amd.rtnorm_batch(F.size, 0.0, 1.0, 1.0, 3.0, F=Fd).cpu().numpy()
    safe = mcmc_amd.rtnorm_batch(F.size, 0.0, 1.0, 1.0, 3.0, F=Fd, tail_safe=True).cpu().numpy()
    assert np.max(np.abs(ref - safe)) < 1e-11
    # lower bound 9 sd out: reference formula gives +Inf (oracle agrees), tail-safe gives the right law
    want = oracle.rtnorm_gibbs(16, 0.0, 1.0, 9.0, np.inf, F[:16])
    assert np.all(np.isinf(want))
    safe = mcmc_amd.rtnorm_batch(F.size, 0.0, 1.0, 9.0, np.inf, F=Fd, tail_safe=True).cpu().numpy()
    assert np.all(np.isfinite(safe)) and safe.min() >= 9.0
    # exact law: P(X > x | X > 9) = Q(x)/Q(9); compare through the survival function
    from scipy.stats import norm
    p = 1.0 - norm.sf(safe) / norm.sf(9.0)
    assert np.max(np.abs(p - F)) < 1e-9
    # the Gibbs kernel reports non-finite states instead of hiding them
    out = mcmc_amd.tmvn_gibbs_batch(64, 5, [0.0, 0.0], np.eye(2), [9.0, -1.0], [np.inf, 1.0], [9.5, 0.0], seed=1)
    st = out["status"].cpu().numpy()
    assert np.all((st & 15) == mcmc_amd.MCMC_CHAIN_NONFINITE_STATE) and np.all((st >> 4) == 1)
    out = mcmc_amd.tmvn_gibbs_batch(64, 5, [0.0, 0.0], np.eye(2), [9.0, -1.0], [np.inf, 1.0], [9.5, 0.0], seed=1,
                                    tail_safe=True)
    assert np.all(out["status"].cpu().numpy() == 0) and bool(cuda.isfinite(out["trace"]).all())


def test_argument_validation(cuda):
    import mcmc_amd

    mu, sigma, lower, upper, init = _c3_params(4)
    with pytest.raises(mcmc_amd.McmcError):
        mcmc_amd.tmvn_gibbs_batch(8, 5, mu, sigma, upper, lower, init)  # lower > upper
    bad = sigma.copy()
    bad[0, 0] = -1.0
    with pytest.raises(mcmc_amd.McmcError):
        mcmc_amd.tmvn_gibbs_batch(8, 5, mu, bad, lower, upper, init)
    with pytest.raises(mcmc_amd.McmcError):
        mcmc_amd.tmvn_gibbs_batch(8, 0, mu, sigma, lower, upper, init)


def test_wide_kernel_philox_replay_sharding_thinning_and_tail(cuda, oracle):
    """The d > 32 kernel (draw / dot wavefront pairs, rows staged through LDS): a Philox run replayed through
    the oracle, invariance to sharding and thinning, and the tail-safe path with running moments only."""
    import mcmc_amd

    d, n_chains, n_sweeps = 48, 200, 12
    mu, sigma, lower, upper, init = _c3_params(d, rho=0.4)
    mu = mu + 0.05 * np.arange(d)
    out = mcmc_amd.tmvn_gibbs_batch(n_chains, n_sweeps, mu, sigma, lower, upper, init, seed=21, chain_offset=9)
    U = mcmc_amd.gibbs_streams(21, n_chains, n_sweeps, d, chain_offset=9).cpu().numpy()
    trace = out["trace"].cpu().numpy()
    for c in (0, 63, 64, 199):
        want = oracle.rtmvnorm_gibbs(n_sweeps, mu, sigma, lower, upper, init, U[:, :, c].reshape(-1))
        assert np.max(np.abs(trace[:, :, c] - want) / np.maximum(1.0, np.abs(want))) < RTOL
    part = mcmc_amd.tmvn_gibbs_batch(72, n_sweeps, mu, sigma, lower, upper, init, seed=21, chain_offset=9 + 128)["trace"]
    assert cuda.equal(out["trace"][:, :, 128:], part)
    thin = mcmc_amd.tmvn_gibbs_batch(n_chains, n_sweeps, mu, sigma, lower, upper, init, seed=21, chain_offset=9, thin=5)["trace"]
    assert thin.shape[0] == 2 and cuda.equal(thin, out["trace"][4::5][:2])
    # C5 shape in small: lower = +4 conditional sd, upper = +Inf, tail-safe draw, last state + moments only
    d = 128
    sigma = 0.3 ** np.abs(np.arange(d)[:, None] - np.arange(d)[None, :])
    _, sd_c = mcmc_amd.tmvn_precompute(sigma)
    lower, upper = 4.0 * sd_c, np.full(d, np.inf)
    o = mcmc_amd.tmvn_gibbs_batch(130, 30, np.zeros(d), sigma, lower, upper, lower + 0.1, seed=5, tail_safe=True, trace=False,
                                  moments=True, burn=10)
    assert np.all(o["status"].cpu().numpy() == 0)
    fin = o["final_state"].cpu().numpy()
    assert np.all(np.isfinite(fin)) and np.all(fin >= lower[:, None])
    m = o["mean"].cpu().numpy()
    assert np.all(m >= lower[:, None]) and np.all(m < lower[:, None] + 1.0)  # mass sits just above the bound
    # the same run with the trace kept gives the same last state and the same running moments
    o2 = mcmc_amd.tmvn_gibbs_batch(130, 30, np.zeros(d), sigma, lower, upper, lower + 0.1, seed=5, tail_safe=True, moments=True,
                                   burn=10)
    assert cuda.equal(o2["final_state"], o["final_state"]) and cuda.equal(o2["mean"], o["mean"])
    tr = o2["trace"][10:].cpu().numpy()
    np.testing.assert_allclose(m, tr.mean(axis=0), rtol=1e-12)
    # M2 in the tail case: the draws sit ~4.2 sd from mu with a spread of ~0.2 sd; the sums run around the bound
    np.testing.assert_allclose(o["m2"].cpu().numpy(), ((tr - tr.mean(axis=0)) ** 2).sum(axis=0), rtol=1e-10)
