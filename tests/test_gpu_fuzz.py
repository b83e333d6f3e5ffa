"""GPU parity on random shapes and settings (seeded): the Gibbs kernels against the oracle with injected uniforms, a
Philox Gibbs run split by sweep_offset against the uninterrupted one, and the Metropolis kernel AUTO picks against a
second kernel on the same Philox run plus a split run (tools/fuzz_rwm.py is the long-running form of the latter)."""
import numpy as np
import pytest

from test_gpu_gibbs import _cond_tol

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(10))
def test_gibbs_random_case_matches_oracle(cuda, oracle, seed):
    import mcmc_amd

    rng = np.random.default_rng(7000 + seed)
    d = int(rng.choice([1, 2, 7, 19, 27, 28, 40, 64, 65, 90, 128, 129, 170]))
    n_chains = int(rng.integers(1, 140))
    n_sweeps = int(rng.integers(1, 9))
    A = rng.standard_normal((d, d))
    sigma = A @ A.T / d + np.eye(d)
    sdm = np.sqrt(np.diag(sigma))
    mu = 0.3 * rng.standard_normal(d)
    lower = mu - sdm * rng.uniform(0.3, 2.0, d)
    upper = mu + sdm * rng.uniform(0.3, 2.0, d)
    lower[rng.random(d) < 0.25] = -np.inf  # one-sided and untruncated coordinates (tmvrnormGibbs.cpp:146 has both)
    upper[rng.random(d) < 0.25] = np.inf
    tail_safe = bool(rng.integers(0, 2))
    init = np.clip(mu[:, None] + 0.2 * rng.standard_normal((d, n_chains)), np.where(np.isfinite(lower), lower, -1e9)[:, None],
                   np.where(np.isfinite(upper), upper, 1e9)[:, None])
    U = rng.random((n_sweeps, d, n_chains))
    out = mcmc_amd.tmvn_gibbs_batch(n_chains, n_sweeps, mu, sigma, lower, upper, cuda.tensor(init, device="cuda"),
                                    U=cuda.tensor(U, device="cuda"), moments=True, tail_safe=tail_safe)
    cuda.cuda.synchronize()
    assert np.all(out["status"].cpu().numpy() == 0)
    trace = out["trace"].cpu().numpy()
    _, P, sd = oracle.tmvn_precompute(sigma)
    for c in sorted({0, n_chains // 2, n_chains - 1}):
        want = oracle.rtmvnorm_gibbs(n_sweeps, mu, sigma, lower, upper, init[:, c], U[:, :, c].reshape(-1))
        tol = _cond_tol(want, init[:, c], mu, P, sd)
        assert np.all(np.abs(trace[:, :, c] - want) <= tol), (d, c, np.max(np.abs(trace[:, :, c] - want) / tol))
        np.testing.assert_allclose(out["mean"].cpu().numpy()[:, c], trace[:, :, c].mean(axis=0), rtol=1e-11, atol=1e-13)
    assert np.all(trace >= lower[None, :, None]) and np.all(trace <= upper[None, :, None])


@pytest.mark.parametrize("seed", range(6))
def test_gibbs_philox_run_split_by_sweep_offset(cuda, seed):
    import mcmc_amd

    rng = np.random.default_rng(7100 + seed)
    d = int(rng.choice([3, 27, 28, 64, 100, 130]))
    n_chains, n, cut = int(rng.integers(1, 200)), int(rng.integers(2, 12)), 0
    cut = int(rng.integers(1, n))
    idx = np.arange(d)
    sigma = 0.4 ** np.abs(idx[:, None] - idx[None, :])
    lower, upper = np.full(d, -0.8), np.where(idx % 2 == 0, np.inf, 1.5)
    kw = dict(seed=int(rng.integers(1, 99)), chain_offset=int(rng.integers(0, 500)), moments=False)
    whole = mcmc_amd.tmvn_gibbs_batch(n_chains, n, np.zeros(d), sigma, lower, upper, np.zeros(d), **kw)
    a = mcmc_amd.tmvn_gibbs_batch(n_chains, cut, np.zeros(d), sigma, lower, upper, np.zeros(d), **kw)
    b = mcmc_amd.tmvn_gibbs_batch(n_chains, n - cut, np.zeros(d), sigma, lower, upper, a["final_state"], sweep_offset=cut, **kw)
    cuda.cuda.synchronize()
    assert cuda.equal(cuda.cat([a["trace"], b["trace"]]), whole["trace"])
    assert cuda.equal(b["final_state"], whole["final_state"])


@pytest.mark.parametrize("seed", range(16))
def test_metropolis_auto_kernel_against_a_second_kernel_and_split_run(cuda, seed):
    import mcmc_amd

    rng = np.random.default_rng(7200 + seed)
    d = int(rng.choice([2, 10, 11, 13, 16, 17, 32, 33, 48, 64, 65, 97, 112, 113, 128]))
    n = int(rng.integers(1, 100)) if rng.random() < 0.5 else int(rng.integers(2 * d, 4 * d + 40))
    nc, thin = int(rng.integers(1, 45)), int(rng.integers(1, 6))
    mode = int(rng.integers(0, 3))
    w, win = (0.0, 0) if mode == 0 else ((float(rng.uniform(0.01, 0.3)), 0) if mode == 1 else (0.0, int(rng.choice([3, 64, 65, 130]))))
    ass = int(rng.integers(0, 30))
    # adapt_shape_start > d with the running-mean rate only: otherwise covmat_empirical is rank deficient and WHEN its
    # factorization fails is a matter of rounding (the kernels then disagree about the iteration, as LAPACK builds would)
    ashs = int(rng.choice([0, d + 8, 2 * d])) if mode == 0 else 0
    A = rng.standard_normal((d, d))
    H = np.linalg.inv(A @ A.T / d + np.eye(d))
    H = 0.5 * (H + H.T)
    m = 0.2 * rng.standard_normal(d)
    args = (mcmc_amd.Target.gaussian(m, H), nc, m, (0.5 / d) * np.eye(d))
    ad = (ass, w, win, ashs)
    kw = dict(seed=int(rng.integers(1, 1000)), chain_offset=int(rng.integers(0, 1000)), thin=thin)
    a = mcmc_amd.adapt_rwm_batch(*args, n, *ad, **kw)
    b = mcmc_amd.adapt_rwm_batch(*args, n, *ad, kernel="lane" if d <= 12 else "block", **kw)
    cuda.cuda.synchronize()
    assert np.all(a["status"].cpu().numpy() == 0) and np.all(b["status"].cpu().numpy() == 0)
    ta, tb = a["theta_trace"].cpu().numpy(), b["theta_trace"].cpu().numpy()
    if ta.size:
        assert np.max(np.abs(ta - tb) / np.maximum(1.0, np.abs(tb))) < 1e-11, (d, n)
    assert cuda.equal(a["acceptance_rate"], b["acceptance_rate"])
    if n > 1:
        cut = int(rng.integers(1, n))
        p1 = mcmc_amd.adapt_rwm_batch(*args, cut, *ad, checkpoint=True, **kw)
        p2 = mcmc_amd.adapt_rwm_batch(*args, n - cut, *ad, resume=p1["state"], **kw)
        cuda.cuda.synchronize()
        assert cuda.equal(cuda.cat([p1["theta_trace"], p2["theta_trace"]]), a["theta_trace"])
        assert cuda.equal(p2["acceptance_rate"], a["acceptance_rate"])
