"""GPU suite: the two non-default options of mcmc_rwm_config that exist for the d > 12 kernels.

* trace_layout = MCMC_TRACE_CHAIN_MAJOR: the same samples as the default layout, bit for bit, stored [chain][t][k].
* factor_update = MCMC_FACTOR_RANK1 (NOT a parity mode): the proposal factor follows covmat_empirical by closed-form
  rank-1 updates (tests/test_rank1_update_math.py has the algebra) with a full factorization every refresh_every
  iterations, instead of the reference's fresh chol every iteration (adaptMCMC_source.cpp:153).  Same algorithm in exact
  arithmetic, so the bar is: bit-identical to the exact mode when every iteration refreshes; otherwise equal decisions
  and trajectories within rounding drift (measured: DESIGN.md section 3.3c), results independent of sharding, and a run
  split through a checkpoint bit-identical to the uninterrupted run."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _target(d, seed=4):
    import mcmc_amd

    A = np.random.default_rng(seed).standard_normal((d, d))
    H = np.linalg.inv(A @ A.T / d + np.eye(d))
    return mcmc_amd.Target.gaussian(np.zeros(d), 0.5 * (H + H.T))


@pytest.mark.parametrize("d,kernel,nc,thin", [(20, "auto", 70, 1), (40, "auto", 33, 3), (64, "row", 9, 2), (20, "block", 5, 1),
                                              (70, "auto", 17, 4), (128, "auto", 3, 5)])
def test_chain_major_trace_holds_the_same_samples(cuda, d, kernel, nc, thin):
    import mcmc_amd

    tgt = _target(d)
    n = 200 + 4 * d  # adapt_shape_start = 2 d accepted jumps at an acceptance rate around 0.6
    kw = dict(seed=3, thin=thin, kernel=kernel)
    a = mcmc_amd.adapt_rwm_batch(tgt, nc, np.zeros(d), _cov0(d), n, 0, 0.0, 0, 2 * d, **kw)
    b = mcmc_amd.adapt_rwm_batch(tgt, nc, np.zeros(d), _cov0(d), n, 0, 0.0, 0, 2 * d, trace_layout="chain_major", **kw)
    assert tuple(b["theta_trace"].shape) == (nc, n // thin, d)
    assert int(a["shape_iterations"].min().item()) > 10 and int((a["status"] != 0).sum().item()) == 0
    assert cuda.equal(b["theta_trace"].permute(1, 2, 0).contiguous(), a["theta_trace"])
    for k in ("acceptance_rate", "final_theta", "final_cov", "status"):
        assert cuda.equal(a[k], b[k]), k


def test_chain_major_is_refused_where_no_kernel_serves_it(cuda):
    import mcmc_amd

    tgt = mcmc_amd.Target.gaussian(np.zeros(4), np.eye(4))
    with pytest.raises(mcmc_amd.McmcError) as e:
        mcmc_amd.adapt_rwm_batch(tgt, 8, np.zeros(4), np.eye(4), 10, trace_layout="chain_major")
    assert e.value.code == -2


def _cov0(d):
    """start covariance: small steps (acceptance ~0.5), but det(covmat) must stay representable -- (0.3 / 128)^128
    underflows, and then the reference rejects everything through the NaN of mvrnorm_pdf (adaptMCMC_source.cpp:56,169-175)"""
    return max(0.3 / d, 0.01) * np.eye(d)


def _first_difference(x, y, tol):
    """per chain: first kept row where the two [n, d, c] traces differ by more than tol (relative), n if never"""
    err = np.max(np.abs(x - y) / np.maximum(1.0, np.abs(y)), axis=1)  # [n, c]
    bad = err > tol
    return np.where(bad.any(axis=0), bad.argmax(axis=0), x.shape[0]), err


@pytest.mark.parametrize("d,kernel", [(20, "block"), (72, "auto"), (100, "auto"), (128, "auto")])
def test_rank1_mode_follows_the_exact_mode(cuda, d, kernel):
    import mcmc_amd

    tgt = _target(d)
    nc, n = 48, 500 + 4 * d
    args = (tgt, nc, np.zeros(d), _cov0(d), n, 0, 0.0, 0, 2 * d)
    exact = mcmc_amd.adapt_rwm_batch(*args, seed=9, kernel=kernel)
    te = exact["theta_trace"].cpu().numpy()
    assert int(exact["shape_iterations"].min().item()) > n // 2 and int((exact["status"] != 0).sum().item()) == 0
    # every iteration a refresh point: the same factorizations as the exact mode, so the same bits
    r1 = mcmc_amd.adapt_rwm_batch(*args, seed=9, kernel=kernel, factor_update="rank1", refresh_every=1)
    assert cuda.equal(r1["theta_trace"], exact["theta_trace"])
    assert cuda.equal(r1["final_cov"], exact["final_cov"]) and cuda.equal(r1["acceptance_rate"], exact["acceptance_rate"])
    for refresh in (16, 64, 1000):  # 1000: one factorization when the shape phase starts, then updates only
        fast = mcmc_amd.adapt_rwm_batch(*args, seed=9, kernel=kernel, factor_update="rank1", refresh_every=refresh)
        assert int((fast["status"] != 0).sum().item()) == 0
        assert cuda.equal(fast["shape_iterations"], exact["shape_iterations"]) or refresh == 1000
        first, err = _first_difference(fast["theta_trace"].cpu().numpy(), te, 1e-9)
        # a decision flips only when |log A - la| is inside the drift: at most a stray chain in 48 x 700 iterations
        assert (first == n).sum() >= nc - 1, (refresh, first)
        same = first == n
        assert err[:, same].max() < 1e-10, (refresh, err[:, same].max())  # measured: ~1e-13


def test_rank1_mode_sharding_resume_and_phase_changes(cuda):
    import mcmc_amd

    d, nc, n = 40, 20, 600
    tgt = _target(d)
    # weighted acceptance rate + size adaptation: the sampler moves between the size and the shape phase several times
    args = (tgt, nc, np.zeros(d), (0.5 / d) * np.eye(d), n, 30, 0.1, 0, 80)
    kw = dict(seed=5, kernel="block", factor_update="rank1", refresh_every=32, thin=2)
    whole = mcmc_amd.adapt_rwm_batch(*args, **kw)
    sh = whole["shape_iterations"].cpu().numpy()
    assert sh.min() > 100 and sh.max() < n - 100 and int((whole["status"] != 0).sum().item()) == 0  # both phases, several switches (checked with the oracle)
    exact = mcmc_amd.adapt_rwm_batch(*args, seed=5, kernel="block", thin=2)
    first, err = _first_difference(whole["theta_trace"].cpu().numpy(), exact["theta_trace"].cpu().numpy(), 1e-9)
    assert (first == n // 2).sum() >= nc - 1
    # sharding: chains 7.. alone
    part = mcmc_amd.adapt_rwm_batch(tgt, nc - 7, np.zeros(d), (0.5 / d) * np.eye(d), n, 30, 0.1, 0, 80, chain_offset=7, **kw)
    assert cuda.equal(part["theta_trace"], whole["theta_trace"][:, :, 7:].contiguous())
    # a run cut in three through checkpoints continues bit for bit (refresh points are absolute iteration numbers)
    pieces, state, rows = [], None, []
    for seg in (151, 250, 199):
        out = mcmc_amd.adapt_rwm_batch(tgt, nc, np.zeros(d), (0.5 / d) * np.eye(d), seg, 30, 0.1, 0, 80, checkpoint=True, resume=state, **kw)
        state = out["state"]
        rows.append(out["theta_trace"])
        pieces.append(out)
    assert cuda.equal(cuda.cat(rows, dim=0), whole["theta_trace"])
    for k in ("acceptance_rate", "final_theta", "final_mean", "final_cov", "status", "shape_iterations"):
        assert cuda.equal(pieces[-1][k], whole[k]), k


def test_rank1_mode_failure_and_refusals(cuda):
    import mcmc_amd

    d = 30
    tgt = _target(d)
    # adapt_shape_start far below d: covmat_empirical is rank deficient at the first factorization (arma::chol would throw, :153)
    out = mcmc_amd.adapt_rwm_batch(tgt, 6, np.zeros(d), (0.3 / d) * np.eye(d), 80, 0, 0.0, 0, 3, seed=1, kernel="block",
                                   factor_update="rank1", refresh_every=8)
    ex = mcmc_amd.adapt_rwm_batch(tgt, 6, np.zeros(d), (0.3 / d) * np.eye(d), 80, 0, 0.0, 0, 3, seed=1, kernel="block")
    st = out["status"].cpu().numpy()
    assert np.all((st & 15) == mcmc_amd.MCMC_CHAIN_CHOL_FAILED)
    np.testing.assert_array_equal(st, ex["status"].cpu().numpy())
    small = mcmc_amd.Target.gaussian(np.zeros(8), np.eye(8))
    for bad in (dict(target=small, d=8), dict(target=tgt, d=d, diagnostics=True), dict(target=tgt, d=d, kernel="row"),
                dict(target=tgt, d=d, refresh_every=0)):
        t, dd = bad.pop("target"), bad.pop("d")
        kw = dict(factor_update="rank1", refresh_every=8)
        kw.update(bad)
        with pytest.raises(mcmc_amd.McmcError):
            mcmc_amd.adapt_rwm_batch(t, 4, np.zeros(dd), np.eye(dd), 20, 0, 0.0, 0, 10, **kw)
