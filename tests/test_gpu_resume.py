"""GPU: checkpoint / resume of chain state (no counterpart in the reference,
whose loop state is lost on return, adaptMCMC_source.cpp:84-105,111-114).

Contract (include/mcmc_amd.h): a run split into launches through d_state_out / d_state_in + iter_offset
produces bit-identical trace rows, rates, moments and final state to one uninterrupted launch.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _gauss(d, rho):
    idx = np.arange(d)
    Sig = rho ** np.abs(idx[:, None] - idx[None, :])
    H = np.linalg.inv(Sig)
    return Sig, 0.5 * (H + H.T)


def _same(torch, a, b, keys):
    for k in keys:
        assert torch.equal(a[k], b[k]) or (torch.isnan(a[k]) == torch.isnan(b[k])).all() and torch.equal(
            torch.nan_to_num(a[k]), torch.nan_to_num(b[k])), k


KEYS = ("final_theta", "final_mean", "final_cov", "acceptance_rate", "status", "shape_iterations")


@pytest.mark.parametrize("kernel,d,ass,ashs,w,win,thin", [
    ("pair", 6, 20, 60, 0.0, 0, 1), ("pair", 10, 30, 90, 0.05, 0, 7), ("pair", 3, 10, 40, 0.0, 20, 3), ("pair", 4, 0, 0, 0.0, 0, 1),
    ("lane", 5, 20, 60, 0.0, 0, 1), ("lane", 12, 15, 50, 0.0, 30, 4),
    ("block", 24, 20, 70, 0.0, 0, 5), ("block", 40, 10, 0, 0.05, 0, 1)])
def test_split_run_equals_uninterrupted_run(cuda, tmp_path, kernel, d, ass, ashs, w, win, thin):
    import mcmc_amd

    nc, n1, n2, n3 = 150, 137, 200, 63
    Sig, H = _gauss(d, 0.7)
    tgt = mcmc_amd.Target.gaussian(0.3 * np.ones(d), H)
    init = mcmc_amd.init_normal(3, nc, np.zeros(d), 1.0, chain_offset=9)
    kw = dict(seed=17, chain_offset=9, kernel=kernel, thin=thin)
    args = (ass, w, win, ashs)
    whole = mcmc_amd.adapt_rwm_batch(tgt, nc, init, 0.5 * np.eye(d), n1 + n2 + n3, *args, **kw)
    a = mcmc_amd.adapt_rwm_batch(tgt, nc, init, 0.5 * np.eye(d), n1, *args, checkpoint=True, **kw)
    assert a["state"].iterations == n1
    # through a file, as a restarted process would
    path = os.path.join(tmp_path, "ckpt.npz")
    a["state"].save(path)
    st = mcmc_amd.RwmState.load(path)
    b = mcmc_amd.adapt_rwm_batch(tgt, nc, init, 0.5 * np.eye(d), n2, *args, resume=st, checkpoint=True, **kw)
    c = mcmc_amd.adapt_rwm_batch(tgt, nc, init, 0.5 * np.eye(d), n3, *args, resume=b["state"], **kw)
    cuda.cuda.synchronize()
    assert int((whole["status"] != 0).sum()) == 0
    parts = cuda.cat([a["theta_trace"], b["theta_trace"], c["theta_trace"]], dim=0)
    assert parts.shape == whole["theta_trace"].shape
    assert cuda.equal(parts, whole["theta_trace"])
    _same(cuda, c, whole, KEYS)
    if ashs:
        assert int(whole["shape_iterations"].max()) > 0  # the split crossed the shape phase


def test_resume_with_injected_streams_diagnostics_and_user_target(cuda, oracle):
    import mcmc_amd

    d, nc, n1, n2 = 4, 70, 90, 110
    src = """
__device__ double mcmc_user_target(const double (&x)[MCMC_USER_D], const double *params)
{
    double s = 0.0;
    for (int k = 0; k < MCMC_USER_D; ++k) s += params[k] * x[k] * x[k] + 0.1 * x[k] * x[(k + 1) % MCMC_USER_D];
    return -0.5 * s;
}
"""
    prm = [1.0, 2.0, 0.5, 1.5]
    tgt = mcmc_amd.Target.compile(src, d, prm)
    rng = np.random.default_rng(1)
    Z = cuda.tensor(rng.standard_normal((n1 + n2, d, nc)), device="cuda")
    U = cuda.tensor(rng.random((n1 + n2, nc)), device="cuda")
    args = (np.zeros(d), np.eye(d))
    ad = (15, 0.0, 0, 50)
    whole = mcmc_amd.adapt_rwm_batch(tgt, nc, *args, n1 + n2, *ad, Z=Z, U=U, diagnostics=True, sigma_trace=True)
    a = mcmc_amd.adapt_rwm_batch(tgt, nc, *args, n1, *ad, Z=Z[:n1].contiguous(), U=U[:n1].contiguous(), diagnostics=True,
                                 sigma_trace=True, checkpoint=True)
    b = mcmc_amd.adapt_rwm_batch(tgt, nc, *args, n2, *ad, Z=Z[n1:].contiguous(), U=U[n1:].contiguous(), diagnostics=True,
                                 sigma_trace=True, resume=a["state"])
    for k in ("theta_trace", "sigma_empirical", "accepted", "log_acceptance", "scaling_sd", "rate_trace"):
        got = cuda.cat([a[k], b[k]], dim=0)
        assert cuda.equal(cuda.nan_to_num(got.double()), cuda.nan_to_num(whole[k].double())), k
    _same(cuda, b, whole, KEYS)
    # the paired-wave kernel with the same user target, Philox streams
    kw = dict(seed=2, kernel="pair")
    whole = mcmc_amd.adapt_rwm_batch(tgt, nc, *args, n1 + n2, *ad, **kw)
    a = mcmc_amd.adapt_rwm_batch(tgt, nc, *args, n1, *ad, checkpoint=True, **kw)
    b = mcmc_amd.adapt_rwm_batch(tgt, nc, *args, n2, *ad, resume=a["state"], **kw)
    assert cuda.equal(cuda.cat([a["theta_trace"], b["theta_trace"]], dim=0), whole["theta_trace"])
    _same(cuda, b, whole, KEYS)


def test_resume_refuses_foreign_states_and_keeps_dead_chains_dead(cuda):
    import mcmc_amd

    d, nc = 6, 66
    Sig, H = _gauss(d, 0.6)
    tgt = mcmc_amd.Target.gaussian(np.zeros(d), H)
    run = lambda n, **kw: mcmc_amd.adapt_rwm_batch(tgt, nc, np.zeros(d), np.eye(d), n, 20, 0.0, 0, 60, seed=5, **kw)
    a = run(100, kernel="pair", checkpoint=True)
    # a state written by the paired-wave kernel, resumed by the one-wave kernel: every chain stops, nothing is sampled
    b = run(50, kernel="lane", resume=a["state"])
    s = b["status"].cpu().numpy()
    assert np.all((s & 15) == mcmc_amd.MCMC_CHAIN_BAD_STATE) and np.all((s >> 4) == 101)
    assert bool(cuda.isnan(b["theta_trace"]).all())
    # the right kernel at the wrong iteration
    wrong = mcmc_amd.RwmState(a["state"].array, 90)
    s = run(50, kernel="pair", resume=wrong)["status"].cpu().numpy()
    assert np.all((s & 15) == mcmc_amd.MCMC_CHAIN_BAD_STATE)
    with pytest.raises(ValueError):
        mcmc_amd.adapt_rwm_batch(tgt, nc + 1, np.zeros(d), np.eye(d), 10, resume=a["state"])
    # a chain that died of a Cholesky failure (shape phase from iteration 1: rank-deficient covmat_empirical, :153) stays dead
    for kernel in ("pair", "lane"):
        kw = dict(seed=5, kernel=kernel)
        whole = mcmc_amd.adapt_rwm_batch(tgt, nc, np.zeros(d), np.eye(d), 100, 0, 0.0, 0, 1, **kw)
        x = mcmc_amd.adapt_rwm_batch(tgt, nc, np.zeros(d), np.eye(d), 40, 0, 0.0, 0, 1, checkpoint=True, **kw)
        y = mcmc_amd.adapt_rwm_batch(tgt, nc, np.zeros(d), np.eye(d), 60, 0, 0.0, 0, 1, resume=x["state"], **kw)
        assert np.all((whole["status"].cpu().numpy() & 15) == mcmc_amd.MCMC_CHAIN_CHOL_FAILED)
        np.testing.assert_array_equal(y["status"].cpu().numpy(), whole["status"].cpu().numpy())
        got = cuda.cat([x["theta_trace"], y["theta_trace"]], dim=0)
        assert cuda.equal(cuda.isnan(got), cuda.isnan(whole["theta_trace"]))
        assert cuda.equal(cuda.nan_to_num(got), cuda.nan_to_num(whole["theta_trace"]))


@pytest.mark.parametrize("d,thin,tail", [(10, 1, False), (7, 4, False), (40, 3, False), (64, 1, True)])
def test_gibbs_split_run_equals_uninterrupted_run(cuda, d, thin, tail):
    """tmvrnormGibbs.cpp:115-139: the chain state is x alone; sweep_offset continues the Philox counter."""
    import mcmc_amd

    nc, n1, n2 = 130, 37, 50
    Sig, _ = _gauss(d, 0.5)
    mu = np.linspace(-0.5, 0.5, d)
    lower = (mu + 2.0) if tail else -np.ones(d)
    upper = np.full(d, np.inf) if tail else 2 * np.ones(d)
    x0 = lower + 0.1
    args = (mu, Sig, lower, upper)
    kw = dict(seed=8, thin=thin, chain_offset=3, tail_safe=tail)
    whole = mcmc_amd.tmvn_gibbs_batch(nc, n1 + n2, *args, x0, **kw)
    a = mcmc_amd.tmvn_gibbs_batch(nc, n1, *args, x0, **kw)
    b = mcmc_amd.tmvn_gibbs_batch(nc, n2, *args, a["final_state"], sweep_offset=n1, **kw)
    assert int((whole["status"] != 0).sum()) == 0
    assert cuda.equal(cuda.cat([a["trace"], b["trace"]], dim=0), whole["trace"])
    assert cuda.equal(b["final_state"], whole["final_state"])


def test_auto_resume_returns_to_the_kernel_that_wrote_the_state(cuda):
    """AUTO picks the paired-wave kernel for a plain Philox run and the one-wave kernel when diagnostics are asked for;
    a checkpoint is in its writer's representation.  A resumed AUTO run therefore goes back to the writer -- the lane
    kernel's state continues under AUTO without diagnostics -- and a pair-kernel state resumed WITH diagnostics is
    refused on the host instead of stopping every chain with MCMC_CHAIN_BAD_STATE."""
    import mcmc_amd

    d, nc = 4, 100
    tgt = mcmc_amd.Target.gaussian(np.zeros(d), np.eye(d))
    ad = (10, 0.0, 0, 40)
    kw = dict(seed=3)
    whole = mcmc_amd.adapt_rwm_batch(tgt, nc, np.ones(d), np.eye(d), 300, *ad, kernel="lane", **kw)
    a = mcmc_amd.adapt_rwm_batch(tgt, nc, np.ones(d), np.eye(d), 120, *ad, checkpoint=True, diagnostics=True, **kw)   # AUTO -> lane
    b = mcmc_amd.adapt_rwm_batch(tgt, nc, np.ones(d), np.eye(d), 180, *ad, resume=a["state"], **kw)                   # AUTO, no diagnostics
    assert int((b["status"] != 0).sum()) == 0
    assert cuda.equal(cuda.cat([a["theta_trace"], b["theta_trace"]]), whole["theta_trace"])
    p = mcmc_amd.adapt_rwm_batch(tgt, nc, np.ones(d), np.eye(d), 120, *ad, checkpoint=True, **kw)                      # AUTO -> pair
    with pytest.raises(mcmc_amd.McmcError) as e:
        mcmc_amd.adapt_rwm_batch(tgt, nc, np.ones(d), np.eye(d), 180, *ad, resume=p["state"], diagnostics=True, **kw)
    assert e.value.code == -2 and "paired-wave" in str(e.value)
