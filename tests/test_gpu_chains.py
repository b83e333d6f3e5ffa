"""GPU suite: the host-pointer multi-chain entry points an R caller reaches the batch kernels through
(mcmc_adaptMCMC_chains, mcmc_rtmvnorm_gibbs_chains; include/mcmc_amd.h).  They do the job of the reference's
mclapply_RNG(X, FUN) (MCMCutilty.R:103-149) -- one sampler call per element of X -- in one launch, and return the list of
chains melt_trace (MCMCutilty.R:19-35) reads.

Bar: bit-identical to the device-pointer batch entry points on the same chains (same kernels, same Philox streams), and
within 1e-12 of the oracle's adaptMCMC / rtmvnorm_gibbs fed the replayed streams."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL = 1e-12


def _rel(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))))


def _gauss(d, rho):
    idx = np.arange(d)
    H = np.linalg.inv(rho ** np.abs(idx[:, None] - idx[None, :]))
    return 0.5 * (H + H.T)


def test_adaptMCMC_chains_equals_batch_entry_and_oracle(cuda, oracle):
    """200 chains through the host-pointer entry equal the same chains through mcmc_adapt_rwm_batch
    bit for bit, and chain 17 equals the oracle replay to 1e-12."""
    import mcmc_amd

    d, nc, n = 10, 200, 400
    H = _gauss(d, 0.9)
    tgt = mcmc_amd.Target.gaussian(np.zeros(d), H)
    rng = np.random.default_rng(8)
    X = [2.0 * rng.standard_normal(d) for _ in range(nc)]  # the list mclapply_RNG takes: one init_theta per chain
    chains = mcmc_amd.adaptMCMC_chains(tgt, X, np.eye(d), n, 30, 0.0, 0, 60, 100, 7.0, covmat_empirical=True)
    assert len(chains) == nc and all(c["status"] == 0 for c in chains)
    assert chains[0]["trace"].shape == (n, d)

    init = cuda.tensor(np.ascontiguousarray(np.asarray(X).T), device="cuda")
    out = mcmc_amd.adapt_rwm_batch(tgt, nc, init, np.eye(d), n, 30, 0.0, 0, 60, seed=7)
    tr, acc, cov = out["theta_trace"].cpu().numpy(), out["acceptance_rate"].cpu().numpy(), out["final_cov"].cpu().numpy()
    for c in range(nc):
        assert np.array_equal(chains[c]["trace"], tr[:, :, c]), c
        assert chains[c]["acceptance_rate"] == acc[c]
        assert np.array_equal(chains[c]["covmat_empirical"], cov[:, c].reshape(d, d, order="F"))

    Z, U = mcmc_amd.rwm_streams(7, nc, n, d)
    for c in (17, 199):
        ref = oracle.adaptMCMC(oracle.gaussian_target(np.zeros(d), H), X[c], np.eye(d), n, 30, 0.0, 0, 60,
                               Z.cpu().numpy()[:, :, c], U.cpu().numpy()[:, c], store_sigma=False)
        assert _rel(chains[c]["trace"], ref["theta_trace"]) < RTOL
        assert chains[c]["acceptance_rate"] == ref["acceptance_rate"]

    df = mcmc_amd.melt_trace(chains[:5])  # the shape the reference's helper consumes
    assert df.shape == (5 * n, d + 2) and list(df.columns[-2:]) == ["chain", "iteration"]


def test_adaptMCMC_chains_thinning_shared_start_and_wide_d(cuda, oracle):
    """thin > 1, one start for all chains (init_per_chain = 0 through the raw C ABI), a chain count that is no multiple
    of 64, and d = 40 (row kernel under AUTO): equal to the batch entry, chain 3 equal to the oracle."""
    import mcmc_amd
    from mcmc_amd import _lib as L

    d, nc, n, thin = 40, 70, 480, 4
    H = _gauss(d, 0.5)
    tgt = mcmc_amd.Target.gaussian(np.zeros(d), H)
    cov0 = 0.05 * np.eye(d)
    start = np.full(d, 0.25)
    ct = tgt._c()
    n_kept = n // thin
    trace = np.empty(nc * n_kept * d)
    acc = np.empty(nc)
    status = np.full(nc, -1, dtype=np.int32)
    rc = L.lib.mcmc_adaptMCMC_chains(C.byref(ct), d, nc, start.ctypes.data_as(L.dp), 0,
                                     np.ascontiguousarray(cov0.ravel(order="F")).ctypes.data_as(L.dp), n, 0, 0.0, 0, 120, 50, 11.0,
                                     0.99, 50.0, thin, trace.ctypes.data_as(L.dp), acc.ctypes.data_as(L.dp), None,
                                     status.ctypes.data_as(L.ip))
    assert rc == 0, L.lib.mcmc_amd_last_error()
    assert np.all(status == 0)
    out = mcmc_amd.adapt_rwm_batch(tgt, nc, start, cov0, n, 0, 0.0, 0, 120, seed=11, thin=thin)
    assert int(out["shape_iterations"].min().item()) > 50  # the covariance adaptation was reached
    tr = out["theta_trace"].cpu().numpy()
    for c in range(nc):
        assert np.array_equal(trace[c * n_kept * d:(c + 1) * n_kept * d].reshape(n_kept, d, order="F"), tr[:, :, c])
    assert np.array_equal(acc, out["acceptance_rate"].cpu().numpy())
    Z, U = mcmc_amd.rwm_streams(11, nc, n, d)
    ref = oracle.adaptMCMC(oracle.gaussian_target(np.zeros(d), H), start, cov0, n, 0, 0.0, 0, 120, Z.cpu().numpy()[:, :, 3],
                           U.cpu().numpy()[:, 3], store_sigma=False)
    got = trace[3 * n_kept * d:4 * n_kept * d].reshape(n_kept, d, order="F")
    assert _rel(got, ref["theta_trace"][thin - 1::thin]) < RTOL


def test_adaptMCMC_chains_reports_failed_chains_and_bad_arguments(cuda):
    import mcmc_amd

    d = 3
    tgt = mcmc_amd.Target.gaussian(np.zeros(d), np.eye(d))
    X = [np.zeros(d)] * 5
    # adapt_shape_start 1 < d: covmat_empirical is rank deficient when the shape phase starts -> arma::chol would throw
    # (adaptMCMC_source.cpp:153); every chain carries its own status word, the call itself succeeds
    chains = mcmc_amd.adaptMCMC_chains(tgt, X, np.eye(d), 50, 0, 0.0, 0, 1, 10, 5.0)
    assert all((c["status"] & 15) == mcmc_amd.MCMC_CHAIN_CHOL_FAILED for c in chains)
    assert all(np.isnan(c["trace"][-1]).all() for c in chains)
    with pytest.raises(mcmc_amd.McmcError):
        mcmc_amd.adaptMCMC_chains(tgt, X, np.eye(d), 50, 0, 0.0, 0, 0, 0, 5.0)      # info = 0 (UB in the reference, :142)
    with pytest.raises(mcmc_amd.McmcError):
        mcmc_amd.adaptMCMC_chains(tgt, X, -np.eye(d), 50, 0, 0.0, 0, 0, 10, 5.0)    # covmat not PD
    with pytest.raises(ValueError):
        mcmc_amd.adaptMCMC_chains(tgt, [np.zeros(3), np.zeros(4)], np.eye(d), 50, 0, 0.0, 0, 0, 10, 5.0)


def test_rtmvnorm_gibbs_chains_equals_batch_entry_and_oracle(cuda, oracle):
    import mcmc_amd

    # the reference's own 4-D example (tmvrnormGibbs.cpp:146), one start per chain inside the support
    mu, sigma = np.arange(1.0, 5.0), np.diag(np.arange(1.0, 5.0))
    lower, upper = np.array([-np.inf, -np.inf, 0.0, 0.0]), np.array([10.0, 10.0, 100.0, 100.0])
    rng = np.random.default_rng(2)
    nc, n, d = 150, 300, 4
    X = [np.array([2.0, 2.0, 50.0, 50.0]) + rng.uniform(-1, 1, d) for _ in range(nc)]
    chains = mcmc_amd.rtmvnorm_gibbs_chains(n, mu, sigma, lower, upper, X, seed=9)
    assert len(chains) == nc and all(c["status"] == 0 for c in chains)
    init = cuda.tensor(np.ascontiguousarray(np.asarray(X).T), device="cuda")
    out = mcmc_amd.tmvn_gibbs_batch(nc, n, mu, sigma, lower, upper, init, seed=9)
    tr = out["trace"].cpu().numpy()
    for c in range(nc):
        assert np.array_equal(chains[c]["trace"], tr[:, :, c]), c
    U = mcmc_amd.gibbs_streams(9, nc, n, d).cpu().numpy()
    for c in (0, 149):
        want = oracle.rtmvnorm_gibbs(n, mu, sigma, lower, upper, X[c], U[:, :, c].reshape(-1))
        assert _rel(chains[c]["trace"], want) < RTOL
    # thinning and the wide kernel (d = 40)
    d = 40
    idx = np.arange(d)
    sigma = 0.5 ** np.abs(idx[:, None] - idx[None, :])
    X = [np.zeros(d)] * 66
    chains = mcmc_amd.rtmvnorm_gibbs_chains(60, np.zeros(d), sigma, np.full(d, -1.0), np.full(d, 2.0), X, seed=3, thin=3)
    out = mcmc_amd.tmvn_gibbs_batch(66, 60, np.zeros(d), sigma, np.full(d, -1.0), np.full(d, 2.0), np.zeros(d), seed=3, thin=3)
    tr = out["trace"].cpu().numpy()
    assert chains[0]["trace"].shape == (20, d)
    for c in range(66):
        assert np.array_equal(chains[c]["trace"], tr[:, :, c])


def test_adaptMCMC_chains_large_trace_takes_the_pipelined_download(cuda):
    """A trace of 289 MB (above four 64 MB chunks, and no multiple of a chunk): the result reaches the caller's pageable
    buffer through the two pinned chunks and the host copy threads (api.hip: download_pipelined) and must equal the batch
    entry's device trace bit for bit, to the last element."""
    import mcmc_amd
    from mcmc_amd import _lib as L

    d, nc, n = 4, 8200, 1100
    tgt = mcmc_amd.Target.gaussian(np.zeros(d), _gauss(d, 0.5))
    start = np.full(d, 0.5)
    ct = tgt._c()
    trace = np.full(nc * n * d, np.nan)
    acc = np.empty(nc)
    status = np.full(nc, -1, dtype=np.int32)
    rc = L.lib.mcmc_adaptMCMC_chains(C.byref(ct), d, nc, start.ctypes.data_as(L.dp), 0,
                                     np.ascontiguousarray(np.eye(d).ravel(order="F")).ctypes.data_as(L.dp), n, 20, 0.0, 0, 0, 50, 13.0,
                                     0.99, 50.0, 1, trace.ctypes.data_as(L.dp), acc.ctypes.data_as(L.dp), None,
                                     status.ctypes.data_as(L.ip))
    assert rc == 0, L.lib.mcmc_amd_last_error()
    assert np.all(status == 0)
    out = mcmc_amd.adapt_rwm_batch(tgt, nc, start, np.eye(d), n, 20, 0.0, 0, 0, seed=13)
    want = out["theta_trace"].permute(2, 1, 0).contiguous().cpu().numpy()  # [chain][k][t]: chain c's n x d matrix, column-major
    assert np.array_equal(trace.reshape(nc, d, n), want)
    assert np.array_equal(acc, out["acceptance_rate"].cpu().numpy())
