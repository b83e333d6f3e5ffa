"""GPU parity: user targets compiled at run time (mcmc_target_compile).

The reference calls an R closure `target` twice per iteration (adaptMCMC_source.cpp:97,158); the
oracle calls a Python closure through the same up-call.  Here the same function is given to the
device as HIP source and to the oracle as a closure; with identical injected streams decisions
must match bit-exactly and trajectories within 1e-12 relative (same bar as test_gpu_rwm.py).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RTOL = 1e-12

# log-density of a d-dimensional "Rosenbrock ridge" with a Gaussian envelope; params = a, b, tau
RIDGE_SRC = r"""
__device__ double mcmc_user_target(const double (&x)[MCMC_USER_D], const double *params)
{
    const double a = params[0], b = params[1], tau = params[2];
    double s = 0.0;
#pragma unroll
    for (int k = 0; k + 1 < MCMC_USER_D; ++k) {
        const double u = x[k + 1] - x[k] * x[k], v = a - x[k];
        s += b * (u * u) + v * v;
    }
    double e = 0.0;
#pragma unroll
    for (int k = 0; k < MCMC_USER_D; ++k) e += x[k] * x[k];
    return -s / 20.0 - tau * e;
}
"""

# the banana of adaptMCMC_analysis.R:19-22 restated as a user target; params = B
BANANA_SRC = r"""
__device__ double mcmc_user_target(const double (&x)[MCMC_USER_D], const double *params)
{
    const double B = params[0];
    const double u = x[1] + B * (x[0] * x[0]) - 100 * B;
    return -(x[0] * x[0]) / 200 - 1.0 / 2 * (u * u);
}
"""

# a target with a hard support boundary: -Inf outside the unit box (the :163-165 reject path), log via fm_log
BOX_SRC = r"""
__device__ double mcmc_user_target(const double (&x)[MCMC_USER_D], const double *params)
{
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < MCMC_USER_D; ++k) {
        if (!(x[k] > 0.0 && x[k] < 1.0)) return -__builtin_inf();
        s += params[k] * mcmc::fm_log<true>(x[k]) + params[MCMC_USER_D + k] * mcmc::fm_log<true>(1.0 - x[k]);
    }
    return s;
}
"""


def ridge(prm):
    a, b, tau = prm

    def f(x):
        x = np.asarray(x, dtype=np.float64).reshape(-1)
        s = 0.0
        for k in range(x.size - 1):
            u = x[k + 1] - x[k] * x[k]
            v = a - x[k]
            s += b * (u * u) + v * v
        e = 0.0
        for k in range(x.size):
            e += x[k] * x[k]
        return -s / 20.0 - tau * e

    return f


def box(prm, d):
    def f(x):
        x = np.asarray(x, dtype=np.float64).reshape(-1)
        if not np.all((x > 0) & (x < 1)):
            return -np.inf
        s = 0.0
        for k in range(d):
            s += prm[k] * np.log(x[k]) + prm[d + k] * np.log(1.0 - x[k])
        return s

    return f


def _rel(a, b):
    return np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b)))


@pytest.mark.parametrize("d", [2, 5, 12])
def test_user_target_values_match_the_host_closure(cuda, oracle, d):
    import mcmc_amd

    prm = [1.0, 5.0, 0.05]
    tgt = mcmc_amd.Target.compile(RIDGE_SRC, d, prm)
    rng = np.random.default_rng(d)
    x = rng.standard_normal((d, 1000)) * 1.5
    got = mcmc_amd.target_eval_batch(tgt, cuda.tensor(x, device="cuda")).cpu().numpy()
    f = ridge(prm)
    want = np.array([f(x[:, c]) for c in range(x.shape[1])])
    np.testing.assert_allclose(got, want, rtol=1e-13, atol=1e-13)
    # the built-in kinds go through the same entry point
    H = np.linalg.inv(0.7 ** np.abs(np.subtract.outer(np.arange(d), np.arange(d))))
    H = 0.5 * (H + H.T)
    m = np.linspace(-1, 1, d)
    got = mcmc_amd.target_eval_batch(mcmc_amd.Target.gaussian(m, H), cuda.tensor(x, device="cuda")).cpu().numpy()
    ot = oracle.gaussian_target(m, H)
    np.testing.assert_allclose(got[:200], [ot(x[:, c]) for c in range(200)], rtol=1e-13, atol=1e-13)


@pytest.mark.parametrize("d,ass,ashs,w,win", [(4, 30, 80, 0.0, 0), (9, 20, 0, 0.05, 0), (12, 25, 70, 0.0, 30)])
def test_user_target_injected_streams_match_the_oracle_closure(cuda, oracle, d, ass, ashs, w, win):
    """adaptMCMC_source.cpp:108-216 with target = a closure (oracle) / the same function compiled (device)."""
    import mcmc_amd

    prm = [1.0, 5.0, 0.05]
    tgt = mcmc_amd.Target.compile(RIDGE_SRC, d, prm)
    nc, n = 70, 700
    rng = np.random.default_rng(100 + d)
    init = 0.5 * rng.standard_normal((d, nc))
    Z = rng.standard_normal((n, d, nc))
    U = rng.random((n, nc))
    out = mcmc_amd.adapt_rwm_batch(tgt, nc, cuda.tensor(init, device="cuda"), 0.25 * np.eye(d), n, ass, w, win, ashs,
                                   Z=cuda.tensor(Z, device="cuda"), U=cuda.tensor(U, device="cuda"), diagnostics=True)
    cuda.cuda.synchronize()
    assert np.all(out["status"].cpu().numpy() == 0)
    tr, acc = out["theta_trace"].cpu().numpy(), out["accepted"].cpu().numpy()
    rate = out["rate_trace"].cpu().numpy()
    ot = oracle.callable_target(ridge(prm))
    for c in (0, 63, 64, 69):
        ref = oracle.adaptMCMC(ot, init[:, c], 0.25 * np.eye(d), n, ass, w, win, ashs, Z[:, :, c], U[:, c], diagnostics=True)
        np.testing.assert_array_equal(acc[:, c], ref["diag"]["accepted"])
        assert _rel(tr[:, :, c], ref["theta_trace"]) < RTOL, c
        np.testing.assert_array_equal(rate[:, c], ref["diag"]["acceptance_rate"])
        assert 0.02 < ref["acceptance_rate"] < 0.9


def test_user_banana_equals_the_builtin_banana_and_pair_kernel_serves_user_targets(cuda, oracle):
    import mcmc_amd

    user = mcmc_amd.Target.compile(BANANA_SRC, 2, [0.03])
    builtin = mcmc_amd.Target.banana(0.03)
    n, nc = 600, 130
    for kernel in ("lane", "pair"):
        a = mcmc_amd.adapt_rwm_batch(user, nc, [10.0, 10.0], np.eye(2), n, 10, 0.0, 0, 20, seed=4, kernel=kernel)
        b = mcmc_amd.adapt_rwm_batch(builtin, nc, [10.0, 10.0], np.eye(2), n, 10, 0.0, 0, 20, seed=4, kernel=kernel)
        np.testing.assert_array_equal(a["acceptance_rate"].cpu().numpy(), b["acceptance_rate"].cpu().numpy())
        assert _rel(a["theta_trace"].cpu().numpy(), b["theta_trace"].cpu().numpy()) < RTOL
    # same handle, other parameter values: no recompilation
    other = user.with_params([0.1])
    a = mcmc_amd.adapt_rwm_batch(other, nc, [3.0, 3.0], np.eye(2), n, 10, 0.0, 0, 20, seed=4)
    b = mcmc_amd.adapt_rwm_batch(mcmc_amd.Target.banana(0.1), nc, [3.0, 3.0], np.eye(2), n, 10, 0.0, 0, 20, seed=4)
    assert _rel(a["theta_trace"].cpu().numpy(), b["theta_trace"].cpu().numpy()) < RTOL

    # d = 7 ridge: the paired-wave production kernel against the one-wave kernel and the oracle closure, Philox run
    d, prm = 7, [1.0, 5.0, 0.05]
    tgt = mcmc_amd.Target.compile(RIDGE_SRC, d, prm)
    kw = dict(seed=21, chain_offset=5)
    pair = mcmc_amd.adapt_rwm_batch(tgt, nc, np.zeros(d), 0.25 * np.eye(d), n, 20, 0.0, 0, 90, kernel="pair", **kw)
    lane = mcmc_amd.adapt_rwm_batch(tgt, nc, np.zeros(d), 0.25 * np.eye(d), n, 20, 0.0, 0, 90, kernel="lane", **kw)
    auto = mcmc_amd.adapt_rwm_batch(tgt, nc, np.zeros(d), 0.25 * np.eye(d), n, 20, 0.0, 0, 90, **kw)
    assert cuda.equal(auto["theta_trace"], pair["theta_trace"])
    assert np.all(pair["status"].cpu().numpy() == 0)
    assert _rel(pair["theta_trace"].cpu().numpy(), lane["theta_trace"].cpu().numpy()) < RTOL
    np.testing.assert_array_equal(pair["acceptance_rate"].cpu().numpy(), lane["acceptance_rate"].cpu().numpy())
    Z, U = mcmc_amd.rwm_streams(21, nc, n, d, chain_offset=5)
    ot = oracle.callable_target(ridge(prm))
    for c in (0, 129):
        ref = oracle.adaptMCMC(ot, np.zeros(d), 0.25 * np.eye(d), n, 20, 0.0, 0, 90, Z.cpu().numpy()[:, :, c],
                               U.cpu().numpy()[:, c], store_sigma=False)
        assert _rel(pair["theta_trace"].cpu().numpy()[:, :, c], ref["theta_trace"]) < RTOL
        assert pair["acceptance_rate"].cpu().numpy()[c] == ref["acceptance_rate"]


def test_user_target_minus_inf_support_boundary_and_errors(cuda, oracle):
    """-Inf outside the support rejects (adaptMCMC_source.cpp:163-165); the chain never leaves the box.
    Stationary law: independent Beta(p_k + 1, q_k + 1)."""
    import mcmc_amd
    from scipy import stats

    d = 3
    prm = np.array([1.0, 2.0, 0.5, 3.0, 1.0, 0.5])
    tgt = mcmc_amd.Target.compile(BOX_SRC, d, prm)
    nc, n = 512, 3000
    rng = np.random.default_rng(8)
    Z = rng.standard_normal((200, d, 4))
    U = rng.random((200, 4))
    out = mcmc_amd.adapt_rwm_batch(tgt, 4, 0.5 * np.ones(d), 0.09 * np.eye(d), 200, 20, 0.0, 0, 60,
                                   Z=cuda.tensor(Z, device="cuda"), U=cuda.tensor(U, device="cuda"), diagnostics=True)
    ref = oracle.adaptMCMC(oracle.callable_target(box(prm, d)), 0.5 * np.ones(d), 0.09 * np.eye(d), 200, 20, 0.0, 0, 60,
                           Z[:, :, 2], U[:, 2], diagnostics=True)
    np.testing.assert_array_equal(out["accepted"].cpu().numpy()[:, 2], ref["diag"]["accepted"])
    assert np.isneginf(ref["diag"]["log_acceptance"]).sum() > 5  # the boundary was hit
    assert _rel(out["theta_trace"].cpu().numpy()[:, :, 2], ref["theta_trace"]) < RTOL
    run = mcmc_amd.adapt_rwm_batch(tgt, nc, 0.5 * np.ones(d), 0.09 * np.eye(d), n, 50, 0.0, 0, 200, seed=3, thin=10)
    x = run["theta_trace"].cpu().numpy()[100:]  # [kept, d, chains]
    assert np.all((x > 0) & (x < 1))
    for k in range(d):
        p = stats.kstest(x[::10, k, :].ravel(), stats.beta(prm[k] + 1, prm[d + k] + 1).cdf).pvalue
        assert p > 1e-3, (k, p)
    # misuse is refused with a message, not a crash
    with pytest.raises(mcmc_amd.McmcError, match="compiled for d=3"):
        mcmc_amd.adapt_rwm_batch(tgt, 4, np.zeros(4), np.eye(4), 10)
    with pytest.raises(mcmc_amd.McmcError, match="undeclared identifier"):
        mcmc_amd.Target.compile("__device__ double mcmc_user_target(const double (&x)[MCMC_USER_D], const double *p) { return y; }", 2)
    with pytest.raises(mcmc_amd.McmcError):
        mcmc_amd.Target.compile(RIDGE_SRC, 129, [1.0, 5.0, 0.05])


# ---- user targets for 12 < d <= 128: one chain per workgroup, the log-density as a sum of per-thread parts ----------
# Bayesian logistic regression: params = tau, n_obs, X [n_obs][d] row-major, y [n_obs]
LOGIT_SRC = r"""
__device__ double mcmc_user_target_part(const double *x, int d, const double *params, int part, int nparts)
{
    const double tau = params[0];
    const int n_obs = (int)params[1];
    const double *X = params + 2, *y = X + (long)n_obs * d;
    double s = 0.0;
    for (int r = part; r < n_obs; r += nparts) {
        double eta = 0.0;
        for (int k = 0; k < d; ++k) eta += X[(long)r * d + k] * x[k];
        const double m = eta > 0 ? eta : 0.0; // y eta - log(1 + exp(eta)), without overflow
        s += y[r] * eta - (m + log(exp(-m) + exp(eta - m)));
    }
    for (int k = part; k < d; k += nparts) s -= 0.5 * tau * x[k] * x[k];
    return s;
}
"""


def logit_problem(d, n_obs, seed):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n_obs, d)) / np.sqrt(d)
    beta = rng.standard_normal(d)
    y = (rng.random(n_obs) < 1.0 / (1.0 + np.exp(-X @ beta))).astype(np.float64)
    tau = 0.5
    prm = np.concatenate([[tau, n_obs], X.ravel(), y])

    def f(x):
        x = np.asarray(x, dtype=np.float64).reshape(-1)
        eta = X @ x
        m = np.maximum(eta, 0.0)
        return float(np.sum(y * eta - (m + np.log(np.exp(-m) + np.exp(eta - m)))) - 0.5 * tau * np.sum(x * x))

    return prm, f


@pytest.mark.parametrize("d,n_obs", [(13, 50), (40, 300), (128, 700)])
def test_wide_user_target_values_match_the_host_closure(cuda, d, n_obs):
    import mcmc_amd

    prm, f = logit_problem(d, n_obs, d)
    tgt = mcmc_amd.Target.compile(LOGIT_SRC, d, prm)
    x = np.random.default_rng(1).standard_normal((d, 200))
    got = mcmc_amd.target_eval_batch(tgt, cuda.tensor(x, device="cuda")).cpu().numpy()
    want = np.array([f(x[:, c]) for c in range(x.shape[1])])
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12)  # the sum is associated per thread, then over the workgroup


@pytest.mark.parametrize("d,ass,ashs,w,win", [(20, 20, 60, 0.0, 0), (48, 15, 0, 0.05, 0)])
def test_wide_user_target_injected_streams_match_the_oracle_closure(cuda, oracle, d, ass, ashs, w, win):
    """adaptMCMC_source.cpp:108-216 with target = a closure (oracle) / the same function compiled into the
    per-workgroup kernel (device).  The log-density sum is associated differently (256 strided parts), so the
    bar is the contract's: decisions equal unless the margin is below 1e-12, trajectories within 1e-12."""
    import mcmc_amd

    prm, f = logit_problem(d, 120, 5)
    tgt = mcmc_amd.Target.compile(LOGIT_SRC, d, prm)
    nc, n = 6, 400
    rng = np.random.default_rng(200 + d)
    init = 0.1 * rng.standard_normal((d, nc))
    Z = rng.standard_normal((n, d, nc))
    U = rng.random((n, nc))
    cov0 = (0.5 / d) * np.eye(d)
    out = mcmc_amd.adapt_rwm_batch(tgt, nc, cuda.tensor(init, device="cuda"), cov0, n, ass, w, win, ashs,
                                   Z=cuda.tensor(Z, device="cuda"), U=cuda.tensor(U, device="cuda"), diagnostics=True)
    cuda.cuda.synchronize()
    assert np.all(out["status"].cpu().numpy() == 0)
    tr, acc = out["theta_trace"].cpu().numpy(), out["accepted"].cpu().numpy()
    ot = oracle.callable_target(f)
    for c in (0, 5):
        ref = oracle.adaptMCMC(ot, init[:, c], cov0, n, ass, w, win, ashs, Z[:, :, c], U[:, c], diagnostics=True)
        diff = np.nonzero(acc[:, c] != ref["diag"]["accepted"])[0]
        if diff.size:
            first = diff[0]
            margin = abs(ref["diag"]["log_A"][first] - ref["diag"]["log_acceptance"][first])
            assert margin <= 1e-12 * max(1.0, abs(ref["diag"]["pdf_term"][first])), (first, margin)
            continue
        assert _rel(tr[:, :, c], ref["theta_trace"]) < RTOL, c
        assert 0.02 < ref["acceptance_rate"] < 0.95
    # Philox run: re-sharding and a split run give the same bits
    kw = dict(seed=9, thin=3)
    whole = mcmc_amd.adapt_rwm_batch(tgt, 40, np.zeros(d), cov0, 300, ass, w, win, ashs, **kw)
    part = mcmc_amd.adapt_rwm_batch(tgt, 10, np.zeros(d), cov0, 300, ass, w, win, ashs, chain_offset=30, **kw)
    assert cuda.equal(whole["theta_trace"][:, :, 30:], part["theta_trace"])
    a = mcmc_amd.adapt_rwm_batch(tgt, 40, np.zeros(d), cov0, 100, ass, w, win, ashs, checkpoint=True, **kw)
    b = mcmc_amd.adapt_rwm_batch(tgt, 40, np.zeros(d), cov0, 200, ass, w, win, ashs, resume=a["state"], **kw)
    assert cuda.equal(cuda.cat([a["theta_trace"], b["theta_trace"]], dim=0), whole["theta_trace"])
    with pytest.raises(mcmc_amd.McmcError, match="mcmc_user_target_part"):
        mcmc_amd.Target.compile(RIDGE_SRC, 20, [1.0, 5.0, 0.05])
