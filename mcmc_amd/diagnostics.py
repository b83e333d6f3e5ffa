"""Cross-chain diagnostics: per-chain moments and ESS on the device, Gelman-Rubin R-hat from
per-GPU partial sums combined by ONE all-reduce (RCCL over xGMI when torch.distributed runs the
"nccl" backend; gloo on CPU for tests).

No counterpart in the reference (coda is loaded at MCMCutilty.R:11 but never called); the
definitions are: R-hat = sqrt(((N-1)/N W + B/N) / W) with W the mean within-chain variance and
B/N the variance of the chain means (Gelman & Rubin 1992, no chain splitting); ESS by Geyer's
initial positive sequence on the autocorrelation.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import check, lib
from .api import _dev_ptr, _stream_ptr, _torch


def chain_moments(trace, burn_kept=0):
    """trace [n_kept, d, n_chains] (CUDA f64) -> (mean, var) each [d, n_chains]."""
    torch = _torch()
    n_kept, d, n = trace.shape
    mean = torch.empty((d, n), dtype=torch.float64, device=trace.device)
    var = torch.empty_like(mean)
    check(lib.mcmc_chain_moments(_dev_ptr(trace), n_kept, int(burn_kept), d, n, _dev_ptr(mean), _dev_ptr(var),
                                 _stream_ptr()))
    return mean, var


def chain_ess(trace, burn_kept=0, max_lag=1000, return_capped=False):
    """Per-chain, per-coordinate effective sample size, [d, n_chains] (Geyer's initial positive sequence, summed up
    to `max_lag`).  With return_capped also an int32 [d, n_chains] array: 1 where the sum was still positive at
    max_lag, i.e. where the cap ended it and the ESS is over-stated."""
    torch = _torch()
    n_kept, d, n = trace.shape
    ess = torch.empty((d, n), dtype=torch.float64, device=trace.device)
    if not return_capped:
        check(lib.mcmc_chain_ess(_dev_ptr(trace), n_kept, int(burn_kept), d, n, int(max_lag), _dev_ptr(ess),
                                 _stream_ptr()))
        return ess
    capped = torch.empty((d, n), dtype=torch.int32, device=trace.device)
    check(lib.mcmc_chain_ess_capped(_dev_ptr(trace), n_kept, int(burn_kept), d, n, int(max_lag), _dev_ptr(ess),
                                    _dev_ptr(capped), _stream_ptr()))
    return ess, capped


def rhat_partial(mean, var):
    """[d, n_chains] per-chain means / variances -> partial sums [3d+1] on the device:
    sum_c m, sum_c m^2, sum_c s^2 per coordinate, then the chain count."""
    torch = _torch()
    d, n = mean.shape
    part = torch.empty((3 * d + 1,), dtype=torch.float64, device=mean.device)
    check(lib.mcmc_rhat_partial(_dev_ptr(mean), _dev_ptr(var), d, n, _dev_ptr(part), _stream_ptr()))
    return part


def rhat_from_partial(part, n_per_chain):
    """Combine (all-reduced) partial sums into R-hat per coordinate. `part` may be a tensor or array."""
    p = part.detach().cpu().numpy() if hasattr(part, "detach") else np.asarray(part, dtype=np.float64)
    d = (p.size - 1) // 3
    M = p[3 * d]
    sm, sm2, ss = p[:d], p[d:2 * d], p[2 * d:3 * d]
    W = ss / M
    B_over_n = (sm2 - sm * sm / M) / (M - 1.0)
    N = float(n_per_chain)
    var_plus = (N - 1.0) / N * W + B_over_n
    return np.sqrt(var_plus / W)


def rhat(mean, var, n_per_chain, group=None):
    """Gelman-Rubin R-hat per coordinate over ALL chains of ALL ranks.

    Each rank reduces its own chains on the device (mcmc_rhat_partial), then one all-reduce(sum)
    of 3d+1 doubles combines the ranks (latency-bound, ~0.3 KB at d=10)."""
    import torch.distributed as dist

    part = rhat_partial(mean, var)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(part, op=dist.ReduceOp.SUM, group=group)
    return rhat_from_partial(part, n_per_chain)


def pooled_cov_partial(mean, cov):
    """[d, n_chains] chain means and [d*d, n_chains] chain covariances (final_mean / final_cov of adapt_rwm_batch)
    -> partial sums [2 d^2 + d + 1] on the device (mcmc_pooled_cov_partial)."""
    torch = _torch()
    d, n = mean.shape
    if tuple(cov.shape) != (d * d, n):
        raise ValueError("cov must be [d*d, n_chains]")
    part = torch.empty((2 * d * d + d + 1,), dtype=torch.float64, device=mean.device)
    check(lib.mcmc_pooled_cov_partial(_dev_ptr(mean), _dev_ptr(cov), d, n, _dev_ptr(part), _stream_ptr()))
    return part


def pooled_cov_from_partial(part):
    """(all-reduced) partial sums -> pooled mean [d] and covariance [d, d] over all chains (equal lengths)."""
    p = part.detach().cpu().numpy() if hasattr(part, "detach") else np.asarray(part, dtype=np.float64)
    d = int(round((-1 + np.sqrt(1 + 8 * (p.size - 1))) / 4))
    assert 2 * d * d + d + 1 == p.size
    M = p[-1]
    sC = p[: d * d].reshape(d, d, order="F")
    smm = p[d * d: 2 * d * d].reshape(d, d, order="F")
    mbar = p[2 * d * d: 2 * d * d + d] / M
    return mbar, (sC + smm) / M - np.outer(mbar, mbar)


def pooled_covariance(mean, cov, group=None):
    """Pooled mean and covariance of the draws of ALL chains of ALL ranks from the per-chain running moments
    (SURVEY.md section 8e): one device reduction per rank, ONE all-reduce(sum) of 2 d^2 + d + 1 doubles."""
    import torch.distributed as dist

    part = pooled_cov_partial(mean, cov)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(part, op=dist.ReduceOp.SUM, group=group)
    return pooled_cov_from_partial(part)
