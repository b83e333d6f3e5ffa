"""Multi-chain glue in the shape the reference's R helpers expect.

The reference's only parallelism is one chain per R worker process: `mclapply_RNG(X, FUN)`
(MCMCutilty.R:103-149) starts a socket cluster, gives every worker its own L'Ecuyer stream
(:137) and returns `list(FUN(X[[1]]), FUN(X[[2]]), ...)`; `melt_trace(chain)` (:19-35) turns such a
list -- every element carrying `$trace`, an iterations x parameters data frame -- into one long data
frame with `chain` and `iteration` columns for the ggplot helpers (:38-95).

Here all chains of the list run in ONE launch (`run_chains`): element c of X is the start point of
chain c, whose random stream is Philox keyed by (seed, chain_offset + c) instead of a worker's
L'Ecuyer stream.  The result has the reference's list-of-chains shape, so `melt_trace` below -- and,
through the same layout in R, the reference's own plotting helpers -- work on it unchanged.
"""
from __future__ import annotations

import numpy as np

from . import api


def chain_list(trace, par_names=None, acceptance_rate=None, chains=None):
    """[n_kept, d, n_chains] trace (CUDA tensor or array) -> list of {"trace": n_kept x d array, ...}.

    `chains` selects a subset (indices) before anything is copied to the host."""
    if hasattr(trace, "detach"):
        if chains is not None:
            trace = trace[:, :, list(chains)]
        trace = trace.detach().cpu().numpy()
    elif chains is not None:
        trace = np.asarray(trace)[:, :, list(chains)]
    trace = np.asarray(trace)
    n_kept, d, n_chains = trace.shape
    names = list(par_names) if par_names is not None else [f"V{k + 1}" for k in range(d)]  # R's default column names
    if len(names) != d:
        raise ValueError(f"{len(names)} parameter names for d = {d}")
    acc = None
    if acceptance_rate is not None:
        acc = acceptance_rate.detach().cpu().numpy() if hasattr(acceptance_rate, "detach") else np.asarray(acceptance_rate)
        if chains is not None:
            acc = acc[list(chains)]
    out = []
    for c in range(n_chains):
        item = {"trace": np.ascontiguousarray(trace[:, :, c]), "par_names": names}
        if acc is not None:
            item["acceptance_rate"] = float(acc[c])
        out.append(item)
    return out


def melt_trace(chain):
    """melt_trace (MCMCutilty.R:19-35): list of chains -> long data frame.

    Columns: the parameters, then `chain` (1-based, categorical) and `iteration` (1-based), rows ordered
    chain by chain exactly as `ldply` stacks them."""
    import pandas as pd

    if len(chain) == 0:
        raise ValueError("empty chain list")
    n = chain[0]["trace"].shape[0]
    names = chain[0].get("par_names") or [f"V{k + 1}" for k in range(chain[0]["trace"].shape[1])]
    for x in chain:
        if x["trace"].shape != chain[0]["trace"].shape:
            raise ValueError("all chains must have the same trace shape (the reference assumes it, :27-30)")
    df = pd.DataFrame(np.concatenate([x["trace"] for x in chain], axis=0), columns=names)
    df["chain"] = pd.Categorical(np.repeat(np.arange(1, len(chain) + 1), n))
    df["iteration"] = np.tile(np.arange(1, n + 1), len(chain))
    return df


def run_chains(X, target, covmat, n_iterations, adapt_size_start=0, acceptance_rate_weight=0.0,
               acceptance_window=0, adapt_shape_start=0, *, seed=0, par_names=None, thin=1, chain_offset=0, **kw):
    """The job `mclapply_RNG(X, function(x) adaptMCMC(target, x, covmat, ...))` does in the reference
    (MCMCutilty.R:103-149), as one GPU launch: X is the list of per-chain start points.

    Returns the reference's list-of-chains shape: element c has `trace` (n_kept x d), `acceptance_rate`,
    `status`; feed it to `melt_trace`."""
    torch = api._torch()
    X = np.asarray(X, dtype=np.float64)
    if X.ndim != 2:
        raise ValueError("X must be a sequence of equally long start vectors")
    n_chains, d = X.shape
    init = torch.tensor(np.ascontiguousarray(X.T), dtype=torch.float64, device="cuda")
    out = api.adapt_rwm_batch(target, n_chains, init, covmat, n_iterations, adapt_size_start, acceptance_rate_weight,
                              acceptance_window, adapt_shape_start, seed=seed, thin=thin, chain_offset=chain_offset, **kw)
    chains = chain_list(out["theta_trace"], par_names, out["acceptance_rate"])
    status = out["status"].cpu().numpy()
    for c, item in enumerate(chains):
        item["status"] = int(status[c])
    return chains
