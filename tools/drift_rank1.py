#!/usr/bin/env python3
"""Drift of MCMC_FACTOR_RANK1 against the exact mode on C4S's inputs (d = 100, covmat 0.05 I, adapt_shape_start 1000):
the same Philox streams through both modes; a chain either keeps the exact mode's decisions for the whole run -- then its
trajectory differs by rounding drift only -- or one decision flips (|log A - la| inside the drift) and the two runs part
for good.  Prints, per refresh_every, how many chains kept every decision and the largest relative trajectory error
among them.  Usage (GPU box): python3 tools/drift_rank1.py [chains] [iterations]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mcmc_amd

chains = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
d = 100
A = np.random.default_rng(4).standard_normal((d, d))
H = np.linalg.inv(A @ A.T / d + np.eye(d)); H = 0.5 * (H + H.T)
tgt = mcmc_amd.Target.gaussian(np.zeros(d), H)
init = mcmc_amd.init_normal(1, chains, np.zeros(d), 1.0)
args = (tgt, chains, init, 0.05 * np.eye(d), n, 0, 0.0, 0, 1000)
ex = mcmc_amd.adapt_rwm_batch(*args, seed=1, thin=10)
te = ex["theta_trace"]
print(f"d={d} chains={chains} iterations={n} thin=10; shape-phase fraction {float(ex['shape_iterations'].double().mean())/n:.3f}; "
      f"chains failed {int((ex['status'] != 0).sum())}")
print("refresh_every  chains_with_all_decisions_equal  max_rel_err_among_them  median_rel_err_last_row  first_split_iteration(min)")
for R in (1, 16, 64, 256, 1 << 20):
    fa = mcmc_amd.adapt_rwm_batch(*args, seed=1, thin=10, factor_update="rank1", refresh_every=R)
    err = ((fa["theta_trace"] - te).abs() / te.abs().clamp(min=1.0)).amax(dim=1)  # [n_kept, chains]
    split = err > 1e-8
    same = ~split.any(dim=0)
    first = torch.where(split.any(dim=0), split.float().argmax(dim=0), torch.full((chains,), err.shape[0], device=err.device))
    mx = float(err[:, same].max()) if bool(same.any()) else float("nan")
    med = float(err[-1, same].median()) if bool(same.any()) else float("nan")
    fs = int(first.min()) * 10
    print(f"{R:>12d}  {int(same.sum()):>8d} of {chains}  {mx:.3e}  {med:.3e}  {fs if fs < n else 'none'}")
