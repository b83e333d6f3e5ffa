#!/usr/bin/env python3
"""Growth of the trajectory error against the oracle with the number of iterations: Philox runs of
the row kernel (d = 63) and the per-workgroup kernel (d = 100) replayed through the oracle, error = max over chains,
coordinates and iterations <= n of |x - ref| / max(1, |ref|).  A random walk does not contract rounding differences,
so the error of a chain grows like a random walk itself (~ sqrt(n)) until a decision flips.
Usage (GPU box): python3 tools/parity_growth.py > profiles/r3_parity_growth.txt"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mcmc_amd
from oracle import oracle as O

O.build()
for d, kernel, ass, ashs, w in ((63, "row", 25, 200, 0.05), (63, "block", 25, 200, 0.05), (100, "block", 0, 300, 0.0)):
    n, nc = 4000, 8
    rng = np.random.default_rng(2000 + d)
    A = rng.standard_normal((d, d))
    H = np.linalg.inv(A @ A.T / d + np.eye(d)); H = 0.5 * (H + H.T)
    m = 0.3 * rng.standard_normal(d)
    cov0 = (0.8 / d) * np.eye(d) + (0.1 / d) * np.ones((d, d))
    init = mcmc_amd.init_normal(9, nc, m, 1.0, chain_offset=500)
    out = mcmc_amd.adapt_rwm_batch(mcmc_amd.Target.gaussian(m, H), nc, init, cov0, n, ass, w, 0, ashs, kernel=kernel, seed=21, chain_offset=500)
    Z, U = mcmc_amd.rwm_streams(21, nc, n, d, chain_offset=500)
    tr, Z, U, x0 = out["theta_trace"].cpu().numpy(), Z.cpu().numpy(), U.cpu().numpy(), init.cpu().numpy()
    errs = []
    for c in range(nc):
        ref = O.adaptMCMC(O.gaussian_target(m, H), x0[:, c], cov0, n, ass, w, 0, ashs, Z[:, :, c], U[:, c], store_sigma=False)["theta_trace"]
        errs.append(np.max(np.abs(tr[:, :, c] - ref) / np.maximum(1.0, np.abs(ref)), axis=1))
    e = np.maximum.accumulate(np.array(errs), axis=1)  # running max per chain
    print(f"d={d} kernel={kernel} adapt {ass}/{ashs} weight {w}: shape-phase fraction {float(out['shape_iterations'].double().mean())/n:.2f}")
    for k in (250, 500, 1000, 2000, 4000):
        print(f"   n <= {k:5d}: max over 8 chains {e[:, k - 1].max():.2e}   median chain {np.median(e[:, k - 1]):.2e}")
