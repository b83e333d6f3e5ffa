"""Differential fuzzing of the Metropolis kernels on the GPU (development tool, not a test): random shapes and settings,
AUTO (lane / pair / row / block by d) against a forced second kernel on the same Philox run, and a run split at a random
iteration against the uninterrupted one (bit for bit).  Usage: python tools/fuzz_rwm.py [n_cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mcmc_amd

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(n_cases):
    d = int(rng.choice([2, 5, 10, 11, 12, 13, 16, 17, 31, 32, 33, 47, 48, 49, 63, 64, 65, 80, 97, 100, 112, 113, 128]))
    n = int(rng.integers(1, 120)) if rng.random() < 0.5 else int(rng.integers(2 * d, 6 * d + 50)); nc = int(rng.integers(1, 45)); thin = int(rng.integers(1, 6))
    mode = int(rng.integers(0, 3))
    w, win = (0.0, 0) if mode == 0 else ((float(rng.uniform(0.01, 0.3)), 0) if mode == 1 else (0.0, int(rng.choice([3, 64, 65, 130]))))
    ass = int(rng.integers(0, 30))
    # adapt_shape_start > d, or covmat_empirical is rank deficient and WHETHER its factorization fails is a matter of rounding
    # (and with the weighted / windowed rate estimates rate * i says nothing about the number of accepted moves)
    ashs = int(rng.choice([0, d + 8, 2 * d])) if mode == 0 else 0
    off = int(rng.integers(0, 1000))
    A = rng.standard_normal((d, d)); Sig = A @ A.T / d + np.eye(d); H = np.linalg.inv(Sig); H = 0.5 * (H + H.T)
    m = 0.2 * rng.standard_normal(d)
    tgt = mcmc_amd.Target.gaussian(m, H)
    cov0 = (0.5 / d) * np.eye(d)
    args = (tgt, nc, m, cov0)
    ad = (ass, w, win, ashs)
    kw = dict(seed=int(rng.integers(1, 1000)), chain_offset=off, thin=thin)
    other = "lane" if d <= 12 else "block"
    try:
        a = mcmc_amd.adapt_rwm_batch(*args, n, *ad, **kw)
        b = mcmc_amd.adapt_rwm_batch(*args, n, *ad, kernel=other, **kw)
        torch.cuda.synchronize()
        sa, sb = a["status"].cpu().numpy(), b["status"].cpu().numpy()
        ta, tb = a["theta_trace"].cpu().numpy(), b["theta_trace"].cpu().numpy()
        # chains whose covmat_empirical is (numerically) singular fail in every kernel, but AT WHICH iteration the pivot
        # turns non-positive is a matter of rounding: compare the chains that are fine in both
        good = (sa == 0) & (sb == 0)
        ok = np.array_equal(sa != 0, sb != 0)
        ta, tb = ta[:, :, good], tb[:, :, good]
        err = float(np.max(np.abs(ta - tb) / np.maximum(1, np.abs(tb)))) if ta.size else 0.0
        ra, rb = a["acceptance_rate"].cpu().numpy()[good], b["acceptance_rate"].cpu().numpy()[good]
        ok = ok and err < 1e-11 and np.array_equal(ra, rb)
        # split run
        cut = int(rng.integers(1, n)) if n > 1 else 0
        if cut:
            p1 = mcmc_amd.adapt_rwm_batch(*args, cut, *ad, checkpoint=True, **kw)
            p2 = mcmc_amd.adapt_rwm_batch(*args, n - cut, *ad, resume=p1["state"], **kw)
            torch.cuda.synchronize()
            whole = torch.cat([p1["theta_trace"], p2["theta_trace"]])
            same = whole.shape == a["theta_trace"].shape and torch.equal(torch.nan_to_num(whole, nan=7.0), torch.nan_to_num(a["theta_trace"], nan=7.0))
            same = same and torch.equal(p2["acceptance_rate"], a["acceptance_rate"]) and torch.equal(p2["status"], a["status"])
        else:
            same = True
        tag = "ok " if (ok and same) else "BAD"
        bad += tag == "BAD"
        print(f"{tag} d={d} n={n} nc={nc} thin={thin} ass={ass} ashs={ashs} w={w:.3f} win={win} off={off} cut={cut} err={err:.2e} "
              f"status_nonzero={int((sa != 0).sum())} kernels_equal={ok} resume_equal={same}", flush=True)
    except mcmc_amd.McmcError as e:
        print(f"ERR d={d} n={n}: {e}", flush=True)
        bad += 1
print("bad cases:", bad)
sys.exit(1 if bad else 0)
