"""Generates tests/golden/*.npz from the CPU oracle (oracle/mcmc_oracle.c).

The reference ships no expected outputs and cannot be built here, so
these vectors are the ORACLE's outputs on the inputs the reference author is known to have
exercised (tmvrnormGibbs.cpp:72-78,144-163; adaptMCMC_analysis.R:19-28,115-134;
deprecated/adaptMCMC.R:145-165) plus edge cases.  "Parity unpinned" applies: they pin the
oracle against regressions and give the GPU tests fixed inputs, they are not reference output.
The only entry with an independent known answer is G1 (hand-derivable).

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def save(name, **kw):
    np.savez_compressed(os.path.join(OUT, name), **kw)
    print(name, {k: np.asarray(v).shape for k, v in kw.items()})


def main():
    O.build()
    rng = np.random.default_rng(20240611)

    # G1 update_sigma KAT (deprecated/adaptMCMC.R:146-155): hand-derived [[4.01,.06],[.06,.44]]
    cov = np.array([[5.0, 0.0], [0.0, 0.1]])
    res = np.array([0.25, 1.5])
    save("g1_update_sigma.npz", cov=cov, residual=res, i=5.0, expected=O.update_sigma(cov, res, 5.0),
         hand=np.array([[4.01, 0.06], [0.06, 0.44]]))

    # G2 rtnorm_gibbs grid (tmvrnormGibbs.cpp:72-78 uses (-Inf, Inf))
    u = np.concatenate([[1e-12, 1e-9, 1e-6, 1e-3], np.linspace(0.01, 0.99, 99), [1 - 1e-3, 1 - 1e-6, 1 - 1e-9, 1 - 1e-12]])
    bounds = np.array([[-np.inf, np.inf], [0.0, np.inf], [-np.inf, -9.0], [6.0, np.inf], [-1.0, 1.0], [9.0, np.inf]])
    outs = np.stack([O.rtnorm_gibbs(u.size, 0.0, 1.0, a, b, u) for a, b in bounds])
    outs2 = np.stack([O.rtnorm_gibbs(u.size, 1.5, 2.5, a, b, u) for a, b in bounds])
    save("g2_rtnorm.npz", u=u, bounds=bounds, out_std=outs, out_mu1p5_sd2p5=outs2)

    # G3 4-D diagonal fixture (tmvrnormGibbs.cpp:146), first 1000 sweeps
    n = 1000
    U = rng.random(n * 4)
    mu, sigma = np.arange(1.0, 5.0), np.diag(np.arange(1.0, 5.0))
    lower, upper = np.array([-np.inf, -np.inf, 0.0, 0.0]), np.array([10.0, 10.0, 100.0, 100.0])
    init = np.array([2.0, 2.0, 50.0, 50.0])
    save("g3_gibbs_diag4.npz", n=n, mu=mu, sigma=sigma, lower=lower, upper=upper, init=init, U=U,
         trace=O.rtmvnorm_gibbs(n, mu, sigma, lower, upper, init, U))

    # G4 2-D correlated fixture (tmvrnormGibbs.cpp:153-159)
    U = rng.random(n * 2)
    mu, sigma = np.array([-1.0, 1.0]), np.array([[1.0, -0.5], [-0.5, 1.0]])
    lower, upper, init = np.array([-np.inf, 0.0]), np.array([10.0, 10.0]), np.array([2.0, 2.0])
    save("g4_gibbs_corr2.npz", n=n, mu=mu, sigma=sigma, lower=lower, upper=upper, init=init, U=U,
         trace=O.rtmvnorm_gibbs(n, mu, sigma, lower, upper, init, U))

    # G5 adaptMCMC on banana (adaptMCMC_analysis.R:19-28) and Goldstein-Price (:115-134), 2000 iterations
    n = 2000
    for name, tgt, init, ass, ashs in (("banana", O.banana_target(0.03), [10.0, 10.0], 10, 20),
                                       ("goldpr", O.goldstein_price_target(), [1.0, 1.0], 100, 150)):
        Z = rng.standard_normal((n, 2))
        Uu = rng.random(n)
        r = O.adaptMCMC(tgt, init, np.eye(2), n, ass, 0.0, 0, ashs, Z, Uu, diagnostics=True)
        dg = r["diag"]
        save(f"g5_adapt_{name}.npz", n=n, init=np.array(init), adapt_size_start=ass, adapt_shape_start=ashs, Z=Z, U=Uu,
             theta_trace=r["theta_trace"], sigma_last=r["sigma_empirical"][:, :, -1],
             sigma_every100=r["sigma_empirical"][:, :, 99::100], acceptance_rate=r["acceptance_rate"],
             accepted=dg["accepted"].astype(np.uint8), log_acceptance=dg["log_acceptance"], log_A=dg["log_A"],
             pdf_term=dg["pdf_term"], scaling_sd=dg["scaling_sd"], rate=dg["acceptance_rate"])

    # G6 the three acceptance-rate modes (adaptMCMC_source.cpp:188-207) on a 3-D Gaussian
    n, d = 1500, 3
    A = rng.standard_normal((d, d))
    Sig = A @ A.T / d + np.eye(d)
    H = np.linalg.inv(Sig)
    H = 0.5 * (H + H.T)
    Z = rng.standard_normal((n, d))
    Uu = rng.random(n)
    g6 = dict(n=n, H=H, Z=Z, U=Uu, init=np.array([1.0, -1.0, 0.5]))
    for tag, w, win in (("mean", 0.0, 0), ("weight", 0.05, 0), ("window", 0.0, 50)):
        r = O.adaptMCMC(O.gaussian_target(np.zeros(d), H), g6["init"], np.eye(d), n, 50, w, win, 200, Z, Uu,
                        store_sigma=False, diagnostics=True)
        g6[f"trace_{tag}"] = r["theta_trace"]
        g6[f"accepted_{tag}"] = r["diag"]["accepted"].astype(np.uint8)
        g6[f"rate_{tag}"] = r["diag"]["acceptance_rate"]
        g6[f"scaling_sd_{tag}"] = r["diag"]["scaling_sd"]
    save("g6_rate_modes.npz", **g6)

    # G7 NaN / -Inf paths: far jumps make exp underflow -> p = -Inf -> NaN -> reject
    n, d = 200, 2
    Z = rng.standard_normal((n, d))
    Z[10] = [40.0, 40.0]      # q = 3200 > 1490: exp(-q/2) = 0
    Z[50] = [1e200, 0.0]      # target overflows to -Inf: immediate reject
    Uu = rng.random(n)
    r = O.rwMCMC(O.gaussian_target(np.zeros(d), np.eye(d)), [0.1, 0.2], np.eye(d), n, Z, Uu, diagnostics=True)
    save("g7_nan_paths.npz", n=n, Z=Z, U=Uu, init=np.array([0.1, 0.2]), theta_trace=r["theta_trace"],
         accepted=r["diag"]["accepted"].astype(np.uint8), log_acceptance=r["diag"]["log_acceptance"],
         acceptance_rate=r["acceptance_rate"])

    # G8 chol failure (shape phase with a rank-deficient covmat_empirical)
    n, d = 100, 3
    Z = rng.standard_normal((n, d))
    Uu = rng.random(n)
    r = O.adaptMCMC(O.gaussian_target(np.zeros(d), np.eye(d)), [0.0, 0.0, 0.0], np.eye(d), n, 0, 0.0, 0, 1, Z, Uu,
                    store_sigma=False)
    assert r["status"] == O.ORC_CHOL_FAILED
    save("g8_chol_fail.npz", n=n, Z=Z, U=Uu, fail_iter=r["fail_iter"], theta_trace=r["theta_trace"])


if __name__ == "__main__":
    main()
