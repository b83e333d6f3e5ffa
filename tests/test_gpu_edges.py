"""GPU: smallest and ragged shapes of every kernel -- one chain, one iteration, d = 1, a chain count that fills no
wavefront, thinning that keeps nothing -- against the oracle fed with the streams the device reports."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RTOL = 1e-12


def _rel(a, b):
    return np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b)))


@pytest.mark.parametrize("kernel,d,nc,n", [("pair", 1, 1, 1), ("pair", 1, 3, 40), ("lane", 1, 1, 1), ("pair", 10, 1, 1), ("lane", 12, 65, 2),
                                           ("block", 13, 1, 1), ("block", 17, 3, 30), ("block", 128, 2, 3)])
def test_rwm_smallest_shapes(cuda, oracle, kernel, d, nc, n):
    import mcmc_amd

    idx = np.arange(d)
    H = np.linalg.inv(0.5 ** np.abs(idx[:, None] - idx[None, :])) if d > 1 else np.array([[2.0]])
    H = 0.5 * (H + H.T)
    ashs = 5 if d == 1 else 0  # a shape phase after 5 accepted jumps needs d <= 5: covmat_empirical must have full rank (:153)
    out = mcmc_amd.adapt_rwm_batch(mcmc_amd.Target.gaussian(np.zeros(d), H), nc, 0.3 * np.ones(d), np.eye(d), n, 2, 0.0, 0, ashs,
                                   seed=4, kernel=kernel)
    assert tuple(out["theta_trace"].shape) == (n, d, nc) and np.all(out["status"].cpu().numpy() == 0)
    Z, U = mcmc_amd.rwm_streams(4, nc, n, d)
    for c in {0, nc - 1}:
        ref = oracle.adaptMCMC(oracle.gaussian_target(np.zeros(d), H), 0.3 * np.ones(d), np.eye(d), n, 2, 0.0, 0, ashs,
                               Z.cpu().numpy()[:, :, c], U.cpu().numpy()[:, c], store_sigma=False)
        assert _rel(out["theta_trace"].cpu().numpy()[:, :, c], ref["theta_trace"]) < RTOL
        assert out["acceptance_rate"].cpu().numpy()[c] == ref["acceptance_rate"]
    # thinning that keeps nothing: no trace rows, the final state is still there
    none = mcmc_amd.adapt_rwm_batch(mcmc_amd.Target.gaussian(np.zeros(d), H), nc, 0.3 * np.ones(d), np.eye(d), n, 2, 0.0, 0, ashs,
                                    seed=4, kernel=kernel, thin=n + 1)
    assert none["theta_trace"].shape[0] == 0
    assert cuda.equal(none["final_theta"], out["final_theta"])


@pytest.mark.parametrize("d,nc,n", [(1, 1, 1), (2, 1, 1), (32, 1, 2), (33, 1, 1), (64, 3, 2), (65, 1, 1), (129, 2, 2), (256, 1, 1)])
def test_gibbs_smallest_shapes(cuda, oracle, d, nc, n):
    import mcmc_amd

    idx = np.arange(d)
    sigma = 0.4 ** np.abs(idx[:, None] - idx[None, :])
    mu, lower, upper, init = 0.1 * np.ones(d), -np.ones(d), 2 * np.ones(d), np.zeros(d)
    out = mcmc_amd.tmvn_gibbs_batch(nc, n, mu, sigma, lower, upper, init, seed=6)
    assert tuple(out["trace"].shape) == (n, d, nc) and np.all(out["status"].cpu().numpy() == 0)
    U = mcmc_amd.gibbs_streams(6, nc, n, d).cpu().numpy()
    for c in {0, nc - 1}:
        want = oracle.rtmvnorm_gibbs(n, mu, sigma, lower, upper, init, U[:, :, c].reshape(-1))
        assert _rel(out["trace"].cpu().numpy()[:, :, c], want) < RTOL
    assert cuda.equal(out["final_state"], out["trace"][-1])


def test_bench_two_ranks_on_one_gpu(cuda):
    """bench.py --gpus 2 run plainly: two child ranks (gloo here, since both share the one GPU of the test box; RCCL
    on a real node), chains keyed by global id, ONE JSON line with n_gpus = 2 and the all-reduced diagnostics."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["MCMC_BENCH_BACKEND"] = "gloo"
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--chains", "512", "--iters", "400",
                        "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    ln = lines[0]
    assert ln["n_gpus"] == 2 and ln["collective_ranks"] == 2 and ln["config"]["chains_per_gpu"] == 512
    assert ln["value"] > 0 and ln["chains_failed"] == 0 and ln["rhat_max"] is not None and ln["ess_per_sec"] > 0
