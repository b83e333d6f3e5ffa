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


def test_rccl_backend_initialises_and_all_reduces_f64(cuda):
    """The "nccl" backend of torch.distributed IS RCCL on ROCm.  A one-GPU box cannot hold two RCCL ranks (one
    communicator per device), so this is the one-rank rehearsal of exactly the calls bench.py and diagnostics.rhat
    issue on a node: init_process_group("nccl", device_id=...), barrier, all-reduce MAX and SUM of float64 device
    tensors, the partial sums of mcmc_rhat_partial through it -- in a fresh process, as under a launcher
    (tools/rccl_check.py).  The driver's 8-GPU run is then not the first time the backend executes."""
    import os
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "rccl_check.py")], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-1500:])
    assert "rccl one-rank rehearsal OK" in p.stdout


def test_batch_calls_return_without_waiting_for_the_stream(cuda):
    """include/mcmc_amd.h: a batch call is asynchronous on its stream -- the parameter block travels in kernel-argument
    fills, nothing queued earlier is waited for.  A long launch followed by a short one: both calls are back on the
    host long before the GPU has finished the first."""
    import time

    import mcmc_amd

    d = 10
    idx = np.arange(d)
    H = np.linalg.inv(0.9 ** np.abs(idx[:, None] - idx[None, :]))
    tgt = mcmc_amd.Target.gaussian(np.zeros(d), 0.5 * (H + H.T))
    tgt4 = mcmc_amd.Target.gaussian(np.zeros(4), np.eye(4))
    out_big = {"theta_trace": cuda.empty((4000, d, 65536), dtype=cuda.float64, device="cuda")}
    mu, sigma = np.zeros(d), 0.5 ** np.abs(idx[:, None] - idx[None, :])
    for _ in range(2):  # the first round loads code objects and warms the allocator
        cuda.cuda.synchronize()
        t0 = time.perf_counter()
        mcmc_amd.adapt_rwm_batch(tgt, 65536, np.ones(d), np.eye(d), 4000, 100, 0.0, 0, 500, seed=1, out=out_big, final=False)
        mcmc_amd.tmvn_gibbs_batch(4096, 10, mu, sigma, np.full(d, -1.0), np.full(d, 2.0), np.zeros(d), seed=2)
        # (kernels without scratch memory behind the long one: the HIP runtime itself holds the launch of a kernel that
        # uses scratch -- the d = 10 paired-wave kernel has 56 bytes per lane, the d <= 16 row kernel 92 -- on the host while
        # an earlier scratch-using launch is in flight and another launch sits between them; measured, not the library's doing)
        mcmc_amd.adapt_rwm_batch(tgt4, 640, np.ones(4), np.eye(4), 50, 10, 0.0, 0, 20, seed=3)
        t1 = time.perf_counter()
        cuda.cuda.synchronize()
        t2 = time.perf_counter()
    assert t2 - t0 > 0.012, t2 - t0             # the long launch alone is ~19 ms of GPU time
    assert t1 - t0 < 0.5 * (t2 - t0), (t1 - t0, t2 - t0)
