"""CPU suite: the N>1 path -- chain sharding and the single all-reduce of moment sums -- with two
gloo ranks.  On GPUs the same code runs over RCCL (backend "nccl"); the per-rank partial sums come
from mcmc_rhat_partial, whose contract (sum m, sum m^2, sum s^2, count) is reproduced here in numpy."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _partial(mean, var):
    d = mean.shape[0]
    out = np.empty(3 * d + 1)
    out[:d], out[d:2 * d], out[2 * d:3 * d], out[3 * d] = mean.sum(1), (mean ** 2).sum(1), var.sum(1), mean.shape[1]
    return out


def _worker(rank, world, port, total_chains, n_per_chain, d, q):
    import torch
    import torch.distributed as dist

    from mcmc_amd.diagnostics import rhat_from_partial
    from mcmc_amd.sharding import all_reduce_sum, shard_chains

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    off, cnt = shard_chains(total_chains, rank, world)
    rng = np.random.default_rng(123)
    draws = rng.standard_normal((n_per_chain, d, total_chains)) + 0.3 * rng.standard_normal((1, d, total_chains))
    mine = draws[:, :, off:off + cnt]
    part = torch.tensor(_partial(mine.mean(0), mine.var(0, ddof=1)))
    all_reduce_sum(part)
    q.put((rank, off, cnt, rhat_from_partial(part, n_per_chain)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_rhat_equals_single_process():
    import torch.multiprocessing as mp

    from mcmc_amd.diagnostics import rhat_from_partial

    total, n, d, world = 37, 200, 3, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, n, d, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [(r[1], r[2]) for r in res] == [(0, 19), (19, 18)]  # contiguous, balanced, covers [0, 37)
    rng = np.random.default_rng(123)
    draws = rng.standard_normal((n, d, total)) + 0.3 * rng.standard_normal((1, d, total))
    want = rhat_from_partial(_partial(draws.mean(0), draws.var(0, ddof=1)), n)
    # textbook Gelman-Rubin on the pooled chains
    m, s2 = draws.mean(0), draws.var(0, ddof=1)
    W, B_over_n = s2.mean(1), m.var(1, ddof=1)
    np.testing.assert_allclose(want, np.sqrt(((n - 1) / n * W + B_over_n) / W), rtol=1e-12)
    for r in res:
        np.testing.assert_allclose(r[3], want, rtol=1e-12)
    assert np.all(want > 1.02)  # the injected between-chain offsets are detected


def test_shard_chains_properties():
    from mcmc_amd.sharding import shard_chains

    for total in (0, 1, 7, 64, 65536, 1_048_576 + 3):
        for world in (1, 2, 3, 8):
            parts = [shard_chains(total, r, world) for r in range(world)]
            assert parts[0][0] == 0 and sum(c for _, c in parts) == total
            for (o1, c1), (o2, _) in zip(parts, parts[1:]):
                assert o1 + c1 == o2
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1
    with pytest.raises(ValueError):
        shard_chains(10, 2, 2)


def test_pooled_covariance_from_partial_sums_of_two_shards():
    """pooled covariance from per-chain running moments; the partial sums of two shards add up
    (the contract of mcmc_pooled_cov_partial reproduced in numpy) to the covariance of all draws pooled."""
    from mcmc_amd.diagnostics import pooled_cov_from_partial

    rng = np.random.default_rng(7)
    n, d, total = 150, 4, 11
    A = rng.standard_normal((d, d))
    draws = np.einsum("kl,tlc->tkc", A, rng.standard_normal((n, d, total))) + rng.standard_normal((1, d, total))
    m = draws.mean(0)                                            # [d, chains]
    Cc = np.einsum("tkc,tlc->klc", draws - m, draws - m) / n     # population covariance per chain, as update_sigma keeps it

    def partial(sl):
        mm, cc = m[:, sl], Cc[:, :, sl]
        return np.concatenate([cc.sum(2).ravel(order="F"), np.einsum("kc,lc->kl", mm, mm).ravel(order="F"), mm.sum(1),
                               [mm.shape[1]]])

    mbar, S = pooled_cov_from_partial(partial(slice(0, 6)) + partial(slice(6, total)))
    flat = draws.transpose(1, 0, 2).reshape(d, -1)
    np.testing.assert_allclose(mbar, flat.mean(1), rtol=1e-12)
    np.testing.assert_allclose(S, np.cov(flat, bias=True), rtol=1e-11, atol=1e-13)


def _run_bench(args, env_extra=None, timeout=300):
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, capture_output=True, text=True,
                       timeout=timeout)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p.returncode, [json.loads(ln) for ln in lines], p.stderr


def test_bench_gpus_flag_starts_that_many_ranks():
    """`python bench.py --gpus 2` run plainly (no launcher) must start two ranks itself, not silently run one
    (the job of mclapply_RNG, MCMCutilty.R:103-149).  --rehearse skips the kernels so this runs without a GPU: the
    rank start-up, the barriers, the max-over-ranks clock and the all-reduce are the real ones (gloo)."""
    rc, lines, err = _run_bench(["--gpus", "2", "--rehearse", "--steps", "2", "--chains", "100"])
    assert rc == 0, err
    assert len(lines) == 1  # rank 0 alone prints
    assert lines[0]["n_gpus"] == 2 and lines[0]["collective_ranks"] == 2 and lines[0]["chain_offsets"] == [0, 100]
    # a launcher that set WORLD_SIZE to something else than --gpus is an error, not a one-GPU run labelled N
    rc, lines, err = _run_bench(["--gpus", "2", "--rehearse"], {"WORLD_SIZE": "1", "RANK": "0"})
    assert rc != 0 and not lines and "WORLD_SIZE=1" in err
    # under a launcher (WORLD_SIZE == --gpus) the process is one rank and starts nothing
    rc, lines, err = _run_bench(["--gpus", "1", "--rehearse", "--steps", "1"], {"WORLD_SIZE": "1", "RANK": "0"})
    assert rc == 0 and lines[0]["n_gpus"] == 1, err
