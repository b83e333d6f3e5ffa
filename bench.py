#!/usr/bin/env python3
"""Headline benchmark: chain-iterations/sec of the adaptive random-walk Metropolis hot path.

Workload at N=1 = BASELINE.json configs[1] (C2): adaptive RW Metropolis on a 10-D correlated
Gaussian (Sigma_ij = 0.9^|i-j|, given as precision), 65,536 chains x 1e4 iterations, FP64,
adapt_size_start 100, adapt_shape_start 500, covmat I, per-chain start ~ N(0, 4I) from
Philox(seed 1), full theta trace written to HBM (52.4 GB).  One "step" = one pass of the hot
path over that batch = ONE kernel launch advancing every chain by n_iterations.

For N>1 every rank runs the same per-GPU workload on its own chains (chain_offset = rank *
n_chains; weak scaling; no data-path collective -- chains are independent).  After the timed
region the ranks combine per-chain moments with ONE all-reduce (RCCL) into R-hat.

Prints one JSON line (see the driver contract) with `roofline` and `cpu_baseline` objects.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_VALU_PEAK_TFLOPS = 78.6  # MI355X FP64 vector peak: 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz (= FP64 matrix peak)
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)


def rwm_flops_per_iter(d, with_chol):
    """proposal d(d+1) + target 2d^2+3d + moments 1.5d(d+1)+3d (+ d^3/3 Cholesky)."""
    f = 4.5 * d * d + 8.5 * d
    return f + (d ** 3) / 3.0 if with_chol else f


def pmc_traffic_bytes(n_chains, n_iter, d):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/r1_c2_pmc.csv), for the
    default C2 shape only: WRITE_SIZE + 2 x FETCH_SIZE (KiB; the x2 is the gfx950 FETCH_SIZE correction of
    MI355X_MICROARCH.md).  bench.py cannot collect PMC counters itself, so other shapes report null."""
    path = os.path.join(ROOT, "profiles", "r1_c2_pmc.csv")
    if (n_chains, n_iter, d) != (65536, 10000, 10) or not os.path.exists(path):
        return None
    vals = {}
    for line in open(path):
        if line.startswith("#") or "," not in line:
            continue
        k, v = line.strip().split(",")
        if k != "counter":
            vals[k] = float(v)
    if "WRITE_SIZE" not in vals or "FETCH_SIZE" not in vals:
        return None
    return (vals["WRITE_SIZE"] + 2.0 * vals["FETCH_SIZE"]) * 1024.0


def c2_problem(d=10, rho=0.9):
    idx = np.arange(d)
    Sig = rho ** np.abs(idx[:, None] - idx[None, :])
    H = np.linalg.inv(Sig)
    return Sig, 0.5 * (H + H.T)


def cpu_baseline(d, n_iter, ass, ashs, budget_s=12.0):
    """The oracle (a port of the reference loop: fresh chol + 2 det + 2 inv per iteration, compiled
    C target instead of an R closure) on the host cores: one chain per thread, same config."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle import oracle as O

    O.build()
    _, H = c2_problem(d)
    tgt = O.gaussian_target(np.zeros(d), H)
    cores = max(1, min(16, os.cpu_count() or 1))
    rng = np.random.default_rng(1)

    def one(seed):
        init = 2.0 * rng.standard_normal(d)
        O.adaptMCMC_seeded(tgt, init, np.eye(d), n_iter, ass, 0.0, 0, ashs, float(seed))
        return n_iter

    t0 = time.perf_counter()
    one(0)
    t_one = time.perf_counter() - t0
    per_thread = max(1, int(budget_s / max(t_one, 1e-3)))
    per_thread = min(per_thread, 64)
    n_chains = cores * per_thread
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        done = sum(ex.map(one, range(1, n_chains + 1)))
    dt = time.perf_counter() - t0
    return {
        "value": done / dt, "unit": "chain-iterations/s", "cores": cores, "kind": "port",
        "sample": f"{n_chains} chains x {n_iter} iterations of the same config, one chain per thread "
                  f"(oracle/mcmc_oracle.c, libstdc++-style mt19937 streams)",
        "per_core": done / dt / cores,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--chains", type=int, default=65536, help="chains per GPU")
    ap.add_argument("--iters", type=int, default=10000)
    ap.add_argument("--d", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ess", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import mcmc_amd
    from mcmc_amd import diagnostics as diag

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    d, n_chains, n_iter = args.d, args.chains, args.iters
    ass, ashs = 100, 500
    Sig, H = c2_problem(d)
    target = mcmc_amd.Target.gaussian(np.zeros(d), H)
    chain_offset = rank * n_chains
    init = mcmc_amd.init_normal(1, n_chains, np.zeros(d), 2.0, chain_offset=chain_offset)
    out = {"theta_trace": torch.empty((n_iter, d, n_chains), dtype=torch.float64, device="cuda")}

    def step():
        return mcmc_amd.adapt_rwm_batch(target, n_chains, init, np.eye(d), n_iter, ass, 0.0, 0, ashs, seed=1,
                                        chain_offset=chain_offset, out=out)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in evs:
        a.record()
        res = step()
        b.record()
    fence()
    elapsed = time.perf_counter() - t0
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))  # HIP events on the launch stream
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- after the timed region: convergence diagnostics (one all-reduce of 3d+1 doubles)
    burn = min(n_iter // 2, 2000)
    mean, var = diag.chain_moments(res["theta_trace"], burn)
    rhat = diag.rhat(mean, var, n_iter - burn)
    status_bad = int((res["status"] != 0).sum().item())
    ess_per_s = None
    if not args.no_ess:
        sub = res["theta_trace"][:, :, : min(1024, n_chains)].contiguous()
        ess = diag.chain_ess(sub, burn, 500)
        ess_chain = ess.min(dim=0).values.mean().item()  # min over coordinates, mean over the sampled chains
        ess_per_s = ess_chain * n_chains * world / (elapsed / args.steps)
    acc_mean = float(res["acceptance_rate"].mean().item())

    if rank == 0:
        units = n_chains * n_iter  # chain-iterations per launch per GPU
        flops_launch = n_chains * (n_iter * rwm_flops_per_iter(d, False) + (n_iter - ass + 1) * (d ** 3) / 3.0)
        achieved_tf = flops_launch / (kern_ms * 1e-3) / 1e12
        bytes_launch = 8.0 * d * units
        line = {
            "metric": "chain-iterations/sec (adaptive RW Metropolis, whole job)",
            "value": units * world * args.steps / elapsed,
            "unit": "chain-iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"C2: adaptive RW Metropolis, {d}-D correlated Gaussian (rho=0.9, precision form), "
                                   f"{n_chains} chains/GPU x {n_iter} iterations, adapt 100/500, full trace to HBM",
                       "chains_per_gpu": n_chains, "iterations": n_iter, "d": d, "parallelism": f"chains sharded x{world}"},
            "roofline": {"bound": "fp64-valu", "achieved": achieved_tf, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved_tf / FP64_VALU_PEAK_TFLOPS, "traffic": pmc_traffic_bytes(n_chains, n_iter, d),
                         "algorithmic_bytes": bytes_launch,
                         "kernel": f"rwm_lane_kernel<{d},gaussian>", "kernel_ms": kern_ms,
                         "flops_per_chain_iteration": rwm_flops_per_iter(d, True),
                         "hbm_write_GBs": bytes_launch / (kern_ms * 1e-3) / 1e9,
                         "hbm_write_frac": bytes_launch / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "ess_per_sec": ess_per_s, "rhat_max": float(np.max(rhat)), "acceptance_rate_mean": acc_mean,
            "chains_failed": status_bad,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(d, n_iter, ass, ashs)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
