#!/usr/bin/env python3
"""Headline benchmark: chain-iterations/sec of the two sampling hot paths on MI355X.

Default workload = BASELINE.json configs[1] (C2): adaptive RW Metropolis on a 10-D correlated
Gaussian (Sigma_ij = 0.9^|i-j|, given as precision), 65,536 chains x 1e4 iterations, FP64,
adapt_size_start 100, adapt_shape_start 500, covmat I, per-chain start ~ N(0, 4I) from
Philox(seed 1), full theta trace written to HBM (52.4 GB).  One "step" = one pass of the hot
path over that batch = ONE kernel launch advancing every chain by n_iterations.

--workload selects the other single-GPU shapes of BASELINE.json / SURVEY.md section 8d:
  c3   TMVN Gibbs d=10, box bounds [-1,2], 1,048,576 chains x 2,000 sweeps, full trace (168 GB)
  c4   adaptive RW Metropolis d=100 AS SPECIFIED (per-GPU share of config 4: 131,072 chains x 5,000
       iterations, covmat I, adapt 200/1000, trace thinned x10).  On these inputs the reference's own size
       adaptation collapses: every chain ends in permanent NaN-reject around iteration 220 (DESIGN.md
       section 3.3); the line reports it (`nan_reject_fraction`, `frozen_chains`), R-hat is undefined (null)
  c4s  the same target and sizes with inputs that DO reach the covariance adaptation: covmat 0.05 I, size
       adaptation off, adapt_shape_start 1000 -- ~45 % of the iterations factor covmat_empirical
  c5   TMVN Gibbs d=128, lower = +4 conditional sd, upper = +Inf (tail path), per-GPU share of
       config 5: 1,048,576 chains x 200 sweeps, last state + running moments only
  r32  adaptive RW Metropolis d=32 on the one-chain-per-lane-row kernel (rwm_row_kernel, 12 < d <= 64): Sigma=AA'/d+I,
       65,536 chains x 3,000 iterations, covmat (0.3/d) I, size adaptation off, adapt_shape_start 180 (most
       iterations factor covmat_empirical), full trace (50 GB)
  r64  the same at d=64 (one chain per wavefront): 32,768 chains x 3,000 iterations, full trace (50 GB)

--gpus N: one process per GPU.  Under torch.distributed.run (WORLD_SIZE set) this process is one rank;
run plainly with N > 1 it starts N fresh child ranks of itself BEFORE anything touches the GPU (the job of
the reference's mclapply_RNG, MCMCutilty.R:103-149: one worker per chain set) and exits with their code.
Every rank runs the same per-GPU workload on its own chains (chain_offset = rank * n_chains; weak scaling;
no data-path collective -- chains are independent).  After the timed region the ranks combine per-chain
moments with ONE all-reduce (RCCL) into R-hat, and their ESS sums with another.

Prints one JSON line (see the driver contract) with `roofline` and `cpu_baseline` objects.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_VALU_PEAK_TFLOPS = 78.6  # MI355X FP64 vector peak: 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz (= FP64 matrix peak)
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
PROFILE_ROUNDS = ("r3", "r2", "r1")  # newest committed rocprofv3 PMC summary first


def rwm_flops_per_iter(d, with_chol):
    """proposal d(d+1) + target 2d^2+3d + moments 1.5d(d+1)+3d; + d^3/3 for the
    Cholesky factorization in shape-phase iterations (the size phase / rwMCMC have no Cholesky term)."""
    f = 4.5 * d * d + 8.5 * d
    return f + (d ** 3) / 3.0 if with_chol else f


def gibbs_flops_per_sweep(d):
    """d * [2(d-1) dot + 2 C_pnorm + C_qnorm + 9], C_pnorm = 50, C_qnorm = 35."""
    return d * (2.0 * d + 142.0)


def pmc_traffic_bytes(wl):
    """HBM bytes per launch REPLAYED from the committed rocprofv3 PMC passes of this same command
    (profiles/<round>_<workload>_pmc.csv, default shape only): WRITE_SIZE + 2 x FETCH_SIZE (KiB; the x2 is the
    gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md).  bench.py cannot collect PMC counters itself, so the
    value is NOT a measurement of this run; `traffic_source` in the line says where it comes from."""
    for rnd in PROFILE_ROUNDS:
        path = os.path.join(ROOT, "profiles", f"{rnd}_{wl}_pmc.csv")
        if not os.path.exists(path):
            continue
        vals = {}
        for line in open(path):
            if line.startswith("#") or "," not in line:
                continue
            k, v = line.strip().split(",")[:2]
            if k != "counter":
                try:
                    vals[k] = float(v)
                except ValueError:
                    pass
        if "WRITE_SIZE" in vals and "FETCH_SIZE" in vals:
            return (vals["WRITE_SIZE"] + 2.0 * vals["FETCH_SIZE"]) * 1024.0, f"profiles/{rnd}_{wl}_pmc.csv"
    return None, None


def ar1(d, rho):
    idx = np.arange(d)
    return rho ** np.abs(idx[:, None] - idx[None, :])


def sym_inv(S):
    H = np.linalg.inv(S)
    return 0.5 * (H + H.T)


def c4_sigma(d=100):
    A = np.random.default_rng(4).standard_normal((d, d))
    return A @ A.T / d + np.eye(d)


def _finite(x):
    """JSON has no Infinity / NaN: an undefined diagnostic is null."""
    x = float(x)
    return x if math.isfinite(x) else None


# --------------------------------------------------------------------------------------------
# CPU baselines: the oracle (a port of the reference loops) on the host cores, one chain per thread
def _threads():
    """Every core this job may use.  On the GPU box os.cpu_count() is the whole machine, the job's share is what
    the scheduler affinity and OMP_NUM_THREADS (16 there) say."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    cap = os.environ.get("OMP_NUM_THREADS")
    if cap and cap.isdigit() and int(cap) > 0:
        n = min(n, int(cap))
    return max(1, n)


def cpu_baseline_rwm(d, H, n_iter, ass, ashs, cov0, init_sd, budget_s=12.0):
    """adaptMCMC as the reference runs it: fresh chol + 2 det + 2 inv per iteration, compiled C target
    instead of an R closure (which favours the reference)."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle import oracle as O

    O.build()
    tgt = O.gaussian_target(np.zeros(d), H)
    cores = _threads()
    rng = np.random.default_rng(1)

    def one(seed):
        init = init_sd * rng.standard_normal(d)
        O.adaptMCMC_seeded(tgt, init, cov0, n_iter, ass, 0.0, 0, ashs, float(seed))
        return n_iter

    t0 = time.perf_counter()
    one(0)
    t_one = time.perf_counter() - t0
    per_thread = min(64, max(1, int(budget_s / max(t_one, 1e-3))))
    n_chains = cores * per_thread
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        done = sum(ex.map(one, range(1, n_chains + 1)))
    dt = time.perf_counter() - t0
    return {
        "value": done / dt, "unit": "chain-iterations/s", "cores": cores, "kind": "port",
        "sample": f"{n_chains} chains x {n_iter} iterations of the same config, one chain per thread on every core of "
                  f"this job's CPU share (oracle/mcmc_oracle.c, libstdc++-style mt19937 streams)",
        "per_core": done / dt / cores,
    }


def cpu_baseline_gibbs(mu, sigma, lower, upper, init, n_sweeps, budget_s=12.0):
    """rtmvnorm_gibbs as the reference runs it (d sub-block inverses up front, copy-and-shed per coordinate).
    The uniforms (the reference's runif(n*d), tmvrnormGibbs.cpp:97) are drawn before the clock starts."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle import oracle as O

    O.build()
    d = len(mu)
    cores = _threads()
    U = [np.random.default_rng(t).random(n_sweeps * d) for t in range(cores)]  # one stream per thread, reused

    def one(k):
        O.rtmvnorm_gibbs(n_sweeps, mu, sigma, lower, upper, init, U[k % cores])
        return n_sweeps

    t0 = time.perf_counter()
    one(0)
    t_one = time.perf_counter() - t0
    per_thread = min(64, max(1, int(budget_s / max(t_one, 1e-3))))
    n_chains = cores * per_thread
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        done = sum(ex.map(one, range(n_chains)))
    dt = time.perf_counter() - t0
    return {
        "value": done / dt, "unit": "chain-sweeps/s", "cores": cores, "kind": "port",
        "sample": f"{n_chains} chains x {n_sweeps} sweeps of the same config, one chain per thread on every core of "
                  f"this job's CPU share (oracle/mcmc_oracle.c; uniforms drawn before the timed region)",
        "per_core": done / dt / cores,
    }


# --------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n):
    """Parent of a plain `bench.py --gpus N` run: N fresh children of this script, one per GPU, with the
    torch.distributed environment set.  Nothing in this process has touched the GPU (torch is not even imported)."""
    port = os.environ.get("MASTER_PORT") or str(_free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        for p in procs:
            rc = max(rc, abs(p.wait()))
            if rc:
                break
    finally:
        for p in procs:  # a failed rank leaves its peers waiting in a collective: end exactly the children started here
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=30)
            except subprocess.TimeoutExpired:
                p.kill()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", choices=["c2", "c3", "c4", "c4s", "c5", "r32", "r64"], default="c2")
    ap.add_argument("--chains", type=int, default=None, help="chains per GPU (default: the workload's)")
    ap.add_argument("--iters", type=int, default=None, help="iterations / sweeps (default: the workload's)")
    ap.add_argument("--fast", action="store_true",
                    help="d > 12 Metropolis workloads: factor_update = MCMC_FACTOR_RANK1 (NOT the parity mode; closed-form rank-1 "
                         "update of the proposal factor, full factorization every --refresh iterations)")
    ap.add_argument("--refresh", type=int, default=64, help="refresh_every of --fast")
    ap.add_argument("--trace-layout", choices=["soa", "chain_major"], default=None,
                    help="default: chain_major where a chain owns a wavefront or workgroup (c4, c4s, r64), else soa")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ess", action="store_true")
    ap.add_argument("--rehearse", action="store_true",
                    help="no GPU, no kernels: only the rank start-up, the barriers and the all-reduces (gloo); for CPU tests")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")

    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))
    world = int(env_world or "1")
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                 f"(python -m torch.distributed.run --nproc-per-node {args.gpus} ... bench.py --gpus {args.gpus}), "
                 f"or run plainly and let --gpus start the ranks")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import torch
    import torch.distributed as dist

    f64 = torch.float64
    backend = os.environ.get("MCMC_BENCH_BACKEND", "gloo" if args.rehearse else "nccl")  # "gloo": N > 1 on a single GPU
    if args.rehearse:
        dev = torch.device("cpu")
    else:
        # one rank per GPU; the modulo only matters when more ranks than GPUs are rehearsed on one box
        local_rank %= max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")  # where the small all-reduced tensors live

    def all_sum(x):
        t = torch.tensor([float(x)], dtype=f64, device=coll_dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())

    wl = args.workload
    defaults = {"c2": (65536, 10000, 20, 3), "c3": (1 << 20, 2000, 5, 2), "c4": (131072, 5000, 1, 0),
                "c4s": (131072, 5000, 1, 0), "c5": (1 << 20, 200, 2, 1), "r32": (65536, 3000, 3, 1), "r64": (32768, 3000, 3, 1)}
    n_chains = args.chains or defaults[wl][0]
    n_iter = args.iters or defaults[wl][1]
    steps = args.steps if args.steps is not None else defaults[wl][2]
    warmup = args.warmup if args.warmup is not None else defaults[wl][3]
    chain_offset = rank * n_chains
    is_rwm = wl in ("c2", "c4", "c4s", "r32", "r64")

    if args.rehearse:
        def fence():
            if world > 1:
                dist.barrier()

        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            time.sleep(0.01)
        fence()
        elapsed = time.perf_counter() - t0
        t = torch.tensor([elapsed], dtype=f64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ranks = all_sum(1.0)
        if rank == 0:
            print(json.dumps({"metric": "rehearsal (no kernels)", "value": 0.0, "unit": "chain-iterations/s", "n_gpus": world,
                              "steps": steps, "warmup": warmup, "ms_per_step": float(t.item()) / max(1, steps) * 1e3,
                              "rehearsal": True, "collective_ranks": int(round(ranks)), "backend": backend,
                              "chain_offsets": [r * n_chains for r in range(world)]}), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    import mcmc_amd
    from mcmc_amd import diagnostics as diag

    if is_rwm:
        if wl == "c2":
            d, ass, ashs, thin, init_sd, c0 = 10, 100, 500, 1, 2.0, 1.0
            Sig = ar1(d, 0.9)
            desc = (f"C2: adaptive RW Metropolis, {d}-D correlated Gaussian (rho=0.9, precision form), {n_chains} chains/GPU x "
                    f"{n_iter} iterations, adapt 100/500, full trace to HBM")
        elif wl == "c4":
            d, ass, ashs, thin, init_sd, c0 = 100, 200, 1000, 10, 1.0, 1.0
            Sig = c4_sigma(d)
            desc = (f"C4 (per-GPU share, as specified): adaptive RW Metropolis, {d}-D Gaussian Sigma=AA'/d+I, {n_chains} chains/GPU x "
                    f"{n_iter} iterations, covmat I, adapt 200/1000, trace thinned x10 -- the reference's size adaptation ends in "
                    f"permanent NaN-reject on these inputs (see nan_reject_fraction)")
        elif wl in ("r32", "r64"):
            d = 32 if wl == "r32" else 64
            ass, ashs, thin, init_sd, c0 = 0, 180, 1, 1.0, 0.3 / d
            Sig = c4_sigma(d)
            desc = (f"{wl.upper()}: adaptive RW Metropolis on the lane-row kernel, {d}-D Gaussian Sigma=AA'/d+I, {n_chains} chains/GPU x "
                    f"{n_iter} iterations, covmat (0.3/d) I, size adaptation off, adapt_shape_start 180, full trace to HBM")
        else:
            d, ass, ashs, thin, init_sd, c0 = 100, 0, 1000, 10, 1.0, 0.05
            Sig = c4_sigma(d)
            desc = (f"C4S (per-GPU share, non-degenerate inputs): adaptive RW Metropolis, {d}-D Gaussian Sigma=AA'/d+I, {n_chains} "
                    f"chains/GPU x {n_iter} iterations, covmat 0.05 I, size adaptation off, adapt_shape_start 1000, trace thinned x10")
        H = sym_inv(Sig)
        cov0 = c0 * np.eye(d)
        target = mcmc_amd.Target.gaussian(np.zeros(d), H)
        init = mcmc_amd.init_normal(1, n_chains, np.zeros(d), init_sd, chain_offset=chain_offset)
        if args.fast and d <= 12:
            ap.error("--fast is served by the per-workgroup kernel (d > 12)")
        layout = args.trace_layout or ("chain_major" if d > 32 else "soa")
        if layout == "chain_major" and d <= 12:
            ap.error("--trace-layout chain_major needs d > 12")
        # final_theta / final_mean / final_cov are not part of the timed path (nothing below reads them); at d = 100 the
        # [d*d][n_chains] final covariances alone would be 10.5 GB of 8-byte-per-line stores from one-chain workgroups
        rwm_kw = dict(trace_layout=layout, final=d <= 12)
        if args.fast:
            rwm_kw.update(factor_update="rank1", refresh_every=args.refresh)
            desc += f"; NON-PARITY fast mode: rank-1 factor updates, refactorization every {args.refresh} iterations"
        desc += f"; trace layout {layout}"
        shape = (n_chains, n_iter // thin, d) if layout == "chain_major" else (n_iter // thin, d, n_chains)
        out = {"theta_trace": torch.empty(shape, dtype=f64, device="cuda")}

        def step():
            return mcmc_amd.adapt_rwm_batch(target, n_chains, init, cov0, n_iter, ass, 0.0, 0, ashs, seed=1, thin=thin,
                                            chain_offset=chain_offset, out=out, **rwm_kw)
        trace_key = "theta_trace"
    else:
        if wl == "c3":
            d, thin, tail_safe, want_trace = 10, 1, False, True
            sigma = ar1(d, 0.5)
            lower, upper, init_state = np.full(d, -1.0), np.full(d, 2.0), np.zeros(d)
            desc = (f"C3: TMVN Gibbs d={d}, Sigma_ij=0.5^|i-j|, box [-1,2], {n_chains} chains/GPU x {n_iter} sweeps, full trace to HBM")
        else:
            d, thin, tail_safe, want_trace = 128, 1, True, False
            sigma = ar1(d, 0.3)
            _, sd_c = mcmc_amd.tmvn_precompute(sigma)
            lower, upper = 4.0 * np.asarray(sd_c), np.full(d, np.inf)
            init_state = lower + 0.1
            desc = (f"C5 (per-GPU share): TMVN Gibbs d={d}, Sigma_ij=0.3^|i-j|, lower=+4 conditional sd, upper=+Inf (tail path), "
                    f"{n_chains} chains/GPU x {n_iter} sweeps, last state + running moments")
        mu = np.zeros(d)
        out = {}
        if want_trace:
            out["trace"] = torch.empty((n_iter // thin, d, n_chains), dtype=f64, device="cuda")

        def step():
            return mcmc_amd.tmvn_gibbs_batch(n_chains, n_iter, mu, sigma, lower, upper, init_state, seed=2 if wl == "c3" else 5,
                                             thin=thin, burn=n_iter // 2, chain_offset=chain_offset, tail_safe=tail_safe,
                                             trace=want_trace, moments=not want_trace, out=out)
        trace_key = "trace"

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    res = None
    for _ in range(warmup):
        res = step()
    fence()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for a, b in evs:
        a.record()  # on torch's current stream, the one adapt_rwm_batch / tmvn_gibbs_batch launch on
        res = step()
        b.record()
    fence()
    elapsed = time.perf_counter() - t0
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))  # HIP events on the launch stream
    if world > 1:
        t = torch.tensor([elapsed], dtype=f64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- after the timed region: convergence diagnostics (one all-reduce of 3d+1 doubles for R-hat, one scalar for ESS)
    extra = {}
    status_bad = int(all_sum((res["status"] != 0).sum().item()))
    if is_rwm and layout == "chain_major":  # the diagnostics kernels read [n_kept][d][n_chains]
        res = dict(res)
        res[trace_key] = res[trace_key].permute(1, 2, 0).contiguous()
        out.clear()
    if trace_key in res:
        n_kept = res[trace_key].shape[0]
        burn = min(n_kept // 2, 2000)
        mean, var = diag.chain_moments(res[trace_key], burn)
        # a chain that never moved after burn-in (C4: permanent NaN-reject) has within-chain variance 0 in every coordinate
        frozen = int(all_sum((var.max(dim=0).values == 0).sum().item()))
        extra["frozen_chains"] = frozen
        if backend == "nccl" or world == 1:
            rh = diag.rhat(mean, var, n_kept - burn)
        else:  # gloo rehearsal on one GPU: the partial sums cross through host memory
            part = diag.rhat_partial(mean, var).cpu()
            dist.all_reduce(part, op=dist.ReduceOp.SUM)
            rh = diag.rhat_from_partial(part, n_kept - burn)
        with np.errstate(all="ignore"):
            extra["rhat_max"] = _finite(np.max(rh))  # null when the within-chain variance is 0 (all chains frozen)
        if not args.no_ess:
            n_s = min(1024, n_chains)
            sub = res[trace_key][:, :, :n_s].contiguous()
            # Geyer's sum is cut at max_lag; a chain whose sum was still positive there has its ESS OVER-stated
            # (tests/test_gpu_diagnostics.py).  Raise the cap until under 1 % of the sampled chains hit it (at most
            # half the series), and say what is left: ess_per_sec counts those chains as the kernel reports them (an upper
            # bracket), ess_per_sec_lower counts them as 0.
            lag_limit = max(1, (n_kept - burn) // 2)
            max_lag = min(500, lag_limit)
            while True:
                ess, capped = diag.chain_ess(sub, burn, max_lag, return_capped=True)
                chain_capped = capped.max(dim=0).values > 0  # any coordinate of the chain
                n_capped = all_sum(chain_capped.sum().item())
                if n_capped <= 0.01 * n_s * world or max_lag >= lag_limit:
                    break
                max_lag = min(2 * max_lag, lag_limit)
            ess_chain = ess.min(dim=0).values  # min over coordinates, per sampled chain
            ess_chain = torch.nan_to_num(ess_chain, nan=0.0, posinf=0.0, neginf=0.0)  # a frozen chain carries no information
            scale = n_chains / n_s
            ess_total = all_sum(ess_chain.sum().item() * scale)  # this rank's chains, then the sum over ranks
            ess_lower = all_sum((ess_chain * (~chain_capped)).sum().item() * scale)
            extra["ess_per_sec"] = _finite(ess_total / (elapsed / steps))
            extra["ess_per_sec_lower"] = _finite(ess_lower / (elapsed / steps))
            extra["ess_max_lag"] = max_lag
            extra["ess_lag_capped_fraction"] = n_capped / (n_s * world)
            extra["ess_sample"] = f"{n_s} chains per rank, {n_kept - burn} kept draws after burn-in, min over coordinates"
    else:
        n_mom = n_iter - n_iter // 2
        mean, var = res["mean"], res["m2"] / max(1, n_mom - 1)
        if backend == "nccl" or world == 1:
            rh = diag.rhat(mean, var, n_mom)
        else:
            part = diag.rhat_partial(mean, var).cpu()
            dist.all_reduce(part, op=dist.ReduceOp.SUM)
            rh = diag.rhat_from_partial(part, n_mom)
        with np.errstate(all="ignore"):
            extra["rhat_max"] = _finite(np.max(rh))
    shape_iters = 0.0
    if is_rwm:
        extra["acceptance_rate_mean"] = all_sum(res["acceptance_rate"].sum().item()) / (n_chains * world)
        shape_iters = float(res["shape_iterations"].to(f64).sum().item())  # this rank's launch (roofline is per launch)
        extra["shape_phase_fraction"] = all_sum(shape_iters) / (n_chains * world * n_iter)
        # how many proposals died in the NaN path of mvrnorm_pdf (adaptMCMC_source.cpp:56,169-175): a sample of this
        # rank's first chains re-run with per-iteration diagnostics (same seed and chain ids)
        ns = min(256, n_chains)
        del res[trace_key]  # room for the diagnostics arrays
        dg = mcmc_amd.adapt_rwm_batch(target, ns, init[:, :ns].contiguous(), cov0, n_iter, ass, 0.0, 0, ashs, seed=1, thin=thin,
                                      chain_offset=chain_offset, trace=False, final=False, diagnostics=True)
        extra["nan_reject_fraction"] = float(torch.isnan(dg["log_acceptance"]).double().mean().item())
        extra["nan_reject_sample"] = f"{ns} chains of rank 0 re-run with per-iteration diagnostics"
        del dg
    extra["chains_failed"] = status_bad
    ranks = int(round(all_sum(1.0)))

    if rank == 0:
        units = n_chains * n_iter  # chain-iterations (chain-sweeps) per launch per GPU
        if is_rwm:
            if args.fast:
                # what the rank-1 mode has to do: per shape-phase iteration the factor update (per entry of the triangle
                # g w + u s and s += p w: 5 flops, 2.5 d^2) instead of d^3/3, plus one factorization per chain when the
                # shape phase starts and one every `refresh` shape-phase iterations
                flops_launch = (units * rwm_flops_per_iter(d, False) + shape_iters * 2.5 * d * d
                                + (shape_iters / args.refresh + n_chains) * (d ** 3) / 3.0)
            else:
                flops_launch = units * rwm_flops_per_iter(d, False) + shape_iters * (d ** 3) / 3.0
            kept = n_iter // thin
            kernel = f"rwm_pair_kernel<{d},gaussian>" if d <= 10 else ("rwm_row_kernel" if d <= 64 and not args.fast else "rwm_block_kernel")
            per_unit = {"flops_size_phase": rwm_flops_per_iter(d, False), "flops_shape_phase": rwm_flops_per_iter(d, True)}
            unit = "chain-iterations/s"
            metric = "chain-iterations/sec (adaptive RW Metropolis, whole job)"
        else:
            flops_launch = units * gibbs_flops_per_sweep(d)
            kept = n_iter // thin if want_trace else 0
            kernel = "gibbs_kernel" if d <= 27 else "gibbs_reg_kernel"
            per_unit = {"flops_per_chain_sweep": gibbs_flops_per_sweep(d)}
            unit = "chain-sweeps/s"
            metric = "chain-iterations/sec (TMVN Gibbs sweeps, whole job)"
        bytes_launch = 8.0 * d * n_chains * kept
        if not is_rwm and not want_trace:
            # moments-only run: last state, mean and M2 leave the kernel once; the running sums themselves live in
            # HBM (16 B read + 16 B written per chain, coordinate and counted sweep, mostly absorbed by L2 / MALL)
            bytes_launch = 24.0 * d * n_chains
        achieved_tf = flops_launch / (kern_ms * 1e-3) / 1e12
        prof_tag = wl + ("_fast" if args.fast else "")
        traffic, traffic_src = pmc_traffic_bytes(prof_tag) if (n_chains, n_iter) == defaults[wl][:2] else (None, None)
        line = {
            "metric": metric,
            "value": units * world * steps / elapsed,
            "unit": unit,
            "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": elapsed / steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": desc, "chains_per_gpu": n_chains, "iterations": n_iter, "d": d,
                       "parallelism": f"chains sharded x{world}"},
            "collective_ranks": ranks, "backend": backend if world > 1 else None,
            "roofline": {"bound": "fp64-valu", "achieved": achieved_tf, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved_tf / FP64_VALU_PEAK_TFLOPS,
                         "traffic": traffic,
                         "traffic_source": (f"{traffic_src}: rocprofv3 --pmc passes of this command, replayed -- not collected in this run"
                                            if traffic_src else None),
                         "algorithmic_bytes": bytes_launch, "algorithmic_flops": flops_launch,
                         "kernel": kernel, "kernel_ms": kern_ms, **per_unit,
                         "hbm_write_GBs": bytes_launch / (kern_ms * 1e-3) / 1e9,
                         "hbm_write_frac": bytes_launch / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
            **extra,
        }
        if world == 1 and not args.no_cpu_baseline:
            if is_rwm:
                line["cpu_baseline"] = cpu_baseline_rwm(d, H, min(n_iter, 10000 if d <= 10 else 300), ass, ashs, cov0, init_sd)
            else:
                line["cpu_baseline"] = cpu_baseline_gibbs(mu, sigma, lower, upper, init_state, min(n_iter, 2000 if d <= 10 else 50))
        print(json.dumps(line, allow_nan=False), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
