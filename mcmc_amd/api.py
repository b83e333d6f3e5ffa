"""Host-side mirror of the reference's Rcpp interface, over the C ABI.

Single-chain functions keep the reference's names, argument order, defaults and return
layout (adaptMCMC_source.cpp:74-76,229-230; tmvrnormGibbs.cpp:11-37,48-49,90-91):

    adaptMCMC, rwMCMC, rtmvnorm_gibbs, rtnorm_gibbs, sub1, sub2, negSubCol, negSubRow

The batch functions (adapt_rwm_batch, tmvn_gibbs_batch, ...) are the many-chain form; their
arrays are torch CUDA tensors (device memory + stream plumbing only -- all arithmetic is in
the HIP library).  Nothing here computes samples on the CPU.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _lib as L
from ._lib import check, dp, lib


# --------------------------------------------------------------------------- targets
@dataclass
class Target:
    """Device log-density: the stand-in for the R closure `target`
    (adaptMCMC_source.cpp:75,97,158)."""

    kind: int
    params: np.ndarray
    d: Optional[int] = None
    _user: Optional["_UserHandle"] = None

    @staticmethod
    def gaussian(mean, precision) -> "Target":
        """log N(mean, precision^-1) up to a constant: -0.5 (x-m)^T H (x-m)."""
        mean = np.ascontiguousarray(mean, dtype=np.float64).reshape(-1)
        H = np.asarray(precision, dtype=np.float64)
        if H.shape != (mean.size, mean.size):
            raise ValueError("precision must be d x d")
        return Target(L.MCMC_TARGET_GAUSSIAN_PREC, np.concatenate([mean, H.ravel(order="F")]), mean.size)

    @staticmethod
    def goldstein_price() -> "Target":
        """adaptMCMC_analysis.R:115-129"""
        return Target(L.MCMC_TARGET_GOLDSTEIN_PRICE, np.zeros(0), 2)

    @staticmethod
    def banana(B: float = 0.03) -> "Target":
        """adaptMCMC_analysis.R:19-22"""
        return Target(L.MCMC_TARGET_BANANA, np.array([float(B)]), 2)

    @staticmethod
    def compile(source: str, d: int, params=()) -> "Target":
        """A user log-density as HIP device source (mcmc_target_compile): the compiled-code
        replacement for an arbitrary R closure `target` (adaptMCMC_source.cpp:75,97,158).

        `source` defines
            __device__ double mcmc_user_target(const double (&x)[MCMC_USER_D], const double *params)
        and is compiled with hipRTC into the Metropolis kernels (d <= 12).  Raises McmcError with the
        compiler's diagnostics if it does not compile."""
        prm = np.ascontiguousarray(params, dtype=np.float64).reshape(-1)
        handle = C.c_void_p()
        log = C.create_string_buffer(1 << 16)
        rc = lib.mcmc_target_compile(source.encode("utf-8"), int(d), int(prm.size), C.byref(handle), log, len(log))
        if rc != L.MCMC_OK:
            raise L.McmcError(rc, lib.mcmc_amd_last_error().decode("utf-8", "replace") + "\n"
                              + log.value.decode("utf-8", "replace"))
        return Target(L.MCMC_TARGET_USER, prm, int(d), _UserHandle(handle.value))

    def with_params(self, params) -> "Target":
        """The same compiled user target with other parameter values (same count)."""
        prm = np.ascontiguousarray(params, dtype=np.float64).reshape(-1)
        if prm.size != np.asarray(self.params).size:
            raise ValueError("the number of parameters is fixed when the target is compiled")
        return Target(self.kind, prm, self.d, self._user)

    def _c(self):
        prm = np.ascontiguousarray(self.params, dtype=np.float64)
        t = L.McmcTarget(self.kind, prm.size, prm.ctypes.data_as(dp) if prm.size else None,
                         self._user.value if self._user is not None else None)
        t._keep = (prm, self._user)
        return t


class _UserHandle:
    """Owns a mcmc_target_compile handle."""

    def __init__(self, value):
        self.value = value

    def __del__(self):
        if getattr(self, "value", None):
            lib.mcmc_target_destroy(self.value)
            self.value = None


def _need_target(target) -> Target:
    if isinstance(target, Target):
        return target
    if callable(target):
        raise TypeError(
            "target is a host callable; a GPU cannot call back into the interpreter "
            "(adaptMCMC_source.cpp:97,158). Pass a mcmc_amd.Target descriptor "
            "(Target.gaussian / goldstein_price / banana), or the function as HIP source "
            "through Target.compile."
        )
    raise TypeError("target must be a mcmc_amd.Target")


def _vec(a):
    return np.ascontiguousarray(a, dtype=np.float64).reshape(-1)


def _colmajor(a, d):
    a = np.asarray(a, dtype=np.float64)
    if a.shape != (d, d):
        raise ValueError(f"expected a {d} x {d} matrix, got {a.shape}")
    return np.ascontiguousarray(a.ravel(order="F"))


def _p(a):
    return None if a is None else a.ctypes.data_as(dp)


# --------------------------------------------------------------------------- single chain (Rcpp drop-in)
def adaptMCMC(target, init_theta, covmat, n_iterations, adapt_size_start, acceptance_rate_weight,
              acceptance_window, adapt_shape_start, info, seedMH, adapt_size_cooling=0.99,
              max_scaling_sd=50.0, *, Z=None, U=None, store_sigma=True, info_log=False):
    """adaptMCMC (adaptMCMC_source.cpp:75-219) on the GPU, one chain.

    Returns the reference's list as a dict: theta_trace (n x d), sigma_empirical (d x d x n),
    acceptance_rate.  Z (n x d) / U (n) inject the random streams in the reference's draw
    order; without them Philox4x32-10 keyed by seedMH is used.
    """
    t = _need_target(target)
    init = _vec(init_theta)
    d = init.size
    cov = _colmajor(covmat, d)
    n = int(n_iterations)
    zz = uu = None
    if Z is not None or U is not None:
        zz = np.ascontiguousarray(Z, dtype=np.float64).reshape(-1)
        uu = _vec(U)
        if zz.size < n * d or uu.size < n:
            raise ValueError("Z must hold n*d normals and U n uniforms")
    trace = np.empty(n * d)
    sig = np.empty(d * d * n) if store_sigma else None
    acc = C.c_double(0.0)
    fail = C.c_int32(0)
    n_info = n // int(info) if int(info) > 0 else 0
    ilog = np.zeros(3 * max(n_info, 1)) if info_log else None
    ct = t._c()
    check(lib.mcmc_adaptMCMC(C.byref(ct), d, _p(init), _p(cov), n, int(adapt_size_start),
                             float(acceptance_rate_weight), int(acceptance_window), int(adapt_shape_start),
                             int(info), float(seedMH), float(adapt_size_cooling), float(max_scaling_sd),
                             _p(zz), _p(uu), _p(trace), _p(sig), C.byref(acc), _p(ilog), C.byref(fail)))
    out = dict(theta_trace=trace.reshape(n, d, order="F"),
               sigma_empirical=None if sig is None else sig.reshape(d, d, n, order="F"),
               acceptance_rate=acc.value)
    if info_log:
        out["info_log"] = ilog[: 3 * n_info].reshape(n_info, 3)
    return out


def rwMCMC(target, init_theta, covmat, n_iterations, info, seedMH, *, Z=None, U=None):
    """rwMCMC (adaptMCMC_source.cpp:230-316): returns theta_trace (n x d), acceptance_rate."""
    t = _need_target(target)
    init = _vec(init_theta)
    d = init.size
    cov = _colmajor(covmat, d)
    n = int(n_iterations)
    zz = uu = None
    if Z is not None or U is not None:
        zz = np.ascontiguousarray(Z, dtype=np.float64).reshape(-1)
        uu = _vec(U)
    trace = np.empty(n * d)
    acc = C.c_double(0.0)
    fail = C.c_int32(0)
    ct = t._c()
    check(lib.mcmc_rwMCMC(C.byref(ct), d, _p(init), _p(cov), n, int(info), float(seedMH), _p(zz), _p(uu),
                          _p(trace), C.byref(acc), C.byref(fail)))
    return dict(theta_trace=trace.reshape(n, d, order="F"), acceptance_rate=acc.value)


def rtmvnorm_gibbs(n, mu, sigma, lower, upper, init_state, *, U=None, seed=0):
    """rtmvnorm_gibbs (tmvrnormGibbs.cpp:91-142): n x d trace. U = the runif(n*d) draws."""
    mu, lower, upper, init = _vec(mu), _vec(lower), _vec(upper), _vec(init_state)
    d = mu.size
    if not (lower.size == upper.size == init.size == d):
        raise ValueError("mu, lower, upper, init_state must have the same length")
    sg = _colmajor(sigma, d)
    n = int(n)
    uu = None
    if U is not None:
        uu = _vec(U)
        if uu.size < n * d:
            raise ValueError("U must hold n*d uniforms")
    trace = np.empty(n * d)
    check(lib.mcmc_rtmvnorm_gibbs(n, d, _p(mu), _p(sg), _p(lower), _p(upper), _p(init), _p(uu),
                                  int(seed), _p(trace)))
    return trace.reshape(n, d, order="F")


def rtnorm_gibbs(n, mu, sigma, a, b, *, F=None, seed=0):
    """rtnorm_gibbs (tmvrnormGibbs.cpp:49-70). F = the runif(n) draws."""
    n = int(n)
    ff = None
    if F is not None:
        ff = _vec(F)
        if ff.size < n:
            raise ValueError("F must hold n uniforms")
    out = np.empty(n)
    check(lib.mcmc_rtnorm_gibbs(n, float(mu), float(sigma), float(a), float(b), _p(ff), int(seed), _p(out)))
    return out


def _init_matrix(X, d_expected=None):
    """list / array of per-chain start vectors (R: X[[c]]) -> (d x n_chains column-major flat, d, n_chains)"""
    X = np.asarray(X, dtype=np.float64)
    if X.ndim != 2:
        raise ValueError("X must be a sequence of equally long start vectors (one per chain)")
    n_chains, d = X.shape
    if d_expected is not None and d != d_expected:
        raise ValueError(f"start vectors have length {d}, expected {d_expected}")
    return np.ascontiguousarray(X).reshape(-1), d, n_chains  # row c of X = column c of the d x n_chains matrix


def adaptMCMC_chains(target, X, covmat, n_iterations, adapt_size_start, acceptance_rate_weight, acceptance_window,
                     adapt_shape_start, info, seedMH, adapt_size_cooling=0.99, max_scaling_sd=50.0, *, thin=1,
                     covmat_empirical=False):
    """`mclapply_RNG(X, function(x) adaptMCMC(target, x, covmat, ...))` (MCMCutilty.R:103-149 around
    adaptMCMC_source.cpp:75-219) as ONE host-pointer call (mcmc_adaptMCMC_chains): X is the list of per-chain
    init_theta.  Returns the reference's list of chains -- element c is a dict with `trace` (n_kept x d, what
    melt_trace reads, MCMCutilty.R:19-35), `acceptance_rate`, `status` (and `covmat_empirical` on request)."""
    t = _need_target(target)
    init, d, n_chains = _init_matrix(X)
    cov = _colmajor(covmat, d)
    n, thin = int(n_iterations), int(thin)
    n_kept = n // thin if thin > 0 else 0
    trace = np.empty(n_chains * n_kept * d)
    acc = np.empty(n_chains)
    status = np.zeros(n_chains, dtype=np.int32)
    cemp = np.empty(n_chains * d * d) if covmat_empirical else None
    ct = t._c()
    check(lib.mcmc_adaptMCMC_chains(C.byref(ct), d, n_chains, _p(init), 1, _p(cov), n, int(adapt_size_start),
                                    float(acceptance_rate_weight), int(acceptance_window), int(adapt_shape_start),
                                    int(info), float(seedMH), float(adapt_size_cooling), float(max_scaling_sd), thin,
                                    _p(trace), _p(acc), _p(cemp), status.ctypes.data_as(L.ip)))
    out = []
    for c in range(n_chains):
        item = dict(trace=trace[c * n_kept * d:(c + 1) * n_kept * d].reshape(n_kept, d, order="F"),
                    acceptance_rate=float(acc[c]), status=int(status[c]))
        if covmat_empirical:
            item["covmat_empirical"] = cemp[c * d * d:(c + 1) * d * d].reshape(d, d, order="F")
        out.append(item)
    return out


def rtmvnorm_gibbs_chains(n, mu, sigma, lower, upper, X, *, seed=0, thin=1, tail_safe=False):
    """`mclapply_RNG(X, function(x) rtmvnorm_gibbs(n, mu, sigma, lower, upper, x))` (MCMCutilty.R:103-149 around
    tmvrnormGibbs.cpp:91-142) as one host-pointer call (mcmc_rtmvnorm_gibbs_chains): list of {trace, status}."""
    mu, lower, upper = _vec(mu), _vec(lower), _vec(upper)
    d = mu.size
    init, _, n_chains = _init_matrix(X, d)
    sg = _colmajor(sigma, d)
    n, thin = int(n), int(thin)
    n_kept = n // thin if thin > 0 else 0
    trace = np.empty(n_chains * n_kept * d)
    status = np.zeros(n_chains, dtype=np.int32)
    check(lib.mcmc_rtmvnorm_gibbs_chains(n, d, n_chains, _p(mu), _p(sg), _p(lower), _p(upper), _p(init), 1, int(seed), thin,
                                         int(bool(tail_safe)), _p(trace), status.ctypes.data_as(L.ip)))
    return [dict(trace=trace[c * n_kept * d:(c + 1) * n_kept * d].reshape(n_kept, d, order="F"), status=int(status[c]))
            for c in range(n_chains)]


def sub1(x, i):
    """x[-i,-i] (tmvrnormGibbs.cpp:12-16), 0-based i."""
    x = np.asarray(x, dtype=np.float64)
    d = x.shape[0]
    buf = _colmajor(x, d)
    out = np.empty((d - 1) * (d - 1))
    check(lib.mcmc_sub1(d, _p(buf), int(i), _p(out)))
    return out.reshape(d - 1, d - 1, order="F")


def sub2(x, a, b):
    """x[a,-b] as a 1 x (nc-1) matrix (tmvrnormGibbs.cpp:20-23), 0-based."""
    x = np.asarray(x, dtype=np.float64)
    nr, nc = x.shape
    buf = np.ascontiguousarray(x.ravel(order="F"))
    out = np.empty(nc - 1)
    check(lib.mcmc_sub2(nr, nc, _p(buf), int(a), int(b), _p(out)))
    return out.reshape(1, nc - 1)


def negSubCol(x, i):
    """x[-i] (tmvrnormGibbs.cpp:27-30)."""
    x = _vec(x)
    out = np.empty(x.size - 1)
    check(lib.mcmc_negSubCol(x.size, _p(x), int(i), _p(out)))
    return out


def negSubRow(x, i):
    """x[-i] (tmvrnormGibbs.cpp:34-37)."""
    x = _vec(x)
    out = np.empty(x.size - 1)
    check(lib.mcmc_negSubRow(x.size, _p(x), int(i), _p(out)))
    return out


def tmvn_precompute(sigma):
    """Regression rows P (d x (d-1)) and conditional sd (d) of tmvrnormGibbs.cpp:100-113."""
    sigma = np.asarray(sigma, dtype=np.float64)
    d = sigma.shape[0]
    sg = _colmajor(sigma, d)
    P = np.zeros(max(d * (d - 1), 1))
    sd = np.zeros(d)
    check(lib.mcmc_tmvn_precompute(d, _p(sg), _p(P), _p(sd)))
    return P[: d * (d - 1)].reshape(d, d - 1), sd


# --------------------------------------------------------------------------- batch (device tensors)
def _torch():
    import torch

    return torch


def _dev_ptr(t, dtype=None):
    if t is None:
        return None
    torch = _torch()
    if not t.is_cuda or not t.is_contiguous():
        raise ValueError("batch arrays must be contiguous CUDA tensors")
    if dtype is not None and t.dtype != dtype:
        raise ValueError(f"expected dtype {dtype}, got {t.dtype}")
    return C.c_void_p(t.data_ptr())


def _stream_ptr():
    torch = _torch()
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def adapt_rwm_batch(target, n_chains, init_theta, covmat, n_iterations, adapt_size_start=0,
                    acceptance_rate_weight=0.0, acceptance_window=0, adapt_shape_start=0, *, seed=0,
                    adapt_size_cooling=0.99, max_scaling_sd=50.0, thin=1, chain_offset=0, Z=None, U=None,
                    trace=True, sigma_trace=False, final=True, diagnostics=False, out=None, device="cuda",
                    kernel="auto", checkpoint=False, resume=None, factor_update="exact", refresh_every=64,
                    trace_layout="soa"):
    """Many independent adaptMCMC chains in one launch (adaptMCMC_source.cpp:75-219 per chain).

    factor_update="rank1" (NOT a parity mode; per-workgroup kernel, d > 12): in the shape phase the proposal factor
    follows covmat_empirical by closed-form rank-1 updates and is refactored every `refresh_every` iterations, instead
    of the reference's fresh chol every iteration (:153).  trace_layout="chain_major" stores theta_trace as
    [n_chains, n_kept, d] (a kept row of a chain contiguous; row / per-workgroup kernels, d > 12).

    init_theta: length-d vector shared by all chains, or a [d, n_chains] CUDA tensor.
    Z [n_iterations, d, n_chains], U [n_iterations, n_chains]: injected streams (CUDA, f64).
    Returns a dict of CUDA tensors: theta_trace [n_kept, d, n_chains], acceptance_rate
    [n_chains], status [n_chains], shape_iterations [n_chains] (iterations proposed from the
    shape-adapted covariance), final_theta / final_mean [d, n_chains], final_cov
    [d*d, n_chains], and with diagnostics=True accepted / log_acceptance / log_A /
    scaling_sd / scaling_multiplier / rate_trace, each [n_iterations, n_chains].
    `out` may carry preallocated tensors under the same keys.  kernel = "auto" | "lane" | "pair" |
    "block" picks the device kernel (auto: per lane for d <= 12 -- the paired-wave variant for Philox runs
    without diagnostics at d <= 10 -- and per 256-thread workgroup above).

    checkpoint=True adds out["state"] = RwmState(array [state_doubles, n_chains], iterations done);
    resume=<that RwmState> continues the run at its next iteration with the same arguments (bit-identical
    to one uninterrupted call; trace rows are kept at absolute multiples of `thin`).  RwmState.save / .load
    move it through a .npz file.
    """
    torch = _torch()
    t = _need_target(target)
    n_chains = int(n_chains)
    n = int(n_iterations)
    out = dict(out or {})
    f64 = torch.float64
    if torch.is_tensor(init_theta):
        d = init_theta.shape[0]
        if tuple(init_theta.shape) != (d, n_chains):
            raise ValueError("init tensor must be [d, n_chains]")
        d_init, h_init = init_theta, None
    else:
        h_init = _vec(init_theta)
        d = h_init.size
        d_init = None
    cov = _colmajor(covmat, d)
    i0 = 0
    if resume is not None:
        i0 = int(resume.iterations)
        if tuple(resume.array.shape) != (_state_rows(d, acceptance_window, acceptance_rate_weight), n_chains):
            raise ValueError("resume state has the wrong shape for this d / n_chains / acceptance_window")
    n_kept = (i0 + n) // int(thin) - i0 // int(thin)

    def buf(key, shape, dtype=f64, want=True):
        if not want:
            return None
        if key not in out:
            out[key] = torch.empty(shape, dtype=dtype, device=device)
        elif tuple(out[key].shape) != tuple(shape):
            raise ValueError(f"out[{key!r}] has shape {tuple(out[key].shape)}, expected {tuple(shape)}")
        return out[key]

    cfg = L.RwmConfig(d=d, n_iterations=n, n_chains=n_chains, chain_offset=int(chain_offset),
                      adapt_size_start=int(adapt_size_start), adapt_shape_start=int(adapt_shape_start),
                      acceptance_window=int(acceptance_window), thin=int(thin),
                      acceptance_rate_weight=float(acceptance_rate_weight),
                      adapt_size_cooling=float(adapt_size_cooling), max_scaling_sd=float(max_scaling_sd),
                      seed=int(seed) & 0xFFFFFFFFFFFFFFFF,
                      rng_mode=L.MCMC_RNG_INJECTED if Z is not None else L.MCMC_RNG_PHILOX,
                      kernel={"auto": L.MCMC_KERNEL_AUTO, "lane": L.MCMC_KERNEL_LANE, "block": L.MCMC_KERNEL_BLOCK,
                              "pair": L.MCMC_KERNEL_PAIR, "row": L.MCMC_KERNEL_ROW}[kernel],
                      iter_offset=i0,
                      factor_update={"exact": L.MCMC_FACTOR_EXACT, "rank1": L.MCMC_FACTOR_RANK1}[factor_update],
                      refresh_every=int(refresh_every),
                      trace_layout={"soa": L.MCMC_TRACE_SOA, "chain_major": L.MCMC_TRACE_CHAIN_MAJOR}[trace_layout])
    if (Z is None) != (U is None):
        raise ValueError("give both Z and U, or neither")
    if Z is not None:
        if tuple(Z.shape) != (n, d, n_chains) or tuple(U.shape) != (n, n_chains):
            raise ValueError("Z must be [n_iterations, d, n_chains] and U [n_iterations, n_chains]")
    io = L.RwmIO()
    io.h_init_theta = _p(h_init)
    io.d_init = _dev_ptr(d_init, f64)
    io.h_covmat = _p(cov)
    io.d_Z = _dev_ptr(Z, f64)
    io.d_U = _dev_ptr(U, f64)
    io.d_trace = _dev_ptr(buf("theta_trace", (n_chains, n_kept, d) if trace_layout == "chain_major" else (n_kept, d, n_chains),
                              want=trace))
    io.d_sigma_trace = _dev_ptr(buf("sigma_empirical", (n_kept, d * d, n_chains), want=sigma_trace))
    io.d_final_theta = _dev_ptr(buf("final_theta", (d, n_chains), want=final))
    io.d_final_mean = _dev_ptr(buf("final_mean", (d, n_chains), want=final))
    io.d_final_cov = _dev_ptr(buf("final_cov", (d * d, n_chains), want=final))
    io.d_acceptance_rate = _dev_ptr(buf("acceptance_rate", (n_chains,)))
    io.d_status = _dev_ptr(buf("status", (n_chains,), torch.int32))
    io.d_shape_iterations = _dev_ptr(buf("shape_iterations", (n_chains,), torch.int32), torch.int32)
    if diagnostics:
        io.d_accepted = _dev_ptr(buf("accepted", (n, n_chains), torch.uint8))
        io.d_log_acceptance = _dev_ptr(buf("log_acceptance", (n, n_chains)))
        io.d_log_A = _dev_ptr(buf("log_A", (n, n_chains)))
        io.d_scaling_sd = _dev_ptr(buf("scaling_sd", (n, n_chains)))
        io.d_scaling_multiplier = _dev_ptr(buf("scaling_multiplier", (n, n_chains)))
        io.d_rate_trace = _dev_ptr(buf("rate_trace", (n, n_chains)))
    if resume is not None:
        io.d_state_in = _dev_ptr(resume.array, f64)
    if checkpoint:
        state = torch.empty((_state_rows(d, acceptance_window, acceptance_rate_weight), n_chains), dtype=f64, device=device)
        io.d_state_out = _dev_ptr(state)
        out["state"] = RwmState(state, i0 + n)
    ct = t._c()
    check(lib.mcmc_adapt_rwm_batch(C.byref(ct), C.byref(cfg), C.byref(io), _stream_ptr()))
    return out


def _state_rows(d, acceptance_window, acceptance_rate_weight):
    """rows of a checkpoint: with a windowed rate over more than 64 outcomes the ring of outcomes rides along"""
    w = int(acceptance_window) if float(acceptance_rate_weight) == 0.0 else 0
    rows = int(lib.mcmc_rwm_state_doubles_window(int(d), w))
    if rows < 0:
        check(rows)
    return rows


@dataclass
class RwmState:
    """Chain state of a Metropolis run (mcmc_rwm_io::d_state_out): a [state_doubles, n_chains] f64 CUDA
    tensor and the number of iterations done.  No counterpart in the reference."""

    array: "object"
    iterations: int

    def save(self, path):
        np.savez(path, array=self.array.cpu().numpy(), iterations=np.int64(self.iterations))

    @staticmethod
    def load(path, device="cuda"):
        torch = _torch()
        with np.load(path) as f:
            return RwmState(torch.tensor(f["array"], dtype=torch.float64, device=device), int(f["iterations"]))


def target_eval_batch(target, x):
    """log-density of `target` at the columns of x ([d, n] CUDA tensor) -> [n] CUDA tensor
    (mcmc_target_eval_batch): checks a compiled target against its host definition."""
    torch = _torch()
    t = _need_target(target)
    d, n = x.shape
    out = torch.empty((n,), dtype=torch.float64, device=x.device)
    ct = t._c()
    check(lib.mcmc_target_eval_batch(C.byref(ct), int(d), int(n), _dev_ptr(x, torch.float64), _dev_ptr(out), _stream_ptr()))
    return out


def rwm_streams(seed, n_chains, n_iterations, d, chain_offset=0, device="cuda"):
    """The N(0,1) / U(0,1) arrays a Philox-mode adapt_rwm_batch run draws (same layouts)."""
    torch = _torch()
    Z = torch.empty((n_iterations, d, n_chains), dtype=torch.float64, device=device)
    U = torch.empty((n_iterations, n_chains), dtype=torch.float64, device=device)
    check(lib.mcmc_rwm_streams(int(seed), int(chain_offset), int(n_chains), int(n_iterations), int(d),
                               _dev_ptr(Z), _dev_ptr(U), _stream_ptr()))
    return Z, U


def init_normal(seed, n_chains, mean, sd, chain_offset=0, device="cuda"):
    """theta0[k, c] = mean[k] + sd * N(0,1) from Philox(seed): a [d, n_chains] CUDA tensor."""
    torch = _torch()
    mean = _vec(mean)
    out = torch.empty((mean.size, n_chains), dtype=torch.float64, device=device)
    check(lib.mcmc_init_normal(int(seed), int(chain_offset), int(n_chains), mean.size, _p(mean), float(sd),
                               _dev_ptr(out), _stream_ptr()))
    return out


def tmvn_gibbs_batch(n_chains, n_sweeps, mu, sigma, lower, upper, init_state, *, seed=0, thin=1, burn=0,
                     chain_offset=0, U=None, tail_safe=False, trace=True, final=True, moments=False,
                     out=None, device="cuda", sweep_offset=0, no_alloc=False):
    """Many independent rtmvnorm_gibbs chains in one launch (tmvrnormGibbs.cpp:91-142 per chain).

    init_state: length-d vector shared by all chains, or a [d, n_chains] CUDA tensor.
    U [n_sweeps, d, n_chains]: injected uniforms.  Returns CUDA tensors: trace
    [n_kept, d, n_chains], final_state [d, n_chains], status [n_chains], and with
    moments=True mean / m2 [d, n_chains] over sweeps > burn.
    Resume: pass the previous call's final_state as init_state and sweep_offset = sweeps done; the
    continuation is bit-identical to one uninterrupted call (the chain state of this sampler is x alone).
    no_alloc=True goes through mcmc_tmvn_gibbs_batch_noalloc: the library allocates nothing (pass every output in
    `out`), d <= 139, calls on one GPU must not overlap.
    """
    torch = _torch()
    f64 = torch.float64
    mu, lower, upper = _vec(mu), _vec(lower), _vec(upper)
    d = mu.size
    sg = _colmajor(sigma, d)
    n_chains, n_sweeps = int(n_chains), int(n_sweeps)
    out = dict(out or {})
    if torch.is_tensor(init_state):
        if tuple(init_state.shape) != (d, n_chains):
            raise ValueError("init tensor must be [d, n_chains]")
        d_init, h_init = init_state, None
    else:
        h_init, d_init = _vec(init_state), None
    if U is not None and tuple(U.shape) != (n_sweeps, d, n_chains):
        raise ValueError("U must be [n_sweeps, d, n_chains]")
    j0 = int(sweep_offset)
    n_kept = (j0 + n_sweeps) // int(thin) - j0 // int(thin)

    def buf(key, shape, dtype=f64, want=True):
        if not want:
            return None
        if key not in out:
            out[key] = torch.empty(shape, dtype=dtype, device=device)
        elif tuple(out[key].shape) != tuple(shape):
            raise ValueError(f"out[{key!r}] has shape {tuple(out[key].shape)}, expected {tuple(shape)}")
        return out[key]

    cfg = L.GibbsConfig(d=d, n_sweeps=n_sweeps, n_chains=n_chains, chain_offset=int(chain_offset),
                        thin=int(thin), burn=int(burn), seed=int(seed) & 0xFFFFFFFFFFFFFFFF,
                        rng_mode=L.MCMC_RNG_INJECTED if U is not None else L.MCMC_RNG_PHILOX,
                        tail_safe=1 if tail_safe else 0, sweep_offset=j0)
    io = L.GibbsIO()
    io.h_mu, io.h_sigma, io.h_lower, io.h_upper = _p(mu), _p(sg), _p(lower), _p(upper)
    io.h_init_state = _p(h_init)
    io.d_init = _dev_ptr(d_init, f64)
    io.d_U = _dev_ptr(U, f64)
    io.d_trace = _dev_ptr(buf("trace", (n_kept, d, n_chains), want=trace))
    io.d_final_state = _dev_ptr(buf("final_state", (d, n_chains), want=final))
    io.d_mean = _dev_ptr(buf("mean", (d, n_chains), want=moments))
    io.d_m2 = _dev_ptr(buf("m2", (d, n_chains), want=moments))
    io.d_status = _dev_ptr(buf("status", (n_chains,), torch.int32))
    entry = lib.mcmc_tmvn_gibbs_batch_noalloc if no_alloc else lib.mcmc_tmvn_gibbs_batch
    check(entry(C.byref(cfg), C.byref(io), _stream_ptr()))
    return out


def gibbs_streams(seed, n_chains, n_sweeps, d, chain_offset=0, device="cuda"):
    """The uniforms a Philox-mode tmvn_gibbs_batch run draws, [n_sweeps, d, n_chains]."""
    torch = _torch()
    U = torch.empty((n_sweeps, d, n_chains), dtype=torch.float64, device=device)
    check(lib.mcmc_gibbs_streams(int(seed), int(chain_offset), int(n_chains), int(n_sweeps), int(d),
                                 _dev_ptr(U), _stream_ptr()))
    return U


def rtnorm_batch(n, mu, sigma, a, b, *, F=None, seed=0, offset=0, tail_safe=False, device="cuda"):
    """n truncated-normal draws on the device (tmvrnormGibbs.cpp:49-70), as a CUDA tensor."""
    torch = _torch()
    out = torch.empty((int(n),), dtype=torch.float64, device=device)
    check(lib.mcmc_rtnorm_batch(int(n), float(mu), float(sigma), float(a), float(b), _dev_ptr(F), int(seed),
                                int(offset), 1 if tail_safe else 0, _dev_ptr(out), _stream_ptr()))
    return out


# --------------------------------------------------------------------------- outside the samplers
def mvrGaussian_pdf(x, mu, sigma, do_log=True):
    """mvrGaussian_pdf (deprecated/truncGaussian.cpp:11-33).  x: a length-d vector -> a float, or a [d, n] CUDA
    tensor of n points -> an [n] CUDA tensor."""
    mu = _vec(mu)
    d = mu.size
    sg = _colmajor(sigma, d)
    if hasattr(x, "is_cuda"):
        torch = _torch()
        out = torch.empty((x.shape[1],), dtype=torch.float64, device=x.device)
        check(lib.mcmc_mvrGaussian_pdf_batch(d, int(x.shape[1]), _dev_ptr(x, torch.float64), _p(mu), _p(sg), 1 if do_log else 0,
                                             _dev_ptr(out), _stream_ptr()))
        return out
    xv = _vec(x)
    out = C.c_double(0.0)
    check(lib.mcmc_mvrGaussian_pdf(d, _p(xv), _p(mu), _p(sg), 1 if do_log else 0, C.byref(out)))
    return out.value


def tmvrGaussian_pdf(x, mu, sigma, lower, upper, cdf=None, *, n_points=1 << 20, seed=0):
    """tmvrGaussian_pdf (deprecated/truncGaussian.cpp:38-78).  cdf = the pmvnorm(lower, upper, mu, sigma) value the
    reference takes from CRAN mvtnorm (:74); None computes it with pmvnorm() below."""
    mu, lower, upper = _vec(mu), _vec(lower), _vec(upper)
    d = mu.size
    sg = _colmajor(sigma, d)
    if cdf is None:
        cdf = pmvnorm(lower, upper, mu, sigma, n_points=n_points, seed=seed)[0]
    if hasattr(x, "is_cuda"):
        torch = _torch()
        out = torch.empty((x.shape[1],), dtype=torch.float64, device=x.device)
        check(lib.mcmc_tmvrGaussian_pdf_batch(d, int(x.shape[1]), _dev_ptr(x, torch.float64), _p(mu), _p(sg), _p(lower), _p(upper),
                                              float(cdf), _dev_ptr(out), _stream_ptr()))
        return out
    xv = _vec(x)
    out = C.c_double(0.0)
    check(lib.mcmc_tmvrGaussian_pdf(d, _p(xv), _p(mu), _p(sg), _p(lower), _p(upper), float(cdf), C.byref(out)))
    return out.value


def pmvnorm(lower, upper, mu, sigma, *, n_points=1 << 20, seed=0):
    """P(lower <= X <= upper) for X ~ N(mu, sigma) (Genz separation of variables on the device): (estimate, std error)."""
    mu, lower, upper = _vec(mu), _vec(lower), _vec(upper)
    sg = _colmajor(sigma, mu.size)
    prob, err = C.c_double(0.0), C.c_double(0.0)
    check(lib.mcmc_pmvnorm(mu.size, _p(lower), _p(upper), _p(mu), _p(sg), int(n_points), int(seed) & 0xFFFFFFFFFFFFFFFF,
                           C.byref(prob), C.byref(err)))
    return prob.value, err.value


def _walk(fn, dim, n, U, seed):
    n = int(n)
    uu = None
    if U is not None:
        uu = _vec(U)
        if uu.size < n - 1:
            raise ValueError("U must hold n-1 uniforms")
    out = np.empty(n * dim)
    check(fn(n, _p(uu), int(seed) & 0xFFFFFFFFFFFFFFFF, _p(out)))
    return out.reshape(n, dim, order="F")


def random_walk2d(n, *, U=None, seed=0):
    """random_walk2d (randomWalkers.cpp:9-37): n x 2 matrix."""
    return _walk(lib.mcmc_random_walk2d, 2, n, U, seed)


def random_walk3d(n, *, U=None, seed=0):
    """random_walk3d (randomWalkers.cpp:40-74): n x 3 matrix."""
    return _walk(lib.mcmc_random_walk3d, 3, n, U, seed)


def random_walk_batch(dim, n, n_walkers, *, U=None, seed=0, walker_offset=0, device="cuda"):
    """Many lattice walkers in one launch: [n, dim, n_walkers] CUDA tensor; U [n-1, n_walkers] injects the uniforms."""
    torch = _torch()
    out = torch.empty((int(n), int(dim), int(n_walkers)), dtype=torch.float64, device=device)
    check(lib.mcmc_random_walk_batch(int(dim), int(n), int(n_walkers), int(walker_offset), _dev_ptr(U), int(seed) & 0xFFFFFFFFFFFFFFFF,
                                     _dev_ptr(out), _stream_ptr()))
    return out


def random_walkGraph(graph, init_pos, n, *, U=None, seed=0):
    """random_walkGraph (randomWalkers.cpp:80-100): the n visited nodes (0-based)."""
    graph = np.asarray(graph, dtype=np.float64)
    nn = graph.shape[0]
    g = _colmajor(graph, nn)
    n = int(n)
    uu = None
    if U is not None:
        uu = _vec(U)
        if uu.size < n - 1:
            raise ValueError("U must hold n-1 uniforms")
    out = np.empty(n)
    check(lib.mcmc_random_walkGraph(nn, _p(g), int(init_pos), n, _p(uu), int(seed) & 0xFFFFFFFFFFFFFFFF, _p(out)))
    return out


def random_walkGraph_batch(graph, init_pos, n, n_walkers, *, U=None, seed=0, walker_offset=0, device="cuda"):
    """Many graph walkers in one launch: [n, n_walkers] CUDA tensor; U [n-1, n_walkers] injects the uniforms."""
    torch = _torch()
    graph = np.asarray(graph, dtype=np.float64)
    nn = graph.shape[0]
    g = _colmajor(graph, nn)
    out = torch.empty((int(n), int(n_walkers)), dtype=torch.float64, device=device)
    check(lib.mcmc_random_walkGraph_batch(nn, _p(g), int(init_pos), int(n), int(n_walkers), int(walker_offset), _dev_ptr(U),
                                          int(seed) & 0xFFFFFFFFFFFFFFFF, _dev_ptr(out), _stream_ptr()))
    return out
