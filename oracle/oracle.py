"""ctypes front-end of the CPU oracle (oracle/mcmc_oracle.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py; never by mcmc_amd/ (the product).

Function names mirror the reference's Rcpp exports (adaptMCMC_source.cpp:74,229;
tmvrnormGibbs.cpp:11,19,26,33,48,90); random streams are injected as arrays.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_STD = None

TARGET_GAUSSIAN_PREC = 0
TARGET_GOLDSTEIN_PRICE = 1
TARGET_BANANA = 2

ORC_OK, ORC_CHOL_FAILED, ORC_BAD_ARG, ORC_SINGULAR = 0, 1, 2, 3

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


class _Diag(C.Structure):
    _fields_ = [
        ("accepted", _ip),
        ("log_acceptance", _dp),
        ("log_A", _dp),
        ("pdf_term", _dp),
        ("scaling_sd", _dp),
        ("acceptance_rate", _dp),
        ("target_propose", _dp),
    ]


TARGET_CB = C.CFUNCTYPE(C.c_double, _dp, C.c_int, C.c_void_p)


def build(force: bool = False) -> None:
    """Compile liboracle.so / libstdcheck.so in place (gcc, seconds)."""
    need = force or not all(
        os.path.exists(os.path.join(_HERE, f)) for f in ("liboracle.so", "libstdcheck.so")
    )
    if not need:
        src_m = max(
            os.path.getmtime(os.path.join(_HERE, f))
            for f in ("mcmc_oracle.c", "stdlib_streams_check.cpp", "Makefile")
        )
        so_m = min(
            os.path.getmtime(os.path.join(_HERE, f)) for f in ("liboracle.so", "libstdcheck.so")
        )
        need = src_m > so_m
    if need:
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_pnorm.restype = C.c_double
        L.orc_pnorm.argtypes = [C.c_double] * 3
        L.orc_qnorm.restype = C.c_double
        L.orc_qnorm.argtypes = [C.c_double]
        L.orc_det.restype = C.c_double
        L.orc_mvrnorm_pdf.restype = C.c_double
        L.orc_target_eval.restype = C.c_double
        L.orc_mt19937_u32.restype = C.c_uint32
        L.orc_mt19937_u32.argtypes = [C.c_uint32, C.c_int]
        L.orc_stdlib_streams.argtypes = [C.c_double, C.c_int, C.c_int, _dp, _dp]
        L.orc_update_sigma.argtypes = [C.c_int, _dp, _dp, C.c_double, _dp]
        L.orc_rtnorm_gibbs.argtypes = [C.c_int] + [C.c_double] * 4 + [_dp, _dp]
        L.orc_pnorm_vec.argtypes = [C.c_int, _dp, C.c_double, C.c_double, _dp]
        _LIB = L
    return _LIB


def stdlib_check_lib():
    global _STD
    if _STD is None:
        path = os.path.join(_HERE, "libstdcheck.so")
        if not os.path.exists(path):
            build()
        _STD = C.CDLL(path)
        _STD.stdcheck_streams.argtypes = [C.c_double, C.c_int, C.c_int, _dp, _dp]
    return _STD


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def _colmajor(a):
    """d x d numpy matrix -> flat column-major buffer."""
    return np.asfortranarray(np.asarray(a, dtype=np.float64)).ravel(order="F").copy()


def _p(a):
    return a.ctypes.data_as(_dp)


# --------------------------------------------------------------------------- scalar pieces
def pnorm(x, mu=0.0, sigma=1.0):
    x = _f64(np.atleast_1d(x))
    out = np.empty_like(x)
    lib().orc_pnorm_vec(x.size, _p(x), float(mu), float(sigma), _p(out))
    return out


def qnorm(p):
    p = _f64(np.atleast_1d(p))
    out = np.empty_like(p)
    lib().orc_qnorm_vec(p.size, _p(p), _p(out))
    return out


def chol(A):
    A = np.asarray(A, dtype=np.float64)
    d = A.shape[0]
    a = _colmajor(A)
    r = np.empty(d * d)
    st = lib().orc_chol_upper(d, _p(a), _p(r))
    return st, r.reshape(d, d, order="F")


def det(A):
    A = np.asarray(A, dtype=np.float64)
    a = _colmajor(A)
    return lib().orc_det(A.shape[0], _p(a))


def inv(A):
    A = np.asarray(A, dtype=np.float64)
    d = A.shape[0]
    a = _colmajor(A)
    r = np.empty(d * d)
    st = lib().orc_inv(d, _p(a), _p(r))
    return st, r.reshape(d, d, order="F")


def update_sigma(sigma, residual, i):
    """adaptMCMC_source.cpp:42-45"""
    sigma = np.asarray(sigma, dtype=np.float64)
    d = sigma.shape[0]
    s = _colmajor(sigma)
    r = _f64(residual)
    out = np.empty(d * d)
    lib().orc_update_sigma(d, _p(s), _p(r), float(i), _p(out))
    return out.reshape(d, d, order="F")


def mvrnorm_pdf(x, mu, sigma):
    """adaptMCMC_source.cpp:48-64"""
    x, mu = _f64(x), _f64(mu)
    s = _colmajor(sigma)
    return lib().orc_mvrnorm_pdf(x.size, _p(x), _p(mu), _p(s))


def stdlib_streams(seedMH, n_iterations, d):
    Z = np.empty(n_iterations * d)
    U = np.empty(n_iterations)
    lib().orc_stdlib_streams(float(seedMH), n_iterations, d, _p(Z), _p(U))
    return Z.reshape(n_iterations, d), U


# --------------------------------------------------------------------------- targets
class Target:
    """A built-in log-density (id + parameter block) or a Python callable."""

    def __init__(self, kind, ctx=None, fn=None):
        self.kind = kind
        self._ctx = None if ctx is None else _f64(ctx)
        self._fn = fn
        self._cb = None
        if fn is not None:
            def _tramp(ptr, d, _ctx):
                th = np.ctypeslib.as_array(ptr, shape=(d,))
                return float(fn(th.copy()))
            self._cb = TARGET_CB(_tramp)

    @property
    def ctx_ptr(self):
        return None if self._ctx is None else self._ctx.ctypes.data_as(C.c_void_p)

    def __call__(self, theta):
        theta = _f64(theta)
        if self._fn is not None:
            return float(self._fn(theta))
        return lib().orc_target_eval(self.kind, _p(theta), theta.size, self.ctx_ptr)


def gaussian_target(mean, precision):
    mean = _f64(mean)
    return Target(TARGET_GAUSSIAN_PREC, np.concatenate([mean, _colmajor(precision)]))


def goldstein_price_target():
    return Target(TARGET_GOLDSTEIN_PRICE)


def banana_target(B=0.03):
    return Target(TARGET_BANANA, np.array([B]))


def callable_target(fn):
    return Target(-1, fn=fn)


# --------------------------------------------------------------------------- samplers
def _make_diag(n, want):
    if not want:
        return None, None
    arrs = dict(
        accepted=np.zeros(n, dtype=np.int32),
        log_acceptance=np.full(n, np.nan),
        log_A=np.full(n, np.nan),
        pdf_term=np.full(n, np.nan),
        scaling_sd=np.full(n, np.nan),
        acceptance_rate=np.full(n, np.nan),
        target_propose=np.full(n, np.nan),
    )
    dg = _Diag(
        arrs["accepted"].ctypes.data_as(_ip),
        *[_p(arrs[k]) for k in ("log_acceptance", "log_A", "pdf_term", "scaling_sd",
                                "acceptance_rate", "target_propose")],
    )
    return dg, arrs


def adaptMCMC(target, init_theta, covmat, n_iterations, adapt_size_start, acceptance_rate_weight,
              acceptance_window, adapt_shape_start, Z, U, adapt_size_cooling=0.99,
              max_scaling_sd=50.0, store_sigma=True, diagnostics=False):
    """adaptMCMC_source.cpp:75-219 with injected streams Z (n x d) and U (n).

    Returns dict(theta_trace n x d, sigma_empirical d x d x n | None, acceptance_rate,
    status, fail_iter[, diag])."""
    init = _f64(init_theta)
    d = init.size
    cov = _colmajor(covmat)
    Z = _f64(Z).reshape(-1)
    U = _f64(U).reshape(-1)
    n = int(n_iterations)
    assert Z.size >= n * d and U.size >= n
    trace = np.zeros(n * d)
    sig = np.zeros(d * d * n) if store_sigma else None
    acc = C.c_double(0.0)
    fail = C.c_int(0)
    dg, arrs = _make_diag(n, diagnostics)
    L = lib()
    common = (d, _p(init), _p(cov), n, int(adapt_size_start), C.c_double(acceptance_rate_weight),
              int(acceptance_window), int(adapt_shape_start), C.c_double(adapt_size_cooling),
              C.c_double(max_scaling_sd), _p(Z), _p(U), _p(trace),
              _p(sig) if sig is not None else None, C.byref(acc), C.byref(fail),
              C.byref(dg) if dg is not None else None)
    if target._cb is not None:
        st = L.orc_adapt_mcmc(target._cb, None, *common)
    else:
        st = L.orc_adapt_mcmc_builtin(target.kind, target.ctx_ptr, *common)
    out = dict(
        theta_trace=trace.reshape(n, d, order="F"),
        sigma_empirical=None if sig is None else sig.reshape(d, d, n, order="F"),
        acceptance_rate=acc.value, status=st, fail_iter=fail.value,
    )
    if arrs is not None:
        out["diag"] = arrs
    return out


def rwMCMC(target, init_theta, covmat, n_iterations, Z, U, diagnostics=False):
    """adaptMCMC_source.cpp:230-316 with injected streams."""
    init = _f64(init_theta)
    d = init.size
    cov = _colmajor(covmat)
    Z = _f64(Z).reshape(-1)
    U = _f64(U).reshape(-1)
    n = int(n_iterations)
    trace = np.zeros(n * d)
    acc = C.c_double(0.0)
    dg, arrs = _make_diag(n, diagnostics)
    L = lib()
    common = (d, _p(init), _p(cov), n, _p(Z), _p(U), _p(trace), C.byref(acc),
              C.byref(dg) if dg is not None else None)
    if target._cb is not None:
        st = L.orc_rw_mcmc(target._cb, None, *common)
    else:
        st = L.orc_rw_mcmc_builtin(target.kind, target.ctx_ptr, *common)
    out = dict(theta_trace=trace.reshape(n, d, order="F"), acceptance_rate=acc.value, status=st)
    if arrs is not None:
        out["diag"] = arrs
    return out


def adaptMCMC_seeded(target, init_theta, covmat, n_iterations, adapt_size_start,
                     acceptance_rate_weight, acceptance_window, adapt_shape_start, seedMH,
                     adapt_size_cooling=0.99, max_scaling_sd=50.0, store_sigma=False):
    """The reference call as a libstdc++ build would run it (RNG inline). CPU-baseline leg."""
    init = _f64(init_theta)
    d = init.size
    cov = _colmajor(covmat)
    n = int(n_iterations)
    trace = np.zeros(n * d)
    sig = np.zeros(d * d * n) if store_sigma else None
    acc = C.c_double(0.0)
    st = lib().orc_adapt_mcmc_seeded(
        target.kind, target.ctx_ptr, d, _p(init), _p(cov), n, int(adapt_size_start),
        C.c_double(acceptance_rate_weight), int(acceptance_window), int(adapt_shape_start),
        C.c_double(seedMH), C.c_double(adapt_size_cooling), C.c_double(max_scaling_sd),
        _p(trace), _p(sig) if sig is not None else None, C.byref(acc))
    return dict(theta_trace=trace.reshape(n, d, order="F"), acceptance_rate=acc.value, status=st)


def sub1(x, i):
    x = np.asarray(x, dtype=np.float64)
    d = x.shape[0]
    a = _colmajor(x)
    out = np.empty((d - 1) * (d - 1))
    lib().orc_sub1(d, _p(a), int(i), _p(out))
    return out.reshape(d - 1, d - 1, order="F")


def sub2(x, a, b):
    x = np.asarray(x, dtype=np.float64)
    nr, nc = x.shape
    buf = _colmajor(x)
    out = np.empty(nc - 1)
    lib().orc_sub2(nr, nc, _p(buf), int(a), int(b), _p(out))
    return out.reshape(1, nc - 1)


def negSubCol(x, i):
    x = _f64(x).reshape(-1)
    out = np.empty(x.size - 1)
    lib().orc_neg_sub(x.size, _p(x), int(i), _p(out))
    return out


negSubRow = negSubCol


def rtnorm_gibbs(n, mu, sigma, a, b, F):
    """tmvrnormGibbs.cpp:49-70 with F = the runif(n) draws."""
    F = _f64(F)
    out = np.empty(n)
    lib().orc_rtnorm_gibbs(int(n), float(mu), float(sigma), float(a), float(b), _p(F), _p(out))
    return out


def tmvn_precompute(sigma):
    """tmvrnormGibbs.cpp:100-113 -> P (d x (d-1)), sd (d)."""
    sigma = np.asarray(sigma, dtype=np.float64)
    d = sigma.shape[0]
    s = _colmajor(sigma)
    P = np.empty(d * max(d - 1, 1))
    sd = np.empty(d)
    st = lib().orc_tmvn_precompute(d, _p(s), _p(P), _p(sd))
    return st, P.reshape(d, max(d - 1, 1))[:, : d - 1], sd


def rtmvnorm_gibbs(n, mu, sigma, lower, upper, init_state, U):
    """tmvrnormGibbs.cpp:91-142 with U = the runif(n*d) draws; returns n x d."""
    mu, lower, upper, init = _f64(mu), _f64(lower), _f64(upper), _f64(init_state)
    d = mu.size
    s = _colmajor(sigma)
    U = _f64(U).reshape(-1)
    assert U.size >= n * d
    trace = np.zeros(n * d)
    st = lib().orc_rtmvnorm_gibbs(int(n), d, _p(mu), _p(s), _p(lower), _p(upper), _p(init),
                                  _p(U), _p(trace))
    if st != ORC_OK:
        raise RuntimeError(f"oracle rtmvnorm_gibbs failed: status {st}")
    return trace.reshape(n, d, order="F")


# ---- SURVEY.md section 8(f) row 4 -------------------------------------------------------------
def mvrGaussian_pdf(x, mu, sigma, do_log=True):
    """deprecated/truncGaussian.cpp:11-33"""
    x, mu = _f64(x), _f64(mu)
    lib().orc_mvrGaussian_pdf.restype = C.c_double
    return lib().orc_mvrGaussian_pdf(x.size, _p(x), _p(mu), _p(_colmajor(sigma)), 1 if do_log else 0)


def tmvrGaussian_pdf(x, mu, sigma, lower, upper, cdf):
    """deprecated/truncGaussian.cpp:38-78 with cdf = the pmvnorm(lower, upper, mu, sigma) value of :74-75"""
    x, mu, lower, upper = _f64(x), _f64(mu), _f64(lower), _f64(upper)
    lib().orc_tmvrGaussian_pdf.restype = C.c_double
    return lib().orc_tmvrGaussian_pdf(x.size, _p(x), _p(mu), _p(_colmajor(sigma)), _p(lower), _p(upper), C.c_double(cdf))


def random_walk(dim, n, U):
    """random_walk2d / random_walk3d (randomWalkers.cpp:9-74): n x dim, U = the n-1 runif draws"""
    U = _f64(U)
    out = np.zeros(n * dim)
    lib().orc_random_walk(int(dim), int(n), _p(U), _p(out))
    return out.reshape(n, dim, order="F")


def random_walkGraph(graph, init_pos, n, U):
    """random_walkGraph (randomWalkers.cpp:80-100), 0-based nodes, U = the n-1 unif_rand draws of sample()"""
    graph = np.asarray(graph, dtype=np.float64)
    U = _f64(U)
    out = np.zeros(n)
    st = lib().orc_random_walkGraph(graph.shape[0], _p(_colmajor(graph)), int(init_pos), int(n), _p(U), _p(out))
    if st != ORC_OK:
        raise RuntimeError(f"oracle random_walkGraph failed: status {st}")
    return out
