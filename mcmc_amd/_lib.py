"""ctypes binding of mcmc_amd/libmcmc_amd.so (the C ABI declared in include/mcmc_amd.h).

There is no CPU fallback: if the HIP library is missing, importing this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmcmc_amd.so")

MCMC_OK = 0
MCMC_E_BAD_ARG = -1
MCMC_E_UNSUPPORTED = -2
MCMC_E_HIP = -3
MCMC_E_NOT_PD = -4
MCMC_E_CHAIN_FAILED = -5

MCMC_CHAIN_OK = 0
MCMC_CHAIN_CHOL_FAILED = 1
MCMC_CHAIN_NONFINITE_STATE = 2
MCMC_CHAIN_SYNC_LOST = 3
MCMC_CHAIN_BAD_STATE = 4

MCMC_RNG_PHILOX = 0
MCMC_RNG_INJECTED = 1

MCMC_KERNEL_AUTO = 0
MCMC_KERNEL_LANE = 1
MCMC_KERNEL_BLOCK = 2
MCMC_KERNEL_PAIR = 3
MCMC_KERNEL_ROW = 4

MCMC_FACTOR_EXACT = 0
MCMC_FACTOR_RANK1 = 1
MCMC_TRACE_SOA = 0
MCMC_TRACE_CHAIN_MAJOR = 1

MCMC_TARGET_GAUSSIAN_PREC = 0
MCMC_TARGET_GOLDSTEIN_PRICE = 1
MCMC_TARGET_BANANA = 2
MCMC_TARGET_USER = 3

dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int32)
vp = C.c_void_p


class McmcTarget(C.Structure):
    _fields_ = [("kind", C.c_int32), ("n_params", C.c_int32), ("h_params", dp), ("user", C.c_void_p)]


class RwmConfig(C.Structure):
    _fields_ = [
        ("d", C.c_int32), ("n_iterations", C.c_int32),
        ("n_chains", C.c_int64), ("chain_offset", C.c_int64),
        ("adapt_size_start", C.c_int32), ("adapt_shape_start", C.c_int32),
        ("acceptance_window", C.c_int32), ("thin", C.c_int32),
        ("acceptance_rate_weight", C.c_double), ("adapt_size_cooling", C.c_double),
        ("max_scaling_sd", C.c_double),
        ("seed", C.c_uint64), ("rng_mode", C.c_int32), ("kernel", C.c_int32), ("iter_offset", C.c_int32),
        ("factor_update", C.c_int32), ("refresh_every", C.c_int32), ("trace_layout", C.c_int32), ("reserved_", C.c_int32),
    ]


class RwmIO(C.Structure):
    _fields_ = [
        ("h_init_theta", dp), ("d_init", vp), ("h_covmat", dp), ("d_Z", vp), ("d_U", vp),
        ("d_trace", vp), ("d_sigma_trace", vp), ("d_final_theta", vp), ("d_final_mean", vp),
        ("d_final_cov", vp), ("d_acceptance_rate", vp), ("d_status", vp),
        ("d_accepted", vp), ("d_log_acceptance", vp), ("d_log_A", vp), ("d_scaling_sd", vp),
        ("d_scaling_multiplier", vp), ("d_rate_trace", vp), ("d_shape_iterations", vp),
        ("d_state_in", vp), ("d_state_out", vp),
    ]


class GibbsConfig(C.Structure):
    _fields_ = [
        ("d", C.c_int32), ("n_sweeps", C.c_int32),
        ("n_chains", C.c_int64), ("chain_offset", C.c_int64),
        ("thin", C.c_int32), ("burn", C.c_int32),
        ("seed", C.c_uint64), ("rng_mode", C.c_int32), ("tail_safe", C.c_int32), ("sweep_offset", C.c_int32),
    ]


class GibbsIO(C.Structure):
    _fields_ = [
        ("h_mu", dp), ("h_sigma", dp), ("h_lower", dp), ("h_upper", dp), ("h_init_state", dp),
        ("d_init", vp), ("d_U", vp), ("d_trace", vp), ("d_final_state", vp), ("d_mean", vp),
        ("d_m2", vp), ("d_status", vp),
    ]


# every symbol include/mcmc_amd.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "mcmc_amd_abi_version": (C.c_int, []),
    "mcmc_amd_last_error": (C.c_char_p, []),
    "mcmc_amd_max_lane_dim": (C.c_int, []),
    "mcmc_amd_max_dim": (C.c_int, []),
    "mcmc_target_compile": (C.c_int, [C.c_char_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p), C.c_char_p, C.c_int64]),
    "mcmc_target_destroy": (C.c_int, [vp]),
    "mcmc_target_eval_batch": (C.c_int, [C.POINTER(McmcTarget), C.c_int32, C.c_int64, vp, vp, vp]),
    "mcmc_adapt_rwm_batch": (C.c_int, [C.POINTER(McmcTarget), C.POINTER(RwmConfig), C.POINTER(RwmIO), vp]),
    "mcmc_rwm_state_doubles": (C.c_int64, [C.c_int32]),
    "mcmc_rwm_state_doubles_window": (C.c_int64, [C.c_int32, C.c_int32]),
    "mcmc_rwm_streams": (C.c_int, [C.c_uint64, C.c_int64, C.c_int64, C.c_int32, C.c_int32, vp, vp, vp]),
    "mcmc_init_normal": (C.c_int, [C.c_uint64, C.c_int64, C.c_int64, C.c_int32, dp, C.c_double, vp, vp]),
    "mcmc_adaptMCMC": (C.c_int, [C.POINTER(McmcTarget), C.c_int32, dp, dp, C.c_int32, C.c_int32, C.c_double,
                                 C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double, dp, dp,
                                 dp, dp, dp, dp, ip]),
    "mcmc_rwMCMC": (C.c_int, [C.POINTER(McmcTarget), C.c_int32, dp, dp, C.c_int32, C.c_int32, C.c_double, dp, dp,
                              dp, dp, ip]),
    "mcmc_adaptMCMC_chains": (C.c_int, [C.POINTER(McmcTarget), C.c_int32, C.c_int64, dp, C.c_int32, dp, C.c_int32, C.c_int32,
                                        C.c_double, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double,
                                        C.c_int32, dp, dp, dp, ip]),
    "mcmc_rtmvnorm_gibbs_chains": (C.c_int, [C.c_int32, C.c_int32, C.c_int64, dp, dp, dp, dp, dp, C.c_int32, C.c_uint64,
                                             C.c_int32, C.c_int32, dp, ip]),
    "mcmc_tmvn_gibbs_batch": (C.c_int, [C.POINTER(GibbsConfig), C.POINTER(GibbsIO), vp]),
    "mcmc_tmvn_gibbs_batch_noalloc": (C.c_int, [C.POINTER(GibbsConfig), C.POINTER(GibbsIO), vp]),
    "mcmc_tmvn_gibbs_noalloc_max_dim": (C.c_int, []),
    "mcmc_tmvn_precompute": (C.c_int, [C.c_int32, dp, dp, dp]),
    "mcmc_gibbs_streams": (C.c_int, [C.c_uint64, C.c_int64, C.c_int64, C.c_int32, C.c_int32, vp, vp]),
    "mcmc_rtmvnorm_gibbs": (C.c_int, [C.c_int32, C.c_int32, dp, dp, dp, dp, dp, dp, C.c_uint64, dp]),
    "mcmc_rtnorm_gibbs": (C.c_int, [C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double, dp, C.c_uint64, dp]),
    "mcmc_rtnorm_batch": (C.c_int, [C.c_int64, C.c_double, C.c_double, C.c_double, C.c_double, vp, C.c_uint64,
                                    C.c_int64, C.c_int32, vp, vp]),
    "mcmc_sub1": (C.c_int, [C.c_int32, dp, C.c_int32, dp]),
    "mcmc_sub2": (C.c_int, [C.c_int32, C.c_int32, dp, C.c_int32, C.c_int32, dp]),
    "mcmc_negSubCol": (C.c_int, [C.c_int32, dp, C.c_int32, dp]),
    "mcmc_negSubRow": (C.c_int, [C.c_int32, dp, C.c_int32, dp]),
    "mcmc_chain_moments": (C.c_int, [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int64, vp, vp, vp]),
    "mcmc_rhat_partial": (C.c_int, [vp, vp, C.c_int32, C.c_int64, vp, vp]),
    "mcmc_pooled_cov_partial": (C.c_int, [vp, vp, C.c_int32, C.c_int64, vp, vp]),
    "mcmc_chain_ess": (C.c_int, [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int32, vp, vp]),
    "mcmc_chain_ess_capped": (C.c_int, [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int32, vp, vp, vp]),
    "mcmc_mvrGaussian_pdf_batch": (C.c_int, [C.c_int32, C.c_int64, vp, dp, dp, C.c_int32, vp, vp]),
    "mcmc_mvrGaussian_pdf": (C.c_int, [C.c_int32, dp, dp, dp, C.c_int32, dp]),
    "mcmc_tmvrGaussian_pdf_batch": (C.c_int, [C.c_int32, C.c_int64, vp, dp, dp, dp, dp, C.c_double, vp, vp]),
    "mcmc_tmvrGaussian_pdf": (C.c_int, [C.c_int32, dp, dp, dp, dp, dp, C.c_double, dp]),
    "mcmc_pmvnorm": (C.c_int, [C.c_int32, dp, dp, dp, dp, C.c_int64, C.c_uint64, dp, dp]),
    "mcmc_random_walk2d": (C.c_int, [C.c_int32, dp, C.c_uint64, dp]),
    "mcmc_random_walk3d": (C.c_int, [C.c_int32, dp, C.c_uint64, dp]),
    "mcmc_random_walk_batch": (C.c_int, [C.c_int32, C.c_int32, C.c_int64, C.c_int64, vp, C.c_uint64, vp, vp]),
    "mcmc_random_walkGraph": (C.c_int, [C.c_int32, dp, C.c_int32, C.c_int32, dp, C.c_uint64, dp]),
    "mcmc_random_walkGraph_batch": (C.c_int, [C.c_int32, dp, C.c_int32, C.c_int32, C.c_int64, C.c_int64, vp, C.c_uint64, vp, vp]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP library is the product and has no CPU fallback. "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C mcmc_amd/csrc`)."
        )
    try:  # torch ships its own copy of the HIP runtime: load it first, so that this library binds the runtime whose
        import torch  # noqa: F401  device buffers and streams it is handed (otherwise: "no ROCm-capable device")
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


class McmcError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"mcmc_amd error {code}: {message}")
        self.code = code


def check(rc: int) -> None:
    if rc != MCMC_OK:
        raise McmcError(rc, lib.mcmc_amd_last_error().decode("utf-8", "replace"))
