"""mcmc_amd -- MI355X (gfx950) batched MCMC engine behind the slwu89/MCMC Rcpp interface.

The product is mcmc_amd/libmcmc_amd.so (hand-written HIP kernels + C ABI, include/mcmc_amd.h);
this package is its host-side mirror of the reference interface.  Importing it loads the
library and raises if it is missing: there is no CPU fallback.
"""
from ._lib import (  # noqa: F401
    MCMC_CHAIN_BAD_STATE, MCMC_CHAIN_CHOL_FAILED, MCMC_CHAIN_NONFINITE_STATE, MCMC_CHAIN_OK, McmcError, lib,
)
from .api import (  # noqa: F401
    RwmState, Target, adaptMCMC, adaptMCMC_chains, rtmvnorm_gibbs_chains, adapt_rwm_batch, gibbs_streams, init_normal, negSubCol, negSubRow, rtmvnorm_gibbs,
    rtnorm_batch, rtnorm_gibbs, rwMCMC, rwm_streams, sub1, sub2, target_eval_batch, tmvn_gibbs_batch, tmvn_precompute,
    mvrGaussian_pdf, tmvrGaussian_pdf, pmvnorm, random_walk2d, random_walk3d, random_walk_batch, random_walkGraph,
    random_walkGraph_batch,
)
from .chains import chain_list, melt_trace, run_chains  # noqa: F401,E402
