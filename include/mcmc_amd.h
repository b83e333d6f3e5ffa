/*
 * mcmc_amd.h -- C ABI of the MI355X (gfx950) batched MCMC engine.
 *
 * This is the drop-in boundary for the two sampling paths of slwu89/MCMC.  Every entry
 * point cites the reference interface it replaces (paths are into the reference repo).
 * Plain pointers and sizes only; no C++ or torch types.  The library is
 * mcmc_amd/libmcmc_amd.so (hipcc --offload-arch=gfx950); INTEGRATION.md shows the
 * Rcpp-side binding.
 *
 * Conventions
 *   - "h_" arguments are HOST pointers, "d_" arguments are DEVICE (HBM) pointers on the
 *     current HIP device.  The single-chain drop-in calls take host pointers only.
 *   - Matrices handed over by the caller are column-major (R / Armadillo layout).
 *   - Batch arrays are structure-of-arrays with the CHAIN index fastest:
 *       state  [d][n_chains]              element (k, c)      at k*n_chains + c
 *       trace  [n_kept][d][n_chains]      element (t, k, c)   at (t*d + k)*n_chains + c
 *     so one wavefront (64 chains, or 64 consecutive lanes) writes 512 contiguous bytes.
 *   - Every call returns MCMC_OK (0) or a negative MCMC_E_* code; mcmc_amd_last_error()
 *     gives the message.  Per-chain failures (what would be a C++ exception inside the
 *     reference's loop) are reported through the d_status / status outputs instead.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  A batch call is
 *     asynchronous on `stream`: its small shared parameter block (and the cooling table)
 *     travels in the arguments of a few fill kernels queued ahead of the sampling kernel, so
 *     the call returns at once and waits for nothing queued on the stream earlier.  It does
 *     wait once (a copy + a stream synchronisation) where such a block exceeds 224 KB -- a
 *     user target's data set, the cooling table of a call of more than 28 671 iterations --
 *     and when a run is resumed under MCMC_KERNEL_AUTO (the checkpoint's kernel tag is read
 *     back).  (The HIP runtime may still hold a launch on the host when kernels that use
 *     scratch memory follow one another on a busy device.)  The single-chain calls and the
 *     density / walker helpers synchronise before returning.
 *   - Random numbers: rng_mode MCMC_RNG_PHILOX draws from Philox4x32-10 keyed by `seed` and
 *     counted by (global chain id, iteration, draw), so results do not depend on how chains
 *     are split over launches or GPUs; MCMC_RNG_INJECTED consumes caller-supplied N(0,1) /
 *     U(0,1) arrays in exactly the order the reference draws them (the parity mode).
 */
#ifndef MCMC_AMD_H
#define MCMC_AMD_H

#ifndef __HIPCC_RTC__ /* the run-time compiler (user targets) brings its own fixed-width types */
#include <stdint.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define MCMC_AMD_ABI_VERSION 2 /* 2: mcmc_rwm_config grew factor_update / refresh_every / trace_layout */

/* ----- return codes ---------------------------------------------------------------- */
#define MCMC_OK 0
#define MCMC_E_BAD_ARG (-1)       /* shape / range / NULL pointer problem, nothing launched   */
#define MCMC_E_UNSUPPORTED (-2)   /* valid request this build has no kernel for               */
#define MCMC_E_HIP (-3)           /* a HIP runtime call failed                                */
#define MCMC_E_NOT_PD (-4)        /* a shared (host-side) matrix is not positive definite     */
#define MCMC_E_CHAIN_FAILED (-5)  /* single-chain call: the chain stopped (see *fail_iter)    */

/* ----- per-chain status word (d_status) -------------------------------------------- */
/* 0 = ran to completion.  Otherwise (iteration << 4) | code, iteration 1-based.          */
#define MCMC_CHAIN_OK 0
#define MCMC_CHAIN_CHOL_FAILED 1      /* arma::chol would have thrown (adaptMCMC_source.cpp:153) */
#define MCMC_CHAIN_NONFINITE_STATE 2  /* Gibbs draw became +-Inf/NaN (tmvrnormGibbs.cpp:132 has no tail handling) */
#define MCMC_CHAIN_SYNC_LOST 3        /* internal hand-shake between the wavefronts of a chain group timed out (a bug, never data) */
#define MCMC_CHAIN_BAD_STATE 4        /* resume: d_state_in was not written by this kernel at iteration iter_offset */
#define MCMC_STATUS_CODE(s) ((s) & 15)
#define MCMC_STATUS_ITER(s) ((s) >> 4)

/* ----- random-number source --------------------------------------------------------- */
#define MCMC_RNG_PHILOX 0
#define MCMC_RNG_INJECTED 1

/* ----- kernel choice for the Metropolis path ----------------------------------------- */
#define MCMC_KERNEL_AUTO 0   /* one chain per lane for d <= mcmc_amd_max_lane_dim() (the paired-wave variant when it
                                serves the request), else per workgroup.  The two per-lane kernels round differently
                                (1e-12 relative), so under AUTO asking for a diagnostics array can change the low bits of
                                a trace; a resumed AUTO run returns to the kernel that wrote the checkpoint, or is
                                refused with MCMC_E_UNSUPPORTED when that kernel cannot serve the call        */
#define MCMC_KERNEL_LANE 1   /* one chain per lane (register-resident state), d <= 12               */
#define MCMC_KERNEL_BLOCK 2  /* one chain per 256-thread workgroup (LDS-resident factor), d <= 128,
                                Gaussian target only                                               */
#define MCMC_KERNEL_PAIR 3   /* one chain per lane, a sampler and a generator wavefront per 64 chains;
                                Philox streams, no per-iteration diagnostics, d <= 10              */
#define MCMC_KERNEL_ROW 4    /* one chain per row of 16 (d <= 16), 32 (d <= 32) or 64 lanes (d <= 64) of a wavefront, one coordinate
                                and one matrix column per lane; Gaussian target, Philox streams, no per-iteration
                                diagnostics.  What AUTO runs for 12 < d <= 64 when it serves the request   */

/* ----- how the proposal factor follows covmat_empirical in the shape phase ------------ */
#define MCMC_FACTOR_EXACT 0  /* the reference: a fresh chol(covmat_proposal) every iteration (adaptMCMC_source.cpp:153);
                                the parity mode and the default                                                   */
#define MCMC_FACTOR_RANK1 1  /* NOT a parity mode.  update_sigma (:42-45) changes covmat_empirical by a rank-1 term, so
                                its factor is updated in O(d^2) -- R_i = sqrt(a) chol(I + p p^T) R_{i-1}, p the whitened
                                residual, in closed form, no d-step recurrence -- and refactored from covmat_empirical
                                every refresh_every iterations (absolute iteration numbers).  Same algorithm in exact
                                arithmetic; trajectories follow the exact mode to rounding drift until a decision
                                flips (DESIGN.md has the measured drift).  Per-workgroup kernel only (d > 12)       */

/* ----- layout of d_trace ------------------------------------------------------------------ */
#define MCMC_TRACE_SOA 0          /* [n_kept][d][n_chains], chain fastest: the default, what every other call reads */
#define MCMC_TRACE_CHAIN_MAJOR 1  /* [n_chains][n_kept][d]: a kept row of a chain is d contiguous doubles.  For the
                                     kernels that give a chain a whole wavefront or workgroup (MCMC_KERNEL_ROW at
                                     32 < d <= 64, MCMC_KERNEL_BLOCK), whose SoA stores fill 8 bytes of a line each    */

/* ----- target log-densities --------------------------------------------------------- */
/*
 * Replaces the R closure `target` of adaptMCMC / rwMCMC (adaptMCMC_source.cpp:75,97,158;
 * :230,246,277), which a GPU cannot call.  A target is a kind plus a host parameter block.
 */
#define MCMC_TARGET_GAUSSIAN_PREC 0   /* -0.5 (x-m)^T H (x-m); params = m[d], H[d*d] column-major */
#define MCMC_TARGET_GOLDSTEIN_PRICE 1 /* adaptMCMC_analysis.R:115-129 (d = 2); no params          */
#define MCMC_TARGET_BANANA 2          /* adaptMCMC_analysis.R:19-22  (d = 2); params = B          */
#define MCMC_TARGET_USER 3            /* a log-density supplied as HIP source, see mcmc_target_compile */

typedef struct mcmc_target {
    int32_t kind;
    int32_t n_params;
    const double *h_params;
    const void *user;              /* MCMC_TARGET_USER: handle from mcmc_target_compile; else NULL */
} mcmc_target;

/*
 * User targets: the compiled-code replacement for an arbitrary R closure `target`
 * (adaptMCMC_source.cpp:75,97,158).  `source` is HIP device code that defines
 *
 *     __device__ double mcmc_user_target(const double (&x)[MCMC_USER_D], const double *params);
 *
 * returning the log-density at x (any non-finite value rejects the proposal, :163-165), for d <= 12
 * (one chain per lane; `params` = the n_params <= 4096 doubles of mcmc_target::h_params, shared by
 * all chains, resident in LDS), and for 12 < d <= 128 (one chain per 256-thread workgroup)
 *
 *     __device__ double mcmc_user_target_part(const double *x, int d, const double *params,
 *                                             int part, int nparts);
 *
 * returning part `part` of `nparts` of the log-density -- the kernel calls it once per thread and adds
 * the parts up, so a sum over observations or coordinates is strided by nparts; x is in LDS, `params`
 * (n_params <= 4 194 304 doubles: room for a data set) in HBM / L2.
 * MCMC_USER_D (= d) and MCMC_USER_NP (= n_params) are predefined; the lean
 * mcmc::fm_log / fm_exp / fm_sqrt_rsq of csrc/fastmath.h and the usual device math are available.
 * The source is compiled for gfx950 with hipRTC together with the Metropolis kernels, so the
 * function is inlined into the sampling loop.  Needs the kernel
 * sources at run time: <library dir>/csrc and <library dir>/../include, or $MCMC_AMD_SRC_DIR,
 * $MCMC_AMD_INCLUDE_DIR; HIP headers from $ROCM_PATH (default /opt/rocm).
 * On a compile error the call returns MCMC_E_BAD_ARG and `log` (capacity log_capacity, may be
 * NULL) receives the compiler's diagnostics.  Compiling needs no GPU; the module is loaded on
 * each device at first use.  Handles are immutable and may be shared between threads.
 */
int mcmc_target_compile(const char *source, int32_t d, int32_t n_params, void **handle,
                        char *log, int64_t log_capacity);
int mcmc_target_destroy(void *handle);
/* log-density of `target` at n points (built-in kinds d <= 12, user targets d <= 128): d_x is [d][n], d_out is [n].  Lets a
 * caller check a compiled target against its host definition before sampling.  No counterpart in
 * the reference (there the closure can simply be called from R). */
int mcmc_target_eval_batch(const mcmc_target *target, int32_t d, int64_t n, const double *d_x,
                           double *d_out, void *stream);

/* ----- library ----------------------------------------------------------------------- */
int mcmc_amd_abi_version(void);
const char *mcmc_amd_last_error(void);
/* largest d served by the one-chain-per-lane Metropolis kernel; above it the per-workgroup kernel runs */
int mcmc_amd_max_lane_dim(void);
int mcmc_amd_max_dim(void);

/* =====================================================================================
 * Adaptive random-walk Metropolis, many chains
 * replaces: List adaptMCMC(Function target, arma::vec init_theta, arma::mat covmat,
 *   int n_iterations, int adapt_size_start, double acceptance_rate_weight,
 *   int acceptance_window, int adapt_shape_start, int info, double seedMH,
 *   double adapt_size_cooling = 0.99, double max_scaling_sd = 50.0)
 *   (adaptMCMC_source.cpp:75-219), run once per chain; and rwMCMC (:230-316), which is the
 *   same loop with adapt_size_start = adapt_shape_start = 0, weight = 0, window = 0.
 * ===================================================================================== */
typedef struct mcmc_rwm_config {
    int32_t d;                     /* length of init_theta                                   */
    int32_t n_iterations;
    int64_t n_chains;              /* chains in THIS call                                    */
    int64_t chain_offset;          /* global id of local chain 0 (Philox counter)            */
    int32_t adapt_size_start;      /* 0 disables size adaptation  (:31,117)                  */
    int32_t adapt_shape_start;     /* 0 disables shape adaptation (:31,131)                  */
    int32_t acceptance_window;     /* 0 = off; 1..65536 (above 64 the outcomes live in a per-chain ring in HBM) */
    int32_t thin;                  /* keep iteration i when i % thin == 0 (>= 1)             */
    double acceptance_rate_weight;
    double adapt_size_cooling;     /* reference default 0.99                                 */
    double max_scaling_sd;         /* reference default 50.0                                 */
    uint64_t seed;                 /* Philox key                                             */
    int32_t rng_mode;              /* MCMC_RNG_PHILOX | MCMC_RNG_INJECTED                     */
    int32_t kernel;                /* MCMC_KERNEL_AUTO (0) | _LANE | _BLOCK | _PAIR: which device kernel runs */
    int32_t iter_offset;           /* resume: iterations already done (0 = fresh start). This call runs iterations
                                      iter_offset+1 .. iter_offset+n_iterations and needs io->d_state_in;
                                      iter_offset + n_iterations < 2^27 (the status word holds the iteration)   */
    int32_t factor_update;         /* MCMC_FACTOR_EXACT (0, the reference) | MCMC_FACTOR_RANK1                    */
    int32_t refresh_every;         /* MCMC_FACTOR_RANK1: full refactorization when iteration % refresh_every == 0 (>= 1) */
    int32_t trace_layout;          /* MCMC_TRACE_SOA (0) | MCMC_TRACE_CHAIN_MAJOR                                 */
    int32_t reserved_;             /* 0                                                                          */
} mcmc_rwm_config;

typedef struct mcmc_rwm_io {
    /* inputs */
    const double *h_init_theta;    /* [d] start shared by all chains, used when d_init == NULL */
    const double *d_init;          /* [d][n_chains] per-chain start, or NULL                 */
    const double *h_covmat;        /* [d*d] column-major, shared (arma::mat covmat)          */
    const double *d_Z;             /* INJECTED: [n_iterations][d][n_chains] normals, Y(k) of iteration i at ((i-1)*d+k)*n_chains+c (:149-152) */
    const double *d_U;             /* INJECTED: [n_iterations][n_chains] uniforms A (:174)   */
    /* outputs, each may be NULL */
    double *d_trace;               /* theta_trace: [n_iterations/thin][d][n_chains] (:185)   */
    double *d_sigma_trace;         /* sigma_empirical: [n_iterations/thin][d*d][n_chains], column-major d x d slices (:214) */
    double *d_final_theta;         /* [d][n_chains]                                          */
    double *d_final_mean;          /* theta_mean: [d][n_chains] (:212)                       */
    double *d_final_cov;           /* covmat_empirical: [d*d][n_chains] column-major (:211)  */
    double *d_acceptance_rate;     /* [n_chains] (:218)                                      */
    int32_t *d_status;             /* [n_chains] per-chain status word                        */
    /* parity diagnostics, each may be NULL: one value per iteration and chain, [n_iterations][n_chains] */
    uint8_t *d_accepted;           /* is_accepted                                            */
    double *d_log_acceptance;      /* log_acceptance after :168-170                           */
    double *d_log_A;               /* log(A), :175                                           */
    double *d_scaling_sd;          /* scaling_sd after the adaptation block :117-139          */
    double *d_scaling_multiplier;  /* scaling_multiplier after the adaptation block (:123)    */
    double *d_rate_trace;          /* acceptance_rate after :188-207                          */
    /* summary, may be NULL */
    int32_t *d_shape_iterations;   /* [n_chains] iterations whose proposal used scaling_sd^2 covmat_empirical (:131-139);
                                      counted from iteration 1, across resumed launches                   */
    /* checkpoint / resume, each may be NULL: [mcmc_rwm_state_doubles_window(d, acceptance_window)][n_chains]
       (= mcmc_rwm_state_doubles(d) rows unless acceptance_window > 64 with acceptance_rate_weight = 0) */
    const double *d_state_in;      /* chain state a previous call wrote (needs cfg->iter_offset = its end)  */
    double *d_state_out;           /* chain state after the last iteration of this call                     */
} mcmc_rwm_io;

int mcmc_adapt_rwm_batch(const mcmc_target *target, const mcmc_rwm_config *cfg,
                         const mcmc_rwm_io *io, void *stream);

/*
 * Checkpoint / resume of chain state (no counterpart in the reference: its loop state lives in
 * local variables and is lost on return or interrupt, adaptMCMC_source.cpp:84-105,111-114).
 * d_state_out receives, per chain, everything the loop carries from one iteration to the next:
 * theta_current, theta_mean, covmat_empirical and the proposal factor (packed upper triangles),
 * target(theta_current), acceptance_rate and the acceptance window, scaling_sd, the proposal
 * scale / determinant term, the status word and the iteration count.  A later call with the
 * same target, covmat, adaptation parameters, seed, chain_offset, thin and kernel, with
 * iter_offset = iterations done and d_state_in = that array, continues the run: its trace rows,
 * rates and final state are bit-identical to those of one uninterrupted call (Philox counters
 * are absolute iteration numbers; injected streams d_Z / d_U, the diagnostics arrays and d_trace
 * hold this call's iterations only; a trace row is kept when the ABSOLUTE iteration is a
 * multiple of thin, so this call writes (iter_offset+n_iterations)/thin - iter_offset/thin rows).
 * The state is in the writing kernel's own representation and carries its tag: a state resumed by
 * another kernel, or at another iter_offset, stops every chain with MCMC_CHAIN_BAD_STATE.
 * The array is plain doubles: copy it to the host and write it to disk to survive the process.
 */
int64_t mcmc_rwm_state_doubles(int32_t d);
/* The same with acceptance_window > 64 and acceptance_rate_weight = 0: the state then also carries the ring of
 * outcomes (2 doubles per 64 outcomes) behind the mcmc_rwm_state_doubles(d) doubles above. */
int64_t mcmc_rwm_state_doubles_window(int32_t d, int32_t acceptance_window);

/* Fills the N(0,1) / U(0,1) arrays that mcmc_adapt_rwm_batch would draw in MCMC_RNG_PHILOX
 * mode for the same (seed, chain_offset, n_chains, n_iterations, d), in the d_Z / d_U layout
 * above.  Lets a checker replay a Philox run through the reference with identical streams.
 * No counterpart in the reference (its RNG is std::mt19937, adaptMCMC_source.cpp:78-80). */
int mcmc_rwm_streams(uint64_t seed, int64_t chain_offset, int64_t n_chains, int32_t n_iterations,
                     int32_t d, double *d_Z, double *d_U, void *stream);

/* Per-chain start points theta0[k][c] = mean[k] + sd * N(0,1) from Philox(seed) (stream tag
 * "init"), layout [d][n_chains].  No counterpart in the reference (callers pass init_theta). */
int mcmc_init_normal(uint64_t seed, int64_t chain_offset, int64_t n_chains, int32_t d,
                     const double *h_mean, double sd, double *d_out, void *stream);

/* ----- single chain, host pointers, reference layout (the Rcpp drop-in) ---------------
 * replaces: adaptMCMC (adaptMCMC_source.cpp:75-219).
 *   theta_trace      n_iterations x d, column-major                     (:92,185,218)
 *   sigma_empirical  d x d x n_iterations (arma::cube) or NULL          (:93,214,218)
 *   acceptance_rate  scalar                                              (:218)
 *   h_Z / h_U        optional injected streams (Z[(i-1)*d+k], U[i-1]); both NULL = Philox
 *                    keyed by seedMH (converted like std::mt19937's seed: to uint64, :78)
 *   info             must be > 0 (the reference computes i % info, :142); info_log, if not
 *                    NULL, receives for every i with i % info == 0 the three numbers the
 *                    reference prints at :144 -- acceptance_rate, scaling_sd,
 *                    scaling_multiplier -- as info_log[3*(i/info - 1) + 0..2]
 *   fail_iter        0, or the 1-based iteration at which chol failed (then the call
 *                    returns MCMC_E_CHAIN_FAILED; rows from that iteration on are NaN)
 */
int mcmc_adaptMCMC(const mcmc_target *target, int32_t d, const double *h_init_theta,
                   const double *h_covmat, int32_t n_iterations, int32_t adapt_size_start,
                   double acceptance_rate_weight, int32_t acceptance_window,
                   int32_t adapt_shape_start, int32_t info, double seedMH,
                   double adapt_size_cooling, double max_scaling_sd, const double *h_Z,
                   const double *h_U, double *h_theta_trace, double *h_sigma_empirical,
                   double *h_acceptance_rate, double *h_info_log, int32_t *fail_iter);

/* replaces: List rwMCMC(Function target, arma::vec init_theta, arma::mat covmat,
 *   int n_iterations, int info, double seedMH) (adaptMCMC_source.cpp:230-316);
 *   returns theta_trace (n x d column-major) and acceptance_rate (:315). */
int mcmc_rwMCMC(const mcmc_target *target, int32_t d, const double *h_init_theta,
                const double *h_covmat, int32_t n_iterations, int32_t info, double seedMH,
                const double *h_Z, const double *h_U, double *h_theta_trace,
                double *h_acceptance_rate, int32_t *fail_iter);

/* ----- many chains, host pointers: the batch path for an R caller -----------------------
 * replaces: mclapply_RNG(X, FUN) (MCMCutilty.R:103-149) around FUN = function(x) adaptMCMC(target, x, covmat, ...)
 *   -- the reference's only way to run several chains: one adaptMCMC call (adaptMCMC_source.cpp:75-219) per element
 *   of X in a socket-cluster worker, results returned as a list -- by ONE call that runs all chains on the GPU.
 *   h_init_theta     init_per_chain != 0: d x n_chains column-major, column c = X[[c]] (chain c's init_theta);
 *                    init_per_chain == 0: [d], one start for all chains
 *   h_theta_trace    n_chains matrices back to back, chain c at h_theta_trace + c*n_kept*d: n_kept x d column-major,
 *                    n_kept = n_iterations / thin -- the `trace` element melt_trace (MCMCutilty.R:19-35) reads per
 *                    chain.  May be NULL.
 *   h_acceptance_rate [n_chains] or NULL (:218); h_covmat_empirical d x d x n_chains (the last sigma_empirical slice
 *                    of every chain, :211,214) or NULL; h_status [n_chains] per-chain status words or NULL (a chain
 *                    whose chol failed has NaN rows from that iteration on; the call itself still returns MCMC_OK).
 *   seedMH           Philox key, converted as at :78; chain c draws from (seedMH, chain id c), so -- unlike the
 *                    reference, where every worker handed the same seedMH repeats one std::mt19937 stream -- the
 *                    chains of one call are independent, and a chain's results do not depend on n_chains.
 * Same kernels and kernel choice as mcmc_adapt_rwm_batch (bit-identical traces); chains are processed in slabs of at
 * most 4 GiB of device trace, so the host result may be larger than the GPU's memory.  Synchronous. */
int mcmc_adaptMCMC_chains(const mcmc_target *target, int32_t d, int64_t n_chains, const double *h_init_theta,
                          int32_t init_per_chain, const double *h_covmat, int32_t n_iterations,
                          int32_t adapt_size_start, double acceptance_rate_weight, int32_t acceptance_window,
                          int32_t adapt_shape_start, int32_t info, double seedMH, double adapt_size_cooling,
                          double max_scaling_sd, int32_t thin, double *h_theta_trace, double *h_acceptance_rate,
                          double *h_covmat_empirical, int32_t *h_status);

/* =====================================================================================
 * Truncated multivariate normal Gibbs sampler, many chains
 * replaces: arma::mat rtmvnorm_gibbs(int n, arma::vec mu, arma::mat sigma, arma::vec lower,
 *   arma::vec upper, arma::vec init_state) (tmvrnormGibbs.cpp:91-142), run once per chain.
 * ===================================================================================== */
typedef struct mcmc_gibbs_config {
    int32_t d;
    int32_t n_sweeps;              /* the reference's n                                       */
    int64_t n_chains;
    int64_t chain_offset;
    int32_t thin;                  /* keep sweep j (1-based) when j % thin == 0               */
    int32_t burn;                  /* sweeps excluded from the running moments                */
    uint64_t seed;
    int32_t rng_mode;
    int32_t tail_safe;             /* 0 = the reference formula (:130-132) verbatim;
                                      1 = same draw computed from whichever tail keeps
                                          precision (needed when a bound is > ~8 sd out,
                                          where the reference returns +-Inf)                  */
    int32_t sweep_offset;          /* resume: sweeps already done (0 = fresh start).  The chain state of the
                                      Gibbs sampler is x alone: pass the previous call's d_final_state as d_init;
                                      Philox counters and thinning continue from sweep_offset + 1, the running
                                      moments (d_mean, d_m2, burn) cover this call's sweeps only              */
} mcmc_gibbs_config;

typedef struct mcmc_gibbs_io {
    const double *h_mu;            /* [d]                                                    */
    const double *h_sigma;         /* [d*d] column-major covariance                           */
    const double *h_lower;         /* [d], -INFINITY allowed                                  */
    const double *h_upper;         /* [d], +INFINITY allowed                                  */
    const double *h_init_state;    /* [d] shared start, used when d_init == NULL              */
    const double *d_init;          /* [d][n_chains] or NULL                                   */
    const double *d_U;             /* INJECTED: [n_sweeps][d][n_chains]; U(l), l = j*d+i, of chain c at (j*d+i)*n_chains+c (:97,132-133) */
    double *d_trace;               /* [n_sweeps/thin][d][n_chains] or NULL (:137)             */
    double *d_final_state;         /* [d][n_chains] or NULL                                   */
    double *d_mean;                /* running mean  of x over sweeps > burn: [d][n_chains] or NULL */
    double *d_m2;                  /* running sum of squared deviations (Welford M2): [d][n_chains] or NULL */
    int32_t *d_status;             /* [n_chains] or NULL                                      */
} mcmc_gibbs_io;

int mcmc_tmvn_gibbs_batch(const mcmc_gibbs_config *cfg, const mcmc_gibbs_io *io, void *stream);

/*
 * The same call for callers that cannot allocate device memory (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  No counterpart in the reference.  mcmc_tmvn_gibbs_batch keeps its shared
 * parameter block (regression rows P_i, sd_i, bounds: tmvrnormGibbs.cpp:100-113) in a hipMallocAsync'ed buffer; this
 * variant keeps it in a static 160 KB device array of the library (one copy per GPU, held until the process ends).
 * Both fill the block through the arguments of small kernels, so neither waits for the stream; this one also calls no
 * hipMalloc / hipFree / hipMemcpy: every step is a kernel launch on `stream`.  Same kernels, bit-identical results.
 * The FIRST call on a device looks up the array's address (hipGetSymbolAddress, which loads the code object): make one
 * warm-up call per device outside hipStreamBeginCapture; every later call is launches only and can be captured.
 * Limits: d <= mcmc_tmvn_gibbs_noalloc_max_dim() (= 139: the block of a larger d does not fit the array; MCMC_E_UNSUPPORTED).
 * Every call on a GPU uses that GPU's one array.  Host threads are serialised per GPU while a call stages and queues
 * its block (a mutex), but calls that overlap ON THE DEVICE -- on two streams while the first call's kernel is still
 * running -- race; calls on different GPUs do not.
 */
int mcmc_tmvn_gibbs_batch_noalloc(const mcmc_gibbs_config *cfg, const mcmc_gibbs_io *io, void *stream);
int mcmc_tmvn_gibbs_noalloc_max_dim(void);

/* The shared precompute of tmvrnormGibbs.cpp:100-113 (regression rows P_i and conditional
 * sd_i), obtained from the precision matrix H = sigma^-1: P_i = -H[i,-i]/H_ii,
 * sd_i = 1/sqrt(H_ii).  h_P is d rows of d-1 entries (row-major), h_sd is [d]. */
int mcmc_tmvn_precompute(int32_t d, const double *h_sigma, double *h_P, double *h_sd);

/* Fills the uniforms a MCMC_RNG_PHILOX Gibbs run would use, in the d_U layout above. */
int mcmc_gibbs_streams(uint64_t seed, int64_t chain_offset, int64_t n_chains, int32_t n_sweeps,
                       int32_t d, double *d_U, void *stream);

/* single chain, host pointers, reference layout.
 * replaces: rtmvnorm_gibbs (tmvrnormGibbs.cpp:91-142); h_trace is n x d column-major;
 * h_U optional (n*d uniforms in the reference's order), NULL = Philox(seed). */
int mcmc_rtmvnorm_gibbs(int32_t n, int32_t d, const double *h_mu, const double *h_sigma,
                        const double *h_lower, const double *h_upper, const double *h_init_state,
                        const double *h_U, uint64_t seed, double *h_trace);

/* many chains, host pointers: replaces mclapply_RNG(X, FUN) (MCMCutilty.R:103-149) around
 * FUN = function(x) rtmvnorm_gibbs(n, mu, sigma, lower, upper, x) (tmvrnormGibbs.cpp:91-142).  h_init_state is
 * d x n_chains column-major (init_per_chain != 0) or [d]; h_trace holds n_chains matrices back to back, each
 * (n / thin) x d column-major; chain c draws its uniforms from Philox (seed, chain id c); h_status [n_chains] or NULL.
 * tail_safe as in mcmc_gibbs_config.  Same kernels as mcmc_tmvn_gibbs_batch (bit-identical); synchronous. */
int mcmc_rtmvnorm_gibbs_chains(int32_t n, int32_t d, int64_t n_chains, const double *h_mu, const double *h_sigma,
                               const double *h_lower, const double *h_upper, const double *h_init_state,
                               int32_t init_per_chain, uint64_t seed, int32_t thin, int32_t tail_safe,
                               double *h_trace, int32_t *h_status);

/* replaces: NumericVector rtnorm_gibbs(int n, double mu, double sigma, double a, double b)
 * (tmvrnormGibbs.cpp:49-70); h_F optional (the runif(n) draws), NULL = Philox(seed). */
int mcmc_rtnorm_gibbs(int32_t n, double mu, double sigma, double a, double b, const double *h_F,
                      uint64_t seed, double *h_out);
/* device-resident variant of the same draw: d_F may be NULL (Philox), d_out is [n]. */
int mcmc_rtnorm_batch(int64_t n, double mu, double sigma, double a, double b, const double *d_F,
                      uint64_t seed, int64_t offset, int32_t tail_safe, double *d_out, void *stream);

/* Index-shedding helpers (host, column-major).
 * replaces: sub1 (tmvrnormGibbs.cpp:12-16)  out = x[-i,-i], (d-1) x (d-1)
 *           sub2 (:20-23)                    out = x[a,-b],  1 x (nc-1)
 *           negSubCol (:27-30), negSubRow (:34-37)  out = x[-i]
 * i, a, b are 0-based as in the reference.  On the device these vanish into indexing. */
int mcmc_sub1(int32_t d, const double *h_x, int32_t i, double *h_out);
int mcmc_sub2(int32_t nr, int32_t nc, const double *h_x, int32_t a, int32_t b, double *h_out);
int mcmc_negSubCol(int32_t n, const double *h_x, int32_t i, double *h_out);
int mcmc_negSubRow(int32_t n, const double *h_x, int32_t i, double *h_out);

/* =====================================================================================
 * Cross-chain diagnostics (no counterpart in the reference: coda is loaded at
 * MCMCutilty.R:11 but never called).  The partial sums are what one RCCL all-reduce
 * carries between GPUs.
 * ===================================================================================== */
/* Per-chain mean and unbiased variance over kept draws t >= burn_kept of a
 * [n_kept][d][n_chains] trace: d_mean, d_var are [d][n_chains]. */
int mcmc_chain_moments(const double *d_trace, int32_t n_kept, int32_t burn_kept, int32_t d,
                       int64_t n_chains, double *d_mean, double *d_var, void *stream);
/* Reduces [d][n_chains] per-chain means/variances to d_partial[3*d + 1]:
 * sum_c m, sum_c m^2, sum_c s^2 (per coordinate) and the chain count. */
int mcmc_rhat_partial(const double *d_mean, const double *d_var, int32_t d, int64_t n_chains,
                      double *d_partial, void *stream);
/* Reduces per-chain theta_mean [d][n_chains] and covmat_empirical [d*d][n_chains] (d_final_mean / d_final_cov of
 * mcmc_adapt_rwm_batch) to d_partial[2*d*d + d + 1]: sum_c C_c, sum_c m_c m_c^T (both d x d column-major),
 * sum_c m_c, and the chain count -- the terms of the pooled covariance
 * mean_c(C_c + m_c m_c^T) - mbar mbar^T over chains of equal length. */
int mcmc_pooled_cov_partial(const double *d_mean, const double *d_cov, int32_t d, int64_t n_chains,
                            double *d_partial, void *stream);
/* Per-chain, per-coordinate effective sample size (Geyer initial positive sequence on the
 * autocorrelation up to max_lag): d_ess is [d][n_chains]. */
int mcmc_chain_ess(const double *d_trace, int32_t n_kept, int32_t burn_kept, int32_t d,
                   int64_t n_chains, int32_t max_lag, double *d_ess, void *stream);
/* The same, and d_capped [d][n_chains] (may be NULL) receives 1 where the sum of autocorrelation pairs was still
 * positive at max_lag -- the cap, not Geyer's stopping rule, ended it, so that chain's ESS is an UPPER bound
 * (tau is under-stated) -- and 0 elsewhere.  A constant series has ESS 0, capped 0. */
int mcmc_chain_ess_capped(const double *d_trace, int32_t n_kept, int32_t burn_kept, int32_t d,
                          int64_t n_chains, int32_t max_lag, double *d_ess, int32_t *d_capped, void *stream);

/* =====================================================================================
 * Outside the two samplers: the multivariate normal density helpers of
 * deprecated/truncGaussian.cpp and the toy walkers of randomWalkers.cpp.
 * ===================================================================================== */
/* replaces: double mvrGaussian_pdf(arma::vec x, arma::vec mu, arma::mat sigma, bool do_log = true)
 * (deprecated/truncGaussian.cpp:11-33): rootTerm * exp(-0.5 (x-mu)^T sigma^-1 (x-mu)), logged afterwards if do_log.
 * Batch form: d_x is [d][n] (n points), d_out [n].  Waits for the launch (its parameter block is a local). */
int mcmc_mvrGaussian_pdf_batch(int32_t d, int64_t n, const double *d_x, const double *h_mu, const double *h_sigma,
                               int32_t do_log, double *d_out, void *stream);
int mcmc_mvrGaussian_pdf(int32_t d, const double *h_x, const double *h_mu, const double *h_sigma, int32_t do_log, double *out);
/* replaces: double tmvrGaussian_pdf(NumericVector x, NumericVector mu, NumericMatrix sigma, NumericVector lower,
 * NumericVector upper) (deprecated/truncGaussian.cpp:38-78): 0 outside [lower, upper] (:45-63), else
 * mvrGaussian_pdf(x, mu, sigma) / cdf -- with the reference's own default do_log = true at :73, i.e. the LOG
 * density over cdf.  `cdf` is what the reference gets from mvtnorm::pmvnorm(lower, upper, mu, sigma) (:68-75), a
 * third-party randomized integral: pass that value, or mcmc_pmvnorm's. */
int mcmc_tmvrGaussian_pdf_batch(int32_t d, int64_t n, const double *d_x, const double *h_mu, const double *h_sigma,
                                const double *h_lower, const double *h_upper, double cdf, double *d_out, void *stream);
int mcmc_tmvrGaussian_pdf(int32_t d, const double *h_x, const double *h_mu, const double *h_sigma, const double *h_lower,
                          const double *h_upper, double cdf, double *out);
/* P(lower <= X <= upper), X ~ N(mu, sigma), d <= 32, by Genz' separation-of-variables estimator over n_points Philox
 * points on the device; *std_err (may be NULL) is the Monte Carlo standard error.  Stands in for the
 * mvtnorm::pmvnorm call of deprecated/truncGaussian.cpp:74 -- same estimand, other random numbers: PARITY UNPINNED. */
int mcmc_pmvnorm(int32_t d, const double *h_lower, const double *h_upper, const double *h_mu, const double *h_sigma,
                 int64_t n_points, uint64_t seed, double *prob, double *std_err);
/* replaces: NumericMatrix random_walk2d(int n) / random_walk3d(int n) (randomWalkers.cpp:9-37, :40-74): row 0 is the
 * origin, step i moves +-1 along axis (r-1)/2 with r = 1 + floor(runif * 2 dim).  h_out is n x dim column-major;
 * h_U optional (the n-1 runif draws in order), NULL = Philox(seed).  Batch form: d_U [n-1][n_walkers] or NULL,
 * d_out [n][dim][n_walkers], walker w has global id walker_offset + w. */
int mcmc_random_walk2d(int32_t n, const double *h_U, uint64_t seed, double *h_out);
int mcmc_random_walk3d(int32_t n, const double *h_U, uint64_t seed, double *h_out);
int mcmc_random_walk_batch(int32_t dim, int32_t n, int64_t n_walkers, int64_t walker_offset, const double *d_U, uint64_t seed,
                           double *d_out, void *stream);
/* replaces: NumericVector random_walkGraph(NumericMatrix graph, int init_pos, int n) (randomWalkers.cpp:80-100): from
 * node p the next node is RcppArmadillo::sample(0..n_nodes-1, 1, false, graph(p, _)) -- probabilities normalised, sorted
 * in descending order, first position whose running sum reaches unif_rand() (rows with equal weights: the tie order
 * of Armadillo's sort is unspecified, here index order).  h_graph is n_nodes x n_nodes column-major, nodes 0-based;
 * h_U optional (n-1 draws).  Batch form: d_U [n-1][n_walkers] or NULL, d_out [n][n_walkers]; waits for the launch. */
int mcmc_random_walkGraph(int32_t n_nodes, const double *h_graph, int32_t init_pos, int32_t n, const double *h_U, uint64_t seed,
                          double *h_out);
int mcmc_random_walkGraph_batch(int32_t n_nodes, const double *h_graph, int32_t init_pos, int32_t n, int64_t n_walkers,
                                int64_t walker_offset, const double *d_U, uint64_t seed, double *d_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MCMC_AMD_H */
