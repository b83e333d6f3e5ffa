// rcpp_shim.cpp -- the Rcpp side of the drop-in (NOT compiled in this repository: the build image
// has no R / Rcpp / RcppArmadillo).  It carries the reference's eight exported signatures
// verbatim (adaptMCMC_source.cpp:74-76,229-230; tmvrnormGibbs.cpp:11-12,19-20,26-27,33-34,48-49,
// 90-91) and each body is: marshal -> one extern "C" call into libmcmc_amd.so -> wrap.
// Two more exports, adaptMCMC_chains and rtmvnorm_gibbs_chains (end of file), do the job of
// mclapply_RNG(X, FUN) (MCMCutilty.R:103-149) in one call: all chains of X in one GPU launch, returned as the
// list of chains melt_trace (MCMCutilty.R:19-35) reads.  That is the fast path: a single chain per call occupies
// one lane of a 256-CU GPU and runs ~10x slower than a CPU core.
//
// Build inside R:  Sys.setenv(PKG_LIBS = "-L<repo>/mcmc_amd -lmcmc_amd -Wl,-rpath,<repo>/mcmc_amd",
//                             PKG_CPPFLAGS = "-I<repo>/include");  Rcpp::sourceCpp("rcpp_shim.cpp")
//
// `target` is either a list describing a device target
//     list(kind = "gaussian", mean = m, precision = H) | list(kind = "goldstein_price") |
//     list(kind = "banana", B = 0.03) |
//     list(kind = "hip", source = "<HIP device source of mcmc_user_target>", params = c(...))
// or an R function.  A GPU cannot call an R closure (adaptMCMC_source.cpp:97,158), so functions are
// refused here with a message: restate the closure as HIP source (kind = "hip", compiled once per
// call by mcmc_target_compile, ~1-3 s) or keep sourcing the reference's own adaptMCMC_source.cpp.
#include <RcppArmadillo.h>
// [[Rcpp::depends(RcppArmadillo)]]
#include "mcmc_amd.h"
using namespace Rcpp;

namespace {
struct TargetDesc {
    mcmc_target t{};
    std::vector<double> prm;
    void *user = nullptr;
    TargetDesc() = default;
    TargetDesc(const TargetDesc &) = delete;
    TargetDesc(TargetDesc &&o) : t(o.t), prm(std::move(o.prm)), user(o.user) { o.user = nullptr; }
    ~TargetDesc()
    {
        if (user) mcmc_target_destroy(user);
    }
};
TargetDesc make_target(SEXP target, int d)
{
    if (Rf_isFunction(target))
        stop("mcmc_amd: `target` is an R function; pass list(kind=...) (kind = \"hip\" takes the function as HIP source) "
             "for the GPU path or use the reference sampler");
    List l(target);
    std::string kind = as<std::string>(l["kind"]);
    TargetDesc out;
    if (kind == "gaussian") {
        arma::vec m = as<arma::vec>(l["mean"]);
        arma::mat H = as<arma::mat>(l["precision"]);
        out.prm.assign(m.begin(), m.end());
        out.prm.insert(out.prm.end(), H.begin(), H.end()); // column-major, as the C ABI expects
        out.t.kind = MCMC_TARGET_GAUSSIAN_PREC;
    } else if (kind == "goldstein_price") {
        out.t.kind = MCMC_TARGET_GOLDSTEIN_PRICE;
    } else if (kind == "banana") {
        out.prm.push_back(l.containsElementNamed("B") ? as<double>(l["B"]) : 0.03);
        out.t.kind = MCMC_TARGET_BANANA;
    } else if (kind == "hip") {
        std::string src = as<std::string>(l["source"]);
        if (l.containsElementNamed("params")) out.prm = as<std::vector<double>>(l["params"]);
        std::vector<char> log(1 << 16);
        if (mcmc_target_compile(src.c_str(), d, (int)out.prm.size(), &out.user, log.data(), (int64_t)log.size()) != MCMC_OK)
            stop("mcmc_amd: %s\n%s", mcmc_amd_last_error(), log.data());
        out.t.kind = MCMC_TARGET_USER;
        out.t.user = out.user;
    } else {
        stop("mcmc_amd: unknown target kind '%s'", kind);
    }
    out.t.n_params = (int)out.prm.size();
    out.t.h_params = out.prm.empty() ? nullptr : out.prm.data();
    return out;
}
void check(int rc)
{
    if (rc != MCMC_OK) stop("mcmc_amd: %s", mcmc_amd_last_error());
}
} // namespace

// [[Rcpp::export]]
List adaptMCMC(SEXP target, arma::vec init_theta, arma::mat covmat, int n_iterations, int adapt_size_start,
               double acceptance_rate_weight, int acceptance_window, int adapt_shape_start, int info, double seedMH,
               double adapt_size_cooling = 0.99, double max_scaling_sd = 50.0)
{
    const int d = init_theta.n_elem;
    TargetDesc td = make_target(target, d);
    arma::mat theta_trace(n_iterations, d);
    arma::cube sigma_empirical(d, d, n_iterations);
    double acceptance_rate = 0.0;
    std::vector<double> info_log(3 * std::max(1, info > 0 ? n_iterations / info : 1));
    int fail_iter = 0;
    check(mcmc_adaptMCMC(&td.t, d, init_theta.memptr(), covmat.memptr(), n_iterations, adapt_size_start,
                         acceptance_rate_weight, acceptance_window, adapt_shape_start, info, seedMH,
                         adapt_size_cooling, max_scaling_sd, nullptr, nullptr, theta_trace.memptr(),
                         sigma_empirical.memptr(), &acceptance_rate, info_log.data(), &fail_iter));
    for (int i = info; i <= n_iterations; i += info) { // the lines the reference prints at :142-146
        const double *r = &info_log[3 * (i / info - 1)];
        Rcout << "At iter: " << i << ", acceptance rate is: " << r[0] << ", scaling_sd: " << r[1]
              << ", scaling_multiplier: " << r[2] << std::endl;
    }
    return List::create(Named("theta_trace") = theta_trace, Named("sigma_empirical") = sigma_empirical,
                        Named("acceptance_rate") = acceptance_rate);
}

// [[Rcpp::export]]
List rwMCMC(SEXP target, arma::vec init_theta, arma::mat covmat, int n_iterations, int info, double seedMH)
{
    const int d = init_theta.n_elem;
    TargetDesc td = make_target(target, d);
    arma::mat theta_trace(n_iterations, d);
    double acceptance_rate = 0.0;
    int fail_iter = 0;
    check(mcmc_rwMCMC(&td.t, d, init_theta.memptr(), covmat.memptr(), n_iterations, info, seedMH, nullptr, nullptr,
                      theta_trace.memptr(), &acceptance_rate, &fail_iter));
    return List::create(Named("theta_trace") = theta_trace, Named("acceptance_rate") = acceptance_rate);
}

// [[Rcpp::export]]
arma::mat sub1(arma::mat x, int i)
{
    arma::mat out(x.n_rows - 1, x.n_cols - 1);
    check(mcmc_sub1(x.n_rows, x.memptr(), i, out.memptr()));
    return out;
}

// [[Rcpp::export]]
arma::mat sub2(arma::mat x, int a, int b)
{
    arma::mat out(1, x.n_cols - 1);
    check(mcmc_sub2(x.n_rows, x.n_cols, x.memptr(), a, b, out.memptr()));
    return out;
}

// [[Rcpp::export]]
arma::vec negSubCol(arma::vec x, int i)
{
    arma::vec out(x.n_elem - 1);
    check(mcmc_negSubCol(x.n_elem, x.memptr(), i, out.memptr()));
    return out;
}

// [[Rcpp::export]]
arma::rowvec negSubRow(arma::rowvec x, int i)
{
    arma::rowvec out(x.n_elem - 1);
    check(mcmc_negSubRow(x.n_elem, x.memptr(), i, out.memptr()));
    return out;
}

// [[Rcpp::export]]
NumericVector rtnorm_gibbs(int n, double mu, double sigma, double a, double b)
{
    // the reference draws runif(n) from R's RNG (tmvrnormGibbs.cpp:52); doing the same here and
    // injecting the draws keeps set.seed() reproducibility identical to the reference
    NumericVector F = runif(n);
    NumericVector out(n);
    check(mcmc_rtnorm_gibbs(n, mu, sigma, a, b, F.begin(), 0, out.begin()));
    return out;
}

// [[Rcpp::export]]
arma::mat rtmvnorm_gibbs(int n, arma::vec mu, arma::mat sigma, arma::vec lower, arma::vec upper, arma::vec init_state)
{
    const int d = mu.n_elem;
    NumericVector U = runif(n * d); // as tmvrnormGibbs.cpp:97, consumed as U[j*d + i]
    arma::mat trace(n, d);
    check(mcmc_rtmvnorm_gibbs(n, d, mu.memptr(), sigma.memptr(), lower.memptr(), upper.memptr(), init_state.memptr(),
                              U.begin(), 0, trace.memptr()));
    return trace;
}

// ---- many chains in one call ------------------------------------------------------------------
// X is what mclapply_RNG's first argument is in the reference (MCMCutilty.R:103): a list of per-chain start
// vectors.  The return value is what mclapply_RNG returns -- a list with one element per chain -- and every element
// carries `trace` (iterations x parameters), which is what melt_trace indexes (MCMCutilty.R:22-24), beside
// `acceptance_rate` and the per-chain `status` word (0 = ran to completion).
namespace {
arma::mat starts_as_columns(List X, int &d)
{
    if (X.size() == 0) stop("mcmc_amd: X is empty");
    d = as<arma::vec>(X[0]).n_elem;
    arma::mat init(d, X.size());
    for (int c = 0; c < X.size(); ++c) {
        arma::vec x = as<arma::vec>(X[c]);
        if ((int)x.n_elem != d) stop("mcmc_amd: X[[%d]] has length %d, X[[1]] has %d", c + 1, (int)x.n_elem, d);
        init.col(c) = x;
    }
    return init;
}
} // namespace

// [[Rcpp::export]]
List adaptMCMC_chains(SEXP target, List X, arma::mat covmat, int n_iterations, int adapt_size_start,
                      double acceptance_rate_weight, int acceptance_window, int adapt_shape_start, int info, double seedMH,
                      double adapt_size_cooling = 0.99, double max_scaling_sd = 50.0, int thin = 1)
{
    int d = 0;
    arma::mat init = starts_as_columns(X, d);
    const int n_chains = init.n_cols, n_kept = thin > 0 ? n_iterations / thin : 0;
    TargetDesc td = make_target(target, d);
    arma::cube traces(n_kept, d, n_chains); // slice c = chain c's n_kept x d matrix, column-major: the C ABI's layout
    arma::vec acc(n_chains);
    IntegerVector status(n_chains);
    check(mcmc_adaptMCMC_chains(&td.t, d, n_chains, init.memptr(), 1, covmat.memptr(), n_iterations, adapt_size_start,
                                acceptance_rate_weight, acceptance_window, adapt_shape_start, info, seedMH,
                                adapt_size_cooling, max_scaling_sd, thin, traces.memptr(), acc.memptr(), nullptr,
                                status.begin()));
    List out(n_chains);
    for (int c = 0; c < n_chains; ++c)
        out[c] = List::create(Named("trace") = DataFrame(wrap(arma::mat(traces.slice(c)))), Named("acceptance_rate") = acc[c],
                              Named("status") = status[c]);
    return out;
}

// [[Rcpp::export]]
List rtmvnorm_gibbs_chains(int n, arma::vec mu, arma::mat sigma, arma::vec lower, arma::vec upper, List X, double seed = 0,
                           int thin = 1)
{
    int d = 0;
    arma::mat init = starts_as_columns(X, d);
    const int n_chains = init.n_cols, n_kept = thin > 0 ? n / thin : 0;
    arma::cube traces(n_kept, d, n_chains);
    IntegerVector status(n_chains);
    check(mcmc_rtmvnorm_gibbs_chains(n, d, n_chains, mu.memptr(), sigma.memptr(), lower.memptr(), upper.memptr(),
                                     init.memptr(), 1, (uint64_t)seed, thin, 0, traces.memptr(), status.begin()));
    List out(n_chains);
    for (int c = 0; c < n_chains; ++c)
        out[c] = List::create(Named("trace") = DataFrame(wrap(arma::mat(traces.slice(c)))), Named("status") = status[c]);
    return out;
}
