"""Development microbenchmarks (GPU): times the batch kernels with HIP events."""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mcmc_amd


def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return min(ts)


def gibbs(d, chains, sweeps, trace=True, rho=0.5):
    idx = np.arange(d)
    sigma = rho ** np.abs(idx[:, None] - idx[None, :])
    out = {}
    if trace:
        out["trace"] = torch.empty((sweeps, d, chains), dtype=torch.float64, device="cuda")
    f = lambda: mcmc_amd.tmvn_gibbs_batch(chains, sweeps, np.zeros(d), sigma, np.full(d, -1.0), np.full(d, 2.0), np.zeros(d),
                                          seed=2, trace=trace, out=out)
    ms = timeit(f)
    rate = chains * sweeps / (ms * 1e-3)
    print(f"gibbs d={d} chains={chains} sweeps={sweeps} trace={trace}: {ms:.2f} ms  {rate:.3e} chain-sweeps/s  "
          f"{rate*d*(2*d+142)/1e12:.2f} TF alg  cycles/coord/wave={ms*1e-3*2.4e9*1024/ (chains/64*sweeps*d):.0f} (if 1024 SIMDs busy)")


def rwm(d, chains, iters, trace=True):
    idx = np.arange(d)
    Sig = 0.9 ** np.abs(idx[:, None] - idx[None, :])
    H = np.linalg.inv(Sig); H = 0.5 * (H + H.T)
    tgt = mcmc_amd.Target.gaussian(np.zeros(d), H)
    init = mcmc_amd.init_normal(1, chains, np.zeros(d), 2.0)
    out = {}
    if trace:
        out["theta_trace"] = torch.empty((iters, d, chains), dtype=torch.float64, device="cuda")
    f = lambda: mcmc_amd.adapt_rwm_batch(tgt, chains, init, np.eye(d), iters, 100, 0.0, 0, 500, seed=1, trace=trace, out=out)
    ms = timeit(f)
    rate = chains * iters / (ms * 1e-3)
    print(f"rwm d={d} chains={chains} iters={iters} trace={trace}: {ms:.2f} ms  {rate:.3e} chain-iter/s  "
          f"cycles/iter/wave={ms*1e-3*2.4e9/iters/max(1,chains/64/1024):.0f}")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="*", default=["gibbs", "rwm"])  # also: block
    args = ap.parse_args()
    if "gibbs" in args.what:
        gibbs(10, 1 << 20, 200)
        gibbs(10, 1 << 20, 200, trace=False)
        gibbs(10, 1 << 16, 400)
        gibbs(4, 1 << 20, 200)
        gibbs(32, 1 << 18, 50)
        gibbs(128, 1 << 17, 10, trace=False)
    if "rwm" in args.what:
        rwm(10, 1 << 16, 2000)
        rwm(10, 1 << 16, 2000, trace=False)
        rwm(10, 1 << 18, 1000)
        rwm(4, 1 << 18, 2000)
        rwm(2, 1 << 18, 2000)


def rwm_block(d, chains, iters, ass=1, ashs=0, thin=10, kernel="auto"):
    rng = np.random.default_rng(4)
    A = rng.standard_normal((d, d))
    Sig = A @ A.T / d + np.eye(d)
    H = np.linalg.inv(Sig); H = 0.5 * (H + H.T)
    tgt = mcmc_amd.Target.gaussian(np.zeros(d), H)
    f = lambda: mcmc_amd.adapt_rwm_batch(tgt, chains, np.zeros(d), 0.01 * np.eye(d), iters, ass, 0.0, 0, ashs, seed=1, thin=thin,
                                         final=False, kernel=kernel)
    ms = timeit(f, reps=2)
    out = f()
    rate = chains * iters / (ms * 1e-3)
    # 4.5 d^2 + 8.5 d per iteration, + d^3/3 only for the iterations that factor covmat_empirical
    # (the shape phase; the size phase keeps the unscaled factor and does no factorization)
    shape = float(out["shape_iterations"].double().sum().item())
    fl = (chains * iters * (4.5 * d * d + 8.5 * d) + shape * d ** 3 / 3) / (ms * 1e-3)
    print(f"rwm-block[{kernel}] d={d} chains={chains} iters={iters} ass={ass}: {ms:.2f} ms  {rate:.3e} chain-iter/s  {fl/1e12:.2f} TF alg "
          f"({fl/78.6e12*100:.1f}% of FP64 peak; shape-phase fraction {shape/(chains*iters):.2f})")


if __name__ == "__main__" and "block" in sys.argv:
    rwm_block(100, 4096, 100)
    rwm_block(100, 768, 200)
    rwm_block(100, 4096, 100, ass=0)
    rwm_block(128, 2048, 100)
    rwm_block(32, 8192, 200)
    rwm_block(16, 16384, 200)


def rwm_block_shape(d, chains, iters, kernel="auto"):
    """shape phase from the first iterations on: small covmat, shape start right after d accepted jumps"""
    rng = np.random.default_rng(4)
    A = rng.standard_normal((d, d))
    Sig = A @ A.T / d + np.eye(d)
    H = np.linalg.inv(Sig); H = 0.5 * (H + H.T)
    tgt = mcmc_amd.Target.gaussian(np.zeros(d), H)
    f = lambda: mcmc_amd.adapt_rwm_batch(tgt, chains, np.zeros(d), (0.3 / d) * np.eye(d), iters, 0, 0.0, 0, 2 * d, seed=1, thin=10,
                                         final=False, kernel=kernel)
    ms = timeit(f, reps=2)
    out = f()
    frac = float(out["shape_iterations"].double().mean().item()) / iters
    bad = int((out["status"] != 0).sum().item())
    fl = chains * iters * (4.5 * d * d + 8.5 * d + frac * d ** 3 / 3) / (ms * 1e-3)
    print(f"rwm-block-shape[{kernel}] d={d} chains={chains} iters={iters}: {ms:.2f} ms  {chains*iters/(ms*1e-3):.3e} chain-iter/s  shape fraction {frac:.2f}  "
          f"{fl/1e12:.2f} TF alg ({fl/78.6e12*100:.1f}% of FP64 peak)  failed {bad}")


if __name__ == "__main__" and "wide" in sys.argv:  # the tile counts above d = 100: NT = 8
    rwm_block(128, 2048, 100)
    rwm_block(120, 2048, 100)
    rwm_block_shape(128, 2048, 800)
    rwm_block_shape(113, 2048, 800)


if __name__ == "__main__" and "shape" in sys.argv:
    rwm_block_shape(100, 3072, 1000)
    rwm_block_shape(64, 3072, 1000)
    rwm_block_shape(32, 6144, 1000)


GAUSS_USER_SRC = r"""
// -0.5 (x-m)^T H (x-m) written the obvious way: params = m[d], H[d*d] column-major
__device__ double mcmc_user_target(const double (&x)[MCMC_USER_D], const double *params)
{
    constexpr int D = MCMC_USER_D;
    double v[D], q = 0.0;
#pragma unroll
    for (int k = 0; k < D; ++k) v[k] = x[k] - params[k];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < D; ++j) s += params[D + j * D + i] * v[j];
        q += v[i] * s;
    }
    return -0.5 * q;
}
"""


def rwm_user(d, chains, iters):
    """C2 shape with the Gaussian target built in vs the same density compiled at run time as a user target"""
    idx = np.arange(d)
    Sig = 0.9 ** np.abs(idx[:, None] - idx[None, :])
    H = np.linalg.inv(Sig); H = 0.5 * (H + H.T)
    import time
    t0 = time.perf_counter()
    user = mcmc_amd.Target.compile(GAUSS_USER_SRC, d, np.concatenate([np.zeros(d), H.ravel(order="F")]))
    t_compile = time.perf_counter() - t0
    init = mcmc_amd.init_normal(1, chains, np.zeros(d), 2.0)
    out = {}
    for name, tgt in (("built-in", mcmc_amd.Target.gaussian(np.zeros(d), H)), ("user (hipRTC)", user)):
        f = lambda: out.update(mcmc_amd.adapt_rwm_batch(tgt, chains, init, np.eye(d), iters, 100, 0.0, 0, 500, seed=1, final=False,
                                                        out=out))
        ms = timeit(f, reps=3)
        print(f"rwm d={d} chains={chains} iters={iters} target {name}: {ms:.2f} ms  {chains*iters/(ms*1e-3):.3e} chain-iter/s"
              + (f"  (compile {t_compile:.1f} s)" if tgt is user else ""))


if __name__ == "__main__" and "user" in sys.argv:
    rwm_user(10, 1 << 16, 2000)
    rwm_user(4, 1 << 18, 1000)


def c1_single_chain():
    """BASELINE config C1: adaptMCMC_analysis.R:115-134, ONE chain x 1e5 iterations through the host-pointer
    drop-in call (PCIe-inclusive: upload of parameters, download of theta_trace and the sigma_empirical cube)."""
    import time
    n = 100000
    tgt = mcmc_amd.Target.goldstein_price()
    for store_sigma in (True, False):
        f = lambda: mcmc_amd.adaptMCMC(tgt, [1.0, 1.0], np.eye(2), n, 100, 0.0, 0, 150, 10000, 123.0, store_sigma=store_sigma)
        f()
        t0 = time.perf_counter(); r = f(); dt = time.perf_counter() - t0
        print(f"C1 single chain n={n} store_sigma={store_sigma}: GPU drop-in call {dt*1e3:.1f} ms  {n/dt:.3e} iter/s  acceptance {r['acceptance_rate']:.3f}")


if __name__ == "__main__" and "c1" in sys.argv:
    c1_single_chain()


if __name__ == "__main__" and "gibbs_mid" in sys.argv:  # crossover between the lane kernel (d <= 32) and the wide kernel
    for d in (16, 24, 32, 34, 48, 64, 96):
        gibbs(d, 1 << 18, 40, trace=False)


if __name__ == "__main__" and "block_mid" in sys.argv:  # where spreading the per-iteration sums over four waves starts to pay
    for d in (48, 64, 72, 80, 96):
        rwm_block(d, 4096, 200)
        rwm_block_shape(d, 3072, 600)


LOGIT_SRC = r"""
__device__ double mcmc_user_target_part(const double *x, int d, const double *params, int part, int nparts)
{
    const double tau = params[0];
    const int n_obs = (int)params[1];
    const double *X = params + 2, *y = X + (long)n_obs * d;
    double s = 0.0;
    for (int r = part; r < n_obs; r += nparts) {
        double eta = 0.0;
        for (int k = 0; k < d; ++k) eta += X[(long)r * d + k] * x[k];
        const double m = eta > 0 ? eta : 0.0;
        s += y[r] * eta - (m + log(exp(-m) + exp(eta - m)));
    }
    for (int k = part; k < d; k += nparts) s -= 0.5 * tau * x[k] * x[k];
    return s;
}
"""


def rwm_user_wide(d, n_obs, chains, iters):
    """Bayesian logistic regression (n_obs observations, d coefficients) as a run-time compiled user target in the
    per-workgroup kernel: 2 n_obs d flops + 2 n_obs transcendental pairs per evaluation"""
    rng = np.random.default_rng(0)
    X = rng.standard_normal((n_obs, d)) / np.sqrt(d)
    y = (rng.random(n_obs) < 0.5).astype(np.float64)
    tgt = mcmc_amd.Target.compile(LOGIT_SRC, d, np.concatenate([[0.5, n_obs], X.ravel(), y]))
    f = lambda: mcmc_amd.adapt_rwm_batch(tgt, chains, np.zeros(d), (0.5 / d) * np.eye(d), iters, 50, 0.0, 0, 3 * d, seed=1, thin=10,
                                         final=False)
    ms = timeit(f, reps=2)
    out = f()
    print(f"rwm user target (logistic regression) d={d} n_obs={n_obs} chains={chains} iters={iters}: {ms:.2f} ms  "
          f"{chains*iters/(ms*1e-3):.3e} chain-iter/s  {chains*iters*n_obs/(ms*1e-3):.3e} likelihood terms/s  "
          f"acceptance {float(out['acceptance_rate'].mean()):.2f}  shape fraction {float(out['shape_iterations'].double().mean())/iters:.2f}")


if __name__ == "__main__" and "user_wide" in sys.argv:
    rwm_user_wide(40, 1000, 4096, 500)
    rwm_user_wide(100, 5000, 2048, 200)


if __name__ == "__main__" and "gibbs_big" in sys.argv:  # 128 < d <= 256: register-state kernel with four dot waves per group
    for d in (160, 200, 256):
        gibbs(d, 1 << 16, 10, trace=False)


if __name__ == "__main__" and "gibbs_cross" in sys.argv:  # lane kernel against the register-state kernel around d = 16..32
    for d in (20, 24, 28, 32):
        gibbs(d, 1 << 18, 40, trace=False)


LOGIT_SRC_T = LOGIT_SRC.replace("X[(long)r * d + k]", "X[(long)k * n_obs + r]")  # X stored [d][n_obs]: consecutive threads, consecutive addresses


def rwm_user_wide_t(d, n_obs, chains, iters):
    rng = np.random.default_rng(0)
    X = rng.standard_normal((n_obs, d)) / np.sqrt(d)
    y = (rng.random(n_obs) < 0.5).astype(np.float64)
    tgt = mcmc_amd.Target.compile(LOGIT_SRC_T, d, np.concatenate([[0.5, n_obs], X.T.ravel(), y]))
    f = lambda: mcmc_amd.adapt_rwm_batch(tgt, chains, np.zeros(d), (0.5 / d) * np.eye(d), iters, 50, 0.0, 0, 3 * d, seed=1, thin=10,
                                         final=False)
    ms = timeit(f, reps=2)
    print(f"rwm user target (logistic regression, X column-major) d={d} n_obs={n_obs} chains={chains} iters={iters}: {ms:.2f} ms  "
          f"{chains*iters/(ms*1e-3):.3e} chain-iter/s  {chains*iters*n_obs/(ms*1e-3):.3e} likelihood terms/s")


if __name__ == "__main__" and "user_wide_t" in sys.argv:
    rwm_user_wide_t(40, 1000, 4096, 500)
    rwm_user_wide_t(100, 5000, 2048, 200)


if __name__ == "__main__" and "mid" in sys.argv:  # the 12 < d < 64 range
    for d in (11, 13, 16, 24, 32):
        for kern in ("auto", "block"):
            if d <= 12 and kern == "auto":
                kern = "row"
            rwm_block(d, 65536 if kern != "block" else 16384, 300, kernel=kern)
            rwm_block_shape(d, 65536 if kern != "block" else 8192, 600, kernel=kern)
    for d in (33, 48, 63, 64):  # one chain per wavefront (row kernel, L = 64) against the per-workgroup kernel
        for kern in ("auto", "block"):
            rwm_block(d, 32768 if kern != "block" else 16384, 300, kernel=kern)
            rwm_block_shape(d, 32768 if kern != "block" else 8192, 600, kernel=kern)


def chains_host(d, nc, n):
    """the host-pointer list-of-chains entry (what an R caller reaches): upload, launch, download, untangle -- end to end"""
    idx = np.arange(d)
    H = np.linalg.inv(0.9 ** np.abs(idx[:, None] - idx[None, :])); H = 0.5 * (H + H.T)
    tgt = mcmc_amd.Target.gaussian(np.zeros(d), H)
    rng = np.random.default_rng(1)
    X = [2.0 * rng.standard_normal(d) for _ in range(nc)]
    f = lambda: mcmc_amd.adaptMCMC_chains(tgt, X, np.eye(d), n, 100, 0.0, 0, 500, 100, 7.0)
    f()
    t0 = time.perf_counter(); out = f(); dt = time.perf_counter() - t0
    gb = nc * n * d * 8 / 1e9
    print(f"adaptMCMC_chains d={d} chains={nc} iters={n}: {dt*1e3:.1f} ms end to end  {nc*n/dt:.3e} chain-iter/s  ({gb:.2f} GB of trace to the host, "
          f"{gb/dt:.1f} GB/s)")


if __name__ == "__main__" and "chains" in sys.argv:
    chains_host(10, 4096, 2000)
    chains_host(10, 65536, 1000)
    chains_host(100, 1024, 1000)
