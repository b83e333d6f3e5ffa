"""GPU: the no-allocation entry point mcmc_tmvn_gibbs_batch_noalloc (static device scratch for the shared parameter
block) against mcmc_tmvn_gibbs_batch, through the C ABI: same kernels, so every output must be bit-identical -- on two
calls in a row, the second with a smaller input (the scratch array keeps the first call's contents)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _params(d, rho):
    idx = np.arange(d)
    sigma = rho ** np.abs(idx[:, None] - idx[None, :])
    k = np.arange(d)
    lower = np.where(k % 3 == 0, -np.inf, -1.0)
    upper = np.where(k % 3 == 1, np.inf, 2.0)
    return 0.05 * k, sigma, lower, upper


def _pool_used_high(reset=False):
    """bytes ever handed out by the device's default stream-ordered pool (hipMallocAsync) since the last reset: the
    allocating entry point raises it, the no-allocation one must not (torch's own counters do not see hipMalloc* calls
    made inside libmcmc_amd.so)"""
    import ctypes as C

    hip = C.CDLL("libamdhip64.so")
    pool = C.c_void_p()
    assert hip.hipDeviceGetDefaultMemPool(C.byref(pool), 0) == 0
    val = C.c_uint64(0)
    if reset:
        assert hip.hipMemPoolSetAttribute(pool, 0x8, C.byref(val)) == 0  # hipMemPoolAttrUsedMemHigh := 0
    assert hip.hipMemPoolGetAttribute(pool, 0x8, C.byref(val)) == 0
    return val.value


def _run(cuda, d, n_chains, n_sweeps, seed, no_alloc, **kw):
    import mcmc_amd

    mu, sigma, lower, upper = _params(d, 0.3)
    f64 = cuda.float64
    out = {  # the caller brings every output (the entry point allocates nothing)
        "trace": cuda.full((n_sweeps, d, n_chains), np.nan, dtype=f64, device="cuda"),
        "final_state": cuda.full((d, n_chains), np.nan, dtype=f64, device="cuda"),
        "mean": cuda.full((d, n_chains), np.nan, dtype=f64, device="cuda"),
        "m2": cuda.full((d, n_chains), np.nan, dtype=f64, device="cuda"),
        "status": cuda.full((n_chains,), -1, dtype=cuda.int32, device="cuda"),
    }
    import gc

    gc.collect()  # tensors of earlier tests must not be collected (and returned to torch's allocator) during the call
    cuda.cuda.synchronize()
    before = cuda.cuda.memory_allocated()
    res = mcmc_amd.tmvn_gibbs_batch(n_chains, n_sweeps, mu, sigma, lower, upper, np.full(d, 0.5), seed=seed, moments=True, burn=1,
                                    out=out, no_alloc=no_alloc, **kw)
    assert cuda.cuda.memory_allocated() == before  # nothing came from torch's allocator either
    cuda.cuda.synchronize()
    return {k: v.clone() for k, v in res.items()}


@pytest.mark.parametrize("shapes", [[(128, 300, 6), (48, 70, 9)],      # wide kernel, then a smaller wide block
                                    [(100, 130, 5), (10, 200, 40)],    # wide kernel, then the one-chain-per-lane kernel
                                    [(27, 100, 10), (3, 65, 30)]])     # lane kernel twice
def test_noalloc_entry_equals_the_allocating_one_on_two_calls_in_a_row(cuda, shapes):
    got = [_run(cuda, d, nc, ns, 11 + i, True, tail_safe=(i == 1)) for i, (d, nc, ns) in enumerate(shapes)]
    want = [_run(cuda, d, nc, ns, 11 + i, False, tail_safe=(i == 1)) for i, (d, nc, ns) in enumerate(shapes)]
    for g, w in zip(got, want):
        assert int((g["status"] != 0).sum()) == 0
        assert bool(cuda.isfinite(g["trace"]).all())
        for k in ("trace", "final_state", "mean", "m2", "status"):
            assert cuda.equal(g[k], w[k]), k


def test_noalloc_entry_rejects_what_does_not_fit_the_scratch(cuda):
    import mcmc_amd
    from mcmc_amd._lib import lib

    dmax = lib.mcmc_tmvn_gibbs_noalloc_max_dim()
    assert 128 <= dmax < 256
    d = dmax + 1
    mu, sigma, lower, upper = _params(d, 0.3)
    with pytest.raises(mcmc_amd.McmcError) as e:
        mcmc_amd.tmvn_gibbs_batch(64, 2, mu, sigma, lower, upper, np.full(d, 0.5), no_alloc=True)
    assert e.value.code == -2 and "static scratch" in str(e.value)
    r = _run(cuda, dmax, 64, 2, 3, True)  # the largest d that fits
    w = _run(cuda, dmax, 64, 2, 3, False)
    assert cuda.equal(r["trace"], w["trace"])


def test_noalloc_entry_allocates_nothing_on_the_device(cuda):
    """torch.cuda.memory_allocated() only sees torch's caching allocator.  Here: the free device memory
    (hipMemGetInfo) and the high-water mark of the stream-ordered pool are unchanged across a no-allocation call,
    while the allocating entry point does move the pool's mark (so the probe can tell the two apart)."""
    _run(cuda, 100, 130, 5, 1, True)   # warm-up: code objects loaded, the scratch address looked up
    _run(cuda, 100, 130, 5, 1, False)
    cuda.cuda.synchronize()
    _pool_used_high(reset=True)
    free0 = cuda.cuda.mem_get_info()[0]
    _run(cuda, 100, 130, 5, 2, True)
    cuda.cuda.synchronize()
    assert _pool_used_high() == 0
    assert cuda.cuda.mem_get_info()[0] == free0
    _run(cuda, 100, 130, 5, 2, False)
    cuda.cuda.synchronize()
    assert _pool_used_high() > 0       # the allocating variant's hipMallocAsync is what the probe sees
