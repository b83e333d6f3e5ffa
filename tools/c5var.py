import sys, os, numpy as np, torch
sys.path.insert(0, "/root/repo")
import mcmc_amd
d=128; idx=np.arange(d); sigma=0.3**np.abs(idx[:,None]-idx[None,:])
_, sd = mcmc_amd.tmvn_precompute(sigma)
def run(n_chains, sweeps, lower, upper, init, tail, moments, label):
    f=lambda: mcmc_amd.tmvn_gibbs_batch(n_chains, sweeps, np.zeros(d), sigma, lower, upper, init, seed=5, tail_safe=tail, trace=False, moments=moments, burn=sweeps//2)
    f(); torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record(); f(); b.record(); torch.cuda.synchronize()
    ms=a.elapsed_time(b)
    groups=n_chains/64; rounds=groups/512
    print(f"{label}: {ms:.1f} ms  {n_chains*sweeps/ms*1e3:.3e} sweeps/s  cycles/step={ms*1e-3*2.4e9/rounds/(sweeps*d):.0f}")
lo4=4*sd; up=np.full(d,np.inf)
n=1<<18
run(n, 40, lo4, up, lo4+0.1, True, True, "tail+moments")
run(n, 40, lo4, up, lo4+0.1, True, False, "tail, no moments")
run(n, 40, np.full(d,-1.0), np.full(d,2.0), np.zeros(d), False, False, "box, plain")
run(n, 40, np.full(d,-1.0), np.full(d,2.0), np.zeros(d), True, True, "box, tail flag+moments")
run(1<<20, 40, lo4, up, lo4+0.1, True, True, "1M chains tail+moments")
