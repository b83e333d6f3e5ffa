"""One-rank RCCL rehearsal of the collectives bench.py and mcmc_amd.diagnostics issue when WORLD_SIZE > 1
(init with device_id, barrier, all-reduce MAX / SUM of float64 on the device).  Run on a GPU box:
    python tools/rccl_check.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dist.barrier()
t = torch.tensor([1.25], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t.item()) == 1.25

import mcmc_amd
from mcmc_amd import diagnostics as diag

g = torch.Generator(device="cuda").manual_seed(3)
mean = torch.randn((4, 256), dtype=torch.float64, device="cuda", generator=g)
var = torch.rand((4, 256), dtype=torch.float64, device="cuda", generator=g) + 0.5
part = diag.rhat_partial(mean, var)
ref = part.clone()
dist.all_reduce(part, op=dist.ReduceOp.SUM)
assert torch.equal(part, ref)
print("rhat", np.round(diag.rhat(mean, var, 100), 4))
dist.barrier()
dist.destroy_process_group()
print("rccl one-rank rehearsal OK")
