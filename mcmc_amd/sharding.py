"""Chain sharding across GPUs.

Chains are independent, so the only "parallelism strategy" is a contiguous split of the global
chain index range: rank g of G owns [offset, offset + count).  Philox counters are keyed by the
GLOBAL chain id (chain_offset + local index), so results do not depend on G.  This is the
device counterpart of the reference's one-chain-per-R-worker `mclapply_RNG`
(MCMCutilty.R:103-149).
"""
from __future__ import annotations


def shard_chains(total_chains: int, rank: int, world_size: int):
    """Contiguous, balanced split: returns (chain_offset, n_chains) of `rank`."""
    if not (0 <= rank < world_size):
        raise ValueError("rank outside world")
    base, rem = divmod(int(total_chains), int(world_size))
    count = base + (1 if rank < rem else 0)
    offset = rank * base + min(rank, rem)
    return offset, count


def all_reduce_sum(t, group=None):
    """One all-reduce(sum) of a small tensor of partial sums (RCCL on GPUs, gloo on CPU)."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t
