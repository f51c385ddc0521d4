"""Data-parallel plumbing of the CycleGAN step (the MirroredStrategy the reference lists as TODO,
cgan.py:8-11): one process per GPU, torch.distributed over RCCL (backend "nccl") on xGMI.

Every replica computes local-batch-mean losses and gradients; ONE all-reduce of the flat gradient
vector of all four networks (0.62 M floats = 2.5 MB in 3-D) sums them and the Adam kernel divides
by the world size, which equals the global-batch mean the reference's TODO asks for.  The
payload is latency-bound on xGMI (a few tens of microseconds), so a single un-bucketed collective
is the right shape; there is no other exchange step on the hot path.
"""
import torch
import torch.distributed as dist


def is_distributed():
    return dist.is_available() and dist.is_initialized()


def allreduce_sum_(flat, group=None, force=False):
    """In-place sum of `flat` over the process group (no-op without one, or with one rank unless `force`:
    the one-GPU rehearsal of the collective, EM2EM.exchange)."""
    if is_distributed() and (force or dist.get_world_size(group) > 1):
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


def shard(items, rank, world_size):
    """Round-robin shard of independent work items (tiles, volumes) -- no collective needed."""
    return list(items)[rank::world_size]


def replica_seed(seed, rank):
    """Independent dropout stream per replica."""
    return int(seed) + int(rank)
