"""Batch sharding of the generator path across the GPUs of one node.

Every op on the path is per-sample (eval-BN uses running statistics, IN/LN/region pooling are per-sample), so the
batch axis shards with no data-path collective; weights are replicated.  The only exchange is the all-gather of the
output shards (RCCL over xGMI when the backend is 'nccl'; 'gloo' in the CPU tests).  One process per GPU.
"""
from typing import Callable, Tuple

import torch
import torch.distributed as dist


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced slice [lo, hi) of `total` samples for `rank`: the first (total % world) ranks get one more."""
    q, r = divmod(total, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def gather_shards(local: torch.Tensor, total: int, group=None) -> torch.Tensor:
    """All-gather per-rank output shards [n_r, ...] (n_r from shard_range) into [total, ...] on every rank."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local
    rank = dist.get_rank(group)
    sizes = [shard_range(total, r, world)[1] - shard_range(total, r, world)[0] for r in range(world)]
    assert local.shape[0] == sizes[rank]
    if len(set(sizes)) == 1:
        out = local.new_empty((total,) + tuple(local.shape[1:]))
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    mx = max(sizes)                      # ragged tail: pad to the largest shard, gather, drop the padding
    pad = local.new_zeros((mx,) + tuple(local.shape[1:]))
    pad[:local.shape[0]] = local
    buf = local.new_empty((world * mx,) + tuple(local.shape[1:]))
    dist.all_gather_into_tensor(buf, pad, group=group)
    return torch.cat([buf[r * mx:r * mx + sizes[r]] for r in range(world)], dim=0)


def sharded_generate(generate: Callable, labels: torch.Tensor, codes: torch.Tensor, noise: torch.Tensor, group=None):
    """Run `generate(labels, codes, noise)` on this rank's slice of a global batch and all-gather the images.
    Inputs are the *global* tensors (identical on every rank); returns the global output [B,3,S,S] on every rank."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    B = labels.shape[0]
    lo, hi = shard_range(B, rank, world)
    local = generate(labels[lo:hi], codes[lo:hi], None if noise is None else noise[lo:hi])
    return gather_shards(local, B, group)
