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


class PipelinedGather:
    """The multi-GPU step of bench.py / a batch renderer: every rank renders its own shard into one of two output buffers
    and the all-gather of step i (a side stream on GPUs: RCCL over xGMI; 50 MB per rank per 16-image step) runs under the
    generator pass of step i+1.  Usage per step::

        out = pg.begin()            # buffer to render into; waits for the gather that read it two steps ago
        render(..., out=out)
        pg.submit()                 # enqueue the all-gather of `out`
    ...
        full = pg.finish()          # all gathers done; [world * n, ...] of the last step, rank r's shard at slot r

    Works on CPU tensors (gloo; gathers are then synchronous), which is how tests/test_parallel_gloo.py drives exactly
    this code with two ranks.  `overlap=False` gathers on the compute stream (bench.py --sync-gather)."""

    def __init__(self, shard_shape, dtype=torch.float32, device='cpu', group=None, overlap=True):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.device = torch.device(device)
        self.cuda = self.device.type == 'cuda'
        self.overlap = overlap and self.cuda and self.world > 1
        self.outs = [torch.empty(tuple(shard_shape), dtype=dtype, device=self.device) for _ in range(2)]
        full = (self.world * shard_shape[0],) + tuple(shard_shape[1:])
        self.gathered = [torch.empty(full, dtype=dtype, device=self.device) for _ in range(2)] if self.world > 1 else None
        self.comm = torch.cuda.Stream(self.device) if self.overlap else None
        self.done = [None, None]
        self.count = 0

    def begin(self) -> torch.Tensor:
        j = self.count & 1
        if self.done[j] is not None:           # the gather that read outs[j] two steps ago
            torch.cuda.current_stream(self.device).wait_event(self.done[j])
            self.done[j] = None
        return self.outs[j]

    def submit(self):
        j = self.count & 1
        self.count += 1
        if self.world == 1:
            return
        if not self.overlap:
            dist.all_gather_into_tensor(self.gathered[j], self.outs[j], group=self.group)
            return
        main = torch.cuda.current_stream(self.device)
        ready = torch.cuda.Event()
        ready.record(main)
        with torch.cuda.stream(self.comm):
            self.comm.wait_event(ready)
            dist.all_gather_into_tensor(self.gathered[j], self.outs[j], group=self.group)
            done = torch.cuda.Event()
            done.record(self.comm)
        self.done[j] = done

    def finish(self) -> torch.Tensor:
        """Waits for every enqueued gather; returns the gathered tensor of the last submitted step (the local buffer when
        world == 1)."""
        if self.cuda:
            torch.cuda.synchronize(self.device)
        j = (self.count - 1) & 1
        return self.outs[j] if self.world == 1 else self.gathered[j]

    def last_local(self) -> torch.Tensor:
        return self.outs[(self.count - 1) & 1]

    def check_slot(self) -> bool:
        """This rank's shard of the last step sits at its slot of the gathered buffer."""
        if self.world == 1:
            return True
        g, n = self.finish(), self.outs[0].shape[0]
        return bool(torch.equal(g[self.rank * n:(self.rank + 1) * n], self.last_local()))
