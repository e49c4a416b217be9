"""Multi-process (world_size 2, gloo, CPU) test of the batch-sharding path used by bench.py --gpus N: each rank runs
its slice (here through the CPU oracle on the tiny generator), shards are all-gathered, and the result must equal
the single-process full-batch result -- for even and ragged batch sizes."""
import os
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, B, q):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    from ctrlhair_amd import parallel
    from ctrlhair_amd import procedural as P
    from oracle import sean_oracle as O
    torch.set_num_threads(2)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    ngf, S = 16, 64
    sd = O.to_torch(P.sean_state_dict(0, ngf))
    labels = torch.from_numpy(P.blocky_labels(B, S, grid=8))
    codes = torch.from_numpy(P.style_codes(B))
    noise = torch.from_numpy(P.noise_planes(B, S, ngf))
    gen = lambda l, c, n: O.generator_forward(sd, l, c, n, ngf)
    out = parallel.sharded_generate(gen, labels, codes, noise)
    if rank == 0:
        q.put(out.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _worker_pipelined(rank, world, port, steps, q):
    """bench.py's multi-GPU step (parallel.PipelinedGather) with two gloo ranks: per step every rank renders ITS shard
    (global sample index offset rank * n, as bench.py does), the gather of each step must hold both shards in rank order."""
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    from ctrlhair_amd import parallel
    from ctrlhair_amd import procedural as P
    from oracle import sean_oracle as O
    torch.set_num_threads(2)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    ngf, S, n = 16, 64, 2
    sd = O.to_torch(P.sean_state_dict(0, ngf))
    pg = parallel.PipelinedGather((n, 3, S, S), torch.float32, 'cpu')
    ok = True
    for step in range(steps):
        first = rank * n + 100 * step          # different inputs every step: a stale buffer would be noticed
        out = pg.begin()
        out.copy_(O.generator_forward(sd, P.blocky_labels(n, S, grid=8, first=first), P.style_codes(n, first=first),
                                      P.noise_planes(n, S, ngf, first=first), ngf))
        pg.submit()
        ok = ok and pg.check_slot()
    full = pg.finish()
    if rank == 0:
        q.put((ok, full.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_pipelined_gather_two_ranks():
    sys.path.insert(0, ROOT)
    from ctrlhair_amd import procedural as P
    from oracle import sean_oracle as O
    ngf, S, n, steps = 16, 64, 2, 3
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_pipelined, args=(r, 2, 29613, steps, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok, got = q.get(timeout=300)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert ok and got.shape == (2 * n, 3, S, S)
    sd = O.to_torch(P.sean_state_dict(0, ngf))
    for r in range(2):          # the last step's gathered buffer = [rank 0's shard | rank 1's shard]
        first = r * n + 100 * (steps - 1)
        ref = O.generator_forward(sd, P.blocky_labels(n, S, grid=8, first=first), P.style_codes(n, first=first),
                                  P.noise_planes(n, S, ngf, first=first), ngf).numpy()
        assert np.abs(got[r * n:(r + 1) * n] - ref).max() <= 1e-5


def _run(B, port):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, B, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = q.get(timeout=300)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return out


def test_shard_ranges():
    from ctrlhair_amd.parallel import shard_range
    for total in (1, 5, 16, 256):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gather_equals_single_process():
    sys.path.insert(0, ROOT)
    from ctrlhair_amd import procedural as P
    from oracle import sean_oracle as O
    ngf, S = 16, 64
    sd = O.to_torch(P.sean_state_dict(0, ngf))
    for B, port in ((4, 29611), (3, 29612)):       # even and ragged
        full = O.generator_forward(sd, P.blocky_labels(B, S, grid=8), P.style_codes(B), P.noise_planes(B, S, ngf), ngf).numpy()
        got = _run(B, port)
        assert got.shape == full.shape
        assert np.abs(got - full).max() <= 1e-5
