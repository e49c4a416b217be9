#!/usr/bin/env python3
"""bench.py -- headline benchmark of the CtrlHair hot path on MI355X.

Workload (BASELINE.json configs[1], SURVEY.md 8d Config 2): SEAN generator forward only, batch 16 per GPU,
synthetic 512x512 blocky label maps + tanh(N(0,1)) style codes + explicit N(0,1) noise planes, procedural
(random, calibrated) weights of the real ngf=64 architecture, fp32 end to end.  One "step" = one generator
pass over one batch already resident in HBM.  With N>1 (one process per GPU, torch.distributed/RCCL) every
rank runs its own batch (weak scaling) and the per-rank output shards are all-gathered over xGMI inside the
timed region, as north_star asks (on a side stream, under the next step's generator pass; --sync-gather serialises it).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 16] [--size 512] [--no-cpu-baseline]

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, exact f32
PEAK_F16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: f16/bf16 MFMA dense peak (spec; the 2:1-sparse figure is not used)
PEAK_HBM_GBS = 8000.0


def cpu_baseline(ngf, S, sd_np, budget_s=30.0):
    """The oracle (torch fp32 CPU restatement of the reference path) timed on this host's cores, on a bounded
    sample of the same workload: 1 warm-up + up to 2 timed single-image forwards at the benchmark size."""
    import torch
    from ctrlhair_amd import procedural as P
    from oracle import sean_oracle as O          # baseline leg only -- never on the measured path
    sd = O.to_torch(sd_np)
    wc = {}
    lab, cd = P.blocky_labels(1, S), P.style_codes(1)
    nz = P.noise_planes(1, S, ngf)
    t0 = time.time()
    O.generator_forward(sd, P.blocky_labels(1, 128, grid=8), cd, P.noise_planes(1, 128, ngf), ngf, weights_cache=wc)
    times = []
    for _ in range(2):
        t = time.time()
        O.generator_forward(sd, lab, cd, nz, ngf, weights_cache=wc)
        times.append(time.time() - t)
        if time.time() - t0 > budget_s:
            break
    best = min(times)
    return {'value': round(1.0 / best, 4), 'unit': 'images/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': f'{len(times)} single-image {S}x{S} generator forwards (min), torch {torch.__version__} CPU, '
                      f'after a 128x128 warm-up'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--size', type=int, default=512)
    ap.add_argument('--ngf', type=int, default=64)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--path', choices=('f32', 'f16x3', 'f16'), default='f16x3',
                    help='conv arithmetic: f16x3 = 3-term split-operand f16 MFMA, f32 accumulate, f32-class accuracy '
                         '(default; max |delta| vs the exact path 1.5e-5); f32 = exact-f32 MFMA (v_mfma_f32_32x32x2_f32); '
                         'f16 = single-term f16 operands (reduced precision, BASELINE configs[4]; informational)')
    ap.add_argument('--dbg', type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument('--sync-gather', action='store_true',
                    help='N > 1: all-gather each step on the compute stream instead of overlapping it with the next step')
    ap.add_argument('--force-dist', action='store_true', help=argparse.SUPPRESS)   # 1-rank process group: exercises the N > 1 code
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.gpus != world and world > 1:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the HIP path has no CPU fallback)')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    from ctrlhair_amd import procedural as P
    from ctrlhair_amd.sean.generator import SeanGenerator
    B, S, ngf = args.batch, args.size, args.ngf
    sd = P.sean_state_dict(0, ngf)
    gen = SeanGenerator(local_rank, f16x3={'f32': 0, 'f16x3': 1, 'f16': 2}[args.path]).load_state_dict(sd, max_batch=B, max_size=S)
    if args.dbg:
        gen.handle.set_option('sean.dbg', args.dbg)
    first = rank * B     # global sample index offset (SURVEY.md 8d Config 4)
    labels = torch.from_numpy(P.blocky_labels(B, S, first=first)).to(dev)
    codes = torch.from_numpy(P.style_codes(B, first=first)).to(dev)
    noise = torch.from_numpy(P.noise_planes(B, S, ngf, first=first)).to(dev)
    # Two output / gather buffers: the RCCL all-gather of step i (side stream, over xGMI) runs under the generator pass of
    # step i+1; the timed region ends after the last gather has completed (device-wide synchronize).
    outs = [torch.empty(B, 3, S, S, dtype=torch.float32, device=dev) for _ in range(2)]
    gathered = [torch.empty(world * B, 3, S, S, dtype=torch.float32, device=dev) for _ in range(2)] if dist else None
    comm = torch.cuda.Stream(dev) if dist else None
    gather_done = [None, None]
    counter = [0]

    def step():
        j = counter[0] & 1
        counter[0] += 1
        main = torch.cuda.current_stream(dev)
        if gather_done[j] is not None:
            main.wait_event(gather_done[j])          # the gather that read outs[j] two steps ago
        gen.generate(labels, codes, noise, out=outs[j])
        if dist is None:
            return
        if args.sync_gather:
            dist.all_gather_into_tensor(gathered[j], outs[j])
            return
        ready = torch.cuda.Event()
        ready.record(main)
        with torch.cuda.stream(comm):
            comm.wait_event(ready)
            dist.all_gather_into_tensor(gathered[j], outs[j])
            done = torch.cuda.Event()
            done.record(comm)
        gather_done[j] = done

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    gen.handle.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    gen.handle.profile_enable(False)
    out = outs[(counter[0] - 1) & 1]
    if dist is not None and args.steps > 0:        # the gathered buffer holds every rank's shard, this rank's at its slot
        g = gathered[(counter[0] - 1) & 1]
        assert torch.equal(g[rank * B:(rank + 1) * B], out), 'all-gather result does not contain this rank\'s shard'
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    prof_ace = gen.handle.profile_read(1)
    prof_plain = gen.handle.profile_read(0)
    prof_all = gen.handle.profile_read(-1)
    assert args.dbg or torch.isfinite(out).all()

    res = None
    if rank == 0:
        ms_step = dt / args.steps * 1e3
        value = world * B * args.steps / dt
        # algorithmic (f32-equivalent) conv FLOP/s of the dominant kernel, hipEvent-timed per launch inside the library
        alg = prof_ace['flops'] / (prof_ace['ms'] * 1e-3) / 1e12 if prof_ace['ms'] > 0 else 0.0
        if args.path == 'f16':
            # reduced-precision configuration (BASELINE.json configs[4] class; NOT the headline): one f16 MFMA product per term
            executed, peak = alg, PEAK_F16_MFMA_TFLOPS
            kname = 'conv_sh16_ws_kernel / conv_sh16_kernel <KS=3,...,EPI_ACE,TERMS=1> (SPADE gamma/beta conv, f16 operands, fused ACE epilogue)'
            dtype = 'f16 operands on MFMA, f32 accumulate, f32 normalisation/modulation (reduced precision: tolerance 5e-2)'
        elif args.path.startswith('f16x3'):
            # every f32 product is executed as 3 f16 MFMA products: utilisation is priced on executed MFMA FLOPs
            executed, peak = 3.0 * alg, PEAK_F16_MFMA_TFLOPS
            kname = 'conv_sh16_ws_kernel / conv_sh16_kernel <KS=3,...,EPI_ACE> (SPADE gamma/beta conv, f16x3 split operands, fused ACE epilogue)'
            dtype = 'f32 storage + f32 accumulate; conv products as 3-term f16 split on MFMA (f32-class: |delta| <= 1.5e-5 vs exact f32)'
        else:
            executed, peak = alg, PEAK_F32_MFMA_TFLOPS
            kname = 'conv_mfma_kernel<KS=3,...,EPI_ACE> (SPADE gamma/beta conv, exact-f32 MFMA, fused ACE epilogue)'
            dtype = 'f32'
        traffic = traffic_detail = None
        tpath = os.path.join(ROOT, 'profiles', 'latest_traffic.json')
        if os.path.exists(tpath):      # HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
            try:
                traffic_detail = json.load(open(tpath)).get(args.path)
                traffic = traffic_detail and round(float(traffic_detail['hbm_bytes']))
            except Exception:
                traffic = traffic_detail = None
        res = {
            'metric': '512x512 edited images/sec (SEAN generator forward), whole job',
            'value': round(value, 3), 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms_step, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': dtype, 'data': 'synthetic (blocky labels, tanh-normal codes, explicit noise planes; '
                                    'procedural calibrated weights, no checkpoint ships with the reference)',
            'config': {'workload': f'SEAN generator forward only, batch {B}/GPU, {S}x{S}, ngf={ngf}, fp32 '
                                   f'(BASELINE.json configs[1])', 'global_batch': world * B, 'conv_path': args.path,
                       'parallelism': f'batch-sharded x{world}' + ((' + RCCL all-gather of outputs' + ('' if args.sync_gather else ' overlapped with the next step')) if dist is not None else '')},
            'roofline': {
                'bound': 'mfma', 'kernel': kname,
                'achieved': round(executed, 2), 'peak': peak, 'unit': 'TFLOP/s',
                'frac': round(executed / peak, 4), 'traffic': traffic, 'traffic_unit': 'HBM bytes per launch (FETCH_SIZE x2 + WRITE_SIZE)',
                'traffic_detail': traffic_detail,
                'algorithmic_f32_tflops': round(alg, 2),
                'launches': prof_ace['launches'], 'avg_launch_ms': round(prof_ace['ms'] / max(prof_ace['launches'], 1), 4),
                'flops_per_launch_avg': prof_ace['flops'] / max(prof_ace['launches'], 1),
                'all_mfma_convs': {'algorithmic_tflops': round(prof_all['flops'] / max(prof_all['ms'], 1e-9) / 1e9, 2),
                                   'ms_per_step': round(prof_all['ms'] / args.steps, 3),
                                   'plain_algorithmic_tflops': round(prof_plain['flops'] / max(prof_plain['ms'], 1e-9) / 1e9, 2)},
                'hbm_algorithmic_gbs': round(prof_all['bytes'] / max(prof_all['ms'], 1e-9) / 1e6, 1),
            },
        }
        if not args.no_cpu_baseline:
            res['cpu_baseline'] = cpu_baseline(ngf, S, sd)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:                      # the ONE JSON line, after every library banner (RCCL prints its version through C stdio)
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(res), flush=True)


if __name__ == '__main__':
    main()
