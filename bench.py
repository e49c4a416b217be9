#!/usr/bin/env python3
"""bench.py -- headline benchmark of the CtrlHair hot path on MI355X.

Workloads
  generator (default; BASELINE.json configs[1], SURVEY.md 8d Config 2): SEAN generator forward only, batch 16 per GPU
      (32 per GPU with --gpus 8 = configs[3], B=256), synthetic 512x512 blocky label maps + tanh(N(0,1)) style codes +
      explicit N(0,1) noise planes, procedural (random, calibrated) weights of the real ngf=64 architecture.
  pipeline (BASELINE.json configs[2], SURVEY.md 8d Config 3): the whole edit at batch 8 -- BiSeNet parse @512 -> label remap
      -> nearest 256 -> shape encoders -> Zencoder @512 -> colour encoder / predictor / generator with slider deltas ->
      shape decoder -> nearest x2 -> SEAN generator @512.
One "step" = one pass over one batch already resident in HBM.  With N>1 (one process per GPU, torch.distributed/RCCL)
every rank runs its own batch (weak scaling) and the output shards are all-gathered over xGMI inside the timed region
(ctrlhair_amd.parallel.PipelinedGather: side stream, under the next step's pass; --sync-gather serialises it).

Two arithmetic legs are timed with the same steps / warm-up, each in its own pass WITHOUT per-launch instrumentation:
  * the headline leg (--path, default f16x3): fp32 storage and accumulation, conv products as a 3-term f16 split on the
    matrix cores with power-of-two operand scaling (ctrlhair_amd/csrc/sh16.h) -- f32-class by construction;
  * "strict_fp32": the same job on the exact-f32 matrix-core path (v_mfma_f32_32x32x2_f32), the reference's arithmetic.
A third, separate pass per leg with hipEvent brackets around every MFMA conv launch gives the roofline figures.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--size 512] [--workload generator|pipeline]
                    [--path f16x3|f32|f16|bf16] [--no-strict-fp32] [--no-cpu-baseline]

Prints ONE JSON line (rank 0).
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, exact f32
PEAK_F16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: f16/bf16 MFMA dense peak (spec; the 2:1-sparse figure is not used)
PEAK_HBM_GBS = 8000.0
CONTRACT_BYTES_PER_IMAGE = 5.805e9        # conv-layer accounting, activations in + out, S=512, fp32
CONTRACT_WEIGHT_BYTES = 1.061e9           # ... + weights once per batch
PATH_OPTION = {'f32': 0, 'f16x3': 1, 'f16': 2, 'bf16': 3}


def csrc_sha():
    """Hash of the sources of the dominant kernels (the MFMA conv headers): profiles/latest_traffic.json records the one its
    PMC passes were taken at; a mismatch means the committed traffic figure is stale and is not reported."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, 'ctrlhair_amd', 'csrc')
    for f in ('conv_mfma.h', 'conv_sh16.h', 'conv_sh16_ws2.h', 'sh16.h'):
        h.update(f.encode())
        h.update(open(os.path.join(d, f), 'rb').read())
    return h.hexdigest()[:16]


def cpu_baseline(ngf, S, sd_np, budget_s=40.0):
    """The oracle (torch fp32 CPU restatement of the reference path) timed on this host's cores on a bounded sample of
    the same workload: one 128x128 warm-up, then up to 3 single-image forwards at the benchmark size (median)."""
    import numpy as np
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
    for _ in range(3):
        t = time.time()
        O.generator_forward(sd, lab, cd, nz, ngf, weights_cache=wc)
        times.append(time.time() - t)
        if time.time() - t0 > budget_s:
            break
    med = float(np.median(times))
    return {'value': round(1.0 / med, 4), 'unit': 'images/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': f'{len(times)} single-image {S}x{S} generator forwards (median; batch 16 would take minutes), torch '
                      f'{torch.__version__} CPU, after a 128x128 warm-up'}


class GeneratorJob:
    """configs[1]: one SEAN generator pass over this rank's batch."""

    def __init__(self, args, path, dev, rank, sd):
        import torch
        from ctrlhair_amd import procedural as P
        from ctrlhair_amd.sean.generator import SeanGenerator
        B, S, ngf = args.batch, args.size, args.ngf
        opts = {'sean.ahead': args.ahead} if args.ahead >= 0 else None
        self.gen = SeanGenerator(dev.index, f16x3=PATH_OPTION[path], options=opts).load_state_dict(sd, max_batch=B, max_size=S)
        if args.dbg:
            self.gen.handle.set_option('sean.dbg', args.dbg)
        first = rank * B     # global sample index offset (SURVEY.md 8d Config 4)
        self.labels = torch.from_numpy(P.blocky_labels(B, S, first=first)).to(dev)
        self.codes = torch.from_numpy(P.style_codes(B, first=first)).to(dev)
        self.noise = torch.from_numpy(P.noise_planes(B, S, ngf, first=first)).to(dev)
        self.handle = self.gen.handle
        self.images = B
        self.out_shape = (B, 3, S, S)

    def step(self, out):
        self.gen.generate(self.labels, self.codes, self.noise, out=out)

    def close(self):
        self.handle.close()


class PipelineJob:
    """configs[2]: the whole edit (ctrlhair_amd.pipeline.EditPipeline) over this rank's batch of portraits."""

    def __init__(self, args, path, dev, rank, weights):
        import torch
        from ctrlhair_amd import procedural as P
        from ctrlhair_amd.pipeline import EditPipeline
        B, S = args.batch, args.size
        self.pipe = EditPipeline(weights, device=dev.index, img_size=S, max_batch=B, f16x3=PATH_OPTION[path])
        self.img = torch.from_numpy(P.synthetic_images(B, S, seed=11 + rank * B)).to(dev)
        self.handle = self.pipe.models.generator.handle
        self.images = B
        self.out_shape = (B, 3, S, S)

    def step(self, out):
        self.pipe.edit(self.img, out=out)

    # algorithmic GFLOP per image of each stage (generator: FLOPs executed after the exact reformulations)
    STAGE_GFLOP = {'parse': 27.5, 'shape_encode': 4.9, 'zencoder': 169.6, 'shape_decode': 32.2, 'generator': 1083.8}

    def stages(self, path):
        """Per-stage time and roofline of one edit (separate instrumented runs, torch events)."""
        ms = self.pipe.stage_times(self.img)
        out = {}
        for k, t in ms.items():
            row = {'ms': round(t, 3)}
            if k in self.STAGE_GFLOP:
                tf = self.STAGE_GFLOP[k] * self.images / t            # GFLOP per ms = TFLOP/s
                # matrix-core path of the stage's dominant convs: exact-f32 MFMA, or 3 executed f16 products per f32 product
                f16 = path != 'f32'          # every stage's convs run on the f16 matrix cores unless the strict-f32 path is on
                terms = 3.0 if (f16 and (path == 'f16x3' or k != 'generator')) else 1.0     # (aux networks: always the 3-term split)
                peak = PEAK_F16_MFMA_TFLOPS if f16 else PEAK_F32_MFMA_TFLOPS
                row.update({'algorithmic_tflops': round(tf, 1), 'bound': 'mfma', 'peak_tflops': peak,
                            'frac': round(terms * tf / peak, 4)})
            else:
                row['bound'] = 'launch latency (three small MLPs + slider arithmetic)'
            out[k] = row
        return out

    def close(self):
        self.pipe.close()


def run_leg(job, args, dist, dev, world):
    """warm-up, then the timed region (barrier + synchronize on both sides, MAX over ranks), then a separate
    instrumented pass for the per-kernel figures."""
    import torch
    from ctrlhair_amd.parallel import PipelinedGather
    pg = PipelinedGather(job.out_shape, torch.float32, dev, overlap=not args.sync_gather)

    def step():
        out = pg.begin()
        job.step(out)
        pg.submit()

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    if args.steps > 0:
        assert pg.check_slot(), 'all-gather result does not contain this rank\'s shard at its slot'
    out = pg.last_local()
    assert args.dbg or args.steps == 0 or bool(torch.isfinite(out).all())
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # instrumented pass (not timed as a whole): hipEvent brackets around every MFMA conv launch, on the launch stream
    nprof = max(1, min(args.steps, 5))
    job.handle.profile_enable(True)
    scratch = pg.outs[0]
    for _ in range(nprof):
        job.step(scratch)
    torch.cuda.synchronize()
    job.handle.profile_enable(False)
    prof = {'ace': job.handle.profile_read(1), 'plain': job.handle.profile_read(0)}
    prof['all'] = job.handle.profile_read(-1)
    prof['steps'] = nprof
    return dt, prof


def roofline_block(path, prof, value, B):
    ace, plain, allk = prof['ace'], prof['plain'], prof['all']
    alg = ace['flops'] / (ace['ms'] * 1e-3) / 1e12 if ace['ms'] > 0 else 0.0     # algorithmic (f32-equivalent) TFLOP/s
    if path in ('f16', 'bf16'):
        executed, peak = alg, PEAK_F16_MFMA_TFLOPS
        kname = f'conv_sh16_ws_kernel / conv_sh16_kernel <KS=3,...,EPI_ACE,TERMS=1> (SPADE gamma/beta conv, {path} operands, fused ACE epilogue)'
    elif path == 'f16x3':
        executed, peak = 3.0 * alg, PEAK_F16_MFMA_TFLOPS      # every f32 product is executed as 3 f16 MFMA products
        kname = 'conv_sh16_ws_kernel / conv_sh16_kernel <KS=3,...,EPI_ACE> (SPADE gamma/beta conv, f16x3 split operands, fused ACE epilogue)'
    else:
        executed, peak = alg, PEAK_F32_MFMA_TFLOPS
        kname = 'conv_mfma_kernel<KS=3,...,EPI_ACE> (SPADE gamma/beta conv, exact-f32 MFMA, fused ACE epilogue)'
    traffic = detail = note = None
    tpath = os.path.join(ROOT, 'profiles', 'latest_traffic.json')
    if os.path.exists(tpath):      # HBM bytes per launch of the dominant kernel from committed rocprofv3 PMC passes
        try:
            detail = json.load(open(tpath)).get(path)
            if detail and detail.get('csrc_sha') == csrc_sha():
                traffic = round(float(detail['hbm_bytes']))
            elif detail:
                note = 'PMC passes on file were taken at other kernel sources (csrc_sha mismatch): not reported'
                detail = None
        except Exception:
            traffic = detail = None
    contract = CONTRACT_BYTES_PER_IMAGE + CONTRACT_WEIGHT_BYTES / B
    return {
        'bound': 'mfma', 'kernel': kname, 'achieved': round(executed, 2), 'peak': peak, 'unit': 'TFLOP/s',
        'frac': round(executed / peak, 4), 'traffic': traffic,
        'traffic_unit': 'HBM bytes per launch (FETCH_SIZE x2 + WRITE_SIZE)', 'traffic_detail': detail, 'traffic_note': note,
        'algorithmic_f32_tflops': round(alg, 2), 'launches': ace['launches'],
        'avg_launch_ms': round(ace['ms'] / max(ace['launches'], 1), 4),
        'flops_per_launch_avg': ace['flops'] / max(ace['launches'], 1),
        'all_mfma_convs': {'algorithmic_tflops': round(allk['flops'] / max(allk['ms'], 1e-9) / 1e9, 2),
                           'ms_per_step': round(allk['ms'] / prof['steps'], 3),
                           'plain_algorithmic_tflops': round(plain['flops'] / max(plain['ms'], 1e-9) / 1e9, 2)},
        'timing': 'hipEvents around each launch in a separate instrumented pass (not the timed region)',
        # north_star's other yardstick: conv-layer contract bytes (each conv reads its input and writes its output once,
        # fp32) against HBM peak.  The path is matrix-core bound (AI ~ 440 FLOP/B): this fraction cannot reach 55 %.
        'hbm_contract': {'bytes_per_image': contract, 'achieved_gbs': round(value * contract / 1e9, 1),
                         'peak_gbs': PEAK_HBM_GBS, 'frac': round(value * contract / 1e9 / PEAK_HBM_GBS, 4)},
    }


DTYPE = {
    'f16x3': 'f32 storage + f32 accumulate; conv products as 3-term f16 split on MFMA with power-of-two operand scaling '
             '(f32-class by construction: csrc/sh16.h)',
    'f32': 'f32 (exact-f32 MFMA)',
    'f16': 'f16 operands on MFMA, f32 accumulate, f32 normalisation/modulation (reduced precision: tolerance 5e-2)',
    'bf16': 'bf16 operands on MFMA, f32 accumulate, f32 normalisation/modulation (reduced precision: tolerance 5e-2)',
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=0, help='per GPU; default 16 (generator; 32 with --gpus 8 = configs[3]) / 8 (pipeline)')
    ap.add_argument('--size', type=int, default=512)
    ap.add_argument('--ngf', type=int, default=64)
    ap.add_argument('--workload', choices=('generator', 'pipeline'), default='generator')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-strict-fp32', action='store_true', help='skip the exact-f32 leg')
    ap.add_argument('--path', choices=tuple(PATH_OPTION), default='f16x3',
                    help='arithmetic of the headline leg: f16x3 = 3-term split-operand f16 MFMA, f32 accumulate, f32-class '
                         '(default); f32 = exact-f32 MFMA; f16 / bf16 = single-term reduced-precision operands (configs[4])')
    ap.add_argument('--dbg', type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument('--ahead', type=int, default=-1, help=argparse.SUPPRESS)       # option sean.ahead (experiments)
    ap.add_argument('--sync-gather', action='store_true',
                    help='N > 1: all-gather each step on the compute stream instead of overlapping it with the next step')
    ap.add_argument('--force-dist', action='store_true', help=argparse.SUPPRESS)   # 1-rank process group: exercises the N > 1 code
    args = ap.parse_args()
    if args.batch <= 0:
        args.batch = 8 if args.workload == 'pipeline' else (32 if args.gpus == 8 else 16)

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

    B, S, ngf = args.batch, args.size, args.ngf
    if args.workload == 'pipeline':
        from ctrlhair_amd.hair_editor import procedural_weights
        weights = procedural_weights(0, ngf)
        sd = weights['sean']
        make = lambda path: PipelineJob(args, path, dev, rank, weights)
        wl = (f'full CtrlHair edit (BiSeNet parse -> shape + colour/texture branches -> Zencoder -> SEAN generator), '
              f'batch {B}/GPU, {S}x{S}, ngf={ngf} (BASELINE.json configs[2])')
        metric = '512x512 edited images/sec (full pipeline), whole job'
    else:
        from ctrlhair_amd import procedural as P
        sd = P.sean_state_dict(0, ngf)
        make = lambda path: GeneratorJob(args, path, dev, rank, sd)
        cfgn = 'configs[3]: B=256 on 8 GPUs' if (world == 8 and B == 32) else 'configs[1]'
        wl = f'SEAN generator forward only, batch {B}/GPU, {S}x{S}, ngf={ngf} (BASELINE.json {cfgn})'
        metric = '512x512 edited images/sec (SEAN generator forward), whole job'

    legs = [args.path] + ([] if (args.no_strict_fp32 or args.path == 'f32') else ['f32'])
    results = {}
    for path in legs:
        job = make(path)
        dt, prof = run_leg(job, args, dist, dev, world)
        value = world * job.images * args.steps / dt if args.steps else 0.0
        results[path] = {'value': round(value, 3), 'ms_per_step': round(dt / max(args.steps, 1) * 1e3, 3),
                         'roofline': roofline_block(path, prof, value / world, B)}
        if hasattr(job, 'stages') and rank == 0:
            results[path]['stages'] = job.stages(path)
        job.close()
        del job
        torch.cuda.empty_cache()

    res = None
    if rank == 0:
        head = results[args.path]
        par = f'batch-sharded x{world}'
        if dist is not None:
            par += ' + RCCL all-gather of outputs' + ('' if args.sync_gather else ' overlapped with the next step')
        res = {
            'metric': metric, 'value': head['value'], 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': head['ms_per_step'], 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': DTYPE[args.path],
            'data': 'synthetic (blocky labels, tanh-normal codes, explicit noise planes; procedural calibrated weights, '
                    'no checkpoint ships with the reference)',
            'config': {'workload': wl, 'global_batch': world * B, 'conv_path': args.path, 'parallelism': par},
            'roofline': head['roofline'],
        }
        if 'stages' in head:
            res['stages'] = head['stages']
        if 'f32' in results and args.path != 'f32':
            s = results['f32']
            res['strict_fp32'] = {'value': s['value'], 'unit': 'images/s', 'ms_per_step': s['ms_per_step'],
                                  'dtype': DTYPE['f32'], 'steps': args.steps, 'warmup': args.warmup,
                                  'note': 'same job, same timed protocol, exact-f32 matrix-core path (the reference\'s arithmetic)',
                                  'roofline': s['roofline']}
            if 'stages' in s:
                res['strict_fp32']['stages'] = s['stages']
        if not args.no_cpu_baseline and world == 1:
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
