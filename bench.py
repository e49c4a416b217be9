#!/usr/bin/env python3
"""bench.py -- headline benchmark of the CtrlHair hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one pass of the hot path over one batch already resident in HBM.  The default run (N = 1) answers, in ONE JSON
line, every configuration of BASELINE.json that fits one GPU:

  top level   configs[1]: SEAN generator forward only, batch 16, synthetic 512x512 blocky label maps + tanh(N(0,1)) style
              codes + explicit N(0,1) noise planes, procedural (random, calibrated) weights of the real ngf=64 architecture,
              on the EXACT-f32 matrix-core path (v_mfma_f32_32x32x2_f32) -- the reference's arithmetic.  `value`, `dtype`,
              `roofline`, the per-step percentiles and `cpu_baseline` all belong to this leg.
  f32_class_f16x3        the same job with the conv products evaluated as a 3-term f16 split on the f16 matrix cores (f32
                         storage and accumulation, power-of-two operand scaling: ctrlhair_amd/csrc/sh16.h).  Narrower than
                         fp32 per product (2^-22), so it is NOT the fp32 number of record; tested to 1e-3 like the exact path.
  face_like_labels       both legs again on face-like label maps (large regions, curved boundaries) instead of the blocky
                         maps of SURVEY.md 8(d) Config 2.
  pipeline               configs[2]: the whole edit at batch 8 (BiSeNet parse @512 -> label remap -> nearest 256 -> shape
                         encoders -> Zencoder @512 -> colour encoder / predictor / generator with slider deltas -> shape
                         decoder -> nearest x2 -> SEAN generator @512), f16x3 and exact-f32 legs, with per-stage times.

With --gpus N > 1 the script starts N ranks itself when it was not launched by torch.distributed.run (it re-executes under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`), one process per GPU over RCCL:
every rank runs its own batch (weak scaling; 32 per GPU with --gpus 8 = configs[3], B=256) and the output shards are
all-gathered over xGMI inside the timed region (ctrlhair_amd.parallel.PipelinedGather: side stream, under the next step's
pass; --sync-gather serialises it).  `n_gpus` is the world size RCCL reports.

Timing: W untimed warm-up steps, then K steps bracketed by barrier + torch.cuda.synchronize() on both sides (max over ranks)
-> `value` / `ms_per_step`; an event is recorded on the launch stream at every step boundary (no synchronisation inside the
region) -> `step_ms` median / p10 / p90.  Per-kernel figures come from a separate instrumented pass (hipEvents around every
MFMA conv launch on the launch stream), never from the timed region.

Other flags: --workload generator|pipeline (only that workload), --path f32|f16x3|f16|bf16 (only that arithmetic as the top
level; f16 / bf16 = single-term reduced-precision operands of configs[4], tolerance 5e-2), --labels blocky|face,
--only-headline (skip the extra blocks), --batch, --size, --no-cpu-baseline.
"""
import argparse
import hashlib
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, exact f32 (at the 2.4 GHz boost clock)
PEAK_F16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: f16/bf16 MFMA dense peak (spec; the 2:1-sparse figure is not used)
PEAK_HBM_GBS = 8000.0
CONTRACT_BYTES_PER_IMAGE = 5.805e9        # conv-layer accounting, activations in + out, S=512, fp32
CONTRACT_WEIGHT_BYTES = 1.061e9           # ... + weights once per batch
DENSE_REFERENCE_FLOP_PER_IMAGE = 2.5696e12    # the reference's dense graph at S=512
PATH_OPTION = {'f32': 0, 'f16x3': 1, 'f16': 2, 'bf16': 3}

DTYPE = {
    'f32': 'f32 (every product and sum an IEEE f32 operation on the f32 matrix cores, v_mfma_f32_16x16x4_f32 / 32x32x2_f32; 3x3 convs as '
           'Winograd: ResBlock convs and the <= 64-pixel SPADE / style convs F(4x4,3x3), SPADE / style convs above F(2x2,3x3) -- f32 operands and accumulation over TRANSFORMED operands, '
           'not a re-association of the direct sum: <= 3e-5 from the reference fixtures on the benchmarked call; --wino 1 / 0 for F(2x2,3x3) / direct)',
    'f16x3': 'f32 storage + f32 accumulate; conv products as 3-term f16 split on MFMA with power-of-two operand scaling '
             '(2^-22 per product: f32-class, not the fp32 number of record; csrc/sh16.h)',
    'f16': 'f16 operands on MFMA, f32 accumulate, f32 normalisation/modulation (reduced precision: tolerance 5e-2)',
    'bf16': 'bf16 operands on MFMA, f32 accumulate, f32 normalisation/modulation (reduced precision: tolerance 5e-2)',
}


def csrc_sha(path='f32'):
    """Hash of the sources of the kernels whose launches roofline.traffic averages over on `path` (the SPADE conv + fused ACE
    epilogue set): profiles/latest_traffic.json records the one its PMC passes were taken at; a mismatch means the committed
    traffic figure is stale and is not reported.  Per path, and only those kernels' files, so that a late edit elsewhere in csrc/
    does not void the evidence of a kernel it did not touch (round 5: the driver line lost its traffic that way)."""
    files = {'f32': ('conv_mfma.h', 'conv_wino.h', 'conv_wino4.h', 'conv_wino4v.h', 'conv_ace_sparse.h'),
             'f32_plain': ('conv_wino.h', 'conv_wino4.h', 'conv_wino4v.h', 'conv_pw.h'),
             'f16x3': ('conv_sh16.h', 'sh16.h', 'ace_sparse.h')}.get(path, ('conv_sh16.h', 'sh16.h', 'ace_sparse.h'))
    h = hashlib.sha256()
    d = os.path.join(ROOT, 'ctrlhair_amd', 'csrc')
    for f in files:
        h.update(f.encode())
        h.update(open(os.path.join(d, f), 'rb').read())
    return h.hexdigest()[:16]


def cpu_baseline(ngf, S, sd_np, budget_s=120.0, weights=None, b16=False):
    """The oracle (torch fp32 CPU restatement of the reference path) timed on this host's cores on a bounded sample of the same
    workload (SURVEY.md 8d asks for Config 2 at B=1 and B=16 and for Config 1; B=16 of the dense reference graph takes minutes
    here, so the sample is: one 128x128 warm-up, single-image forwards at the benchmark size while the budget lasts (median), and
    -- when the full weight set is at hand -- ONE Config-1 edit (256x256 portrait through Backend.set_input_img / sliders /
    output on the CPU restatement of every network))."""
    import numpy as np
    import torch
    from ctrlhair_amd import procedural as P
    from oracle import sean_oracle as O          # baseline leg only -- never on the measured path
    sd = O.to_torch(sd_np)
    from ctrlhair_amd.hostutil import cpu_quota
    quota = cpu_quota()                          # (the box shows 256 cores to a container whose CFS quota is 16 CPUs: a pool of 128 threads
    if torch.get_num_threads() > quota:          #  spin-waits the quota away and the process is descheduled for most of every 100 ms period)
        torch.set_num_threads(quota)
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
        if time.time() - t0 + times[-1] > budget_s:
            break
    med = float(np.median(times))
    out = {'value': round(1.0 / med, 4), 'unit': 'images/s', 'cores': torch.get_num_threads(), 'cores_visible': os.cpu_count(), 'cpu_quota': quota, 'kind': 'port',
           'sample': f'{len(times)} single-image {S}x{S} generator forwards (median; the dense reference graph), torch '
                     f'{torch.__version__} CPU, after a 128x128 warm-up',
           'b1_seconds': [round(t, 2) for t in times],
           'b16': {'run': False, 'estimate_seconds_per_batch': round(16 * med, 1),
                   'why': 'no cross-sample op in the graph: a batch of 16 is 16 of these forwards; skipped by --no-cpu-b16'}}
    if b16:
        t = time.time()
        O.generator_forward(sd, P.blocky_labels(16, S), P.style_codes(16), P.noise_planes(16, S, ngf), ngf, weights_cache=wc)
        dt = time.time() - t
        out['b16'] = {'run': True, 'seconds_per_batch': round(dt, 1), 'images_per_s': round(16.0 / dt, 4)}
    if weights is not None:
        try:
            from ctrlhair_amd.ui.backend import Backend
            from tests.oracle_models import OracleModels      # CPU provider of every network (test infrastructure)
            be = Backend(2.5, blending=False, models=OracleModels(weights, ngf))
            img = np.ascontiguousarray(P.synthetic_images(1, 256, seed=11)[0].transpose(1, 2, 0))
            img = np.clip((img * 0.5 + 0.5) * 255.0, 0, 255).astype(np.uint8)
            t = time.time()
            be.set_input_img(img_rgb=img)
            be.change_curliness(1.0)
            be.change_texture(1.5, 0)
            be.change_shape(-1.0, 0)
            be.output()
            out['config1_edit'] = {'seconds': round(time.time() - t, 2), 'what': 'one 256x256 portrait: set_input_img (parse, shape / '
                                   'style / colour encoders) + three slider moves + output(), every network on the CPU oracle'}
        except Exception as e:          # never lose the run over the baseline
            out['config1_edit'] = {'error': f'{type(e).__name__}: {e}'}
    return out


def interactive_b1(ngf, S, sd_np, weights, dev):
    """BASELINE.json configs[0] on the GPU: ONE image at a time, the reference's own mode of use (hair_editor.py:159-179, ui/backend.py:147-175).
    Generator render latency (batch 1, S x S, inputs resident) on both arithmetic paths, and one whole Config-1 edit on the HIP Backend --
    the call sequence the CPU baseline times (cpu_baseline.config1_edit)."""
    import numpy as np
    import torch
    from ctrlhair_amd import procedural as P
    from ctrlhair_amd.sean.generator import SeanGenerator
    from ctrlhair_amd.hostutil import cap_threads_to_cpu_quota
    out = {'what': f'one {S}x{S} image per call (batch 1), median of 30 calls after 5 warm-up calls, inputs resident in HBM', 'unit': 'ms per image',
           'host_threads': cap_threads_to_cpu_quota()}
    for path, mode, opts in (('f32', 0, None), ('f16x3', 1, None), ('f32_batch_invariant', 0, {'sean.batch_invariant': 1})):
        g = SeanGenerator(dev.index or 0, f16x3=mode, options=opts).load_state_dict(sd_np, max_batch=1, max_size=S)
        l = torch.from_numpy(P.blocky_labels(1, S)).to(dev)
        c = torch.from_numpy(P.style_codes(1)).to(dev)
        n = torch.from_numpy(P.noise_planes(1, S, ngf)).to(dev)
        for _ in range(5):
            g.generate(l, c, n)
        torch.cuda.synchronize(dev)
        ts = []
        for _ in range(30):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.generate(l, c, n)
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1))
        out[path] = {'render_ms': round(float(np.median(ts)), 3), 'p10': round(float(np.percentile(ts, 10)), 3), 'p90': round(float(np.percentile(ts, 90)), 3)}
        g.handle.close()
        del g
        torch.cuda.empty_cache()
    if weights is not None:
        try:
            from ctrlhair_amd.ui.backend import Backend
            img = np.ascontiguousarray(P.synthetic_images(1, 256, seed=11)[0].transpose(1, 2, 0))
            img = np.clip((img * 0.5 + 0.5) * 255.0, 0, 255).astype(np.uint8)
            for path, f16 in (('f32', False), ('f16x3', True)):
                be = Backend(2.5, blending=False, weights=weights, device=dev.index or 0, f16x3=f16)
                ts = []
                for rep in range(9):
                    torch.cuda.synchronize(dev)
                    t = time.time()
                    be.set_input_img(img_rgb=img)
                    be.change_curliness(1.0)
                    be.change_texture(1.5, 0)
                    be.change_shape(-1.0, 0)
                    be.output()
                    torch.cuda.synchronize(dev)
                    ts.append((time.time() - t) * 1e3)
                out[path]['config1_edit_ms'] = round(float(np.median(ts[1:])), 2)       # (first repetition: warm-up)
                out[path]['config1_edit_ms_max'] = round(float(np.max(ts[1:])), 2)
                del be
                torch.cuda.empty_cache()
            out['config1_edit'] = ('one 256x256 portrait: set_input_img (parse, shape / style / colour encoders) + three slider moves + output() on the HIP '
                                   'Backend, host wall-clock incl. the cv2-equivalent host code; the same calls as cpu_baseline.config1_edit')
        except Exception as e:
            out['config1_edit'] = {'error': f'{type(e).__name__}: {e}'}
    return out


def make_labels(kind, B, S, first):
    import numpy as np
    from ctrlhair_amd import procedural as P
    if kind == 'face':
        return np.stack([P.face_like_labels(S, 500 + first + b) for b in range(B)])
    return P.blocky_labels(B, S, first=first)


class GeneratorJob:
    """configs[1]: one SEAN generator pass over this rank's batch."""

    def __init__(self, args, path, dev, rank, sd, labels='blocky'):
        import torch
        from ctrlhair_amd import procedural as P
        from ctrlhair_amd.sean.generator import SeanGenerator
        B, S, ngf = args.batch, args.size, args.ngf
        opts = {'sean.sparse': args.sparse, 'sean.wino': args.wino}
        if args.ahead >= 0:
            opts['sean.ahead'] = args.ahead
        if args.overlap >= 0:
            opts['sean.overlap'] = args.overlap
        for kv in (args.opt or []):              # experiments: any ch_set_option pair
            k, v = kv.split('=')
            opts[k] = int(v)
        if args.sparse_th:
            opts['sean.sparse_th'] = args.sparse_th
        if args.compact is not None:
            opts['sean.sh16_compact'] = args.compact
        if args.dbg:
            opts['sean.dbg'] = args.dbg          # (some experiment bits act at ch_finalize)
        self.gen = SeanGenerator(dev.index, f16x3=PATH_OPTION[path], options=opts).load_state_dict(sd, max_batch=B, max_size=S)
        if args.dbg:
            self.gen.handle.set_option('sean.dbg', args.dbg)
        self.dev, self.rank, self.args = dev, rank, args
        first = rank * B     # global sample index offset (SURVEY.md 8d Config 4)
        self.codes = torch.from_numpy(P.style_codes(B, first=first)).to(dev)
        self.noise = torch.from_numpy(P.noise_planes(B, S, ngf, first=first)).to(dev)
        self.set_labels(labels)
        self.handle = self.gen.handle
        self.images = B
        self.out_shape = (B, 3, S, S)

    def set_labels(self, kind):
        import torch
        a = self.args
        self.labels = torch.from_numpy(make_labels(kind, a.batch, a.size, self.rank * a.batch)).to(self.dev)

    def step(self, out):
        self.gen.generate(self.labels, self.codes, self.noise, out=out)

    def close(self):
        self.handle.close()


class PipelineJob:
    """configs[2]: the whole edit (ctrlhair_amd.pipeline.EditPipeline) over this rank's batch of portraits."""

    def __init__(self, args, path, dev, rank, weights, B):
        import torch
        from ctrlhair_amd import procedural as P
        from ctrlhair_amd.pipeline import EditPipeline
        S = args.size
        # throughput configuration: full run-ahead mode up to this batch (+1.6 % at 8 x 512^2 for 4.3 GB of per-ACE buffers)
        opts = {'sean.ahead': max(B, 2)}
        if args.dbg:
            opts['sean.dbg'] = args.dbg
        if args.compact is not None:
            opts['sean.sh16_compact'] = args.compact
        for kv in (args.opt or []):              # experiments: any ch_set_option pair of the generator
            k, v = kv.split('=')
            opts[k] = int(v)
        self.pipe = EditPipeline(weights, device=dev.index, img_size=S, max_batch=B, f16x3=PATH_OPTION[path], options=opts)
        self.img = torch.from_numpy(P.synthetic_images(B, S, seed=11 + rank * B)).to(dev)
        self.handle = self.pipe.models.generator.handle
        self.images = B
        self.scale = (S / 512.0) ** 2          # the per-image GFLOP figures below are quoted at 512 x 512; every stage is convolutional
        self.out_shape = (B, 3, S, S)

    def step(self, out):
        self.pipe.edit(self.img, out=out)

    # dense GFLOP per image of each stage (generator: the dense evaluation after the LUT reformulations) and the share of
    # them the matrix cores execute on the exact-f32 path where it is known analytically: Winograd F(2x2,3x3) runs 16 of 36 products
    # of a 3x3 stride-1 conv (Zencoder: the 256 -> 512 conv and the four phase convs of the ConvTranspose; BiSeNet / shape decoder:
    # their 3x3 stride-1 layers, the rest direct).  The generator's share is MEASURED (device-side counters of the work lists).
    STAGE_GFLOP = {'parse': 27.5, 'shape_encode': 4.9, 'zencoder': 169.6, 'shape_decode': 32.2, 'generator': 1083.8}
    F32_EXECUTED_SHARE = {'parse': 0.72, 'shape_encode': 1.0, 'zencoder': 0.48, 'shape_decode': 0.50}

    def stages(self, path):
        """Per-stage time and roofline of one edit (separate instrumented runs, torch events).  `frac` prices the FLOPs the matrix
        cores EXECUTED (f16x3: three f16 products per f32 product) against the peak of the unit that ran them."""
        import torch
        ms = self.pipe.stage_times(self.img)
        gen_share = None
        try:        # executed / dense of the generator stage on this workload's label maps (instrumented pass, not timed)
            self.handle.profile_enable(True)
            self.pipe.edit(self.img)
            torch.cuda.synchronize()
            self.handle.profile_enable(False)
            allk = self.handle.profile_read(-1)
            gen_share = allk['flops_executed'] / max(allk['flops'], 1.0)
        except Exception:
            gen_share = None
        out = {}
        for k, t in ms.items():
            row = {'ms': round(t, 3)}
            if k in self.STAGE_GFLOP:
                dense = self.STAGE_GFLOP[k] * self.scale * self.images / t            # GFLOP per ms = TFLOP/s
                f16 = path != 'f32'          # every stage's convs run on the f16 matrix cores unless the strict-f32 path is on
                share = gen_share if k == 'generator' else (1.0 if f16 else self.F32_EXECUTED_SHARE[k])
                row.update({'dense_tflops': round(dense, 1), 'bound': 'mfma', 'peak_tflops': PEAK_F16_MFMA_TFLOPS if f16 else PEAK_F32_MFMA_TFLOPS})
                if share is not None:
                    ex = dense * share * (3.0 if path == 'f16x3' else 1.0)
                    row.update({'executed_over_dense': round(share, 4), 'executed_tflops': round(ex, 1),
                                'executed_share_source': 'work-list counters' if k == 'generator' else ('3 f16 products per f32 product' if f16 else 'analytic (layer mix)'),
                                'frac': round(ex / row['peak_tflops'], 4)})
            else:
                row['bound'] = 'launch latency (three small MLPs + slider arithmetic)'
            out[k] = row
        return out

    def close(self):
        self.pipe.close()


def percentiles(ms):
    import numpy as np
    if not ms:
        return None
    a = np.asarray(ms, dtype=np.float64)
    return {'median': round(float(np.median(a)), 3), 'p10': round(float(np.percentile(a, 10)), 3),
            'p90': round(float(np.percentile(a, 90)), 3), 'n': int(a.size)}


def run_leg(job, args, dist, dev, world, steps=None, warmup=None, profile=True):
    """warm-up, then the timed region (barrier + synchronize on both sides, MAX over ranks), then a separate
    instrumented pass for the per-kernel figures."""
    import torch
    from ctrlhair_amd.parallel import PipelinedGather
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    pg = PipelinedGather(job.out_shape, torch.float32, dev, overlap=not args.sync_gather)

    def step():
        out = pg.begin()
        job.step(out)
        pg.submit()

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    sync()
    # host time to enqueue one step's launches, measured OUTSIDE the timed region on an EMPTY queue (synchronise, enqueue one step, stop the
    # clock, synchronise): inside the region the HIP queue is full and step() blocks on it -- that figure (kept below as
    # host_step_call_ms_in_region) is queue back-pressure, not enqueue cost (round 5 review)
    enq = []
    for _ in range(min(5, max(steps, 0))):
        sync()
        h0 = time.perf_counter()
        step()
        enq.append((time.perf_counter() - h0) * 1e3)
    sync()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    call_ms = []
    t0 = time.perf_counter()
    for i in range(steps):
        marks[i].record()            # on the launch stream; nothing waits on it inside the region
        h0 = time.perf_counter()
        step()
        call_ms.append((time.perf_counter() - h0) * 1e3)
    marks[steps].record()
    sync()
    dt = time.perf_counter() - t0
    step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)]
    if steps > 0:
        assert pg.check_slot(), 'all-gather result does not contain this rank\'s shard at its slot'
    out = pg.last_local()
    assert args.dbg or steps == 0 or bool(torch.isfinite(out).all())
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    prof = None
    if profile:
        # instrumented pass (not timed as a whole): hipEvent brackets around every MFMA conv launch, on the launch stream
        nprof = max(1, min(steps, 3))
        job.handle.profile_enable(True)
        scratch = pg.outs[0]
        for _ in range(nprof):
            job.step(scratch)
        torch.cuda.synchronize()
        job.handle.profile_enable(False)
        prof = {'ace': job.handle.profile_read(1), 'plain': job.handle.profile_read(0), 'interior': job.handle.profile_read(3)}
        prof['all'] = job.handle.profile_read(-1)
        prof['steps'] = nprof
    value = world * job.images * steps / dt if steps else 0.0
    return {'value': round(value, 3), 'ms_per_step': round(dt / max(steps, 1) * 1e3, 3), 'step_ms': percentiles(step_ms),
            'host_enqueue_ms_per_step': percentiles(enq), 'host_step_call_ms_in_region': percentiles(call_ms), 'steps': steps, 'warmup': warmup}, prof


def roofline_block(path, prof, value, B, sustained):
    ace, plain, allk, inter = prof['ace'], prof['plain'], prof['all'], prof['interior']
    n = max(ace['launches'], 1)
    t_ace = ace['ms'] * 1e-3
    dense = ace['flops'] / t_ace / 1e12 if t_ace > 0 else 0.0            # dense-equivalent f32 TFLOP/s of the SPADE convs
    useful = ace['flops_executed'] / t_ace / 1e12 if t_ace > 0 else 0.0  # f32 FLOPs the matrix cores were asked for
    if path in ('f16', 'bf16'):
        executed, peak, pk = useful, PEAK_F16_MFMA_TFLOPS, 'f16'
        kname = f'conv_sh16_ws_kernel / conv_sh16_kernel <KS=3,...,EPI_ACE,TERMS=1> (SPADE gamma/beta conv, {path} operands, fused ACE epilogue)'
    elif path == 'f16x3':
        executed, peak, pk = 3.0 * useful, PEAK_F16_MFMA_TFLOPS, 'f16'      # every f32 product is executed as 3 f16 MFMA products
        kname = 'conv_sh16_ws_kernel / conv_sh16_kernel <KS=3,...,EPI_ACE> (SPADE gamma/beta conv, f16x3 split operands, fused ACE epilogue)'
    else:
        executed, peak, pk = useful, PEAK_F32_MFMA_TFLOPS, 'f32'
        kname = ('wino_ace_gather_kernel (SPADE gamma/beta conv + style convs as Winograd F(2x2,3x3) on the exact-f32 matrix cores over '
                 'tasks of 64 boundary quads, fused ACE epilogue: the levels above 64 pixels) + wino4v_kernel<1> (the same conv as F(4x4,3x3) over '
                 'every tile with the hidden activations pre-transformed by wino4v_pack_kernel, csrc/conv_wino4v.h: 32 / 64 pixels; '
                 'sean.wino4v=0: wino4_ace_kernel) + conv_mfma_kernel<KS=3,...,EPI_ACE> for the ACEs below 32 pixels; --wino 0: conv_ace_sparse_kernel')
    # Which kernel set is the DOMINANT one?  Until round 6 the SPADE convs were (22 of 46 ms); with the straight-edge reduction (sean.edge,
    # csrc/ace_sparse.h) most of their boundary pixels leave the matrix cores and on label maps with straight region borders the ResBlock
    # convs (3x3 as Winograd F(4x4,3x3) + the learned 1x1 shortcuts) take more of the step: the block describes whichever set took longer
    # in this run, the other one is kept beside it (`other_set`).
    other = None
    tkey = path
    dom = 'spade_convs'
    if path == 'f32':
        spade_blk = {'set': 'spade_convs', 'kernel': kname, 'achieved': round(useful, 2), 'frac': round(useful / peak, 4), 'launches': ace['launches'],
                     'avg_launch_ms': round(ace['ms'] / n, 4), 'ms_per_step': round(ace['ms'] / prof['steps'], 3),
                     'executed_over_dense': round(ace['flops_executed'] / max(ace['flops'], 1.0), 4)}
        if plain['ms'] > ace['ms'] and plain['launches'] > 0:
            dom, tkey, other = 'resblock_convs', 'f32_plain', spade_blk
            ace = plain
            n = max(ace['launches'], 1)
            t_ace = ace['ms'] * 1e-3
            dense = ace['flops'] / t_ace / 1e12
            useful = executed = ace['flops_executed'] / t_ace / 1e12
            kname = ('wino4_plain_kernel (ResBlock 3x3 convs as Winograd F(4x4,3x3) on v_mfma_f32_16x16x4_f32, in-kernel input transform: the levels above '
                     '64 pixels) + wino4v_kernel<0> (the same with the input pre-transformed by wino4v_pack_kernel, csrc/conv_wino4v.h: G_middle / up_0) + '
                     'wino_plain_kernel (F(2x2,3x3): the 16-pixel head block) + pw_conv_kernel (the learned 1x1 shortcuts, csrc/conv_pw.h)')
        else:
            pt = plain['ms'] * 1e-3
            other = None if not plain['launches'] else {
                'set': 'resblock_convs', 'achieved': round(plain['flops_executed'] / pt / 1e12, 2),
                'frac': round(plain['flops_executed'] / pt / 1e12 / peak, 4), 'launches': plain['launches'],
                'avg_launch_ms': round(plain['ms'] / plain['launches'], 4), 'ms_per_step': round(plain['ms'] / prof['steps'], 3)}
    traffic = traffic_raw = detail = note = None
    tpath = os.path.join(ROOT, 'profiles', 'latest_traffic.json')
    if os.path.exists(tpath):      # HBM bytes per launch of the dominant kernel from committed rocprofv3 PMC passes
        try:
            detail = json.load(open(tpath)).get(tkey)
            if detail and detail.get('csrc_sha') == csrc_sha(tkey):
                traffic = round(float(detail['hbm_bytes']))
                traffic_raw = round(float(detail['fetch_raw']) + float(detail['write']))
            elif detail:
                note = 'PMC passes on file were taken at other kernel sources (csrc_sha mismatch): not reported'
                detail = None
        except Exception:
            traffic = detail = None
    alg_bytes = ace['bytes'] / n
    contract = CONTRACT_BYTES_PER_IMAGE + CONTRACT_WEIGHT_BYTES / B
    steps = prof['steps']
    sus = (sustained or {}).get(pk)
    blk = {
        'bound': 'mfma', 'dominant_set': dom, 'other_set': other, 'ms_per_step': round(ace['ms'] / prof['steps'], 3),
        'kernel': kname, 'achieved': round(executed, 2), 'peak': peak, 'unit': 'TFLOP/s',
        'frac': round(executed / peak, 4),
        'peak_sustained': sus, 'frac_of_sustained': round(executed / sus, 4) if sus else None,
        'peak_note': 'peak = spec at the 2.4 GHz boost clock; peak_sustained = MFMA-only loop measured on this device in this run '
                     '(ch_mfma_peak)',
        'traffic': traffic, 'traffic_unit': 'HBM bytes per launch (FETCH_SIZE x2 + WRITE_SIZE: the guide\'s gfx950 correction, an upper bound)',
        'traffic_uncorrected': traffic_raw,
        'algorithmic_bytes_per_launch': round(alg_bytes),
        'algorithmic_bytes_note': 'sparse ACE launches: hidden activations + x + output of the BOUNDARY pixels only (the minimum); the exact-f32 '
                                  'gather kernel fetches the 4 x 4 patch of every boundary quad once per pair of row tiles (overlapping patches, '
                                  'mostly L2 hits), the f16x3 kernel stages the patch of every tile that holds a boundary pixel',
        'traffic_ratio': round(traffic / alg_bytes, 3) if (traffic and alg_bytes > 0) else None,
        'traffic_ratio_uncorrected': round(traffic_raw / alg_bytes, 3) if (traffic_raw and alg_bytes > 0) else None,
        'traffic_detail': detail, 'traffic_note': note,
        'launches': ace['launches'], 'avg_launch_ms': round(ace['ms'] / n, 4),
        'flops_executed_per_launch_avg': ace['flops_executed'] / n, 'flops_dense_per_launch_avg': ace['flops'] / n,
        'executed_over_dense': round(ace['flops_executed'] / max(ace['flops'], 1.0), 4),
        'dense_equivalent_f32_tflops': round(dense, 2),
        'note': 'executed = FLOPs the matrix cores ran (exact SPADE-interior reduction: pixels with a uniform 5x5 label '
                'neighbourhood take per-label constants, csrc/ace_sparse.h; Winograd F(2x2,3x3): 16 products per quad and channel '
                'instead of 36, csrc/conv_wino.h); utilisation is priced on executed FLOPs only',
        'all_mfma_convs': {
            'executed_tflops': round(allk['flops_executed'] / max(allk['ms'], 1e-9) / 1e9, 2),
            'ms_per_step': round(allk['ms'] / steps, 3),
            'plain_tflops': round(plain['flops_executed'] / max(plain['ms'], 1e-9) / 1e9, 2),
            'flops_executed_per_image': allk['flops_executed'] / steps / B,
            'flops_dense_per_image_after_lut': allk['flops'] / steps / B,
            'flops_dense_reference_per_image': DENSE_REFERENCE_FLOP_PER_IMAGE},
        'interior_pass': None if not inter['launches'] else {
            'bound': 'hbm', 'kernel': 'ace_gtable_kernel + ace_interior_*_kernel (elementwise modulation of the interior pixels)',
            'ms_per_step': round(inter['ms'] / steps, 3), 'algorithmic_gb_per_step': round(inter['bytes'] / steps / 1e9, 3),
            'achieved_gbs': round(inter['bytes'] / max(inter['ms'], 1e-9) / 1e6, 1), 'peak_gbs': PEAK_HBM_GBS,
            'frac': round(inter['bytes'] / max(inter['ms'], 1e-9) / 1e6 / PEAK_HBM_GBS, 4)},
        'timing': 'hipEvents around each launch in a separate instrumented pass (not the timed region)',
        # north_star's other yardstick: conv-layer contract bytes (each conv reads its input and writes its output once,
        # fp32) against HBM peak.  The path is matrix-core bound (AI ~ 440 FLOP/B): this fraction cannot reach 55 %.
        'hbm_contract': {'bytes_per_image': contract, 'achieved_gbs': round(value * contract / 1e9, 1),
                         'peak_gbs': PEAK_HBM_GBS, 'frac': round(value * contract / 1e9 / PEAK_HBM_GBS, 4)},
    }
    return blk


def respawn_under_torchrun(n):
    """--gpus N without a launcher: start N ranks ourselves (one per GPU) and let rank 0 print the JSON line."""
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < n:
        raise SystemExit(f'bench.py --gpus {n}: only {have} GPU(s) visible on this node -- refusing to run a smaller job '
                         f'under the label n_gpus={n}')
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execvpe(sys.executable, cmd, env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=0, help='per GPU; default 16 for every --gpus N (generator) / 8 (pipeline); the 32-per-GPU '
                    'shape of configs[3] rides along as the block configs3_b32_per_gpu at N = 1 and N = 8')
    ap.add_argument('--size', type=int, default=512)
    ap.add_argument('--ngf', type=int, default=64)
    ap.add_argument('--workload', choices=('all', 'generator', 'pipeline'), default='all')
    ap.add_argument('--labels', choices=('blocky', 'face'), default='blocky', help='label maps of the top-level generator legs')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-b16', action='store_true', help='(default since round 6; kept for old command lines)')
    ap.add_argument('--no-cpu-b16', action='store_true', help='CPU baseline: skip the ONE real batch-16 forward of the dense graph (SURVEY.md 8(d): Config 2 at B=16; about a minute of host time)')
    ap.add_argument('--only-headline', action='store_true', help='top-level leg only (no f16x3 / face-like / pipeline blocks)')
    ap.add_argument('--no-strict-fp32', action='store_true', help=argparse.SUPPRESS)      # (older tools: implies --only-headline)
    ap.add_argument('--path', choices=tuple(PATH_OPTION), default='f32',
                    help='arithmetic of the top-level leg: f32 = exact-f32 MFMA (default, the reference\'s arithmetic); f16x3 = 3-term '
                         'split-operand f16 MFMA, f32 accumulate; f16 / bf16 = single-term reduced-precision operands (configs[4])')
    ap.add_argument('--dbg', type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument('--ahead', type=int, default=-1, help=argparse.SUPPRESS)       # option sean.ahead (experiments)
    ap.add_argument('--opt', action='append', help=argparse.SUPPRESS)      # key=value pairs for ch_set_option (experiments)
    ap.add_argument('--overlap', type=int, default=-1, help='exact-f32 path: CUs of the side streams that run the HBM-bound kernels beside the '
                    'convs (option sean.overlap; 0 = serial schedule; default: the library\'s)')
    ap.add_argument('--sparse', type=int, default=1, help='0: every pixel through the SPADE convs (no interior reduction)')
    ap.add_argument('--wino', type=int, default=2, help='exact-f32 path (option sean.wino): 2 = ResBlock convs as Winograd F(4x4,3x3), SPADE convs as '
                    'F(2x2,3x3) (default); 1 = F(2x2,3x3) everywhere; 0 = 3x3 convs evaluated directly')
    ap.add_argument('--sparse-th', type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument('--compact', type=int, default=None, help=argparse.SUPPRESS)
    ap.add_argument('--sync-gather', action='store_true',
                    help='N > 1: all-gather each step on the compute stream instead of overlapping it with the next step')
    ap.add_argument('--force-dist', action='store_true', help='1-rank process group: exercises the RCCL code path on one GPU')
    args = ap.parse_args()
    if args.no_strict_fp32:
        args.only_headline = True
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        respawn_under_torchrun(args.gpus)
    # ONE per-GPU batch for every N of a scaling curve (N = 1 must be the workload of N = 8)
    gen_batch = args.batch if args.batch > 0 else 16
    user_batch = args.batch
    pipe_batch = args.batch if args.batch > 0 else 8

    import torch
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.gpus != world:
        raise SystemExit(f'--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the HIP path has no CPU fallback)')
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f'rank {rank}: no GPU {local_rank} on this node ({torch.cuda.device_count()} visible)')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    # host thread pool: torch sizes it by the visible cores; under a container CPU quota (16 CPUs on the 1-GPU box) the pool's spin-waits get
    # the process descheduled for most of a 100 ms period -- with N ranks on one node each rank gets its share of the quota
    from ctrlhair_amd.hostutil import cpu_quota
    torch.set_num_threads(max(1, min(torch.get_num_threads(), cpu_quota() // max(1, world))))
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        world = dist.get_world_size()            # what RCCL actually sees

    S, ngf = args.size, args.ngf
    do_gen = args.workload in ('all', 'generator')
    do_pipe = args.workload == 'pipeline' or (args.workload == 'all' and world == 1 and not args.only_headline)
    extras = world == 1 and not args.only_headline
    res = {}
    sustained = None
    sd = None
    all_weights = None

    if do_gen:
        from ctrlhair_amd import procedural as P
        args.batch = gen_batch
        sd = P.sean_state_dict(0, ngf)
        legs = [args.path] + (['f16x3'] if (extras and args.path == 'f32') else [])
        blocks = {}
        for path in legs:
            job = GeneratorJob(args, path, dev, rank, sd, labels=args.labels)
            if sustained is None and rank == 0:
                try:
                    sustained = {'f32': round(job.handle.mfma_peak(0, 30), 1), 'f16': round(job.handle.mfma_peak(1, 30), 1),
                                 'unit': 'TFLOP/s', 'how': 'MFMA-only loop, 2 waves per SIMD, random operands, ~30 ms (ch_mfma_peak)'}
                except Exception as e:         # never lose the run over the helper
                    sustained = {'error': str(e)}
            r, prof = run_leg(job, args, dist, dev, world)
            r['roofline'] = roofline_block(path, prof, r['value'] / world, gen_batch, sustained)
            r['dtype'] = DTYPE[path]
            if extras and args.labels == 'blocky':      # the same handle on face-like label maps (shorter run)
                job.set_labels('face')
                rf, pf = run_leg(job, args, dist, dev, world, steps=max(5, args.steps // 3), warmup=max(2, args.warmup // 3))
                rb = roofline_block(path, pf, rf['value'], gen_batch, sustained)
                rf.update({'executed_over_dense_spade': rb['executed_over_dense'],
                           'flops_executed_per_image': rb['all_mfma_convs']['flops_executed_per_image'],
                           'dominant_kernel_tflops': rb['achieved'], 'dominant_kernel_frac': rb['frac']})
                r['face'] = rf
            blocks[path] = r
            job.close()
            del job
            torch.cuda.empty_cache()
        head = blocks[args.path]
        # side legs of the exact-f32 number of record (shorter runs, same protocol): the per-GPU shape of configs[3] (32 images
        # per GPU), and at N = 1 the same job without the Winograd convs and without the SPADE-interior reduction (worst case)
        side = {}
        if args.path == 'f32' and (extras or world == 8) and user_batch == 0:
            variants = [('configs3_b32_per_gpu', {'batch': 32}), ('configs4_bf16_b32_per_gpu', {'batch': 32, '_path': 'bf16'})]
            if extras:
                variants += [('winograd_f2x2_only', {'wino': 1}), ('direct_convs_no_winograd', {'wino': 0}),
                             ('dense_worst_case_no_interior_reduction', {'sparse': 0}),
                             # the label-INDEPENDENT path: every level's SPADE convs as dense F(4x4,3x3) (what an adversarial label map costs at most)
                             ('dense_all_levels_f4x4', {'sparse': 0, 'opt': (args.opt or []) + ['sean.wino4_ace=512']}),
                             # what bit-identical results across batch sizes cost at B = 16 (option sean.batch_invariant: no split-K, no sample-pair
                             # tiles at 16 pixels; the F(4x4) / F(2x2) rule does not bind at this size)
                             ('batch_invariant_mode', {'opt': (args.opt or []) + ['sean.batch_invariant=1']}),
                             # round 6's two label-dependent reductions off: straight-edge pixels through the boundary conv again, patches from the planes
                             ('no_straight_edge_reduction', {'opt': (args.opt or []) + ['sean.edge=0']})]
            for name, over in variants:
                a2 = argparse.Namespace(**vars(args))
                vpath = over.get('_path', 'f32')
                for k, v in over.items():
                    if not k.startswith('_'):
                        setattr(a2, k, v)
                try:
                    job = GeneratorJob(a2, vpath, dev, rank, sd, labels=args.labels)
                    r2, p2 = run_leg(job, a2, dist, dev, world, steps=max(5, args.steps // 3), warmup=max(2, args.warmup // 3))
                    rb = roofline_block(vpath, p2, r2['value'] / world, a2.batch, sustained)
                    side[name] = {'value': r2['value'], 'unit': 'images/s', 'ms_per_step': r2['ms_per_step'], 'steps': r2['steps'],
                                  'batch_per_gpu': a2.batch, 'global_batch': world * a2.batch,
                                  'dtype': DTYPE[vpath] if (vpath != 'f32' or a2.wino == 2) else
                                  ('f32, IEEE f32 products and sums on the f32 matrix cores; ' +
                                   {0: 'direct 3x3 convs (no Winograd: the fmaf chains of the reference)', 1: 'every 3x3 conv as Winograd F(2x2,3x3)'}[min(max(a2.wino, 0), 1)]),
                                  'all_mfma_convs': rb['all_mfma_convs'], 'dominant_kernel_tflops': rb['achieved'],
                                  'dominant_kernel_frac': rb['frac'], 'executed_over_dense_spade': rb['executed_over_dense']}
                    if vpath == 'bf16':
                        side[name]['note'] = ('BASELINE.json configs[4] per-GPU shape: bf16 operands on MFMA, f32 accumulate; tolerance 5e-2 '
                                              '(tests/test_hip_sean_generator.py [bf16-32]); no reference counterpart (normalization.py:139,153 fail under autocast)')
                    job.close()
                    del job
                except Exception as e:           # e.g. not enough memory for the 32-image handle next to another process
                    side[name] = {'error': f'{type(e).__name__}: {e}'}
                torch.cuda.empty_cache()
        par = f'batch-sharded x{world}'
        if dist is not None:
            par += ' + RCCL all-gather of outputs' + ('' if args.sync_gather else ' overlapped with the next step')
        cfgn = 'configs[1] per GPU' if gen_batch == 16 else f'{gen_batch} per GPU'
        res = {
            'metric': '512x512 edited images/sec (SEAN generator forward), whole job', 'value': head['value'], 'unit': 'images/s',
            'n_gpus': world, 'steps': head['steps'], 'warmup': head['warmup'], 'ms_per_step': head['ms_per_step'],
            'step_ms': head['step_ms'], 'host_enqueue_ms_per_step': head['host_enqueue_ms_per_step'],
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': DTYPE[args.path],
            'data': f'synthetic ({args.labels} labels, tanh-normal codes, explicit noise planes; procedural calibrated weights, '
                    'no checkpoint ships with the reference)',
            'config': {'workload': f'SEAN generator forward only, batch {gen_batch}/GPU, {S}x{S}, ngf={ngf} (BASELINE.json {cfgn})',
                       'global_batch': world * gen_batch, 'conv_path': args.path, 'parallelism': par, 'labels': args.labels,
                       'spade_interior_reduction': bool(args.sparse), 'winograd': ({0: 'off', 1: 'F(2x2,3x3)', 2: 'ResBlock convs and the SPADE convs up to 64 pixels F(4x4,3x3), SPADE convs above F(2x2,3x3) over boundary quads'}[min(max(args.wino, 0), 2)] if args.path == 'f32' else 'n/a')},
            'roofline': head['roofline'], 'sustained_peaks': sustained,
        }
        res.update(side)
        if 'f16x3' in blocks and args.path != 'f16x3':
            b = blocks['f16x3']
            res['f32_class_f16x3'] = {k: b[k] for k in ('value', 'ms_per_step', 'step_ms', 'steps', 'warmup', 'dtype', 'roofline')}
            res['f32_class_f16x3']['unit'] = 'images/s'
            res['f32_class_f16x3']['note'] = 'same job, same timed protocol; narrower than fp32 per product, reported beside the fp32 number'
        if any('face' in b for b in blocks.values()):
            res['face_like_labels'] = {'labels': 'ctrlhair_amd.procedural.face_like_labels (large regions, curved boundaries), same '
                                                 'batch / size / weights / codes / noise', 'unit': 'images/s'}
            for path, b in blocks.items():
                if 'face' in b:
                    res['face_like_labels'][path] = b['face']

    if do_pipe:
        from ctrlhair_amd.hair_editor import procedural_weights
        args.batch = pipe_batch
        weights = procedural_weights(0, ngf)
        all_weights = weights
        if sd is None:
            sd = weights['sean']
        if args.workload == 'all':
            plegs = ['f16x3', 'f32']
        else:           # --workload pipeline: the top level follows --path; the other f32-class arithmetic rides along as a block
            other = 'f16x3' if args.path == 'f32' else 'f32'
            plegs = [args.path] + ([] if args.only_headline else [other])
        pb = {}
        for path in plegs:
            job = PipelineJob(args, path, dev, rank, weights, pipe_batch)
            r, _ = run_leg(job, args, dist, dev, world, profile=False)
            if rank == 0:
                r['stages'] = job.stages(path)
            r['dtype'] = DTYPE[path]
            pb[path] = r
            job.close()
            del job
            torch.cuda.empty_cache()
        wl = (f'full CtrlHair edit (BiSeNet parse -> shape + colour/texture branches -> Zencoder -> SEAN generator), '
              f'batch {pipe_batch}/GPU, {S}x{S}, ngf={ngf} (BASELINE.json configs[2])')
        if args.workload == 'pipeline':
            head = pb[plegs[0]]
            res = {'metric': '512x512 edited images/sec (full pipeline), whole job', 'value': head['value'], 'unit': 'images/s',
                   'n_gpus': world, 'steps': head['steps'], 'warmup': head['warmup'], 'ms_per_step': head['ms_per_step'],
                   'step_ms': head['step_ms'], 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
                   'dtype': DTYPE[plegs[0]], 'data': 'synthetic portraits; procedural calibrated weights',
                   'config': {'workload': wl, 'global_batch': world * pipe_batch, 'conv_path': plegs[0]},
                   'stages': head.get('stages')}
            if len(plegs) > 1:
                res['f32_class_f16x3' if plegs[1] == 'f16x3' else 'strict_fp32'] = pb[plegs[1]]
        else:
            res['pipeline'] = {'workload': wl, 'unit': 'images/s', 'metric': '512x512 edited images/sec (full pipeline)'}
            for path, r in pb.items():
                res['pipeline'][path] = r

    if rank == 0 and do_gen and extras and world == 1 and user_batch == 0:
        try:
            res['interactive_b1'] = interactive_b1(ngf, S, sd, all_weights, dev)
        except Exception as e:          # never lose the line over a side block
            res['interactive_b1'] = {'error': f'{type(e).__name__}: {e}'}
    if rank == 0 and do_gen and not args.no_cpu_baseline and world == 1:
        res['cpu_baseline'] = cpu_baseline(ngf, S, sd, weights=all_weights, b16=not args.no_cpu_b16)
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
