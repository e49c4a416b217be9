#!/usr/bin/env python3
"""Per-stage timings of the CtrlHair path on one MI355X (Zencoder, BiSeNet, shape enc/dec, colour MLPs,
and BASELINE config 3 = the full pipeline at batch 8).  Writes one JSON document to stdout.

    python tools/bench_stages.py > profiles/r01_stages.json
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ctrlhair_amd import procedural as P           # noqa: E402
from ctrlhair_amd.hair_editor import HipModels, procedural_weights   # noqa: E402


def timeit(fn, warm=2, it=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


def main():
    B, S = 8, 512
    dev = torch.device('cuda', 0)
    w = procedural_weights(0, 64)
    m = HipModels(w, device=0, img_size=S, max_batch=B, f16x3=False)          # exact-f32 SEAN path
    from ctrlhair_amd.sean.generator import SeanGenerator
    gen16 = SeanGenerator(0, f16x3=1).load_state_dict(w['sean'], max_batch=B, max_size=S)
    img = torch.from_numpy(P.synthetic_images(B, S)).to(dev)
    lab512 = torch.from_numpy(P.blocky_labels(B, S)).to(dev)
    lab256 = lab512[:, ::2, ::2].contiguous()
    codes = torch.from_numpy(P.style_codes(B)).to(dev)
    noise = torch.from_numpy(P.noise_planes(B, S, 64)).to(dev)
    out = {'device': torch.cuda.get_device_name(0), 'batch': B, 'size': S, 'unit': 'ms per batch'}

    out['zencoder_f32'] = timeit(lambda: m.generator.encode(img, lab512))
    out['zencoder_f16x3_ms'] = timeit(lambda: gen16.encode(img, lab512))
    out['bisenet_f32'] = timeit(lambda: m.face_parsing.parse_tensor(img))
    out['shape_encode_f32'] = timeit(lambda: m.mask_generator.encode_labels(lab256))
    hc, fc = m.mask_generator.encode_labels(lab256)
    out['shape_decode_f32'] = timeit(lambda: m.mask_generator.decode_labels(hc, fc))
    out['shape_decode_f32_batch1'] = timeit(lambda: m.mask_generator.decode_labels(hc[:1], fc[:1]))
    hair = codes[:, 13].contiguous()
    def color():
        d = m.solver_feature.dis({'code': hair})
        r = m.solver_feature.rgb_model({'code': hair})
        return m.solver_feature.gen({'noise': d['noise'], 'noise_curliness': d['noise_curliness'], 'rgb_mean': r['rgb_mean'],
                                     'pca_std': r['pca_std']})['code']
    out['color_mlps_f32'] = timeit(color)
    out['sean_generator_f32'] = timeit(lambda: m.generator.generate(lab512, codes, noise), warm=1, it=3)
    out['sean_generator_f16x3'] = timeit(lambda: gen16.generate(lab512, codes, noise), warm=1, it=5)

    # single-image latency of the render call behind Backend.output() (ui/backend.py:147-175), explicit noise
    for s_ in (256, 512):
        l1 = lab512[:1, ::(512 // s_), ::(512 // s_)].contiguous()
        n1 = torch.from_numpy(P.noise_planes(1, s_, 64)).to(dev)
        out[f'sean_generator_f16x3_batch1_{s_}'] = timeit(lambda: gen16.generate(l1, codes[:1], n1), warm=2, it=10)
        out[f'sean_generator_f32_batch1_{s_}'] = timeit(lambda: m.generator.generate(l1, codes[:1], n1), warm=2, it=5)

    # BASELINE config 3: full pipeline, batch 8 (BiSeNet@512 -> remap -> nearest 256 -> shape enc -> Zencoder@512 ->
    # colour enc/pred/gen with a slider delta -> shape dec -> nearest x2 -> generator@512), blending off.
    def pipeline():
        lab, _ = m.face_parsing.parse_tensor(img)
        l256 = lab[:, ::2, ::2].contiguous()
        hcode, fcode = m.mask_generator.encode_labels(l256)
        c = gen16.encode(img, lab)
        h = c[:, 13].contiguous()
        d = m.solver_feature.dis({'code': h})
        r = m.solver_feature.rgb_model({'code': h})
        c[:, 13] = m.solver_feature.gen({'noise': d['noise'] + 0.5, 'noise_curliness': d['noise_curliness'] + 1.0,
                                         'rgb_mean': r['rgb_mean'], 'pca_std': r['pca_std']})['code']
        newlab = m.mask_generator.decode_labels(hcode - 0.1, fcode)
        lab_up = newlab.repeat_interleave(2, 1).repeat_interleave(2, 2).contiguous()
        return gen16.generate(lab_up, c.contiguous(), None, seed=1)
    t = timeit(pipeline, warm=1, it=5)
    out['pipeline_config3_f16x3_ms'] = t
    out['pipeline_config3_images_per_s'] = B / t * 1e3
    # blending step after the generator (8f N3): mask construction + Poisson CG, one image
    import time
    from ctrlhair_amd.blending import PoissonBlender
    from oracle import poisson_oracle as PO         # CPU baseline leg only
    blender = PoissonBlender(m.generator.handle, m.device)
    for s_ in (256, 512):
        ys, xs = np.mgrid[0:s_, 0:s_]
        hair = ((ys - 0.3 * s_) ** 2 / (0.28 * s_) ** 2 + (xs - 0.5 * s_) ** 2 / (0.33 * s_) ** 2 <= 1).astype(np.uint8)
        src = ((P.synthetic_images(1, s_, seed=5)[0].transpose(1, 2, 0) * 0.5 + 0.5) * 247 + 4).astype(np.uint8)
        tgt = np.clip(src.astype(np.int32) + 17, 4, 251).astype(np.uint8)
        st, tt = torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda()
        mt = torch.from_numpy(1 - hair).cuda()
        out[f'poisson_blend_{s_}_ms'] = timeit(lambda: blender(st, tt, mt), warm=2, it=5)
        out[f'poisson_blend_{s_}_cg_iterations'] = blender.last_iters
        if s_ == 256:
            t0 = time.time()
            PO.poisson_blending(src, tgt, 1 - hair)
            out['poisson_blend_256_cpu_oracle_ms'] = (time.time() - t0) * 1e3
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
