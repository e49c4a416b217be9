"""Run-to-run determinism stress of the exact-f32 generator (round 6): the same call repeated must give identical images, with the
straight-edge reduction / patch source on and off.   python tools/determinism_stress.py [reps]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctrlhair_amd import procedural as P
from ctrlhair_amd.sean.generator import SeanGenerator
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ngf, S, B = 64, 512, 2
sd = P.sean_state_dict(0, ngf)
codes, noise = P.style_codes(B, seed=81), P.noise_planes(B, S, ngf, seed=82)
sets = {'blocky': P.blocky_labels(B, S, grid=8), 'face': np.stack([P.face_like_labels(S, 40 + b) for b in range(B)])}
diag = (np.add.outer(np.arange(S), np.arange(S)) % 19).astype(np.uint8)
sets['diag'] = np.repeat(diag[None], B, 0)
for edge, patch in ((1, 1), (1, 0), (0, 1), (0, 0)):
    g = SeanGenerator(0, f16x3=0, options={'sean.wino4_force': 1, 'sean.edge': edge, 'sean.patch': patch}).load_state_dict(sd, max_batch=B, max_size=S)
    dev = g.device
    for name, lab in sets.items():
        l, c, n = torch.from_numpy(lab).to(dev), torch.from_numpy(codes).to(dev), torch.from_numpy(noise).to(dev)
        ref = None
        bad = 0
        worst = 0.0
        for r in range(reps):
            out = g.generate(l, c, n)
            torch.cuda.synchronize()
            o = out.cpu().numpy()
            if ref is None:
                ref = o
            elif not np.array_equal(ref, o):
                bad += 1
                worst = max(worst, float(np.abs(ref - o).max()))
        print(f'edge={edge} patch={patch} {name}: {bad} of {reps - 1} repeats differ (max {worst:.3e})', flush=True)
    g.handle.close()
