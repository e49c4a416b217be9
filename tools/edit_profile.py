"""Where one Config-1 edit (set_input_img + three slider moves + output()) spends its wall-clock on the HIP Backend:
    python tools/edit_profile.py [f32|f16x3]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctrlhair_amd import procedural as P
from ctrlhair_amd.hair_editor import procedural_weights
from ctrlhair_amd.ui.backend import Backend

path = sys.argv[1] if len(sys.argv) > 1 else 'f32'
w = procedural_weights(0, 64)
be = Backend(2.5, blending=False, weights=w, device=0, f16x3=(path == 'f16x3'))
if len(sys.argv) > 2 and sys.argv[2] == 'serial':      # everything on one stream (Backend.overlap)
    be.overlap = False
img = np.ascontiguousarray(P.synthetic_images(1, 256, seed=11)[0].transpose(1, 2, 0))
img = np.clip((img * 0.5 + 0.5) * 255.0, 0, 255).astype(np.uint8)
steps = [('set_input_img', lambda: be.set_input_img(img_rgb=img)), ('change_curliness', lambda: be.change_curliness(1.0)),
         ('change_texture', lambda: be.change_texture(1.5, 0)), ('change_shape', lambda: be.change_shape(-1.0, 0)), ('output', lambda: be.output())]
for rep in range(14):
    ts = []
    for name, fn in steps:
        torch.cuda.synchronize()
        t = time.time()
        fn()
        torch.cuda.synchronize()
        ts.append((name, (time.time() - t) * 1e3))
    if rep:
        print(path, ' '.join(f'{n} {v:.2f} ms' for n, v in ts), f'| total {sum(v for _, v in ts):.2f} ms')
