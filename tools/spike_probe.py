import os, sys, time, gc, cProfile, pstats, io
import numpy as np, torch
sys.path.insert(0, '/root/repo')
from ctrlhair_amd import procedural as P
from ctrlhair_amd.hair_editor import procedural_weights
from ctrlhair_amd.ui.backend import Backend
w = procedural_weights(0, 64)
be = Backend(2.5, blending=False, weights=w, device=0, f16x3=False)
img = np.ascontiguousarray(P.synthetic_images(1, 256, seed=11)[0].transpose(1, 2, 0))
img = np.clip((img * 0.5 + 0.5) * 255.0, 0, 255).astype(np.uint8)
be.set_input_img(img_rgb=img); be.output()
if len(sys.argv) > 1: gc.disable()
ts = []
slow = None
for i in range(40):
    torch.cuda.synchronize(); t = time.time()
    pr = cProfile.Profile(); pr.enable()
    be.output()
    torch.cuda.synchronize()
    pr.disable()
    dt = (time.time() - t) * 1e3
    ts.append(dt)
    if dt > 40 and slow is None:
        s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(14); slow = s.getvalue()
print('output() ms:', ' '.join(f'{t:.0f}' for t in ts), 'reserved MB', torch.cuda.memory_reserved() >> 20)
print(slow[:3000] if slow else 'no slow call')
