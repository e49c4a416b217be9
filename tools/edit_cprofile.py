import os, sys, time, cProfile, pstats, io
import numpy as np, torch
sys.path.insert(0, '/root/repo')
from ctrlhair_amd import procedural as P
from ctrlhair_amd.hair_editor import procedural_weights
from ctrlhair_amd.ui.backend import Backend
be = Backend(2.5, blending=False, weights=procedural_weights(0, 64), device=0, f16x3=False)
img = np.ascontiguousarray(P.synthetic_images(1, 256, seed=11)[0].transpose(1, 2, 0))
img = np.clip((img * 0.5 + 0.5) * 255.0, 0, 255).astype(np.uint8)
for _ in range(3): be.set_input_img(img_rgb=img); be.change_shape(-1.0, 0); be.output()
pr = cProfile.Profile(); pr.enable()
for _ in range(10): be.set_input_img(img_rgb=img); be.change_shape(-1.0, 0); be.output()
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(22); print(s.getvalue()[:4500])
