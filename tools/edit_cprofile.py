"""cProfile of the host side of one Config-1 edit on the HIP Backend (which Python / numpy calls the wall-clock of an edit goes to):
    python tools/edit_cprofile.py [f32|f16x3] [set_input_img|output|change_shape]"""
import cProfile
import os
import pstats
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctrlhair_amd import procedural as P
from ctrlhair_amd.hair_editor import procedural_weights
from ctrlhair_amd.ui.backend import Backend

path = sys.argv[1] if len(sys.argv) > 1 else 'f32'
what = sys.argv[2] if len(sys.argv) > 2 else 'set_input_img'
be = Backend(2.5, blending=False, weights=procedural_weights(0, 64), device=0, f16x3=(path == 'f16x3'))
img = np.ascontiguousarray(P.synthetic_images(1, 256, seed=11)[0].transpose(1, 2, 0))
img = np.clip((img * 0.5 + 0.5) * 255.0, 0, 255).astype(np.uint8)
calls = {'set_input_img': lambda: be.set_input_img(img_rgb=img), 'change_shape': lambda: be.change_shape(-1.0, 0), 'output': lambda: be.output()}
for _ in range(5):
    for fn in calls.values():
        fn()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    calls[what]()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(45)
