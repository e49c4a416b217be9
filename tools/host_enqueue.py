#!/usr/bin/env python3
"""Host-side enqueue time vs device time of one EditPipeline.edit() (B=8, 512x512): shows whether the host keeps ahead of
the GPU (enqueue << total) or a hidden synchronisation / launch-bound section exists.  Run on the GPU box."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctrlhair_amd.pipeline import EditPipeline  # noqa: E402

pipe = EditPipeline(img_size=512, max_batch=8, f16x3=1)
img = (torch.rand(8, 3, 512, 512) * 2 - 1).cuda()
out = torch.empty(8, 3, 512, 512, device='cuda')
for _ in range(3):
    pipe.edit(img, out=out)
torch.cuda.synchronize()
for n in (1, 5):
    t0 = time.perf_counter()
    for _ in range(n):
        pipe.edit(img, out=out)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'{n} edits: host enqueue {1e3 * (t1 - t0) / n:.2f} ms/edit, until idle {1e3 * (t2 - t0) / n:.2f} ms/edit')
# per-stage host time
m = pipe.models
torch.cuda.synchronize()
ts = {}
def hst(name, fn):
    t0 = time.perf_counter(); r = fn(); ts[name] = 1e3 * (time.perf_counter() - t0); return r
labels = hst('parse', lambda: pipe.parse(img))
lat = hst('analyse', lambda: pipe.analyse(img, labels))
from ctrlhair_amd.pipeline import DEFAULT_SLIDERS
lat = hst('sliders', lambda: pipe.apply_sliders(lat, DEFAULT_SLIDERS))
hst('render', lambda: pipe.render(lat, seed=1, out=out))
t0 = time.perf_counter(); torch.cuda.synchronize(); ts['drain'] = 1e3 * (time.perf_counter() - t0)
print({k: round(v, 2) for k, v in ts.items()})
