#!/usr/bin/env python3
"""A/B of EditPipeline.edit's stream usage on one box (B=8, 512x512): sequential / shape branch on a side stream /
shape branch + BiSeNet underneath the Zencoder.  ms per 8 edits, median of 5 runs of 10 edits."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctrlhair_amd.pipeline import EditPipeline  # noqa: E402

pipe = EditPipeline(img_size=512, max_batch=8, f16x3=1)
img = (torch.rand(8, 3, 512, 512) * 2 - 1).cuda()
out = torch.empty(8, 3, 512, 512, device='cuda')
for name, ov, sp in (('sequential', False, False), ('shape branch aside', True, False), ('+ parse under Zencoder', True, True)) * 2:
    pipe.overlap, pipe.split_encode = ov, sp
    for _ in range(3):
        pipe.edit(img, out=out)
    ts = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            pipe.edit(img, out=out)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 100)
    print(f'{name:26s} {sorted(ts)[2]:.3f} ms')
