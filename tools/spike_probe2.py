"""Which part of a batch-1 render + read-back stalls for ~85 ms every second or third call? (host clock around each part)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctrlhair_amd import procedural as P
from ctrlhair_amd.sean.generator import SeanGenerator
S = 256
g = SeanGenerator(0, f16x3=0).load_state_dict(P.sean_state_dict(0, 64), max_batch=1, max_size=S)
dev = g.device
l = torch.from_numpy(P.blocky_labels(1, S)).to(dev); c = torch.from_numpy(P.style_codes(1)).to(dev); n = torch.from_numpy(P.noise_planes(1, S, 64)).to(dev)
pin = torch.empty((1, 3, S, S), dtype=torch.float32, pin_memory=True)
mode = sys.argv[1] if len(sys.argv) > 1 else 'sync+cpu'
rows = []
for i in range(40):
    t0 = time.time()
    out = g.generate(l, c, n)
    t1 = time.time()
    torch.cuda.synchronize()
    t2 = time.time()
    if mode == 'sync+cpu': h = out.cpu()
    elif mode == 'sync+pin': pin.copy_(out)
    elif mode == 'sleep': time.sleep(0.004)
    t3 = time.time()
    rows.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
print(mode, 'enqueue / sync / copy ms:', ' '.join(f'{a:.0f}/{b:.0f}/{c:.0f}' for a, b, c in rows))
