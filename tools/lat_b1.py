"""Single-image render latency (the call behind Backend.output), eager launches vs hipGraph replay:
    python tools/lat_b1.py [S] [key=value ...]      (ch_set_option pairs, e.g. sean.wino=1)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctrlhair_amd import procedural as P
from ctrlhair_amd.sean.generator import SeanGenerator


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
OPTS = {kv.split('=')[0]: int(kv.split('=')[1]) for kv in sys.argv[2:]}
LEGS = [l for l in ((1, 'f16x3'), (0, 'f32')) if os.environ.get('LEG', l[1]) == l[1]]
for mode, name in LEGS:
    g = SeanGenerator(0, f16x3=mode, options=OPTS).load_state_dict(P.sean_state_dict(0, 64), max_batch=1, max_size=S)
    dev = g.device
    l = torch.from_numpy(P.blocky_labels(1, S)).to(dev)
    c = torch.from_numpy(P.style_codes(1)).to(dev)
    n = torch.from_numpy(P.noise_planes(1, S, 64)).to(dev)
    eager = timeit(lambda: g.generate(l, c, n))
    graph, out = g.capture(l, c, n)
    replay = timeit(graph.replay)
    print(f'{name} S={S} batch 1: eager {eager:.3f} ms, hipGraph replay {replay:.3f} ms per image')
    del graph, g
