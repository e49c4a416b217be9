"""Feasibility probe: two half-batch generator handles on two streams against one full-batch handle.
Run on the GPU box:  python tools/two_stream.py [--path f32]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctrlhair_amd import procedural as P                     # noqa: E402
from ctrlhair_amd.sean.generator import SeanGenerator        # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--path', default='f32')
    ap.add_argument('--steps', type=int, default=20)
    a = ap.parse_args()
    ngf, S, B = 64, 512, 16
    sd = P.sean_state_dict(0, ngf)
    dev = torch.device('cuda', 0)
    lab = torch.from_numpy(P.blocky_labels(B, S, grid=16)).to(dev)
    codes = torch.from_numpy(P.style_codes(B)).to(dev)
    noise = torch.from_numpy(P.noise_planes(B, S, ngf)).to(dev)
    f16x3 = int(a.path == 'f16x3')

    def timed(fn, n):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    one = SeanGenerator(0, f16x3=f16x3).load_state_dict(sd, max_batch=B, max_size=S)
    ms1 = timed(lambda: one.generate(lab, codes, noise), a.steps)
    one.handle.close()
    del one
    torch.cuda.empty_cache()
    h = [SeanGenerator(0, f16x3=f16x3).load_state_dict(sd, max_batch=B // 2, max_size=S) for _ in range(2)]
    st = [torch.cuda.Stream(dev) for _ in range(2)]
    parts = [(lab[i * 8:(i + 1) * 8].contiguous(), codes[i * 8:(i + 1) * 8].contiguous(), noise[i * 8:(i + 1) * 8].contiguous()) for i in range(2)]

    def two():
        for i in range(2):
            with torch.cuda.stream(st[i]):
                h[i].generate(*parts[i])

    def serial():
        for i in range(2):
            h[i].generate(*parts[i])

    ms_serial = timed(serial, a.steps)
    ms2 = timed(two, a.steps)
    print(f'{a.path}: one handle B=16: {ms1:.2f} ms ({B / ms1 * 1e3:.1f} images/s); two B=8 handles, one stream: {ms_serial:.2f} ms; '
          f'two streams: {ms2:.2f} ms ({B / ms2 * 1e3:.1f} images/s)')


if __name__ == '__main__':
    main()
