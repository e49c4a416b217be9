"""Produce ctrlhair_amd/data/sean_calib.npz: BN running statistics + conv_img gain that make the
procedural (random) SEAN generator non-degenerate.

Dev-time generator (run once in the build container, result committed).  Uses the oracle's
calibration mode: a forward pass that records per-ACE batch statistics of (x + noise) and
normalises with them, so every later ACE sees O(1) inputs.

    python tests/golden/make_calibration.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from ctrlhair_amd import procedural as P          # noqa: E402
from oracle import sean_oracle as O               # noqa: E402

CONFIGS = [  # (seed, ngf, S, B)
    (0, 64, 256, 2),
    (0, 16, 64, 4),
]


def main():
    out = {}
    if os.path.exists(P.CALIB_PATH):
        z = np.load(P.CALIB_PATH)
        out.update({k: z[k] for k in z.files})
    for seed, ngf, S, B in CONFIGS:
        sd = O.to_torch(P.sean_state_dict(seed, ngf, calibrated=False, with_zencoder=False))
        labels = P.blocky_labels(B, S, seed=99)
        codes = P.style_codes(B, seed=98)
        noise = P.noise_planes(B, S, ngf, seed=97)
        stats = {}
        img = O.generator_forward(sd, labels, codes, noise, ngf, stats_out=stats)
        pre = f'ngf{ngf}_seed{seed}/'
        for k in [k for k in out if k.startswith(pre)]:
            del out[k]
        for k, v in stats.items():
            out[pre + k] = np.asarray(v, np.float32)
        print(f'ngf={ngf} S={S}: calib-mode out std {float(img.std()):.3f} |y|>0.99: '
              f'{float((img.abs() > 0.99).float().mean()):.4f} gain {stats["conv_img.gain"]:.4f}')
    np.savez_compressed(P.CALIB_PATH, **out)
    # verify eval-mode forward with the stored table
    for seed, ngf, S, B in CONFIGS:
        sd = O.to_torch(P.sean_state_dict(seed, ngf, with_zencoder=False))
        img = O.generator_forward(sd, P.blocky_labels(1, S), P.style_codes(1), P.noise_planes(1, S, ngf), ngf)
        print(f'ngf={ngf} S={S} eval: std {float(img.std()):.3f} mean|y| {float(img.abs().mean()):.3f} '
              f'sat {float((img.abs() > 0.99).float().mean()):.4f}')


if __name__ == '__main__':
    main()
