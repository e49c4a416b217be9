#!/usr/bin/env python3
"""Golden vectors for the Poisson blending step: outputs of the REFERENCE function (imported from
/root/reference/poisson_blending.py, numpy + scipy only) on seeded synthetic inputs.  Run in the build container:

    python tests/golden/make_poisson_golden.py        -> tests/golden/poisson_golden.npz

Cases cover: a blob mask away from the border, a mask touching the image border (the reference's border rows), an
all-zero mask (nothing to solve: output = target), an all-one mask, non-square images, with_gamma on/off."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, '/root/reference')
from poisson_blending import poisson_blending       # noqa: E402  (reference, read-only)


def smooth_image(rng, H, W):
    """Low-frequency colour field + mild noise, uint8 [H,W,3] (portrait-like statistics, no zeros for the gamma power)."""
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float64)
    img = np.zeros((H, W, 3))
    for c in range(3):
        a, b, p = rng.uniform(0.5, 2.5, 3)
        img[:, :, c] = 128 + 70 * np.sin(a * ys / H * np.pi + p) * np.cos(b * xs / W * np.pi) + rng.normal(0, 6, (H, W))
    return np.clip(img, 4, 251).astype(np.uint8)


def blob_mask(rng, H, W, cy, cx, ry, rx, invert=False):
    ys, xs = np.mgrid[0:H, 0:W]
    m = (((ys - cy) / ry) ** 2 + ((xs - cx) / rx) ** 2 <= 1.0).astype(np.uint8)
    return 1 - m if invert else m


def main():
    rng = np.random.default_rng(20260928)
    cases = {}

    def add(name, H, W, mask, with_gamma=True):
        src, tgt = smooth_image(rng, H, W), smooth_image(rng, H, W)
        out = poisson_blending(src.copy(), tgt.copy(), mask.copy()[..., None], with_gamma=with_gamma)
        cases[name] = dict(src=src, tgt=tgt, mask=mask, out=out, gamma=np.array(with_gamma))

    add('blob_inside_48x40', 48, 40, blob_mask(rng, 48, 40, 22, 19, 12, 10))
    add('hair_like_64x64', 64, 64, blob_mask(rng, 64, 64, 20, 32, 18, 22, invert=True))      # solve everywhere but the blob
    add('touches_border_40x56', 40, 56, blob_mask(rng, 40, 56, 0, 28, 22, 20, invert=True))
    add('all_zero_32x32', 32, 32, np.zeros((32, 32), np.uint8))
    add('all_one_32x24', 32, 24, np.ones((32, 24), np.uint8))
    add('no_gamma_40x40', 40, 40, blob_mask(rng, 40, 40, 18, 20, 9, 13, invert=True), with_gamma=False)
    flat = {f'{k}/{f}': v for k, d in cases.items() for f, v in d.items()}
    np.savez_compressed(os.path.join(HERE, 'poisson_golden.npz'), **flat)
    print('wrote', len(cases), 'cases')


if __name__ == '__main__':
    main()
