"""Generate committed golden vectors by running the *reference* modules (imported from
/root/reference, CPU, via refharness.py) on procedural weights + seeded synthetic inputs.

Build-container only.  Fixtures hold inputs/seeds and expected outputs -- never weights, never
reference source.  Large outputs are stored as crops + a stride-4 subsample + per-channel sums.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from ctrlhair_amd import procedural as P      # noqa: E402
import refharness as R                        # noqa: E402

CROPS = [(0, 0), (37, 101), (128, 64), (-64, -64)]   # top-left corners of 64x64 crops (negative = from end)


def face_like_labels(S, seed):
    """Concentric/elliptic blobs roughly like a parsing map (background, skin, hair cap, eyes, mouth...)."""
    rng = np.random.Generator(np.random.Philox(key=[seed, 77]))
    yy, xx = np.mgrid[0:S, 0:S].astype(np.float32) / S
    lab = np.zeros((S, S), np.uint8)
    def ell(cx, cy, rx, ry): return ((xx - cx) / rx) ** 2 + ((yy - cy) / ry) ** 2 < 1
    lab[ell(.5, .45, .42, .48)] = 13          # hair
    lab[ell(.5, .55, .27, .35)] = 1           # skin
    lab[ell(.5, .95, .22, .2)] = 17           # neck
    lab[ell(.5, 1.1, .5, .2)] = 18            # cloth
    lab[ell(.38, .48, .06, .03)] = 4; lab[ell(.62, .48, .06, .03)] = 5      # eyes
    lab[ell(.38, .42, .08, .015)] = 6; lab[ell(.62, .42, .08, .015)] = 7    # brows
    lab[ell(.5, .6, .05, .08)] = 2            # nose
    lab[ell(.5, .74, .1, .03)] = 11; lab[ell(.5, .77, .09, .025)] = 12      # lips
    lab[ell(.22, .55, .03, .08)] = 8; lab[ell(.78, .55, .03, .08)] = 9      # ears
    j = rng.integers(0, S - 8, size=(6, 2))
    for (a, b) in j:                           # a few tiny specks that vanish at low resolution
        lab[a:a + 3, b:b + 3] = 15
    return lab


def summarize(img):
    """img [B,3,S,S] -> dict of crops / subsample / channel sums."""
    B, _, S, _ = img.shape
    out = {'sums': img.astype(np.float64).sum(axis=(2, 3)), 'sub4': img[:, :, ::4, ::4].copy()}
    c = min(64, S)
    for i, (y, x) in enumerate(CROPS):
        y = y % S; x = x % S
        y = min(y, S - c); x = min(x, S - c)
        out[f'crop{i}'] = img[:, :, y:y + c, x:x + c].copy()
        out[f'crop{i}_yx'] = np.array([y, x])
    return out


def median_codes():
    z = np.load(os.path.join(ROOT, 'ctrlhair_amd', 'data', 'mean_style_code.npz'))
    return z['median'].astype(np.float32)


def case(name, ngf, S, B, ui, wseed=0, lseed=1234, cseed=2024, nseed=7, grid=16, labels='blocky', codes='tanh'):
    sd = P.sean_state_dict(wseed, ngf)
    if labels == 'blocky':
        lab = P.blocky_labels(B, S, seed=lseed, grid=grid)
    else:
        lab = np.stack([face_like_labels(S, lseed + b) for b in range(B)])
    cd = P.style_codes(B, seed=cseed)
    if codes == 'median':
        cd = np.repeat(median_codes()[None], B, 0)
    nz = P.noise_planes(B, S, ngf, seed=nseed)
    img = R.run_generator(sd, lab, cd, nz, ngf, ui_mode=ui)
    meta = dict(ngf=ngf, S=S, B=B, ui=int(ui), wseed=wseed, cseed=cseed, nseed=nseed,
                codes_kind=codes)
    out = {'labels': lab, **{'meta_' + k: np.array(v) for k, v in meta.items()}}
    if S <= 64:
        out['image'] = img
    else:
        out.update(summarize(img))
    path = os.path.join(HERE, f'sean_gen_{name}.npz')
    np.savez_compressed(path, **out)
    print(name, 'std', img.std(), 'bytes', os.path.getsize(path))


def zenc_case(name, S, B, labels='blocky', lseed=1234, iseed=31, grid=16):
    sd = P.sean_state_dict(0, 16)     # Zencoder weights do not depend on ngf
    lab = P.blocky_labels(B, S, seed=lseed, grid=grid) if labels == 'blocky' else \
        np.stack([face_like_labels(S, lseed + b) for b in range(B)])
    img = P.synthetic_images(B, S, seed=iseed)
    codes = R.run_zencoder(sd, img, lab)
    path = os.path.join(HERE, f'sean_zenc_{name}.npz')
    np.savez_compressed(path, labels=lab, codes=codes, meta_S=np.array(S), meta_B=np.array(B), meta_iseed=np.array(iseed))
    print('zenc', name, 'absent rows', int((np.abs(codes).sum(-1) == 0).sum()), 'bytes', os.path.getsize(path))


def main():
    if 'zenc' in sys.argv[1:] or len(sys.argv) == 1:
        zenc_case('S64_B2', 64, 2, grid=8)
        zenc_case('S256_face', 256, 1, labels='face', lseed=41, iseed=42)
        zenc_case('S512_B1', 512, 1, lseed=51, iseed=52)
    if len(sys.argv) > 1 and 'gen' not in sys.argv[1:]:
        return
    case('ngf16_S64_B3', 16, 64, 3, False, grid=8)
    case('ngf16_S64_ui', 16, 64, 1, True, grid=8, lseed=5, cseed=6, nseed=8)
    case('ngf16_S128_face', 16, 128, 2, False, labels='face', codes='median', lseed=11, nseed=12)
    case('ngf64_S256_ui', 64, 256, 1, True)
    case('ngf64_S256_face_B2', 64, 256, 2, False, labels='face', codes='median', lseed=21, nseed=22)
    case('ngf64_S512_ui', 64, 512, 1, True, lseed=31, cseed=32, nseed=33)


if __name__ == '__main__':
    main()
