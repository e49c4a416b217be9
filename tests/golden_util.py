"""Helpers shared by the golden-vector tests: load a fixture made by tests/golden/make_golden.py,
regenerate its seeded inputs, and compare an image against the stored crops / subsample / sums."""
import os

import numpy as np

from ctrlhair_amd import procedural as P

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

SEAN_CASES = ['ngf16_S64_B3', 'ngf16_S64_ui', 'ngf16_S128_face', 'ngf64_S256_ui', 'ngf64_S256_face_B2',
              'ngf64_S512_ui', 'ngf64_S512_B2']


class Case:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, f'sean_gen_{name}.npz'))
        self.z = z
        self.name = name
        m = {k[5:]: z[k].item() for k in z.files if k.startswith('meta_')}
        self.ngf, self.S, self.B, self.ui = m['ngf'], m['S'], m['B'], bool(m['ui'])
        self.wseed = m['wseed']
        self.labels = z['labels']
        if m['codes_kind'] == 'median':
            med = np.load(os.path.join(os.path.dirname(P.CALIB_PATH), 'mean_style_code.npz'))['median']
            self.codes = np.repeat(med[None].astype(np.float32), self.B, 0)
        else:
            self.codes = P.style_codes(self.B, seed=m['cseed'])
        self.noise = P.noise_planes(self.B, self.S, self.ngf, seed=m['nseed'])

    def state_dict(self):
        return P.sean_state_dict(self.wseed, self.ngf)

    def max_abs_diff(self, img: np.ndarray) -> float:
        """max |img - golden| over everything the fixture stores (full image, or crops + stride-4 subsample),
        plus the per-channel-sum check scaled to a per-pixel figure."""
        z = self.z
        assert img.shape == (self.B, 3, self.S, self.S), img.shape
        return self.diff_samples(img, range(self.B))

    def diff_samples(self, img: np.ndarray, rows) -> float:
        """The same comparison for `img` [len(rows),3,S,S] holding the fixture's samples `rows` (a fixture's samples
        embedded in a larger batch)."""
        z = self.z
        rows = list(rows)
        zz = {k: (z[k][rows] if (k in ('image', 'sub4', 'sums') or (k.startswith('crop') and not k.endswith('_yx'))) else z[k])
              for k in z.files}
        return self._diff(zz, img)

    def _diff(self, z, img):
        if 'image' in z:
            return float(np.abs(img - z['image']).max())
        d = float(np.abs(img[:, :, ::4, ::4] - z['sub4']).max())
        i = 0
        while f'crop{i}' in z:
            y, x = z[f'crop{i}_yx']
            c = z[f'crop{i}']
            d = max(d, float(np.abs(img[:, :, y:y + c.shape[2], x:x + c.shape[3]] - c).max()))
            i += 1
        mean_err = np.abs(img.astype(np.float64).sum(axis=(2, 3)) - z['sums']).max() / (self.S * self.S)
        return max(d, float(mean_err))


ZENC_CASES = ['S64_B2', 'S256_face', 'S512_B1']


class ZencCase:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, f'sean_zenc_{name}.npz'))
        self.S, self.B = int(z['meta_S']), int(z['meta_B'])
        self.labels, self.codes = z['labels'], z['codes']
        self.img = P.synthetic_images(self.B, self.S, seed=int(z['meta_iseed']))

    def state_dict(self):
        return P.sean_state_dict(0, 16)
