"""Vectors pinning the host-side colour / resize / morphology helpers (ctrlhair_amd/hostutil.py, oracle/poisson_oracle.py)
to OpenCV's 8-bit algorithms.  cv2 is not installed in this image (and has no network to come from), so the expected values
are DERIVED here, independently of hostutil, by restating the integer / fixed-point algorithms OpenCV documents for these
calls -- the ones the reference makes at ui/backend.py:73,100,113 and hair_editor.py:121-128,297-305:

  cv2.cvtColor(uint8, COLOR_RGB2HSV)      imgproc colour conversions, 8-bit path: V = max, S = 255 * diff / V, H = 30 * sector
        arithmetic / diff (H in [0,180)), evaluated in 12-bit fixed point with the division tables
        sdiv[v] = round(255 * 4096 / v), hdiv[d] = round(180 * 4096 / (6 d)), results rounded by adding half before the shift.
  cv2.cvtColor(uint8, COLOR_HSV2RGB)      8-bit path: (h, s/255, v/255) through the float sector formula
        tab = {v, v(1-s), v(1-s f), v(1-s(1-f))}, then * 255 and round-half-even (cvRound), saturated.
  cv2.resize(uint8, INTER_LINEAR)         half-pixel centres, clamped taps, 11-bit coefficients (x 2048, rounded, second tap =
        2048 - first), horizontal pass in int32, vertical pass ((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16), + 2, >> 2.
  cv2.getStructuringElement(MORPH_ELLIPSE, (k, k))   row i: dy = i - r, dx = round(c * sqrt((r^2 - dy^2) / r^2)), ones on
        [c - dx, c + dx]  (r = c = k // 2).

    python tests/golden/make_host_vectors.py     -> tests/golden/host_vectors.npz
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def cv_round(x):
    return np.rint(x)          # round half to even, like cvRound / lrint


def rgb2hsv_u8(rgb):
    rgb = rgb.astype(np.int64)
    r, g, b = rgb[..., 0], rgb[..., 1], rgb[..., 2]
    v = np.maximum(np.maximum(r, g), b)
    vmin = np.minimum(np.minimum(r, g), b)
    diff = v - vmin
    shift = 12
    sdiv = np.zeros(256, np.int64)
    hdiv = np.zeros(256, np.int64)
    for i in range(1, 256):
        sdiv[i] = int(cv_round((255 << shift) / (1.0 * i)))
        hdiv[i] = int(cv_round((180 << shift) / (6.0 * i)))
    vr = v == r
    vg = v == g
    h = np.where(vr, g - b, np.where(vg, b - r + 2 * diff, r - g + 4 * diff))
    s = (diff * sdiv[v] + (1 << (shift - 1))) >> shift
    h = (h * hdiv[diff] + (1 << (shift - 1))) >> shift
    h = np.where(h < 0, h + 180, h)
    return np.stack([h, s, v], -1).astype(np.uint8)


def hsv2rgb_u8(hsv):
    h = hsv[..., 0].astype(np.float32) * np.float32(6.0 / 180.0)
    s = hsv[..., 1].astype(np.float32) * np.float32(1.0 / 255.0)
    v = hsv[..., 2].astype(np.float32) * np.float32(1.0 / 255.0)
    sector = np.floor(h).astype(np.int64)
    f = h - sector
    sector = sector % 6
    tab = np.stack([v, v * (1 - s), v * (1 - s * f), v * (1 - s * (1 - f))], -1)
    idx = np.array([[1, 3, 0], [1, 0, 2], [3, 0, 1], [0, 2, 1], [0, 1, 3], [2, 1, 0]])    # (b, g, r) per sector
    sel = idx[sector]
    b = np.take_along_axis(tab, sel[..., 0:1], -1)[..., 0]
    g = np.take_along_axis(tab, sel[..., 1:2], -1)[..., 0]
    r = np.take_along_axis(tab, sel[..., 2:3], -1)[..., 0]
    s0 = s == 0
    r, g, b = np.where(s0, v, r), np.where(s0, v, g), np.where(s0, v, b)
    return np.clip(cv_round(np.stack([r, g, b], -1) * np.float32(255.0)), 0, 255).astype(np.uint8)


def resize_linear_u8(img, size):
    w, h = size
    H, W = img.shape[:2]
    a = img.astype(np.int64)

    def taps(n_out, n_in):
        scale = n_in / n_out
        fx = (np.arange(n_out) + 0.5) * scale - 0.5
        sx = np.floor(fx).astype(np.int64)
        fx = (fx - sx).astype(np.float32)
        lo = sx < 0
        sx = np.where(lo, 0, sx)
        fx = np.where(lo, np.float32(0), fx)
        hi = sx >= n_in - 1
        sx = np.where(hi, n_in - 1, sx)
        fx = np.where(hi, np.float32(0), fx)
        c0 = np.clip(cv_round((1.0 - fx) * 2048.0), -32768, 32767).astype(np.int64)
        c1 = np.clip(cv_round(fx * 2048.0), -32768, 32767).astype(np.int64)
        return sx, np.minimum(sx + 1, n_in - 1), c0, c1

    x0, x1, a0, a1 = taps(w, W)
    y0, y1, b0, b1 = taps(h, H)
    sh = (1, -1) + (1,) * (a.ndim - 2)
    rows = a[:, x0] * a0.reshape(sh) + a[:, x1] * a1.reshape(sh)            # horizontal pass, x 2048
    r0, r1 = rows[y0], rows[y1]
    sv = (-1, 1) + (1,) * (a.ndim - 2)
    out = (((b0.reshape(sv) * (r0 >> 4)) >> 16) + ((b1.reshape(sv) * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def ellipse(k):
    r = c = k // 2
    out = np.zeros((k, k), np.uint8)
    inv_r2 = 1.0 / (r * r) if r else 0.0
    for i in range(k):
        dy = i - r
        if abs(dy) <= r:
            dx = int(cv_round(c * np.sqrt((r * r - dy * dy) * inv_r2)))
            out[i, max(c - dx, 0):min(c + dx + 1, k)] = 1
    return out


def main():
    rng = np.random.default_rng(20260928)
    rgb = np.concatenate([rng.integers(0, 256, size=(4000, 3)),
                          np.array([[0, 0, 0], [255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255], [128, 128, 128], [10, 10, 9],
                                    [200, 100, 50], [50, 100, 200], [100, 200, 50], [1, 0, 0], [254, 255, 255], [17, 16, 17]])]).astype(np.uint8)
    hsv_in = np.stack([rng.integers(0, 180, 4000), rng.integers(0, 256, 4000), rng.integers(0, 256, 4000)], 1).astype(np.uint8)
    img = rng.integers(0, 256, size=(37, 53, 3)).astype(np.uint8)
    smooth = (np.add.outer(np.arange(64) * 3, np.arange(64) * 2) % 256).astype(np.uint8)
    out = {'rgb': rgb, 'rgb_to_hsv': rgb2hsv_u8(rgb), 'hsv': hsv_in, 'hsv_to_rgb': hsv2rgb_u8(hsv_in),
           'img': img, 'img_to_64x48': resize_linear_u8(img, (64, 48)), 'img_to_20x15': resize_linear_u8(img, (20, 15)),
           'smooth': smooth, 'smooth_to_256': resize_linear_u8(smooth, (256, 256)), 'smooth_to_32': resize_linear_u8(smooth, (32, 32)),
           'ellipse5': ellipse(5), 'ellipse13': ellipse(13), 'ellipse19': ellipse(19)}
    np.savez_compressed(os.path.join(HERE, 'host_vectors.npz'), **out)
    print({k: v.shape for k, v in out.items()})
    print(ellipse(13))


if __name__ == '__main__':
    main()
