"""Host-side (numpy) helpers of the editing API: the handful of cv2 / dataset-table calls the reference's
`ui/backend.py` and `hair_editor.py` make around the networks.  cv2 is not installed in this image, so each has a
numpy implementation with cv2's conventions; when cv2 *is* importable it is used instead.  None of this is on the GPU
hot path (SURVEY.md 7 "host-side dependencies of the drop-in surface").
"""
import os
import pickle
from bisect import bisect_left, bisect_right

import numpy as np

HAIR_IDX = 13
PARSING_LABEL_LIST = ['background', 'skin_other', 'nose', 'eye_g', 'l_eye', 'r_eye', 'l_brow', 'r_brow',
                      'l_ear', 'r_ear', 'mouth', 'u_lip', 'l_lip', 'hair', 'hat',
                      'ear_r', 'neck_l', 'neck', 'cloth']          # global_value_utils.py:49-52
TEMP_FOLDER = 'temp_folder'


def _cv2():
    try:
        import cv2
        return cv2 if hasattr(cv2, 'cvtColor') else None      # (an import-only stub module is not cv2)
    except Exception:
        return None


def resize_bilinear(img: np.ndarray, size) -> np.ndarray:
    """cv2.resize(img, (w, h)) (INTER_LINEAR: half-pixel centres, edge clamp, no anti-aliasing).  uint8 in/out
    (round-half-up like cv2's fixed point to within 1 LSB)."""
    cv2 = _cv2()
    if cv2 is not None:
        return cv2.resize(img, tuple(size))
    w, h = size
    H, W = img.shape[:2]
    if (H, W) == (h, w):
        return img.copy()
    a = img.astype(np.float32)

    def axis(n_out, n_in):
        t = (np.arange(n_out, dtype=np.float32) + 0.5) * (n_in / n_out) - 0.5
        i0 = np.floor(t).astype(np.int64)
        f = t - i0
        return np.clip(i0, 0, n_in - 1), np.clip(i0 + 1, 0, n_in - 1), f

    y0, y1, fy = axis(h, H)
    x0, x1, fx = axis(w, W)
    fy = fy.reshape(-1, 1, *([1] * (a.ndim - 2)))
    fx = fx.reshape(1, -1, *([1] * (a.ndim - 2)))
    top = a[y0][:, x0] * (1 - fx) + a[y0][:, x1] * fx
    bot = a[y1][:, x0] * (1 - fx) + a[y1][:, x1] * fx
    out = top * (1 - fy) + bot * fy
    return np.clip(np.floor(out + 0.5), 0, 255).astype(img.dtype) if img.dtype == np.uint8 else out.astype(img.dtype)


def resize_nearest(img: np.ndarray, size) -> np.ndarray:
    """cv2.resize(..., interpolation=cv2.INTER_NEAREST): src = floor(dst * in/out)."""
    w, h = size
    H, W = img.shape[:2]
    ys = np.minimum((np.arange(h) * (H / h)).astype(np.int64), H - 1)
    xs = np.minimum((np.arange(w) * (W / w)).astype(np.int64), W - 1)
    return img[ys][:, xs]


def rgb_to_hsv_u8(rgb: np.ndarray) -> np.ndarray:
    """cv2.cvtColor(uint8 RGB, COLOR_RGB2HSV): H in [0,180), S,V in [0,255]."""
    cv2 = _cv2()
    if cv2 is not None:
        return cv2.cvtColor(rgb.astype('uint8'), cv2.COLOR_RGB2HSV)
    a = rgb.astype(np.float32)
    r, g, b = a[..., 0], a[..., 1], a[..., 2]
    v = a.max(-1)
    mn = a.min(-1)
    d = v - mn
    s = np.where(v > 0, d / np.maximum(v, 1e-12) * 255.0, 0.0)
    dd = np.maximum(d, 1e-12)
    h = np.where(v == r, (g - b) / dd, np.where(v == g, 2.0 + (b - r) / dd, 4.0 + (r - g) / dd)) * 60.0
    h = np.where(d == 0, 0.0, h)
    h = np.where(h < 0, h + 360.0, h) / 2.0
    out = np.stack([np.floor(h + 0.5) % 180, np.floor(s + 0.5), v], -1)
    return np.clip(out, 0, 255).astype(np.uint8)


def hsv_to_rgb_u8(hsv: np.ndarray) -> np.ndarray:
    """cv2.cvtColor(uint8 HSV, COLOR_HSV2RGB)."""
    cv2 = _cv2()
    if cv2 is not None:
        return cv2.cvtColor(hsv.astype('uint8'), cv2.COLOR_HSV2RGB)
    a = hsv.astype(np.float32)
    h, s, v = a[..., 0] * 2.0, a[..., 1] / 255.0, a[..., 2]
    c = v * s
    hp = (h / 60.0) % 6.0
    x = c * (1 - np.abs(hp % 2 - 1))
    z = np.zeros_like(c)
    sel = np.floor(hp).astype(np.int64)
    r = np.choose(sel, [c, x, z, z, x, c])
    g = np.choose(sel, [x, c, c, x, z, z])
    b = np.choose(sel, [z, z, x, c, c, x])
    m = v - c
    return np.clip(np.floor(np.stack([r + m, g + m, b + m], -1) + 0.5), 0, 255).astype(np.uint8)


_MASK_COLORS = np.array([[0, 128, 64], [204, 0, 0], [76, 153, 0], [204, 204, 0], [51, 51, 255], [204, 0, 204], [0, 255, 255],
                         [51, 255, 255], [102, 51, 0], [255, 0, 0], [102, 204, 0], [255, 255, 0], [0, 0, 153], [0, 0, 204],
                         [255, 51, 153], [0, 204, 204], [0, 51, 0], [255, 153, 51], [0, 204, 0]], np.uint8)


def mask_to_rgb(pred: np.ndarray, draw_type: int = 2) -> np.ndarray:
    """util/mask_color_util.py:15-64."""
    if pred.ndim == 3 and pred.shape[0] == 1:
        pred = pred[0]
    color = _MASK_COLORS.copy()
    for cc in range(len(color)):
        if draw_type == 2 and cc != HAIR_IDX:
            color[cc] = [255, 255, 255]
        elif draw_type == 1 and cc != HAIR_IDX and cc != 0:
            color[cc] = [237, 28, 36]
    lut = np.full((256, 3), 0, np.uint8)
    lut[:19] = color
    lut[255] = 255
    return lut[pred.astype(np.uint8)]


class DistTranslation:
    """util/color_from_hsv_to_gaussian.py:16-33: HSV value <-> N(0,1) quantile through the dataset's sorted HSV table
    (dataset_info_ctrlhair/hsv_stat_dict_ordered.pkl, not shipped).  `table` [N,3] (each column sorted) may be injected;
    otherwise the pickle is loaded if present, else a documented synthetic table (uniform quantiles of H in [0,179],
    S,V in [0,255]) is used so the API stays functional."""

    def __init__(self, table: np.ndarray = None, root: str = 'dataset_info_ctrlhair'):
        if table is None:
            path = os.path.join(root, 'hsv_stat_dict_ordered.pkl')
            if os.path.exists(path):
                with open(path, 'rb') as f:
                    table = pickle.load(f)
            else:
                q = (np.arange(4096) + 0.5) / 4096
                table = np.stack([q * 179.0, q * 255.0, q * 255.0], 1)
        self.cols_hsv = np.asarray(table)

    def gaussian_to_val(self, dim, val):
        import scipy.stats as st
        n = self.cols_hsv.shape[0]
        return self.cols_hsv[min(int(st.norm.cdf(val) * n), n - 1)][dim]

    def val_to_gaussian(self, dim, val):
        import scipy.stats as st
        col = self.cols_hsv[:, dim]
        return st.norm.ppf((bisect_left(col, val) + bisect_right(col, val)) / 2 / self.cols_hsv.shape[0])


def seeded_directions(n: int, dim: int, seed: int) -> np.ndarray:
    """Stand-in for the trained direction pickles (model_trained/*/{texture,shape}_dir_used, hair_editor.py:82-119):
    n orthonormal vectors in R^dim from a seeded QR (SURVEY.md 8d Config 3)."""
    rng = np.random.Generator(np.random.Philox(key=[seed, dim]))
    q, _ = np.linalg.qr(rng.standard_normal((dim, n)))
    return np.ascontiguousarray(q.T).astype(np.float32)
