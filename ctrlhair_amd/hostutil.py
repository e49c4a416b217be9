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


_CV2 = [False, None]          # [probed, module]: a failed import costs ~6 ms of path search, and output() asks on every call


def _cv2():
    if not _CV2[0]:
        try:
            import cv2
            _CV2[1] = cv2 if hasattr(cv2, 'cvtColor') else None      # (an import-only stub module is not cv2)
        except Exception:
            _CV2[1] = None
        _CV2[0] = True
    return _CV2[1]


def cpu_quota():
    """CPUs this process may actually use: the cgroup CFS quota (cpu.max / cfs_quota_us) if one is set, else os.cpu_count()."""
    n = os.cpu_count() or 1
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:                       # cgroup v2
            q, per = f.read().split()
        if q != 'max':
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        try:
            with open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us') as f:      # cgroup v1
                q = int(f.read())
            with open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as f:
                per = int(f.read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return n


def cap_threads_to_cpu_quota():
    """torch sizes its CPU thread pool by the visible cores (256 on the MI355X box) even when the container's CFS quota is 16 CPUs: the pool's
    spin-waits around the tiny host-side tensor ops of an edit then exhaust the quota and the whole process is descheduled for the rest of
    the 100 ms period -- every second or third Backend.output() took 90 ms instead of 5 (tools/spike_probe.py).  Never raises the count."""
    import torch
    q = cpu_quota()
    if torch.get_num_threads() > q:
        torch.set_num_threads(q)
    return q


_PINNED = {}
_PINNED_LOCK = None


def _pinned_lock():
    global _PINNED_LOCK
    if _PINNED_LOCK is None:
        import threading
        _PINNED_LOCK = threading.Lock()
    return _PINNED_LOCK


def to_host(t):
    """Device tensor -> numpy array through a cached PINNED staging buffer per (shape, dtype).  A plain `.cpu()` allocates fresh pageable
    memory for every call, which the runtime has to lock page by page before the DMA: on the ROCm 7 stack of the target box that path
    stalls for ~85 ms on every second or third 0.8 MB image (tools/spike_probe.py), i.e. more than the whole edit.  The result is a
    private copy (the staging buffer is reused by the next call)."""
    import torch
    if not isinstance(t, torch.Tensor):
        return np.asarray(t)
    t = t.detach()
    if not t.is_cuda:
        return t.numpy()
    key = (tuple(t.shape), t.dtype)
    with _pinned_lock():               # (two threads with tensors of one shape would otherwise share the staging buffer between copy_ and copy)
        buf = _PINNED.get(key)
        if buf is None:
            if len(_PINNED) > 64:
                _PINNED.clear()
            buf = _PINNED[key] = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        buf.copy_(t.contiguous())          # synchronous: the data has landed when copy_ returns
        return buf.numpy().copy()


def _linear_taps(n_out: int, n_in: int):
    """Source taps and 11-bit weights of one axis of cv2.resize(INTER_LINEAR): half-pixel centres, taps clamped to the image,
    weights rounded to 1/2048 (the two weights always sum to 2048)."""
    f = (np.arange(n_out) + 0.5) * (n_in / n_out) - 0.5
    i0 = np.floor(f).astype(np.int64)
    f = (f - i0).astype(np.float32)
    f = np.where((i0 < 0) | (i0 >= n_in - 1), np.float32(0), f)
    i0 = np.clip(i0, 0, n_in - 1)
    w1 = np.rint(f * 2048.0).astype(np.int64)
    w0 = np.rint((1.0 - f) * 2048.0).astype(np.int64)
    return i0, np.minimum(i0 + 1, n_in - 1), w0, w1


def resize_bilinear(img: np.ndarray, size) -> np.ndarray:
    """cv2.resize(img, (w, h)) for uint8 images, in OpenCV's own fixed-point arithmetic (11-bit tap weights, int32 rows,
    vertical pass with the >> 4 / >> 16 / + 2 / >> 2 rounding): bit-exact against tests/golden/host_vectors.npz.  Other
    dtypes: plain float interpolation with the same taps."""
    cv2 = _cv2()
    if cv2 is not None:
        return cv2.resize(img, tuple(size))
    w, h = size
    H, W = img.shape[:2]
    if (H, W) == (h, w):
        return img.copy()
    x0, x1, a0, a1 = _linear_taps(w, W)
    y0, y1, b0, b1 = _linear_taps(h, H)
    tail = (1,) * (img.ndim - 2)
    if img.dtype == np.uint8:
        a = img.astype(np.int64)
        rows = a[:, x0] * a0.reshape((1, -1) + tail) + a[:, x1] * a1.reshape((1, -1) + tail)
        top, bot = rows[y0] >> 4, rows[y1] >> 4
        out = (((b0.reshape((-1, 1) + tail) * top) >> 16) + ((b1.reshape((-1, 1) + tail) * bot) >> 16) + 2) >> 2
        return np.clip(out, 0, 255).astype(np.uint8)
    a = img.astype(np.float32)
    fx = (a1 / 2048.0).astype(np.float32).reshape((1, -1) + tail)
    fy = (b1 / 2048.0).astype(np.float32).reshape((-1, 1) + tail)
    rows = a[:, x0] * (1 - fx) + a[:, x1] * fx
    return (rows[y0] * (1 - fy) + rows[y1] * fy).astype(img.dtype)


def resize_nearest(img: np.ndarray, size) -> np.ndarray:
    """cv2.resize(..., interpolation=cv2.INTER_NEAREST): src = floor(dst * in/out)."""
    w, h = size
    H, W = img.shape[:2]
    ys = np.minimum((np.arange(h) * (H / h)).astype(np.int64), H - 1)
    xs = np.minimum((np.arange(w) * (W / w)).astype(np.int64), W - 1)
    return img[ys][:, xs]


_HSV_SHIFT = 12
_SDIV = np.array([0] + [int(np.rint((255 << _HSV_SHIFT) / v)) for v in range(1, 256)], np.int64)
_HDIV = np.array([0] + [int(np.rint((180 << _HSV_SHIFT) / (6.0 * d))) for d in range(1, 256)], np.int64)


def rgb_to_hsv_u8(rgb: np.ndarray) -> np.ndarray:
    """cv2.cvtColor(uint8 RGB, COLOR_RGB2HSV): H in [0,180), S,V in [0,255] -- OpenCV's 8-bit integer path (12-bit fixed
    point with division tables), bit-exact against tests/golden/host_vectors.npz."""
    cv2 = _cv2()
    if cv2 is not None:
        return cv2.cvtColor(rgb.astype('uint8'), cv2.COLOR_RGB2HSV)
    a = rgb.astype(np.int64)
    r, g, b = a[..., 0], a[..., 1], a[..., 2]
    v = a.max(-1)
    d = v - a.min(-1)
    half = 1 << (_HSV_SHIFT - 1)
    h = np.where(v == r, g - b, np.where(v == g, b - r + 2 * d, r - g + 4 * d))
    h = (h * _HDIV[d] + half) >> _HSV_SHIFT
    h = np.where(h < 0, h + 180, h)
    sat = (d * _SDIV[v] + half) >> _HSV_SHIFT
    return np.stack([h, sat, v], -1).astype(np.uint8)


def hsv_to_rgb_u8(hsv: np.ndarray) -> np.ndarray:
    """cv2.cvtColor(uint8 HSV, COLOR_HSV2RGB): float32 sector formula on (h * 6/180, s/255, v/255), times 255, rounded half to
    even and saturated (bit-exact against tests/golden/host_vectors.npz)."""
    cv2 = _cv2()
    if cv2 is not None:
        return cv2.cvtColor(hsv.astype('uint8'), cv2.COLOR_HSV2RGB)
    a = hsv.astype(np.float32)
    h, s, v = a[..., 0] * np.float32(6.0 / 180.0), a[..., 1] * np.float32(1.0 / 255.0), a[..., 2] * np.float32(1.0 / 255.0)
    sec = np.floor(h).astype(np.int64)
    f = h - sec
    sec = sec % 6
    p0, p1, p2, p3 = v, v * (1 - s), v * (1 - s * f), v * (1 - s * (1 - f))
    r = np.choose(sec, [p0, p2, p1, p1, p3, p0])
    g = np.choose(sec, [p3, p0, p0, p2, p1, p1])
    b = np.choose(sec, [p1, p1, p3, p0, p0, p2])
    return np.clip(np.rint(np.stack([r, g, b], -1) * np.float32(255.0)), 0, 255).astype(np.uint8)


_MASK_COLORS = np.array([[0, 128, 64], [204, 0, 0], [76, 153, 0], [204, 204, 0], [51, 51, 255], [204, 0, 204], [0, 255, 255],
                         [51, 255, 255], [102, 51, 0], [255, 0, 0], [102, 204, 0], [255, 255, 0], [0, 0, 153], [0, 0, 204],
                         [255, 51, 153], [0, 204, 204], [0, 51, 0], [255, 153, 51], [0, 204, 0]], np.uint8)


def mask_to_rgb(pred: np.ndarray, draw_type: int = 2) -> np.ndarray:
    """util/mask_color_util.py:15-64."""
    if pred.ndim == 3 and pred.shape[0] == 1:
        pred = pred[0]
    lut = _MASK_LUT.get(draw_type)
    if lut is None:
        color = _MASK_COLORS.copy()
        for cc in range(len(color)):
            if draw_type == 2 and cc != HAIR_IDX:
                color[cc] = [255, 255, 255]
            elif draw_type == 1 and cc != HAIR_IDX and cc != 0:
                color[cc] = [237, 28, 36]
        lut = np.zeros((256, 3), np.uint8)
        lut[:19] = color
        lut[255] = 255
        _MASK_LUT[draw_type] = lut
    return np.take(lut, pred.astype(np.uint8), axis=0)       # (three times as fast as lut[pred] on a 256 x 256 map)


_MASK_LUT = {}


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
