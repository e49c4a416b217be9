"""The host-side colour / resize / morphology helpers against tests/golden/host_vectors.npz: vectors derived (independently of
hostutil, tests/golden/make_host_vectors.py) from OpenCV's 8-bit algorithms for the cv2 calls the reference makes at
ui/backend.py:73,100,113 and hair_editor.py:121-128,297-305.  cv2 itself is not installed here."""
import os

import numpy as np
import torch

from ctrlhair_amd import hostutil as U
from tests.golden_util import GOLDEN

Z = np.load(os.path.join(GOLDEN, 'host_vectors.npz'))


def test_rgb_hsv_conversions_bit_exact():
    assert np.array_equal(U.rgb_to_hsv_u8(Z['rgb'][None])[0], Z['rgb_to_hsv'])
    assert np.array_equal(U.hsv_to_rgb_u8(Z['hsv'][None])[0], Z['hsv_to_rgb'])
    # primaries: the documented values of the 8-bit conversion
    assert U.rgb_to_hsv_u8(np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [128, 128, 128]]], np.uint8))[0].tolist() == \
        [[0, 255, 255], [60, 255, 255], [120, 255, 255], [0, 0, 128]]


def test_device_side_conversions_equal_host():
    """ctrlhair_amd.pipeline's torch versions (used on the GPU inside EditPipeline) are the same arithmetic."""
    from ctrlhair_amd import pipeline as PL
    assert np.array_equal(PL.rgb_to_hsv_u8(torch.from_numpy(Z['rgb'].astype(np.float32))).numpy(), Z['rgb_to_hsv'].astype(np.float32))
    assert np.array_equal(PL.hsv_to_rgb_u8(torch.from_numpy(Z['hsv'].astype(np.float32))).numpy(), Z['hsv_to_rgb'].astype(np.float32))


def test_bilinear_resize_bit_exact():
    assert np.array_equal(U.resize_bilinear(Z['img'], (64, 48)), Z['img_to_64x48'])
    assert np.array_equal(U.resize_bilinear(Z['img'], (20, 15)), Z['img_to_20x15'])
    assert np.array_equal(U.resize_bilinear(Z['smooth'], (256, 256)), Z['smooth_to_256'])
    assert np.array_equal(U.resize_bilinear(Z['smooth'], (32, 32)), Z['smooth_to_32'])
    f = U.resize_bilinear(Z['smooth'].astype(np.float32), (32, 32))          # float images: same taps, no rounding
    assert np.abs(f - Z['smooth_to_32']).max() <= 0.51


def test_structuring_elements():
    from oracle import poisson_oracle as PO
    for k in (5, 13, 19):
        assert np.array_equal(PO.ellipse_kernel(k), Z[f'ellipse{k}'])
    assert Z['ellipse5'].tolist() == [[0, 0, 1, 0, 0], [1, 1, 1, 1, 1], [1, 1, 1, 1, 1], [1, 1, 1, 1, 1], [0, 0, 1, 0, 0]]


def test_cpu_quota_and_thread_cap():
    """hostutil.cpu_quota: between 1 and the visible cores; cap_threads_to_cpu_quota never raises torch's thread count."""
    import os
    import torch
    from ctrlhair_amd import hostutil as U
    q = U.cpu_quota()
    assert 1 <= q <= (os.cpu_count() or 1)
    before = torch.get_num_threads()
    assert U.cap_threads_to_cpu_quota() == q
    assert torch.get_num_threads() == min(before, q)


def test_to_host_returns_private_copies():
    """hostutil.to_host: numpy in, numpy out; CPU tensors as arrays; (device tensors: a private copy per call, checked on the GPU box by the
    Backend tests, which compare successive output() results)."""
    import numpy as np
    import torch
    from ctrlhair_amd import hostutil as U
    a = np.arange(6, dtype=np.float32).reshape(2, 3)
    assert np.array_equal(U.to_host(a), a)
    t = torch.arange(6, dtype=torch.float32).reshape(2, 3)
    assert np.array_equal(U.to_host(t), a)
