"""CPU tests of the split-operand number format (ctrlhair_amd/csrc/sh16.h) on its numpy emulation: with the
power-of-two scaling rules the 3-term f16 product sum is f32-class at every weight / activation magnitude; without
them it is not (the round-1 behaviour VERDICT.md measured)."""
import numpy as np
import pytest

from tests import sh16_emul as E

K = 1152          # 128 channels x 9 taps: the SPADE gamma/beta reduction length


def _case(sigma_w, mag_x, seed=0):
    rng = np.random.default_rng(seed)
    w = (rng.standard_normal((64, K)) * sigma_w).astype(np.float32)
    x = rng.standard_normal(K).astype(np.float32)
    x = (np.maximum(x, 0.2 * x) * mag_x).astype(np.float32)
    exact = (w.astype(np.float64) * x.astype(np.float64)).sum(-1)
    norm = (np.abs(w).astype(np.float64) * np.abs(x)).sum(-1)      # normwise error scale of a dot product
    f32 = np.array([np.dot(w[i], x) for i in range(w.shape[0])], np.float32)   # what an fp32 fma chain gives (any order)
    return w, x, exact, norm, f32


@pytest.mark.parametrize('sigma_w', [1e3, 1.0, 2e-2, 1e-4, 1e-6])
def test_scaled_weights_are_f32_class(sigma_w):
    w, x, exact, norm, f32 = _case(sigma_w, 1.0)
    got = np.array([E.dot3(w[i], x, E.row_scale(w[i]), E.ACT_SCALE) for i in range(w.shape[0])])
    err = np.abs(got - exact) / norm
    ref = np.abs(f32 - exact) / norm
    print(f'sigma_w={sigma_w:g}: split {err.max():.2e}  fp32 {ref.max():.2e}')
    assert err.max() <= 2.0 ** -21            # 22-bit operands: a few ulp of fp32 relative to sum |w||x|


@pytest.mark.parametrize('sigma_w,floor', [(1e-4, 1e-5), (1e-6, 1e-3)])
def test_unscaled_weights_are_not(sigma_w, floor):
    """Documents why the scaling exists: un-scaled small weights lose their lo halves to f16 subnormals."""
    w, x, exact, norm, _ = _case(sigma_w, 1.0)
    got = np.array([E.dot3(w[i], x) for i in range(w.shape[0])])
    assert (np.abs(got - exact) / norm).max() > floor


@pytest.mark.parametrize('mag_x', [1e-6, 1e-3, 0.05, 1.0, 40.0, 6e4, 3e7])
def test_activation_magnitudes_with_dynamic_scale(mag_x):
    """First pass at scale 8; if max|x*8| leaves [0.5, 65504] the tensor is rewritten at the corrected scale."""
    w, x, exact, norm, _ = _case(0.02, mag_x)
    amax = float(np.abs(x * np.float32(E.ACT_SCALE)).max())
    s = E.ACT_SCALE * E.dyn_extra(amax)
    assert float(np.abs(x).max()) * s <= E.SH16_MAX
    got = np.array([E.dot3(w[i], x, E.row_scale(w[i]), s) for i in range(w.shape[0])])
    assert (np.abs(got - exact) / norm).max() <= 2.0 ** -21


def test_window_edges():
    assert E.dyn_extra(0.0) == 1.0 and E.dyn_extra(0.5) == 1.0 and E.dyn_extra(65504.0) == 1.0
    assert E.dyn_extra(65505.0) * 65505.0 < 2 ** 15 and E.dyn_extra(65505.0) * 65505.0 >= 2 ** 14
    assert E.dyn_extra(0.49) * 0.49 >= 2 ** 14
    assert E.dyn_extra(float('inf')) == 1.0
    for b in (1e-9, 0.3, 1.0, 255.9, 256.0, 1e5, 1e30):
        assert 2 ** 14 <= b * E.scale_for_bound(b) < 2 ** 15
    # saturating conversion: finite out-of-range values stay finite, hi + lo extends the range to 2 x 65504
    hi, lo = E.split(np.array([7e4, -1e6], np.float32))
    assert hi[0] == 65504 and hi[0] + lo[0] == 70000 and hi[1] == -65504 and np.isfinite(lo).all()
