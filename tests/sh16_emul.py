"""numpy emulation of the split-operand number format of ctrlhair_amd/csrc/sh16.h (test infrastructure): f16 hi/lo split
with power-of-two scales, the 3-term product sum the f16 MFMA path evaluates, and the scale selection rules."""
import numpy as np

SH16_MAX = 65504.0
ACT_SCALE = 8.0


def scale_for_bound(bound: float) -> float:
    if not (bound > 0) or not np.isfinite(bound):
        return 1.0
    m, e = np.frexp(np.float32(bound))
    return float(np.ldexp(1.0, 15 - int(e)))


def dyn_extra(amax: float) -> float:
    if amax == 0 or (0.5 <= amax <= SH16_MAX) or not np.isfinite(amax):
        return 1.0
    return scale_for_bound(amax)


def split(x, s=1.0):
    """-> (hi, lo) as float64 arrays holding f16 values of x*s (saturating like MODE.FP16_OVFL)."""
    t = np.asarray(x, np.float32) * np.float32(s)
    with np.errstate(over='ignore'):
        hi = np.clip(t, -SH16_MAX, SH16_MAX).astype(np.float16)
        lo = np.clip(t - hi.astype(np.float32), -SH16_MAX, SH16_MAX).astype(np.float16)
    return hi.astype(np.float64), lo.astype(np.float64)


def dot3(w, x, wscale=1.0, xscale=1.0):
    """sum_k w_k x_k the way conv_sh16.h computes it: three f16 x f16 products per term (exact in the f32
    accumulator; accumulated here in f64), then the exact power-of-two undo."""
    wh, wl = split(w, wscale)
    xh, xl = split(x, xscale)
    acc = (wl * xh).sum(-1) + (wh * xl).sum(-1) + (wh * xh).sum(-1)
    return acc / (wscale * xscale)


def row_scale(w) -> float:
    """2^k with max|row| * 2^k in [2^14, 2^15) (net_common.h sh16_row_exponent)."""
    return scale_for_bound(float(np.abs(w).max()))
