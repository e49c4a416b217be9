"""Procedural (synthetic) weights for the SEAN generator, keyed exactly like the
reference's ``latest_net_G.pth`` state dict.

No trained checkpoint ships with the reference (external_model_params/ is
git-ignored), so benchmarks and parity tests run on weights produced here.  They are a
pure function of ``(seed, ngf)`` plus the small committed calibration table
``ctrlhair_amd/data/sean_calib.npz`` (BN running statistics and the conv_img gain that
make the random net non-degenerate: un-saturated tanh output, O(1) activations).  The
table is produced once by ``tests/golden/make_calibration.py``; it is data, not code.

Key families follow SURVEY.md Appendix A / the reference modules:
  sean_codes/models/networks/generator.py:24-54, architecture.py:21-67,154-176,
  normalization.py:71-106,191-247 and torch.nn.utils.spectral_norm
  (weight_orig / weight_u / weight_v buffers).
"""
import os
import zlib
from typing import Dict, Optional

import numpy as np

from .sean import arch

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data')
CALIB_PATH = os.path.join(_DATA, 'sean_calib.npz')


def _rng(seed: int, name: str) -> np.random.Generator:
    # Philox is counter based and platform independent: same (seed, name) -> same stream everywhere.
    return np.random.Generator(np.random.Philox(key=[np.uint64(seed), np.uint64(zlib.crc32(name.encode()))]))


def _normal(seed, name, shape, std):
    return (_rng(seed, name).standard_normal(shape, dtype=np.float32) * np.float32(std)).astype(np.float32)


def _xavier(seed, name, shape):
    # shape = [Cout, Cin, kh, kw] (or [out, in] for Linear)
    rf = int(np.prod(shape[2:])) if len(shape) > 2 else 1
    fan_in, fan_out = shape[1] * rf, shape[0] * rf
    return _normal(seed, name, shape, np.sqrt(2.0 / (fan_in + fan_out)))


def power_iterate(w: np.ndarray, seed: int, name: str, iters: int = 60):
    """Converged spectral-norm buffers (u, v) for a conv weight, in torch's convention
    (torch/nn/utils/spectral_norm.py: weight_mat = weight.reshape(Cout, -1);
    v = normalize(W^T u); u = normalize(W v)).  float64 iterations, float32 result."""
    wm = w.reshape(w.shape[0], -1).astype(np.float64)
    u = _rng(seed, name + '.u0').standard_normal(wm.shape[0])
    u /= np.linalg.norm(u) + 1e-12
    v = None
    for _ in range(iters):
        v = wm.T @ u
        v /= np.linalg.norm(v) + 1e-12
        u = wm @ v
        u /= np.linalg.norm(u) + 1e-12
    return u.astype(np.float32), v.astype(np.float32)


def _spectral_conv(sd, seed, prefix, cout, cin, k, bias=True):
    w = _xavier(seed, prefix + '.weight_orig', (cout, cin, k, k))
    u, v = power_iterate(w, seed, prefix)
    sd[prefix + '.weight_orig'] = w
    sd[prefix + '.weight_u'] = u
    sd[prefix + '.weight_v'] = v
    if bias:
        sd[prefix + '.bias'] = _normal(seed, prefix + '.bias', (cout,), 0.05)


def _ace(sd, seed, a: arch.AceSpec):
    p, C = a.name, a.channels
    sd[p + '.blending_gamma'] = _normal(seed, p + '.blending_gamma', (1,), 1.0)
    sd[p + '.blending_beta'] = _normal(seed, p + '.blending_beta', (1,), 1.0)
    sd[p + '.noise_var'] = _normal(seed, p + '.noise_var', (C,), 0.1)
    for q in (p + '.Spade.param_free_norm', p + '.param_free_norm'):
        sd[q + '.running_mean'] = np.zeros((C,), np.float32)
        sd[q + '.running_var'] = np.ones((C,), np.float32)
        sd[q + '.num_batches_tracked'] = np.zeros((), np.int64)
    # SPADE (normalization.py:237-247): gains chosen so actv is O(1) and gamma/beta_spade ~ 0.5
    sd[p + '.Spade.mlp_shared.0.weight'] = _normal(seed, p + '.Spade.mlp_shared.0.weight',
                                                   (arch.SPADE_HIDDEN, arch.LABEL_NC, 3, 3), 1.0 / 3.0)
    sd[p + '.Spade.mlp_shared.0.bias'] = _normal(seed, p + '.Spade.mlp_shared.0.bias', (arch.SPADE_HIDDEN,), 0.2)
    for g in ('mlp_gamma', 'mlp_beta'):
        sd[f'{p}.Spade.{g}.weight'] = _normal(seed, f'{p}.Spade.{g}.weight', (C, arch.SPADE_HIDDEN, 3, 3), 0.0208)
        sd[f'{p}.Spade.{g}.bias'] = _normal(seed, f'{p}.Spade.{g}.bias', (C,), 0.05)
    if a.styled:
        for j in range(arch.LABEL_NC):
            sd[f'{p}.fc_mu{j}.weight'] = _xavier(seed, f'{p}.fc_mu{j}.weight', (arch.STYLE_LEN, arch.STYLE_LEN))
            sd[f'{p}.fc_mu{j}.bias'] = _normal(seed, f'{p}.fc_mu{j}.bias', (arch.STYLE_LEN,), 0.1)
        for g in ('conv_gamma', 'conv_beta'):
            sd[f'{p}.{g}.weight'] = _normal(seed, f'{p}.{g}.weight', (C, arch.STYLE_LEN, 3, 3), 0.0165)
            sd[f'{p}.{g}.bias'] = _normal(seed, f'{p}.{g}.bias', (C,), 0.05)


def load_calibration(seed: int, ngf: int) -> Optional[Dict[str, np.ndarray]]:
    if not os.path.exists(CALIB_PATH):
        return None
    z = np.load(CALIB_PATH)
    pre = f'ngf{ngf}_seed{seed}/'
    out = {k[len(pre):]: z[k] for k in z.files if k.startswith(pre)}
    return out or None


def sean_state_dict(seed: int = 0, ngf: int = 64, calibrated: bool = True,
                    with_zencoder: bool = True) -> Dict[str, np.ndarray]:
    """Full SPADEGenerator state dict (982 entries at any ngf) as numpy arrays."""
    sd: Dict[str, np.ndarray] = {}
    if with_zencoder:
        # architecture.py:158-176: conv 3->32, 32->64 s2, 64->128 s2, ConvT 128->256, conv 256->512
        for idx, shape in ((1, (32, 3, 3, 3)), (4, (64, 32, 3, 3)), (7, (128, 64, 3, 3)),
                           (10, (128, 256, 3, 3)), (14, (512, 256, 3, 3))):
            n = f'Zencoder.model.{idx}'
            sd[n + '.weight'] = _xavier(seed, n + '.weight', shape)
            nb = shape[1] if idx == 10 else shape[0]   # ConvTranspose2d weight is [in, out, k, k]
            sd[n + '.bias'] = _normal(seed, n + '.bias', (nb,), 0.05)
    sd['fc.weight'] = _normal(seed, 'fc.weight', (16 * ngf, arch.LABEL_NC, 3, 3), 1.0 / 3.0)
    sd['fc.bias'] = _normal(seed, 'fc.bias', (16 * ngf,), 0.1)
    for blk in arch.blocks(ngf):
        _spectral_conv(sd, seed, blk.name + '.conv_0', blk.fmid, blk.fin, 3)
        _spectral_conv(sd, seed, blk.name + '.conv_1', blk.fout, blk.fmid, 3)
        if blk.learned_shortcut:
            _spectral_conv(sd, seed, blk.name + '.conv_s', blk.fout, blk.fin, 1, bias=False)
    for a in arch.aces(ngf):
        _ace(sd, seed, a)
    sd['conv_img.weight'] = _xavier(seed, 'conv_img.weight', (3, ngf, 3, 3))
    sd['conv_img.bias'] = _normal(seed, 'conv_img.bias', (3,), 0.05)

    if calibrated:
        cal = load_calibration(seed, ngf)
        if cal is None:
            raise FileNotFoundError(
                f'no calibration for seed={seed} ngf={ngf} in {CALIB_PATH}; run tests/golden/make_calibration.py')
        apply_calibration(sd, cal)
    return sd


def apply_calibration(sd: Dict[str, np.ndarray], cal: Dict[str, np.ndarray]) -> None:
    for k, v in cal.items():
        if k == 'conv_img.gain':
            sd['conv_img.weight'] = (sd['conv_img.weight'] * np.float32(v)).astype(np.float32)
        else:
            assert k in sd and sd[k].shape == v.shape, k
            sd[k] = v.astype(np.float32)


# ---- synthetic inputs of SURVEY.md 8(d) Config 2 -------------------------------------------------

def blocky_labels(B: int, S: int, seed: int = 1234, grid: int = 16, first: int = 0) -> np.ndarray:
    """uint8 [B,S,S]: nearest-upsampled grid x grid map of randint(0,19), seed+global sample index."""
    out = np.empty((B, S, S), np.uint8)
    rep = S // grid
    for b in range(B):
        g = _rng(seed + first + b, 'labels').integers(0, arch.LABEL_NC, size=(grid, grid), dtype=np.uint8)
        out[b] = np.repeat(np.repeat(g, rep, axis=0), rep, axis=1)
    return out


def style_codes(B: int, seed: int = 2024, first: int = 0) -> np.ndarray:
    """float32 [B,19,512] = tanh(N(0,1)) (real codes are tanh-bounded, architecture.py:175)."""
    out = np.empty((B, arch.LABEL_NC, arch.STYLE_LEN), np.float32)
    for b in range(B):
        out[b] = np.tanh(_rng(seed + first + b, 'codes').standard_normal((arch.LABEL_NC, arch.STYLE_LEN),
                                                                         dtype=np.float32))
    return out


def noise_planes(B: int, S: int, ngf: int = 64, seed: int = 7, first: int = 0) -> np.ndarray:
    """float32 [B, noise_floats_per_sample]: per sample the 18 planes n_k[W,H] (k in ACE execution
    order) that normalization.py:111 would draw with randn(B, W, H, 1), flattened and concatenated."""
    n = arch.noise_floats_per_sample(S, ngf)
    out = np.empty((B, n), np.float32)
    for b in range(B):
        out[b] = _rng(seed + first + b, 'noise').standard_normal((n,), dtype=np.float32)
    return out


def synthetic_images(B: int, S: int, seed: int = 31, first: int = 0) -> np.ndarray:
    """float32 [B,3,S,S] in (-1,1): smooth colour blobs (bilinear-upsampled 8x8 grid) + fine noise -- a stand-in for
    portraits (no dataset ships)."""
    out = np.empty((B, 3, S, S), np.float32)
    t = (np.arange(S, dtype=np.float32) + 0.5) / S * 7.0
    i0 = np.clip(np.floor(t).astype(np.int64), 0, 6)
    f = (t - i0).astype(np.float32)
    for b in range(B):
        r = _rng(seed + first + b, 'image')
        g = r.standard_normal((3, 8, 8), dtype=np.float32)
        rows = g[:, i0, :] * (1 - f)[None, :, None] + g[:, i0 + 1, :] * f[None, :, None]
        img = rows[:, :, i0] * (1 - f)[None, None, :] + rows[:, :, i0 + 1] * f[None, None, :]
        img = img + 0.15 * r.standard_normal((3, S, S), dtype=np.float32)
        out[b] = np.tanh(img)
    return out
