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
import functools
import os
import threading
import zlib
from typing import Dict, Optional

import numpy as np

from .sean import arch

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data')
CALIB_PATH = os.path.join(_DATA, 'sean_calib.npz')


# Held by every state-dict builder below and by ctrlhair_amd.checkpoints.expected(), which swaps the tensor makers for shape-only
# stand-ins while it runs: a builder in another thread must never see the swapped makers (re-entrant: builders call each other).
MAKER_LOCK = threading.RLock()


def _locked(fn):
    @functools.wraps(fn)
    def wrapper(*a, **k):
        with MAKER_LOCK:
            return fn(*a, **k)
    return wrapper


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


@_locked
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


def face_like_labels(S: int, seed: int) -> np.ndarray:
    """uint8 [S,S]: concentric / elliptic blobs roughly like a CelebAMask-HQ parsing map (background, skin, hair cap, eyes,
    mouth ...) -- large uniform regions with curved boundaries, unlike the blocky maps of Config 2."""
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


# ---- procedural weights of the other networks on the path (same key names / shapes as the reference modules) ----

def _bn(sd, seed, p, C):
    sd[p + '.weight'] = (0.5 + _rng(seed, p + '.weight').random(C, dtype=np.float32)).astype(np.float32)
    sd[p + '.bias'] = _normal(seed, p + '.bias', (C,), 0.1)
    sd[p + '.running_mean'] = _normal(seed, p + '.running_mean', (C,), 0.1)
    sd[p + '.running_var'] = (0.5 + _rng(seed, p + '.running_var').random(C, dtype=np.float32)).astype(np.float32)
    sd[p + '.num_batches_tracked'] = np.zeros((), np.int64)


@_locked
def shape_state_dict(seed: int = 0) -> Dict[str, np.ndarray]:
    """shape_branch/model.py Generator(cfg 054): hair/face MaskEncoder (7 x conv4x4 s2 + custom LayerNorm) and
    MaskDecoder (Linear -> 7 x [up, conv3x3, LayerNorm] -> conv3x3).  241.0 M parameters."""
    sd: Dict[str, np.ndarray] = {}

    def he(name, shape):   # fan-in scaled normal: LayerNorm follows every conv, so only the ratio to the bias matters
        fan_in = int(np.prod(shape[1:]))
        sd[name] = _normal(seed, name, shape, np.sqrt(2.0 / fan_in))

    for side, cin0, odim in (('hair', 41, 16), ('face', 58, 1024)):
        cin = cin0
        for l in range(7):
            cout = min(2048, 32 << l)
            p = f'{side}_encoder.layers.{l}'
            he(p + '.conv.weight', (cout, cin, 4, 4))
            sd[p + '.conv.bias'] = _normal(seed, p + '.conv.bias', (cout,), 0.05)
            sd[p + '.norm.gamma'] = (0.25 + 0.75 * _rng(seed, p + '.norm.gamma').random(cout, dtype=np.float32)).astype(np.float32)
            sd[p + '.norm.beta'] = _normal(seed, p + '.norm.beta', (cout,), 0.1)
            cin = cout
        heads = ['out_layer'] + (['std_out_layer'] if side == 'hair' else [])
        for hd in heads:
            sd[f'{side}_encoder.{hd}.fc.weight'] = _normal(seed, f'{side}_encoder.{hd}.fc.weight', (odim, 8192), 1.0 / 64)
            sd[f'{side}_encoder.{hd}.fc.bias'] = _normal(seed, f'{side}_encoder.{hd}.fc.bias', (odim,), 0.05)
    for side, idim, oc in (('hair', 1040, 1), ('face', 1024, 18)):
        d = f'{side}_decoder'
        sd[d + '.in_layer.fc.weight'] = _normal(seed, d + '.in_layer.fc.weight', (8192, idim), 1.0 / 32)
        sd[d + '.in_layer.fc.bias'] = _normal(seed, d + '.in_layer.fc.bias', (8192,), 0.1)
        ci = 2048
        for l in range(7):
            co = min(32 << (6 - l), 2048)
            p = f'{d}.layers.{2 * l + 1}'
            he(p + '.conv.weight', (co, ci, 3, 3))
            sd[p + '.conv.bias'] = _normal(seed, p + '.conv.bias', (co,), 0.05)
            sd[p + '.norm.gamma'] = (0.25 + 0.75 * _rng(seed, p + '.norm.gamma').random(co, dtype=np.float32)).astype(np.float32)
            sd[p + '.norm.beta'] = _normal(seed, p + '.norm.beta', (co,), 0.1)
            ci = co
        sd[d + '.out_layer.conv.weight'] = _normal(seed, d + '.out_layer.conv.weight', (oc, 32, 3, 3), 0.25)
        sd[d + '.out_layer.conv.bias'] = _normal(seed, d + '.out_layer.conv.bias', (oc,), 0.5)
    return sd


@_locked
def color_state_dicts(seed: int = 0) -> Dict[str, Dict[str, np.ndarray]]:
    """color_texture_branch cfg 045: {'gen': EigenGenerator, 'dis': Discriminator (encoder), 'rgb': Predictor p004}."""
    gen: Dict[str, np.ndarray] = {}
    gen['main_layer_in.weight'] = _normal(seed, 'ct.gen.in.w', (256, 5), 0.01)     # inputs are 0..255 colours / pca_std ~ 20..120
    gen['main_layer_in.bias'] = _normal(seed, 'ct.gen.in.b', (256,), 0.1)
    for k in range(4):
        o = 512 if k == 3 else 256
        gen[f'main_layer_mid.{k}.1.weight'] = _xavier(seed, f'ct.gen.mid{k}.w', (o, 256))
        gen[f'main_layer_mid.{k}.1.bias'] = _normal(seed, f'ct.gen.mid{k}.b', (o,), 0.05)
        q, _ = np.linalg.qr(_rng(seed, f'ct.gen.U{k}').standard_normal((256, 2)))
        gen[f'subspaces.{k}.U'] = np.ascontiguousarray(q.T).astype(np.float32)     # orthonormal rows (model_eigengan.py:18)
        gen[f'subspaces.{k}.L'] = np.array([6.0, 3.0], np.float32) * np.float32(0.2 + 0.1 * k)
        gen[f'subspaces.{k}.mu'] = _normal(seed, f'ct.gen.mu{k}', (256,), 0.05)
    dis: Dict[str, np.ndarray] = {}
    for k in range(5):
        i, o = (512 if k == 0 else 256), (11 if k == 4 else 256)
        dis[f'net.{k}.fc.weight'] = _xavier(seed, f'ct.dis{k}.w', (o, i))
        dis[f'net.{k}.fc.bias'] = _normal(seed, f'ct.dis{k}.b', (o,), 0.05)
    rgb: Dict[str, np.ndarray] = {}
    for k in range(4):
        i, o = (512 if k == 0 else 256), (4 if k == 3 else 256)
        rgb[f'net.{k}.fc.weight'] = _xavier(seed, f'ct.rgb{k}.w', (o, i))
        rgb[f'net.{k}.fc.bias'] = _normal(seed, f'ct.rgb{k}.b', (o,), 0.05)
        if k < 3:
            _bn(rgb, seed + 17, f'net.{k}.norm', 256)
    # make the predictor emit plausible colours: rgb ~ 128 +- 60, pca_std ~ 60 +- 20
    rgb['net.3.fc.weight'] = (rgb['net.3.fc.weight'] * np.array([[60.0], [60.0], [60.0], [25.0]], np.float32)).astype(np.float32)
    rgb['net.3.fc.bias'] = np.array([128.0, 110.0, 90.0, 60.0], np.float32)
    return {'gen': gen, 'dis': dis, 'rgb': rgb}


@_locked
def bisenet_state_dict(seed: int = 0) -> Dict[str, np.ndarray]:
    """external_code/face_parsing/model.py BiSeNet(19) (+ resnet.py Resnet18): 13.3 M parameters, convs without bias,
    BatchNorm with random affine/running statistics so that BN folding is exercised."""
    sd: Dict[str, np.ndarray] = {}

    def conv(name, cout, cin, k, gain=2.0):
        sd[name + '.weight'] = _normal(seed, 'bise.' + name, (cout, cin, k, k), np.sqrt(gain / (cin * k * k)))

    conv('cp.resnet.conv1', 64, 3, 7)
    _bn(sd, seed, 'cp.resnet.bn1', 64)
    chans = [64, 64, 128, 256, 512]
    for L in range(1, 5):
        for i in range(2):
            cin, cout = (chans[L - 1] if i == 0 else chans[L]), chans[L]
            p = f'cp.resnet.layer{L}.{i}'
            conv(p + '.conv1', cout, cin, 3)
            _bn(sd, seed, p + '.bn1', cout)
            conv(p + '.conv2', cout, cout, 3, gain=1.0)
            _bn(sd, seed, p + '.bn2', cout)
            if i == 0 and L > 1:
                conv(p + '.downsample.0', cout, cin, 1, gain=1.0)
                _bn(sd, seed, p + '.downsample.1', cout)
    for arm, cin in (('cp.arm16', 256), ('cp.arm32', 512)):
        conv(arm + '.conv.conv', 128, cin, 3)
        _bn(sd, seed, arm + '.conv.bn', 128)
        conv(arm + '.conv_atten', 128, 128, 1)
        _bn(sd, seed, arm + '.bn_atten', 128)
    for hd in ('cp.conv_head32', 'cp.conv_head16'):
        conv(hd + '.conv', 128, 128, 3)
        _bn(sd, seed, hd + '.bn', 128)
    conv('cp.conv_avg.conv', 128, 512, 1)
    _bn(sd, seed, 'cp.conv_avg.bn', 128)
    conv('ffm.convblk.conv', 256, 256, 1)
    _bn(sd, seed, 'ffm.convblk.bn', 256)
    conv('ffm.conv1', 64, 256, 1)
    conv('ffm.conv2', 256, 64, 1)
    for out, cin, mid in (('conv_out', 256, 256), ('conv_out16', 128, 64), ('conv_out32', 128, 64)):
        conv(out + '.conv.conv', mid, cin, 3)
        _bn(sd, seed, out + '.conv.bn', mid)
        conv(out + '.conv_out', 19, mid, 1, gain=0.05)
    return sd
