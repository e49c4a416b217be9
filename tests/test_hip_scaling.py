"""GPU tests that the f16 MFMA paths are f32-class BY CONSTRUCTION (ctrlhair_amd/csrc/sh16.h), not only on the weight
distribution the other tests use: weights rescaled by 1e-4 ... 1e+3 (compensated in the following layer), activations
driven to ~6e4 and to ~1e-6, and labels outside 0..18.  Bar: |HIP - oracle| <= 1e-3 per pixel on every path (the oracle runs
the same modified weights), plus normwise f32-class agreement of the affected conv outputs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 1e-3
PATHS = ['f32', 'f16x3ws', 'f16x3nows']
DBG = {'f16x3ws': 64, 'f16x3nows': 128}


def _gen(sd, path, max_batch, max_size):
    from ctrlhair_amd.sean.generator import SeanGenerator
    g = SeanGenerator(0, f16x3=0 if path == 'f32' else 1).load_state_dict(sd, max_batch=max_batch, max_size=max_size)
    if path != 'f32':
        g.handle.set_option('sean.dbg', DBG[path])
    return g


def _run(gen, labels, codes, noise, taps=()):
    dev = gen.device
    bufs = {}
    for name, shape in taps:
        bufs[name] = torch.zeros(shape, dtype=torch.float32, device=dev)
        gen.handle.sean_set_tap(name, bufs[name].data_ptr())
    out = gen.generate(torch.from_numpy(labels).to(dev), torch.from_numpy(codes).to(dev), torch.from_numpy(noise).to(dev))
    torch.cuda.synchronize()
    for name, _ in taps:
        gen.handle.sean_set_tap(name, None)
    return out.cpu().numpy(), {k: v.cpu().numpy() for k, v in bufs.items()}


def _scale_effective_conv(sd, prefix, f):
    """Effective spectral-normed weight W_orig / (u . W v) times f, bias times f."""
    sd[prefix + '.weight_u'] = (sd[prefix + '.weight_u'] / np.float32(f)).astype(np.float32)
    if prefix + '.bias' in sd:
        sd[prefix + '.bias'] = (sd[prefix + '.bias'] * np.float32(f)).astype(np.float32)


def _inputs(B, S, ngf):
    from ctrlhair_amd import procedural as P
    return P.blocky_labels(B, S, grid=8), P.style_codes(B), P.noise_planes(B, S, ngf)


_oracle_cache = {}


def _oracle(key, sd, labels, codes, noise, ngf, want_taps=False):
    """CPU oracle forward, evaluated once per test variant (the `path` parametrisation re-uses it: the oracle at ngf=64,
    256x256 takes tens of seconds)."""
    from oracle import sean_oracle as O
    if key not in _oracle_cache:
        taps = {} if want_taps else None
        ref = O.generator_forward(O.to_torch(sd), labels, codes, noise, ngf, taps=taps).numpy()
        _oracle_cache[key] = (ref, taps)
    return _oracle_cache[key]


@pytest.mark.parametrize('path', PATHS)
@pytest.mark.parametrize('f', [2.0 ** -13, 1e3])
def test_weight_magnitudes(hip_lib, path, f):
    """up_2 of an ngf=16, S=128 generator: conv_0's effective weights x f (undone by ace_1's BN statistics), the SPADE
    hidden layer x f (undone in mlp_gamma/beta), the style convs x f (undone in fc_mu, which makes the style projections
    f times smaller or larger than usual)."""
    from ctrlhair_amd import procedural as P
    from oracle import sean_oracle as O
    ngf, S, B = 16, 128, 2
    sd = dict(P.sean_state_dict(0, ngf))
    blk = 'up_2'
    _scale_effective_conv(sd, blk + '.conv_0', f)
    a1 = blk + '.ace_1'
    sd[a1 + '.param_free_norm.running_mean'] = sd[a1 + '.param_free_norm.running_mean'] * np.float32(f)
    sd[a1 + '.param_free_norm.running_var'] = sd[a1 + '.param_free_norm.running_var'] * np.float32(f) ** 2
    sd[a1 + '.noise_var'] = sd[a1 + '.noise_var'] * np.float32(f)
    a0 = blk + '.ace_0'
    for k in ('.Spade.mlp_shared.0.weight', '.Spade.mlp_shared.0.bias'):
        sd[a0 + k] = sd[a0 + k] * np.float32(f)
    for k in ('.Spade.mlp_gamma.weight', '.Spade.mlp_beta.weight'):
        sd[a0 + k] = sd[a0 + k] / np.float32(f)
    for k in ('.conv_gamma.weight', '.conv_beta.weight'):
        sd[a0 + k] = sd[a0 + k] * np.float32(f)
    for j in range(19):
        for k in ('.weight', '.bias'):
            sd[f'{a0}.fc_mu{j}{k}'] = sd[f'{a0}.fc_mu{j}{k}'] / np.float32(f)
    labels, codes, noise = _inputs(B, S, ngf)
    ref, taps = _oracle(('weights', f), sd, labels, codes, noise, ngf, want_taps=True)
    gen = _gen(sd, path, 4, S)         # max_batch 4: the style LUT is built by the f16x3 GEMM (<= 3 would take the f32 GEMV)
    names = [blk + '.h0', blk + '.dx', blk + '.h1']
    img, got = _run(gen, labels, codes, noise, [(n, tuple(taps[n].shape)) for n in names])
    for n in names:
        r = taps[n].numpy()
        rel = float(np.abs(got[n] - r).max() / np.abs(r).max())
        print(f'{path} f={f:g} {n}: max|ref| {np.abs(r).max():.3e}  normwise err {rel:.2e}')
        assert rel <= 2e-5, (n, rel)
    d = float(np.abs(img - ref).max())
    print(f'{path} f={f:g}: image max |delta| {d:.3e}')
    assert np.isfinite(img).all() and d <= TOL
    if path != 'f32':          # f = 1e3 makes the style projections ~1e-3: their slot must show a rescaled tensor
        rep = gen.handle.sean_scale_report()
        assert (rep['ace'] > 0).sum() == 18 and (rep['style'] > 0).sum() == 15
        if f > 1:
            assert ((rep['style'] > 0) & (rep['style'] < 0.5)).any()
    gen.handle.close()


@pytest.mark.parametrize('path', PATHS)
@pytest.mark.parametrize('target', [6e4, 3e-6])
def test_activation_magnitudes(hip_lib, path, target):
    """The ACE outputs of up_2 (ngf=64, S=256, B=3: large enough for the fused-shortcut kernel) are driven to max |h| ~
    `target` (h' = F h) and the following convs are divided by F, so the image is unchanged.  ace_s and ace_0 get different F: the fused conv_1 + conv_s sees two inputs whose recorded
    scales differ."""
    from ctrlhair_amd import procedural as P
    from oracle import sean_oracle as O
    ngf, S, B = 64, 256, 3
    base = P.sean_state_dict(0, ngf)
    labels, codes, noise = _inputs(B, S, ngf)
    ref0, taps = _oracle(('base64', ), base, labels, codes, noise, ngf, want_taps=True)
    blk = 'up_2'
    sd = dict(base)
    plan = {'ace_s': ('hs', 'conv_s', 1.0), 'ace_0': ('h0', 'conv_0', 1.0), 'ace_1': ('h1', 'conv_1', 2.0 ** -10)}
    factors = {}
    for ace, (tapname, conv, rel) in plan.items():
        hmax = float(taps[f'{blk}.{tapname}'].abs().max())
        F = float(2.0 ** np.round(np.log2(target * rel / hmax)))
        if ace == 'ace_1' and target < 1:
            F = 1.0                       # (small targets: leave h1 alone, hs and h0 go tiny)
        factors[ace] = F
        a = f'{blk}.{ace}'
        if F >= 1:        # h' = F h through gamma' = F gamma + F - 1, beta' = F beta
            for g in ('gamma', 'beta'):
                for pre in ('.Spade.mlp_', '.conv_'):
                    sd[f'{a}{pre}{g}.weight'] = sd[f'{a}{pre}{g}.weight'] * np.float32(F)
                    b = sd[f'{a}{pre}{g}.bias'] * np.float32(F)
                    sd[f'{a}{pre}{g}.bias'] = (b + np.float32(F - 1)) if g == 'gamma' else b
        else:             # F << 1: (1 + gamma') would cancel catastrophically in fp32 -- shrink the normalised input instead
            #               (running_var / F^2: the eps of the BN only loses weight) and beta' = F beta
            sd[a + '.param_free_norm.running_var'] = sd[a + '.param_free_norm.running_var'] / np.float32(F) ** 2
            for pre in ('.Spade.mlp_', '.conv_'):
                sd[f'{a}{pre}beta.weight'] = sd[f'{a}{pre}beta.weight'] * np.float32(F)
                sd[f'{a}{pre}beta.bias'] = sd[f'{a}{pre}beta.bias'] * np.float32(F)
        sd[f'{blk}.{conv}.weight_u'] = (sd[f'{blk}.{conv}.weight_u'] * np.float32(F)).astype(np.float32)
    ref, _ = _oracle(('act', target), sd, labels, codes, noise, ngf)
    assert np.abs(ref - ref0).max() <= 1e-4        # the compensation is exact up to fp32 rounding
    gen = _gen(sd, path, 4, S)
    img, _ = _run(gen, labels, codes, noise)
    d = float(np.abs(img - ref).max())
    print(f'{path} target={target:g} factors={factors}: image max |delta| {d:.3e}')
    assert np.isfinite(img).all() and d <= TOL
    if path != 'f32':
        rep = gen.handle.sean_scale_report()['ace']
        out_of_window = (rep > 65504) if target > 1 else ((rep > 0) & (rep < 0.5))
        print('recorded maxima (|h| x first-pass scale):', np.array2string(rep, precision=3))
        # h0 of up_2 was rewritten with a corrected scale (hs is not: the power of two that aligns conv_s with conv_1 is part
        # of its first-pass scale, and the compensating conv_s weights move it with the activations)
        assert out_of_window.sum() >= 1
    gen.handle.close()


@pytest.mark.parametrize('path', PATHS)
def test_fused_shortcut_with_diverging_scales(hip_lib, path):
    """conv_1 + conv_s share accumulators.  Here the two inputs' magnitudes diverge at run time WITHOUT matching weights:
    h1 of up_2 is driven to ~6e4 and hs to ~4e3 (both leave the first-pass window by different amounts, so their recorded
    scales differ), the block output grows by ~2^13 and up_3's batch-norm statistics absorb it."""
    from ctrlhair_amd import procedural as P
    from oracle import sean_oracle as O
    ngf, S, B = 64, 256, 3
    base = P.sean_state_dict(0, ngf)
    labels, codes, noise = _inputs(B, S, ngf)
    _, taps = _oracle(('base64', ), base, labels, codes, noise, ngf, want_taps=True)
    sd = dict(base)
    F1 = float(2.0 ** np.round(np.log2(6e4 / float(taps['up_2.h1'].abs().max()))))
    F2 = float(2.0 ** np.round(np.log2(4e3 / float(taps['up_2.hs'].abs().max()))))
    for ace, F in (('ace_1', F1), ('ace_s', F2)):
        a = f'up_2.{ace}'
        for g in ('gamma', 'beta'):
            for pre in ('.Spade.mlp_', '.conv_'):
                sd[f'{a}{pre}{g}.weight'] = sd[f'{a}{pre}{g}.weight'] * np.float32(F)
                b = sd[f'{a}{pre}{g}.bias'] * np.float32(F)
                sd[f'{a}{pre}{g}.bias'] = (b + np.float32(F - 1)) if g == 'gamma' else b
    sd['up_2.conv_1.bias'] = sd['up_2.conv_1.bias'] * np.float32(F1)
    for ace in ('ace_s', 'ace_0'):                     # up_3 normalises its input: statistics follow the x F1 growth
        a = f'up_3.{ace}.param_free_norm'
        sd[a + '.running_mean'] = sd[a + '.running_mean'] * np.float32(F1)
        sd[a + '.running_var'] = sd[a + '.running_var'] * np.float32(F1) ** 2
        sd[f'up_3.{ace}.noise_var'] = sd[f'up_3.{ace}.noise_var'] * np.float32(F1)
    ref, _ = _oracle(('fused', ), sd, labels, codes, noise, ngf)
    assert 0.05 < ref.std() < 0.9                     # still an image, not a saturated one
    gen = _gen(sd, path, 4, S)
    img, _ = _run(gen, labels, codes, noise)
    d = float(np.abs(img - ref).max())
    print(f'{path} F1={F1:g} F2={F2:g}: image max |delta| {d:.3e}')
    assert np.isfinite(img).all() and d <= TOL
    if path != 'f32':
        rep = gen.handle.sean_scale_report()['ace']
        print('recorded maxima:', np.array2string(rep, precision=3))
        assert (rep > 65504).sum() >= 2            # h1 and hs of up_2 were both rewritten, by different factors
    gen.handle.close()


@pytest.mark.parametrize('path', PATHS)
def test_labels_outside_the_class_range(hip_lib, path):
    """ids >= 19 (255 is the API's "no class", e.g. ch_shape_encode) are an all-zero one-hot: no table row, no style."""
    from ctrlhair_amd import procedural as P
    from oracle import sean_oracle as O
    ngf, S, B = 16, 64, 2
    sd = P.sean_state_dict(0, ngf)
    labels, codes, noise = P.blocky_labels(B, S, grid=8), P.style_codes(B), P.noise_planes(B, S, ngf)
    labels = labels.copy()
    labels[0, 8:40, 16:48] = 255
    labels[1, :, :5] = 19
    labels[1, 60:, 50:] = 200
    ref = O.generator_forward(O.to_torch(sd), labels, codes, noise, ngf).numpy()
    gen = _gen(sd, path, B, S)
    img, _ = _run(gen, labels, codes, noise)
    d = float(np.abs(img - ref).max())
    print(f'{path}: labels with 19/200/255: max |delta| {d:.3e}')
    assert np.isfinite(img).all() and d <= TOL
    gen.handle.close()


def _bisenet_scaled(F):
    """BiSeNet weights whose every feature map is F times the original while the logits are unchanged: the network is
    positively homogeneous up to its BN shifts, so the first BN scales its output (gamma, beta x F), every later BN on the
    feature path shifts with it (running_mean, beta x F) and the three places that squash features into attention weights
    or logits divide F out again (conv_atten, ffm.conv1, conv_out.conv_out x 1/F)."""
    from ctrlhair_amd import procedural as P
    sd = {k: np.array(v, copy=True) for k, v in P.bisenet_state_dict(0).items()}
    F = np.float32(F)
    for k in list(sd):
        if k.startswith('conv_out16.') or k.startswith('conv_out32.') or 'bn_atten' in k:
            continue
        if k == 'cp.resnet.bn1.weight' or k == 'cp.resnet.bn1.bias':
            sd[k] = sd[k] * F
        elif k.startswith('cp.resnet.bn1.'):
            continue
        elif k.endswith('.running_mean') or (k.endswith('.bias') and ('bn' in k.split('.')[-2] or k.split('.')[-2] == '1')):
            sd[k] = sd[k] * F                          # BN running_mean / beta (downsample.1 is a BN too)
    for k in ('cp.arm16.conv_atten.weight', 'cp.arm32.conv_atten.weight', 'ffm.conv1.weight', 'conv_out.conv_out.weight'):
        sd[k] = sd[k] / F
    return sd


@pytest.mark.parametrize('F', [3e4, 1e-5])
def test_bisenet_activation_magnitudes(hip_lib, F):
    """The BiSeNet trunk keeps f32 activations and splits them into f16 pairs while a conv stages them, at the scale the
    producer's recorded maximum dictates (conv_sh16.h INC4).  With every feature map x 3e4 (maxima beyond 65504 / 8) or x 1e-5
    (far below the first-pass window) the logits must still match the oracle on the same weights."""
    from ctrlhair_amd import lib, models
    from ctrlhair_amd import procedural as P
    from oracle import aux_oracle as A
    from oracle import sean_oracle as O
    sd = _bisenet_scaled(F)
    img = P.synthetic_images(2, 128, seed=77)
    rl, rlab = A.bisenet_forward(O.to_torch(sd), img)
    base, _ = A.bisenet_forward(O.to_torch(P.bisenet_state_dict(0)), img)
    assert float((rl - base).abs().max()) <= 1e-3          # the construction leaves the function unchanged (f32 round-off)
    dev = torch.device('cuda', 0)
    for f16x3 in (True, False):
        fp = models.FaceParsing(lib.Handle(0), dev).load_state_dict(sd, max_batch=2, max_size=128, f16x3=f16x3)
        lab, lg = fp.parse_tensor(torch.from_numpy(img).to(dev), want_logits=True)
        torch.cuda.synchronize()
        d = float((lg.cpu() - rl).abs().max())
        print(f'F={F:g} f16x3={f16x3}: max |logit delta| vs oracle {d:.3e}')
        assert d <= TOL
        top2 = torch.topk(rl, 2, dim=1).values
        assert not ((lab.cpu() != rlab) & ((top2[:, 0] - top2[:, 1]) > 1e-3)).any()
        fp.handle.close()
