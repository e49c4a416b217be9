"""GPU parity of the HIP shape branch, colour MLPs and BiSeNet (through the C ABI and the reference-shaped host
shims) vs the oracle and the reference-made golden vectors.  Float outputs: |delta| <= 1e-3.  Label maps: identical
except where the reference's own top-2 margin is below 1e-3 (an fp32 tie)."""
import os

import numpy as np
import pytest
import torch

from tests.golden_util import GOLDEN

pytestmark = pytest.mark.gpu
TOL = 1e-3
_c = {}
MAX_TIE_FRACTION = 5e-4      # label pixels allowed to differ from the reference's argmax (all of them inside its low-margin set)


def _tie_account(what, bad, low_margin):
    """Integer outputs are an argmax over fp32 logits that legitimately differ at the 1e-6 level: a pixel may flip only where the
    REFERENCE's own top-2 margin is tiny, and the number of flips is measured and bounded -- 'identical up to ties' as a
    statement with a count."""
    nbad, nlow, n = int(bad.sum()), int(low_margin.sum()), bad.size
    print(f'{what}: {nbad} of {n} pixels differ ({nbad / n:.2e}); low-margin pixels of the reference: {nlow} ({nlow / n:.2e})')
    assert not (bad & ~low_margin).any(), f'{what}: {int((bad & ~low_margin).sum())} differences outside the low-margin set'
    assert nbad <= MAX_TIE_FRACTION * n, f'{what}: {nbad} tie flips exceed {MAX_TIE_FRACTION:.0e} of {n} pixels'


def env():
    if not _c:
        from ctrlhair_amd import lib, models
        from ctrlhair_amd import procedural as P
        h = lib.Handle(0)
        dev = torch.device('cuda', 0)
        _c['h'] = h
        _c['dev'] = dev
        _c['shape'] = models.ShapeGenerator(h, dev).load_state_dict(P.shape_state_dict(0), max_batch=2)
        cs = P.color_state_dicts(0)
        _c['color'] = models.ColorTextureModels(h, dev).load_state_dicts(cs['gen'], cs['dis'], cs['rgb'], max_batch=4)
        _c['bise'] = models.FaceParsing(h, dev).load_state_dict(P.bisenet_state_dict(0), max_batch=2, max_size=512)
    return _c


def test_color_mlps(hip_lib):
    e = env()
    z = np.load(os.path.join(GOLDEN, 'color_045.npz'))
    code = torch.from_numpy(z['code']).to(e['dev'])       # B=5 > max_batch=4 -> chunked
    d = e['color'].dis({'code': code})
    r = e['color'].rgb_model({'code': code})
    g = e['color'].gen({'noise': d['noise'], 'noise_curliness': d['noise_curliness'], 'rgb_mean': r['rgb_mean'],
                        'pca_std': r['pca_std']})['code']
    ei = e['color'].edit_infer(code, {'noise_curliness': torch.full((5, 1), 1.0, device=e['dev']),
                                      'rgb_mean': r['rgb_mean'], 'pca_std': r['pca_std']})
    torch.cuda.synchronize()
    assert np.abs(d['noise'].cpu().numpy() - z['noise']).max() <= TOL
    assert np.abs(d['noise_curliness'].cpu().numpy() - z['noise_curliness']).max() <= TOL
    assert np.abs(d['adv'].cpu().numpy() - z['adv']).max() <= TOL
    assert np.abs(r['rgb_mean'].cpu().numpy() - z['rgb_mean']).max() <= TOL * 10   # values ~1e2: 1e-5 relative
    assert np.abs(r['pca_std'].cpu().numpy() - z['pca_std']).max() <= TOL * 10
    assert np.abs(g.cpu().numpy() - z['gen_code']).max() <= TOL
    assert np.abs(ei.cpu().numpy() - z['edit_infer']).max() <= TOL


def test_shape_branch_golden(hip_lib):
    e = env()
    z = np.load(os.path.join(GOLDEN, 'shape_054.npz'))
    sg = e['shape']
    lab = torch.from_numpy(z['labels']).to(e['dev'])
    hc, fc = sg.encode_labels(lab)
    torch.cuda.synchronize()
    assert np.abs(hc.cpu().numpy() - z['hair_code']).max() <= TOL
    assert np.abs(fc.cpu().numpy() - z['face_code']).max() <= TOL
    # reference-shaped API on one-hot tensors gives the same codes
    from ctrlhair_amd.models import mask_label_to_one_hot, mask_one_hot_to_label, split_hair_face
    hair, face = split_hair_face(mask_label_to_one_hot(lab[:, None]))
    assert torch.equal(sg.forward_hair_encoder(hair, testing=True), hc)
    assert torch.equal(sg.forward_face_encoder(face), fc)
    # decode from the golden codes
    ghc, gfc = torch.from_numpy(z['hair_code']).to(e['dev']), torch.from_numpy(z['face_code']).to(e['dev'])
    hl = sg.forward_hair_decoder(ghc, gfc)
    fl = sg.forward_face_decoder(gfc)
    probs = sg.forward_decode_by_code(ghc, gfc)
    out = sg.decode_labels(ghc, gfc)
    torch.cuda.synchronize()
    assert np.abs(hl.cpu().numpy()[:, :, ::4, ::4] - z['hair_logit_sub4']).max() <= TOL
    assert np.abs(fl.cpu().numpy()[:, :, ::4, ::4] - z['face_logit_sub4']).max() <= TOL
    assert np.abs(fl.cpu().numpy()[:, :, 96:160, 96:160] - z['face_logit_crop']).max() <= TOL
    bad = out.cpu().numpy() != z['out_labels']
    _tie_account('shape decoder labels', bad, z['margin'].astype(np.float32) <= 1e-3)
    assert torch.equal(mask_one_hot_to_label(probs).to(torch.uint8), out)
    assert torch.equal(mask_one_hot_to_label(sg.forward_decoder(hl, fl)).to(torch.uint8), out)
    assert float((probs.sum(1) - 1).abs().max()) < 1e-4


def test_shape_branch_vs_oracle_fresh(hip_lib):
    from ctrlhair_amd import procedural as P
    from oracle import aux_oracle as A
    from oracle import sean_oracle as O
    e = env()
    lab = P.blocky_labels(1, 256, seed=777, grid=8)
    sd = O.to_torch(P.shape_state_dict(0))
    rh, rf = A.shape_encode(sd, lab)
    hc, fc = e['shape'].encode_labels(torch.from_numpy(lab).to(e['dev']))
    torch.cuda.synchronize()
    assert float((hc.cpu() - rh).abs().max()) <= TOL and float((fc.cpu() - rf).abs().max()) <= TOL
    rhl, rfl, rprobs, rlab = A.shape_decode(sd, rh, rf)
    probs = e['shape'].forward_decode_by_code(rh.to(e['dev']), rf.to(e['dev']))
    torch.cuda.synchronize()
    assert float((probs.cpu() - rprobs).abs().max()) <= TOL


@pytest.mark.parametrize('name', ['256', '512'])
def test_bisenet_golden(hip_lib, name):
    from ctrlhair_amd import procedural as P
    e = env()
    z = np.load(os.path.join(GOLDEN, f'bisenet_{name}.npz'))
    img = torch.from_numpy(P.synthetic_images(int(z['meta_B']), int(z['meta_S']), seed=int(z['meta_seed']))).to(e['dev'])
    lab, lg = e['bise'].parse_tensor(img, want_logits=True)
    torch.cuda.synchronize()
    lg = lg.cpu().numpy()
    assert np.abs(lg[:, :, ::8, ::8] - z['logits_sub8']).max() <= TOL
    assert np.abs(lg[:, :, 100:164, 60:124] - z['logits_crop']).max() <= TOL
    bad = lab.cpu().numpy() != z['labels']
    _tie_account(f'BiSeNet labels {name}', bad, z['margin'].astype(np.float32) <= 1e-3)
    lab2, _ = e['bise'].parse_tensor(img)
    assert torch.equal(lab, lab2)


def test_bisenet_stagewise_vs_oracle(hip_lib):
    from ctrlhair_amd import procedural as P
    from oracle import aux_oracle as A
    from oracle import sean_oracle as O
    e = env()
    img = P.synthetic_images(3, 128, seed=99)      # B=3 > max_batch=2, non-512 size
    rl, rlab = A.bisenet_forward(O.to_torch(P.bisenet_state_dict(0)), img)
    lab, lg = e['bise'].parse_tensor(torch.from_numpy(img).to(e['dev']), want_logits=True)
    torch.cuda.synchronize()
    assert float((lg.cpu() - rl).abs().max()) <= TOL
    top2 = torch.topk(rl, 2, dim=1).values
    bad = lab.cpu() != rlab
    assert not (bad & ((top2[:, 0] - top2[:, 1]) > 1e-3)).any()


def test_parsing_img_surface(hip_lib):
    from ctrlhair_amd import procedural as P
    e = env()
    img = ((P.synthetic_images(1, 256, seed=5)[0].transpose(1, 2, 0) * 0.5 + 0.5) * 255).astype(np.uint8)
    parsing, pil = e['bise'].parsing_img(img)
    assert parsing.shape == (512, 512) and pil.size == (512, 512) and parsing.max() <= 18
    celeba = e['bise'].swap_parsing_label_to_celeba_mask(parsing)
    assert celeba.shape == (512, 512) and celeba.dtype == np.uint8


def test_shape_decoder_exact_f32_path_matches_golden_too(hip_lib):
    """Option shape.f16x3 = 0 keeps every shape-decoder conv on the exact-f32 kernels (the default, tested above, runs the
    decoder from 4x4 up on the f16x3 split-operand kernels): same golden bar, and the two paths agree far inside it."""
    from ctrlhair_amd import lib, models
    from ctrlhair_amd import procedural as P
    e = env()
    z = np.load(os.path.join(GOLDEN, 'shape_054.npz'))
    h = lib.Handle(0)
    sg = models.ShapeGenerator(h, e['dev']).load_state_dict(P.shape_state_dict(0), max_batch=2, f16x3=False)     # (sets shape.f16x3 = 0)
    ghc, gfc = torch.from_numpy(z['hair_code']).to(e['dev']), torch.from_numpy(z['face_code']).to(e['dev'])
    fl = sg.forward_face_decoder(gfc)
    hl = sg.forward_hair_decoder(ghc, gfc)
    fl2 = e['shape'].forward_face_decoder(gfc)
    hl2 = e['shape'].forward_hair_decoder(ghc, gfc)
    torch.cuda.synchronize()
    assert np.abs(fl.cpu().numpy()[:, :, ::4, ::4] - z['face_logit_sub4']).max() <= TOL
    assert np.abs(hl.cpu().numpy()[:, :, ::4, ::4] - z['hair_logit_sub4']).max() <= TOL
    d = max(float((fl - fl2).abs().max()), float((hl - hl2).abs().max()))
    print('shape decoder f16x3 vs exact f32: max |delta| of the logits', d, ' |logit| max', float(fl.abs().max()))
    assert d <= 2e-4
    h.close()


def test_exact_f32_aux_convs_winograd_equals_direct(hip_lib):
    """Option aux.wino (default 1): the exact-f32 kernels of the shape decoder (shape_branch/model.py:138-143: nearest x2 up-sampling
    folded into each 3x3 conv) and of BiSeNet (face_parsing/resnet.py:36-48 BasicBlocks with residual + ReLU, model.py ARM / head /
    output convs) run their 3x3 stride-1 convs as Winograd F(2x2,3x3) where the output fits the tiles (levels of 16 pixels: pairs of
    samples); against the direct evaluation of the same library (aux.wino = 0), far inside the golden bar."""
    from ctrlhair_amd import lib, models
    from ctrlhair_amd import procedural as P
    e = env()
    z = np.load(os.path.join(GOLDEN, 'shape_054.npz'))
    ghc, gfc = torch.from_numpy(z['hair_code']).to(e['dev']), torch.from_numpy(z['face_code']).to(e['dev'])
    img = torch.from_numpy(P.synthetic_images(2, 512, seed=77)).to(e['dev'])
    outs = {}
    for wino in (1, 0):
        h = lib.Handle(0)
        h.set_option('aux.wino', wino)
        sg = models.ShapeGenerator(h, e['dev']).load_state_dict(P.shape_state_dict(0), max_batch=2, f16x3=False)
        fp = models.FaceParsing(h, e['dev']).load_state_dict(P.bisenet_state_dict(0), max_batch=2, max_size=512, f16x3=False)
        lab, lg = fp.parse_tensor(img, want_logits=True)
        lab1, lg1 = fp.parse_tensor(img[:1], want_logits=True)          # one sample: the 16-pixel level cannot pair samples
        outs[wino] = [sg.forward_face_decoder(gfc).cpu(), sg.forward_hair_decoder(ghc, gfc).cpu(), lg.cpu(), lg1.cpu()]
        torch.cuda.synchronize()
        h.close()
    for a, b, what in zip(outs[1], outs[0], ('face decoder', 'hair decoder', 'BiSeNet logits', 'BiSeNet logits, one sample')):
        d, m = float((a - b).abs().max()), float(b.abs().max())
        print(f'{what}: max |winograd - direct| = {d:.3e}  (max |value| {m:.3g})')
        assert d <= 1e-4 * max(1.0, m), what
    assert float((outs[1][2][:1] - outs[1][3]).abs().max()) <= 1e-4 * max(1.0, float(outs[1][3].abs().max()))


def test_bisenet_non_square_and_exact_f32_option(hip_lib):
    """H != W (256 x 384: partial tiles in both the 32x16 and the 16x16 / 8x8 tilings, stride-2 space-to-depth staging with
    different strides per axis), on the default f16x3 trunk and with option bisenet.f16x3 = 0 (exact-f32 kernels): both
    against the oracle."""
    from ctrlhair_amd import lib, models
    from ctrlhair_amd import procedural as P
    from oracle import aux_oracle as A
    from oracle import sean_oracle as O
    e = env()
    sd = P.bisenet_state_dict(0)
    img = np.ascontiguousarray(P.synthetic_images(2, 384, seed=123)[:, :, :256, :])         # [2, 3, 256, 384]
    rl, rlab = A.bisenet_forward(O.to_torch(sd), img)
    top2 = torch.topk(rl, 2, dim=1).values
    for f16x3 in (True, False):
        fp = models.FaceParsing(lib.Handle(0), e['dev']).load_state_dict(sd, max_batch=2, max_size=512, f16x3=f16x3)
        lab, lg = fp.parse_tensor(torch.from_numpy(img).to(e['dev']), want_logits=True)
        torch.cuda.synchronize()
        assert float((lg.cpu() - rl).abs().max()) <= TOL
        assert not ((lab.cpu() != rlab) & ((top2[:, 0] - top2[:, 1]) > 1e-3)).any()
        fp.handle.close()


def test_shape_encoder_layer0_label_table_equals_the_conv(hip_lib):
    """Exact-f32 shape encoders (shape.f16x3 = 0), option shape.enc_lut (default 1): layer 0 -- a 4x4 stride-2 conv over one-hot mask
    channels + 40 constant positional channels (shape_branch/model.py:74-79,96-100) -- is evaluated as posconst + 16 table rows per
    output pixel straight from the label map (misc_kernels.hip shape_enc_l0) instead of a conv over materialised inputs.  Same real
    number in another f32 association: codes against the conv evaluation of the same library (enc_lut = 0) far inside the golden bar,
    against the reference-made fixture, on 'no class' labels at the image frame, and for the one-encoder calls (NULL outputs)."""
    from ctrlhair_amd import lib, models
    from ctrlhair_amd import procedural as P
    e = env()
    z = np.load(os.path.join(GOLDEN, 'shape_054.npz'))
    lab = z['labels'].copy()
    extra = P.blocky_labels(1, 256, seed=31, grid=16).copy()
    extra[:, :3, :] = 255          # 'no class' along the top frame, hair (13) along the left one
    extra[:, :, :2] = 13
    labs = torch.from_numpy(np.concatenate([lab, extra])[:2]).to(e['dev'])
    codes = {}
    for lut in (1, 0):
        h = lib.Handle(0)
        h.set_option('shape.enc_lut', lut)
        sg = models.ShapeGenerator(h, e['dev']).load_state_dict(P.shape_state_dict(0), max_batch=2, f16x3=False)
        hc, fc = sg.encode_labels(labs)
        hair, face = models.split_hair_face(models.mask_label_to_one_hot(labs[:, None]))
        hc1, fc1 = sg.forward_hair_encoder(hair, testing=True), sg.forward_face_encoder(face)      # one encoder per call
        torch.cuda.synchronize()
        assert torch.equal(hc1, hc) and torch.equal(fc1, fc)
        codes[lut] = (hc.cpu().numpy(), fc.cpu().numpy())
        h.close()
    n = lab.shape[0] if lab.shape[0] < 2 else 2
    assert np.abs(codes[1][0][:n] - z['hair_code'][:n]).max() <= TOL and np.abs(codes[1][1][:n] - z['face_code'][:n]).max() <= TOL
    dh, df = float(np.abs(codes[1][0] - codes[0][0]).max()), float(np.abs(codes[1][1] - codes[0][1]).max())
    print(f'shape encoder layer 0, label table vs conv: max |delta| hair code {dh:.3e}, face code {df:.3e}')
    assert 0 < dh + df and dh <= 2e-5 and df <= 2e-5
