"""BASELINE.json configs[2] at full size on the HIP library: the whole edit (BiSeNet@512 -> remap -> nearest 256 -> shape
encoders -> Zencoder@512 -> colour MLPs + sliders -> shape decoder -> nearest x2 -> SEAN generator@512, ngf=64) against
tests/golden/pipeline512.npz, a fixture composed from the REFERENCE's own modules in the order of ui/backend.py:67-175
(tests/golden/make_golden.py pipeline).  Then `Backend(img_size=512, max_batch=8)` -- the drop-in API -- against the
pipeline on the same portrait and sliders."""
import os

import numpy as np
import pytest
import torch

from tests.golden_util import GOLDEN

pytestmark = pytest.mark.gpu
TOL = 1e-3


@pytest.fixture(scope='module', params=[1, 0], ids=['f16x3', 'f32'])
def pipe(hip_lib, request):
    """Both arithmetic legs bench.py reports for configs[2]: the split-operand f16 MFMA path and the exact-f32 path (Winograd
    Zencoder / BiSeNet / shape-decoder routes and the Winograd generator chained, as `pipeline.f32` of the bench line runs them)."""
    from ctrlhair_amd.pipeline import EditPipeline
    p = EditPipeline(device=0, img_size=512, max_batch=8, f16x3=request.param)
    yield p
    p.close()


def _fixture():
    return np.load(os.path.join(GOLDEN, 'pipeline512.npz'))


def _img_err(z, img):
    d = float(np.abs(img[:, :, ::4, ::4] - z['sub4']).max())
    i = 0
    while f'crop{i}' in z.files:
        y, x = z[f'crop{i}_yx']
        c = z[f'crop{i}']
        d = max(d, float(np.abs(img[:, :, y:y + c.shape[2], x:x + c.shape[3]] - c).max()))
        i += 1
    S = img.shape[-1]
    return max(d, float(np.abs(img.astype(np.float64).sum(axis=(2, 3)) - z['sums']).max() / (S * S)))


def test_stages_against_reference_fixture(pipe):
    """Every stage on the reference's inputs of that stage: continuous outputs <= 1e-3, label maps identical except where
    the reference's own top-2 margin is < 5e-3 (an argmax tie)."""
    from ctrlhair_amd import procedural as P
    z = _fixture()
    B, S, ngf = int(z['meta_B']), int(z['meta_S']), int(z['meta_ngf'])
    dev = pipe.device
    img = torch.from_numpy(P.synthetic_images(B, S, seed=int(z['meta_iseed']))).to(dev)
    # parse
    lab = pipe.parse(img).cpu().numpy()
    low = np.unpackbits(z['labels_low_margin'])[:lab.size].reshape(lab.shape).astype(bool)
    diff = lab != z['labels']
    print(f'parse: {int(diff.sum())} of {diff.size} label pixels differ, {int(low.sum())} low-margin pixels')
    assert not (diff & ~low).any()
    assert diff.sum() <= 5e-4 * diff.size      # 'identical up to ties' with a measured bound
    labels = torch.from_numpy(z['labels']).to(dev)
    # analyse on the reference's label map
    lat = pipe.analyse(img, labels)
    for key, ref in (('shape', 'hair_code'), ('face', 'face_code'), ('codes', 'codes'), ('rgb_mean', 'rgb_mean'),
                     ('pca_std', 'pca_std'), ('texture', 'texture'), ('curliness', 'curliness')):
        got, want = lat[key].cpu().numpy(), z[ref]
        tol = TOL * max(1.0, float(np.abs(want).max()))      # rgb_mean / pca_std live on a 0..255 scale
        d = float(np.abs(got - want).max())
        print(f'{key}: max |delta| {d:.2e} (|ref| max {np.abs(want).max():.2f})')
        assert d <= tol, key
    # sliders + colour generator + shape decoder, fed with the reference's latents
    ref_lat = {'shape': torch.from_numpy(z['hair_code']).to(dev), 'face': torch.from_numpy(z['face_code']).to(dev),
               'codes': torch.from_numpy(z['codes']).to(dev), 'rgb_mean': torch.from_numpy(z['rgb_mean']).to(dev),
               'pca_std': torch.from_numpy(z['pca_std']).to(dev), 'texture': torch.from_numpy(z['texture']).to(dev),
               'curliness': torch.from_numpy(z['curliness']).to(dev)}
    from ctrlhair_amd.pipeline import DEFAULT_SLIDERS
    ed = pipe.apply_sliders(ref_lat, DEFAULT_SLIDERS)
    assert np.array_equal(ed['hsv'].cpu().numpy(), z['hsv'].astype(np.float32))
    assert np.array_equal(ed['rgb'].cpu().numpy(), z['rgb'].astype(np.float32))
    noise = torch.from_numpy(P.noise_planes(B, S, ngf, seed=int(z['meta_nseed']))).to(dev)
    image, mask = pipe.render(ed, noise=noise)
    mask = mask.cpu().numpy()
    lowm = np.unpackbits(z['mask_low_margin'])[:mask.size].reshape(mask.shape).astype(bool)
    dm = mask != z['mask']
    print(f'shape decoder: {int(dm.sum())} of {dm.size} label pixels differ, {int(lowm.sum())} low-margin pixels')
    assert not (dm & ~lowm).any()
    assert dm.sum() <= 5e-4 * dm.size
    # generator on the reference's mask (steps over possible ties of the decoder)
    image, _ = pipe.render(ed, noise=noise, mask=torch.from_numpy(z['mask']).to(dev))
    torch.cuda.synchronize()
    d = _img_err(z, image.cpu().numpy())
    print(f'image: max |delta| vs reference fixture {d:.3e}')
    assert d <= TOL


def test_end_to_end_batch8(pipe):
    """B=8 through edit(): samples 0-1 are the fixture's portraits (labels / image agree with it wherever no argmax tie is
    involved), every sample equals the same sample run alone up to argmax near-ties (no cross-sample op), output finite and
    tanh-bounded."""
    from ctrlhair_amd import procedural as P
    z = _fixture()
    S, ngf, dev = 512, 64, pipe.device
    img = torch.from_numpy(np.concatenate([P.synthetic_images(2, S, seed=int(z['meta_iseed'])),
                                           P.synthetic_images(6, S, seed=501)])).to(dev)
    nz = torch.from_numpy(np.concatenate([P.noise_planes(2, S, ngf, seed=int(z['meta_nseed'])),
                                          P.noise_planes(6, S, ngf, seed=502)])).to(dev)
    st = {}
    out = pipe.edit(img, noise=nz, stages=st).cpu().numpy()
    assert np.isfinite(out).all() and np.abs(out).max() <= 1.0
    lab = st['labels'].cpu().numpy()[:2]
    low = np.unpackbits(z['labels_low_margin'])[:lab.size].reshape(lab.shape).astype(bool)
    assert not ((lab != z['labels']) & ~low).any()
    if np.array_equal(lab, z['labels']) and np.array_equal(st['mask'].cpu().numpy()[:2], z['mask']):
        assert _img_err(z, out[:2]) <= TOL           # no tie anywhere: the un-forced run reproduces the reference image
    # sample 5 alone.  The networks are per-sample, but split-K follows the grid size, i.e. the batch: logits move in the last
    # bits, so the two parsings / decoded masks may differ at argmax near-ties (and only there) ...
    st1 = {}
    pipe.edit(img[5:6], noise=nz[5:6], stages=st1)
    x = ((img[5:6] * 0.5 + 0.5) - pipe.mean) / pipe.std
    lg = pipe.models.face_parsing.parse_tensor(x, want_logits=True)[1]
    top2 = torch.topk(lg, 2, dim=1).values
    flip = st1['labels'][0] != st['labels'][5]
    assert not (flip & ((top2[0, 0] - top2[0, 1]) > 1e-3)).any()
    # ... and with the batch run's parsing and mask forced, the rest of the edit is the same image
    one = pipe.edit(img[5:6], noise=nz[5:6], labels=st['labels'][5:6], mask=st['mask'][5:6]).cpu().numpy()
    # (batch composition changes kernel choices -- a single sample cannot use the sample-pair tiles of the 16-pixel level -- and what
    # differs there in the last bits passes through the F(4x4,3x3) convs of the exact-f32 leg: measured <= 3e-5)
    assert np.abs(one[0] - out[5]).max() <= 1e-4


@pytest.mark.parametrize('f16x3', [True, False], ids=['f16x3', 'f32'])
def test_backend_512_matches_pipeline(hip_lib, f16x3):
    """The drop-in API at BASELINE size: Backend(img_size=512, max_batch=8).set_input_img / change_* / output() on one
    portrait == the pipeline on the same portrait, sliders and noise (uint8 image, +-1 level)."""
    from ctrlhair_amd import procedural as P
    from ctrlhair_amd.pipeline import EditPipeline
    from ctrlhair_amd.ui.backend import Backend
    S, ngf = 512, 64
    be = Backend(2.5, blending=False, img_size=S, max_batch=8, f16x3=f16x3)
    img_u8 = np.clip((P.synthetic_images(1, S, seed=11)[0].transpose(1, 2, 0) * 0.5 + 0.5) * 255.0, 0, 255).astype(np.uint8)
    be.set_input_img(img_u8)
    be.noise = torch.from_numpy(P.noise_planes(1, S, ngf, seed=93)).to(be.device)
    be.change_curliness(1.0)
    be.change_texture(1.5, 0)
    be.change_shape(-1.0, 0)
    be.change_color(1.0, 2)
    got = be.output()
    assert got.shape == (S, S, 3) and got.dtype == np.uint8
    pipe = EditPipeline(models=be.models, img_size=S)
    x = torch.from_numpy(be.preprocess_img(img_u8).astype(np.float32)).to(be.device)
    ref = pipe.edit(x, noise=be.noise, labels=torch.from_numpy(be.input_mask[None].astype(np.uint8)).to(be.device),
                    mask=torch.from_numpy(be.cur_mask[None].astype(np.uint8)).to(be.device))
    want = (ref[0].cpu().numpy().transpose(1, 2, 0) * 127.5 + 127.5).astype(np.uint8)
    d = np.abs(got.astype(np.int32) - want.astype(np.int32))
    print('Backend vs pipeline: max level difference', int(d.max()), 'pixels differing', int((d > 0).sum()))
    assert d.max() <= 1
    be.models.generator.handle.close()


def test_parse_at_256_goes_through_the_512_resize(hip_lib):
    """img_size = 256 (the reference's own default): EditPipeline.parse must see what HairEditor.get_mask sees -- a 512x512
    bilinear resize through BiSeNet, labels nearest-resized back (my_parsing_util.py:34-35, hair_editor.py:331-335) -- and
    not a 256x256 parse, whose labels differ."""
    from ctrlhair_amd import procedural as P
    from ctrlhair_amd.hair_editor import HairEditor, procedural_weights
    from ctrlhair_amd.pipeline import EditPipeline
    w = procedural_weights(0, 16)
    he = HairEditor(weights=w, device=0, img_size=256, max_batch=2)
    pipe = EditPipeline(models=he.models, img_size=256)
    img = P.synthetic_images(1, 256, seed=77)                       # [-1, 1], NCHW
    u8 = np.clip(np.round((img[0].transpose(1, 2, 0) * 0.5 + 0.5) * 255.0), 0, 255).astype(np.uint8)
    ref = he.get_mask(u8)                                             # PIL bilinear -> 512 -> parse -> cv2-style nearest -> 256
    t = torch.from_numpy(u8.transpose(2, 0, 1)[None].astype(np.float32) / 127.5 - 1.0).to(pipe.device)
    got = pipe.parse(t)[0].cpu().numpy()
    direct = he.models.face_parsing.parse_tensor(((t * 0.5 + 0.5) - pipe.mean) / pipe.std)[0][0].cpu().numpy()
    mism = float((got != ref).mean())
    print(f'pipeline.parse vs get_mask at 256: {100 * mism:.3f} % of the labels differ; a direct 256x256 parse: '
          f'{100 * float((direct != ref).mean()):.2f} %')
    dm = float((direct != ref).mean())
    assert mism <= 0.01 and dm >= 10 * mism                    # PIL and torch bilinear agree up to 8-bit rounding ties
