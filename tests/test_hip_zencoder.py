"""GPU parity of the HIP Zencoder (ch_sean_encode) vs oracle and reference-made golden vectors."""
import numpy as np
import pytest
import torch

from tests.golden_util import ZENC_CASES, ZencCase

pytestmark = pytest.mark.gpu
TOL = 1e-3
_gen = {}


PATHS = ['f32', 'f16x3']


def gen(path='f32'):
    if path not in _gen:
        from ctrlhair_amd import procedural as P
        from ctrlhair_amd.sean.generator import SeanGenerator
        _gen[path] = SeanGenerator(0, f16x3=int(path == 'f16x3')).load_state_dict(P.sean_state_dict(0, 16), max_batch=2,
                                                                                   max_size=512)
    return _gen[path]


@pytest.mark.parametrize('path', PATHS)
def test_feature_map_and_codes_vs_oracle(hip_lib, path):
    from ctrlhair_amd import procedural as P
    from oracle import sean_oracle as O
    g = gen(path)
    B, S = 3, 128     # B > max_batch exercises chunking
    lab, img = P.blocky_labels(B, S, grid=8, seed=9), P.synthetic_images(B, S, seed=10)
    taps = {}
    ref = O.zencoder_forward(O.to_torch(P.sean_state_dict(0, 16)), img, lab, taps=taps).numpy()
    feat = torch.zeros(2, 512, S // 2, S // 2, device=g.device)
    g.handle.sean_set_tap('zenc.feat', feat.data_ptr())
    codes = g.encode(torch.from_numpy(img).to(g.device), torch.from_numpy(lab).to(g.device))
    torch.cuda.synchronize()
    g.handle.sean_set_tap('zenc.feat', None)
    # tap holds the last chunk (sample 2 -> slot 0)
    assert float((feat[0].cpu() - taps['zenc.feat'][2]).abs().max()) <= TOL
    assert np.abs(codes.cpu().numpy() - ref).max() <= TOL


@pytest.mark.parametrize('path', PATHS)
@pytest.mark.parametrize('name', ZENC_CASES)
def test_golden(hip_lib, name, path):
    c = ZencCase(name)
    g = gen(path)
    codes = g.encode(torch.from_numpy(c.img).to(g.device), torch.from_numpy(c.labels).to(g.device))
    torch.cuda.synchronize()
    out = codes.cpu().numpy()
    assert np.isfinite(out).all()
    assert np.abs(out - c.codes).max() <= TOL
    # absent regions are exactly zero rows (architecture.py:199)
    absent = np.abs(c.codes).sum(-1) == 0
    assert (out[absent] == 0).all()


def test_size_not_a_multiple_of_the_tiles(hip_lib):
    """S = 288: every stage of the f16 path (space-to-depth stride-2 convs at 144 / 72, depth-to-space ConvTranspose at 72,
    sliced instance norms) meets partial tiles and slices; bar: the oracle on the same inputs."""
    from ctrlhair_amd import procedural as P
    from oracle import sean_oracle as O
    g = gen('f16x3')
    B, S = 2, 288
    lab, img = P.blocky_labels(B, S, grid=9, seed=19), P.synthetic_images(B, S, seed=20)
    ref = O.zencoder_forward(O.to_torch(P.sean_state_dict(0, 16)), img, lab).numpy()
    codes = g.encode(torch.from_numpy(img).to(g.device), torch.from_numpy(lab).to(g.device))
    torch.cuda.synchronize()
    assert float(np.abs(codes.cpu().numpy() - ref).max()) <= TOL


def test_split_encode_equals_encode(hip_lib):
    """ch_sean_encode_features + ch_sean_encode_regions == ch_sean_encode (the same kernels; the region means accumulate with
    float atomics, so equality is up to their summation order), and a regions call without its features call is refused."""
    from ctrlhair_amd import procedural as P
    g = gen('f16x3')
    B, S = 2, 256
    lab = torch.from_numpy(P.blocky_labels(B, S, grid=8, seed=29)).to(g.device)
    img = torch.from_numpy(P.synthetic_images(B, S, seed=30)).to(g.device)
    whole = g.encode(img, lab)
    g.encode_features(img)
    parts = g.encode_regions(lab)
    torch.cuda.synchronize()
    again = g.encode(img, lab)
    torch.cuda.synchronize()
    print('run-to-run', float((whole - again).abs().max()), 'split vs whole', float((whole - parts).abs().max()))
    assert float((whole - parts).abs().max()) <= 1e-6
    with pytest.raises(RuntimeError):
        g.encode_regions(lab)                  # consumed: needs a new encode_features


def test_convtranspose_as_phase_gemms(hip_lib):
    """Option sean.convt_gemm (exact-f32 path, default 1): ConvTranspose2d(128, 256, k3, s2, p1, op1) (architecture.py:167-170) as four phase
    GEMMs over shifted views of its input, the following InstanceNorm + lrelu reading the phase planes (misc_kernels.hip convt_shift4 /
    instnorm_act_d2s), taken by calls with at least 16384 input pixels per chunk -- against the Winograd phase convs of the same library
    (option 0) on the feature map and the codes, and against the oracle."""
    from ctrlhair_amd import procedural as P
    from ctrlhair_amd.sean.generator import SeanGenerator
    from oracle import sean_oracle as O
    sd = P.sean_state_dict(0, 16)
    B, S = 4, 256
    on = SeanGenerator(0, f16x3=0).load_state_dict(sd, max_batch=B, max_size=S)
    off = SeanGenerator(0, f16x3=0, options={'sean.convt_gemm': 0}).load_state_dict(sd, max_batch=B, max_size=S)
    lab, img = P.blocky_labels(B, S, grid=8, seed=39), P.synthetic_images(B, S, seed=40)
    out = {}
    for name, g in (('on', on), ('off', off)):
        feat = torch.zeros(B, 512, S // 2, S // 2, device=g.device)
        g.handle.sean_set_tap('zenc.feat', feat.data_ptr())
        codes = g.encode(torch.from_numpy(img).to(g.device), torch.from_numpy(lab).to(g.device))
        torch.cuda.synchronize()
        g.handle.sean_set_tap('zenc.feat', None)
        out[name] = (feat.cpu().numpy(), codes.cpu().numpy())
    df, dc = float(np.abs(out['on'][0] - out['off'][0]).max()), float(np.abs(out['on'][1] - out['off'][1]).max())
    ref = O.zencoder_forward(O.to_torch(sd), img, lab).numpy()
    dr = float(np.abs(out['on'][1] - ref).max())
    print(f'phase GEMMs vs Winograd phase convs: feature map {df:.3e}, codes {dc:.3e}; codes vs oracle {dr:.3e}')
    assert np.isfinite(out['on'][0]).all() and df <= 2e-5 and dc <= 2e-6 and dr <= TOL
    assert df > 0, 'both handles took the same route'
    on.handle.close()
    off.handle.close()
