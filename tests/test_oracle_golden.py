"""Pin the oracle (oracle/sean_oracle.py) against golden vectors produced by the imported reference
modules (tests/golden/make_golden.py).  CPU only."""
import pytest

from oracle import sean_oracle as O
from tests.golden_util import SEAN_CASES, Case

TOL = 1e-5   # oracle and reference run the same ATen kernels; batched vs per-sample conv paths differ ~2e-6


@pytest.mark.parametrize('name', SEAN_CASES)
def test_oracle_matches_reference_golden(name):
    c = Case(name)
    sd = O.to_torch(c.state_dict())
    img = O.generator_forward(sd, c.labels, c.codes, c.noise, c.ngf).numpy()
    assert c.max_abs_diff(img) <= TOL


def test_noise_matters():
    """A different noise seed must move the output by far more than the parity tolerance, otherwise the
    1e-3 bar would not detect a wrong noise layout (SURVEY.md 7 'stochastic noise')."""
    import numpy as np
    from ctrlhair_amd import procedural as P
    c = Case('ngf16_S64_ui')
    sd = O.to_torch(c.state_dict())
    a = O.generator_forward(sd, c.labels, c.codes, c.noise, c.ngf).numpy()
    b = O.generator_forward(sd, c.labels, c.codes, P.noise_planes(1, c.S, c.ngf, seed=12345), c.ngf).numpy()
    assert np.abs(a - b).max() > 5e-2
    # transposed noise planes (the classic mistake) must also be visible
    nz = c.noise.copy()
    off = 0
    from ctrlhair_amd.sean import arch
    for r in arch.noise_plane_sizes(c.S, c.ngf):
        nz[:, off:off + r * r] = nz[:, off:off + r * r].reshape(-1, r, r).transpose(0, 2, 1).reshape(-1, r * r)
        off += r * r
    t = O.generator_forward(sd, c.labels, c.codes, nz, c.ngf).numpy()
    assert np.abs(a - t).max() > 5e-2


from tests.golden_util import ZENC_CASES, ZencCase  # noqa: E402


@pytest.mark.parametrize('name', ZENC_CASES)
def test_zencoder_oracle_matches_reference_golden(name):
    import numpy as np
    c = ZencCase(name)
    codes = O.zencoder_forward(O.to_torch(c.state_dict()), c.img, c.labels).numpy()
    assert np.abs(codes - c.codes).max() <= 1e-5


def _g(name):
    import os
    import numpy as np
    from tests.golden_util import GOLDEN
    return np.load(os.path.join(GOLDEN, name))


def test_shape_oracle_matches_reference_golden():
    import numpy as np
    from ctrlhair_amd import procedural as P
    from oracle import aux_oracle as A
    z = _g('shape_054.npz')
    sd = O.to_torch(P.shape_state_dict(0))
    hc, fc = A.shape_encode(sd, z['labels'])
    assert np.abs(hc.numpy() - z['hair_code']).max() <= 1e-5 and np.abs(fc.numpy() - z['face_code']).max() <= 1e-5
    hl, fl, probs, lab = A.shape_decode(sd, z['hair_code'], z['face_code'])
    assert np.abs(hl.numpy()[:, :, ::4, ::4] - z['hair_logit_sub4']).max() <= 1e-4
    assert np.abs(fl.numpy()[:, :, ::4, ::4] - z['face_logit_sub4']).max() <= 1e-4
    bad = lab.numpy() != z['out_labels']
    assert not (bad & (z['margin'].astype(np.float32) > 1e-3)).any()


def test_color_oracle_matches_reference_golden():
    import numpy as np
    import torch
    from ctrlhair_amd import procedural as P
    from oracle import aux_oracle as A
    z = _g('color_045.npz')
    cs = {k: O.to_torch(v) for k, v in P.color_state_dicts(0).items()}
    e = A.color_encode(cs['dis'], z['code']).numpy()
    assert np.abs(e[:, 1:9] - z['noise']).max() <= 1e-5 and np.abs(e[:, 9:10] - z['noise_curliness']).max() <= 1e-5
    p = A.color_predict(cs['rgb'], z['code']).numpy()
    assert np.abs(p[:, :3] - z['rgb_mean']).max() <= 1e-3 and np.abs(p[:, 3:] - z['pca_std']).max() <= 1e-3
    cond = np.concatenate([z['noise_curliness'], z['rgb_mean'], z['pca_std']], 1)
    g = A.color_generate(cs['gen'], z['noise'], cond).numpy()
    assert np.abs(g - z['gen_code']).max() <= 1e-5


@pytest.mark.parametrize('name', ['256', '512'])
def test_bisenet_oracle_matches_reference_golden(name):
    import numpy as np
    from ctrlhair_amd import procedural as P
    from oracle import aux_oracle as A
    z = _g(f'bisenet_{name}.npz')
    img = P.synthetic_images(int(z['meta_B']), int(z['meta_S']), seed=int(z['meta_seed']))
    lg, lab = A.bisenet_forward(O.to_torch(P.bisenet_state_dict(0)), img)
    assert np.abs(lg.numpy()[:, :, ::8, ::8] - z['logits_sub8']).max() <= 1e-4
    assert np.abs(lg.numpy()[:, :, 100:164, 60:124] - z['logits_crop']).max() <= 1e-4
    bad = lab.numpy() != z['labels']
    assert not (bad & (z['margin'].astype(np.float32) > 1e-3)).any()
