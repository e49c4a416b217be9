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
