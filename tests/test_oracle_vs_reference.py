"""Direct module-vs-oracle comparison, only where /root/reference exists (build container)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
import refharness as R   # noqa: E402

pytestmark = pytest.mark.skipif(not R.available(), reason='reference checkout not present')


@pytest.mark.parametrize('B,ui,seed', [(2, False, 100), (1, True, 200)])
def test_generator_fresh_inputs(B, ui, seed):
    from ctrlhair_amd import procedural as P
    from oracle import sean_oracle as O
    ngf, S = 16, 64
    sd = P.sean_state_dict(0, ngf)
    lab = P.blocky_labels(B, S, seed=seed, grid=4)
    codes = P.style_codes(B, seed=seed + 1)
    nz = P.noise_planes(B, S, ngf, seed=seed + 2)
    ref = R.run_generator(sd, lab, codes, nz, ngf, ui_mode=ui)
    mine = O.generator_forward(O.to_torch(sd), lab, codes, nz, ngf).numpy()
    assert np.abs(ref - mine).max() <= 1e-5
