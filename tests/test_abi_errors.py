"""Error behaviour of the C ABI (GPU): every misuse returns a non-zero status with a message in ch_last_error and leaves the
handle usable -- the counterpart of the reference raising from load_state_dict(strict=True) / asserts (hair_editor.py:70-71,
util/util.py:202-208).  Nothing here may silently fall back."""
import ctypes as C

import numpy as np
import pytest
import torch

from ctrlhair_amd import lib as L
from ctrlhair_amd import procedural as P

pytestmark = pytest.mark.gpu
OK, ERR_ARG, ERR_HIP, ERR_STATE, ERR_WEIGHTS = 0, 1, 2, 3, 4


def raw(h, fn, *args):
    rc = getattr(h.lib, fn)(h._h, *args)
    return rc, h.lib.ch_last_error(h._h).decode()


def test_calls_before_finalize_and_bad_arguments(hip_lib):
    h = L.Handle(0)
    buf = torch.zeros(1 << 20, device='cuda')
    p = buf.data_ptr()
    rc, msg = raw(h, 'ch_sean_generate', p, p, None, 0, p, 1, 64, None)
    assert rc == ERR_STATE and 'not finalized' in msg
    rc, msg = raw(h, 'ch_sean_encode', p, p, p, 1, 64, None)
    assert rc == ERR_STATE
    rc, msg = raw(h, 'ch_shape_decode', p, p, None, None, p, None, 1, None)
    assert rc in (ERR_STATE, ERR_HIP) and msg
    rc, msg = raw(h, 'ch_finalize', 7, 1, 64)
    assert rc == ERR_ARG and 'unknown model' in msg
    rc, msg = raw(h, 'ch_set_option', b'no.such.option', 1)
    assert rc == ERR_ARG and 'no.such.option' in msg
    rc, msg = raw(h, 'ch_poisson_blend', p, p, p, p, 2, 2, 1, 10, 1e-7, None, None)
    assert rc == ERR_ARG                                             # H, W >= 3
    rc, msg = raw(h, 'ch_blend_mask', None, p, p, 8, 8, None)
    assert rc == ERR_ARG


def test_missing_and_misshapen_weights_are_reported_by_name(hip_lib):
    sd = P.sean_state_dict(0, 16)
    keys = sorted(sd)
    h = L.Handle(0)
    for k in keys[1:]:                                               # first tensor never loaded
        h.load_tensor(L.MODEL_SEAN, k, np.asarray(sd[k]))
    rc, msg = raw(h, 'ch_finalize', L.MODEL_SEAN, 1, 64)
    assert rc == ERR_WEIGHTS and 'missing tensor' in msg and keys[0] in msg
    h2 = L.Handle(0)
    for k in keys:
        a = np.asarray(sd[k])
        h2.load_tensor(L.MODEL_SEAN, k, a.reshape(-1)[:-1].copy() if k == keys[3] and a.size > 1 else a)
    rc, msg = raw(h2, 'ch_finalize', L.MODEL_SEAN, 1, 64)
    assert rc == ERR_WEIGHTS and 'wrong dtype/size' in msg and keys[3] in msg
    with pytest.raises(TypeError):
        h2.load_tensor(L.MODEL_SEAN, 'x', np.zeros(3, np.float16))    # only f32 / i64 cross the boundary
    # the failed handle recovers once the tensors are complete
    for k in keys:
        h2.load_tensor(L.MODEL_SEAN, k, np.asarray(sd[k]))
    assert raw(h2, 'ch_finalize', L.MODEL_SEAN, 1, 64)[0] == OK


def test_size_limits_and_option_order(hip_lib):
    from ctrlhair_amd.sean.generator import SeanGenerator
    gen = SeanGenerator(0, f16x3=True).load_state_dict(P.sean_state_dict(0, 16), max_batch=2, max_size=64)
    lab = torch.from_numpy(P.blocky_labels(1, 128, grid=8)).cuda()
    codes = torch.from_numpy(P.style_codes(1)).cuda()
    out = torch.empty(1, 3, 128, 128, device='cuda')
    rc, msg = raw(gen.handle, 'ch_sean_generate', lab.data_ptr(), codes.data_ptr(), None, 1, out.data_ptr(), 1, 128, None)
    assert rc != OK and msg                                            # larger than the arena sized at ch_finalize
    rc, msg = raw(gen.handle, 'ch_sean_generate', lab.data_ptr(), codes.data_ptr(), None, 1, out.data_ptr(), 1, 48, None)
    assert rc != OK and msg                                            # not a multiple of 32 (5 stride-2 levels)
    rc, msg = raw(gen.handle, 'ch_set_option', b'sean.f16x3', 0)
    assert rc == ERR_STATE and 'precede' in msg                        # arithmetic is fixed at ch_finalize
    # still healthy
    lab64 = torch.from_numpy(P.blocky_labels(1, 64, grid=8)).cuda()
    img = gen.generate(lab64, codes, None, seed=3)
    assert torch.isfinite(img).all()
