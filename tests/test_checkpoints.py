"""Reference checkpoint tree reader / writer / validator."""
import os

import numpy as np
import pytest

from ctrlhair_amd import checkpoints as C
from ctrlhair_amd import hostutil as U
from ctrlhair_amd import procedural as P
from ctrlhair_amd.hair_editor import procedural_weights

NGF = 16


def small_weights():
    w = procedural_weights(0, 64)
    w['sean'] = P.sean_state_dict(0, NGF)
    return w


@pytest.fixture(scope='module')
def tree(tmp_path_factory):
    root = str(tmp_path_factory.mktemp('ckpt'))
    w = small_weights()
    tdirs, sdirs = U.seeded_directions(2, 8, seed=45), U.seeded_directions(4, 16, seed=54)
    C.write_reference_layout(root, w, tdirs, sdirs, ddp_prefix=True)
    return root, w, tdirs, sdirs, C.reference_checkpoints(root)      # read back once (the shape VAE alone is ~1 GB)


def test_layout_roundtrip(tree):
    root, w, tdirs, sdirs, r = tree
    for rel in (C.SEAN_FILE, C.BISENET_FILE):
        assert os.path.isfile(os.path.join(root, rel))
    for which in C.EXPERIMENTS:
        d = os.path.join(root, *C.EXPERIMENTS[which], 'checkpoints')
        assert open(os.path.join(d, 'latest_checkpoint')).read().endswith('.ckpt\n')
    for m in C.MODELS:
        assert sorted(r[m]) == sorted(w[m]), m                       # 'module.' prefixes stripped
        for k in w[m]:
            assert np.array_equal(np.asarray(r[m][k]), np.asarray(w[m][k])), (m, k)
    assert len(r['texture_dirs']) == 2 and len(r['shape_dirs']) == 4
    assert all(np.allclose(a, b) for a, b in zip(r['texture_dirs'], tdirs))
    assert all(np.allclose(a, b) for a, b in zip(r['shape_dirs'], sdirs))


def test_experiment_dir_falls_back_to_config_id(tmp_path):
    os.makedirs(tmp_path / 'model_trained' / 'shape' / '054__renamed_by_user' / 'checkpoints')
    assert C.experiment_dir(str(tmp_path), 'shape').endswith('054__renamed_by_user')
    assert C.experiment_dir(str(tmp_path), 'color_texture').endswith(C.EXPERIMENTS['color_texture'][1])   # nothing there: default


def test_validate_is_strict(tree):
    root, w, _, _, r = tree
    assert C.validate(r, ngf=NGF) == []
    bad = {m: dict(r[m]) for m in C.MODELS}
    k0 = sorted(bad['shape'])[0]
    del bad['shape'][k0]
    k1 = sorted(bad['sean'])[0]
    bad['sean'][k1] = np.zeros((1, 2, 3), np.float32)
    bad['bisenet']['not.a.real.key'] = np.zeros(3, np.float32)
    msgs = C.validate(bad, ngf=NGF)
    assert any('missing key ' + k0 in m for m in msgs) and any('shape of ' + k1 in m for m in msgs)
    assert any('unexpected key not.a.real.key' in m for m in msgs) and len(msgs) == 3
    assert C.validate(r, ngf=64) != []                                # wrong generator width is reported, not ignored


def test_npz_container_roundtrip(tree, tmp_path):
    root, w, _, _, r = tree
    C.save_npz(str(tmp_path / 'all.npz'), r)
    back = C.load_npz(str(tmp_path / 'all.npz'))
    assert back.get('_origin') == r.get('_origin') == 'reference'      # the arithmetic default of a released checkpoint survives
    for m in C.MODELS:
        assert sorted(back[m]) == sorted(w[m])
        assert all(np.array_equal(back[m][k], np.asarray(w[m][k])) for k in w[m])
    assert len(back['texture_dirs']) == 2 and len(back['shape_dirs']) == 4
    assert C.main(['check', root]) == 1 and C.validate(back, ngf=NGF) == []     # CLI validates at ngf=64 -> reports the tiny G


@pytest.mark.gpu
def test_editor_from_checkpoint_tree_equals_in_memory_weights(hip_lib, tree):
    import torch
    from ctrlhair_amd.hair_editor import HairEditor
    root, w, tdirs, sdirs, _ = tree
    a = HairEditor(True, True, weights=root, device=0, f16x3=True)
    b = HairEditor(True, True, weights=w, device=0, texture_dirs=tdirs, shape_dirs=sdirs, f16x3=True)
    # (a released checkpoint tree defaults to the exact-f32 path, in-memory / procedural weights to the split-operand path)
    d = HairEditor(True, True, weights=root, device=0)
    assert d.models.generator.f16x3 is False and b.models.generator.f16x3 is True
    # ... and so does the same tree read into memory first (reference_checkpoints tags its dict)
    from ctrlhair_amd import checkpoints as C
    from ctrlhair_amd.hair_editor import is_released_checkpoint
    mem = C.reference_checkpoints(root)
    assert is_released_checkpoint(mem) and is_released_checkpoint(root) and not is_released_checkpoint(w)
    e = HairEditor(True, True, weights=mem, device=0)
    assert e.models.generator.f16x3 is False
    assert all(torch.equal(x, y) for x, y in zip(a.texture_dirs, b.texture_dirs))
    assert all(torch.equal(x, y) for x, y in zip(a.shape_dirs, b.shape_dirs))
    labels, codes = P.blocky_labels(1, 256, seed=3), P.style_codes(1, seed=4)
    noise = torch.from_numpy(P.noise_planes(1, 256, NGF, seed=5)).cuda()
    ia, ib = a.gen_imgs(codes, labels, noise=noise), b.gen_imgs(codes, labels, noise=noise)
    assert torch.equal(ia, ib)
