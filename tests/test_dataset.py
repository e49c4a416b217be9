"""Dataset-scale drivers: host logic on CPU, batched masks / codes against the per-image API on GPU."""
import os
import pickle

import numpy as np
import pytest

from ctrlhair_amd import dataset as D


def test_shard_batches_and_keys():
    files = [f'{i:03d}.png' for i in range(11)]
    parts = [D.shard(files, r, 3) for r in range(3)]
    assert sorted(sum(parts, [])) == files and max(map(len, parts)) - min(map(len, parts)) <= 1
    assert parts[1] == files[1::3]
    assert D.shard([], 0, 2) == [] and D.shard(files[:1], 1, 2) == []          # empty / ragged shards
    with pytest.raises(ValueError):
        D.shard(files, 3, 3)
    assert list(D.batches(files, 4)) == [files[0:4], files[4:8], files[8:11]]
    assert D.code_key('CelebaMask_HQ', '123.jpg') == 'CelebaMask_HQ___123'


def test_formats_roundtrip(tmp_path):
    lab = (np.arange(64 * 64).reshape(64, 64) % 19).astype(np.uint8)
    D.write_label_png(str(tmp_path / 'a.png'), lab)
    assert np.array_equal(D.read_gray(str(tmp_path / 'a.png')), lab)          # lossless 8-bit single channel
    cdir = tmp_path / 'sean_code'
    cdir.mkdir()
    codes = {D.code_key('ds', f'{i}.png'): np.full((19, 512), i, np.float32) for i in range(3)}
    for k, v in codes.items():
        with open(cdir / (k + '.pkl'), 'wb') as f:
            pickle.dump(v, f)
    merged = D.merge_pickle_dir_to_dict(str(cdir), str(tmp_path / 'sean_code_dict.pkl'))
    with open(tmp_path / 'sean_code_dict.pkl', 'rb') as f:
        back = pickle.load(f)
    assert sorted(back) == sorted(codes) == sorted(merged)
    assert all(np.array_equal(back[k], codes[k]) and back[k].dtype == np.float32 for k in codes)
    assert D.list_images(str(cdir)) == []                                      # only image extensions are listed


@pytest.mark.gpu
def test_batched_masks_and_codes_equal_per_image_api(hip_lib, tmp_path):
    from PIL import Image
    import torch
    from ctrlhair_amd import procedural as P
    from ctrlhair_amd.hair_editor import HairEditor, procedural_weights
    w = procedural_weights(0, 64)
    w['sean'] = P.sean_state_dict(0, 16)
    he = HairEditor(True, True, weights=w, device=0, img_size=256, max_batch=2)
    img_dir, label_dir, code_dir = tmp_path / 'ds' / 'images_256', tmp_path / 'ds' / 'label', tmp_path / 'codes'
    img_dir.mkdir(parents=True)
    imgs = ((P.synthetic_images(5, 256, seed=21).transpose(0, 2, 3, 1) * 0.5 + 0.5) * 255).astype(np.uint8)
    for i, im in enumerate(imgs):
        Image.fromarray(im).save(img_dir / f'{i:02d}.png')
    # masks: two ranks with ragged shards (3 + 2 files), batch 2
    done = sum((D.extract_masks(he, str(img_dir), str(label_dir), batch=2, rank=r, world=2) for r in range(2)), [])
    assert sorted(done) == D.list_images(str(img_dir))
    for i, im in enumerate(imgs):
        parsing, _ = he.face_parsing.parsing_img(im)
        ref = he.face_parsing.swap_parsing_label_to_celeba_mask(parsing).astype(np.uint8)
        got = D.read_gray(str(label_dir / f'{i:02d}.png'))
        # batch-2 and batch-1 launches may pick different split-K schedules: logits agree to ~1e-6, so labels can differ
        # only at argmax near-ties (random weights make those far more common than trained ones)
        assert got.shape == (512, 512) and (got != ref).mean() < 1e-3
    # codes: batch 2 over 5 files (chunks 2 + 2 + 1), merged dict in the reference's format
    out = D.encode_sean_codes(he, str(img_dir), str(label_dir), str(code_dir), 'ds', batch=2)
    merged = D.merge_pickle_dir_to_dict(str(code_dir), str(tmp_path / 'sean_code_dict.pkl'))
    assert sorted(merged) == [f'ds___{i:02d}' for i in range(5)] == sorted(out)
    for i, im in enumerate(imgs):
        lab = D.read_gray(str(label_dir / f'{i:02d}.png'))
        ref = he.get_code(he.preprocess_img(im), he.preprocess_mask(lab)).cpu().numpy()[0]
        c = merged[f'ds___{i:02d}']
        assert c.shape == (19, 512) and c.dtype == np.float32
        assert np.abs(c - ref).max() <= 1e-5


@pytest.mark.gpu
def test_files_on_disk_against_the_oracle(hip_lib, tmp_path):
    """What lands on disk (label PNGs, code pickles) against the CPU ORACLE of the two networks on the same image files:
    dataset_scripts/script_get_mask.py:55-71 and script_get_sean_code.py:40-62 in the reference's formats."""
    from PIL import Image
    import torch
    from ctrlhair_amd import procedural as P
    from ctrlhair_amd.hair_editor import HairEditor, procedural_weights
    from ctrlhair_amd.models import _MEAN, _STD
    from oracle import aux_oracle as A
    from oracle import sean_oracle as O
    w = procedural_weights(0, 64)
    w['sean'] = P.sean_state_dict(0, 16)
    he = HairEditor(True, True, weights=w, device=0, img_size=256, max_batch=2)
    img_dir, label_dir, code_dir = tmp_path / 'ds' / 'images_256', tmp_path / 'ds' / 'label', tmp_path / 'codes'
    img_dir.mkdir(parents=True)
    imgs = ((P.synthetic_images(3, 256, seed=77).transpose(0, 2, 3, 1) * 0.5 + 0.5) * 255).astype(np.uint8)
    for i, im in enumerate(imgs):
        Image.fromarray(im).save(img_dir / f'{i:02d}.png')
    D.extract_masks(he, str(img_dir), str(label_dir), batch=2)
    D.encode_sean_codes(he, str(img_dir), str(label_dir), str(code_dir), 'ds', batch=2)
    bsd = {k: torch.from_numpy(np.asarray(v)) for k, v in w['bisenet'].items()}
    ssd = O.to_torch(w['sean'])
    mean, std = torch.tensor(_MEAN).view(1, 3, 1, 1), torch.tensor(_STD).view(1, 3, 1, 1)
    for i in range(3):
        im = np.asarray(Image.open(img_dir / f'{i:02d}.png').convert('RGB'))
        # masks: PIL bilinear resize to 512, ToTensor + Normalize, BiSeNet, argmax, remap (my_parsing_util.py:31-54)
        big = np.asarray(Image.fromarray(im).resize((512, 512), Image.BILINEAR))
        x = (torch.from_numpy(big.copy()).permute(2, 0, 1)[None].float() / 255.0 - mean) / std
        lg, ref = A.bisenet_forward(bsd, x)                # (logits, CelebAMask-HQ ids)
        top2 = torch.topk(lg, 2, dim=1).values
        ref = ref.numpy()[0]
        got = D.read_gray(str(label_dir / f'{i:02d}.png'))
        low = ((top2[:, 0] - top2[:, 1]) < 5e-3).numpy()[0]
        assert got.shape == ref.shape and not ((got != ref) & ~low).any()
        # codes: the file's label map, nearest-resized to the image size, through the oracle Zencoder
        lab = he.preprocess_mask(got)[0]
        with torch.no_grad():
            rc = O.zencoder_forward(ssd, he.preprocess_img(im).astype(np.float32), lab).numpy()[0]
        with open(code_dir / f'ds___{i:02d}.pkl', 'rb') as f:
            c = pickle.load(f)
        assert c.shape == (19, 512) and c.dtype == np.float32 and np.abs(c - rc).max() <= 1e-3
    he.models.generator.handle.close()
