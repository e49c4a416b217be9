"""Checkpoint I/O for the reference's own weight files.

The reference ships no weights; users download two folders (README.md:37-43) whose layout is fixed by the code that reads them:

  external_model_params/sean_checkpoints/CelebA-HQ_pretrained/latest_net_G.pth     state dict, util/util.py:202-208
  external_model_params/face_parsing_79999_iter.pth                                state dict, my_parsing_util.py:42-43
  model_trained/color_texture/045__color_texture_final/checkpoints/               {'Model_G','Model_D',...}, hair_editor.py:63-71
  model_trained/color_texture/045__color_texture_final/texture_dir_used/*.pkl     pickled tensors [8], hair_editor.py:82-91
  model_trained/color_encoder/p004___pca_std/checkpoints/                          {'Predictor'}, hair_editor.py:77-79
  model_trained/shape/054__succeed__049__gan_fake_0.5_from_noise/checkpoints/      {'Model_G','Model_D'}, hair_editor.py:100-108
  model_trained/shape/054__.../shape_dir_used/*.pkl                                pickled tensors [16], hair_editor.py:110-119
  <ckpt dir>/latest_checkpoint     text file, first line = newest .ckpt file name (my_torchlib/utils.py:25-36)

`reference_checkpoints(root)` reads that tree into the weights dict `HipModels` takes; `write_reference_layout` writes the same
tree (tests; also documents the layout); `validate` is the `load_state_dict(strict=True)` check of the reference done up front,
on key names and shapes, against the architecture the library implements; `save_npz` / `load_npz` are a torch-free container
of the same tensors (one file, `<model>/<key>` entries).

    python -m ctrlhair_amd.checkpoints check   <root>            # key/shape report, exit code 1 on mismatch
    python -m ctrlhair_amd.checkpoints convert <root> out.npz    # pack everything into one npz
"""
import functools
import os
import pickle
from typing import Dict, List, Tuple

import numpy as np

EXPERIMENTS = {
    'color_texture': ('model_trained/color_texture', '045__color_texture_final'),          # color_texture_branch/config.py:18
    'color_encoder': ('model_trained/color_encoder', 'p004___pca_std'),                    # predictor_config.py:31
    'shape': ('model_trained/shape', '054__succeed__049__gan_fake_0.5_from_noise'),        # shape_branch/config.py:18
}
SEAN_FILE = 'external_model_params/sean_checkpoints/CelebA-HQ_pretrained/latest_net_G.pth'
BISENET_FILE = 'external_model_params/face_parsing_79999_iter.pth'
MODELS = ('sean', 'shape', 'color_gen', 'color_dis', 'color_rgb', 'bisenet')


def experiment_dir(root: str, which: str) -> str:
    """The experiment folder: the configured name, else the first folder starting with the config id (the reference selects
    configs by `experiment_name.startswith(config_id)`, config.py:44)."""
    base, name = EXPERIMENTS[which]
    d = os.path.join(root, base, name)
    if os.path.isdir(d):
        return d
    cid = name.split('_')[0]
    parent = os.path.join(root, base)
    if os.path.isdir(parent):
        for cand in sorted(os.listdir(parent)):
            if cand.startswith(cid) and os.path.isdir(os.path.join(parent, cand)):
                return os.path.join(parent, cand)
    return d


def load_checkpoint(ckpt_dir_or_file: str):
    """my_torchlib/utils.py:25-36 (newest file named on the first line of `latest_checkpoint`)."""
    import torch
    path = ckpt_dir_or_file
    if os.path.isdir(path):
        with open(os.path.join(path, 'latest_checkpoint')) as f:
            path = os.path.join(path, f.readline().strip())
    return torch.load(path, map_location='cpu')


def strip_module(sd: Dict) -> Dict:
    """DDP prefixes are dropped when the first key starts with 'module' (hair_editor.py:65-68)."""
    keys = list(sd)
    return {k[7:]: v for k, v in sd.items()} if keys and keys[0].startswith('module') else dict(sd)


def _load_dirs(d: str) -> List[np.ndarray]:
    if not os.path.isdir(d):
        return []
    out = []
    for name in sorted(os.listdir(d)):                      # hair_editor.py:84-90: sorted file names
        with open(os.path.join(d, name), 'rb') as f:
            v = pickle.load(f)
        out.append(np.asarray(v.detach().cpu().numpy() if hasattr(v, 'detach') else v, dtype=np.float32))
    return out


def reference_checkpoints(root: str = '.') -> Dict[str, object]:
    """-> {'sean','shape','color_gen','color_dis','color_rgb','bisenet': state dicts; 'texture_dirs','shape_dirs': lists;
    '_origin': 'reference' -- HairEditor / EditPipeline default to the exact-f32 path for a dict that carries this tag; save_npz / load_npz keep it,
    a caller that rebuilds the dict by hand must carry it over or pass f16x3 explicitly}."""
    import torch
    ct = load_checkpoint(os.path.join(experiment_dir(root, 'color_texture'), 'checkpoints'))
    sh = load_checkpoint(os.path.join(experiment_dir(root, 'shape'), 'checkpoints'))
    rgb = load_checkpoint(os.path.join(experiment_dir(root, 'color_encoder'), 'checkpoints'))
    return {'sean': torch.load(os.path.join(root, SEAN_FILE), map_location='cpu'),
            'shape': strip_module(sh['Model_G']), 'color_gen': strip_module(ct['Model_G']),
            'color_dis': strip_module(ct['Model_D']), 'color_rgb': strip_module(rgb['Predictor']),
            'bisenet': torch.load(os.path.join(root, BISENET_FILE), map_location='cpu'),
            'texture_dirs': _load_dirs(os.path.join(experiment_dir(root, 'color_texture'), 'texture_dir_used')),
            'shape_dirs': _load_dirs(os.path.join(experiment_dir(root, 'shape'), 'shape_dir_used')),
            '_origin': 'reference'}


def write_reference_layout(root: str, weights: Dict[str, dict], texture_dirs=(), shape_dirs=(), ddp_prefix: bool = True,
                           step: int = 1) -> None:
    """Write `weights` as the checkpoint tree the reference reads (torch.save files, `latest_checkpoint` lists, pickled
    direction tensors).  ddp_prefix adds the 'module.' prefix the released CtrlHair checkpoints carry."""
    import torch

    def t(sd, prefix=''):
        return {prefix + k: torch.as_tensor(np.asarray(v)) for k, v in sd.items()}

    def save_ckpt(which, obj):
        d = os.path.join(root, *EXPERIMENTS[which], 'checkpoints')
        os.makedirs(d, exist_ok=True)
        name = '%07d.ckpt' % step
        torch.save(obj, os.path.join(d, name))
        with open(os.path.join(d, 'latest_checkpoint'), 'w') as f:
            f.write(name + '\n')

    def save_dirs(which, sub, dirs):
        d = os.path.join(root, *EXPERIMENTS[which], sub)
        os.makedirs(d, exist_ok=True)
        for i, v in enumerate(dirs):
            with open(os.path.join(d, '%02d.pkl' % i), 'wb') as f:
                pickle.dump(torch.as_tensor(np.asarray(v, dtype=np.float32)), f)

    pre = 'module.' if ddp_prefix else ''
    for path, key in ((SEAN_FILE, 'sean'), (BISENET_FILE, 'bisenet')):
        os.makedirs(os.path.dirname(os.path.join(root, path)), exist_ok=True)
        torch.save(t(weights[key]), os.path.join(root, path))
    save_ckpt('color_texture', {'Model_G': t(weights['color_gen'], pre), 'Model_D': t(weights['color_dis'], pre), 'step': step})
    save_ckpt('color_encoder', {'Predictor': t(weights['color_rgb'], pre), 'step': step})
    save_ckpt('shape', {'Model_G': t(weights['shape'], pre), 'Model_D': {}, 'step': step})
    if len(texture_dirs):
        save_dirs('color_texture', 'texture_dir_used', texture_dirs)
    if len(shape_dirs):
        save_dirs('shape', 'shape_dir_used', shape_dirs)


def _shape(v) -> Tuple[int, ...]:
    return tuple(int(x) for x in (v.shape if hasattr(v, 'shape') else np.asarray(v).shape))


@functools.lru_cache(maxsize=4)
def expected(ngf: int = 64) -> Dict[str, Dict[str, Tuple[int, ...]]]:
    """Key -> shape tables of the architectures the library implements (ctrlhair_amd.procedural builds exactly the reference's
    state-dict keys; values are irrelevant here)."""
    from . import procedural as P
    from .hair_editor import procedural_weights
    # Only the shapes are wanted: the random-tensor makers of ctrlhair_amd.procedural are swapped for zero-stride views for
    # the duration of the call (no hundreds of MB of Gaussians, no power iterations); calibration tables are not applied.
    view = lambda shape: np.broadcast_to(np.float32(0), tuple(int(d) for d in shape))
    # P.MAKER_LOCK: the weight builders take the same (re-entrant) lock, so a concurrent procedural_weights() / sean_state_dict()
    # call in another thread can never see the patched makers
    with P.MAKER_LOCK:
      saved = (P._normal, P._xavier, P.power_iterate, P.load_calibration)
      try:
        P._normal = lambda seed, name, shape, std: view(shape)
        P._xavier = lambda seed, name, shape: view(shape)
        P.power_iterate = lambda w, seed, name, iters=60: (view((w.shape[0],)), view((int(np.prod(w.shape[1:])),)))
        P.load_calibration = lambda seed, ngf: {}              # (an empty table: nothing to apply, shapes unchanged)
        w = procedural_weights(0, ngf)
      finally:
        P._normal, P._xavier, P.power_iterate, P.load_calibration = saved
    return {m: {k: _shape(v) for k, v in w[m].items()} for m in MODELS}


def validate(weights: Dict[str, dict], ngf: int = 64, optional=('num_batches_tracked',)) -> List[str]:
    """strict=True semantics up front: every expected key present with the expected shape, no unknown keys (bookkeeping
    buffers such as BatchNorm's num_batches_tracked may be absent or present).  Returns a list of problems (empty = ok)."""
    problems = []
    exp = expected(ngf)
    for m in MODELS:
        if m not in weights:
            problems.append(f'{m}: missing model')
            continue
        have = {k: _shape(v) for k, v in weights[m].items()}
        for k, s in exp[m].items():
            if k not in have:
                if not any(k.endswith(o) for o in optional):
                    problems.append(f'{m}: missing key {k} {s}')
            elif have[k] != s:
                problems.append(f'{m}: shape of {k} is {have[k]}, expected {s}')
        for k in have:
            if k not in exp[m] and not any(k.endswith(o) for o in optional):
                problems.append(f'{m}: unexpected key {k} {have[k]}')
    return problems


def split_problems(problems: List[str]) -> Tuple[List[str], List[str]]:
    """(fatal, benign): missing keys and shape mismatches stop a load; keys the implemented architectures do not use (an extra
    buffer in a newer checkpoint) are only reported -- the reference would have needed strict loading per module for them."""
    benign = [p for p in problems if ': unexpected key ' in p]
    return [p for p in problems if p not in benign], benign


def save_npz(path: str, weights: Dict[str, object]) -> None:
    flat = {}
    for m in MODELS:
        for k, v in weights[m].items():
            a = v.detach().cpu().numpy() if hasattr(v, 'detach') else np.asarray(v)
            flat[f'{m}/{k}'] = a
    for name in ('texture_dirs', 'shape_dirs'):
        for i, v in enumerate(weights.get(name, ())):
            flat[f'{name}/{i:02d}'] = np.asarray(v, dtype=np.float32)
    if weights.get('_origin') is not None:      # the tag that decides the default arithmetic (exact f32 for a released checkpoint) survives the round trip
        flat['_origin'] = np.asarray(str(weights['_origin']))
    np.savez(path, **flat)


def load_npz(path: str) -> Dict[str, object]:
    d = np.load(path)
    out: Dict[str, object] = {m: {} for m in MODELS}
    out['texture_dirs'], out['shape_dirs'] = [], []
    for key in sorted(d.files):
        if key == '_origin':
            out['_origin'] = str(d[key])
            continue
        m, k = key.split('/', 1)
        if m in ('texture_dirs', 'shape_dirs'):
            out[m].append(d[key])
        else:
            out[m][k] = d[key]
    return out


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description='check / convert the reference checkpoint tree')
    ap.add_argument('job', choices=('check', 'convert'))
    ap.add_argument('root')
    ap.add_argument('out', nargs='?')
    args = ap.parse_args(argv)
    w = reference_checkpoints(args.root)
    problems = validate(w)
    for m in MODELS:
        n = sum(int(np.prod(_shape(v))) for v in w[m].values())
        print(f'{m:10s} {len(w[m]):5d} tensors {n / 1e6:9.2f} M values')
    print(f"texture_dirs {len(w['texture_dirs'])}, shape_dirs {len(w['shape_dirs'])}")
    for p in problems:
        print('PROBLEM', p)
    if args.job == 'convert':
        if not args.out:
            ap.error('convert needs an output path')
        save_npz(args.out, w)
        print('wrote', args.out)
    return 1 if problems else 0


if __name__ == '__main__':
    raise SystemExit(main())
