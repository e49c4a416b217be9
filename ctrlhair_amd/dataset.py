"""Dataset-scale mask extraction and SEAN style-code encoding.

The reference walks a directory one image at a time (dataset_scripts/script_get_mask.py:55-71 -> `label/<name>.png`,
dataset_scripts/script_get_sean_code.py:40-62 -> `sean_code/<dataset>___<name>.pkl` merged into `sean_code_dict.pkl` by
dataset_scripts/utils.py:14-21).  Here the same BiSeNet / Zencoder kernels run over batches, and the file list is
sharded over ranks (one process per GPU; no collective on the data path -- rank r owns files r, r+W, r+2W, ...).

On-disk formats are the reference's:
  label/<name>.png                      8-bit single-channel PNG, CelebAMask-HQ ids, 512x512 (script_get_mask.py:44-50)
  sean_code/<dataset>___<name>.pkl      pickle of float32 [19,512] (script_get_sean_code.py:56-62)
  sean_code_dict.pkl                    pickle of {'<dataset>___<name>': float32 [19,512]} (utils.py:14-21)

    python -m ctrlhair_amd.dataset masks  <root> <dataset> [--batch 16]
    python -m ctrlhair_amd.dataset codes  <root> <dataset> [--batch 16]
    (under torch.distributed.run for several GPUs; RANK / WORLD_SIZE / LOCAL_RANK are read from the environment)
"""
import os
import pickle
from typing import Dict, Iterable, List, Sequence

import numpy as np

IMG_EXT = ('.png', '.jpg', '.jpeg', '.bmp')


# ---- host logic (no GPU) ---------------------------------------------------------------------------------------------
def list_images(img_dir: str) -> List[str]:
    """Sorted image file names of a directory (script_get_sean_code.py:30-35 sorts the joined list)."""
    return sorted(f for f in os.listdir(img_dir) if f.lower().endswith(IMG_EXT))


def shard(items: Sequence, rank: int, world: int) -> List:
    """Round-robin shard of a sorted list: equal sizes up to one item, independent of the batch size."""
    if not 0 <= rank < world:
        raise ValueError(f'rank {rank} outside world {world}')
    return list(items[rank::world])


def batches(items: Sequence, n: int) -> Iterable[List]:
    for i in range(0, len(items), n):
        yield list(items[i:i + n])


def code_key(dataset: str, file_name: str) -> str:
    """'%s___%s' % (dataset_name, base_name[:-4])  (script_get_sean_code.py:60)."""
    return '%s___%s' % (dataset, os.path.splitext(file_name)[0])


def merge_pickle_dir_to_dict(dir_name: str, target_path: str) -> Dict[str, np.ndarray]:
    """dataset_scripts/utils.py:14-21."""
    res = {}
    for f_name in sorted(os.listdir(dir_name)):
        if f_name.endswith('.pkl'):
            with open(os.path.join(dir_name, f_name), 'rb') as f:
                res[f_name[:-4]] = pickle.load(f)
    with open(target_path, 'wb') as f:
        pickle.dump(res, f)
    return res


def read_rgb(path: str) -> np.ndarray:
    from PIL import Image
    return np.asarray(Image.open(path).convert('RGB'))


def read_gray(path: str) -> np.ndarray:
    from PIL import Image
    return np.asarray(Image.open(path).convert('L'))


def write_label_png(path: str, label: np.ndarray) -> None:
    from PIL import Image
    Image.fromarray(np.asarray(label, dtype=np.uint8), mode='L').save(path)


# ---- batched drivers (HairEditor on the HIP library) -----------------------------------------------------------------
def extract_masks(editor, img_dir: str, label_dir: str, batch: int = 16, rank: int = 0, world: int = 1,
                  parse_size: int = 512) -> List[str]:
    """BiSeNet-parse every image of `img_dir` (this rank's shard) and write `label_dir/<name>.png`.
    Per image identical to FaceParsing.parsing_img + swap_parsing_label_to_celeba_mask (my_parsing_util.py:31-54)."""
    import torch
    from PIL import Image
    os.makedirs(label_dir, exist_ok=True)
    fp = editor.face_parsing
    done = []
    for names in batches(shard(list_images(img_dir), rank, world), batch):
        x = torch.cat([fp.normalise(np.asarray(Image.fromarray(read_rgb(os.path.join(img_dir, n)))
                                               .resize((parse_size, parse_size), Image.BILINEAR))) for n in names], dim=0)
        labels, _ = fp.parse_tensor(x)                       # uint8 [B,512,512], CelebAMask-HQ ids
        labels = labels.cpu().numpy()
        for n, lab in zip(names, labels):
            write_label_png(os.path.join(label_dir, os.path.splitext(n)[0] + '.png'), lab)
            done.append(n)
    return done


def encode_sean_codes(editor, img_dir: str, label_dir: str, code_dir: str, dataset: str, batch: int = 16, rank: int = 0,
                      world: int = 1) -> Dict[str, np.ndarray]:
    """Zencoder style codes of every (image, label) pair of this rank's shard -> `code_dir/<dataset>___<name>.pkl`.
    Per image identical to HairEditor.get_code(preprocess_img(img), preprocess_mask(label)) (hair_editor.py:121-157)."""
    import torch
    os.makedirs(code_dir, exist_ok=True)
    gen = editor.models.generator
    out = {}
    for names in batches(shard(list_images(img_dir), rank, world), batch):
        imgs = np.concatenate([editor.preprocess_img(read_rgb(os.path.join(img_dir, n))) for n in names], axis=0)
        labs = np.concatenate([editor.preprocess_mask(read_gray(os.path.join(label_dir, os.path.splitext(n)[0] + '.png')))[0]
                               for n in names], axis=0)
        codes = gen.encode(torch.from_numpy(imgs.astype(np.float32)).to(editor.device),
                           torch.from_numpy(labs.astype(np.uint8)).to(editor.device)).cpu().numpy()
        for n, c in zip(names, codes):
            key = code_key(dataset, n)
            with open(os.path.join(code_dir, key + '.pkl'), 'wb') as f:
                pickle.dump(c, f)
            out[key] = c
    return out


def _dist_env():
    return int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('LOCAL_RANK', '0'))


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    ap.add_argument('job', choices=('masks', 'codes'))
    ap.add_argument('root')
    ap.add_argument('dataset')
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--img-size', type=int, default=256)
    ap.add_argument('--weights', default='reference', help="'reference' (the Google-Drive checkpoints under the reference's "
                                                           "paths) or 'procedural'")
    args = ap.parse_args(argv)
    import torch
    from .hair_editor import HairEditor
    rank, world, local = _dist_env()
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('gloo', rank=rank, world_size=world)      # only a barrier before the merge
    torch.cuda.set_device(local)
    he = HairEditor(True, True, weights=args.weights, device=local, img_size=args.img_size, max_batch=args.batch)
    base = os.path.join(args.root, args.dataset)
    if args.job == 'masks':
        n = len(extract_masks(he, os.path.join(base, 'images_256'), os.path.join(base, 'label'), args.batch, rank, world))
    else:
        code_dir = os.path.join(args.root, 'hair_info_all_dataset', 'sean_code')
        n = len(encode_sean_codes(he, os.path.join(base, 'images_256'), os.path.join(base, 'label'), code_dir, args.dataset,
                                  args.batch, rank, world))
        if dist is not None:
            dist.barrier()
        if rank == 0:
            merge_pickle_dir_to_dict(code_dir, os.path.join(args.root, 'sean_code_dict.pkl'))
    print(f'rank {rank}/{world}: {n} files')
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
