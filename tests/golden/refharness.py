"""Import harness for the *reference* (XuyangGuo/CtrlHair at /root/reference) on CPU.

Build-container only: /root/reference does not exist on the GPU box, and nothing under
ctrlhair_amd/, bench.py or the -m gpu tests imports this file.  It is used by
make_golden.py (to generate committed fixtures) and by tests/test_oracle_vs_reference.py
(skipped when the reference is absent).  Shims are the ones documented in SURVEY.md 8c /
Appendix A: stub modules for import-only dependencies, Tensor.cuda -> identity, no
model-zoo download.  No reference source is copied; modules are imported where they lie.
"""
import argparse
import os
import sys
import types

REF = '/root/reference'


def available() -> bool:
    return os.path.isdir(os.path.join(REF, 'sean_codes'))


_booted = False


def boot():
    global _booted
    if _booted:
        return
    if not available():
        raise RuntimeError('reference not present')
    import torch
    import torch.utils.model_zoo as mz
    sys.path.insert(0, REF)
    for n in ['cv2', 'torchvision', 'torchvision.models', 'torchvision.transforms', 'dlib', 'tensorboardX']:
        sys.modules.setdefault(n, types.ModuleType(n))
    torch.Tensor.cuda = lambda self, *a, **k: self       # normalization.py:111 hard-codes .cuda()
    mz.load_url = lambda *a, **k: {}                     # resnet.py:83 downloads at construction
    _booted = True


def make_generator(ngf: int, S: int, status: str = 'UI_mode'):
    boot()
    import warnings
    warnings.filterwarnings('ignore')
    from sean_codes.models.networks.generator import SPADEGenerator
    opt = argparse.Namespace(ngf=ngf, semantic_nc=19, label_nc=19, norm_G='spectralspadesyncbatch3x3',
                             num_upsampling_layers='normal', crop_size=S, aspect_ratio=1.0, status=status)
    return SPADEGenerator(opt).eval()


class NoiseFeeder:
    """Replaces torch.randn while the reference generator runs so that the 18 draws of
    normalization.py:111 return our explicit planes (in execution order)."""

    def __init__(self, planes):
        self.planes = list(planes)
        self.i = 0

    def __enter__(self):
        import torch
        self._orig = torch.randn

        def fake(*shape, **kw):
            p = self.planes[self.i]
            self.i += 1
            shape = tuple(shape[0]) if len(shape) == 1 and not isinstance(shape[0], int) else tuple(shape)
            assert tuple(p.shape) + (1,) == shape, (p.shape, shape)
            return p.reshape(shape).clone()
        torch.randn = fake
        return self

    def __exit__(self, *a):
        import torch
        torch.randn = self._orig


def run_generator(sd_np, labels, codes, noise, ngf, ui_mode=False):
    """Reference SPADEGenerator forward.  ui_mode=True: literal UI_mode path with obj_dic (B must be 1,
    generator.py:72 + normalization.py:120-139).  Otherwise the batched 'else' branch (:141-153) with the
    Zencoder replaced by the given per-sample codes (SURVEY.md 8c shim 6)."""
    import torch
    from oracle import sean_oracle as O
    labels_t = torch.from_numpy(labels)
    B, S = labels.shape[0], labels.shape[-1]
    G = make_generator(ngf, S)
    G.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()}, strict=True)
    G.eval()
    seg = O.one_hot(labels_t)
    planes = O.split_noise(torch.from_numpy(noise), S, ngf)
    codes_t = torch.from_numpy(codes)
    with torch.no_grad(), NoiseFeeder(planes) as nf:
        if ui_mode:
            assert B == 1
            for m in G.modules():
                if hasattr(m, 'status'):
                    m.status = 'UI_mode'
            obj_dic = {str(j): {'ACE': codes_t[0, j]} for j in range(19)}
            out = G(seg, None, obj_dic=obj_dic)
        else:
            for m in G.modules():
                if hasattr(m, 'status'):
                    m.status = 'train'
            G.Zencoder.forward = lambda input, segmap: codes_t
            out = G(seg, torch.zeros(B, 3, S, S))
        assert nf.i == 18
    return out.numpy()


def run_zencoder(sd_np, img, labels):
    """Reference Zencoder (architecture.py:155-207) on CPU; sd_np: full generator state dict."""
    import torch
    from oracle import sean_oracle as O
    G = make_generator(16, img.shape[-1])        # Zencoder is independent of ngf; tiny G keeps construction cheap
    zsd = {k[len('Zencoder.'):]: torch.from_numpy(v) for k, v in sd_np.items() if k.startswith('Zencoder.')}
    G.Zencoder.load_state_dict(zsd, strict=True)
    with torch.no_grad():
        return G.Zencoder(torch.from_numpy(img), O.one_hot(torch.from_numpy(labels))).numpy()


class _Dict(dict):
    """Stand-in for addict.Dict (not installed): attribute access, missing key -> empty falsy _Dict, nested dicts
    wrapped (needed by shape_branch/config.py:10 and color_texture_branch/config.py:10)."""

    def __init__(self, *a, **k):
        super().__init__()
        for key, v in dict(*a, **k).items():
            self[key] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, _Dict):
            v = _Dict(v)
        super().__setitem__(k, v)

    def __getattr__(self, k):
        if k.startswith('__'):
            raise AttributeError(k)
        if k not in self:
            return _Dict()
        return self[k]

    def __setattr__(self, k, v):
        self[k] = v

    def __missing__(self, k):
        return _Dict()


def boot_cfg():
    boot()
    if 'addict' not in sys.modules:
        m = types.ModuleType('addict')
        m.Dict = _Dict
        sys.modules['addict'] = m
    tv = sys.modules['torchvision.transforms']
    for n in ('Compose', 'ToTensor', 'Normalize', 'Resize'):
        if not hasattr(tv, n):
            setattr(tv, n, lambda *a, **k: (lambda x: x))
    sys.modules['torchvision'].transforms = tv


def make_shape_generator():
    boot_cfg()
    import warnings
    warnings.filterwarnings('ignore')
    from shape_branch.config import cfg
    from shape_branch.model import Generator
    return Generator(cfg).eval()


def make_color_solver():
    boot_cfg()
    import warnings
    warnings.filterwarnings('ignore')
    import torch
    from color_texture_branch.config import cfg
    from color_texture_branch.solver import Solver
    return Solver(cfg, torch.device('cpu'), -1, training=False)


def make_bisenet():
    boot_cfg()
    from external_code.face_parsing.model import BiSeNet
    return BiSeNet(19).eval()
