"""Host-side mirrors of the reference's model objects for the rest of the CtrlHair path, each a thin shim over
the C ABI (no arithmetic here beyond slicing / concatenation):

  ShapeGenerator      <-> shape_branch/model.py::Generator            (solver_mask.gen, hair_editor.py:93-108)
  ColorTextureModels  <-> color_texture_branch/solver.py::Solver      (.gen / .dis / .rgb_model / edit_infer)
  FaceParsing         <-> external_code/face_parsing/my_parsing_util.py::FaceParsing

Method names, argument meaning and return types follow the reference so that ui/backend.py-style code runs
unchanged on top of them.
"""
from typing import Dict, Optional

import numpy as np
import torch

from . import hostutil as U
from . import lib as _lib

HAIR_IDX = 13   # global_value_utils.py:52


def _np(v):
    a = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
    return a if a.dtype in (np.float32, np.int64) else a.astype(np.float32)


def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


# ---- shape_branch/shape_util.py -------------------------------------------------------------------------------------

def mask_label_to_one_hot(img: torch.Tensor) -> torch.Tensor:
    """shape_util.py:6-14 (255 -> dropped class)."""
    img = img.clone()
    img[img == 255] = 19
    bs, _, h, w = img.shape
    oh = torch.zeros(bs, 20, h, w, dtype=torch.float32, device=img.device).scatter_(1, img.long(), 1.0)
    return oh[:, :-1]


def mask_one_hot_to_label(one_hot: torch.Tensor) -> torch.Tensor:
    """shape_util.py:17-20."""
    mask = torch.argmax(one_hot, dim=1)
    mask[one_hot.max(dim=1)[0] == 0] = 255
    return mask


def split_hair_face(mask: torch.Tensor):
    """shape_util.py:23-26."""
    return mask[:, [HAIR_IDX]], torch.cat([mask[:, :HAIR_IDX], mask[:, HAIR_IDX + 1:]], dim=1)


class ShapeGenerator:
    """shape_branch/model.py::Generator (inference methods :164-199) on the HIP library."""

    def __init__(self, handle: _lib.Handle, device: torch.device):
        self.handle, self.device = handle, device

    def load_state_dict(self, sd: Dict[str, object], max_batch: int = 4, f16x3: bool = True):
        """f16x3: the decoder's 3x3 convs from 4x4 resolution up on the split-operand f16 MFMA kernels (f32-class; default) or,
        False, every conv on the exact-f32 kernels (option shape.f16x3 of the library)."""
        self.handle.set_option('shape.f16x3', 1 if f16x3 else 0)
        for k, v in sd.items():
            if 'std_out_layer' in k:
                continue        # VAE std head: unused with testing=True (shape_branch/model.py:164-169)
            self.handle.load_tensor(_lib.MODEL_SHAPE, k, _np(v))
        self.handle.finalize(_lib.MODEL_SHAPE, max_batch, 256)
        return self

    # fast path used by Backend: label map in, both codes out
    def encode_labels(self, labels: torch.Tensor):
        assert labels.is_cuda and labels.dtype == torch.uint8 and labels.shape[-2:] == (256, 256)
        labels = labels.reshape(-1, 256, 256).contiguous()
        B = labels.shape[0]
        hair = torch.empty(B, 16, device=labels.device)
        face = torch.empty(B, 1024, device=labels.device)
        self.handle.call('ch_shape_encode', labels.data_ptr(), hair.data_ptr(), face.data_ptr(), B, _stream(labels.device))
        return hair, face

    def forward_hair_encoder(self, hair: torch.Tensor, testing: bool = False):
        """hair: one-hot [B,1,256,256] (shape_branch/model.py:164-169).  Only testing=True (VAE mean) is served."""
        if not testing:
            raise NotImplementedError('VAE resampling is a training path (shape_branch/model.py:110-113)')
        lab = torch.where(hair[:, 0] > 0.5, torch.tensor(HAIR_IDX, device=hair.device), torch.tensor(255, device=hair.device))
        lab = lab.to(torch.uint8).contiguous()
        B = lab.shape[0]
        out = torch.empty(B, 16, device=hair.device)
        self.handle.call('ch_shape_encode', lab.data_ptr(), out.data_ptr(), None, B, _stream(hair.device))
        return out

    def forward_face_encoder(self, face: torch.Tensor):
        """face: one-hot [B,18,256,256] (the 19 classes without hair), shape_branch/model.py:171-173."""
        mx, idx = face.max(dim=1)
        idx = torch.where(idx >= HAIR_IDX, idx + 1, idx)
        lab = torch.where(mx > 0.5, idx, torch.full_like(idx, 255)).to(torch.uint8).contiguous()
        B = lab.shape[0]
        out = torch.empty(B, 1024, device=face.device)
        self.handle.call('ch_shape_encode', lab.data_ptr(), None, out.data_ptr(), B, _stream(face.device))
        return out

    def _decode(self, hair_code, face_code, want_hair=False, want_face=False, want_labels=False, want_probs=False):
        face_code = face_code.float().contiguous()
        B, dev = face_code.shape[0], face_code.device
        hc = hair_code.float().contiguous() if hair_code is not None else None
        hl = torch.empty(B, 1, 256, 256, device=dev) if want_hair else None
        fl = torch.empty(B, 18, 256, 256, device=dev) if want_face else None
        lab = torch.empty(B, 256, 256, dtype=torch.uint8, device=dev) if (want_labels or want_probs) else None
        pr = torch.empty(B, 19, 256, 256, device=dev) if want_probs else None
        p = lambda t: t.data_ptr() if t is not None else None
        self.handle.call('ch_shape_decode', p(hc), face_code.data_ptr(), p(hl), p(fl), p(lab), p(pr), B, _stream(dev))
        return hl, fl, lab, pr

    def forward_hair_decoder(self, hair_code, face_code):
        return self._decode(hair_code, face_code, want_hair=True)[0]

    def forward_face_decoder(self, face_code):
        return self._decode(None, face_code, want_face=True)[1]

    def forward_decoder(self, hair_logit, face_logit):
        """softmax mask [B,19,256,256] (shape_branch/model.py:184-187)."""
        B, dev = face_logit.shape[0], face_logit.device
        hair_logit, face_logit = hair_logit.float().contiguous(), face_logit.float().contiguous()
        lab = torch.empty(B, 256, 256, dtype=torch.uint8, device=dev)
        pr = torch.empty(B, 19, 256, 256, device=dev)
        self.handle.call('ch_shape_combine', hair_logit.data_ptr(), face_logit.data_ptr(), lab.data_ptr(), pr.data_ptr(), B,
                         _stream(dev))
        return pr

    def forward_decode_by_code(self, hair_code, face_code):
        return self._decode(hair_code, face_code, want_probs=True)[3]

    def decode_labels(self, hair_code, face_code) -> torch.Tensor:
        """fast path: uint8 label map [B,256,256] == mask_one_hot_to_label(forward_decode_by_code(...))."""
        return self._decode(hair_code, face_code, want_labels=True)[2]


class _Callable:
    def __init__(self, fn):
        self._fn = fn

    def __call__(self, data):
        return self._fn(data)

    def eval(self):
        return self


class ColorTextureModels:
    """color_texture_branch/solver.py::Solver(training=False): .gen, .dis, .rgb_model callables + edit_infer."""

    def __init__(self, handle: _lib.Handle, device: torch.device):
        self.handle, self.device = handle, device
        self.gen = _Callable(self._gen)
        self.dis = _Callable(self._dis)
        self.rgb_model = _Callable(self._rgb)

    def load_state_dicts(self, gen: Dict, dis: Dict, rgb: Dict, max_batch: int = 16):
        for pre, sd in (('gen.', gen), ('dis.', dis), ('rgb.', rgb)):
            for k, v in sd.items():
                k = k[7:] if k.startswith('module.') else k      # hair_editor.py:65-68 strips DDP prefixes
                self.handle.load_tensor(_lib.MODEL_COLOR, pre + k, _np(v))
        self.handle.finalize(_lib.MODEL_COLOR, max_batch, 0)
        return self

    def _f(self, t):
        if not isinstance(t, torch.Tensor):
            t = torch.as_tensor(t)
        return t.to(self.device).float().contiguous()

    def _gen(self, data):      # EigenGenerator.forward, model_eigengan.py:62-84
        noise = self._f(data['noise'])
        cond = torch.cat([self._f(data['noise_curliness']), self._f(data['rgb_mean']), self._f(data['pca_std'])], dim=1).contiguous()
        B = noise.shape[0]
        code = torch.empty(B, 512, device=self.device)
        self.handle.call('ch_color_generate', noise.data_ptr(), cond.data_ptr(), code.data_ptr(), B, _stream(self.device))
        return {'code': code}

    def _dis(self, data):      # Discriminator.forward, model.py:108-127 (cfg 045 slices)
        code = self._f(data['code'])
        B = code.shape[0]
        out = torch.empty(B, 11, device=self.device)
        self.handle.call('ch_color_encode', code.data_ptr(), out.data_ptr(), B, _stream(self.device))
        return {'adv': out[:, 0:1], 'noise': out[:, 1:9], 'noise_curliness': out[:, 9:10]}     # slices only: a list index is an H2D copy

    def _rgb(self, data):      # Predictor.forward, predictor_model.py:32-41 (predict_dict: rgb_mean 3, pca_std 1)
        code = self._f(data['code'])
        B = code.shape[0]
        out = torch.empty(B, 4, device=self.device)
        self.handle.call('ch_color_predict', code.data_ptr(), out.data_ptr(), B, _stream(self.device))
        return {'rgb_mean': out[:, 0:3], 'pca_std': out[:, 3:4]}

    def edit_infer(self, hair_code, data):     # solver.py:78-83
        self.inner_code = self.dis({'code': hair_code})
        for ke in data:
            self.inner_code[ke] = data[ke]
        self.res = self.gen(self.inner_code)
        return self.res['code']


BISENET_TO_CELEBA = np.array([0, 1, 6, 7, 4, 5, 3, 8, 9, 15, 2, 10, 11, 12, 17, 16, 18, 13, 14], np.uint8)
_CELEBA_TO_BISENET = np.argsort(BISENET_TO_CELEBA).astype(np.uint8)
_MEAN = (0.485, 0.456, 0.406)     # my_parsing_util.py:27
_STD = (0.229, 0.224, 0.225)


class FaceParsing:
    """external_code/face_parsing/my_parsing_util.py::FaceParsing with the network on the HIP library.
    Unlike the reference's class-level lazy singleton (:24,38-44) an instance owns its handle; `parsing_img` and
    `swap_parsing_label_to_celeba_mask` keep the reference's signatures."""

    def __init__(self, handle: _lib.Handle, device: torch.device):
        self.handle, self.device = handle, device

    def load_state_dict(self, sd: Dict[str, object], max_batch: int = 8, max_size: int = 512, f16x3: bool = True):
        """f16x3: the stride-1 convs on the split-operand f16 MFMA kernels (f32-class; default) or, False, every conv on the
        exact-f32 kernels (option bisenet.f16x3 of the library)."""
        self.handle.set_option('bisenet.f16x3', 1 if f16x3 else 0)
        for k, v in sd.items():
            if k.startswith('conv_out16.') or k.startswith('conv_out32.'):
                continue     # auxiliary heads: outputs discarded at inference (my_parsing_util.py:45 takes [0])
            self.handle.load_tensor(_lib.MODEL_BISENET, k, _np(v))
        self.handle.finalize(_lib.MODEL_BISENET, max_batch, max_size)
        return self

    def parse_tensor(self, img: torch.Tensor, want_logits: bool = False):
        """img: cuda float32 [B,3,H,W] already normalised -> (labels uint8 [B,H,W] CelebA ids, logits or None)."""
        img = img.float().contiguous()
        B, _, H, W = img.shape
        lab = torch.empty(B, H, W, dtype=torch.uint8, device=img.device)
        lg = torch.empty(B, 19, H, W, device=img.device) if want_logits else None
        self.handle.call('ch_bisenet_parse', img.data_ptr(), lab.data_ptr(), lg.data_ptr() if want_logits else None, B, H, W,
                         _stream(img.device))
        return lab, lg

    def normalise(self, img_u8_hwc: np.ndarray) -> torch.Tensor:
        """ToTensor + Normalize (my_parsing_util.py:25-28) on device."""
        t = torch.from_numpy(np.array(img_u8_hwc, copy=True)).to(self.device).permute(2, 0, 1).float() / 255.0
        if getattr(self, '_norm', None) is None:          # (two tiny host-to-device copies per call otherwise)
            self._norm = (torch.tensor(_MEAN, device=self.device).view(3, 1, 1), torch.tensor(_STD, device=self.device).view(3, 1, 1))
        mean, std = self._norm
        return ((t - mean) / std)[None]

    def parsing_img(self, img, image_size: int = 512):
        """my_parsing_util.py:31-47: PIL bilinear resize to 512, normalise, net, argmax -> (parsing in BiSeNet ids,
        resized PIL image)."""
        from PIL import Image
        pil = img if isinstance(img, Image.Image) else Image.fromarray(np.asarray(img).astype('uint8'))
        image = pil.resize((image_size, image_size), Image.BILINEAR)
        lab, _ = self.parse_tensor(self.normalise(np.asarray(image)))
        parsing = _CELEBA_TO_BISENET[U.to_host(lab[0])]          # (pinned staging buffer: hostutil.to_host)
        return parsing, image

    @staticmethod
    def swap_parsing_label_to_celeba_mask(parsing):
        """my_parsing_util.py:50-54."""
        return BISENET_TO_CELEBA[np.asarray(parsing)]
