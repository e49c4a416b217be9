"""Mirror of /root/reference/hair_editor.py::HairEditor on the MI355X library: same method names, argument meaning
and return types (hair_editor.py:40-335), the networks replaced by C-ABI shims.

Differences that are deliberate and documented (SURVEY.md 7 'host-side dependencies'):
  * no argv parsing / checkpoint globbing in the constructor: weights come in as state dicts (`weights=`), either real
    checkpoints in the reference's formats or ctrlhair_amd.procedural ones;
  * cv2 / dlib / scipy are imported lazily and only by the CPU pre/post-processing helpers that need them;
  * `load_average_feature` reads the 19 median codes once (packed copy of
    sean_codes/styles_test/mean_style_code/median/*/ACE.npy) instead of re-globbing .npy files on every call.
"""
import os
from typing import Dict, Optional

import numpy as np
import torch

from . import hostutil as U
from . import lib as _lib
from .hostutil import HAIR_IDX, PARSING_LABEL_LIST

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data')


def change_status(model, new_status):
    """hair_editor.py:33-36 (a no-op for the HIP generator, which has no per-module status)."""
    for m in model.modules():
        if hasattr(m, 'status'):
            m.status = new_status


def procedural_weights(seed: int = 0, ngf: int = 64) -> Dict[str, dict]:
    from . import procedural as P
    cs = P.color_state_dicts(seed)
    return {'sean': P.sean_state_dict(seed, ngf), 'shape': P.shape_state_dict(seed), 'color_gen': cs['gen'],
            'color_dis': cs['dis'], 'color_rgb': cs['rgb'], 'bisenet': P.bisenet_state_dict(seed)}


from .checkpoints import reference_checkpoints      # noqa: E402,F401  (reference checkpoint tree reader)


class HipModels:
    """Builds every network of the path on one ch_handle (one GPU)."""

    def __init__(self, weights: Dict[str, dict], device: int = 0, img_size: int = 256, max_batch: int = 1, f16x3=None,
                 options: Optional[Dict[str, int]] = None):
        """f16x3: arithmetic of the SEAN generator / Zencoder convs (see SeanGenerator): True = split-operand f16 MFMA,
        f32-class results (default); False = exact-f32 MFMA; 2 = single-term f16 (reduced precision).
        options: extra ch_set_option pairs applied before the generator is finalised (e.g. {'sean.ahead': 8})."""
        from .models import ColorTextureModels, FaceParsing, ShapeGenerator
        from .sean.generator import SeanGenerator
        from .sean.pix2pix_model import Pix2PixModel
        if f16x3 is None:            # by the weights: exact f32 for a released checkpoint, f16x3 otherwise (is_released_checkpoint)
            f16x3 = not is_released_checkpoint(weights)
        self.generator = SeanGenerator(device, f16x3=f16x3, options=options).load_state_dict(weights['sean'], max_batch=max_batch,
                                                                                           max_size=img_size)
        h, dev = self.generator.handle, self.generator.device
        self.device = dev
        self.sean_model = Pix2PixModel(self.generator)
        self.solver_feature = ColorTextureModels(h, dev).load_state_dicts(weights['color_gen'], weights['color_dis'],
                                                                          weights['color_rgb'], max_batch=max(max_batch, 16))
        self.mask_generator = ShapeGenerator(h, dev).load_state_dict(weights['shape'], max_batch=max(max_batch, 1),
                                                                     f16x3=bool(f16x3))      # exact f32 everywhere when f16x3=False
        self.face_parsing = FaceParsing(h, dev).load_state_dict(weights['bisenet'], max_batch=max(max_batch, 1), max_size=512,
                                                                f16x3=bool(f16x3))
        from .blending import PoissonBlender
        self.blender = PoissonBlender(h, dev)       # blending step after the generator (Backend(blending=True))


def is_released_checkpoint(weights) -> bool:
    """True for 'reference', a checkpoint directory, or a dict made by checkpoints.reference_checkpoints() (tagged '_origin'):
    such weights default to the exact-f32 path everywhere (HairEditor, EditPipeline); procedural / untagged dicts to f16x3."""
    if isinstance(weights, str):
        return weights != 'procedural'
    return isinstance(weights, dict) and weights.get('_origin') == 'reference'


class HairEditor:
    """This is the basic module (hair_editor.py:40-43); ctrlhair_amd.ui.backend.Backend succeeds this class."""

    def __init__(self, load_feature_model=True, load_mask_model=True, *, weights='procedural', device: int = 0,
                 img_size: int = 256, models=None, texture_dirs=None, shape_dirs=None, max_batch: int = 1, f16x3=None,
                 cap_threads: bool = None):
        """f16x3: None = by the weights: True (split-operand f16 MFMA, f32-class, ~3x faster) for procedural weights, on which
        every parity test runs; False (exact-f32 MFMA, the reference's arithmetic) for a released checkpoint tree
        (weights='reference' or a directory) -- the split-operand path is tested on heavy-tailed synthetic weights
        (tests/test_hip_robust_weights.py) but has never seen the real checkpoints, which cannot be fetched here.  Pass
        True / False to choose explicitly.
        cap_threads: the constructor lowers torch's intra-op thread count to the container's CPU quota when the pool is larger (a pool
        larger than the CFS quota stalls every other edit for ~85 ms).  That is a PROCESS-WIDE setting the reference does not touch:
        pass False (or set CTRLHAIR_NO_THREAD_CAP=1) to leave the embedding application's thread pool alone."""
        if f16x3 is None:
            f16x3 = not is_released_checkpoint(weights)
        if cap_threads is None:
            cap_threads = os.environ.get('CTRLHAIR_NO_THREAD_CAP', '') in ('', '0')
        if cap_threads:
            U.cap_threads_to_cpu_quota()
        if models is None:
            if weights == 'procedural':
                weights = procedural_weights()
            elif weights == 'reference' or (isinstance(weights, str) and os.path.isdir(weights)):
                # the reference's checkpoint tree ('reference' = the current directory, like the reference itself)
                weights = reference_checkpoints('.' if weights == 'reference' else weights)
                from .checkpoints import split_problems, validate
                fcw = weights['sean'].get('fc.weight') if isinstance(weights.get('sean'), dict) else None
                ngf = int(fcw.shape[0]) // 16 if fcw is not None else 64      # generator width as the C library infers it
                # strict=True semantics before anything is uploaded; keys the architectures do not use are reported, not fatal
                problems, extras = split_problems(validate(weights, ngf=ngf))
                if extras:
                    import warnings
                    warnings.warn('checkpoint tree has keys the implemented architectures do not use (ignored):\n  ' + '\n  '.join(extras[:20]))
                    for msg in extras:
                        mname, key = msg.split(': unexpected key ')[0], msg.split(': unexpected key ')[1].split(' ')[0]
                        weights[mname].pop(key, None)
                if problems:
                    raise RuntimeError('checkpoint tree does not match the implemented architectures:\n  ' + '\n  '.join(problems[:20]))
                if texture_dirs is None and weights.get('texture_dirs'):
                    texture_dirs = weights['texture_dirs']          # hair_editor.py:82-91
                if shape_dirs is None and weights.get('shape_dirs'):
                    shape_dirs = weights['shape_dirs']              # hair_editor.py:110-119
            models = HipModels(weights, device=device, img_size=img_size, max_batch=max_batch, f16x3=f16x3)
        self.models = models
        self.sean_model = models.sean_model
        self.img_size = img_size                     # hair_editor.py:50 (256 in the reference)
        self.device = models.device
        self.face_parsing = models.face_parsing
        if load_feature_model:
            self.solver_feature = models.solver_feature
            self.feature_encoder = self.solver_feature.dis
            self.feature_generator = self.solver_feature.gen
            self.feature_rgb_predictor = self.solver_feature.rgb_model
            dirs = texture_dirs if texture_dirs is not None else U.seeded_directions(2, 8, seed=45)
            self.texture_dirs = [torch.as_tensor(d).float().to(self.device) for d in dirs]
        if load_mask_model:
            self.mask_generator = models.mask_generator
            dirs = shape_dirs if shape_dirs is not None else U.seeded_directions(4, 16, seed=54)
            self.shape_dirs = [torch.as_tensor(d).float().to(self.device) for d in dirs]
        self._median = None

    # ---- pre-processing (hair_editor.py:121-128) --------------------------------------------------------------
    def preprocess_img(self, img):
        img = U.resize_bilinear(np.asarray(img).astype('uint8'), (self.img_size, self.img_size))
        return (np.transpose(img, [2, 0, 1]) / 127.5 - 1.0)[None, ...]

    def preprocess_mask(self, mask_img):
        mask_img = U.resize_nearest(np.asarray(mask_img).astype('uint8'), (self.img_size, self.img_size))
        return mask_img[None, None, :, :]

    def load_average_feature(self):
        """hair_editor.py:130-147: {str(i): {'ACE': tensor[512]}} of the per-category median style codes."""
        if self._median is None:
            self._median = torch.from_numpy(np.load(os.path.join(_DATA, 'mean_style_code.npz'))['median']).to(self.device)
        return {str(i): {'ACE': self._median[i]} for i in range(19)}

    # ---- networks --------------------------------------------------------------------------------------------
    def get_code(self, hair_img, hair_parsing):
        """hair_editor.py:149-157 -> style codes [1,19,512]."""
        data = {'label': torch.as_tensor(np.asarray(hair_parsing), dtype=torch.float32), 'instance': torch.tensor(0),
                'image': torch.as_tensor(np.asarray(hair_img), dtype=torch.float32), 'path': ['temp/temp_npy']}
        change_status(self.sean_model, 'test')
        return self.sean_model(data, mode='style_code')

    def _obj_dic(self, code):
        obj_dic = self.load_average_feature()
        present = (code[0] != 0).any(dim=1).tolist()  # ONE reduction + read-back instead of the reference's 19 `torch.all(...)` tests
        for idx in range(19):
            if present[idx]:                          # all-zero row (absent region) -> keep the median code, :165-168
                obj_dic[str(idx)]['ACE'] = code[0, idx]
        return obj_dic

    def gen_img(self, code, parsing, noise=None):
        """hair_editor.py:159-179 -> generated image [3,S,S] in [-1,1]."""
        if not isinstance(code, torch.Tensor):
            code = torch.tensor(code)
        code = code.to(self.device)
        # hair_editor.py:162-176 builds obj_dic from the median codes and the non-zero rows of code[0] (19 `torch.all` read-backs) and
        # the model stacks it again: the same [19, 512] tensor from ONE device-side select, no read-back (the shim takes it as
        # data['codes']; _obj_dic() stays for callers that want the dictionary)
        data = {'label': torch.as_tensor(np.asarray(parsing) if not isinstance(parsing, torch.Tensor) else parsing,
                                         dtype=torch.float32),
                'instance': torch.tensor(0), 'image': None}
        if getattr(self.sean_model, 'accepts_codes', False):
            if self._median is None:
                self.load_average_feature()
            present = (code[0] != 0).any(dim=1, keepdim=True)
            data['obj_dic'] = None
            data['codes'] = torch.where(present, code[0].float(), self._median.to(code.device))
        else:                                             # a model object with the reference's interface only
            data['obj_dic'] = self._obj_dic(code)
        if noise is not None:
            data['noise'] = noise
        change_status(self.sean_model, 'UI_mode')
        return self.sean_model(data, mode='UI_mode')[0]

    def gen_imgs(self, codes, parsings, noise=None, seed: int = 0):
        """Batched gen_img (the reference only offers batch 1, hair_editor.py:159-179, and loops
        over it in shape_branch/validation_in_train.py:114-121 / color_texture_branch/solver.py:270-299).
        codes [B,19,512], parsings [B,S,S] or [B,1,S,S] label maps -> images [B,3,S,S].  Per sample, all-zero code rows
        (absent regions) fall back to the median style code exactly like gen_img (:165-168)."""
        codes = torch.as_tensor(codes).to(self.device).float()
        lab = torch.as_tensor(np.asarray(parsings) if not isinstance(parsings, torch.Tensor) else parsings)
        lab = lab.reshape(lab.shape[0], lab.shape[-2], lab.shape[-1]).to(self.device).to(torch.uint8).contiguous()
        if self._median is None:
            self.load_average_feature()
        zero = (codes == 0).all(dim=2, keepdim=True)
        codes = torch.where(zero, self._median[None].expand_as(codes), codes).contiguous()
        return self.models.generator.generate(lab, codes, noise, seed=seed)

    def generate_by_sean(self, face_img_code, hair_code, target_seg, noise=None):
        """hair_editor.py:181-206 (face_img_code [19,512], hair_code [512])."""
        obj_dic = self.load_average_feature()
        for idx in range(19):
            cur_code = hair_code if idx == HAIR_IDX else face_img_code[idx]
            if not torch.all(face_img_code == 0):
                obj_dic[str(idx)]['ACE'] = cur_code
        data = {'label': torch.as_tensor(np.asarray(target_seg), dtype=torch.float32), 'instance': torch.tensor(0),
                'obj_dic': obj_dic, 'image': None}
        if noise is not None:
            data['noise'] = noise
        change_status(self.sean_model, 'UI_mode')
        return self.sean_model(data, mode='UI_mode')[0]

    def generate_instance_transfer_img(self, face_img, face_parsing, hair_img, hair_parsing, target_seg, edit_data=None,
                                       temp_path='temp'):
        """hair_editor.py:208-231."""
        face_img_code = self.get_code(face_img, face_parsing)
        hair_img_code = face_img_code if hair_img is None else self.get_code(hair_img, hair_parsing)
        hair_code = hair_img_code[0, HAIR_IDX]
        if edit_data is not None:
            hair_code = self.solver_feature.edit_infer(hair_code[None, ...], edit_data)[0]
        return self.generate_by_sean(face_img_code[0], hair_code, target_seg)

    def get_mask(self, img_rgb):
        """hair_editor.py:331-335: BiSeNet parse @512 -> CelebAMask ids -> nearest resize to img_size."""
        parsing, _ = self.face_parsing.parsing_img(img_rgb)
        parsing = self.face_parsing.swap_parsing_label_to_celeba_mask(parsing)
        return U.resize_nearest(parsing.astype('uint8'), (self.img_size, self.img_size))

    def get_hair_color(self, img):
        """hair_editor.py:233-243 (needs cv2 for the 19x19 elliptical erosion)."""
        cv2 = U._cv2()
        if cv2 is None:
            raise RuntimeError('get_hair_color needs cv2 (elliptical erosion), which is not installed')
        parsing = self.get_mask_fullres(img, 1024)
        img = cv2.resize(np.asarray(img).astype('uint8'), (1024, 1024))
        hair_mask = cv2.erode((parsing == HAIR_IDX).astype('uint8'),
                              cv2.getStructuringElement(cv2.MORPH_ELLIPSE, ksize=(19, 19)), iterations=1)
        return img[hair_mask.astype('bool')].mean(axis=0)

    def get_mask_fullres(self, img_rgb, size):
        parsing, _ = self.face_parsing.parsing_img(img_rgb)
        return U.resize_nearest(self.face_parsing.swap_parsing_label_to_celeba_mask(parsing).astype('uint8'), (size, size))

    # ---- post-processing (hair_editor.py:257-310) --------------------------------------------------------------
    def postprocess_blending(self, face_img, res_img, face_parsing, target_parsing, verbose_print=False, blending=True,
                             blender=None):
        def from_tensor_order_to_cv2(tensor_img, is_mask=False):
            if isinstance(tensor_img, torch.Tensor) and tensor_img.is_cuda and not is_mask and tensor_img.dtype == torch.float32 and \
                    tensor_img.shape[-3:-2] == (3,) and tensor_img.dim() in (3, 4) and tensor_img.shape[-1] > 3:
                # the same float32 multiply, add and truncation as below, on the device: a quarter of the bytes cross the bus and the
                # host does no arithmetic (the caller's .astype('uint8') is then a no-op)
                t = tensor_img[0] if tensor_img.dim() == 4 else tensor_img
                return U.to_host((t * 127.5 + 127.5).to(torch.uint8).permute(1, 2, 0).contiguous())
            if isinstance(tensor_img, torch.Tensor):
                tensor_img = U.to_host(tensor_img)
            if len(tensor_img.shape) == 4:
                tensor_img = tensor_img[0]
            if len(tensor_img.shape) == 2:
                tensor_img = tensor_img[None, ...]
            if tensor_img.shape[2] <= 3:
                return tensor_img
            res = np.transpose(tensor_img, [1, 2, 0])
            if not is_mask:
                res = res * 127.5 + 127.5
            return res

        res_img = from_tensor_order_to_cv2(res_img).astype('uint8')
        if not blending:
            return res_img, None
        target_parsing = from_tensor_order_to_cv2(target_parsing, is_mask=True)
        face_img = from_tensor_order_to_cv2(face_img).astype('uint8')
        face_parsing = from_tensor_order_to_cv2(face_parsing, is_mask=True)
        if blender is None:
            blender = getattr(self.models, 'blender', None)
        from .blending import PoissonBlender
        if isinstance(blender, PoissonBlender):
            # HIP path: mask construction (hair_editor.py:297-305) and the Poisson solve
            # (poisson_blending.py:29-87) both on the library; no cv2, no scipy
            def at_image_size(m):        # [H,W,1] label map -> [H,W] at the image's size (512 mode: masks live at 256)
                m = np.asarray(m).reshape(m.shape[0], m.shape[1]).astype('uint8')
                return m if m.shape == res_img.shape[:2] else U.resize_nearest(m, res_img.shape[:2])
            res_mask_dilated = blender.blend_mask(at_image_size(target_parsing), at_image_size(face_parsing))
            if face_img.shape[:2] != res_img.shape[:2]:
                face_img = U.resize_bilinear(face_img, res_img.shape[:2])
            out = blender(face_img, res_img, 1 - res_mask_dilated, with_gamma=True)
            return out, U.to_host(res_mask_dilated)[..., None]
        # injected reference blender (`blender=poisson_blending.poisson_blending`): needs cv2 for the dilations
        cv2 = U._cv2()
        if blender is None or cv2 is None:
            raise RuntimeError('blending=True needs the HIP models (HipModels.blender) or cv2 plus an injected Poisson '
                               'blender; construct Backend(..., blending=False) or pass blender=')
        res_mask = np.logical_or(target_parsing == HAIR_IDX, face_parsing == HAIR_IDX).astype('uint8')
        k13 = cv2.getStructuringElement(cv2.MORPH_ELLIPSE, ksize=(13, 13))
        k5 = cv2.getStructuringElement(cv2.MORPH_ELLIPSE, ksize=(5, 5))
        d13 = cv2.dilate(res_mask, k13, iterations=1)[..., None]
        d5 = cv2.dilate(res_mask, k5, iterations=1)[..., None]
        bg_mask = (target_parsing == PARSING_LABEL_LIST.index('background'))
        res_mask_dilated = d13 * (1 - bg_mask) + d5 * bg_mask
        return blender(face_img, res_img, 1 - res_mask_dilated, with_gamma=True), res_mask_dilated

    def crop_face(self, img_rgb, save_path=None):
        """hair_editor.py:312-329: dlib landmark alignment -- CPU pre-processing, out of scope."""
        raise NotImplementedError('crop_face needs dlib landmark models (external_code/crop.py); crop offline')
