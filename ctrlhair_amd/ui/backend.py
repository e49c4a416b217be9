"""Mirror of /root/reference/ui/backend.py::Backend (the editing API CtrlHair scripts call) on the MI355X library.
Every public method of ui/backend.py:67-462 is present with the same name, argument order and return types; line
references are given per method.  Shape transfer by photo (`transfer_latent_representation('shape')`) needs the
reference's ARAP warping tool chain (wrap_codes/, dlib) and accepts an injected `warper` instead.
"""
import contextlib
import os

import numpy as np
import torch

from .. import hostutil as U
from ..hair_editor import HairEditor
from ..hostutil import HAIR_IDX, TEMP_FOLDER, mask_to_rgb
from ..models import mask_label_to_one_hot, mask_one_hot_to_label, split_hair_face  # noqa: F401 (API parity)


def generate_noise(bs, dim, label=None):
    """my_torchlib/train_utils.py:44-51."""
    noise = torch.randn((bs, dim))
    if label is not None:
        noise = (noise.abs() * label).float()
    return noise


class LatentRepresentation:                       # ui/backend.py:31-37
    def __init__(self):
        self.color = None
        self.curliness = None
        self.shape = None
        self.texture = None
        self.face = None


class Backend(HairEditor):
    def __init__(self, maximum_value_fe, blending=True, temp_path=os.path.join(TEMP_FOLDER, 'demo_output'), *,
                 hsv_table=None, warper=None, blender=None, **editor_kwargs):
        """ui/backend.py:45-65.  Keyword-only extras: hsv_table (DistTranslation data), warper (shape-transfer warp
        function), blender (Poisson blender), and HairEditor's weights/device/img_size/models."""
        super().__init__(True, True, **editor_kwargs)
        self.target_img = None
        self.input_img = None
        self.target_mask = None
        self.input_mask = None
        self.cur_latent = None
        self.target_latent = None
        self.cur_mask = None
        self.input_sean_code = None
        self.target_size = 256
        self.maximum_value_fe = maximum_value_fe
        self.temp_path = temp_path
        self.blending = blending
        self.dist_translation = U.DistTranslation(hsv_table)
        self.warper = warper
        self.blender = blender
        self.noise = None          # optional pinned noise planes for repeatable output() (tests / A-B comparisons)

    def _side_stream(self):
        """The side stream of parse_img (None when overlap is off: Backend.overlap = False, or without a GPU)."""
        if not getattr(self, 'overlap', True) or not torch.cuda.is_available():
            return None
        if getattr(self, '_side', None) is None:
            self._side = torch.cuda.Stream(self.device)
        return self._side

    # ---- analysis (ui/backend.py:67-106) ---------------------------------------------------------------------
    def _mask_for_sean(self, mask256):
        """The shape branch is fixed at 256x256 (shape_branch/model.py:85-89); for img_size 512 the label map is
        nearest-upsampled x2 before the SEAN generator (SURVEY.md 8d Config 3)."""
        if self.img_size == mask256.shape[-1]:
            return mask256
        return U.resize_nearest(mask256.astype('uint8'), (self.img_size, self.img_size))

    def parse_img(self, img_rgb, target_img=False):
        img_ts = U.resize_bilinear(np.asarray(img_rgb), (self.target_size, self.target_size))
        # Batch-1 latency: the networks of this call are chains of small kernels that leave most of the device idle, and only some of them
        # depend on each other -- the Zencoder's convolutions need the image only (the parsing enters its region means at the end:
        # ch_sean_encode_features / _regions), the shape branch (two encoders, two decoders: :81-90) the parsing only.  With a side stream
        # the Zencoder's convolutions run underneath BiSeNet and the shape branch underneath the region means and the colour MLPs (same
        # kernels, same results; Backend.overlap = False keeps everything on one stream).
        side = self._side_stream() if self.device.type == 'cuda' else None
        main = torch.cuda.current_stream(self.device) if side is not None else None
        img_pre = self.preprocess_img(img_rgb)
        gen = self.models.generator if side is not None and hasattr(self, 'models') and hasattr(self.models, 'generator') else None
        if gen is not None and img_pre.shape[0] <= gen.max_batch:
            img_dev = torch.as_tensor(img_pre).to(self.device).float()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                gen.encode_features(img_dev)
        else:
            gen = None
        lr = LatentRepresentation()
        fp = getattr(self, 'face_parsing', None)
        mask_dev = None
        if side is not None and self.img_size in (256, 512) and hasattr(fp, 'parse_tensor') and hasattr(fp, 'normalise'):
            # get_mask (hair_editor.py:331-335) with the label map kept on the device: BiSeNet's ids come back as CelebAMask ids, the
            # nearest resize 512 -> img_size (-> 256 for the shape branch) is a strided view (cv2.INTER_NEAREST: floor(dst * 2)), and the
            # networks below start from it without a round trip through the host; the host copy the caller gets is taken at the end.
            from PIL import Image
            pil = img_rgb if isinstance(img_rgb, Image.Image) else Image.fromarray(np.asarray(img_rgb).astype('uint8'))
            lab512, _ = fp.parse_tensor(fp.normalise(np.asarray(pil.resize((512, 512), Image.BILINEAR))))
            mask_dev = lab512[0, ::512 // self.img_size, ::512 // self.img_size].contiguous()
            mask_tensor = (mask_dev[::self.img_size // 256, ::self.img_size // 256].contiguous() if self.img_size != 256 else mask_dev)[None]
            mask = mask_batch = None
        else:
            mask = self.get_mask(img_rgb)                                     # [img_size, img_size] CelebA ids
            mask256 = U.resize_nearest(mask, (256, 256)) if mask.shape[0] != 256 else mask
            mask_batch = self.preprocess_mask(mask)
            mask_tensor = torch.tensor(mask256[None], dtype=torch.uint8, device=self.device)
        if side is not None:
            main.wait_stream(side)                # (the Zencoder's feature map)
            side.wait_stream(main)
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            lr.shape, lr.face = self.mask_generator.encode_labels(mask_tensor)   # == one-hot, split, two encoders (:81-86)
            decoded_dev = self.mask_generator.decode_labels(lr.shape, lr.face)   # :87-90
        # hair appearance: Zencoder code of the hair region -> colour statistics and texture / curliness latents (:93-105)
        if gen is not None and mask_dev is not None:
            codes = gen.encode_regions(mask_dev[None])
        elif gen is not None:
            codes = gen.encode_regions(torch.as_tensor(np.asarray(mask_batch)).to(self.device).to(torch.uint8)[:, 0].contiguous())
        else:
            if mask_batch is None:
                mask_batch = self.preprocess_mask(U.to_host(mask_dev))
            codes = self.get_code(img_pre, mask_batch)
        hair = codes[:, HAIR_IDX]
        stats = self.feature_rgb_predictor({'code': hair})
        rgb_u8 = np.clip(U.to_host(stats['rgb_mean']), 0, 255).astype('uint8')
        lr.color = {'hsv': torch.tensor(U.rgb_to_hsv_u8(rgb_u8[None, ...])).to(self.device)[0], 'pca_std': stats['pca_std']}
        latents = self.feature_encoder({'code': hair})
        lr.curliness, lr.texture = latents['noise_curliness'], latents['noise']
        if side is not None:
            main.wait_stream(side)
        decoded = U.to_host(decoded_dev)[0]
        if mask is None:
            mask = U.to_host(mask_dev)
        return img_ts, decoded, lr, mask, codes, hair

    def _convert_u8(self, t, fn):
        arr = U.to_host(t).astype('uint8')            # the reference truncates to uint8 before cv2.cvtColor
        return torch.tensor(fn(arr[None, ...])).to(self.device)[0]

    def tensor_hsv_to_rgb(self, hsv):             # :108-115
        return self._convert_u8(hsv, U.hsv_to_rgb_u8)

    def tensor_rgb_to_hsv(self, rgb):             # :117-125
        return self._convert_u8(rgb, U.rgb_to_hsv_u8)

    def set_input_img(self, img_rgb):             # :127-135
        parsed = self.parse_img(img_rgb)
        (self.input_img, self.cur_mask, self.cur_latent, self.input_mask, self.input_sean_code,
         self.input_hair_feature) = parsed
        return self.input_img, mask_to_rgb(self.cur_mask, draw_type=1)

    def set_target_img(self, img_rgb):            # :137-145
        parsed = self.parse_img(img_rgb)
        self.target_img, self.target_latent, self.target_mask, self.target_hair_feature = (parsed[0], parsed[2], parsed[3],
                                                                                          parsed[5])
        return self.target_img, mask_to_rgb(self.target_mask, draw_type=1)

    # ---- render (ui/backend.py:147-175) ----------------------------------------------------------------------
    def output(self, target_latent=None, feature=None):
        lat = self.cur_latent if target_latent is None else target_latent
        mask = self.cur_mask if target_latent is None else self.refresh_cur_mask(lat)[0]
        if feature is None:
            # colour: an explicit mean RGB wins -- and, as in the reference (:157), it is read from self.target_latent
            rgb = self.target_latent.color['rgb_mean'] if 'rgb_mean' in lat.color else self.tensor_hsv_to_rgb(lat.color['hsv'])
            feature = self.feature_generator({'noise': lat.texture, 'noise_curliness': lat.curliness, 'rgb_mean': rgb,
                                              'pca_std': lat.color['pca_std']})['code']
        self.input_sean_code[:, HAIR_IDX] = feature                      # in-place, like :170
        rendered = self.gen_img(self.input_sean_code, self._mask_for_sean(mask)[None, None, ...], noise=self.noise)
        return self.postprocess_blending(self.input_img, rendered, self.input_mask, mask, blending=self.blending,
                                         blender=self.blender)[0]

    # ---- batched rendering -------------------------------------------------------------------------
    def copy_latent(self, latent=None):
        """Deep copy of a LatentRepresentation (default: the current one); the face code is shared, like interpolate()."""
        latent = self.cur_latent if latent is None else latent
        out = LatentRepresentation()
        for att in ('curliness', 'shape', 'texture'):
            out.__setattr__(att, latent.__getattribute__(att).clone())
        out.color = {k: v.clone() for k, v in latent.color.items()}
        out.face = latent.face
        return out

    def outputs(self, latents, noise=None):
        """Batched output(): renders a list of LatentRepresentation with ONE shape-decoder, ONE colour-generator and ONE SEAN
        generator call (the reference offers only batch-1 output() and loops over it: shape_branch/validation_in_train.py:
        114-121, color_texture_branch/solver.py:270-299, README "Editing with Batch").  Image i equals what
        `output(latents[i])` returns for the same noise; cur_latent / cur_mask are left untouched.  Returns (list of uint8
        RGB images, uint8 label maps [N,256,256])."""
        latents = list(latents)
        n = len(latents)
        if n == 0:
            return [], np.zeros((0, 256, 256), dtype=np.uint8)
        shape = torch.cat([l.shape for l in latents], dim=0)
        face = torch.cat([l.face for l in latents], dim=0)
        masks = self.mask_generator.decode_labels(shape, face)                       # [N,256,256] uint8 on device
        rgb = torch.cat([l.color['rgb_mean'] if 'rgb_mean' in l.color else self.tensor_hsv_to_rgb(l.color['hsv'])
                         for l in latents], dim=0)
        data = {'noise': torch.cat([l.texture for l in latents], dim=0),
                'noise_curliness': torch.cat([l.curliness for l in latents], dim=0),
                'rgb_mean': rgb, 'pca_std': torch.cat([l.color['pca_std'] for l in latents], dim=0)}
        feature = self.feature_generator(data)['code']                               # [N,512]
        codes = self.input_sean_code.expand(n, -1, -1).clone()
        codes[:, HAIR_IDX] = feature
        masks_np = U.to_host(masks)
        lab = np.stack([self._mask_for_sean(m) for m in masks_np])
        if noise is None and self.noise is not None:                                 # pinned planes: same draw for every image
            noise = self.noise.expand(n, -1).contiguous() if self.noise.shape[0] == 1 else self.noise
        imgs = self.gen_imgs(codes, lab, noise=noise)
        out = [self.postprocess_blending(self.input_img, imgs[i], self.input_mask, masks_np[i], blending=self.blending,
                                         blender=self.blender)[0] for i in range(n)]
        return out, masks_np

    def sweep(self, att_name, idx, values, noise=None):
        """Slider sweep: the images output() would give after change_<att_name>(v, idx) for each v in `values`, rendered as
        one batch (att_name in 'curliness' | 'color' | 'shape' | 'texture'; idx ignored for curliness).  The current
        latent is restored afterwards."""
        saved, saved_mask = self.cur_latent, self.cur_mask
        latents = []
        try:
            for v in values:
                self.cur_latent = self.copy_latent(saved)
                if att_name == 'curliness':
                    self.change_curliness(v)
                elif att_name == 'color':
                    self.change_color(v, idx)
                elif att_name == 'texture':
                    self.change_texture(v, idx)
                elif att_name == 'shape':          # change_shape() would also decode a mask per value; outputs() batches that
                    self.cur_latent.shape = self.cur_latent.shape + \
                        (v - torch.dot(self.cur_latent.shape[0], self.shape_dirs[idx])) * self.shape_dirs[idx]
                else:
                    raise ValueError(f'unknown attribute {att_name!r}')
                latents.append(self.cur_latent)
        finally:
            self.cur_latent, self.cur_mask = saved, saved_mask
        return self.outputs(latents, noise=noise)

    def interpolate_grid(self, latent1, latent2, alphas, att_name=None, noise=None):
        """Images along latent1 -> latent2 for every alpha, one batch: interpolate() (all attributes) or
        interpolate_each_att(att_name) per alpha (ui/backend.py:323-395)."""
        lat = [self.interpolate(latent1, latent2, a) if att_name is None else
               self.interpolate_each_att(latent1, latent2, a, att_name) for a in alphas]
        return self.outputs(lat, noise=noise)

    # ---- sliders (ui/backend.py:177-264) ---------------------------------------------------------------------
    def change_curliness(self, val):
        self.cur_latent.curliness[0] = val

    def change_color(self, val, idx):
        color = self.cur_latent.color
        if idx == 3:      # slider in [-max, max] -> colour variance (pca_std) in [20, 120]
            color['pca_std'][0] = 20 + 100 * (val + self.maximum_value_fe) / 2 / self.maximum_value_fe
        else:             # h / s / v sliders live in the dataset's Gaussianised space
            color['hsv'][0][idx] = self.dist_translation.gaussian_to_val(idx, val)

    def change_shape(self, val, idx):
        self.continue_change_with_direction('shape', self.shape_dirs[idx], val)
        self.refresh_cur_mask()

    def change_texture(self, val, idx):
        self.continue_change_with_direction('texture', self.texture_dirs[idx], val)

    def get_curliness_be2fe(self):
        return self.cur_latent.curliness[0]

    def get_color_be2fe(self):
        hsv = U.to_host(self.cur_latent.color['hsv'])[0]
        sliders = [self.dist_translation.val_to_gaussian(i, hsv[i]) for i in range(3)]
        spread = (self.cur_latent.color['pca_std'][0] - 20) / 100            # inverse of change_color(.., 3)
        return (*sliders, spread * 2 * self.maximum_value_fe - self.maximum_value_fe)

    def get_shape_be2fe(self):
        return [torch.dot(self.cur_latent.shape[0], self.shape_dirs[idx]) for idx in range(4)]

    def get_texture_be2fe(self):
        return [torch.dot(self.cur_latent.texture[0], self.texture_dirs[idx]) for idx in range(2)]

    # ---- transfer (ui/backend.py:266-302) ---------------------------------------------------------------------
    def transfer_latent_representation(self, flag, refresh=True):
        if flag == 'shape':
            if self.warper is None:
                raise RuntimeError("transfer_latent_representation('shape') warps the target hair mask with the reference's "
                                   "ARAP tool chain (wrap_codes.mask_adaptor.wrap_by_imgs: dlib + my_arap binaries); pass "
                                   "Backend(..., warper=wrap_by_imgs) to enable it")
            wt, _ = self.warper(self.target_img, self.input_img, wrap_temp_folder=self.temp_path, need_crop=False)
            wt = self.preprocess_mask(wt)
            self.warp_target = wt[0, 0]
            w256 = U.resize_nearest(wt[0, 0], (256, 256))
            hair_code, face_code = self.mask_generator.encode_labels(torch.tensor(w256[None], dtype=torch.uint8,
                                                                                  device=self.device))
            self.target_latent.shape = hair_code
            self.target_latent.face = face_code
            self.refresh_cur_mask()
        src = getattr(self.target_latent, flag)          # tensors are cloned, the colour dict entry by entry
        setattr(self.cur_latent, flag, src.clone() if isinstance(src, torch.Tensor) else {k: v.clone() for k, v in src.items()})
        if flag == 'shape' and refresh:
            self.refresh_cur_mask()
        if flag == 'texture':                            # texture and curliness travel together (:300-302)
            self.transfer_latent_representation('curliness')

    def refresh_cur_mask(self, target_latent=None):      # :304-315
        lat = self.cur_latent if target_latent is None else target_latent
        # The reference decodes twice per shape move (change_shape -> continue_change_with_direction -> refresh_cur_mask, then
        # refresh_cur_mask again, :217-218,461-462): the second call sees the very same latent tensors and gets the first call's mask.
        key = (lat.shape, lat.shape._version, lat.face, lat.face._version)
        c = getattr(self, '_mask_cache', None)
        if c is not None and c[0][0] is key[0] and c[0][1] == key[1] and c[0][2] is key[2] and c[0][3] == key[3]:
            self.cur_mask = c[1].copy()
        else:
            mask = U.to_host(self.mask_generator.decode_labels(lat.shape, lat.face))[0]
            self._mask_cache = (key, mask)
            self.cur_mask = mask.copy()
        return self.cur_mask, mask_to_rgb(self.cur_mask, draw_type=1)

    def get_cur_mask(self):
        return mask_to_rgb(self.cur_mask, draw_type=1)

    # ---- interpolation (ui/backend.py:323-395) ------------------------------------------------------------------
    def interpolate_hsv(self, hsv1, hsv2, alpha):
        mix = self.tensor_hsv_to_rgb(hsv1) * (1 - alpha) + self.tensor_hsv_to_rgb(hsv2) * alpha     # blend in RGB, not in hue
        return self.tensor_rgb_to_hsv(mix)

    def interpolate_triple(self, latent1, latent2, latent3, alpha1, alpha2, alpha3):
        return self.interpolate(self.interpolate(latent1, latent2, alpha2 / (alpha1 + alpha2)), latent3, alpha3)
    @staticmethod
    def _lerp(a, b, alpha):
        return a * (1 - alpha) + b * alpha

    def _lerp_color(self, c1, c2, alpha):
        return {'pca_std': self._lerp(c1['pca_std'], c2['pca_std'], alpha), 'hsv': self.interpolate_hsv(c1['hsv'], c2['hsv'], alpha)}

    def interpolate(self, latent1, latent2, alpha):
        out = LatentRepresentation()
        for name in ('curliness', 'shape', 'texture'):
            setattr(out, name, self._lerp(getattr(latent1, name), getattr(latent2, name), alpha))
        out.color = self._lerp_color(latent1.color, latent2.color, alpha)
        out.face = self.cur_latent.face
        return out

    def interpolate_each_att(self, latent1, latent2, alpha, att_name):
        """Start from the current latent and move only one attribute family along latent1 -> latent2: 'shape'; 'curliness'
        or 'texture' (always both); anything else = colour."""
        out = self.copy_latent()
        moved = {'shape': ('shape',), 'curliness': ('curliness', 'texture'), 'texture': ('curliness', 'texture')}.get(att_name)
        if moved is None:
            out.color = self._lerp_color(latent1.color, latent2.color, alpha)
        else:
            for name in moved:
                setattr(out, name, self._lerp(getattr(latent1, name), getattr(latent2, name), alpha))
            out.color = {k: self.cur_latent.color[k].clone() for k in ('hsv', 'pca_std')}
        return out

    @staticmethod
    def show_hair_region(mask, non_hair_value=0):
        mask_rgb = mask_to_rgb(mask, draw_type=1)
        mask_rgb[mask != HAIR_IDX] = non_hair_value
        return mask_rgb

    def directly_change_hair_mask(self, hair_mask):      # :410-422
        face = self.mask_generator.forward_face_decoder(self.cur_latent.face)
        lo, hi = face.min(), face.max()
        # a hair "logit" that beats every face logit inside the drawn region (hi + 1) and loses everywhere else (lo - 1)
        drawn = torch.tensor(hair_mask == HAIR_IDX)[None, None, ...].type_as(face).to(self.device)
        hair = drawn * (hi - lo + 2) + lo - 1
        self.cur_mask = U.to_host(mask_one_hot_to_label(self.mask_generator.forward_decoder(hair, face)))[0]

    def get_random_texture(self):
        self.cur_latent.texture = generate_noise(1, 8).to(self.device)

    def get_random_shape(self):
        self.cur_latent.shape = generate_noise(1, 16).to(self.device)
        self.refresh_cur_mask()

    def get_random_curliness(self):
        self.cur_latent.curliness = generate_noise(1, 1).to(self.device)

    def continue_change_with_direction(self, att_name, direction, val):     # :450-462
        cur = getattr(self.cur_latent, att_name)          # move along `direction` until the projection equals `val`
        setattr(self.cur_latent, att_name, cur + (val - torch.dot(cur[0], direction)) * direction)
        if att_name == 'shape':
            self.refresh_cur_mask()
