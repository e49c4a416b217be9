"""Mirror of /root/reference/ui/backend.py::Backend (the editing API CtrlHair scripts call) on the MI355X library.
Every public method of ui/backend.py:67-462 is present with the same name, argument order and return types; line
references are given per method.  Shape transfer by photo (`transfer_latent_representation('shape')`) needs the
reference's ARAP warping tool chain (wrap_codes/, dlib) and accepts an injected `warper` instead.
"""
import copy
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

    # ---- analysis (ui/backend.py:67-106) ---------------------------------------------------------------------
    def _mask_for_sean(self, mask256):
        """The shape branch is fixed at 256x256 (shape_branch/model.py:85-89); for img_size 512 the label map is
        nearest-upsampled x2 before the SEAN generator (SURVEY.md 8d Config 3)."""
        if self.img_size == mask256.shape[-1]:
            return mask256
        return U.resize_nearest(mask256.astype('uint8'), (self.img_size, self.img_size))

    def parse_img(self, img_rgb, target_img=False):
        img_ts = U.resize_bilinear(np.asarray(img_rgb), (self.target_size, self.target_size))
        mask = self.get_mask(img_rgb)                                     # [img_size, img_size] CelebA ids
        lr = LatentRepresentation()
        mask256 = U.resize_nearest(mask, (256, 256)) if mask.shape[0] != 256 else mask
        mask_batch = self.preprocess_mask(mask)
        mask_tensor = torch.tensor(mask256[None], dtype=torch.uint8, device=self.device)
        hair_code, face_code = self.mask_generator.encode_labels(mask_tensor)   # == one-hot, split, two encoders (:81-86)
        lr.shape = hair_code
        lr.face = face_code
        out_mask = self.mask_generator.decode_labels(hair_code, face_code).cpu().numpy()[0]   # :87-90
        # infer feature (:93-105)
        input_code = self.get_code(self.preprocess_img(img_rgb), mask_batch)
        hair_feature = input_code[:, HAIR_IDX]
        out_color = self.feature_rgb_predictor({'code': hair_feature})
        c = out_color['rgb_mean'].detach().cpu().numpy()
        c_hsv = U.rgb_to_hsv_u8(np.clip(c, 0, 255)[None, ...].astype('uint8'))
        lr.color = {'hsv': torch.tensor(c_hsv).to(self.device)[0], 'pca_std': out_color['pca_std']}
        out_enc = self.feature_encoder({'code': hair_feature})
        lr.curliness = out_enc['noise_curliness']
        lr.texture = out_enc['noise']
        return img_ts, out_mask, lr, mask, input_code, hair_feature

    def tensor_hsv_to_rgb(self, hsv):             # :108-115
        c = hsv.detach().cpu().numpy()
        return torch.tensor(U.hsv_to_rgb_u8(c[None, ...].astype('uint8'))).to(self.device)[0]

    def tensor_rgb_to_hsv(self, rgb):             # :117-125
        c = rgb.detach().cpu().numpy()
        return torch.tensor(U.rgb_to_hsv_u8(c[None, ...].astype('uint8'))).to(self.device)[0]

    def set_input_img(self, img_rgb):             # :127-135
        self.input_img, self.cur_mask, self.cur_latent, \
            self.input_mask, self.input_sean_code, self.input_hair_feature = self.parse_img(img_rgb)
        return self.input_img, mask_to_rgb(self.cur_mask, draw_type=1)

    def set_target_img(self, img_rgb):            # :137-145
        self.target_img, _, self.target_latent, \
            self.target_mask, _, self.target_hair_feature = self.parse_img(img_rgb)
        return self.target_img, mask_to_rgb(self.target_mask, draw_type=1)

    # ---- render (ui/backend.py:147-175) ----------------------------------------------------------------------
    def output(self, target_latent=None, feature=None):
        if target_latent is None:
            target_latent = self.cur_latent
            target_mask = self.cur_mask
        else:
            target_mask = self.refresh_cur_mask(target_latent)[0]
        if 'rgb_mean' in target_latent.color:
            target_color_rgb = self.target_latent.color['rgb_mean']
        else:
            target_color_rgb = self.tensor_hsv_to_rgb(target_latent.color['hsv'])
        if feature is None:
            data = {'noise': target_latent.texture, 'noise_curliness': target_latent.curliness,
                    'rgb_mean': target_color_rgb, 'pca_std': target_latent.color['pca_std']}
            feature = self.feature_generator(data)['code']
        self.input_sean_code[:, HAIR_IDX] = feature                      # in-place, like :170
        edit_img = self.gen_img(self.input_sean_code, self._mask_for_sean(target_mask)[None, None, ...], noise=self.noise)
        output_img, _ = self.postprocess_blending(self.input_img, edit_img, self.input_mask, target_mask,
                                                  blending=self.blending, blender=self.blender)
        return output_img

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
        masks_np = masks.cpu().numpy()
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
        if idx == 3:
            val = (val + self.maximum_value_fe) / 2 / self.maximum_value_fe
            self.cur_latent.color['pca_std'][0] = val * 100 + 20
        else:
            val = self.dist_translation.gaussian_to_val(idx, val)
            self.cur_latent.color['hsv'][0][idx] = val

    def change_shape(self, val, idx):
        self.continue_change_with_direction('shape', self.shape_dirs[idx], val)
        self.refresh_cur_mask()

    def change_texture(self, val, idx):
        self.continue_change_with_direction('texture', self.texture_dirs[idx], val)

    def get_curliness_be2fe(self):
        return self.cur_latent.curliness[0]

    def get_color_be2fe(self):
        c_hsv = self.cur_latent.color['hsv'].detach().cpu().numpy()[0]
        color0 = self.dist_translation.val_to_gaussian(0, c_hsv[0])
        color1 = self.dist_translation.val_to_gaussian(1, c_hsv[1])
        color2 = self.dist_translation.val_to_gaussian(2, c_hsv[2])
        var_fe = (self.cur_latent.color['pca_std'][0] - 20) / 100 * 2 * self.maximum_value_fe - self.maximum_value_fe
        return color0, color1, color2, var_fe

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
        target_att = self.target_latent.__getattribute__(flag)
        if isinstance(target_att, torch.Tensor):
            self.cur_latent.__setattr__(flag, target_att.clone())
        else:
            cp_dict = copy.copy(target_att)
            for ke in cp_dict:
                cp_dict[ke] = cp_dict[ke].clone()
            self.cur_latent.__setattr__(flag, cp_dict)
        if flag == 'shape' and refresh:
            self.refresh_cur_mask()
        if flag == 'texture':
            self.transfer_latent_representation('curliness')

    def refresh_cur_mask(self, target_latent=None):      # :304-315
        if target_latent is None:
            target_latent = self.cur_latent
        out_mask = self.mask_generator.decode_labels(target_latent.shape, target_latent.face).cpu().numpy()[0]
        self.cur_mask = out_mask
        return out_mask, mask_to_rgb(out_mask, draw_type=1)

    def get_cur_mask(self):
        return mask_to_rgb(self.cur_mask, draw_type=1)

    # ---- interpolation (ui/backend.py:323-395) ------------------------------------------------------------------
    def interpolate_hsv(self, hsv1, hsv2, alpha):
        rgb1 = self.tensor_hsv_to_rgb(hsv1)
        rgb2 = self.tensor_hsv_to_rgb(hsv2)
        return self.tensor_rgb_to_hsv(rgb1 * (1 - alpha) + rgb2 * alpha)

    def interpolate_triple(self, latent1, latent2, latent3, alpha1, alpha2, alpha3):
        latent12 = self.interpolate(latent1, latent2, alpha2 / (alpha1 + alpha2))
        return self.interpolate(latent12, latent3, alpha3)

    def interpolate(self, latent1, latent2, alpha):
        result_latent = LatentRepresentation()
        for att in ['curliness', 'shape', 'texture']:
            result_latent.__setattr__(att, latent1.__getattribute__(att) * (1 - alpha) + latent2.__getattribute__(att) * alpha)
        color_dic = {'pca_std': latent1.color['pca_std'] * (1 - alpha) + latent2.color['pca_std'] * alpha,
                     'hsv': self.interpolate_hsv(latent1.color['hsv'], latent2.color['hsv'], alpha)}
        result_latent.color = color_dic
        result_latent.face = self.cur_latent.face
        return result_latent

    def interpolate_each_att(self, latent1, latent2, alpha, att_name):
        result_latent = LatentRepresentation()
        for att in ['curliness', 'shape', 'texture']:
            result_latent.__setattr__(att, self.cur_latent.__getattribute__(att).clone())
        if att_name == 'shape':
            color_dic = {s: self.cur_latent.color[s].clone() for s in ['hsv', 'pca_std']}
            result_latent.__setattr__(att_name, latent1.__getattribute__(att_name) * (1 - alpha) +
                                      latent2.__getattribute__(att_name) * alpha)
        elif att_name in ['curliness', 'texture']:
            color_dic = {s: self.cur_latent.color[s].clone() for s in ['hsv', 'pca_std']}
            for a in ('curliness', 'texture'):
                result_latent.__setattr__(a, latent1.__getattribute__(a) * (1 - alpha) + latent2.__getattribute__(a) * alpha)
        else:
            color_dic = {'pca_std': latent1.color['pca_std'] * (1 - alpha) + latent2.color['pca_std'] * alpha,
                         'hsv': self.interpolate_hsv(latent1.color['hsv'], latent2.color['hsv'], alpha)}
        result_latent.color = color_dic
        result_latent.face = self.cur_latent.face
        return result_latent

    @staticmethod
    def show_hair_region(mask, non_hair_value=0):
        mask_rgb = mask_to_rgb(mask, draw_type=1)
        mask_rgb[mask != HAIR_IDX] = non_hair_value
        return mask_rgb

    def directly_change_hair_mask(self, hair_mask):      # :410-422
        hair_mask = hair_mask == HAIR_IDX
        face_logit = self.mask_generator.forward_face_decoder(self.cur_latent.face)
        hair_logit = torch.tensor(hair_mask)[None, None, ...].type_as(face_logit).to(self.device)
        hair_logit = hair_logit * (face_logit.max() - face_logit.min() + 2) + face_logit.min() - 1
        mask = self.mask_generator.forward_decoder(hair_logit, face_logit)
        self.cur_mask = mask_one_hot_to_label(mask).cpu().numpy()[0]

    def get_random_texture(self):
        self.cur_latent.texture = generate_noise(1, 8).to(self.device)

    def get_random_shape(self):
        self.cur_latent.shape = generate_noise(1, 16).to(self.device)
        self.refresh_cur_mask()

    def get_random_curliness(self):
        self.cur_latent.curliness = generate_noise(1, 1).to(self.device)

    def continue_change_with_direction(self, att_name, direction, val):     # :450-462
        att = self.cur_latent.__getattribute__(att_name)
        att = att + (val - torch.dot(att[0], direction)) * direction
        self.cur_latent.__setattr__(att_name, att)
        if att_name == 'shape':
            self.refresh_cur_mask()
