"""The whole CtrlHair edit as ONE batched, device-resident pass (BASELINE.json configs[2], SURVEY.md 8d Config 3):

    portrait -> BiSeNet parse @512 -> CelebAMask ids -> nearest 256 -> shape encoders            (ui/backend.py:67-90)
             -> Zencoder @S -> hair code -> colour predictor + colour/texture encoder              (ui/backend.py:93-105)
             -> sliders (curliness, texture / shape directions, HSV in the Gaussianised space)     (ui/backend.py:177-264,450-462)
             -> colour/texture generator -> new hair code; shape decoder -> new mask               (ui/backend.py:147-170,304-315)
             -> nearest x2 -> SEAN generator @S                                                    (hair_editor.py:159-179)

It is what `Backend.set_input_img(img); Backend.change_*(...); Backend.output()` computes for one portrait, restated over
a batch of portraits with every tensor left on the GPU (the reference round-trips through numpy / cv2 between the
networks and handles one image at a time).  `ctrlhair_amd.ui.backend.Backend` remains the drop-in API;
tests/test_pipeline.py pins this composition to a fixture made with the reference's own modules and Backend to this.
"""
from typing import Dict, Optional

import numpy as np
import torch

from . import hostutil as U
from .hostutil import HAIR_IDX

# SURVEY.md 8d Config 3 slider deltas
DEFAULT_SLIDERS = {'curliness': 1.0, 'texture': (0, 1.5), 'shape': (0, -1.0), 'hsv_gaussian': (2, 1.0)}

_MEAN = (0.485, 0.456, 0.406)     # external_code/face_parsing/my_parsing_util.py:27
_STD = (0.229, 0.224, 0.225)


_HSV_TABLES: Dict[torch.device, tuple] = {}


def _hsv_tables(device):
    """OpenCV's division tables on `device`, uploaded once: a host-to-device copy from pageable memory inside the edit would
    make the host wait for the stream to drain (tools/host_enqueue.py), i.e. serialise consecutive edits."""
    t = _HSV_TABLES.get(device)
    if t is None:
        t = _HSV_TABLES[device] = (torch.as_tensor(U._SDIV).to(device), torch.as_tensor(U._HDIV).to(device))
    return t


def rgb_to_hsv_u8(rgb: torch.Tensor) -> torch.Tensor:
    """cv2.cvtColor(uint8 RGB, COLOR_RGB2HSV) on [N,3] uint8-valued tensors, on the tensor's device: the same 12-bit
    fixed-point arithmetic as hostutil.rgb_to_hsv_u8 (H in [0,180), S, V in [0,255])."""
    a = rgb.long()
    r, g, b = a[:, 0], a[:, 1], a[:, 2]
    v = a.max(dim=1).values
    d = v - a.min(dim=1).values
    sdiv, hdiv = _hsv_tables(rgb.device)
    half = 1 << (U._HSV_SHIFT - 1)
    h = torch.where(v == r, g - b, torch.where(v == g, b - r + 2 * d, r - g + 4 * d))
    h = (h * hdiv[d] + half) >> U._HSV_SHIFT
    h = torch.where(h < 0, h + 180, h)
    s = (d * sdiv[v] + half) >> U._HSV_SHIFT
    return torch.stack([h, s, v], dim=1).float()


def hsv_to_rgb_u8(hsv: torch.Tensor) -> torch.Tensor:
    """cv2.cvtColor(uint8 HSV, COLOR_HSV2RGB) on [N,3] (hostutil.hsv_to_rgb_u8: float32 sector formula, round half to even)."""
    a = hsv.float()
    h, s, v = a[:, 0] * (6.0 / 180.0), a[:, 1] * (1.0 / 255.0), a[:, 2] * (1.0 / 255.0)
    sec = torch.floor(h).long()
    f = h - sec
    sec = sec % 6
    p0, p1, p2, p3 = v, v * (1 - s), v * (1 - s * f), v * (1 - s * (1 - f))
    pick = lambda opts: torch.stack(opts, dim=1).gather(1, sec[:, None])[:, 0]
    r, g, b = pick([p0, p2, p1, p1, p3, p0]), pick([p3, p0, p0, p2, p1, p1]), pick([p1, p1, p3, p0, p0, p2])
    return torch.round(torch.stack([r, g, b], dim=1) * 255.0).clamp(0, 255)


class EditPipeline:
    def __init__(self, weights: Optional[Dict[str, dict]] = None, device: int = 0, img_size: int = 512, max_batch: int = 8,
                 f16x3=None, models=None, texture_dirs=None, shape_dirs=None, hsv_table=None, options=None):
        """f16x3: None = by the weights, like HairEditor: 0 (exact f32) for a released checkpoint (a dict made by
        checkpoints.reference_checkpoints), 1 (split-operand f16 MFMA, f32-class) for procedural / untagged weights."""
        from .hair_editor import HipModels, is_released_checkpoint, procedural_weights
        if f16x3 is None:
            f16x3 = 0 if is_released_checkpoint(weights) else 1
        if models is None:
            models = HipModels(weights if weights is not None else procedural_weights(), device=device, img_size=img_size,
                               max_batch=max_batch, f16x3=f16x3, options=options)
        self.models = models
        self.device = models.device
        self.img_size = img_size
        td = texture_dirs if texture_dirs is not None else U.seeded_directions(2, 8, seed=45)      # hair_editor.py:82-91
        sdirs = shape_dirs if shape_dirs is not None else U.seeded_directions(4, 16, seed=54)      # hair_editor.py:110-119
        self.texture_dirs = torch.as_tensor(np.asarray(td)).float().to(self.device)
        self.shape_dirs = torch.as_tensor(np.asarray(sdirs)).float().to(self.device)
        self.dist_translation = U.DistTranslation(hsv_table)
        import os
        med = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data', 'mean_style_code.npz'))['median']
        self.median = torch.from_numpy(med.astype(np.float32)).to(self.device)
        self.side = torch.cuda.Stream(self.device)      # edit(): the shape branch runs here, underneath the Zencoder
        self.overlap = True
        self.split_encode = False      # edit(): BiSeNet underneath the Zencoder's convs (ch_sean_encode_features / _regions);
                                       # measured: no further gain once the shape branch runs aside (tools/edit_modes.py)
        self.mean = torch.tensor(_MEAN, device=self.device).view(1, 3, 1, 1)
        self.std = torch.tensor(_STD, device=self.device).view(1, 3, 1, 1)

    def close(self):
        self.models.generator.handle.close()

    # ---- stages (each returns device tensors; citations: module docstring) -----------------------------------------------
    def parse(self, img: torch.Tensor) -> torch.Tensor:
        """img [B,3,S,S] in [-1,1] -> CelebAMask-HQ label map uint8 [B,S,S].  As the reference does for every img_size
        (my_parsing_util.py:34-35, hair_editor.py:331-335): the parser always sees a 512x512 bilinear resize of the portrait
        (8-bit, like the PIL image it resizes), and the label map is nearest-resized back to S (cv2 INTER_NEAREST:
        src = floor(dst * 512 / S)); ToTensor + Normalize, net, argmax, remap in between."""
        S = img.shape[-1]
        x01 = img * 0.5 + 0.5
        if S != 512:
            # PIL resamples in two passes, horizontal then vertical, each stored as 8-bit (ImagingResample)
            u8 = torch.round(x01 * 255.0).clamp(0, 255)
            for size in ((S, 512), (512, 512)):
                u8 = torch.nn.functional.interpolate(u8, size=size, mode='bilinear', align_corners=False, antialias=S > 512)
                u8 = torch.round(u8).clamp(0, 255)
            x01 = u8 / 255.0
        lab = self.models.face_parsing.parse_tensor((x01 - self.mean) / self.std)[0]
        if S != 512:
            idx = (torch.arange(S, device=lab.device) * 512) // S
            lab = lab[:, idx][:, :, idx].contiguous()
        return lab

    def analyse(self, img: torch.Tensor, labels: torch.Tensor):
        """-> dict(shape [B,16], face [B,1024], codes [B,19,512], rgb_mean [B,3], pca_std [B,1], texture [B,8],
        curliness [B,1]) -- Backend.parse_img's latent representation, batched."""
        m = self.models
        S = img.shape[-1]
        lab256 = labels if S == 256 else labels[:, ::S // 256, ::S // 256].contiguous()       # cv2 INTER_NEAREST: floor(dst * in/out)
        shape, face = m.mask_generator.encode_labels(lab256)
        codes = m.generator.encode(img, labels)
        hair = codes[:, HAIR_IDX].contiguous()
        stats = m.solver_feature.rgb_model({'code': hair})
        lat = m.solver_feature.dis({'code': hair})
        return {'shape': shape, 'face': face, 'codes': codes, 'rgb_mean': stats['rgb_mean'], 'pca_std': stats['pca_std'],
                'texture': lat['noise'], 'curliness': lat['noise_curliness']}

    def apply_sliders(self, lat: dict, sliders: dict, only_shape: bool = False) -> dict:
        """Backend.change_curliness / change_texture / change_shape / change_color on every sample.  only_shape: just the
        shape direction (what edit() runs on its side stream; `lat` then only needs 'shape')."""
        out = dict(lat)
        move = lambda cur, d, val: cur + (val - cur @ d)[:, None] * d[None]                     # continue_change_with_direction
        if only_shape:
            if sliders.get('shape') is not None:
                idx, val = sliders['shape']
                out['shape'] = move(lat['shape'], self.shape_dirs[idx], val)
            return out
        # colour lives as uint8 HSV in the reference (ui/backend.py:100: the predicted mean RGB is truncated to uint8)
        hsv = rgb_to_hsv_u8(lat['rgb_mean'].clamp(0, 255).floor())
        if sliders.get('hsv_gaussian') is not None:
            idx, val = sliders['hsv_gaussian']
            hsv = hsv.clone()
            hsv[:, idx] = float(np.uint8(self.dist_translation.gaussian_to_val(idx, val)))   # stored into a uint8 tensor (:193)
        out['hsv'] = hsv
        out['rgb'] = hsv_to_rgb_u8(hsv)                                                         # tensor_hsv_to_rgb (:108-115)
        if sliders.get('curliness') is not None:
            out['curliness'] = torch.full_like(lat['curliness'], float(sliders['curliness']))
        if sliders.get('texture') is not None:
            idx, val = sliders['texture']
            out['texture'] = move(lat['texture'], self.texture_dirs[idx], val)
        if sliders.get('shape') is not None:
            idx, val = sliders['shape']
            out['shape'] = move(lat['shape'], self.shape_dirs[idx], val)
        return out

    def render(self, lat: dict, noise: Optional[torch.Tensor] = None, seed: int = 0, out: Optional[torch.Tensor] = None,
               mask: Optional[torch.Tensor] = None):
        """Backend.output(): colour/texture generator -> hair code, shape decoder -> mask, SEAN generator.
        Returns (image [B,3,S,S] in [-1,1], mask uint8 [B,256,256])."""
        m = self.models
        feature = m.solver_feature.gen({'noise': lat['texture'], 'noise_curliness': lat['curliness'], 'rgb_mean': lat['rgb'],
                                        'pca_std': lat['pca_std']})['code']
        codes = lat['codes'].clone()
        codes[:, HAIR_IDX] = feature                                                            # ui/backend.py:170
        zero = (codes == 0).all(dim=2, keepdim=True)                                            # absent regions keep the median
        codes = torch.where(zero, self.median[None].expand_as(codes), codes).contiguous()       # code (hair_editor.py:165-168)
        if mask is None:
            mask = m.mask_generator.decode_labels(lat['shape'], lat['face'])                    # ui/backend.py:304-315
        r = self.img_size // 256
        lab = mask if r == 1 else mask.repeat_interleave(r, 1).repeat_interleave(r, 2).contiguous()
        return m.generator.generate(lab, codes, noise, seed=seed, out=out), mask

    def edit(self, img: torch.Tensor, sliders: Optional[dict] = None, noise: Optional[torch.Tensor] = None, seed: int = 1,
             out: Optional[torch.Tensor] = None, labels: Optional[torch.Tensor] = None, mask: Optional[torch.Tensor] = None,
             stages: Optional[dict] = None) -> torch.Tensor:
        """img: cuda float32 [B,3,S,S] in [-1,1] (what HairEditor.preprocess_img produces).  `labels` / `mask` override the
        parsed label map / the decoded mask (a caller-supplied parsing; the tests use them to step over argmax ties).
        `stages` (optional dict) receives every intermediate tensor."""
        sliders = DEFAULT_SLIDERS if sliders is None else sliders
        m, S = self.models, img.shape[-1]
        split_enc = self.overlap and self.split_encode and labels is None and img.shape[0] <= m.generator.max_batch
        if split_enc:
            # The Zencoder's convolutions need the image only (the labels enter its region means at the very end): they start
            # on the main stream while BiSeNet -- ~70 small kernels -- parses the image on the side stream underneath them.
            main = torch.cuda.current_stream(self.device)
            self.side.wait_stream(main)
            m.generator.encode_features(img)
            with torch.cuda.stream(self.side):
                labels = self.parse(img)
                labels_ready = torch.cuda.Event()
                labels_ready.record(self.side)
        elif labels is None:
            labels = self.parse(img)
        if not self.overlap:
            lat = self.analyse(img, labels)
            lat = self.apply_sliders(lat, sliders)
            image, mask = self.render(lat, noise=noise, seed=seed, out=out, mask=mask)
            if stages is not None:
                stages.update(lat, labels=labels, mask=mask, image=image)
            return image
        # The shape branch (encoders -> shape slider -> decoders -> mask) depends on the parsing only, the appearance branch
        # (Zencoder -> colour MLPs -> sliders -> colour generator) on image + parsing: they meet at the generator.  The
        # shape branch is a chain of small, latency-bound kernels, so it runs on a side stream underneath the Zencoder's
        # large convs instead of in front of them (same kernels, same results; ~2 ms per 8 edits).
        main = torch.cuda.current_stream(self.device)
        if not split_enc:
            self.side.wait_stream(main)
        with torch.cuda.stream(self.side):
            lab256 = labels if S == 256 else labels[:, ::S // 256, ::S // 256].contiguous()
            shape, face = m.mask_generator.encode_labels(lab256)
            shape_new = self.apply_sliders({'shape': shape}, sliders, only_shape=True)['shape']
            if mask is None:
                mask = m.mask_generator.decode_labels(shape_new, face)
        if split_enc:
            main.wait_event(labels_ready)
            codes = m.generator.encode_regions(labels)
        else:
            codes = m.generator.encode(img, labels)
        hair = codes[:, HAIR_IDX].contiguous()
        stats = m.solver_feature.rgb_model({'code': hair})
        enc = m.solver_feature.dis({'code': hair})
        lat = {'shape': shape_new, 'face': face, 'codes': codes, 'rgb_mean': stats['rgb_mean'], 'pca_std': stats['pca_std'],
               'texture': enc['noise'], 'curliness': enc['noise_curliness']}
        lat = self.apply_sliders(lat, {k: v for k, v in sliders.items() if k != 'shape'})
        main.wait_stream(self.side)
        image, mask = self.render(lat, noise=noise, seed=seed, out=out, mask=mask)
        if stages is not None:
            stages.update(lat, labels=labels, mask=mask, image=image)
        return image

    STAGES = ('parse', 'shape_encode', 'zencoder', 'colour', 'shape_decode', 'generator')

    def stage_times(self, img: torch.Tensor, reps: int = 3, before=None) -> Dict[str, float]:
        """ms per stage of one edit() over `img` (torch events on the current stream, averaged over `reps` runs; measurement
        only -- edit() itself is never instrumented).  Keys: STAGES = parse (BiSeNet), shape_encode, zencoder, colour (3 MLPs +
        sliders), shape_decode, generator.  `before(name)`, if given, runs ahead of every stage (tools/stage_trace.py launches
        a marker kernel there so that a rocprofv3 kernel trace can be cut into stages)."""
        m, S = self.models, img.shape[-1]
        acc: Dict[str, float] = {}

        def timed(name, fn):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if before is not None:
                before(name)
            e0.record()
            r = fn()
            e1.record()
            acc.setdefault(name, []).append((e0, e1))
            return r

        for _ in range(reps):
            labels = timed('parse', lambda: self.parse(img))
            lab256 = labels if S == 256 else labels[:, ::S // 256, ::S // 256].contiguous()
            shape, face = timed('shape_encode', lambda: m.mask_generator.encode_labels(lab256))
            codes = timed('zencoder', lambda: m.generator.encode(img, labels))

            def colour():
                hair = codes[:, HAIR_IDX].contiguous()
                st = m.solver_feature.rgb_model({'code': hair})
                la = m.solver_feature.dis({'code': hair})
                lat = self.apply_sliders({'shape': shape, 'face': face, 'codes': codes, 'rgb_mean': st['rgb_mean'], 'pca_std': st['pca_std'],
                                          'texture': la['noise'], 'curliness': la['noise_curliness']}, DEFAULT_SLIDERS)
                lat['feature'] = m.solver_feature.gen({'noise': lat['texture'], 'noise_curliness': lat['curliness'],
                                                       'rgb_mean': lat['rgb'], 'pca_std': lat['pca_std']})['code']
                return lat
            lat = timed('colour', colour)
            mask = timed('shape_decode', lambda: m.mask_generator.decode_labels(lat['shape'], lat['face']))

            def gen():
                c = lat['codes'].clone()
                c[:, HAIR_IDX] = lat['feature']
                zero = (c == 0).all(dim=2, keepdim=True)
                c = torch.where(zero, self.median[None].expand_as(c), c).contiguous()
                r = self.img_size // 256
                lab = mask if r == 1 else mask.repeat_interleave(r, 1).repeat_interleave(r, 2).contiguous()
                return m.generator.generate(lab, c, None, seed=1)
            timed('generator', gen)
        torch.cuda.synchronize()
        return {k: sum(a.elapsed_time(b) for a, b in v) / len(v) for k, v in acc.items()}
