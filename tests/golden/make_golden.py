"""Generate committed golden vectors by running the *reference* modules (imported from
/root/reference, CPU, via refharness.py) on procedural weights + seeded synthetic inputs.

Build-container only.  Fixtures hold inputs/seeds and expected outputs -- never weights, never
reference source.  Large outputs are stored as crops + a stride-4 subsample + per-channel sums.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from ctrlhair_amd import procedural as P      # noqa: E402
import refharness as R                        # noqa: E402

CROPS = [(0, 0), (37, 101), (128, 64), (-64, -64)]   # top-left corners of 64x64 crops (negative = from end)


face_like_labels = P.face_like_labels      # (moved to the package: bench.py's face-like workload variant uses it too)


def summarize(img):
    """img [B,3,S,S] -> dict of crops / subsample / channel sums."""
    B, _, S, _ = img.shape
    out = {'sums': img.astype(np.float64).sum(axis=(2, 3)), 'sub4': img[:, :, ::4, ::4].copy()}
    c = min(64, S)
    for i, (y, x) in enumerate(CROPS):
        y = y % S; x = x % S
        y = min(y, S - c); x = min(x, S - c)
        out[f'crop{i}'] = img[:, :, y:y + c, x:x + c].copy()
        out[f'crop{i}_yx'] = np.array([y, x])
    return out


def median_codes():
    z = np.load(os.path.join(ROOT, 'ctrlhair_amd', 'data', 'mean_style_code.npz'))
    return z['median'].astype(np.float32)


def case(name, ngf, S, B, ui, wseed=0, lseed=1234, cseed=2024, nseed=7, grid=16, labels='blocky', codes='tanh'):
    sd = P.sean_state_dict(wseed, ngf)
    if labels == 'blocky':
        lab = P.blocky_labels(B, S, seed=lseed, grid=grid)
    else:
        lab = np.stack([face_like_labels(S, lseed + b) for b in range(B)])
    cd = P.style_codes(B, seed=cseed)
    if codes == 'median':
        cd = np.repeat(median_codes()[None], B, 0)
    nz = P.noise_planes(B, S, ngf, seed=nseed)
    img = R.run_generator(sd, lab, cd, nz, ngf, ui_mode=ui)
    meta = dict(ngf=ngf, S=S, B=B, ui=int(ui), wseed=wseed, cseed=cseed, nseed=nseed,
                codes_kind=codes)
    out = {'labels': lab, **{'meta_' + k: np.array(v) for k, v in meta.items()}}
    if S <= 64:
        out['image'] = img
    else:
        out.update(summarize(img))
    path = os.path.join(HERE, f'sean_gen_{name}.npz')
    np.savez_compressed(path, **out)
    print(name, 'std', img.std(), 'bytes', os.path.getsize(path))


def zenc_case(name, S, B, labels='blocky', lseed=1234, iseed=31, grid=16):
    sd = P.sean_state_dict(0, 16)     # Zencoder weights do not depend on ngf
    lab = P.blocky_labels(B, S, seed=lseed, grid=grid) if labels == 'blocky' else \
        np.stack([face_like_labels(S, lseed + b) for b in range(B)])
    img = P.synthetic_images(B, S, seed=iseed)
    codes = R.run_zencoder(sd, img, lab)
    path = os.path.join(HERE, f'sean_zenc_{name}.npz')
    np.savez_compressed(path, labels=lab, codes=codes, meta_S=np.array(S), meta_B=np.array(B), meta_iseed=np.array(iseed))
    print('zenc', name, 'absent rows', int((np.abs(codes).sum(-1) == 0).sum()), 'bytes', os.path.getsize(path))


def aux_cases():
    import torch
    from oracle import aux_oracle as A
    from oracle import sean_oracle as O
    # ---- shape branch: encode + decode on blocky and face-like label maps (incl. 255 'no class' pixels)
    G = R.make_shape_generator()
    sd = P.shape_state_dict(0)
    G.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    lab = np.stack([P.blocky_labels(1, 256, seed=61)[0], face_like_labels(256, 62), face_like_labels(256, 63)])
    lab[0, :32, :48] = 255
    with torch.no_grad():
        hair, face = A.split_hair_face(A.label_to_onehot19(lab))
        hc = G.forward_hair_encoder(hair, testing=True)
        fc = G.forward_face_encoder(face)
        hl = G.forward_hair_decoder(hc, fc)
        fl = G.forward_face_decoder(fc)
        probs = G.forward_decoder(hl, fl)
        out = torch.argmax(probs, dim=1).to(torch.uint8)
        top2 = torch.topk(probs, 2, dim=1).values
    np.savez_compressed(os.path.join(HERE, 'shape_054.npz'), labels=lab, hair_code=hc.numpy(), face_code=fc.numpy(),
                        hair_logit_sub4=hl.numpy()[:, :, ::4, ::4], face_logit_sub4=fl.numpy()[:, :, ::4, ::4],
                        face_logit_crop=fl.numpy()[:, :, 96:160, 96:160], out_labels=out.numpy(),
                        margin=(top2[:, 0] - top2[:, 1]).numpy().astype(np.float16))
    print('shape classes', np.unique(out.numpy()), 'min margin', float((top2[:, 0] - top2[:, 1]).min()))
    # ---- colour MLPs
    S = R.make_color_solver()
    cs = P.color_state_dicts(0)
    for nm, mod in (('gen', S.gen), ('dis', S.dis), ('rgb', S.rgb_model)):
        mod.load_state_dict({k: torch.from_numpy(v) for k, v in cs[nm].items()})
    code = torch.from_numpy(P.style_codes(5, seed=71)[:, 13])
    with torch.no_grad():
        d = S.dis({'code': code})
        r = S.rgb_model({'code': code})
        data = {'noise': d['noise'], 'noise_curliness': d['noise_curliness'], 'rgb_mean': r['rgb_mean'], 'pca_std': r['pca_std']}
        g = S.gen(data)['code']
        ei = S.edit_infer(code, {'noise_curliness': torch.full((5, 1), 1.0), 'rgb_mean': r['rgb_mean'], 'pca_std': r['pca_std']})
    np.savez_compressed(os.path.join(HERE, 'color_045.npz'), code=code.numpy(), noise=d['noise'].numpy(),
                        noise_curliness=d['noise_curliness'].numpy(), adv=d['adv'].numpy(), rgb_mean=r['rgb_mean'].numpy(),
                        pca_std=r['pca_std'].numpy(), gen_code=g.numpy(), edit_infer=ei.numpy())
    print('color rgb', r['rgb_mean'][0].numpy(), 'gen std', float(g.std()))
    # ---- BiSeNet
    N = R.make_bisenet()
    bs = P.bisenet_state_dict(0)
    N.load_state_dict({k: torch.from_numpy(v) for k, v in bs.items()})
    for name, Bn, Hn, seed in (('256', 2, 256, 81), ('512', 1, 512, 82)):
        img = P.synthetic_images(Bn, Hn, seed=seed)
        with torch.no_grad():
            lg = N(torch.from_numpy(img))[0]
        lut = np.array(A.BISENET_TO_CELEBA, np.uint8)
        top2 = torch.topk(lg, 2, dim=1).values
        np.savez_compressed(os.path.join(HERE, f'bisenet_{name}.npz'), meta_B=np.array(Bn), meta_S=np.array(Hn),
                            meta_seed=np.array(seed), logits_sub8=lg.numpy()[:, :, ::8, ::8],
                            logits_crop=lg.numpy()[:, :, 100:164, 60:124], labels=lut[lg.argmax(1).numpy()],
                            margin=(top2[:, 0] - top2[:, 1]).numpy().astype(np.float16))
        print('bisenet', name, 'classes', np.unique(lg.argmax(1).numpy()), 'logit std', float(lg.std()))


def pipeline_case(name='pipeline512', S=512, B=2, ngf=64, iseed=11, nseed=93):
    """BASELINE.json configs[2] composed from the REFERENCE's own modules in the order of ui/backend.py:67-175 (Backend
    itself needs cv2 / dlib / trained checkpoints): BiSeNet -> label remap -> shape encoders -> Zencoder -> colour
    predictor / encoder -> sliders (SURVEY.md 8d Config 3) -> colour generator, shape decoder -> SPADEGenerator.
    Stores every stage's output so that each HIP stage can be checked on the reference's inputs of that stage."""
    import torch
    from ctrlhair_amd import hostutil as U
    from ctrlhair_amd.pipeline import DEFAULT_SLIDERS
    from oracle import aux_oracle as A
    from oracle import sean_oracle as O
    w = {'sean': P.sean_state_dict(0, ngf), 'shape': P.shape_state_dict(0), 'color': P.color_state_dicts(0),
         'bisenet': P.bisenet_state_dict(0)}
    img = P.synthetic_images(B, S, seed=iseed)                                  # [-1,1], what preprocess_img yields
    out = {'meta_S': np.array(S), 'meta_B': np.array(B), 'meta_ngf': np.array(ngf), 'meta_iseed': np.array(iseed),
           'meta_nseed': np.array(nseed)}
    with torch.no_grad():
        # my_parsing_util.py:25-47: ToTensor + Normalize, BiSeNet, argmax of output [0]; :50-54 remap to CelebAMask ids
        N = R.make_bisenet()
        N.load_state_dict({k: torch.from_numpy(v) for k, v in w['bisenet'].items()})
        mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
        std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
        lg = N((torch.from_numpy(img) * 0.5 + 0.5 - mean) / std)[0]
        top2 = torch.topk(lg, 2, dim=1).values
        labels = np.array(A.BISENET_TO_CELEBA, np.uint8)[lg.argmax(1).numpy()]
        out['labels'] = labels
        out['labels_low_margin'] = np.packbits(((top2[:, 0] - top2[:, 1]) < 5e-3).numpy())
        # ui/backend.py:79-90 (shape branch at 256: cv2 INTER_NEAREST down-sampling of the label map)
        lab256 = np.stack([U.resize_nearest(l, (256, 256)) for l in labels])
        G = R.make_shape_generator()
        G.load_state_dict({k: torch.from_numpy(v) for k, v in w['shape'].items()})
        hair, face = A.split_hair_face(A.label_to_onehot19(lab256))
        hc = G.forward_hair_encoder(hair, testing=True)
        fc = G.forward_face_encoder(face)
        out['hair_code'], out['face_code'] = hc.numpy(), fc.numpy()
        # :93-105 Zencoder codes, colour statistics, texture / curliness latents
        codes = torch.from_numpy(R.run_zencoder(w['sean'], img, labels))
        out['codes'] = codes.numpy()
        Sv = R.make_color_solver()
        for nm, mod in (('gen', Sv.gen), ('dis', Sv.dis), ('rgb', Sv.rgb_model)):
            mod.load_state_dict({k: torch.from_numpy(v) for k, v in w['color'][nm].items()})
        hairf = codes[:, 13]
        col = Sv.rgb_model({'code': hairf})
        enc = Sv.dis({'code': hairf})
        out['rgb_mean'], out['pca_std'] = col['rgb_mean'].numpy(), col['pca_std'].numpy()
        out['texture'], out['curliness'] = enc['noise'].numpy(), enc['noise_curliness'].numpy()
        hsv = U.rgb_to_hsv_u8(col['rgb_mean'].numpy().astype('uint8')[None])[0]                 # :99-101
        # sliders (SURVEY.md 8d Config 3): change_color (:192-199), change_curliness, change_texture / change_shape (:450-462)
        sl = DEFAULT_SLIDERS
        idx, val = sl['hsv_gaussian']
        hsv[:, idx] = U.DistTranslation().gaussian_to_val(idx, val)
        rgb = torch.from_numpy(U.hsv_to_rgb_u8(hsv[None])[0])                                   # tensor_hsv_to_rgb :108-115
        out['hsv'], out['rgb'] = hsv, rgb.numpy()
        tdirs = torch.from_numpy(U.seeded_directions(2, 8, seed=45))
        sdirs = torch.from_numpy(U.seeded_directions(4, 16, seed=54))
        move = lambda cur, d, v: cur + (v - cur @ d)[:, None] * d[None]
        tex = move(enc['noise'], tdirs[sl['texture'][0]], sl['texture'][1])
        shp = move(hc, sdirs[sl['shape'][0]], sl['shape'][1])
        curl = torch.full_like(enc['noise_curliness'], sl['curliness'])
        feat = Sv.gen({'noise': tex, 'noise_curliness': curl, 'rgb_mean': rgb, 'pca_std': col['pca_std']})['code']   # :162-169
        out['feature'] = feat.numpy()
        probs = G.forward_decode_by_code(shp, fc)                                                # :304-315
        t2 = torch.topk(probs, 2, dim=1).values
        mask = torch.argmax(probs, dim=1).to(torch.uint8).numpy()
        out['mask'] = mask
        out['mask_low_margin'] = np.packbits(((t2[:, 0] - t2[:, 1]) < 5e-3).numpy())
        codes2 = codes.clone()
        codes2[:, 13] = feat                                                                     # :170
        med = torch.from_numpy(median_codes())
        zero = (codes2 == 0).all(dim=2, keepdim=True)                                            # hair_editor.py:165-168
        codes2 = torch.where(zero, med[None].expand_as(codes2), codes2)
        lab_up = np.stack([U.resize_nearest(m, (S, S)) for m in mask])
        nz = P.noise_planes(B, S, ngf, seed=nseed)
        image = R.run_generator(w['sean'], lab_up, codes2.numpy(), nz, ngf, ui_mode=False)
    out.update(summarize(image))
    path = os.path.join(HERE, f'{name}.npz')
    np.savez_compressed(path, **out)
    print(name, 'image std', image.std(), 'hair px', int((mask == 13).sum()), 'classes', np.unique(labels), 'bytes',
          os.path.getsize(path))


def main():
    if 'pipeline' in sys.argv[1:]:
        pipeline_case()
        return
    if 'b2' in sys.argv[1:]:
        case('ngf64_S512_B2', 64, 512, 2, False, lseed=41, cseed=42, nseed=43)
        return
    if 'aux' in sys.argv[1:] or len(sys.argv) == 1:
        aux_cases()
    if 'zenc' in sys.argv[1:] or len(sys.argv) == 1:
        zenc_case('S64_B2', 64, 2, grid=8)
        zenc_case('S256_face', 256, 1, labels='face', lseed=41, iseed=42)
        zenc_case('S512_B1', 512, 1, lseed=51, iseed=52)
    if len(sys.argv) > 1 and 'gen' not in sys.argv[1:]:
        return
    case('ngf16_S64_B3', 16, 64, 3, False, grid=8)
    case('ngf16_S64_ui', 16, 64, 1, True, grid=8, lseed=5, cseed=6, nseed=8)
    case('ngf16_S128_face', 16, 128, 2, False, labels='face', codes='median', lseed=11, nseed=12)
    case('ngf64_S256_ui', 64, 256, 1, True)
    case('ngf64_S256_face_B2', 64, 256, 2, False, labels='face', codes='median', lseed=21, nseed=22)
    case('ngf64_S512_ui', 64, 512, 1, True, lseed=31, cseed=32, nseed=33)
    case('ngf64_S512_B2', 64, 512, 2, False, lseed=41, cseed=42, nseed=43)


if __name__ == '__main__':
    main()
