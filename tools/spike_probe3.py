"""Segments of Backend.output() with a device synchronisation behind each (host clock): which one stalls?"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctrlhair_amd import procedural as P
from ctrlhair_amd.hair_editor import procedural_weights
from ctrlhair_amd.ui.backend import Backend
from ctrlhair_amd.hostutil import HAIR_IDX
be = Backend(2.5, blending=False, weights=procedural_weights(0, 64), device=0, f16x3=False)
img = np.ascontiguousarray(P.synthetic_images(1, 256, seed=11)[0].transpose(1, 2, 0))
img = np.clip((img * 0.5 + 0.5) * 255.0, 0, 255).astype(np.uint8)
be.set_input_img(img_rgb=img); be.output()
rows = []
for i in range(30):
    lat, mask = be.cur_latent, be.cur_mask
    t0 = time.time()
    rgb = be.target_latent.color['rgb_mean'] if 'rgb_mean' in lat.color else be.tensor_hsv_to_rgb(lat.color['hsv'])
    torch.cuda.synchronize(); t1 = time.time()
    feature = be.feature_generator({'noise': lat.texture, 'noise_curliness': lat.curliness, 'rgb_mean': rgb, 'pca_std': lat.color['pca_std']})['code']
    torch.cuda.synchronize(); t2 = time.time()
    be.input_sean_code[:, HAIR_IDX] = feature
    m = be._mask_for_sean(mask)[None, None, ...]
    torch.cuda.synchronize(); t3 = time.time()
    rendered = be.gen_img(be.input_sean_code, m, noise=be.noise)
    torch.cuda.synchronize(); t4 = time.time()
    out = be.postprocess_blending(be.input_img, rendered, be.input_mask, mask, blending=False, blender=None)[0]
    t5 = time.time()
    rows.append([(b - a) * 1e3 for a, b in ((t0, t1), (t1, t2), (t2, t3), (t3, t4), (t4, t5))])
print('rgb / feature / code+mask / gen_img / postprocess ms:')
print(' '.join('/'.join(f'{v:.0f}' for v in r) for r in rows))
