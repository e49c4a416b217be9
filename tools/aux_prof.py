"""Kernel-level profile target for the three auxiliary networks at B=8 (run under rocprofv3 --kernel-trace --stats)."""
import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ctrlhair_amd import procedural as P
from ctrlhair_amd.hair_editor import HipModels, procedural_weights
B = 8
w = procedural_weights(0, 64)
w['sean'] = P.sean_state_dict(0, 16)
m = HipModels(w, device=0, img_size=64, max_batch=B)
img = torch.from_numpy(P.synthetic_images(B, 512)).cuda()
which = sys.argv[1] if len(sys.argv) > 1 else 'all'
for _ in range(3):
    if which in ('all', 'bisenet'):
        lab, _ = m.face_parsing.parse_tensor(img)
    else:
        lab = torch.zeros(B, 512, 512, dtype=torch.uint8, device='cuda')
    l256 = lab[:, ::2, ::2].contiguous()
    if which in ('all', 'shape'):
        h, f = m.mask_generator.encode_labels(l256)
        m.mask_generator.decode_labels(h, f)
torch.cuda.synchronize()
