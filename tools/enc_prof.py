import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from ctrlhair_amd import procedural as P
from ctrlhair_amd.sean.generator import SeanGenerator
B, S = 8, 512
g = SeanGenerator(0, f16x3=1).load_state_dict(P.sean_state_dict(0, 64), max_batch=B, max_size=S)
img = torch.from_numpy(P.synthetic_images(B, S)).cuda()
lab = torch.from_numpy(P.blocky_labels(B, S)).cuda()
for _ in range(3):
    g.encode(img, lab)
torch.cuda.synchronize()
