"""Debug helper (round 6): where do sean.edge = 1 and sean.edge = 0 differ?  Classifies the offending pixels on the host."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctrlhair_amd import procedural as P
from ctrlhair_amd.sean.generator import SeanGenerator
from ctrlhair_amd.sean import arch

ngf, S, B = 64, 256, 1
sd = P.sean_state_dict(0, ngf)
lab = np.stack([P.face_like_labels(S, 40 + b) for b in range(B)])
codes, noise = P.style_codes(B, seed=71), P.noise_planes(B, S, ngf, seed=72)
outs = {}
taps = {}
for e in (1, 0):
    g = SeanGenerator(0, f16x3=0, options={'sean.edge': e}).load_state_dict(sd, max_batch=B, max_size=S)
    names = ['up_2.xs', 'up_2.dx', 'up_2', 'up_3.xs', 'up_3.dx', 'up_3']
    bufs = {}
    for n in names:
        blk = n.split('.')[0]
        r = S // {'up_1': 4, 'up_2': 2, 'up_3': 1}[blk]
        c = {'up_1': 4 * ngf, 'up_2': 2 * ngf, 'up_3': ngf}[blk]
        bufs[n] = torch.zeros((B, c, r, r), device=g.device)
        g.handle.sean_set_tap(n, bufs[n].data_ptr())
    dev = g.device
    out = g.generate(torch.from_numpy(lab).to(dev), torch.from_numpy(codes).to(dev), torch.from_numpy(noise).to(dev))
    torch.cuda.synchronize()
    outs[e] = out.cpu().numpy()
    taps[e] = {k: v.cpu().numpy() for k, v in bufs.items()}
    g.handle.close()
print('final max diff', np.abs(outs[1] - outs[0]).max())

def classify(l):
    H, W = l.shape
    cls = np.full((H, W), 2, np.uint8)      # 0 interior, 1 edge, 2 boundary
    for y in range(2, H - 2):
        for x in range(2, W - 2):
            w = l[y - 2:y + 3, x - 2:x + 3]
            if (w == w[2, 2]).all() and w[2, 2] < 19:
                cls[y, x] = 0
                continue
            for o in (0, 1):
                ww = w if o == 0 else w.T
                if (ww == ww[0:1, :]).all():
                    ll = ww[0]
                    A, Bl = ll[0], ll[4]
                    if A < 19 and Bl < 19 and A != Bl:
                        s = 1
                        while s < 4 and ll[s] == A:
                            s += 1
                        if all(ll[j] == (A if j < s else Bl) for j in range(5)):
                            cls[y, x] = 1
    return cls
for n in taps[1]:
    d = np.abs(taps[1][n] - taps[0][n])
    print(n, 'max diff', d.max())
    if d.max() > 1e-4:
        blk = n.split('.')[0]
        k = {'up_1': 4, 'up_2': 2, 'up_3': 1}[blk]
        l = lab[0][::k, ::k]
        cls = classify(l)
        bad = d[0].max(0) > 1e-4
        print('  bad pixels', bad.sum(), 'by class (interior, edge, boundary):', [(bad & (cls == c)).sum() for c in range(3)], 'class sizes', [(cls == c).sum() for c in range(3)])
        ys, xs = np.nonzero(bad)
        for y, x in list(zip(ys, xs))[:4]:
            print('  at', y, x, 'cls', cls[y, x], 'chan argmax', d[0][:, y, x].argmax(), 'diff', d[0][:, y, x].max())
            print(l[max(0, y - 2):y + 3, max(0, x - 2):x + 3])
