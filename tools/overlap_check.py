# overlap mode + dynamic task claiming at the benchmark size: B = 16 / 512^2 / ngf = 64 against the serial schedule, several label sets, repeated
import numpy as np, torch, sys
sys.path.insert(0, '.')
from ctrlhair_amd import procedural as P
from ctrlhair_amd.sean.generator import SeanGenerator
ngf, B, S = 64, 16, 512
sd = P.sean_state_dict(0, ngf)
codes, noise = P.style_codes(B), P.noise_planes(B, S, ngf)
labs = {'blocky': P.blocky_labels(B, S), 'face': np.stack([P.face_like_labels(S, 500 + b) for b in range(B)])}
def run(g, lab):
    dev = g.device
    o = g.generate(torch.from_numpy(lab).to(dev), torch.from_numpy(codes).to(dev), torch.from_numpy(noise).to(dev))
    torch.cuda.synchronize()
    return o.cpu().numpy()
ser = SeanGenerator(0, f16x3=0, options={'sean.overlap': 0}).load_state_dict(sd, max_batch=B, max_size=S)
ref = {k: run(ser, v) for k, v in labs.items()}
ser.handle.close()
ov = SeanGenerator(0, f16x3=0, options={'sean.overlap': 64}).load_state_dict(sd, max_batch=B, max_size=S)
bad = 0
for rep in range(6):
    for k, v in labs.items():
        got = run(ov, v)
        eq = np.array_equal(got, ref[k])
        if not eq: bad += 1; print('MISMATCH', rep, k, float(np.abs(got - ref[k]).max()))
print('overlap + dynamic claiming vs serial: %d mismatching runs of 12' % bad)
