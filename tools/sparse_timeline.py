#!/usr/bin/env python3
"""Per-wave cycle breakdown of conv_ace_sparse_kernel (option sean.dbg bit 256): lane 0 of every wave stamps s_memtime at block
start / after the prologue barrier / after the k-loop / after the epilogue of ONE ACE launch of the B=16, 512x512 exact-f32 pass.

    python tools/sparse_timeline.py [ace_index=16] [labels=blocky|face] > profiles/<name>.md
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ctrlhair_amd import procedural as P                      # noqa: E402
from ctrlhair_amd.sean.generator import SeanGenerator         # noqa: E402


def main():
    sel = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    kind = sys.argv[2] if len(sys.argv) > 2 else 'blocky'
    B, S, ngf = 16, 512, 64
    gen = SeanGenerator(0, f16x3=0).load_state_dict(P.sean_state_dict(0, ngf), max_batch=B, max_size=S)
    dev = gen.device
    labn = P.blocky_labels(B, S) if kind == 'blocky' else np.stack([P.face_like_labels(S, 500 + b) for b in range(B)])
    lab = torch.from_numpy(labn).to(dev)
    cd = torch.from_numpy(P.style_codes(B)).to(dev)
    nz = torch.from_numpy(P.noise_planes(B, S, ngf)).to(dev)
    gen.generate(lab, cd, nz)
    gen.handle.set_option('sean.dbg_sel', sel)
    gen.handle.set_option('sean.dbg', 256 | (int(sys.argv[3]) if len(sys.argv) > 3 else 0))
    nmax = 32768 * 4
    zero = np.zeros(nmax * 5, np.int64)
    for _ in range(2):
        gen.generate(lab, cd, nz)
    torch.cuda.synchronize()
    buf = np.zeros(nmax * 5, np.int64)
    gen.handle._check(gen.handle.lib.ch_sean_debug_read(gen.handle._h, buf.ctypes.data_as(C.c_void_p), buf.nbytes), 'debug_read')
    st = buf.reshape(nmax, 5)
    # (the scratch holds other kernels' data beyond the stamped blocks: keep plausible records only)
    ok = (st[:, 4] >= 0) & (st[:, 4] <= 4) & (st[:, 0] > 0) & (st[:, 1] >= st[:, 0]) & (st[:, 2] >= st[:, 1]) & (st[:, 3] >= st[:, 2]) & \
        (st[:, 3] - st[:, 0] < 10 ** 8)
    if ok.any():
        last = st[ok, 0].max()                       # the most recent stamped launch only
        ok &= (last - st[:, 0] < 10 ** 8) & (last - st[:, 0] >= 0)
    print(f'(records: plausible NSUB {int(((st[:, 4] >= 0) & (st[:, 4] <= 4) & (st[:, 0] > 0)).sum())}, kept {int(ok.sum())})', file=sys.stderr)
    if not ok.any():
        print(st[:8], file=sys.stderr)
        return
    st = st[ok]
    t0 = st[:, 0].min()
    span = st[:, 3].max() - t0
    print(f'# conv_ace_sparse_kernel cycle stamps (s_memtime), ACE launch index {sel}, {kind} labels (B={B}, {S}x{S}, ngf={ngf})\n')
    print(f'waves stamped {len(st)}; launch span {span} counts\n')
    print('| NSUB | waves | prologue (stage chunk 0 + barrier) | k-loop | epilogue (incl. store drain) | total | epilogue share |')
    print('|---|---|---|---|---|---|---|')
    for ns in sorted(set(st[:, 4])):
        m = st[st[:, 4] == ns]
        pro, kl, ep = m[:, 1] - m[:, 0], m[:, 2] - m[:, 1], m[:, 3] - m[:, 2]
        tot = np.median(pro) + np.median(kl) + np.median(ep)
        print(f'| {ns} | {len(m)} | {np.median(pro):.0f} | {np.median(kl):.0f} | {np.median(ep):.0f} | {tot:.0f} | {100 * np.median(ep) / tot:.1f} % |')
    # concurrency: sum of wave busy time / (span * resident wave slots)
    busy = float((st[:, 3] - st[:, 0]).sum())
    print(f'\nsum of wave lifetimes / launch span = {busy / span:.0f} concurrent waves (256 CUs x 8 = 2048 slots)')


if __name__ == '__main__':
    main()
