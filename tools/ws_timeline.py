#!/usr/bin/env python3
"""Per-tile cycle breakdown of the wave-specialised SPADE/ACE conv kernel (option sean.dbg bit 256): for one ACE launch of the
B=16, 512x512 generator pass, wave 0 of every block stamps s_memtime at tile start, end of the k-loop and end of the epilogue.

    python tools/ws_timeline.py [ace_index=16] > profiles/<name>.md
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
    B, S, ngf = 16, 512, 64
    gen = SeanGenerator(0, f16x3=1).load_state_dict(P.sean_state_dict(0, ngf), max_batch=B, max_size=S)
    dev = gen.device
    lab = torch.from_numpy(P.blocky_labels(B, S)).to(dev)
    cd = torch.from_numpy(P.style_codes(B)).to(dev)
    nz = torch.from_numpy(P.noise_planes(B, S, ngf)).to(dev)
    gen.generate(lab, cd, nz)
    gen.handle.set_option('sean.dbg_sel', sel)
    gen.handle.set_option('sean.dbg', 256 | (int(sys.argv[2]) if len(sys.argv) > 2 else 0))
    for _ in range(2):
        gen.generate(lab, cd, nz)
    torch.cuda.synchronize()
    nblk, nt = 256, 64
    buf = np.zeros(nblk * nt * 3, np.int64)
    gen.handle._check(gen.handle.lib.ch_sean_debug_read(gen.handle._h, buf.ctypes.data_as(C.c_void_p), buf.nbytes), 'debug_read')
    st = buf.reshape(nblk, nt, 3)
    valid = st[:, :, 2] > st[:, :, 0]
    ntile = valid.sum(1)
    main_c = (st[:, :, 1] - st[:, :, 0])[valid]
    epi_c = (st[:, :, 2] - st[:, :, 1])[valid]
    gap = (st[:, 1:, 0] - st[:, :-1, 2])[valid[:, 1:] & valid[:, :-1]]
    span = np.array([st[b, ntile[b] - 1, 2] - st[b, 0, 0] for b in range(nblk) if ntile[b] > 0])
    print(f'# conv_sh16_ws_kernel cycle stamps, ACE launch index {sel} (B={B}, {S}x{S}, ngf={ngf})\n')
    print(f'blocks {int((ntile > 0).sum())}, tiles per block {ntile[ntile > 0].min()}..{ntile.max()} (stamped up to 64)\n')
    print('| phase | median cycles | p10 | p90 | share of tile |')
    print('|---|---|---|---|---|')
    tot = np.median(main_c) + np.median(epi_c) + (np.median(gap) if gap.size else 0)
    for name, v in (('k-loop (MFMA)', main_c), ('epilogue (incl. store drain)', epi_c), ('gap to next tile', gap)):
        if v.size:
            print(f'| {name} | {np.median(v):.0f} | {np.percentile(v, 10):.0f} | {np.percentile(v, 90):.0f} | {100 * np.median(v) / tot:.1f} % |')
    print(f'\nblock span (first tile start -> last stamped tile end): median {np.median(span):.0f} cycles')


if __name__ == '__main__':
    main()
