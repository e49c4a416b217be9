#!/usr/bin/env python3
"""Kernel-level view of one Config-1 edit on the HIP Backend (ui/backend.py:147-175: set_input_img + slider moves + output(), one 256x256
portrait, batch 1):

  run (GPU box):  rocprofv3 --kernel-trace -d D -o t -- python tools/edit_trace.py run [f32|f16x3]
  summarise:      python tools/edit_trace.py summary D/t_results.db > profiles/rNN_edit_trace_<path>.md

`run` puts a marker kernel (torch.cuda._sleep) ahead of every Backend call; `summary` cuts the trace at the markers and prints, per call,
the number of dispatches, the sum of kernel time, the span from the first kernel's start to the last one's end (= what the GPU needs,
launch gaps included) and the largest kernels."""
import collections
import os
import re
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REPS = 6
CALLS = ('set_input_img', 'change_curliness', 'change_texture', 'change_shape', 'output')


def run(path):
    import numpy as np
    import torch
    from ctrlhair_amd import procedural as P
    from ctrlhair_amd.hair_editor import procedural_weights
    from ctrlhair_amd.ui.backend import Backend
    be = Backend(2.5, blending=False, weights=procedural_weights(0, 64), device=0, f16x3=(path == 'f16x3'))
    img = np.ascontiguousarray(P.synthetic_images(1, 256, seed=11)[0].transpose(1, 2, 0))
    img = np.clip((img * 0.5 + 0.5) * 255.0, 0, 255).astype(np.uint8)
    steps = [lambda: be.set_input_img(img_rgb=img), lambda: be.change_curliness(1.0), lambda: be.change_texture(1.5, 0),
             lambda: be.change_shape(-1.0, 0), lambda: be.output()]
    for _ in range(3):
        for fn in steps:
            fn()
    torch.cuda.synchronize()
    torch.cuda._sleep(1000)
    torch.cuda._sleep(1000)                    # two markers in a row = start of the measured part
    for _ in range(REPS):
        for fn in steps:
            torch.cuda.synchronize()
            torch.cuda._sleep(1000)
            fn()
    torch.cuda.synchronize()


def short(n):
    n = re.sub(r'\(.*$', '', n)
    n = n.replace('void ', '').replace('chk::', '')
    return n if len(n) < 90 else n[:87] + '...'


def summary(db):
    c = sqlite3.connect(db)
    rows = c.execute('select name, start, end from kernels order by start').fetchall()
    mark = [('spin_kernel' in r[0] or '_sleep' in r[0]) for r in rows]
    first = next(i for i in range(len(rows) - 1) if mark[i] and mark[i + 1]) + 2
    segs, cur = [], None
    for r, m in zip(rows[first:], mark[first:]):
        if m:
            cur = []
            segs.append(cur)
        elif cur is not None:
            cur.append(r)
    assert len(segs) == REPS * len(CALLS), (len(segs), REPS, len(CALLS))
    print(f'# Kernel trace of one Config-1 edit on the HIP Backend (256x256 portrait, batch 1; {REPS} repetitions; rocprofv3 --kernel-trace)\n')
    print('| call | dispatches | kernel time ms | GPU span ms (first start .. last end) | busy % |')
    print('|---|---|---|---|---|')
    tot = [0.0, 0.0]
    per = {}
    for ci, name in enumerate(CALLS):
        ss = [segs[r * len(CALLS) + ci] for r in range(REPS)]
        n = sum(len(s) for s in ss) / REPS
        kt = sum(sum(e - st for _, st, e in s) for s in ss) / REPS / 1e6
        sp = sum((max(e for _, _, e in s) - min(st for _, st, _ in s)) if s else 0 for s in ss) / REPS / 1e6
        tot[0] += kt
        tot[1] += sp
        print(f'| {name} | {n:.0f} | {kt:.3f} | {sp:.3f} | {100 * kt / sp if sp else 0:.0f} |')
        d = collections.OrderedDict()
        for s in ss:
            for nm, st, e in s:
                v = d.setdefault(short(nm), [0, 0])
                v[0] += 1
                v[1] += e - st
        per[name] = d
    print(f'| all five | | {tot[0]:.3f} | {tot[1]:.3f} | |\n')
    for name in ('set_input_img', 'change_shape', 'output'):
        print(f'## {name}: largest kernels (per call)\n')
        print('| kernel | calls | ms | avg us |')
        print('|---|---|---|---|')
        for k, (cnt, dur) in sorted(per[name].items(), key=lambda kv: -kv[1][1])[:14]:
            print(f'| `{k}` | {cnt / REPS:.1f} | {dur / REPS / 1e6:.3f} | {dur / cnt / 1e3:.1f} |')
        print()


if __name__ == '__main__':
    if sys.argv[1] == 'run':
        run(sys.argv[2] if len(sys.argv) > 2 else 'f32')
    else:
        summary(sys.argv[2])
