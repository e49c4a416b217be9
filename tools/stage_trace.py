#!/usr/bin/env python3
"""Per-stage kernel trace of the Config-3 edit pipeline (BASELINE.json configs[2]: B=8, 512x512, ngf=64).

  run (on the GPU box):   rocprofv3 --kernel-trace -d D -o t -- python tools/stage_trace.py run [--f16x3 1]
  summarise:              python tools/stage_trace.py summary D/t_results.db > profiles/rNN_aux_kernel_trace.md

`run` executes EditPipeline.stage_times with a marker kernel (torch.cuda._sleep) ahead of every stage; `summary` orders
the trace by start time, cuts it at the markers and prints, per stage, the kernels with calls / total / average per
pipeline step."""
import argparse
import collections
import os
import re
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REPS = 5


def run(args):
    import torch
    from ctrlhair_amd.pipeline import EditPipeline
    pipe = EditPipeline(img_size=512, max_batch=8, f16x3=args.f16x3)
    g = torch.Generator(device='cpu').manual_seed(3)
    img = (torch.rand(8, 3, 512, 512, generator=g) * 2 - 1).cuda()
    pipe.stage_times(img, reps=2)                                   # warm-up (lazy kernel attributes, allocator)
    torch.cuda.synchronize()
    torch.cuda._sleep(1000)                                         # a second marker in a row = start of the measured part
    t = pipe.stage_times(img, reps=REPS, before=lambda name: torch.cuda._sleep(1000))
    print({k: round(v, 3) for k, v in t.items()})


def short(n):
    n = re.sub(r'\(.*$', '', n)
    n = n.replace('void ', '').replace('chk::', '')
    return n if len(n) < 110 else n[:107] + '...'


def summary(args):
    from ctrlhair_amd.pipeline import EditPipeline
    stages = EditPipeline.STAGES
    c = sqlite3.connect(args.db)
    rows = c.execute('select name, start, end - start, grid_x / workgroup_x, lds_size, vgpr_count from kernels order by start').fetchall()
    is_mark = [('spin_kernel' in r[0] or '_sleep' in r[0]) for r in rows]
    # measured part starts at the first pair of consecutive markers
    first = next(i for i in range(len(rows) - 1) if is_mark[i] and is_mark[i + 1]) + 1
    per = [collections.OrderedDict() for _ in stages]
    k = -1
    for r, mk in zip(rows[first:], is_mark[first:]):
        if mk:
            k += 1
            continue
        d = per[k % len(stages)].setdefault((short(r[0]), r[3], r[4], r[5]), [0, 0])
        d[0] += 1
        d[1] += r[2]
    assert k + 1 == REPS * len(stages), (k, REPS, len(stages))
    print(f'# Per-stage kernel trace of the edit pipeline (B=8, 512x512, ngf=64; {REPS} steps; rocprofv3 --kernel-trace)\n')
    print('Cut at marker kernels launched ahead of every stage (tools/stage_trace.py).  ms / us are per pipeline step.\n')
    for name, tab in zip(stages, per):
        tot = sum(v[1] for v in tab.values()) / REPS / 1e6
        n = sum(v[0] for v in tab.values()) / REPS
        print(f'## {name}: {tot:.3f} ms of kernel time in {n:.0f} dispatches per step\n')
        print('| kernel | grid | LDS B | VGPR | calls/step | ms/step | avg us | % of stage |')
        print('|---|---|---|---|---|---|---|---|')
        for (kn, g, l, v), (cnt, dur) in sorted(tab.items(), key=lambda kv: -kv[1][1])[:args.top]:
            print(f'| `{kn}` | {g} | {l} | {v} | {cnt / REPS:.1f} | {dur / REPS / 1e6:.3f} | {dur / cnt / 1e3:.1f} | '
                  f'{100 * dur / REPS / 1e6 / tot:.1f} |')
        print()


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest='cmd', required=True)
    a = sub.add_parser('run')
    a.add_argument('--f16x3', type=int, default=1)
    b = sub.add_parser('summary')
    b.add_argument('db')
    b.add_argument('--top', type=int, default=14)
    args = ap.parse_args()
    run(args) if args.cmd == 'run' else summary(args)
