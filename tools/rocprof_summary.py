#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average / share, plus, for the
conv kernels, per-(grid,lds) variants.  Usage: rocprof_summary.py results.db > profiles/<name>.md"""
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r'\(.*$', '', n)
    return n.replace('void ', '').replace('chk::', '')


def main(path):
    c = sqlite3.connect(path)
    # the ACE conv kernels are launched twice per layer; the second ("repair") pass returns at once (csrc/sh16.h) and is
    # listed separately so that averages describe real launches
    rows = c.execute("select case when name like '%conv_sh16%' and duration <= 20000 then name || ' [second pass: early exit]' else name end n, "
                     "count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                     "group by n order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows)
    print(f'# rocprofv3 --kernel-trace summary ({path.split("/")[-1]})\n')
    print(f'total kernel time {total / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches\n')
    print('| kernel | calls | total ms | avg us | min us | max us | % |')
    print('|---|---|---|---|---|---|---|')
    for n, k, s, a, mn, mx in rows:
        print(f'| `{short(n)}` | {k} | {s / 1e6:.3f} | {a / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} | {100 * s / total:.2f} |')
    print('\n## conv_mfma_kernel dispatch shapes (grid blocks, LDS bytes, VGPRs)\n')
    print('| kernel | grid | lds | vgpr | calls | avg us |')
    print('|---|---|---|---|---|---|')
    for n, g, l, v, k, a in c.execute("select name, grid_x/workgroup_x, lds_size, vgpr_count, count(*), avg(duration) from kernels "
                                      "where (name like '%conv_mfma%' or name like '%conv_ace_sparse%') group by name, grid_x, lds_size order by avg(duration)*count(*) desc limit 40"):
        print(f'| `{short(n)}` | {g} | {l} | {v} | {k} | {a / 1e3:.1f} |')


if __name__ == '__main__':
    main(sys.argv[1])
