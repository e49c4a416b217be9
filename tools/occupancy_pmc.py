#!/usr/bin/env python3
"""Average resident waves per kernel from a rocprofv3 PMC pass with SQ_WAVE_CYCLES, SQ_BUSY_CYCLES, SQ_WAVES and
GRBM_GUI_ACTIVE: prints SQ_WAVE_CYCLES / GRBM_GUI_ACTIVE (wave-cycles per GPU cycle, i.e. waves resident on the chip, in
the counter's units) per kernel / grid, to compare kernels against one whose occupancy is known.
Usage: occupancy_pmc.py results.db [name filter]"""
import collections
import sqlite3
import sys


def main(db, flt='conv_sh16'):
    c = sqlite3.connect(db)
    rows = c.execute("select dispatch_id, kernel_name, grid_size, counter_name, value from counters_collection "
                     "where kernel_name like ?", (f'%{flt}%',)).fetchall()
    per = collections.defaultdict(dict)
    for d, k, g, n, v in rows:
        per[(d, k, g)][n] = per[(d, k, g)].get(n, 0) + v
    agg = collections.defaultdict(list)
    for (d, k, g), vals in per.items():
        if 'SQ_WAVE_CYCLES' in vals and vals.get('GRBM_GUI_ACTIVE', 0) > 0:
            agg[(k[k.find(flt):][:80], g)].append((vals['SQ_WAVE_CYCLES'] / vals['GRBM_GUI_ACTIVE'], vals.get('SQ_WAVES', 0),
                                                   vals.get('SQ_BUSY_CYCLES', 0) / vals['GRBM_GUI_ACTIVE'], vals['GRBM_GUI_ACTIVE']))
    print('| kernel | grid threads | launches | wave-cycles / GUI cycle | SQ_WAVES | busy / GUI | GUI cycles |')
    print('|---|---|---|---|---|---|---|')
    for (k, g), v in sorted(agg.items(), key=lambda kv: -sum(x[3] for x in kv[1])):
        n = len(v)
        print(f'| `{k}` | {g} | {n} | {sum(x[0] for x in v) / n:.1f} | {sum(x[1] for x in v) / n:.0f} | {sum(x[2] for x in v) / n:.2f} | '
              f'{sum(x[3] for x in v) / n:.0f} |')


if __name__ == '__main__':
    main(*sys.argv[1:])
