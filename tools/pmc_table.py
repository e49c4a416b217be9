#!/usr/bin/env python3
"""Per-kernel averages of the counters in a rocprofv3 --pmc run (rocpd sqlite).  Usage: pmc_table.py results.db [min_us]"""
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
min_ns = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 50e3
names = [r[0] for r in c.execute("select distinct counter_name from counters_collection")]
rows = {}
for k, g, cn, n, v, d in c.execute("select kernel_name, grid_size, counter_name, count(*), avg(value), avg(duration) from counters_collection "
                                   "where duration >= ? group by kernel_name, grid_size, counter_name", (min_ns,)):
    key = (re.sub(r'\(.*$', '', k).replace('void ', '').replace('chk::', ''), g)
    rows.setdefault(key, {'n': n, 'us': d / 1e3})[cn] = v
print('| kernel | grid | n | avg us | ' + ' | '.join(names) + ' |')
print('|---|---|---|---|' + '---|' * len(names))
for key, r in sorted(rows.items(), key=lambda kv: -kv[1]['us'] * kv[1]['n'])[:30]:
    print(f'| `{key[0][:70]}` | {key[1]} | {r["n"]} | {r["us"]:.1f} | ' + ' | '.join(f'{r.get(n, 0):.4g}' for n in names) + ' |')
