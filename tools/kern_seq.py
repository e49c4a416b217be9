#!/usr/bin/env python3
"""Launch-ordered durations of the kernels matching a pattern in a rocprofv3 (rocpd sqlite) trace, last N launches.
Usage: kern_seq.py results.db <name pattern> [N]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rows = c.execute("select name, grid_x/workgroup_x, duration from kernels where name like ? order by start desc limit ?",
                 ('%' + sys.argv[2] + '%', n)).fetchall()
for name, g, d in reversed(rows):
    print(f'{name[:60]:60s} grid {g:7d}  {d / 1e3:9.1f} us')
