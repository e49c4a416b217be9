#!/usr/bin/env python3
"""Update profiles/latest_traffic.json (read by bench.py for roofline.traffic) from a tools/profile_bench.sh run.

    tools/make_traffic.py <prof_dir with fetch/ write/> <path key: f16x3|f32> <kernel name substring> <source note>

Per-launch HBM bytes of the dominant kernel = FETCH_SIZE (KiB, doubled per MI355X_MICROARCH.md's gfx950 128-byte request
correction -- an upper bound) + WRITE_SIZE (KiB), each from its own --pmc pass, averaged over that kernel's launches."""
import json
import os
import sqlite3
import sys


def avg_kib(db, counter, sub):
    c = sqlite3.connect(db)
    # launches shorter than 20 us are the second ("repair") passes of the ACE kernels, which return at once (csrc/sh16.h)
    # `sub`: one or more kernel-name substrings separated by '|' (the launches bench.py's roofline averages over)
    subs = [t for t in sub.split('|') if t]
    cond = ' or '.join('kernel_name like ?' for _ in subs)
    rows = list(c.execute(f"select count(*), avg(value) from counters_collection where counter_name=? and ({cond}) "
                          "and duration > 20000", (counter, *[f'%{t}%' for t in subs])))
    return rows[0]


def main(d, key, sub, note):
    nf, f = avg_kib(os.path.join(d, 'fetch', 't_results.db'), 'FETCH_SIZE', sub)
    nw, w = avg_kib(os.path.join(d, 'write', 't_results.db'), 'WRITE_SIZE', sub)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, 'profiles', 'latest_traffic.json')
    cur = json.load(open(path)) if os.path.exists(path) else {}
    sys.path.insert(0, root)
    import bench
    cur[key] = {'unit': "bytes per launch (avg over the kernel's launches)", 'kernel': sub, 'csrc_sha': bench.csrc_sha(key),
                'fetch_raw': f * 1024, 'fetch_x2_gfx950_corrected': 2 * f * 1024, 'write': w * 1024,
                'hbm_bytes': (2 * f + w) * 1024, 'launches_sampled': nf, 'source': note}
    json.dump(cur, open(path, 'w'), indent=1)
    print(json.dumps(cur[key]))


if __name__ == '__main__':
    main(*sys.argv[1:5])
