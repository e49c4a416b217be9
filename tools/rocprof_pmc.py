#!/usr/bin/env python3
"""Per-kernel PMC summary from rocprofv3 (rocpd sqlite) counter-collection runs.
Usage: rocprof_pmc.py <prof_dir containing fetch/ write/ sq/ subdirs> > profiles/<name>_pmc.md

FETCH_SIZE / WRITE_SIZE are KiB per dispatch.  Per MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 counts
128-byte requests of wide coalesced streams at 64 B -> the corrected read traffic is up to 2x the raw figure; both
are printed.  Counters come from separate passes (TCC slot limits)."""
import os
import re
import sqlite3
import sys


def short(n):
    return re.sub(r'\(.*$', '', n).replace('void ', '').replace('chk::', '')


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    q = ("select kernel_name, grid_size, count(*), avg(value), sum(value), avg(duration) from counters_collection "
         "where counter_name=? group by kernel_name, grid_size")
    return {(short(r[0]), r[1]): r[2:] for r in c.execute(q, (counter,))}


def main(d):
    f = per_kernel(os.path.join(d, 'fetch', 't_results.db'), 'FETCH_SIZE')
    w = per_kernel(os.path.join(d, 'write', 't_results.db'), 'WRITE_SIZE')
    sq = {}
    for name in ('SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_WAVE_CYCLES', 'SQ_LDS_BANK_CONFLICT',
                 'SQ_LDS_IDX_ACTIVE', 'GRBM_GUI_ACTIVE'):
        sq[name] = per_kernel(os.path.join(d, 'sq', 't_results.db'), name)
    print('# rocprofv3 PMC summary (separate --pmc passes: FETCH_SIZE | WRITE_SIZE | SQ/GRBM)\n')
    print('Per dispatch averages.  `fetch x2` = FETCH_SIZE doubled (gfx950 128-B request correction, upper bound).\n')
    print('| kernel | grid (threads) | calls | fetch MB | fetch x2 MB | write MB | MFMA busy/(GUI_ACTIVE*1024) % (raw; counter appears to cover 1 of 8 XCDs -> x8) | LDS conflict % | avg ms (pmc pass) |')
    print('|---|---|---|---|---|---|---|---|---|')
    keys = sorted(f, key=lambda k: -f[k][2])
    for k in keys[:40]:
        n, favg, fsum, dur = f[k]
        wavg = w.get(k, (0, 0, 0, 0))[1]
        busy = sq['SQ_VALU_MFMA_BUSY_CYCLES'].get(k)
        gui = sq['GRBM_GUI_ACTIVE'].get(k)
        util = ''
        if busy and gui and gui[1] > 0:
            # MfmaUtil = sum(MFMA busy cycles over SIMDs) / (GUI_ACTIVE * #SIMD); 256 CU * 4 SIMD
            util = f'{100.0 * busy[1] / (gui[1] * 1024):.1f}'
        lc = sq['SQ_LDS_BANK_CONFLICT'].get(k)
        la = sq['SQ_LDS_IDX_ACTIVE'].get(k)
        lds = f'{100.0 * lc[1] / la[1]:.1f}' if lc and la and la[1] > 0 else ''
        print(f'| `{k[0]}` | {k[1]} | {n} | {favg / 1024:.1f} | {2 * favg / 1024:.1f} | {wavg / 1024:.1f} | {util} | {lds} | {dur / 1e6:.3f} |')


if __name__ == '__main__':
    main(sys.argv[1])
