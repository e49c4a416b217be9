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
    # dispatches shorter than 20 us of the conv kernels are the ACE kernels' second passes (return at once, csrc/sh16.h): left out
    q = ("select kernel_name, grid_size, count(*), avg(value), sum(value), avg(duration) from counters_collection "
         "where counter_name=? and not (kernel_name like '%conv_sh16%' and duration <= 20000) group by kernel_name, grid_size")
    return {(short(r[0]), r[1]): r[2:] for r in c.execute(q, (counter,))}


def calibration(d):
    """MFMA-only loop (tools/mfma_peak.hip) profiled with the same counters: returns (busy / GUI_ACTIVE of its longest
    dispatches, its measured fraction of the 2.5 PFLOP/s dense peak) or None."""
    db, log = os.path.join(d, 'peak', 't_results.db'), os.path.join(d, 'peak.log')
    if not (os.path.exists(db) and os.path.exists(log)):
        return None
    c = sqlite3.connect(db)
    q = ("select a.value, b.value from counters_collection a join counters_collection b on a.dispatch_id = b.dispatch_id "
         "where a.counter_name='SQ_VALU_MFMA_BUSY_CYCLES' and b.counter_name='GRBM_GUI_ACTIVE' and a.kernel_name like '%mfma_loop%' "
         "order by b.value desc limit 3")
    try:
        rows = list(c.execute(q))
    except Exception:
        return None
    tf = [float(m.group(1)) for m in re.finditer(r'([0-9.]+) TFLOP/s', open(log).read())]
    if not rows or not tf:
        return None
    r0 = sum(a / b for a, b in rows) / len(rows)
    return r0, max(tf[-3:]) / 2500.0, max(tf[-3:])


def main(d):
    cal = calibration(d)
    f = per_kernel(os.path.join(d, 'fetch', 't_results.db'), 'FETCH_SIZE')
    w = per_kernel(os.path.join(d, 'write', 't_results.db'), 'WRITE_SIZE')
    sq = {}
    for name in ('SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_WAVE_CYCLES', 'SQ_LDS_BANK_CONFLICT',
                 'SQ_LDS_IDX_ACTIVE', 'GRBM_GUI_ACTIVE'):
        sq[name] = per_kernel(os.path.join(d, 'sq', 't_results.db'), name)
    print('# rocprofv3 PMC summary (separate --pmc passes: FETCH_SIZE | WRITE_SIZE | SQ/GRBM)\n')
    print('Per dispatch averages.  `fetch x2` = FETCH_SIZE doubled (gfx950 128-B request correction, upper bound).\n')
    if cal:
        print(f'MFMA pipe busy is CALIBRATED: SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE of each kernel divided by the same ratio of an '
              f'MFMA-only loop (tools/mfma_peak.hip, ratio {cal[0]:.1f}, measured {cal[2]:.0f} TFLOP/s = {100 * cal[1]:.1f} % of the 2.5 PFLOP/s '
              f'dense peak in the same profiled run), times that loop\'s fraction of peak -- i.e. the share of the matrix pipe\'s issue slots '
              f'the kernel filled, whatever the operand type: 100 % = 157.3 TFLOP/s for the kernels built on f32 MFMAs (wino_*, pw_conv, '
              f'conv_mfma, conv_ace_sparse), 2.5 PFLOP/s for those on f16 / bf16 MFMAs (conv_sh16*).\n')
    print("| kernel | grid (threads) | calls | fetch MB | fetch x2 MB | write MB | MFMA pipe busy, % of the pipe's issue slots (calibrated; f32-MFMA kernels: of 157.3 TFLOP/s, f16: of 2.5 PFLOP/s) | LDS conflict % | avg ms (pmc pass) |")
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
            util = f'{100.0 * (busy[1] / gui[1]) / cal[0] * cal[1]:.1f}' if cal else f'raw {busy[1] / gui[1]:.1f}'
        lc = sq['SQ_LDS_BANK_CONFLICT'].get(k)
        la = sq['SQ_LDS_IDX_ACTIVE'].get(k)
        lds = f'{100.0 * lc[1] / la[1]:.1f}' if lc and la and la[1] > 0 else ''
        print(f'| `{k[0]}` | {k[1]} | {n} | {favg / 1024:.1f} | {2 * favg / 1024:.1f} | {wavg / 1024:.1f} | {util} | {lds} | {dur / 1e6:.3f} |')


if __name__ == '__main__':
    main(sys.argv[1])
