#!/usr/bin/env python3
"""Fill / refresh the measured numbers of DESIGN.md section 6 from bench.py JSON lines:
    tools/fill_design.py <generator bench json> <pipeline bench json>
Placeholders @@NAME@@ are replaced on first use; later runs update the values between the invisible markers it leaves."""
import json
import re
import sys

ROOT = __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))


def main(gen_path, pipe_path):
    g = json.loads(open(gen_path).read().strip().splitlines()[-1])
    p = json.loads(open(pipe_path).read().strip().splitlines()[-1])
    s32 = g['strict_fp32']
    vals = {
        'GEN': f"{g['value']:.1f}", 'GENMS': f"{g['ms_per_step']:.1f}", 'GENTF': f"{g['roofline']['achieved']:.0f}",
        'GENFRAC': f"{100 * g['roofline']['frac']:.1f}", 'F32': f"{s32['value']:.1f}", 'F32MS': f"{s32['ms_per_step']:.1f}",
        'F32TF': f"{s32['roofline']['achieved']:.1f}", 'F32FRAC': f"{100 * s32['roofline']['frac']:.1f}",
        'HBMFRAC': f"{100 * g['roofline']['hbm_contract']['frac']:.1f}",
        'PIPE': f"{p['value']:.1f}", 'PIPEMS': f"{p['ms_per_step']:.1f}", 'PIPEF32': f"{p['strict_fp32']['value']:.1f} images/s",
    }
    st = p.get('stages', {})
    for k, row in st.items():
        vals['ST_' + k] = f"{row['ms']:.2f}" if row['ms'] < 3 else f"{row['ms']:.1f}"
    if st:
        vals['ST_aux'] = f"{sum(r['ms'] for k, r in st.items() if k != 'generator'):.1f}"
    path = ROOT + '/DESIGN.md'
    s = open(path).read()
    for k, v in vals.items():
        s = s.replace(f'@@{k}@@', f'<!--{k}-->{v}<!--/{k}-->')
        s = re.sub(rf'<!--{k}-->.*?<!--/{k}-->', f'<!--{k}-->{v}<!--/{k}-->', s)
    open(path, 'w').write(s)
    print(vals)


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
