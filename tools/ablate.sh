#!/bin/bash
# perf ablations of the f16x3 conv kernels via the library's sean.dbg switches
for d in ${@:-0 64 68}; do
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --only-headline --path f16x3 --dbg $d 2>/dev/null | tail -1 > /tmp/ab.json
python - <<PY
import json
d=json.load(open('/tmp/ab.json'))
r=d['roofline']
print('dbg', $d, 'step_ms', d['ms_per_step'], 'ace_ms/step', round(r['avg_launch_ms']*18,2), 'all_conv_ms/step', r['all_mfma_convs']['ms_per_step'], 'plain_tf', r['all_mfma_convs']['plain_algorithmic_tflops'])
PY
done
