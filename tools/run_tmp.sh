#!/bin/bash
exec < /dev/null
cd /root/repo
mkdir -p gpurun_out
for L in face blocky; do for o in 64 128; do
  timeout 300 python bench.py --only-headline --steps 20 --warmup 5 --no-cpu-baseline --labels $L --opt sean.wino4_ace=$o 2>/dev/null > gpurun_out/w4a_${L}_$o.json
  python - <<PY
import json
l=[x for x in open('gpurun_out/w4a_${L}_$o.json') if x.startswith('{')]
j=json.loads(l[-1]); print('$L wino4_ace=$o', j['value'], j['ms_per_step'])
PY
done; done
