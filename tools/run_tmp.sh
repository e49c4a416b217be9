#!/bin/bash
exec < /dev/null
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_zencoder.py -x -q -s 2>&1 | tail -25 > gpurun_out/t_zenc.log
cat gpurun_out/t_zenc.log
for o in sean.convt_gemm=1 sean.convt_gemm=0; do
  timeout 400 python bench.py --workload pipeline --path f32 --only-headline --steps 12 --warmup 3 --no-cpu-baseline --opt $o > gpurun_out/p_$o.json 2> gpurun_out/p_$o.err
  python - <<PY
import json
l=[x for x in open('gpurun_out/p_$o.json') if x.startswith('{')]
if l:
    j=json.loads(l[-1]); st=j.get('stages')
    print('$o', j['value'], j['ms_per_step'], {k:v['ms'] for k,v in (st or {}).items()})
else:
    print('$o', 'no line'); print(open('gpurun_out/p_$o.err').read()[-800:])
PY
done
