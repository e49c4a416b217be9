#!/bin/bash
exec < /dev/null
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_sparse_ace.py -x -q -s -k "straight_edge or sparse_equals_dense" 2>&1 | tail -60 > gpurun_out/t_edge16.log
for p in bf16 f16 f16x3; do
  timeout 300 python bench.py --only-headline --path $p --batch 32 --steps 6 --warmup 2 --no-cpu-b16 > gpurun_out/b_$p.json 2> gpurun_out/b_$p.err
done
timeout 300 python bench.py --only-headline --path bf16 --batch 16 --steps 6 --warmup 2 --no-cpu-b16 > gpurun_out/b_bf16_16.json 2>> gpurun_out/b_bf16.err
cat gpurun_out/t_edge16.log
for f in gpurun_out/b_bf16.json gpurun_out/b_f16.json gpurun_out/b_f16x3.json gpurun_out/b_bf16_16.json; do python -c "
import json,sys
l=[x for x in open('$f') if x.startswith('{')]
j=json.loads(l[-1]); print('$f', j['value'], j['ms_per_step'])
"; done
