#!/bin/bash
exec < /dev/null
cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_hip_wino.py -x -q -k "straight_edge" 2>&1 | tail -3
cd ctrlhair_amd/csrc
for a in 5 6 4 5 6; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -DACE_T4_ES_BITS=$a -c ace_sparse.hip -o build/ace_sparse.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libctrlhair_hip.so build/*.o
  for L in blocky face; do
  (cd /root/repo && timeout 300 python bench.py --only-headline --steps 20 --warmup 5 --no-cpu-baseline --labels $L > gpurun_out/es_${a}_$L.json 2> gpurun_out/es_${a}_$L.err)
  python - <<PY
import json
l=[x for x in open('/root/repo/gpurun_out/es_${a}_$L.json') if x.startswith('{')]
j=json.loads(l[-1]); print('ES_BITS $a $L', j['value'], j['ms_per_step'], j['roofline']['interior_pass']['ms_per_step'], j['roofline']['interior_pass']['achieved_gbs'])
PY
  done
done
