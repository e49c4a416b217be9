#!/bin/bash
exec < /dev/null
cd /root/repo
mkdir -p gpurun_out
cd ctrlhair_amd/csrc
for a in 0 1 2 3 0 2; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -DACE_T4_AHEAD=$a -c ace_sparse.hip -o build/ace_sparse.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libctrlhair_hip.so build/*.o
  (cd /root/repo && timeout 300 python bench.py --only-headline --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ahead_$a.json 2> gpurun_out/ahead_$a.err)
  python - <<PY
import json
l=[x for x in open('/root/repo/gpurun_out/ahead_$a.json') if x.startswith('{')]
j=json.loads(l[-1]); print('AHEAD $a', j['value'], j['ms_per_step'], j['roofline']['interior_pass']['ms_per_step'], j['roofline']['interior_pass']['achieved_gbs'])
PY
done
