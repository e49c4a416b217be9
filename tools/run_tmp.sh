#!/bin/bash
exec < /dev/null
cd /root/repo
mkdir -p gpurun_out
cd ctrlhair_amd/csrc
for a in 0 512 1024 2048 0; do
  if [ $a = 0 ]; then D=""; else D="-DACE_T4_PAD=$a"; fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result $D -c ace_sparse.hip -o build/ace_sparse.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libctrlhair_hip.so build/*.o
  (cd /root/repo && timeout 300 python bench.py --only-headline --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/pad_$a.json 2>/dev/null)
  python - <<PY
import json
l=[x for x in open('/root/repo/gpurun_out/pad_$a.json') if x.startswith('{')]
j=json.loads(l[-1]); print('PAD floats $a', j['value'], j['ms_per_step'], j['roofline']['interior_pass']['ms_per_step'])
PY
done
