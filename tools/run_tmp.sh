#!/bin/bash
exec < /dev/null
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_backend.py tests/test_pipeline.py -x -q -m gpu 2>&1 | tail -8
for m in overlap serial overlap serial; do
  echo "== f32 $m"; timeout 200 python tools/edit_profile.py f32 $m 2>&1 | tail -4
done
for m in overlap serial; do
  echo "== f16x3 $m"; timeout 200 python tools/edit_profile.py f16x3 $m 2>&1 | tail -3
done
