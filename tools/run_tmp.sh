#!/bin/bash
exec < /dev/null
cd /root/repo
timeout 900 python -m pytest tests/test_backend.py -x -q -m gpu 2>&1 | tail -4
for m in overlap overlap; do
  echo "== f32 $m"; timeout 200 python tools/edit_profile.py f32 $m 2>&1 | tail -3
done
