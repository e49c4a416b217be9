#!/bin/bash
# same-box A/B of two builds of the library: tools/ab_bench.sh <other .so> [bench.py args]   (A = the tree's library, B = the other)
set -u
OTHER=$1; shift
export PYTHONPATH=$PWD
L=ctrlhair_amd/libctrlhair_hip.so
cp $L /tmp/lib_a.so
run() { python bench.py --only-headline --no-cpu-baseline --steps 30 "$@" 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$TAG', r['value'], r['ms_per_step'])"; }
for i in 1 2; do
  TAG=A; cp /tmp/lib_a.so $L; run "$@"
  TAG=B; cp $OTHER $L; run "$@"
done
cp /tmp/lib_a.so $L
