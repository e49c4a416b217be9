#!/bin/bash
# kernel traces of the two builds on one box: tools/ab_trace.sh <other .so>
set -u
OTHER=$1
R=$PWD; export PYTHONPATH=$R
L=$R/ctrlhair_amd/libctrlhair_hip.so
cp $L /tmp/lib_a.so
cd /tmp && export TMPDIR=/tmp
for T in A B; do
  if [ $T = A ]; then cp /tmp/lib_a.so $L; else cp $R/$OTHER $L; fi
  rm -rf /tmp/pt; rocprofv3 --kernel-trace --stats -d /tmp/pt -o t -- python $R/bench.py --only-headline --no-cpu-baseline --steps 5 --warmup 2 > /tmp/pt.log 2>&1
  echo "== $T"; python $R/tools/rocprof_summary.py /tmp/pt/t_results.db | sed -n 3,24p
done
cp /tmp/lib_a.so $L
