#!/bin/bash
exec < /dev/null
cd /root/repo
mkdir -p gpurun_out/w4s
for a in bench 1 3 4 8 12 16 32 63 bench2; do
  if [ $a = bench ] || [ $a = bench2 ]; then b=tools/wino4_bench.bin; else b=tools/w4s_abl$a.bin; fi
  timeout 200 $b > gpurun_out/w4s/abl_$a.txt 2>&1
done
ls -la gpurun_out/w4s
