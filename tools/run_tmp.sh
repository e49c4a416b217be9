#!/bin/bash
exec < /dev/null
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/t_full.log
cat gpurun_out/t_full.log
timeout 2700 bash tools/profile_round.sh r06 > gpurun_out/profile_round.log 2>&1
echo "profile_round rc=$?"
timeout 700 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
echo "bench rc=$?"
timeout 60 python tools/bench_brief.py < gpurun_out/bench_default.json
