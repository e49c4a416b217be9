#!/bin/bash
exec < /dev/null
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/t_full.log
cat gpurun_out/t_full.log
timeout 2700 bash tools/profile_round.sh r06 > gpurun_out/profile_round.log 2>&1
echo "profile_round rc=$?"
export TMPDIR=/tmp
for P in f32 f16x3; do
  D=/tmp/edit_$P; rm -rf $D
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace -d $D -o t -- python /root/repo/tools/edit_trace.py run $P > /root/repo/gpurun_out/edit_trace_$P.log 2>&1)
  timeout 120 python tools/edit_trace.py summary $D/t_results.db > gpurun_out/r06_edit_trace_$P.md 2>> gpurun_out/edit_trace_$P.log
  timeout 200 python tools/edit_profile.py $P 2>&1 | tail -4 > gpurun_out/edit_profile_$P.txt
done
cat gpurun_out/edit_profile_f32.txt
timeout 700 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
echo "bench rc=$?"
timeout 60 python tools/bench_brief.py < gpurun_out/bench_default.json
