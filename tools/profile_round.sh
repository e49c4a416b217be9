#!/bin/bash
# One round's rocprofv3 evidence, summarised ON the GPU box (the raw rocpd databases exceed gpurun's 64 MiB return limit):
#   tools/profile_round.sh <round tag, e.g. r02>   -> gpurun_out/profiles_<tag>/*.md, latest_traffic.json
# Copy the results into profiles/ afterwards.
set -u
exec < /dev/null
TAG=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
for P in f16x3 f32; do
  bash $R/tools/profile_bench.sh ${TAG}_$P --path $P > $OUT/${P}_run.log 2>&1
  D=$R/gpurun_out/prof_${TAG}_$P
  python $R/tools/rocprof_summary.py $D/trace/t_results.db > $OUT/${TAG}_${P}_kernel_trace.md 2>> $OUT/${P}_run.log
  python $R/tools/rocprof_pmc.py $D > $OUT/${TAG}_${P}_pmc.md 2>> $OUT/${P}_run.log
  # the launches bench.py's roofline block averages over: every SPADE conv + fused ACE epilogue of a step
  if [ $P = f16x3 ]; then K='conv_sh16_ws_kernel<3, 32, 16, 1, 1, 3|conv_sh16_kernel<3, 16, 16, 2, 1, 3|conv_sh16_kernel<3, 32, 16, 1, 1, 3'
  else K='wino_ace_gather_kernel|wino4v_kernel<1>|wino4_ace_kernel|wino_ace_kernel|conv_ace_sparse_kernel|conv_mfma_kernel<3, 1, 2, 32, 8, 1, 16, 1,|conv_mfma_kernel<3, 1, 2, 16, 16, 1, 16, 1,'; fi
  python $R/tools/make_traffic.py $D $P "$K" "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), profiles/${TAG}_${P}_pmc.md" >> $OUT/${P}_run.log 2>&1
  # exact f32: also the ResBlock conv set (3x3 as F(4x4,3x3) / F(2x2,3x3) + the 1x1 shortcuts) -- the dominant set once the straight-edge reduction applies
  if [ $P = f32 ]; then python $R/tools/make_traffic.py $D f32_plain 'wino4_plain_kernel|wino4v_kernel<0>|wino_plain_kernel|pw_conv_kernel' "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), profiles/${TAG}_${P}_pmc.md" >> $OUT/${P}_run.log 2>&1; fi
  grep '^{' $D/trace.log | tail -1 > $OUT/${TAG}_${P}_bench_line.json      # (rocprofv3 logs after the JSON line)
  cp $D/peak.log $OUT/${TAG}_${P}_mfma_peak.log 2>/dev/null
  rm -rf $D
done
cp $R/profiles/latest_traffic.json $OUT/latest_traffic.json
cd /tmp && export TMPDIR=/tmp
D=$R/gpurun_out/prof_${TAG}_pipe
rocprofv3 --kernel-trace --stats -d $D/trace -o t -- python $R/bench.py --workload pipeline --path f16x3 --only-headline --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pipe_run.log 2>&1
python $R/tools/rocprof_summary.py $D/trace/t_results.db > $OUT/${TAG}_pipeline_kernel_trace.md 2>> $OUT/pipe_run.log
grep '^{' $OUT/pipe_run.log | tail -1 > $OUT/${TAG}_pipeline_bench_line.json
rm -rf $D
ls -la $OUT
bash $R/tools/stage_trace.sh ${TAG} 1 > $OUT/stage_run.log 2>&1
cp $R/gpurun_out/${TAG}_aux_kernel_trace.md $OUT/${TAG}_aux_kernel_trace.md
bash $R/tools/stage_trace.sh ${TAG} 0 > $OUT/stage_run_f32.log 2>&1          # the exact-f32 aux nets (Config 3's number of record)
cp $R/gpurun_out/${TAG}_aux_f32_kernel_trace.md $OUT/${TAG}_aux_f32_kernel_trace.md
# exact-f32 pipeline trace too
D=$R/gpurun_out/prof_${TAG}_pipe32
cd /tmp
rocprofv3 --kernel-trace --stats -d $D/trace -o t -- python $R/bench.py --workload pipeline --path f32 --only-headline --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pipe32_run.log 2>&1 < /dev/null
python $R/tools/rocprof_summary.py $D/trace/t_results.db > $OUT/${TAG}_pipeline_f32_kernel_trace.md 2>> $OUT/pipe32_run.log < /dev/null
grep '^{' $OUT/pipe32_run.log | tail -1 > $OUT/${TAG}_pipeline_f32_bench_line.json
rm -rf $D
# the traffic file LAST: whatever ran above, the committed figure belongs to the sources of this very snapshot
cp $R/profiles/latest_traffic.json $OUT/latest_traffic.json
ls -la $OUT
