#!/bin/bash
# Run on the GPU box (via gpurun): kernel trace + separate PMC passes of the SAME bench command.
# Usage: tools/profile_bench.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/{trace,fetch,write,sq}/...
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-strict-fp32 $*"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $R/bench.py $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o t -- python $R/bench.py $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o t -- python $R/bench.py $ARGS > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace -d $OUT/sq -o t -- python $R/bench.py $ARGS > $OUT/sq.log 2>&1
# calibration of the MFMA-busy counter: the same counters on an MFMA-only loop whose TFLOP/s is known (tools/mfma_peak.hip)
if [ -x $R/tools/mfma_peak.bin ]; then
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $OUT/peak -o t -- $R/tools/mfma_peak.bin > $OUT/peak.log 2>&1
fi
for d in trace fetch write sq; do tail -1 $OUT/$d.log | cut -c1-300; done
ls -la $OUT/*/
