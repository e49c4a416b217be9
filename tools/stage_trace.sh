#!/bin/bash
# tools/stage_trace.sh <tag>  ->  gpurun_out/<tag>_aux_kernel_trace.md  (run on the GPU box; see tools/stage_trace.py)
set -u
TAG=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
D=$R/gpurun_out/prof_${TAG}_stages
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $D -o t -- python $R/tools/stage_trace.py run > $R/gpurun_out/${TAG}_stage_run.log 2>&1
tail -2 $R/gpurun_out/${TAG}_stage_run.log
python $R/tools/stage_trace.py summary $D/t_results.db > $R/gpurun_out/${TAG}_aux_kernel_trace.md
rm -rf $D
