#!/bin/bash
# tools/stage_trace.sh <tag> [f16x3 = 1 | 0]  ->  gpurun_out/<tag>_aux_kernel_trace.md (f16x3 legs) / <tag>_aux_f32_kernel_trace.md (exact f32)
# (run on the GPU box; see tools/stage_trace.py)
set -u
TAG=$1
F16=${2:-1}
SUF=$([ "$F16" = 0 ] && echo _f32 || echo "")
R=${GRAFT_REPO_ROOT:-/root/repo}
D=$R/gpurun_out/prof_${TAG}_stages$SUF
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace -d $D -o t -- python $R/tools/stage_trace.py run --f16x3 $F16 > $R/gpurun_out/${TAG}_stage_run$SUF.log 2>&1 < /dev/null
tail -2 $R/gpurun_out/${TAG}_stage_run$SUF.log
python $R/tools/stage_trace.py summary $D/t_results.db --top 20 > $R/gpurun_out/${TAG}_aux${SUF}_kernel_trace.md < /dev/null
rm -rf $D
