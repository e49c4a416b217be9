#!/bin/bash
# per-kernel time of one bench run: tools/kern_time.sh <tag> [bench args]  -> gpurun_out/kt_<tag>/
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/kt_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT -o t -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --only-headline $* > $OUT/log 2>&1
python $R/tools/rocprof_summary.py $OUT/t_results.db | head -${KT_LINES:-16}
