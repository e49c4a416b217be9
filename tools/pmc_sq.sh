#!/bin/bash
# SQ stall breakdown of one bench run: tools/pmc_sq.sh <tag> [bench args]  -> gpurun_out/sq_<tag>/
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/sq_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace -d $OUT/a -o t -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --only-headline $* > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --kernel-trace -d $OUT/b -o t -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --only-headline $* > $OUT/b.log 2>&1
tail -1 $OUT/a.log | cut -c1-120; tail -1 $OUT/b.log | cut -c1-120
