#!/bin/bash
# PMC passes over a stand-alone kernel driver, summarised on the box: tools/pmc_kernel.sh <binary + args> -- <kernel name substring>
# (one pass per counter group: `rocprofv3 --pmc` alone, no trace domains besides the kernel trace)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
CMD=$1; KN=$2
OUT=$R/gpurun_out/pmc_kernel
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -oE "\bSQ_[A-Z0-9_]+" | sort -u > $OUT/avail_sq.txt
i=0
while read -r G; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $G --kernel-trace --output-format csv -d $OUT/p$i -o t -- $CMD > $OUT/p$i.log 2>&1 || echo "pass $i failed: $G"
  F=$(find $OUT/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$F" ] && python3 - "$F" "$KN" <<'PY'
import csv,sys,collections
f,kn=sys.argv[1],sys.argv[2]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(f)):
    if kn in r['Kernel_Name']:
        acc[r['Counter_Name']][r['Dispatch_Id']]+=float(r['Counter_Value'])
for c,d in acc.items():
    v=sorted(d.values()); print(f"{c:36s} launches {len(v):4d}  sum {sum(v):.4g}  max-launch {v[-1]:.4g}")
PY
done <<'GROUPS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC
SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INST_CYCLES_VMEM
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL
SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_F32
SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_CMD_FIFO_FULL SQ_IFETCH
GROUPS
