#!/bin/bash
# Hardware counters of one kernel of a stand-alone benchmark binary, a few counters per pass (run on the GPU box):
#   tools/pmc_counters.sh <kernel name substring> "<counters of pass 1>" ["<counters of pass 2>" ...] -- <command ...>
set -u
K=$1; shift
SETS=()
while [ "$1" != "--" ]; do SETS+=("$1"); shift; done
shift
cd /tmp && export TMPDIR=/tmp
for set in "${SETS[@]}"; do
  rm -rf /tmp/pm
  rocprofv3 --pmc $set --kernel-trace -d /tmp/pm -o t -- "$@" > /dev/null 2>&1
  python3 - "$K" <<'PY'
import sqlite3, sys
c = sqlite3.connect('/tmp/pm/t_results.db')
for r in c.execute("select counter_name, count(*), avg(value), avg(duration) from counters_collection where kernel_name like ? group by counter_name order by counter_name", ('%' + sys.argv[1] + '%',)):
    print(f'{r[0]:36s} launches {r[1]:4d}  avg {r[2]:16.1f}  avg duration {r[3] / 1e3:9.1f} us')
PY
done
rm -rf /tmp/pm
