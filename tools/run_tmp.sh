#!/bin/bash
# scratch script for gpurun calls (always: stdin closed, every command under timeout)
exec < /dev/null
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 300 python bench.py --only-headline --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python tools/bench_brief.py
