#!/bin/bash
exec < /dev/null
cd /root/repo
mkdir -p gpurun_out
timeout 120 tools/wino_bench.bin pw 2>&1 | tail -8
timeout 600 python -m pytest tests/test_hip_wino.py -x -q -k "grouped or winograd_equals" 2>&1 | tail -3
timeout 300 python bench.py --only-headline --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python tools/bench_brief.py
