#!/bin/bash
exec < /dev/null
cd /root/repo
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_hip_sean_generator.py -x -q -k "graph_replay" 2>&1 | tail -15 > gpurun_out/t_graph.log
cat gpurun_out/t_graph.log
export TMPDIR=/tmp
D=/tmp/prof_bf16; rm -rf $D; mkdir -p $D
timeout 400 rocprofv3 --kernel-trace --stats -d $D -o t -- python bench.py --only-headline --path bf16 --batch 32 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bf16_trace_run.log 2>&1
timeout 120 python tools/rocprof_summary.py $D/t_results.db > gpurun_out/bf16_b32_kernel_trace.md 2>> gpurun_out/bf16_trace_run.log
head -30 gpurun_out/bf16_b32_kernel_trace.md
