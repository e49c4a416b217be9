#!/bin/bash
exec < /dev/null
cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_hip_sparse_ace.py -x -q -s -k "straight_edge and 416" 2>&1 | tail -25 > gpurun_out/t_416.log
cat gpurun_out/t_416.log
