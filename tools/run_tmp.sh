#!/bin/bash
exec < /dev/null
cd /root/repo
timeout 900 python -m pytest tests/test_backend.py tests/test_pipeline.py tests/test_bench.py -x -q -m gpu 2>&1 | tail -6
