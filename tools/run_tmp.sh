#!/bin/bash
exec < /dev/null
cd /root/repo
timeout 900 python -m pytest tests/test_backend.py -x -q -m gpu 2>&1 | tail -4
