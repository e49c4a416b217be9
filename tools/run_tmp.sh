cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
exec < /dev/null
for i in 1 2 3; do timeout 900 python -m pytest tests/test_hip_wino.py -x -q -m gpu 2>&1 | grep -E 'passed|failed|AssertionError|assert |max \||^E ' | cut -c1-300; done > gpurun_out/t_patch.txt
