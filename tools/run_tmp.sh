cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
exec < /dev/null
timeout 1500 python -m pytest tests -x -q -m gpu --durations=8 2>&1 | tail -25 > gpurun_out/t_full1.txt
