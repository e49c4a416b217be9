cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
exec < /dev/null
timeout 900 python -m pytest tests/test_hip_sean_generator.py -x -q -m gpu -k "batch_invariant" 2>&1 | tail -12 > gpurun_out/t_binv.txt
