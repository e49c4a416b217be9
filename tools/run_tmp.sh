cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
exec < /dev/null
timeout 600 python tools/edge_debug.py 2>&1 | grep 'max diff\|bad pixels' > gpurun_out/edge_debug.txt
timeout 1200 python -m pytest tests/test_hip_wino.py tests/test_hip_sparse_ace.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/t_edge.txt
timeout 900 python -m pytest tests/test_hip_sean_generator.py -x -q -m gpu -k "golden or stagewise or batch_invariant or alternating" 2>&1 | tail -8 >> gpurun_out/t_edge.txt
