cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_wino.py -x -q -m gpu -k "pretransformed or f4x4" 2>&1 | tail -15 > gpurun_out/t_v.txt
timeout 600 python bench.py --only-headline --no-cpu-baseline --steps 20 > gpurun_out/b_v1.json 2> gpurun_out/b_v1.err
timeout 600 python bench.py --only-headline --no-cpu-baseline --steps 20 --opt sean.wino4v=0 > gpurun_out/b_v0.json 2> gpurun_out/b_v0.err
