cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
exec < /dev/null
timeout 900 python -m pytest tests/test_hip_aux_models.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/t_sk.txt
timeout 400 python bench.py --only-headline --no-cpu-baseline --steps 10 --workload pipeline --path f32 > gpurun_out/b_sk_pipe.json 2> gpurun_out/b_sk_pipe.err
