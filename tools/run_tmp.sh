cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
exec < /dev/null
timeout 900 python -m pytest tests/test_hip_aux_models.py tests/test_pipeline.py -x -q -m gpu 2>&1 | tail -12 > gpurun_out/t_lut.txt
timeout 400 python bench.py --only-headline --no-cpu-baseline --steps 10 --workload pipeline --path f32 > gpurun_out/b_lut_pipe.json 2> gpurun_out/b_lut_pipe.err
