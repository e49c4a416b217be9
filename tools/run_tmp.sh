cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
exec < /dev/null
timeout 900 python -m pytest tests/test_hip_aux_models.py tests/test_pipeline.py tests/test_backend.py -x -q -m gpu 2>&1 | tail -12 > gpurun_out/t_ovl.txt
timeout 400 python bench.py --only-headline --no-cpu-baseline --steps 10 --workload pipeline --path f32 > gpurun_out/b_ovl_pipe.json 2> gpurun_out/b_ovl_pipe.err
timeout 400 python bench.py --only-headline --no-cpu-baseline --steps 10 --workload pipeline --path f16x3 > gpurun_out/b_ovl_pipe16.json 2> gpurun_out/b_ovl_pipe16.err
(timeout 300 python tools/edit_profile.py f32 | tail -3; timeout 300 python tools/edit_profile.py f16x3 | tail -2) > gpurun_out/edit_profile2.txt 2>&1
