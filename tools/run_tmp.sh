cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
exec < /dev/null
timeout 600 python -m pytest tests/test_hip_sparse_ace.py tests/test_hip_wino.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/t_int2.txt
timeout 400 python bench.py --only-headline --no-cpu-baseline --steps 20 > gpurun_out/b_two.json 2> gpurun_out/b_two.err
timeout 400 python bench.py --only-headline --no-cpu-baseline --steps 20 --labels face > gpurun_out/b_two_face.json 2> gpurun_out/b_two_face.err
timeout 400 python bench.py --only-headline --no-cpu-baseline --steps 10 --workload pipeline --path f32 > gpurun_out/b_two_pipe.json 2> gpurun_out/b_two_pipe.err
for f in b_two b_two_face; do timeout 20 python tools/bench_brief.py $f < gpurun_out/$f.json; done > gpurun_out/b_two_brief.txt 2>&1
