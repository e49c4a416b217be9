cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
exec < /dev/null
timeout 1200 python -m pytest tests/test_hip_wino.py tests/test_hip_sparse_ace.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/t_edge.txt
timeout 400 python bench.py --only-headline --no-cpu-baseline --steps 20 > gpurun_out/b_edge1.json 2> gpurun_out/b_edge1.err
timeout 400 python bench.py --only-headline --no-cpu-baseline --steps 20 --labels face > gpurun_out/b_edge1_face.json 2> gpurun_out/b_edge1_face.err
timeout 400 python bench.py --only-headline --no-cpu-baseline --steps 20 --opt sean.edge=0 > gpurun_out/b_edge0.json 2> gpurun_out/b_edge0.err
for f in b_edge1 b_edge1_face b_edge0; do timeout 20 python tools/bench_brief.py $f < gpurun_out/$f.json; done > gpurun_out/b_edge_brief.txt 2>&1
