cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
exec < /dev/null
( time timeout 1500 python bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err ) 2> gpurun_out/r06_bench_default.time
