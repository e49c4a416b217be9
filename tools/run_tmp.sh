cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
exec < /dev/null
timeout 900 python -m pytest tests/test_hip_sean_generator.py -x -q -m gpu -k "bf16 or f16 or single_term or reduced" 2>&1 | tail -8 > gpurun_out/t_bf16.txt
timeout 400 python bench.py --only-headline --no-cpu-baseline --steps 10 --path bf16 --batch 32 > gpurun_out/b_bf16_s1.json 2> gpurun_out/b_bf16_s1.err
timeout 400 python bench.py --only-headline --no-cpu-baseline --steps 10 --path bf16 --batch 32 --dbg 268435456 > gpurun_out/b_bf16_s0.json 2> gpurun_out/b_bf16_s0.err
for f in b_bf16_s1 b_bf16_s0; do timeout 20 python tools/bench_brief.py $f < gpurun_out/$f.json; done > gpurun_out/b_bf16_brief2.txt 2>&1
