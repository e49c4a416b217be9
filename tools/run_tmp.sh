cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2; do
 echo "== halo64 run $i"; timeout 300 tools/wino4_bench.bin | cut -c1-140
 echo "== halo32 run $i"; timeout 300 tools/w4_halo32.bin | cut -c1-140
done > gpurun_out/w4_halo.txt 2>&1
timeout 900 python -m pytest tests/test_hip_sean_generator.py -x -q -m gpu -k "alternating or golden_batch16_full_size" 2>&1 | tail -15 > gpurun_out/t_alt.txt
timeout 600 python -m pytest tests/test_hip_wino.py -x -q -m gpu 2>&1 | tail -5 >> gpurun_out/t_alt.txt
