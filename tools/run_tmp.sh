cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
exec < /dev/null
cp ctrlhair_amd/libctrlhair_hip.so /tmp/lib_u4.so
for U in u4 u2 u1 u4 u2 u1; do
  if [ $U = u4 ]; then cp /tmp/lib_u4.so ctrlhair_amd/libctrlhair_hip.so; else cp ctrlhair_amd/libctrlhair_$U.so ctrlhair_amd/libctrlhair_hip.so; fi
  timeout 300 python bench.py --only-headline --no-cpu-baseline --steps 20 > gpurun_out/b_$U.json 2> gpurun_out/b_$U.err
  timeout 20 python tools/bench_brief.py $U < gpurun_out/b_$U.json
done > gpurun_out/b_unroll_brief.txt 2>&1
cp /tmp/lib_u4.so ctrlhair_amd/libctrlhair_hip.so
