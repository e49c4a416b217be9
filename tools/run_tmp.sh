cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for a in 64 128 256; do echo "== W4V_ABL=$a"; timeout 200 tools/w4v_abl$a.bin | grep 'V route' | sed -n '7,9p' | cut -c40-150; done > gpurun_out/w4v_abl3.txt 2>&1
