cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
exec < /dev/null
for r in 1 2; do for a in wino4_bench w4v_nocarry w4v_rb8 w4v_rb2; do echo "== $a run $r"; timeout 200 tools/$a.bin | grep 'V route' | sed -n '6,10p' | cut -c14-150; done; done > gpurun_out/w4v_tune.txt 2>&1
