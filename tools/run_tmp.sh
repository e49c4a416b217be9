cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
exec < /dev/null
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for P in f32 f16x3; do
D=$R/gpurun_out/prof_edit_$P
timeout 600 rocprofv3 --kernel-trace -d $D -o t -- python $R/tools/edit_trace.py run $P > $R/gpurun_out/edit_trace_$P.log 2>&1
timeout 100 python $R/tools/edit_trace.py summary $D/t_results.db > $R/gpurun_out/r06_edit_trace_$P.md 2>> $R/gpurun_out/edit_trace_$P.log
rm -rf $D
done
