cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
exec < /dev/null
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > gpurun_out/t_full3.txt
bash tools/profile_round.sh r06 > gpurun_out/profile_round_r06.log 2>&1
