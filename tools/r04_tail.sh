cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04B
bash tools/profile_round.sh r04B/prof > gpurun_out/r04B/profile_round.log 2>&1; tail -45 gpurun_out/r04B/profile_round.log
