cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04u
O=gpurun_out/r04u
(timeout -k 10 1000 python -m pytest tests -m gpu -x -q 2>&1 | tail -6) > $O/tests.txt 2>&1
tail -4 $O/tests.txt
bash tools/profile_round.sh r04u/prof > $O/profile_round.log 2>&1; tail -40 $O/profile_round.log
