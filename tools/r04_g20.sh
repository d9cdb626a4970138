cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04v
O=gpurun_out/r04v
(timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=6 2>&1 | tail -14) > $O/tests.txt 2>&1
tail -12 $O/tests.txt
python3 tools/ab.py -k 64 -r 4 nofuse default > $O/ab_k64.txt 2>&1; cat $O/ab_k64.txt
python3 tools/ab.py -k 20 -r 4 nofuse default > $O/ab_k20.txt 2>&1; cat $O/ab_k20.txt
python3 tools/ab.py -k 4 -r 4 nofuse default > $O/ab_k4.txt 2>&1; cat $O/ab_k4.txt
WORKLOAD=cfg2 python3 tools/ab.py -k 64 -r 3 nofuse default > $O/ab_cfg2.txt 2>&1; cat $O/ab_cfg2.txt
