cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04t
O=gpurun_out/r04t
python3 tools/ab.py -k 64 -r 4 prev pushlim > $O/ab_k64.txt 2>&1; cat $O/ab_k64.txt
python3 tools/ab.py -k 20 -r 4 prev pushlim > $O/ab_k20.txt 2>&1; cat $O/ab_k20.txt
python3 tools/ab.py -k 1 -r 6 prev pushlim > $O/ab_k1.txt 2>&1; cat $O/ab_k1.txt
WORKLOAD=dragon871k python3 tools/ab.py -k 64 -r 3 prev pushlim > $O/ab_dragon.txt 2>&1; cat $O/ab_dragon.txt
WORKLOAD=cfg2 python3 tools/ab.py -k 64 -r 3 prev pushlim > $O/ab_cfg2.txt 2>&1; cat $O/ab_cfg2.txt
