cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04q
O=gpurun_out/r04q
python3 tools/ab.py -k 1 -r 6 default sm16 sm48 td1 td4 > $O/ab_k1.txt 2>&1; cat $O/ab_k1.txt
python3 tools/ab.py -k 2 -r 4 default sm16 sm48 td1 td4 > $O/ab_k2.txt 2>&1; cat $O/ab_k2.txt
