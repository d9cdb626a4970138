cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04j
O=gpurun_out/r04j
python3 tools/ab.py -k 64 -r 4 base default noodd nomax nohits nocert > $O/ab_k64.txt 2>&1; cat $O/ab_k64.txt
python3 tools/ab.py -k 1 -r 4 base default noodd nocert > $O/ab_k1.txt 2>&1; cat $O/ab_k1.txt
(timeout -k 10 600 python3 tests/fuzz_parity.py --wild2 100000 5000 > $O/fuzz_wild2.txt 2>&1; tail -3 $O/fuzz_wild2.txt)
(timeout -k 10 600 python3 tests/fuzz_parity.py --wild 100000 7000 > $O/fuzz_wild.txt 2>&1; tail -3 $O/fuzz_wild.txt)
(timeout -k 10 600 python3 tests/fuzz_parity.py --wild2 300000 4000 > $O/fuzz_wild2b.txt 2>&1; tail -3 $O/fuzz_wild2b.txt)
(timeout -k 10 600 python3 tests/fuzz_parity.py --wild 300000 4000 > $O/fuzz_wildb.txt 2>&1; tail -3 $O/fuzz_wildb.txt)
