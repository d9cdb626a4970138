cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04c
O=gpurun_out/r04c
(timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > $O/tests.txt 2>&1
tail -4 $O/tests.txt
(timeout -k 10 300 python3 tests/fuzz_parity.py --lattice 0 1500 > $O/fuzz_lattice.txt 2>&1; tail -3 $O/fuzz_lattice.txt)
for w in lattice box; do timeout -k 10 200 python3 tools/order_soak.py $w --passes 16 --chunks 4 > $O/soak_$w.txt 2>&1; tail -2 $O/soak_$w.txt; done
timeout -k 10 300 python3 tools/order_soak.py cfg3 --passes 64 --chunks 32 > $O/soak_cfg3.txt 2>&1; tail -1 $O/soak_cfg3.txt
timeout -k 10 300 python3 tools/order_soak.py cfg3 --passes 64 --chunks 8 --mode 5 > $O/soak_cfg3_m5.txt 2>&1; tail -1 $O/soak_cfg3_m5.txt
for w in cfg2 tree cluster dragon871k; do timeout -k 10 300 python3 tools/order_soak.py $w --passes 64 --chunks 8 > $O/soak_$w.txt 2>&1; tail -1 $O/soak_$w.txt; done
python3 tools/ab.py -k 64 -r 4 base default > $O/ab_k64.txt 2>&1; cat $O/ab_k64.txt
python3 tools/ab.py -k 1 -r 5 base default > $O/ab_k1.txt 2>&1; cat $O/ab_k1.txt
