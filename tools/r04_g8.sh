cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04h
O=gpurun_out/r04h
(timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > $O/tests.txt 2>&1
tail -3 $O/tests.txt
for w in cluster cfg2 tree box; do timeout -k 10 400 python3 tools/order_soak.py $w --passes 64 --chunks 16 > $O/soak_$w.txt 2>&1; tail -1 $O/soak_$w.txt; done
timeout -k 10 300 python3 tools/order_soak.py cfg3 --passes 64 --chunks 32 > $O/soak_cfg3.txt 2>&1; tail -1 $O/soak_cfg3.txt
timeout -k 10 300 python3 tools/order_soak.py cfg3 --passes 64 --chunks 16 --mode 5 > $O/soak_cfg3_m5.txt 2>&1; tail -1 $O/soak_cfg3_m5.txt
timeout -k 10 300 python3 tools/order_soak.py dragon871k --passes 64 --chunks 8 > $O/soak_dragon871k.txt 2>&1; tail -1 $O/soak_dragon871k.txt
(timeout -k 10 600 python3 tests/fuzz_parity.py 100000 7000 > $O/fuzz_plain.txt 2>&1; tail -1 $O/fuzz_plain.txt)
(timeout -k 10 600 python3 tests/fuzz_parity.py --wild 100000 7000 > $O/fuzz_wild.txt 2>&1; tail -1 $O/fuzz_wild.txt)
(timeout -k 10 600 python3 tests/fuzz_parity.py --wild2 100000 5000 > $O/fuzz_wild2.txt 2>&1; tail -1 $O/fuzz_wild2.txt)
(timeout -k 10 600 python3 tests/fuzz_parity.py --lattice 100000 5000 > $O/fuzz_lattice.txt 2>&1; tail -1 $O/fuzz_lattice.txt)
python3 tools/ab.py -k 64 -r 3 base default > $O/ab_k64.txt 2>&1; cat $O/ab_k64.txt
