cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04A
O=gpurun_out/r04A
(timeout -k 10 1000 python -m pytest tests -m gpu -x -q 2>&1 | tail -5) > $O/tests.txt 2>&1
tail -3 $O/tests.txt
python3 tools/ab.py -k 64 -r 4 withmax default > $O/ab_k64.txt 2>&1; cat $O/ab_k64.txt
python3 tools/ab.py -k 1 -r 5 withmax default > $O/ab_k1.txt 2>&1; cat $O/ab_k1.txt
timeout -k 10 300 python3 tools/order_soak.py cfg3 --passes 64 --chunks 48 > $O/soak_cfg3.txt 2>&1; tail -n 1 $O/soak_cfg3.txt
timeout -k 10 200 python3 tools/order_soak.py cfg3 --passes 64 --chunks 16 --mode 5 > $O/soak_cfg3_m5.txt 2>&1; tail -n 1 $O/soak_cfg3_m5.txt
for w in tree cfg2 box lattice; do timeout -k 10 300 python3 tools/order_soak.py $w --passes 64 --chunks 24 > $O/soak_$w.txt 2>&1; tail -n 1 $O/soak_$w.txt; done
for w in cluster dragon871k; do timeout -k 10 400 python3 tools/order_soak.py $w --passes 64 --chunks 12 > $O/soak_$w.txt 2>&1; tail -n 1 $O/soak_$w.txt; done
(timeout -k 10 400 python3 tests/fuzz_parity.py --lattice 800000 4000 > $O/fuzz_lattice.txt 2>&1; tail -n 1 $O/fuzz_lattice.txt)
