cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04b
O=gpurun_out/r04b
(timeout -k 10 300 python3 tests/fuzz_parity.py --lattice 0 1500 > $O/fuzz_lattice.txt 2>&1; tail -3 $O/fuzz_lattice.txt)
for w in lattice box; do timeout -k 10 200 python3 tools/order_soak.py $w --passes 16 --chunks 4 > $O/soak_$w.txt 2>&1; tail -2 $O/soak_$w.txt; done
timeout -k 10 300 python3 tools/order_soak.py cfg3 --passes 64 --chunks 32 > $O/soak_cfg3.txt 2>&1; tail -1 $O/soak_cfg3.txt
timeout -k 10 300 python3 tools/order_soak.py cfg3 --passes 64 --chunks 8 --mode 5 > $O/soak_cfg3_m5.txt 2>&1; tail -1 $O/soak_cfg3_m5.txt
for w in cfg2 tree cluster dragon871k; do timeout -k 10 300 python3 tools/order_soak.py $w --passes 64 --chunks 8 > $O/soak_$w.txt 2>&1; tail -1 $O/soak_$w.txt; done
