cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04s
O=gpurun_out/r04s
timeout -k 10 300 python3 tools/order_soak.py cfg3 --passes 64 --chunks 128 > $O/soak_cfg3.txt 2>&1; tail -1 $O/soak_cfg3.txt
timeout -k 10 200 python3 tools/order_soak.py cfg3 --frame 3840x2160 --passes 64 --chunks 8 > $O/soak_cfg3_4k.txt 2>&1; tail -1 $O/soak_cfg3_4k.txt
timeout -k 10 200 python3 tools/order_soak.py cfg3 --passes 64 --chunks 32 --mode 5 > $O/soak_cfg3_m5.txt 2>&1; tail -1 $O/soak_cfg3_m5.txt
timeout -k 10 200 python3 tools/order_soak.py cfg3 --passes 3 --chunks 300 --mode 0 > $O/soak_cfg3_k3.txt 2>&1; tail -1 $O/soak_cfg3_k3.txt
for w in tree cfg2 box; do timeout -k 10 300 python3 tools/order_soak.py $w --passes 64 --chunks 64 > $O/soak_$w.txt 2>&1; tail -1 $O/soak_$w.txt; done
for w in cluster dragon871k; do timeout -k 10 400 python3 tools/order_soak.py $w --passes 64 --chunks 24 > $O/soak_$w.txt 2>&1; tail -1 $O/soak_$w.txt; done
timeout -k 10 300 python3 tools/order_soak.py lattice --passes 64 --chunks 16 > $O/soak_lattice.txt 2>&1; tail -1 $O/soak_lattice.txt
