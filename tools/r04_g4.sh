cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04d
O=gpurun_out/r04d
export GPUART_LIBDIR=$GRAFT_REPO_ROOT/gpuart_amd/lib_ab/base
timeout -k 10 300 python3 tools/order_soak.py cfg3 --passes 64 --chunks 32 > $O/soak_base_cfg3.txt 2>&1; tail -1 $O/soak_base_cfg3.txt
timeout -k 10 300 python3 tools/order_soak.py tree --passes 64 --chunks 8 > $O/soak_base_tree.txt 2>&1; tail -1 $O/soak_base_tree.txt
timeout -k 10 300 python3 tools/order_soak.py cfg3 --passes 64 --chunks 8 --mode 5 > $O/soak_base_cfg3_m5.txt 2>&1; tail -1 $O/soak_base_cfg3_m5.txt
