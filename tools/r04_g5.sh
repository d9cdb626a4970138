cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04e
O=gpurun_out/r04e
for v in band10 band8 band6; do
export GPUART_LIBDIR=$GRAFT_REPO_ROOT/gpuart_amd/lib_ab/$v
timeout -k 10 300 python3 tools/order_soak.py cfg3 --passes 64 --chunks 32 > $O/soak_${v}_cfg3.txt 2>&1; echo $v; tail -1 $O/soak_${v}_cfg3.txt
timeout -k 10 300 python3 tools/order_soak.py tree --passes 64 --chunks 8 > $O/soak_${v}_tree.txt 2>&1; tail -1 $O/soak_${v}_tree.txt
done
unset GPUART_LIBDIR
python3 tools/ab.py -k 64 -r 3 base default band10 band8 band6 > $O/ab_k64.txt 2>&1; cat $O/ab_k64.txt
