cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04m
O=gpurun_out/r04m
export SWEEP_K=64
python3 tools/sweep.py REFILL_LANES=8,16,24,32 LEAF_LANES=8,16,24,32 > $O/sweep_refill_leaf.txt 2>&1; tail -20 $O/sweep_refill_leaf.txt
python3 tools/sweep.py WAVES_PER_CU=6,8,10,12,16 CHUNK=64,128,256 > $O/sweep_waves_chunk.txt 2>&1; tail -18 $O/sweep_waves_chunk.txt
python3 tools/sweep.py LEAF_SHARE=2,3,4,6 > $O/sweep_leaf_share.txt 2>&1; tail -6 $O/sweep_leaf_share.txt
