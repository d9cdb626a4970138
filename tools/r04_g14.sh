cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04n
O=gpurun_out/r04n
TILE_K=20 python3 tools/tile_overhead.py > $O/tile_scaling_one_gpu_k20.txt 2>&1; cat $O/tile_scaling_one_gpu_k20.txt
python3 tools/tile_overhead.py > $O/tile_scaling_one_gpu.txt 2>&1; cat $O/tile_scaling_one_gpu.txt
GPUART_HIP_LEAF_LANES=24 TILE_K=20 python3 tools/tile_overhead.py > $O/tile_scaling_one_gpu_k20_leaf24.txt 2>&1; cat $O/tile_scaling_one_gpu_k20_leaf24.txt
