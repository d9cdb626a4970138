cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04z
O=gpurun_out/r04z
TILE_K=20 python3 tools/tile_overhead.py > $O/tile_scaling_one_gpu_k20.txt 2>&1; cat $O/tile_scaling_one_gpu_k20.txt
python3 tools/tile_overhead.py > $O/tile_scaling_one_gpu.txt 2>&1; cat $O/tile_scaling_one_gpu.txt
