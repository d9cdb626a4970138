cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04f
O=gpurun_out/r04f
timeout -k 10 420 python3 tools/order_rays.py tree 400 $O/rays_tree.npz > $O/rays_tree.txt 2>&1; tail -5 $O/rays_tree.txt
timeout -k 10 420 python3 tools/order_rays.py cfg3 400 $O/rays_cfg3.npz > $O/rays_cfg3.txt 2>&1; tail -5 $O/rays_cfg3.txt
