set -e
cd $GRAFT_REPO_ROOT
python3 tools/bvh_build_time.py scene_d
python3 tools/bvh_build_time.py big
for s in scene_d big cluster; do python3 tools/setprims_time.py $s; GPUART_BVH_THREADS=1 python3 tools/setprims_time.py $s; done
