cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04o
O=gpurun_out/r04o
SWEEP_K=64 python3 tools/sweep.py LEAF_LANES=16,20,24,28 CHUNK=128,256 > $O/sweep_k64.txt 2>&1; grep ms/pass $O/sweep_k64.txt
SWEEP_K=64 python3 tools/sweep.py LEAF_LANES=16,20,24,28 CHUNK=128,256 > $O/sweep_k64b.txt 2>&1; grep ms/pass $O/sweep_k64b.txt
SWEEP_K=20 python3 tools/sweep.py LEAF_LANES=16,20,24,28 CHUNK=128,256 > $O/sweep_k20.txt 2>&1; grep ms/pass $O/sweep_k20.txt
