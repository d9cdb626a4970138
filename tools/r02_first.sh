set -e
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02_pytest1.log 2>&1 || { tail -30 gpurun_out/r02_pytest1.log; exit 1; }
tail -3 gpurun_out/r02_pytest1.log
timeout -k 10 120 tools/ubench/ubench json > gpurun_out/r02_ubench.txt 2>&1
cat gpurun_out/r02_ubench.txt | head -60
bash tools/pmc_sets.sh r02a 8
