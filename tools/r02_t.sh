set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02_pytest2.log 2>&1 || { tail -40 gpurun_out/r02_pytest2.log; exit 1; }
tail -3 gpurun_out/r02_pytest2.log
for k in 1 2 4 64; do echo "auto K=$k: $(timeout -k 10 120 python3 tools/run_passes.py $k 5 | sort | head -2 | tr '\n' ' ')"; done
