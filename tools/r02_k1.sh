set -e
cd $GRAFT_REPO_ROOT
for k in 1 2 4; do for w in 8 16 24; do
  echo "mode 3 K=$k WAVES_PER_CU=$w: $(GPUART_MODE=3 GPUART_HIP_WAVES_PER_CU=$w timeout -k 10 120 python3 tools/run_passes.py $k 6 | sort | head -2 | tr '\n' ' ')"
done; done
