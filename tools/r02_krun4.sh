set -e
cd $GRAFT_REPO_ROOT
big="GPUART_HIP_BATCH_MPATHS=128 GPUART_HIP_PLAN_RUN_PERCENT=100000 GPUART_HIP_LANE_BUDGET_MB=65536"
for rf in 8 16 24 32 48; do
  echo "refill_lanes $rf: $(env $big GPUART_HIP_REFILL_LANES=$rf timeout -k 10 120 python3 tools/run_passes.py 64 2 | sort | head -1)"
done
for ll in 8 24 32; do
  echo "leaf_lanes $ll: $(env $big GPUART_HIP_LEAF_LANES=$ll timeout -k 10 120 python3 tools/run_passes.py 64 2 | sort | head -1)"
done
for ls in 2 4 6; do
  echo "leaf_share $ls: $(env $big GPUART_HIP_LEAF_SHARE=$ls timeout -k 10 120 python3 tools/run_passes.py 64 2 | sort | head -1)"
done
