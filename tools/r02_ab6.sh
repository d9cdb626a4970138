set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python3 tools/ab.py -k 64 -r 3 base default
for p in 50 75 150 300; do echo "PLAN_RUN_PERCENT $p: $(GPUART_HIP_PLAN_RUN_PERCENT=$p python3 tools/run_passes.py 64 3 | sort | head -1)"; done
GPUART_MODE=5 GPUART_HIP_BATCH_MPATHS=128 GPUART_HIP_PLAN_RUN_PERCENT=100000 GPUART_HIP_LANE_BUDGET_MB=65536 python3 tools/run_passes.py 64 2 | sort | head -1
