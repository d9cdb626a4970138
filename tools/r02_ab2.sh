set -e
cd $GRAFT_REPO_ROOT
export GPUART_HIP_BATCH_MPATHS=128 GPUART_HIP_PLAN_RUN_PERCENT=100000 GPUART_HIP_LANE_BUDGET_MB=65536
python3 tools/ab.py -k 64 -r 2 default fifo
python3 tools/ab.py -k 1 -r 3 default fifo
