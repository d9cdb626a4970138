set -e
R=$GRAFT_REPO_ROOT
cd $R
one="GPUART_HIP_PLAN_RUN_PERCENT=100000 GPUART_HIP_BATCH_MPATHS=128 GPUART_HIP_LANE_BUDGET_MB=65536"
for k in 1 2 4 8 16; do
  echo "== K=$k mode 3 default: $(GPUART_MODE=3 timeout -k 10 120 python3 tools/run_passes.py $k 5 | sort | head -2 | tr '\n' ' ')"
  echo "   K=$k mode 0 default: $(GPUART_MODE=0 timeout -k 10 120 python3 tools/run_passes.py $k 5 | sort | head -2 | tr '\n' ' ')"
  for w in 8 12 16 20; do
    echo "   K=$k k_run single run, waves/CU $w: $(env $one GPUART_HIP_RUN_WAVES_PER_CU=$w timeout -k 10 120 python3 tools/run_passes.py $k 5 | sort | head -2 | tr '\n' ' ')"
  done
done
