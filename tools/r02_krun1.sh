set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 120 python __graft_entry__.py smoke > gpurun_out/krun_smoke.log 2>&1 || { tail -20 gpurun_out/krun_smoke.log; exit 1; }
tail -2 gpurun_out/krun_smoke.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/krun_pytest.log 2>&1 || { tail -40 gpurun_out/krun_pytest.log; exit 1; }
tail -3 gpurun_out/krun_pytest.log
for m in 3 0; do for k in 1 8 64; do echo "mode $m K $k"; GPUART_MODE=$m timeout -k 10 120 python tools/run_passes.py $k 4 | tail -2; done; done
for w in 12 16 20 24; do echo "RUN_WAVES_PER_CU $w"; GPUART_HIP_RUN_WAVES_PER_CU=$w timeout -k 10 120 python tools/run_passes.py 64 3 | tail -1; done
