cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT
for mp in 2 4 8 16 32; do
  GPUART_HIP_BATCH_MPATHS=$mp GPUART_HIP_PLAN_RUN_PERCENT=1000 GPUART_HIP_LANE_BUDGET_MB=65536 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU -d $R/gpurun_out/pmc_b$mp -o x -- python3 $R/tools/run_passes.py 32 > $R/gpurun_out/pmc_b$mp.log 2>&1 || exit 1
done
cd $R && for mp in 2 4 8 16 32; do echo "== batch paths ${mp}M"; python3 tools/pmc_summary.py gpurun_out/pmc_b$mp | grep k_trace; done
