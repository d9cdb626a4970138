set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
big="GPUART_HIP_BATCH_MPATHS=128 GPUART_HIP_PLAN_RUN_PERCENT=1000 GPUART_HIP_LANE_BUDGET_MB=65536"
# timing: one run of 64 passes (128M paths) vs default run sizes, k_run at several grid sizes
for w in 8 12 20; do
  echo "k_run one 64-pass run, RUN_WAVES_PER_CU $w"; env $big GPUART_HIP_RUN_WAVES_PER_CU=$w timeout -k 10 120 python3 $R/tools/run_passes.py 64 3 | tail -1
done
echo "mode 3, one 64-pass run"; env $big GPUART_MODE=3 timeout -k 10 120 python3 $R/tools/run_passes.py 64 3 | tail -1
for m in 0 3; do
for big2 in "$big" "X=1"; do
OUT=$R/gpurun_out/pmc_long_m${m}_$(echo $big2 | cut -c1-3); mkdir -p $OUT
env $big2 GPUART_MODE=$m timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_WAIT_ANY -d $OUT -o x -- python3 $R/tools/run_passes.py 64 > $OUT/log.txt 2>&1
echo "== mode $m env $big2"; python3 $R/tools/pmc_summary.py $OUT | grep -E "k_run|k_trace|k_shade|k_gen"
done; done
