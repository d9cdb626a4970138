# VALU instruction count and lane utilisation of k_trace against the size of its persistent grid (one pass per run, one
# run at a time: the profiler serialises launches) — the data behind profiles/r01_v7/pmc_valu_vs_grid.txt.
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT
run() {
  GPUART_HIP_WAVES_PER_CU=$1 GPUART_HIP_BATCH_MPATHS=1 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES -d $R/gpurun_out/pmc_w$1 -o x -- python3 $R/tools/run_passes.py 2 > $R/gpurun_out/pmc_w$1.log 2>&1
}
run 1 && run 2 && run 4 && run 8 && run 16 && cd $R && for w in 1 2 4 8 16; do echo "== WAVES_PER_CU=$w"; python3 tools/pmc_summary.py gpurun_out/pmc_w$w | grep k_trace; done
