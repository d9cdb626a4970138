cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT
run() { # mux waves
  GPUART_HIP_MUX=$1 GPUART_HIP_WAVES_PER_CU=$2 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES -d $R/gpurun_out/pmc_m$1_w$2 -o x -- python3 $R/tools/run_passes.py 2 > $R/gpurun_out/pmc_m$1_w$2.log 2>&1
}
run 0 1 && run 0 2 && run 0 4 && run 0 6 && run 0 12 && run 1 1 && run 1 2 && run 1 3 && run 1 6 && cd $R && for d in m0_w1 m0_w2 m0_w4 m0_w6 m0_w12 m1_w1 m1_w2 m1_w3 m1_w6; do echo "== $d"; python3 tools/pmc_summary.py gpurun_out/pmc_$d | grep k_trace; done
