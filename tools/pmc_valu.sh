# VALU instruction counts of the k_trace launches of 2 passes (serialised by the profiler):  bash tools/pmc_valu.sh <tag>
set -e
R=$GRAFT_REPO_ROOT; TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS -d $R/gpurun_out/pmc_$TAG -o x -- python3 $R/tools/run_passes.py ${2:-40} > $R/gpurun_out/pmc_$TAG.log 2>&1
cd $R && python3 tools/pmc_summary.py gpurun_out/pmc_$TAG | grep k_trace
