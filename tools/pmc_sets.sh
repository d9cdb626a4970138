# Collects SQ / TA / TCP / TCC counters of the bench workload in separate rocprofv3 --pmc passes (one small set per pass:
# a set that asks for more than the hardware can collect at once aborts the profiler), keeping only counters the box
# offers.   bash tools/pmc_sets.sh <tag> [passes]     -> gpurun_out/pmc_<tag>/<set>/..., summary in gpurun_out/pmc_<tag>.txt
R=$GRAFT_REPO_ROOT; TAG=${1:-x}; K=${2:-8}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_$TAG; mkdir -p $OUT
rocprofv3 --list-avail > $OUT/avail.txt 2>&1 || true
pick() { for c in "$@"; do if grep -qw -- "$c" $OUT/avail.txt; then printf "%s " "$c"; fi; done; }
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  set -- $(pick $line)
  [ $# -eq 0 ] && continue
  echo "== set $i: $*" >> $OUT/log.txt
  timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $OUT/s$i -o x -- python3 $R/tools/run_passes.py $K >> $OUT/log.txt 2>&1 || { echo "set $i failed" >> $OUT/log.txt; }
done <<SETS
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_BRANCH
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU
SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_IFETCH SQ_INSTS_VALU SQ_WAVE_CYCLES
TA_TA_BUSY_sum TA_BUSY_avr TA_BUFFER_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum
TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
GRBM_GUI_ACTIVE GRBM_COUNT SQ_CYCLES
SETS
cd $R && for d in $OUT/s*; do python3 tools/pmc_summary.py $d | grep -E "k_trace|k_shade|k_run" ; done > gpurun_out/pmc_$TAG.txt 2>&1
cat gpurun_out/pmc_$TAG.txt
