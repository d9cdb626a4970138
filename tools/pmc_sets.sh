# Collects SQ / TCP / TCC counters of the bench workload in separate rocprofv3 --pmc passes (one small set per pass:
# a set that asks for more than the hardware can collect at once aborts the profiler), keeping only counters the box
# offers. TA_* and TCP_GATE / TCP_*_STALL counters are not asked for: rocprofiler refuses them on gfx950 ("error code 38: Request exceeds
# the capabilities of the hardware to collect", even one at a time), aborts with signal 6 and then sits in its own signal handler until
# the timeout below kills it — the tool stalls, no kernel ever runs (profiles/r03/pmc_ta_tcp_sets_abort.txt).
#   bash tools/pmc_sets.sh <tag> [passes]     -> gpurun_out/pmc_<tag>/s<i>/..., summary in gpurun_out/pmc_<tag>.txt
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
  timeout -k 10 90 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $OUT/s$i -o x -- python3 $R/tools/run_passes.py $K >> $OUT/log.txt 2>&1 || { echo "set $i failed" >> $OUT/log.txt; exit 1; }
done <<SETS
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_BRANCH
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
GRBM_GUI_ACTIVE SQ_CYCLES SQ_BUSY_CU_CYCLES
SETS
cd $R && for d in $OUT/s*; do python3 tools/pmc_summary.py $d | grep -E "k_trace|k_shade|k_run|k_gen" ; done > gpurun_out/pmc_$TAG.txt 2>&1
python3 - <<PY >> gpurun_out/pmc_$TAG.txt
import csv,collections
d=collections.defaultdict(list)
for r in csv.DictReader(open("$OUT/s2/x_kernel_trace.csv")):
    n=r['Kernel_Name'].replace('(anonymous namespace)::','').split('(')[0]
    d[n].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
for n,v in d.items(): print('kernel-trace (serialised by --pmc)', n, len(v), 'launches, total %.3f ms'%(sum(v)/1e6))
PY
cat gpurun_out/pmc_$TAG.txt
