# Counters of ONE k_run launch (a single pass observed alone, K = 1): instruction cache, wave wait states, instruction mix.
#   bash tools/pmc_krun.sh <tag> [libdir-variant]
R=$GRAFT_REPO_ROOT; TAG=${1:-x}
[ -n "$2" ] && export GPUART_LIBDIR=$R/gpuart_amd/lib_ab/$2
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
  timeout -k 10 90 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $OUT/s$i -o x -- python3 $R/tools/run_passes.py 1 3 >> $OUT/log.txt 2>&1 || { echo "set $i failed" >> $OUT/log.txt; exit 1; }
done <<SETS
SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES
SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_IFETCH
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_BRANCH
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU
SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_I8
SETS
cd $R && for d in $OUT/s*; do python3 tools/pmc_summary.py $d | grep -E "k_run" ; done > gpurun_out/pmc_$TAG.txt 2>&1
cat gpurun_out/pmc_$TAG.txt
