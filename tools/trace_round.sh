# The kernel-trace part of profile_round.sh alone:  bash tools/trace_round.sh <name>  -> gpurun_out/<name>/kernel_stats.csv, trace_vs_events.txt
set -e
NAME=${1:-trace}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o x -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile > $OUT/trace.log 2>&1
cd $R
cp $OUT/trace/x_kernel_stats.csv $OUT/kernel_stats.csv
python3 tools/trace_agreement.py $OUT/trace/x_kernel_trace.csv $OUT/trace.log > $OUT/trace_vs_events.txt
rm -rf $OUT/trace
cat $OUT/trace_vs_events.txt
