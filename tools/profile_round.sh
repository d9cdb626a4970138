# Collects the judged evidence of the current build on the GPU box:  bash tools/profile_round.sh <name>
#   gpurun_out/<name>/{kernel_stats.csv, bench.json, pmc_traffic.json}; copy into profiles/<name>/ afterwards.
set -e
NAME=${1:-prof}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$NAME
CMD="bench.py --steps 16 --warmup 8 --no-cpu-baseline"
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# the kernel trace profiles the DEFAULT bench command, so that its averages can be held against bench.json
timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o x -- python3 $R/bench.py --no-cpu-baseline > $OUT/trace.log 2>&1
timeout -k 10 250 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/fetch -o x -- python3 $R/$CMD > $OUT/fetch.log 2>&1
timeout -k 10 250 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/write -o x -- python3 $R/$CMD > $OUT/write.log 2>&1
cd $R
cp $OUT/trace/x_kernel_stats.csv $OUT/kernel_stats.csv
python3 tools/trace_agreement.py $OUT/trace/x_kernel_trace.csv $OUT/trace.log > $OUT/trace_vs_events.txt
python3 tools/pmc_traffic.py $OUT/fetch $OUT/write $OUT/pmc_traffic.json "python3 $CMD" > /dev/null
cp $OUT/pmc_traffic.json profiles/pmc_traffic.json
timeout -k 10 250 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/bench.json
head -8 $OUT/kernel_stats.csv
rm -rf $OUT/trace $OUT/fetch $OUT/write
