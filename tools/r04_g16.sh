cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04p
O=gpurun_out/r04p
(timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "frames_vs_reference or full_size_frame or scheduling_knobs" 2>&1 | tail -3) > $O/tests_subset.txt 2>&1; cat $O/tests_subset.txt
WORKLOAD=dragon871k python3 tools/knob_ab.py -k 64 -r 4 "XCD_QUEUES=0" "XCD_QUEUES=1" > $O/xcd_ab_dragon871k.txt 2>&1; cat $O/xcd_ab_dragon871k.txt
WORKLOAD=cfg3 python3 tools/knob_ab.py -k 64 -r 3 "XCD_QUEUES=0" "XCD_QUEUES=1" > $O/xcd_ab_cfg3.txt 2>&1; cat $O/xcd_ab_cfg3.txt
timeout -k 10 900 python bench.py --workload dragon871k > $O/bench_dragon871k.json 2> $O/bench_dragon871k.err; tail -c 600 $O/bench_dragon871k.json
