cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04z
O=gpurun_out/r04z
(timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=5 2>&1 | tail -12) > $O/tests.txt 2>&1
tail -10 $O/tests.txt
WORKLOAD=cfg2 python3 tools/ab.py -k 64 -r 4 base default > $O/cfg2_ab.txt 2>&1; cat $O/cfg2_ab.txt
echo "nearest-first forced on cfg2:"; GPUART_HIP_NEAREST_MIN_PRIMS=0 WORKLOAD=cfg2 python3 tools/run_passes.py 64 3 | tail -1
timeout -k 10 600 python bench.py --workload cfg2 > $O/bench_cfg2.json 2> $O/bench_cfg2.err; python3 -c "
import json; d=json.loads(open('$O/bench_cfg2.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['ms_per_frame_single'], d['rewalked_queries_per_step'])"
