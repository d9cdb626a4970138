# The bench lines of a build without the rest of profile_round.sh:  bash tools/bench_lines.sh <name>   -> gpurun_out/<name>/bench*.json
set -e
NAME=${1:-lines}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$NAME
mkdir -p $OUT
cd $R
timeout -k 10 400 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > $OUT/bench_driver_config.json 2> $OUT/bench_driver_config.err
for w in cfg2 cluster tree dragon871k; do timeout -k 10 400 python3 bench.py --workload $w > $OUT/bench_$w.json 2> $OUT/bench_$w.err; done
timeout -k 10 400 python3 bench.py --frame 3840x2160 --steps 16 --warmup 4 > $OUT/bench_4k.json 2> $OUT/bench_4k.err
timeout -k 10 400 python3 bench.py --steps 1 --warmup 2 --no-cpu-baseline --no-profile > $OUT/bench_steps1.json 2> $OUT/bench_steps1.err
python3 - <<PY
import json
for f in ["bench", "bench_driver_config", "bench_cfg2", "bench_cluster", "bench_tree", "bench_dragon871k", "bench_4k", "bench_steps1"]:
    d = json.loads(open("$OUT/%s.json" % f).read().strip().split("\n")[-1])
    r = d["roofline"]
    print("%-20s %8.1f Mrays/s (reference-defined %8.1f)  %.4f ms/step  single %s  contract frac %s (algorithmic B per launch / launch time / 8 TB/s)  useful lanes %s = valu %s x lane_util %s  visits frac %s  L1 frac %s  traffic_frac %s  waves %s  cpu %s" % (
        f, d["value"], d["mrays_reference_defined_per_s"], d["ms_per_step"], d["ms_per_frame_single"], r.get("frac"), (r.get("lane_slots") or {}).get("frac"), r["valu_issue"].get("frac"),
        r["valu_issue"].get("lane_util"), r["node_visits"].get("frac"), r["l1_accesses"].get("frac"), r["hbm"].get("traffic_frac"), r.get("k_trace_wave_states"), (d.get("cpu_baseline") or {}).get("value")))
PY
