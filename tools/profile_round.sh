# Collects the judged evidence of the current build on the GPU box:  bash tools/profile_round.sh <name>
#   gpurun_out/<name>/...; copy into profiles/<name>/ afterwards.
set -e
NAME=${1:-prof}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# the kernel trace profiles the DRIVER's bench command (--steps 20 --warmup 5; minus its own rocprofv3 child runs and the CPU baseline),
# so that its averages can be held against the HIP-event figures of the same process
timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o x -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile > $OUT/trace.log 2>&1
cd $R
cp $OUT/trace/x_kernel_stats.csv $OUT/kernel_stats.csv
python3 tools/trace_agreement.py $OUT/trace/x_kernel_trace.csv $OUT/trace.log > $OUT/trace_vs_events.txt
rm -rf $OUT/trace
timeout -k 10 400 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > $OUT/bench_driver_config.json 2> $OUT/bench_driver_config.err
for w in cfg2 cluster tree dragon871k; do timeout -k 10 400 python3 bench.py --workload $w > $OUT/bench_$w.json 2> $OUT/bench_$w.err; done
timeout -k 10 400 python3 bench.py --frame 3840x2160 --steps 16 --warmup 4 > $OUT/bench_4k.json 2> $OUT/bench_4k.err
timeout -k 10 400 python3 bench.py --steps 1 --warmup 2 --no-cpu-baseline --no-profile > $OUT/bench_steps1.json 2> $OUT/bench_steps1.err
python3 tools/tile_overhead.py > $OUT/tile_scaling_one_gpu.txt 2>&1
TILE_K=20 python3 tools/tile_overhead.py > $OUT/tile_scaling_one_gpu_k20.txt 2>&1
# BASELINE.json's cfg5 as far as one GPU can show it: a rank's share of the 7680x4320 frame among 8 ranks, 256 passes, against the whole frame
TILE_FRAME=7680x4320 TILE_K=256 TILE_N=1,8 TILE_REPS=2 python3 tools/tile_overhead.py > $OUT/tile_scaling_one_gpu_cfg5.txt 2>&1
python3 tools/time_direct.py > $OUT/direct_lighting.txt 2>&1
(python3 tools/bvh_build_time.py scene_d; python3 tools/bvh_build_time.py big; for s in scene_d big cluster tree; do python3 tools/setprims_time.py $s; done) > $OUT/bvh_build.txt 2>&1
timeout -k 10 200 tools/ubench/ubench > $OUT/ubench.txt 2>&1
python3 - <<PY
import json
for f in ["bench", "bench_driver_config", "bench_cfg2", "bench_cluster", "bench_tree", "bench_dragon871k", "bench_4k", "bench_steps1"]:
    d = json.loads(open("$OUT/%s.json" % f).read().strip().split("\n")[-1])
    r = d["roofline"]
    print("%-20s %8.1f Mrays/s (reference-defined %8.1f)  %.4f ms/step  single %s  contract frac %s (algorithmic B per launch / launch time / 8 TB/s)  useful lanes %s = valu %s x lane_util %s  L1 frac %s  traffic_frac %s  waves %s  cpu %s" % (
        f, d["value"], d["mrays_reference_defined_per_s"], d["ms_per_step"], d["ms_per_frame_single"], r.get("frac"), (r.get("lane_slots") or {}).get("frac"), r["valu_issue"].get("frac"),
        r["valu_issue"].get("lane_util"), r["l1_accesses"].get("frac"), r["hbm"].get("traffic_frac"), r.get("k_trace_wave_states"), (d.get("cpu_baseline") or {}).get("value")))
PY
cat $OUT/trace_vs_events.txt $OUT/tile_scaling_one_gpu.txt $OUT/bvh_build.txt | tail -30
