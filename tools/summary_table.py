#!/usr/bin/env python3
"""The table at the head of profiles/<round>/summary.txt from the bench lines of tools/profile_round.sh / bench_lines.sh.
python3 tools/summary_table.py <dir with bench*.json>"""
import json
import os
import sys

d0 = sys.argv[1]
rows = [("(default: K = 64)", "bench"), ("--steps 20 --warmup 5", "bench_driver_config"), ("--workload cfg2", "bench_cfg2"), ("--workload cluster", "bench_cluster"),
        ("--workload tree", "bench_tree"), ("--workload dragon871k", "bench_dragon871k"), ("--frame 3840x2160", "bench_4k"), ("--steps 1", "bench_steps1")]
print("line (python bench.py ...)   ms/pass   Mrays/s  ref-def.  one pass   NF opt-in  contract      lane    visits       L1      HBM CPU port")
print("                        (median)  executed   Mrays/s  alone ms     ms/pass      frac     slots      frac     frac  traffic  Mrays/s")
f = lambda v, fmt: (fmt % v) if isinstance(v, (int, float)) else "        -"[-len(fmt % 0.0):]
ok = True
for name, fn in rows:
    p = os.path.join(d0, fn + ".json")
    if not os.path.exists(p):
        continue
    d = json.loads(open(p).read().strip().split("\n")[-1])
    r = d["roofline"]
    nf = (d.get("nearest_first_opt_in") or {})
    ok = ok and d.get("frames_verified") is True and (not nf or nf.get("same_frame_as_default") is True)
    print("%-22s %9.4f %9.1f %9.1f %9.3f   %9s %9s %9s %9s %8s %8s %8s" % (
        name, d["ms_per_step"], d["value"], d["mrays_reference_defined_per_s"], d["ms_per_frame_single"], f(nf.get("ms_per_step"), "%9.3f"), f(r.get("frac"), "%9.3f"),
        f((r.get("lane_slots") or {}).get("frac"), "%9.3f"), f((r.get("node_visits") or {}).get("frac"), "%9.3f"), f((r.get("l1_accesses") or {}).get("frac"), "%8.3f"),
        f((r.get("hbm") or {}).get("traffic_frac"), "%8.3f"), f((d.get("cpu_baseline") or {}).get("value"), "%8.3f")))
print()
print("every line: frames_verified = %s (the timed accumulator == the mode-1 and mode-4 replays, bit for bit); nearest_first_opt_in.same_frame_as_default = %s" % (ok, ok))
d = json.loads(open(os.path.join(d0, "bench_driver_config.json")).read().strip().split("\n")[-1])
r = d["roofline"]
pl = r.get("per_launch") or {}
print("contract frac = executed algorithmic bytes of one k_trace launch / its average duration (HIP events) / 8 TB/s; driver's line: %.0f %s; PMC HBM traffic per launch %.0f MB" % (r.get("achieved"), r.get("unit"), (r.get("traffic") or 0) / 1e6))
print("k_trace waves at the driver's line: %s" % (r.get("k_trace_wave_states"),))
ks = r.get("kernels") or []
print("kernels at the driver's line (ms summed per pass; launches overlap): " + ", ".join("%s %.3f" % (k.get("kernel"), k.get("ms_summed_per_pass", float("nan"))) for k in ks[:6]))
